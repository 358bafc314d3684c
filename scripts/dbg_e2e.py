import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
from oracle import gssd_oracle as O
from gssd import synth
from models.ssd_multiphase_custom_group import build_ssd
NETS = {
    'gssd': (dict(), (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)),
    'gssd_sa': (dict(use_self_attention=True, use_self_attention_base=True), (True, 4, 4, 1, True, True, True, 0, 1, False, False, 1)),
    'gssdpp': (dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True), (True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)),
}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
offs = np.cumsum([0, 38*38*4, 19*19*6, 100*6, 25*6, 9*4, 4])
for name, (flags, args) in NETS.items():
    net = build_ssd('train', 300, 2, *args)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
    net.load_state_dict(sd); net = net.cuda().train()
    x = synth.synth_images(B, seed=5)
    with torch.no_grad():
        loc, conf, _ = net(x.cuda())
        lo, co, upd = O.gssd_forward(sd, x, **flags)
    for nm, a, b in (('loc', loc.cpu(), lo), ('conf', conf.cpu(), co)):
        errs = []
        for i in range(6):
            d = (a[:, offs[i]:offs[i+1]] - b[:, offs[i]:offs[i+1]]).abs().max().item()
            errs.append(d / b.abs().max().item())
        print(name, nm, ' '.join(f'{e:.2e}' for e in errs))
