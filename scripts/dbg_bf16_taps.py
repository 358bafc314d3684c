"""Per-stage comparison of the HIP bf16 forward against the bf16-rounded oracle (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')]
import numpy as np, torch
from gssd import synth
from models.ssd_multiphase_custom_group import build_ssd
from oracle import gssd_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else 'gssd'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
NETS = {'gssd': (dict(), (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)),
        'gssdpp': (dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True),
                   (True, 4, 4, 1, True, True, True, 1, 4, True, False, 1))}
flags, args = NETS[name]
net = build_ssd('train', 300, 2, *args)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
net.load_state_dict(sd)
dev = torch.device('cuda:0')
net = net.to(dev).train()
net.compute_dtype = 'bf16'
x = synth.synth_images(B, seed=5)
with torch.no_grad():
    loc, conf, _ = net(x.to(dev))
    taps = {}
    lo, co, _ = O.gssd_forward(sd, x, taps=taps, bf16=True, **flags)
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm()), float(((a - b).abs() > 1e-6 * b.abs().max()).double().mean())
plan = net._engine._last_plan
nchw = lambda t: t.float().permute(0, 3, 1, 2)
print('loc', rel(loc, lo), 'conf', rel(conf, co))
for kind, r in plan.rec:
    if kind == 'convbn' and r['name'].startswith('vgg') and r.get('xf') is None:
        i = int(r['name'].split('.')[1])
        key = f'vgg.{i + 3}' if r['pool'] else f'vgg.{i + 2}'
        if key in taps and taps[key].shape == nchw(r['out']).shape:
            print(r['name'], '->', key, 'maxrel %.2e l2 %.2e frac-differing %.4f' % rel(nchw(r['out']), taps[key]))
    if kind == 'dcn':
        print('dcn out', rel(nchw(r['out']), taps['dcn0.out']))
    if kind == 'l2norm':
        print('l2norm', rel(nchw(r['out']), taps['l2norm']))
for i, (s, H, C) in enumerate(plan.sources):
    print('source', i, rel(nchw(s), taps[f'source{i}']))
