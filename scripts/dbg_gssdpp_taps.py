"""Per-stage comparison of the HIP GSSD++ forward against the oracle at a given batch size (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')]
import numpy as np, torch
from gssd import synth
from models.ssd_multiphase_custom_group import build_ssd
from oracle import gssd_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
args = (True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)
flags = dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True)
net = build_ssd('train', 300, 2, *args)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
net.load_state_dict(sd)
dev = torch.device('cuda:0')
net = net.to(dev).train()
x = synth.synth_images(B, seed=seed)
torch.set_num_threads(16)
with torch.no_grad():
    loc, conf, _ = net(x.to(dev))
    taps = {}
    lo, co, _ = O.gssd_forward(sd, x, taps=taps, **flags)
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max())
plan = net._engine._last_plan
nchw = lambda t: t.permute(0, 3, 1, 2)
print('loc', rel(loc, lo), 'conf', rel(conf, co))
for kind, r in plan.rec:
    if kind == 'convbn' and r['name'].startswith('vgg') and r.get('xf') is None:
        i = int(r['name'].split('.')[1])
        key = f'vgg.{i + 3}' if r['pool'] else f'vgg.{i + 2}'
        if key in taps and taps[key].shape == nchw(r['out']).shape:
            print(r['name'], '->', key, rel(nchw(r['out']), taps[key]))
    if kind == 'sa':
        print('sa', r['H'], r['C'], 'out absmax', float(r['out'].abs().max()), 'S rowmax mean', float(r['S'][:, :, :r['N']].max(-1)[0].mean()))
    if kind == 'dcn':
        print('dcn out', rel(nchw(r['out']), taps['dcn0.out']), 'om', float(r['om'].abs().max()))
    if kind == 'l2norm':
        print('l2norm', rel(nchw(r['out']), taps['l2norm']))
print('sab0.out', rel(nchw([r for k, r in plan.rec if k == 'sa'][0]['out']), taps['sab0.out']))
for i, (s, H, C) in enumerate(plan.sources):
    print('source', i, rel(nchw(s), taps[f'source{i}']))
