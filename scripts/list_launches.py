"""Every tagged launch of the GSSD++ forward plan at batch 32 with its shape (M x Cout x K, groups, map) and eager duration:
python scripts/list_launches.py [f32|bf16]   (GPU box)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')); sys.path.insert(0, ROOT)
import ctypes as C
import torch
from gssd import synth, _lib
from models.ssd_multiphase_custom_group import build_ssd
dev = torch.device('cuda:0')
net = build_ssd('train', 300, 2, True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)
net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
net = net.to(dev).train()
if len(sys.argv) > 1 and sys.argv[1] == 'bf16':
    net.compute_dtype = 'bf16'
x = synth.synth_images(32, seed=100).to(dev)
class EL(list):
    only = None
with torch.no_grad():
    for _ in range(3):
        net(x)
    ev = EL()
    net.__dict__['_events'] = ev
    net(x)
    torch.cuda.synchronize()
    net.__dict__['_events'] = None
plan = net._engine._last_plan
descs = {}
for st in plan.steps:
    if st.tag is not None and st.keep is not None:
        k = st.keep[0] if isinstance(st.keep, (tuple, list)) else st.keep
        if isinstance(k, _lib.ConvDesc):
            descs[id(st.tag)] = k
tot = 0.0
for tag, e0, e1 in ev:
    us = 1e3 * e0.elapsed_time(e1)
    tot += us
    d = descs.get(id(tag))
    shp = f'M {d.B * d.Ho * d.Wo:6d} ({d.Ho:3d}x{d.Wo:3d}) Cout {d.Cout:5d} g {d.groups} K {d.K:5d} k{d.KH} s{d.stride} mode {d.out_mode} splitk {d.split_k}' if d is not None else ''
    print(f'{us:8.1f} us  {tag[0]:34s} {getattr(tag, "layer", None) or "":18s} {shp}')
print(f'total {tot / 1e3:.3f} ms over {len(ev)} launches')
