"""Forward error budget of one seed pair: HIP fp32 and the CPU fp32 oracle, each against the same graph in float64 (CPU oracle with double
weights / input).  usage: python3 scripts/dbg_fwd64.py [wseed xseed]"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import gssd_oracle as O
from gssd import synth
from gpu_common import NETS
from models.ssd_multiphase_custom_group import build_ssd
wseed, xseed = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (31337, 14)
flags, args = NETS['gssdpp']
net = build_ssd('train', 300, 2, *args)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
sd = synth.synth_state_dict(shapes, seed=wseed)
net.load_state_dict(sd)
dev = torch.device('cuda:0')
net = net.to(dev).train()
x = synth.synth_images(4, seed=xseed)
with torch.no_grad():
    loc, conf, _ = net(x.to(dev))
    lo, co, _ = O.gssd_forward(sd, x, **flags)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    l64, c64, _ = O.gssd_forward(sd64, x.double(), **flags)
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())
def where(a, b):
    d = (a.double().cpu() - b.double().cpu()).abs()
    i = int(d.reshape(d.shape[0], d.shape[1], -1).amax(2).amax(0).argmax())
    return i
print(f'seed ({wseed}, {xseed})')
print('HIP fp32 vs CPU fp32 : loc %.2e conf %.2e  (prior of the worst conf entry: %d of 8732)' % (rel(loc, lo), rel(conf, co), where(conf, co)))
print('HIP fp32 vs float64  : loc %.2e conf %.2e  (prior %d)' % (rel(loc, l64), rel(conf, c64), where(conf, c64)))
print('CPU fp32 vs float64  : loc %.2e conf %.2e  (prior %d)' % (rel(lo, l64), rel(co, c64), where(co, c64)))
