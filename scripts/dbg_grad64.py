import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
from oracle import gssd_oracle as O
from gssd import synth
from models.ssd_multiphase_custom_group import build_ssd
args = (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)
net = build_ssd('train', 300, 2, *args)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
net.load_state_dict(sd); net = net.cuda().train()
B = 4
x = synth.synth_images(B, seed=9)
rng = np.random.default_rng(0)
r1 = torch.from_numpy(rng.normal(size=(B, 8732, 4)).astype(np.float32)); r2 = torch.from_numpy(rng.normal(size=(B, 8732, 2)).astype(np.float32))
cut = int(sys.argv[1]) if len(sys.argv) > 1 else 8728
r1[:, cut:] = 0; r2[:, cut:] = 0
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
keys = ['vgg.0.weight', 'vgg.14.weight', 'vgg.31.weight', 'vgg.44.weight', 'extras.2.weight', 'fuse_11.weight', 'bn_fuse_21.bias', 'loc.0.weight', 'L2Norm.weight']
def oracle_grads(dtype):
    sdg = {k: (v.clone().to(dtype).requires_grad_() if (v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))) else (v.to(dtype) if v.is_floating_point() else v)) for k, v in sd.items()}
    lo, co, _ = O.gssd_forward(sdg, x.to(dtype))
    ((lo * r1.to(dtype)).sum() + (co * r2.to(dtype)).sum()).backward()
    return {k: sdg[k].grad for k in keys}
g64 = oracle_grads(torch.float64); g32 = oracle_grads(torch.float32)
named = dict(net.named_parameters())
res = {}
for mode in ('hip', 'aten'):
    for p in net.parameters(): p.grad = None
    net.__dict__['_force_aten_backward'] = (mode == 'aten')
    loc, conf, _ = net(x.cuda())
    ((loc * r1.cuda()).sum() + (conf * r2.cuda()).sum()).backward()
    res[mode] = {k: named[k].grad.clone() for k in keys}
for k in keys:
    print(f'{k:18s} cpu32 {rel(g32[k], g64[k]):.1e}  hip {rel(res["hip"][k], g64[k]):.1e}  aten {rel(res["aten"][k], g64[k]):.1e}')
