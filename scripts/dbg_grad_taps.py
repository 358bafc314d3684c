"""Where do the HIP gradients of the branch layers leave the float64 ones?  d(source_i) / d(l2norm) of the HIP backward plan vs CPU
autograd through the oracle in fp32 and fp64 (plain GSSD, B = 4)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import gssd_oracle as O
from gssd import synth
from models.ssd_multiphase_custom_group import build_ssd
dev = torch.device('cuda:0')
args = (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)
flags = dict()
net = build_ssd('train', 300, 2, *args)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
net.load_state_dict(sd); net = net.to(dev).train()
x = synth.synth_images(4, seed=9)
rng = np.random.default_rng(0)
r1 = torch.from_numpy(rng.normal(size=(4, 8732, 4)).astype(np.float32)); r2 = torch.from_numpy(rng.normal(size=(4, 8732, 2)).astype(np.float32))
r1[:, 8728:] = 0; r2[:, 8728:] = 0
loc, conf, _ = net(x.to(dev))
((loc * r1.to(dev)).sum() + (conf * r2.to(dev)).sum()).backward()
plan = net._engine._last_plan
bwd = plan.backward_plan()
def run(dt):
    s = {k: (v.to(dt).requires_grad_() if (v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))) else (v.to(dt) if v.is_floating_point() else v)) for k, v in sd.items()}
    taps = {}
    lo, co, _ = O.gssd_forward(s, x.to(dt), taps=taps, **flags)
    for t in taps.values():
        if t.requires_grad: t.retain_grad()
    ((lo * r1.to(dt)).sum() + (co * r2.to(dt)).sum()).backward()
    return taps
t32, t64 = run(torch.float32), run(torch.float64)
def l2(a, b): a, b = a.double(), b.double(); return float((a - b).norm() / b.norm())
for i, (s, H, Cc) in enumerate(plan.sources):
    g = bwd.gbuf.get(s.data_ptr())
    name = f'source{i}'
    if g is None or name not in t64: continue
    gh = g.permute(0, 3, 1, 2).cpu()
    print(name, 'fwd hip-vs-64', f'{l2(s.permute(0, 3, 1, 2).cpu(), t64[name]):.1e}', 'fwd cpu32-vs-64', f'{l2(t32[name], t64[name]):.1e}',
          '| grad hip-vs-64', f'{l2(gh, t64[name].grad):.1e}', 'cpu32-vs-64', f'{l2(t32[name].grad, t64[name].grad):.1e}')
for kind, r in plan.rec:
    if kind == 'l2norm':
        g = bwd.gbuf.get(r['out'].data_ptr())
        print('d(l2norm out): hip-vs-64', f"{l2(g.permute(0, 3, 1, 2).cpu(), t64['l2norm'].grad):.1e}", 'cpu32-vs-64', f"{l2(t32['l2norm'].grad, t64['l2norm'].grad):.1e}")
        g = bwd.gbuf.get(r['x_in'].data_ptr())
        print('d(l2norm in) available', g is not None)
named = dict(net.named_parameters())
s32 = {}
import ctypes as C
from gssd import _lib
lib = _lib.lib
st = torch.cuda.current_stream().cuda_stream
for kind, r in plan.rec:
    if kind == 'convbn' and r['name'] in ('fuse_11', 'fuse_21', 'extras.2', 'vgg.31'):
        bn, raw, B, Ho, Cout = r['bn'], r['raw'], plan.B, r['Ho'], r['Cout']
        dout = bwd.gbuf[r['out'].data_ptr()]
        print(r['name'], 'pool', r['pool'], 'relu', r['relu'], 'raw mean/std', float(raw.mean()), float(raw.std()))
        # exact reference in float64 on the HIP tensors
        raw64 = raw.double().permute(0, 3, 1, 2).cpu().requires_grad_()
        y = torch.nn.functional.batch_norm(raw64, None, None, bn.weight.double().cpu(), bn.bias.double().cpu(), True, 0.1, bn.eps)
        if r['relu']: y = torch.relu(y)
        if r['pool']:
            pk, ps, pp, ceil = r['pool']
            y = torch.nn.functional.max_pool2d(y, pk, ps, pp, ceil_mode=ceil)
        y.backward(dout.double().permute(0, 3, 1, 2).cpu())
        ref = raw64.grad
        # HIP
        sc, sh, pd_ = (torch.empty(Cout, device=dev) for _ in range(3))
        _lib.check(lib.gssd_bn_finalize_f32(r['stats'].data_ptr(), float(B * Ho * Ho), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                            bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps), 2, Cout,
                                            sc.data_ptr(), sh.data_ptr(), pd_.data_ptr(), r.get('stats_rep', 0), st))
        pool = r['pool']; pk, ps, pp = (pool[0], pool[1], pool[2]) if pool else (0, 1, 0)
        dz = torch.zeros(B, Ho, Ho, Cout, device=dev); sums = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
        _lib.check(lib.gssd_bn_bwd_reduce_f32(dout.data_ptr(), raw.data_ptr(), sc.data_ptr(), sh.data_ptr(), dz.data_ptr(), sums.data_ptr(), B, Ho, Ho,
                                              Cout, r['Hp'], r['Hp'], pk, ps, pp, int(r['relu']), st))
        ca, cb, cc, dg, db = (torch.empty(Cout, device=dev) for _ in range(5))
        _lib.check(lib.gssd_bn_bwd_finalize_f32(r['stats'].data_ptr(), float(B * Ho * Ho), sums.data_ptr(), bn.weight.data_ptr(), float(bn.eps), Cout,
                                                ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), dg.data_ptr(), db.data_ptr(), r.get('stats_rep', 0), st))
        cs = torch.zeros(Cout, dtype=torch.float64, device=dev)
        _lib.check(lib.gssd_bn_bwd_apply_f32(dz.data_ptr(), raw.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), B * Ho * Ho, Cout, cs.data_ptr(), st))
        torch.cuda.synchronize()
        print('   d(raw) hip vs float64 on the same inputs:', f'{l2(dz.permute(0, 3, 1, 2).cpu(), ref):.2e}',
              ' fwd stats check: mean', float((r['stats'][:Cout] / (B * Ho * Ho)).cpu().sub(raw.double().mean(dim=(0, 1, 2)).cpu()).abs().max()))
for i, (s, H, Cc) in enumerate(plan.sources[:5]):
    name = f'source{i}'
    mh = (s.permute(0, 3, 1, 2).cpu() > 0)
    m32, m64 = (t32[name] > 0), (t64[name] > 0)
    n = m64.numel()
    print(name, 'mask flips vs float64: hip', int((mh != m64).sum()), 'cpu32', int((m32 != m64).sum()), 'of', n,
          ' | max |z| among hip flips', float(t64[name][mh != m64].abs().max()) if (mh != m64).any() else 0.0)
