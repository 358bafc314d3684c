"""Teacher-forced gradient parity: EVERY parameter tensor of the HIP training step against float64 autograd
through the oracle graph (models/ssd_multiphase_custom_group.py:217-400 restated in oracle/gssd_oracle.py), with the graph's
discontinuous decisions -- ReLU masks, max-pool arg-maxes and the deformable conv's sampling cells -- taken from the HIP path.

tests/test_gpu_training.py::test_backward_gradients lets both sides take their own decisions; a forward difference of 1e-7 then flips a
handful of ReLU / arg-max decisions between the two implementations and each flip moves single weight-gradient entries by ~1 % (seen
equally between CPU fp32 and CPU float64), which is why that test can only hold 2e-2 on 11-15 tensors.  Here the float64 graph is
evaluated with the decisions the HIP backward really took -- reconstructed exactly from what its kernel reads: the raw conv output the
forward stored, and the BatchNorm scale / shift of gssd_bn_finalize_f32 (csrc/backward.hip::bn_bwd_reduce_kernel decides on
z = fma(raw, scale, shift): ReLU mask z > 0, first maximum of max(z, 0) in window scan order) -- so the two gradients differ by
arithmetic only and every tensor can be held to a tight bound (BOUND()).  Round 5, first run of this test: it exposed that the oracle's
spectral-norm restatement differentiated THROUGH the power iteration (the reference runs it under torch.no_grad()): the float64
"reference" gradients of all 48 spectral-normed weights were off by 1e-2 .. 7e-2 -- fixed in oracle/gssd_oracle.py::spectral_weight and pinned
against the imported reference's own autograd (tests/golden/grad.npz, tests/test_oracle_golden.py::test_self_attn_gradients_vs_reference).
"""
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import gssd_oracle as O          # noqa: E402
from gssd import synth                       # noqa: E402

pytestmark = pytest.mark.gpu

NETS = {
    'gssd': (dict(), (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)),
    'gssdpp': (dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True),
               (True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)),
}


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def hip_decisions(plan, dev):
    """ReLU masks / pool arg-maxes of every conv + BatchNorm record of a HIP forward plan, from the tensors its backward reads."""
    from gssd import _lib
    lib = _lib.lib
    st = torch.cuda.current_stream().cuda_stream
    dec = O.Decisions()
    for kind, r in plan.rec:
        if kind == 'pool':
            # a stand-alone pool (pool4 = vgg[33] reads the activated conv4_3 map -- behind the deformable conv in GSSD++): the backward
            # kernel takes the first maximum of the stored fp32 input in window scan order
            assert (r['k'], r['s'], r['p']) == (2, 2, 0), r
            xin = r['x_in'].permute(0, 3, 1, 2).float().cpu()
            ceil = r['Hp'] != (r['H'] - r['k']) // r['s'] + 1
            _, idx = F.max_pool2d(xin, r['k'], r['s'], r['p'], ceil_mode=ceil, return_indices=True)
            dec.pool_idx['vgg.33'] = idx
            continue
        if kind == 'dcn':
            # the bilinear sampler's cells from the fp32 offsets the HIP forward stored, with the kernel's own arithmetic
            # (csrc/dcn_x6.hip::setups: py = (float)(h - 1 + tap / 3) + dy in fp32, floorf, the -1 < p < H gate)
            dg, H = r['dg'], r['H']
            om = r['om'].float().cpu()[..., :27 * dg]                          # [B, H, W, 27 dg]: offsets d*18 + 2*tap (+1), then masks
            off = om[..., :18 * dg].reshape(om.shape[0], H, H, dg, 9, 2).permute(0, 3, 4, 5, 1, 2)      # [B, dg, tap, yx, H, W]
            hh = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)
            ww = torch.arange(H, dtype=torch.float32).view(1, 1, 1, H)
            cells = dict(y0=[], x0=[], valid=[])
            for k in range(9):
                py = (hh - 1 + k // 3) + off[:, :, k, 0]
                px = (ww - 1 + k % 3) + off[:, :, k, 1]
                cells['valid'].append((py > -1) & (px > -1) & (py < H) & (px < H))
                cells['y0'].append(torch.floor(py))
                cells['x0'].append(torch.floor(px))
            dec.dcn_cells[f'dcn_list.{r["li"]}'] = cells
            continue
        if kind != 'convbn':
            continue
        bn, raw, B, Ho, Cout = r['bn'], r['raw'], plan.B, r['Ho'], r['Cout']
        sc, sh, pd_ = (torch.empty(Cout, device=dev) for _ in range(3))
        # training = 2: the batch statistics again, no second running update (what the backward plan itself calls)
        _lib.check(lib.gssd_bn_finalize_f32(r['stats'].data_ptr(), float(B * Ho * Ho), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                            bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps), 2, Cout,
                                            sc.data_ptr(), sh.data_ptr(), pd_.data_ptr(), r.get('stats_rep', 0), st))
        torch.cuda.synchronize()
        # z as the kernel's fma(raw, scale, shift): exact product and sum in float64, one rounding to fp32
        z = (raw.double() * sc.double() + sh.double()).float().permute(0, 3, 1, 2).cpu()
        name = r['name']
        if name.startswith('vgg.'):
            i = int(name.split('.')[1])
            relu_site, pool_site = f'vgg.{i + 2}', f'vgg.{i + 3}'
        elif name.startswith('extras.'):
            i = int(name.split('.')[1])
            relu_site, pool_site = f'extras.{i + 1}', None
        else:
            assert name.startswith('fuse_'), name
            relu_site, pool_site = 'bn_' + name, None
        assert r['relu'], name                                  # every BatchNorm of this graph is followed by a ReLU
        dec.relu_mask[relu_site] = z > 0
        if r['pool']:
            k, s, p, ceil = r['pool']
            _, idx = F.max_pool2d(torch.relu(z), k, s, p, ceil_mode=bool(ceil), return_indices=True)      # first maximum wins, as the kernel
            dec.pool_idx[pool_site] = idx
    return dec


# (gssdpp, 32) -- round 6, VERDICT r5 item 1b: the batch bench.py's `full_step` times.  Only at this size do the data gradients of conv3_2 .. conv4_3
# run csrc/conv_wino_x6.hip (>= 8 192 Winograd tiles), conv6 / conv7 / fuse_21 / the 19 x 19 Self_Attn convs and the deformable conv's d(cols) GEMM
# csrc/conv_x6.hip (M >= 4 096), and the wgrad / BatchNorm-backward kernels their batch-32 tilings: the float64 graph costs ~80 GB of host memory
# and a minute or two of the GPU box's 256 host cores, once.
@pytest.mark.parametrize('name,B', [('gssd', 4), ('gssdpp', 4), ('gssdpp', 32)])
def test_teacher_forced_gradients_all_parameters(dev, name, B):
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args = NETS[name]
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=1111)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    x = synth.synth_images(B, seed=9)
    rng = np.random.default_rng(0)
    r1 = torch.from_numpy(rng.normal(size=(B, 8732, 4)).astype(np.float32))
    r2 = torch.from_numpy(rng.normal(size=(B, 8732, 2)).astype(np.float32))
    loc, conf, _ = net(x.to(dev))
    ((loc * r1.to(dev)).sum() + (conf * r2.to(dev)).sum()).backward()
    plan = net._engine._last_plan
    if B >= 24:
        # the kernel mix this batch exists for: the three-plane kernels really took their launches, forward and backward
        fwd_k = {st.tag[0].split('/')[0] for st in plan.steps if st.tag is not None}
        assert {'conv_wino_x6<64>', 'conv_x6<128>', 'dcn_x6<128x256>', 'flash_attn_x6<64,256>'} <= fwd_k, fwd_k
        import ctypes as C
        from gssd import _lib
        bwd_descs = [a for fn, args in plan._bwd.steps if getattr(fn, '__name__', '') == 'gssd_conv2d_nhwc_f32' for a in args]
        n_wx6 = sum(_lib.lib.gssd_conv_wino_x6_takes(a) == 1 for a in bwd_descs)
        n_cx6 = sum(_lib.lib.gssd_conv_x6_takes(a) == 1 for a in bwd_descs)
        print(f'backward plan at B = {B}: {n_wx6} data-gradient launches on conv_wino_x6, {n_cx6} on conv_x6')
        assert n_wx6 >= 5 and n_cx6 >= 6, (n_wx6, n_cx6)
    dec = hip_decisions(plan, dev)
    # float64 autograd through the oracle with the HIP path's decisions (spectral norm's u / v: the pre-step values, as the HIP forward used)
    sd64 = {k: (v.double().requires_grad_() if (v.is_floating_point() and not k.endswith(('running_mean', 'running_var', 'weight_u', 'weight_v')))
                else (v.double() if v.is_floating_point() else v)) for k, v in sd.items()}
    lo, co, _ = O.gssd_forward(sd64, x.double(), decide=dec, **flags)
    sites = set(dec.relu_mask) | set(dec.pool_idx) | set(dec.dcn_cells)
    assert dec.used == sites, dec.used ^ sites
    # the forced graph is the HIP forward's graph: outputs agree to fp32 rounding
    fw = max(float((loc.detach().cpu().double() - lo.detach()).abs().max() / lo.detach().abs().max()),
             float((conf.detach().cpu().double() - co.detach()).abs().max() / co.detach().abs().max()))
    ((lo * r1.double()).sum() + (co * r2.double()).sum()).backward()
    named = dict(net.named_parameters())

    def l2rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / max(float(b.norm()), 1e-300))
    errs, zero_grad, skipped = {}, {}, []
    gmax = max(float(g.grad.abs().max()) for g in sd64.values() if torch.is_tensor(g) and g.requires_grad and g.grad is not None)
    for k, p in named.items():
        ref = sd64[k].grad
        if ref is None or p.grad is None:
            assert ref is None and p.grad is None, k            # exactly the parameters the reference graph leaves out
            skipped.append(k)
            continue
        if float(ref.abs().max()) < 1e-9 * gmax:
            # a conv bias in front of a train-mode BatchNorm: mathematically zero gradient, pure rounding noise on both sides
            zero_grad[k] = float(p.grad.abs().max())
            continue
        errs[k] = l2rel(p.grad, ref)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:12]
    lines = [f'{name} B={B}: teacher-forced gradients, {len(errs)} tensors compared (+ {len(zero_grad)} zero-gradient biases, {len(skipped)} unused); '
             f'forward vs the forced float64 graph {fw:.2e}',
             'worst: ' + ' '.join(f'{k} {v:.1e}' for k, v in worst),
             f'median {float(np.median(list(errs.values()))):.1e}  tensors > 1e-4: {sum(v > 1e-4 for v in errs.values())}  > 1e-3: {sum(v > 1e-3 for v in errs.values())}']
    print('\n'.join(lines))
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, f'grad_teacher_forced_{name}' + (f'_b{B}' if B != 4 else '') + '.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n' + '\n'.join(f'{k} {v:.3e}' for k, v in sorted(errs.items(), key=lambda kv: -kv[1])) + '\n')
    assert fw < 1e-4
    assert len(errs) + len(zero_grad) + len(skipped) == len(named)
    assert all(v < 1e-2 * gmax for v in zero_grad.values()), zero_grad
    bad = {k: v for k, v in errs.items() if v > BOUND(k)}
    assert not bad, bad
    assert float(np.median(list(errs.values()))) < 2e-4


SMALL_MAP = re.compile(r'^(extras\.|fuse_[456]1|bn_fuse_[456]1|self_attn(_base)?_list\.[345]\.|loc\.[345]|conf\.[345])')


def BOUND(k):
    """Per-tensor bound on the relative L2 error vs float64.  Layers on the 38 x 38 .. 10 x 10 maps (the whole trunk, the deformable conv,
    Self_Attn blocks 0-2, fuse_11 / 21 / 31, heads 0-2): 3e-4 (measured <= 1.4e-4).  Layers that sit on or behind the 5 x 5 / 3 x 3 / 1 x 1
    maps -- train-mode BatchNorm over 4 x 25 .. 4 x 1 values per channel at this batch amplifies fp32 rounding in BOTH directions of the
    graph -- 2e-3 (measured <= 9.2e-4); Self_Attn's sigma there (a scalar: one cancelling sum over the block) 5e-3 (measured 1.9e-3)."""
    if SMALL_MAP.match(k):
        return 5e-3 if k.endswith('sigma') else 2e-3
    return 3e-4
