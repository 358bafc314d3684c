"""GPU tests of the training step: HIP backward plan vs CPU autograd through the oracle (free-running decisions; the teacher-forced
all-parameter check lives in tests/test_gpu_grad.py), branch / leaf streams, the gradient-segment hook, autograd integration, SGD steps.

Tolerances (BASELINE.json north_star): integer / index outputs bit-exact; fp32 activations and losses <= 1e-4 relative
(max-abs-diff / max-abs-ref per tensor).  Everything goes through the C ABI (ctypes -> libgssd_hip.so).
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_common import *                      # noqa: E402,F401,F403  (fixtures dev / ops, rel, TOL, nhwc / nchw, NETS, FLAG_NETS, same_detections)
from gpu_common import O, synth, ROOT, _stage_errors     # noqa: E402,F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['gssd', 'gssdpp'])
def test_backward_gradients(dev, name):
    """Training step plumbing: HIP forward + loss, gradients through the HIP loss backward and the interim ATen
    recomputation of the network, against CPU autograd through the oracle graph."""
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args = NETS[name]
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=1111)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    x = synth.synth_images(4, seed=9)
    rng = np.random.default_rng(0)
    r1 = torch.from_numpy(rng.normal(size=(4, 8732, 4)).astype(np.float32))
    r2 = torch.from_numpy(rng.normal(size=(4, 8732, 2)).astype(np.float32))
    r1[:, 8728:] = 0          # the 1x1 map's BatchNorm over 4 values has an ill-conditioned backward: keep it out
    r2[:, 8728:] = 0
    loc, conf, _ = net(x.to(dev))
    assert loc.requires_grad and conf.requires_grad
    ((loc * r1.to(dev)).sum() + (conf * r2.to(dev)).sum()).backward()
    sdg = {k: (v.clone().requires_grad_() if (v.is_floating_point() and not k.endswith(('running_mean', 'running_var',
                                                                                         'weight_u', 'weight_v'))) else v)
           for k, v in sd.items()}
    lo, co, _ = O.gssd_forward(sdg, x, **flags)
    ((lo * r1).sum() + (co * r2).sum()).backward()
    named = dict(net.named_parameters())
    keys = ['vgg.0.weight', 'vgg.14.weight', 'vgg.30.bias', 'vgg.31.weight', 'vgg.44.weight', 'extras.2.weight', 'fuse_11.weight',
            'bn_fuse_21.bias', 'loc.0.weight', 'conf.3.bias', 'L2Norm.weight']
    if name == 'gssdpp':
        keys += ['self_attn_list.0.snconv1x1_theta.weight_orig', 'self_attn_base_list.0.sigma', 'dcn_list.0.weight',
                 'dcn_list.0.conv_offset_mask.weight']
    def l2rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / b.norm())
    errs = {k: l2rel(named[k].grad, sdg[k].grad) for k in keys if k != 'vgg.30.bias'}
    print('gradient L2-relative errors vs CPU autograd', {k: f'{v:.1e}' for k, v in errs.items()})
    # Metric: relative L2 error per tensor.  ReLU masks and max-pool arg-maxes are discontinuous: a forward difference of
    # 1e-7 flips a handful of them between two implementations, and one flip moves single weight-gradient entries by
    # ~1/sqrt(pixels) ~ 1 % (seen equally in CPU-fp32 vs CPU-fp64, scripts/dbg_grad64.py), so max-abs is not meaningful
    # here.  A conv bias in front of a train-mode BatchNorm has a mathematically zero gradient (pure rounding noise).
    assert errs['loc.0.weight'] < 1e-4 and errs['conf.3.bias'] < 1e-4
    assert float(named['vgg.30.bias'].grad.abs().max()) < 1e-2 * float(named['vgg.31.bias'].grad.abs().max())
    # self-attention's sigma is a scalar whose gradient is one heavily cancelling sum over the whole map: looser bound
    assert max(v for k, v in errs.items() if not k.endswith('sigma')) < 2e-2, errs
    assert all(v < 6e-2 for k, v in errs.items() if k.endswith('sigma')), errs
    # The arbiter: the same graph in float64 (CPU autograd through the oracle with double weights / input).  Both fp32 results -- the
    # HIP backward and the CPU fp32 autograd -- leave it through ONE mechanism: ReLU / max-pool decisions at |z| <~ 1e-5 that a 1e-6
    # forward difference flips.  scripts/dbg_grad_taps.py on this very case: every backward kernel reproduces float64 to 5e-8 on the
    # same inputs and d(source_i) to 3e-7; at source 0 three of 2 957 312 ReLU decisions differ from float64 on the HIP path and none on
    # the CPU fp32 path -- those three units ARE the 1.9e-3 of fuse_11 / L2Norm (sqrt(3 / 1.5 M active units)); in the trunk, where
    # both paths flip hundreds of decisions, both sit at the same 5-6e-3.  So: per tensor, HIP must be as close to float64 as CPU fp32
    # is, up to a handful of flips (3x + 5e-3); the head layers (no decision downstream) must match to 1e-5.
    sd64 = {k: (v.double().requires_grad_() if (v.is_floating_point() and not k.endswith(('running_mean', 'running_var', 'weight_u',
                                                                                          'weight_v')))
                else (v.double() if v.is_floating_point() else v)) for k, v in sd.items()}
    lo64, co64, _ = O.gssd_forward(sd64, x.double(), **flags)
    ((lo64 * r1.double()).sum() + (co64 * r2.double()).sum()).backward()
    e_hip64 = {k: l2rel(named[k].grad, sd64[k].grad) for k in keys if k != 'vgg.30.bias'}
    e_cpu64 = {k: l2rel(sdg[k].grad, sd64[k].grad) for k in keys if k != 'vgg.30.bias'}
    print('vs float64: HIP', {k: f'{v:.1e}' for k, v in e_hip64.items()}, 'CPU fp32', {k: f'{v:.1e}' for k, v in e_cpu64.items()})
    for k in e_hip64:
        # Self_Attn's projection weights and sigma: gradients that are heavily cancelling sums over the 1444 x 1444 map (the float64
        # value is ~1e-3 of the sum of magnitudes), so fp32 accumulation ORDER shows: MFMA chains of ~2000 sequential terms here,
        # blocked sums in oneDNN -- 1.2e-2 vs 4e-4 on theta, 1.1e-2 vs 1.1e-3 on sigma
        slack = 3e-2 if ('snconv1x1' in k or k.endswith('sigma')) else 5e-3
        assert e_hip64[k] <= 3.0 * e_cpu64[k] + slack, (k, e_hip64[k], e_cpu64[k])
    assert e_hip64['loc.0.weight'] < 1e-5 and e_hip64['conf.3.bias'] < 1e-5
    # the HIP backward plan against the whole-graph ATen recomputation on the same device (tests/aten_shadow.py); the
    # spectral-norm u / v the HIP forward used are the ones the module holds now
    from aten_shadow import shadow_param_grads
    sg = dict(zip([k for k, _ in net.named_parameters()], shadow_param_grads(net, x.to(dev), r1.to(dev), r2.to(dev))))
    e2 = {k: l2rel(named[k].grad, sg[k]) for k in keys if k != 'vgg.30.bias'}
    print('HIP vs ATen backward (L2-relative)', {k: f'{v:.1e}' for k, v in e2.items()})
    assert max(v for k, v in e2.items() if not k.endswith('sigma')) < 2e-2, e2
    assert all(v < 6e-2 for k, v in e2.items() if k.endswith('sigma')), e2


def test_backward_branch_streams_equal_single_stream(dev):
    """gssd/backward.py runs the backward of branch blocks 1 .. 5 on their own streams beside the trunk's and the trunk's
    parameter-gradient launches on a "leaf" stream; the same step list on one stream (the mode used with a gradient-segment hook /
    GSSD_BWD_STREAMS=0) must give the same gradients (split-K atomics aside)."""
    import copy
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args = NETS['gssdpp']
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth.synth_state_dict(shapes, seed=1111))
    net = net.to(dev).train()
    sd0 = copy.deepcopy(net.state_dict())
    x = synth.synth_images(4, seed=9).to(dev)
    rng = np.random.default_rng(0)
    r1 = torch.from_numpy(rng.normal(size=(4, 8732, 4)).astype(np.float32)).to(dev)
    r2 = torch.from_numpy(rng.normal(size=(4, 8732, 2)).astype(np.float32)).to(dev)

    def grads(single):
        net.load_state_dict(sd0)
        net.zero_grad(set_to_none=True)
        loc, conf, _ = net(x)
        bp = net._engine._last_plan.backward_plan()
        assert bp.hoisted in ([6, 5, 4, 3, 2], [6, 5, 4, 3, 2, 1])
        bp.single_stream = single
        try:
            ((loc * r1).sum() + (conf * r2).sum()).backward()
        finally:
            bp.single_stream = False
        return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    # Gradients that are mathematically zero (conv biases in front of a train-mode BatchNorm, phi's bias, ...) hold the rounding noise of
    # atomically accumulated sums and differ between any two runs; so the yardstick per tensor is the difference between two
    # single-stream runs.  Several rounds: a missing stream dependency shows as a sporadic mismatch.
    for rep in range(3):
        ga, gb, gc = grads(False), grads(True), grads(True)
        assert ga.keys() == gb.keys()
        gmax = max(float(v.norm()) for v in gb.values())
        for k in ga:
            d_ms, d_ss, ref = float((ga[k] - gb[k]).norm()), float((gc[k] - gb[k]).norm()), float(gb[k].norm())
            # (+ 1e-6 of the largest gradient: tensors that are pure noise -- e.g. theta / phi of the 1 x 1 map's attention block,
            # whose single-key softmax has no gradient -- are noise on both sides of the comparison)
            assert d_ms <= 3.0 * d_ss + 1e-5 * ref + 1e-6 * gmax, (k, d_ms, d_ss, ref, gmax)


def test_gradient_segment_hook_sees_final_ranges(dev):
    """gssd.dist.OverlappedGradReducer's contract with the multi-stream backward: when the hook fires for a range of the flat gradient
    buffer, every launch writing into that range -- on the main, branch and leaf streams -- is ordered before work enqueued from the
    hook.  A fake hook snapshots its range on the current stream; the snapshots must equal the final gradients."""
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args = NETS['gssdpp']
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth.synth_state_dict(shapes, seed=1111))
    net = net.to(dev).train()
    x = synth.synth_images(4, seed=9).to(dev)
    rng = np.random.default_rng(0)
    r1 = torch.from_numpy(rng.normal(size=(4, 8732, 4)).astype(np.float32)).to(dev)
    r2 = torch.from_numpy(rng.normal(size=(4, 8732, 2)).astype(np.float32)).to(dev)
    snaps = {}

    def hook(k, flat_slice):
        snaps[k] = (flat_slice, flat_slice.clone())
    net._engine.grad_segment_hook = hook
    try:
        for rep in range(3):
            snaps.clear()
            net.zero_grad(set_to_none=True)
            loc, conf, _ = net(x)
            ((loc * r1).sum() + (conf * r2).sum()).backward()
            torch.cuda.synchronize()
            assert sorted(snaps) == [0, 1, 2, 3]
            assert sum(s[0].numel() for s in snaps.values()) == net._engine._last_plan.backward_plan().flat.numel()
            for k, (final, snap) in snaps.items():
                assert torch.equal(final, snap), k
    finally:
        net._engine.grad_segment_hook = None


def test_training_steps_reduce_loss(dev):
    """The driver's step sequence (train_lesion_multiphase_v2.py:242-253) on a fixed synthetic batch: forward, MultiBoxLoss,
    backward (HIP), SGD(momentum 0.9, wd 5e-4).  The loss must fall, and the weights must track the same steps taken with
    the ATen backward."""
    import copy
    from models.ssd_multiphase_custom_group import build_ssd
    from layers.modules import MultiBoxLoss
    flags, args = NETS['gssd']
    net = build_ssd('train', 300, 2, *args)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
    net = net.to(dev).train()
    twin = copy.deepcopy(net)
    from aten_shadow import shadow_param_grads
    x = synth.synth_images(8, seed=21).to(dev)
    tg = [t.to(dev) for t in synth.synth_targets(8, seed=21)]
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    hist = {}
    for tag, m in (('hip', net), ('aten', twin)):
        opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
        losses = []
        for _ in range(4):
            opt.zero_grad()
            if tag == 'hip':
                ll, lc = crit(m(x), tg)
                (ll + lc).backward()
            else:
                # same HIP forward and loss; the network gradient from the ATen recomputation instead of the HIP plan
                with torch.no_grad():
                    loc, conf, pri = m(x)
                loc.requires_grad_(), conf.requires_grad_()
                ll, lc = crit((loc, conf, pri), tg)
                (ll + lc).backward()
                for p_, g_ in zip(m.parameters(), shadow_param_grads(m, x, loc.grad, conf.grad)):
                    p_.grad = g_
            opt.step()
            losses.append(float(ll + lc))
        hist[tag] = losses
    print('loss trajectories', hist)
    assert all(np.isfinite(hist['hip'])) and hist['hip'][-1] < 0.7 * hist['hip'][0]
    # SGD with momentum on a 20-layer train-mode-BN net amplifies rounding differences step by step (ReLU / arg-max flips, fp32
    # atomics in the split-K weight gradients): four steps track within 1 %, per step
    assert all(abs(a - b) < 1e-2 * abs(b) for a, b in zip(hist['hip'], hist['aten'])), hist
    w1, w2 = dict(net.named_parameters()), dict(twin.named_parameters())
    for k in ('vgg.0.weight', 'vgg.24.weight', 'fuse_21.weight', 'loc.2.weight'):
        assert rel(w1[k], w2[k]) < 1e-2, k


def test_backward_runs_against_its_own_forward(dev):
    """a second forward before .backward() must not hand stale activations to the HIP backward.  Two micro-batches
    summed into one loss get a plan instance each; a plan re-run behind a pending backward's back raises."""
    from models.ssd_multiphase_custom_group import build_ssd
    from gssd._lib import GssdError
    flags, args = NETS['gssd']
    net = build_ssd('train', 300, 2, *args)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
    net = net.to(dev).train()
    xa, xb = synth.synth_images(2, seed=31).to(dev), synth.synth_images(2, seed=32).to(dev)
    rng = np.random.default_rng(1)
    ra = torch.from_numpy(rng.normal(size=(2, 8732, 4)).astype(np.float32)).to(dev)
    ra[:, 8728:] = 0

    def grads_of(xs):
        for p in net.parameters():
            p.grad = None
        outs = [net(x) for x in xs]                      # all forwards first, then ONE backward
        sum((o[0] * ra).sum() for o in outs).backward()
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    # BN running stats / batch statistics are per forward, so (a then b in one graph) == grad(a) + grad(b)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    gab = grads_of([xa, xb])
    net.load_state_dict(sd0)
    ga = grads_of([xa])
    gb = grads_of([xb])
    for k in ('vgg.0.weight', 'vgg.24.weight', 'fuse_21.weight', 'loc.0.weight', 'extras.4.weight'):
        assert rel(gab[k], ga[k] + gb[k]) < 1e-4, k
    # a no-grad forward while a backward is pending must take another plan (the pending one stays intact)
    loc, conf, _ = net(xa)
    with torch.no_grad():
        net(xb)
    (loc * ra).sum().backward()
    # retained graph + a fresh forward on the same plan: the second backward would read overwritten buffers -> raises
    for p in net.parameters():
        p.grad = None
    loc, conf, _ = net(xa)
    (loc * ra).sum().backward(retain_graph=True)
    net(xb)[0].sum().backward()
    with pytest.raises((GssdError, RuntimeError)):
        (loc * ra).sum().backward()


def test_autograd_grad_and_hooks_see_gradients(dev):
    """torch.autograd.grad / tensor hooks / non-leaf parameters receive the HIP gradients (the no-copy p.grad
    hand-out is only taken for a plain .backward() on hook-free leaves)."""
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args = NETS['gssd']
    net = build_ssd('train', 300, 2, *args)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
    net = net.to(dev).train()
    x = synth.synth_images(2, seed=33).to(dev)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    loc, conf, _ = net(x)
    loc[:, :8728].sum().backward()
    ref = net.loc[0].weight.grad.clone()
    assert net.loc[0].weight.grad._base is not None          # fast path: a view of the plan's flat gradient buffer
    net.load_state_dict(sd0)
    loc, conf, _ = net(x)
    g, = torch.autograd.grad(loc[:, :8728].sum(), [net.loc[0].weight])
    assert g is not None and rel(g, ref) < 1e-5
    net.load_state_dict(sd0)
    for p in net.parameters():
        p.grad = None
    seen = []
    h = net.loc[0].weight.register_hook(lambda gr: seen.append(gr.clone()))
    loc, conf, _ = net(x)
    loc[:, :8728].sum().backward()
    h.remove()
    assert len(seen) == 1 and rel(seen[0], ref) < 1e-5 and rel(net.loc[0].weight.grad, ref) < 1e-5


def test_plan_follows_reseated_storage(dev):
    """parameters / buffers whose storage is replaced without going through ``_apply`` (``p.data = ...``,
    ``bn.running_mean = ...``) rebuild the launch plan instead of leaving kernels on the old pointers."""
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args = NETS['gssd']
    net = build_ssd('train', 300, 2, *args)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    x = synth.synth_images(2, seed=34).to(dev)
    with torch.no_grad():
        l0, c0, _ = net(x)
        bn = net.vgg[1]
        bn.running_mean = bn.running_mean.clone() + 0.25         # new storage, new values
        net.bn_fuse_11.bias.data = net.bn_fuse_11.bias.data.clone() + 0.5
        l1, c1, _ = net(x)
        sd2 = {k: v.clone() for k, v in net.state_dict().items()}
        lo, co, _ = O.gssd_forward({k: v.cpu() for k, v in sd2.items()}, x.cpu(), training=False, **flags)
    assert rel(l1, lo) < TOL and rel(c1, co) < TOL and rel(l1, l0) > 1e-3
