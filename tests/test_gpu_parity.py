"""GPU parity tests, end to end: the GSSD / GSSD++ / vanilla nets against the reference-generated fixtures and the CPU oracle (batch 4
fixtures, the B = 32 gate of north_star on full tensors, constructor flags, visualize outputs, graph replay, determinism, the reference
driver sequence).

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


@pytest.mark.parametrize('name', list(NETS))
def test_end_to_end(dev, golden, name):
    from models.ssd_multiphase_custom_group import build_ssd
    from layers.modules import MultiBoxLoss
    flags, args = NETS[name]
    g = golden('e2e')
    net = build_ssd('train', 300, 2, *args)
    keys = sorted(net.state_dict().keys())
    assert keys == [str(k) for k in g[f'{name}.keys']]                              # checkpoint key parity
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert [str(shapes[k]) for k in keys] == [str(s) for s in g[f'{name}.shapes']]
    sd = synth.synth_state_dict(shapes, seed=1111)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    x = synth.synth_images(4, seed=5)
    tg = synth.synth_targets(4, seed=5)
    with torch.no_grad():
        loc, conf, pri = net(x.to(dev))
        ll, lc = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)((loc, conf, pri), tg)
    assert loc.shape == (4, 8732, 4) and conf.shape == (4, 8732, 2) and pri.shape == (8732, 4)
    # (1) vs the reference's sampled outputs
    l, c = loc.cpu().numpy().reshape(-1), conf.cpu().numpy().reshape(-1)
    assert np.abs(l[g[f'{name}.loc_idx']] - g[f'{name}.loc_val']).max() / g[f'{name}.loc_absmax'] < TOL
    assert np.abs(c[g[f'{name}.conf_idx']] - g[f'{name}.conf_val']).max() / g[f'{name}.conf_absmax'] < TOL
    assert rel(ll, g[f'{name}.loss'][0]) < TOL and rel(lc, g[f'{name}.loss'][1]) < TOL
    # (2) full tensors vs the oracle
    with torch.no_grad():
        lo, co, upd = O.gssd_forward(sd, x, **flags)
    assert rel(loc, lo) < TOL and rel(conf, co) < TOL
    # (3) state the training forward mutates
    after = net.state_dict()
    for k, v in upd.items():
        assert rel(after[k], v) < TOL, k
    for k in ('vgg.1.running_mean', 'vgg.1.running_var', 'bn_fuse_11.running_mean', 'extras.15.running_var'):
        assert rel(after[k], g[f'{name}.after.{k}']) < TOL, k
    assert int(after['vgg.1.num_batches_tracked']) == 1
    # (4) test phase twin: strict load, eval-mode BN, fused softmax + Detect.  As in make_golden.py, one more
    # training forward with BN momentum 1.0 first makes the running statistics describe these activations.
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    with torch.no_grad():
        net(x.to(dev))
    net_t = build_ssd('test', 300, 2, *args)
    net_t.load_state_dict(net.state_dict())
    net_t = net_t.to(dev).eval()
    with torch.no_grad():
        det = net_t(x.to(dev)).cpu().numpy()
    assert det.shape == (4, 2, 200, 5)
    ref = g[f'{name}.det']
    assert np.array_equal(det[..., 0] > 0, ref[..., 0] > 0)
    assert same_detections(det, ref, 5e-5)


@pytest.mark.parametrize('name', ['gssd', 'gssdpp'])
def test_end_to_end_seed_sweep_margin(dev, name):
    """the 1e-4 gate held with 25 % headroom on ONE seed.  Five weight / image seed pairs, full tensors against the
    (fixture-pinned) oracle, with the per-stage error table printed (pytest -s) and written to gpurun_out/ so the layer that eats the
    budget is named.

    What the sweep shows (round 3, scripts/dbg_fwd64.py): the error is NOT the Winograd trunk's (GSSD_NO_WINOGRAD=1 gives the same 5 -
    7e-5) -- it is fp32 rounding noise amplified by the train-mode BatchNorms of the extras chain, whose statistics at this test's
    batch of 4 run over 4 x 9 values on the 3 x 3 map and over FOUR values on the 1 x 1 map.  The worst entries always sit on the four
    priors of the 1 x 1 map (8728 .. 8731).  There the fp32 REFERENCE is itself only defined to ~5e-5: against the same graph in
    float64 the CPU fp32 oracle is off by 4 - 5e-5 and the HIP path by 5 - 7e-5, and when the two errors have opposite signs their
    difference touches 1e-4 (seed 31337: 1.04e-4; before the Winograd epilogue of round 3 changed the order of the batch sums: 9.9e-5).
    Gates: priors 0 .. 8727 and both losses against the fp32 oracle at 1e-4 (north_star); the four 1 x 1-map priors against FLOAT64 at
    1e-4, and no worse than twice the fp32 oracle's own distance from float64 (+ 2e-5)."""
    from models.ssd_multiphase_custom_group import build_ssd
    from layers.modules import MultiBoxLoss
    flags, args = NETS[name]
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net = net.to(dev).train()
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    pri = O.prior_box()
    lines, worst, worst_tail = [], 0.0, 0.0
    NT = 8728                                   # priors of the 1 x 1 map start here

    def part(a, b, lo_, hi_):                   # max error over a prior range, relative to the whole reference tensor's max
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a[:, lo_:hi_] - b[:, lo_:hi_]).abs().max() / b.abs().max())
    for wseed, xseed in ((1111, 11), (2024, 12), (7, 13), (31337, 14), (99, 15)):
        sd = synth.synth_state_dict(shapes, seed=wseed)
        net.load_state_dict(sd)
        x = synth.synth_images(4, seed=xseed)
        tg = synth.synth_targets(4, seed=xseed)
        taps = {}
        with torch.no_grad():
            loc, conf, _ = net(x.to(dev))
            ll, lc = crit((loc, conf, torch.from_numpy(pri).to(dev)), tg)
            lo, co, _ = O.gssd_forward(sd, x, taps=taps, **flags)
            sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
            l64, c64, _ = O.gssd_forward(sd64, x.double(), **flags)
        rl, rc = O.multibox_loss(lo.numpy(), co.numpy(), pri, [t.numpy() for t in tg])[:2]
        e = dict(loc=part(loc, lo, 0, NT), conf=part(conf, co, 0, NT), loss_l=rel(ll, rl), loss_c=rel(lc, rc))
        tail = dict(hip_ref=max(part(loc, lo, NT, None), part(conf, co, NT, None)),
                    hip_f64=max(part(loc, l64, NT, None), part(conf, c64, NT, None)),
                    ref_f64=max(part(lo, l64, NT, None), part(co, c64, NT, None)))
        st = _stage_errors(net, taps)
        lines.append(f'{name} weights {wseed} images {xseed}: ' + ' '.join(f'{k} {v:.2e}' for k, v in e.items()) +
                     '  | 1x1-map priors: HIP-oracle %.2e HIP-float64 %.2e oracle-float64 %.2e' % (tail['hip_ref'], tail['hip_f64'],
                                                                                                   tail['ref_f64']))
        lines += [f'    {k:18s} {v:.2e}' for k, v in st]
        worst = max(worst, *e.values())
        worst_tail = max(worst_tail, tail['hip_f64'])
        assert tail['hip_f64'] < TOL and tail['hip_f64'] <= 2.0 * tail['ref_f64'] + 2e-5, lines[-len(st) - 1]
    lines.append(f'{name}: worst of 5 seeds {worst:.2e} vs the fp32 oracle on priors 0..8727 and the losses, {worst_tail:.2e} vs float64 on '
                 f'the 1x1-map priors (gate {TOL:.0e})')
    print('\n'.join(lines))
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, f'parity_margin_{name}.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
    assert worst < TOL, lines[-1]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_pooled_raw_plans_are_bit_identical(dev, dtype):
    """A forward that no backward follows (torch.no_grad()) runs a plan whose pooled trunk layers store max- / min-pooled RAW maps (by the
    sign of the BatchNorm weight, GSSD_CONV_POOL2) and skip the BatchNorm + ReLU + pool pass; `net.pooled_raw = False` keeps the
    ordinary plan.  Max-pooling commutes with the monotone BatchNorm + ReLU, so the two plans must agree BIT FOR BIT -- outputs and
    the running statistics they update -- with positive, negative and zero BatchNorm weights on the pooled layers."""
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args = NETS['gssdpp']
    net = build_ssd('train', 300, 2, *args)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
    for k in ('vgg.4.weight', 'vgg.11.weight', 'vgg.21.weight'):           # BatchNorms of conv1_2, conv2_2, conv3_3
        g = sd[k].clone()
        g[1::3] *= -1.0
        g[5] = 0.0
        sd[k] = g
    x = synth.synth_images(4, seed=5).to(dev)
    outs = {}
    for pooled in (True, False):
        net.load_state_dict(sd)
        net = net.to(dev).train()
        net.compute_dtype = dtype
        net.pooled_raw = pooled
        with torch.no_grad():
            loc, conf, _ = net(x)
        plan = net._engine._last_plan
        npool = sum(1 for kind, r in plan.rec if kind == 'convbn' and r.get('pooled'))
        assert npool == ((3 if dtype == 'f32' else 2) if pooled else 0), npool
        outs[pooled] = (loc.clone(), conf.clone(), {k: v.clone() for k, v in net.state_dict().items() if 'running' in k})
    print('pooled vs unpooled plan: max |d loc|', float((outs[True][0] - outs[False][0]).abs().max()), 'max |d conf|',
          float((outs[True][1] - outs[False][1]).abs().max()))
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    for k, v in outs[True][2].items():
        assert torch.equal(v, outs[False][2][k]), k
    # and a grad-enabled forward never takes the pooled plan (its backward needs the raw maps)
    net.pooled_raw = True
    net.compute_dtype = 'f32'
    loc, conf, _ = net(x)
    assert loc.requires_grad and not any(r.get('pooled') for kind, r in net._engine._last_plan.rec if kind == 'convbn')


@pytest.mark.parametrize('name', list(FLAG_NETS))
def test_constructor_flags(dev, golden, name):
    """The driver-reachable non-default flags (train_lesion_multiphase_v2.py:49-77, all passed positionally at :142-145):
    --use_fuseconv False, --batch_norm False, --max_pool_factor 2 / 3, --feature_scale 2.  Checkpoint keys, forward and loss against
    what the imported reference computed (flags.npz), full tensors and mutated state against the oracle, and the HIP backward
    against the same graph recomputed with ATen on the device (tests/aten_shadow.py)."""
    from models.ssd_multiphase_custom_group import build_ssd
    from layers.modules import MultiBoxLoss
    flags, args, gkeys = FLAG_NETS[name]
    g = golden('flags')
    net = build_ssd('train', 300, 2, *args)
    keys = sorted(net.state_dict().keys())
    assert keys == [str(k) for k in g[f'{name}.keys']]
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert [str(shapes[k]) for k in keys] == [str(s) for s in g[f'{name}.shapes']]
    sd = synth.synth_state_dict(shapes, seed=1111)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    x = synth.synth_images(2, seed=5)
    tg = synth.synth_targets(2, seed=5)
    rng = np.random.default_rng(1)
    r1 = torch.from_numpy(rng.normal(size=(2, 8732, 4)).astype(np.float32))
    r2 = torch.from_numpy(rng.normal(size=(2, 8732, 2)).astype(np.float32))
    if args[0]:
        r1[:, 8728:] = 0          # the 1x1 map's BatchNorm over 2 values has an ill-conditioned backward: keep it out
        r2[:, 8728:] = 0
    loc, conf, pri = net(x.to(dev))
    with torch.no_grad():
        ll, lc = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)((loc.detach(), conf.detach(), pri), tg)
    l, c = loc.detach().cpu().numpy().reshape(-1), conf.detach().cpu().numpy().reshape(-1)
    assert np.abs(l[g[f'{name}.loc_idx']] - g[f'{name}.loc_val']).max() / g[f'{name}.loc_absmax'] < TOL
    assert np.abs(c[g[f'{name}.conf_idx']] - g[f'{name}.conf_val']).max() / g[f'{name}.conf_absmax'] < TOL
    assert rel(ll, g[f'{name}.loss'][0]) < TOL and rel(lc, g[f'{name}.loss'][1]) < TOL
    with torch.no_grad():
        lo, co, upd = O.gssd_forward(sd, x, **flags)
    # the last source is a 1 x 1 map: with B = 2 its BatchNorms normalise TWO values per channel (+-1 wherever |x1 - x2| >> sqrt(eps),
    # a steep function of x1 - x2 elsewhere), so the last 4 priors and those layers' statistics amplify 1e-6 differences
    tail = 4 if args[0] else 0
    e_l, e_c = rel(loc.detach()[:, :8732 - tail], lo[:, :8732 - tail]), rel(conf.detach()[:, :8732 - tail], co[:, :8732 - tail])
    if not (e_l < TOL and e_c < TOL):
        # Two fp32 evaluations of a B = 2 train-mode graph can sit up to ~1.1e-4 apart (fs2pp: 2048-channel maps; measured with either
        # deformable-conv kernel in the plan) while both are within 1e-4 of the exact result.  The arbiter of test_parity_margin_five_seeds:
        # the same graph in float64 -- HIP must be within the gate of it and no farther from it than twice the fp32 oracle is (+ 2e-5).
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        with torch.no_grad():
            l64, c64, _ = O.gssd_forward(sd64, x.double(), **flags)
        n_ = 8732 - tail
        h64 = max(rel(loc.detach()[:, :n_], l64[:, :n_]), rel(conf.detach()[:, :n_], c64[:, :n_]))
        r64 = max(rel(lo[:, :n_], l64[:, :n_]), rel(co[:, :n_], c64[:, :n_]))
        print(f'{name}: HIP vs fp32 oracle {max(e_l, e_c):.2e}; HIP vs float64 {h64:.2e}; fp32 oracle vs float64 {r64:.2e}')
        assert h64 < TOL and h64 <= 2.0 * r64 + 2e-5, (name, e_l, e_c, h64, r64)
        # only the configuration this arbiter was written for may need it: a regression that pushes another config past the
        # fp32 gate must not be absorbed here
        assert name in FLOAT64_ARBITER_ALLOWED, f'{name}: HIP vs fp32 oracle {max(e_l, e_c):.2e} exceeds {TOL:.0e} (float64 arbiter reserved for {FLOAT64_ARBITER_ALLOWED})'
    assert rel(loc.detach(), lo) < 2e-2 and rel(conf.detach(), co) < 2e-2
    after = net.state_dict()
    for k, v in upd.items():
        assert rel(after[k], v) < (1e-2 if k.startswith(('bn_fuse_61', 'bn_fuse_list1.3', 'extras.15')) else TOL), k
    for k in [k for k in g.files if k.startswith(f'{name}.after.')]:
        assert rel(after[k[len(name) + 7:]], g[k]) < TOL, k
    # backward
    ((loc * r1.to(dev)).sum() + (conf * r2.to(dev)).sum()).backward()
    named = dict(net.named_parameters())
    from aten_shadow import shadow_param_grads
    sg = dict(zip([k for k, _ in net.named_parameters()], shadow_param_grads(net, x.to(dev), r1.to(dev), r2.to(dev))))

    def l2rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / b.norm())
    errs = {k: l2rel(named[k].grad, sg[k]) for k in gkeys}
    print(name, 'HIP vs ATen backward (L2-relative)', {k: f'{v:.1e}' for k, v in errs.items()})
    # (per tensor, relative L2: a handful of ReLU / max-pool decisions flip between two fp32 implementations, see
    # test_backward_gradients -- which also covers the sigma gates: scalars that are one heavily cancelling sum over the whole map,
    # 1e-1 run-to-run noise at B = 2)
    assert all(np.isfinite(v) and v < 2e-2 for k, v in errs.items()), errs
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    unused = [k for k, p in named.items() if p.grad is None]
    assert unused == [k for k in unused if sg[k] is None], unused          # exactly the parameters the reference graph leaves out


def test_visualize_outputs(dev):
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args = NETS['gssdpp']
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=1111)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    x = synth.synth_images(2, seed=5)
    with torch.no_grad():
        out, offs, attnb, attn = net(x.to(dev), visualize=True)
        taps = {}
        O.gssd_forward(sd, x, training=True, taps=taps, **flags)
    assert len(offs) == 1 and offs[0].shape == (2, 72, 38, 38)
    assert rel(offs[0], taps['dcn0.offset']) < TOL
    assert [a.shape[1] for a in attnb] == [1444, 361, 100, 25, 9, 1] and len(attn) == 6
    assert all(abs(a.sum(-1) - 1).max() < 1e-4 for a in attnb + attn)
    # the materialised maps (extra launches of the visualize plan; the output path is flash-style and never reads them)
    # (a probability is exp(logit - max): an fp32 logit of magnitude ~1e2 carries ~1e-5 of absolute rounding, which the exp turns
    # into the same RELATIVE error of the probability -- hence 1e-3 here, 1e-4 on everything downstream of the row sums)
    errs = [(rel(attnb[i], taps[f'sab{i}.attn']), rel(attn[i], taps[f'sa{i}.attn'])) for i in range(6)]
    print('attention map errors', errs)
    assert max(max(e) for e in errs) < 1e-3, errs
    # ... and the visualize plan's train tuple equals the plain plan's (eval mode: no forward mutates spectral norm's u / v)
    net.eval()
    with torch.no_grad():
        out_vis = net(x.to(dev), visualize=True)[0]
        out_plain = net(x.to(dev))
    assert rel(out_vis[0], out_plain[0]) < 1e-6 and rel(out_vis[1], out_plain[1]) < 1e-6


def test_visualize_outputs_pooled_keys(dev):
    """visualize=True with max_pool_factor = 3 (layers/self_attn.py:57-59): the maps are [B, N, Nk], Nk = max(H // 3, 1)^2 pooled keys;
    test phase (eval-mode BatchNorm, Detect) runs on the same plan."""
    from models.ssd_multiphase_custom_group import build_ssd
    flags, args, _ = FLAG_NETS['mpf3_sa']
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=1111)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    x = synth.synth_images(2, seed=5)
    with torch.no_grad():
        out, offs, attnb, attn = net(x.to(dev), visualize=True)
        taps = {}
        O.gssd_forward(sd, x, training=True, taps=taps, **flags)
    assert offs == [] and [tuple(a.shape[1:]) for a in attn] == [(1444, 144), (361, 36), (100, 9), (25, 1), (9, 1), (1, 1)]
    assert all(abs(a.sum(-1) - 1).max() < 1e-4 for a in attnb + attn)
    errs = [(rel(attnb[i], taps[f'sab{i}.attn']), rel(attn[i], taps[f'sa{i}.attn'])) for i in range(6)]
    print('pooled attention map errors', errs)
    assert max(max(e) for e in errs) < 1e-3, errs
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    with torch.no_grad():
        net(x.to(dev))
    net_t = build_ssd('test', 300, 2, *args)
    net_t.load_state_dict(net.state_dict())
    net_t = net_t.to(dev).eval()
    with torch.no_grad():
        det = net_t(x.to(dev))
        lo, co, _ = O.gssd_forward({k: v.cpu() for k, v in net.state_dict().items()}, x, training=False, **flags)
    ref = O.detect(2, 0, 200, 0.01, 0.45, lo.numpy(), O.softmax_scores(co.numpy()), O.prior_box())
    assert det.shape == (2, 2, 200, 5)
    assert np.array_equal(det.cpu().numpy()[..., 0] > 0, ref[..., 0] > 0)
    # (B = 2: the running variances of the 1 x 1 map's BatchNorms come from two values per channel and can be ~eps, so eval mode
    # divides by sqrt(~eps) there and a 1e-6 forward difference becomes 1e-3 on that map's four boxes)
    assert same_detections(det.cpu().numpy(), ref, 2e-3)


@pytest.mark.parametrize('name', ['gssd', 'gssdpp'])
def test_full_size_properties(dev, name):
    """BASELINE.json configs[1] / configs[2] size (B=32, the benchmarked one): size-independent properties instead of a
    12 s (GSSD) / 30 s (GSSD++) CPU forward."""
    from models.ssd_multiphase_custom_group import build_ssd
    from layers.modules import MultiBoxLoss
    flags, args = NETS[name]
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth.synth_state_dict(shapes, seed=1111))
    net = net.to(dev)
    x = synth.synth_images(32, seed=11).to(dev)
    tg = synth.synth_targets(32, seed=11)
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    with torch.no_grad():
        net.train()
        loc, conf, pri = net(x)
        ll, lc = crit((loc, conf, pri), tg)
    assert torch.isfinite(loc).all() and torch.isfinite(conf).all()
    # loss kernels at full size vs the oracle fed the same predictions
    rl, rc = O.multibox_loss(loc.cpu().numpy(), conf.cpu().numpy(), pri.cpu().numpy(), [t.numpy() for t in tg])
    assert rel(ll, rl) < TOL and rel(lc, rc) < TOL
    # eval mode: images are independent -> image i of the batch-32 run == the same image in a batch of 2
    with torch.no_grad():
        net.eval()
        l32, c32, _ = net(x)
        l2, c2, _ = net(x[5:7])
    assert rel(l32[5:7], l2) < 1e-5 and rel(c32[5:7], c2) < 1e-5
    # Detect on the full batch: descending scores, survivors pairwise IoU <= 0.45, matches the oracle
    from gssd import ops
    det, keep, cnt = ops.detect(l32, c32, pri, 2, conf_is_logits=True, want_keep=True)
    det = det.cpu().numpy()
    sc = det[:, 1, :, 0]
    assert (np.diff(sc, axis=1) <= 0).all() and (det[:, 0] == 0).all()
    b = 3
    n = int(cnt[b, 1])
    bx = det[b, 1, :n, 1:]
    iou = O.jaccard(bx, bx)
    assert (iou[np.triu_indices(n, 1)] <= 0.45 + 1e-6).all()
    ref = O.detect(2, 0, 200, 0.01, 0.45, l32[b:b + 1].cpu().numpy(), O.softmax_scores(c32[b:b + 1].cpu().numpy()),
                   pri.cpu().numpy())
    assert np.array_equal(det[b:b + 1], ref)


@pytest.mark.parametrize('name', ['gssd', 'gssdpp'])
def test_full_size_parity(dev, name, golden):
    """the gate of north_star (<= 1e-4 per tensor) at the BENCHMARKED batch, B = 32 (configs[1] / configs[2]), on the
    FULL loc / conf tensors -- every prior, including the four of the 1 x 1 map (8728 .. 8731) that the batch-4 sweep arbitrates in
    float64 -- and both losses, against the fixture-pinned fp32 oracle (models/ssd_multiphase_custom_group.py:217-400 restated in
    oracle/gssd_oracle.py).  Three weight / image seed pairs; no float64 arbitration, no carve-out.  At B = 32 the extras' train-mode
    BatchNorms (models/...group.py:350-372) see 32 values per channel on the 1 x 1 map instead of 4.  Every number is printed and
    written to gpurun_out/parity_b32_<name>.txt (copied to profiles/ per round)."""
    from models.ssd_multiphase_custom_group import build_ssd
    from layers.modules import MultiBoxLoss
    flags, args = NETS[name]
    net = build_ssd('train', 300, 2, *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net = net.to(dev).train()
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    pri = O.prior_box()
    NT = 8728
    lines, worst = [], 0.0
    gref = golden('e2e_b32')

    def part(a, b, lo_, hi_):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a[:, lo_:hi_] - b[:, lo_:hi_]).abs().max() / b.abs().max())
    for wseed, xseed in ((1111, 11), (2024, 12), (31337, 14)):
        sd = synth.synth_state_dict(shapes, seed=wseed)
        net.load_state_dict(sd)
        x = synth.synth_images(32, seed=xseed)
        tg = synth.synth_targets(32, seed=xseed)
        with torch.no_grad():
            loc, conf, _ = net(x.to(dev))
            ll, lc = crit((loc, conf, torch.from_numpy(pri).to(dev)), tg)
            lo, co, upd = O.gssd_forward(sd, x, **flags)
        rl, rc = O.multibox_loss(lo.numpy(), co.numpy(), pri, [t.numpy() for t in tg])[:2]
        e = dict(loc=rel(loc, lo), conf=rel(conf, co), loss_l=rel(ll, rl), loss_c=rel(lc, rc),
                 loc_1x1=part(loc, lo, NT, None), conf_1x1=part(conf, co, NT, None))
        after = net.state_dict()
        e['state'] = max(rel(after[k], v) for k, v in upd.items())
        if (wseed, xseed) == (int(gref['wseed']), int(gref['xseed'])):
            # ... and against the imported REFERENCE's own B = 32 outputs for this seed pair (tests/golden/make_golden_b32.py): 2 048 sampled
            # values of each tensor, every image's 1 x 1-map priors, both losses, four mutated buffers -- the benchmarked kernel mix
            # (conv_x6 / dcn_x6 take their launches only from M >= 4096) tied to the reference directly, not only through the oracle
            l, c = loc.cpu().numpy(), conf.cpu().numpy()
            e['ref_loc'] = float(np.abs(l.reshape(-1)[gref[f'{name}.loc_idx']] - gref[f'{name}.loc_val']).max() / gref[f'{name}.loc_absmax'])
            e['ref_conf'] = float(np.abs(c.reshape(-1)[gref[f'{name}.conf_idx']] - gref[f'{name}.conf_val']).max() / gref[f'{name}.conf_absmax'])
            e['ref_loc_1x1'] = float(np.abs(l[:, NT:] - gref[f'{name}.loc_1x1']).max() / gref[f'{name}.loc_absmax'])
            e['ref_conf_1x1'] = float(np.abs(c[:, NT:] - gref[f'{name}.conf_1x1']).max() / gref[f'{name}.conf_absmax'])
            e['ref_loss_l'], e['ref_loss_c'] = rel(ll, gref[f'{name}.loss'][0]), rel(lc, gref[f'{name}.loss'][1])
            e['ref_state'] = max(rel(after[k], gref[f'{name}.after.{k}']) for k in
                                 ('vgg.1.running_mean', 'vgg.41.running_var', 'bn_fuse_11.running_mean', 'extras.15.running_var'))
        lines.append(f'{name} B=32 weights {wseed} images {xseed}: ' + ' '.join(f'{k} {v:.2e}' for k, v in e.items()))
        worst = max(worst, *e.values())
    lines.append(f'{name} B=32: worst of 3 seed pairs {worst:.2e} (gate {TOL:.0e}; full tensors incl. priors 8728..8731, both losses, '
                 f'every mutated buffer)')
    print('\n'.join(lines))
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, f'parity_b32_{name}.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
    assert worst < TOL, lines


def test_vanilla_ssd_config0(dev, golden):
    """BASELINE.json configs[0] on the HIP engine: vanilla VGG-SSD300 (models/ssd.py:48-108), 1 phase, batch 2, forward +
    MultiBoxLoss against the reference fixture and the oracle, and the backward (HIP plan) against CPU autograd through the
    oracle graph: finite, and equal within fp32 noise."""
    from models.ssd import build_ssd
    from layers.modules import MultiBoxLoss
    g = golden('e2e')
    net = build_ssd('train', 300, 2)
    keys = sorted(net.state_dict().keys())
    assert keys == [str(k) for k in g['ssd.keys']]
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=1111)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    x = synth.synth_images(2, seed=6, channels=3)
    tg = synth.synth_targets(4, 5)[:2]
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    loc, conf, pri = net(x.to(dev))
    assert loc.requires_grad and loc.shape == (2, 8732, 4) and conf.shape == (2, 8732, 2)
    ll, lc = crit((loc, conf, pri), tg)
    l, c = loc.detach().cpu().numpy().reshape(-1), conf.detach().cpu().numpy().reshape(-1)
    assert rel(l[g['ssd.loc_idx']], g['ssd.loc_val']) < TOL and rel(c[g['ssd.conf_idx']], g['ssd.conf_val']) < TOL
    assert rel(ll, g['ssd.loss'][0]) < TOL and rel(lc, g['ssd.loss'][1]) < TOL
    (ll + lc).backward()
    named = dict(net.named_parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    # oracle: full tensors and CPU autograd of the same loss
    sdg = {k: v.clone().requires_grad_() for k, v in sd.items()}
    lo, co = O.vanilla_ssd_forward(sdg, x)
    assert rel(loc, lo) < TOL and rel(conf, co) < TOL
    # d(loss)/d(loc, conf) from the HIP loss kernels (tested on their own in test_multibox_loss), pushed through CPU autograd
    loc2 = loc.detach().clone().requires_grad_()
    conf2 = conf.detach().clone().requires_grad_()
    l2, c2 = crit((loc2, conf2, pri), tg)
    (l2 + c2).backward()
    torch.autograd.backward((lo, co), (loc2.grad.cpu(), conf2.grad.cpu()))

    def l2rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / max(float(b.norm()), 1e-30))
    errs = {k: l2rel(named[k].grad, sdg[k].grad) for k in named}
    print('vanilla SSD gradient L2-relative errors vs CPU autograd:', {k: f'{v:.1e}' for k, v in errs.items()})
    # ReLU masks / pool arg-maxes flip between two fp32 implementations (see test_backward_gradients): relative L2 per tensor
    assert max(errs.values()) < 1e-2, {k: v for k, v in errs.items() if v >= 1e-2}


def test_reference_driver_sequence(dev, tmp_path):
    """The call sequence of the reference driver (train_lesion_multiphase_v2.py:117-120 default CUDA tensor type; :142-145
    positional 15-argument build_ssd; :587-601 weights_init on extras / loc / conf, .cuda(), nn.DataParallel, deepcopy per
    cross-validation fold; :606-627 SGD with a separate DCN parameter group selected by the ``module.dcn_list`` name prefix;
    :242-253 forward, zero_grad, MultiBoxLoss, backward, check_grad_norm, clip_grad_norm_, step; :647 module.load_weights;
    :399-407 strict state-dict hand-over to a test-phase twin) -- replayed unchanged against the drop-in modules."""
    import copy
    import torch.nn as nn
    from models.ssd_multiphase_custom_group import build_ssd, weights_init
    from layers.modules import MultiBoxLoss
    from gssd._lib import GssdError
    torch.set_default_tensor_type('torch.cuda.FloatTensor')
    try:
        torch.manual_seed(1111)                                      # the driver seeds too (:4); keeps the test independent of test order
        torch.cuda.manual_seed(1111)
        net = build_ssd('train', 300, 2, True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)
        net.extras.apply(weights_init)
        net.loc.apply(weights_init)
        net.conf.apply(weights_init)
        net = net.cuda()
        net = torch.nn.DataParallel(net)
        net_cv = [copy.deepcopy(net) for _ in range(2)]
        del net
        opts = []
        for m in net_cv:
            named = list(m.named_parameters())
            p_dcn = [p_ for k, p_ in named if k.startswith('module.dcn_list')]
            p_rest = [p_ for k, p_ in named if not k.startswith('module.dcn_list')]
            assert len(p_dcn) == 4 and len(p_rest) + 4 == len(named)
            opts.append(torch.optim.SGD([{'params': p_rest}, {'params': p_dcn, 'lr': 1e-4}], lr=1e-3, momentum=0.9, weight_decay=5e-4))
        criterion = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
        images = synth.synth_images(4, seed=41).cuda()
        targets = [t.cuda() for t in synth.synth_targets(4, seed=41)]
        losses = []
        for idx in range(2):
            net_cv[idx].train()
            for it in range(2):
                out = net_cv[idx](images)
                opts[idx].zero_grad()
                loss_l, loss_c = criterion(out, targets)
                loss = loss_l + loss_c
                loss.backward()
                grad_norm = sum(float(p_.grad.data.norm(2)) ** 2 for p_ in net_cv[idx].parameters() if p_.grad is not None) ** 0.5
                assert np.isfinite(grad_norm) and grad_norm > 0
                nn.utils.clip_grad_norm_(net_cv[idx].parameters(), 10.0)
                opts[idx].step()
                losses.append(float(loss))
        # the folds start as identical copies (BatchNorm sums and split-K heads accumulate with atomics: equal to rounding, not bitwise)
        assert all(np.isfinite(losses)) and losses[1] < losses[0] and abs(losses[0] - losses[2]) <= 1e-5 * abs(losses[0]), losses
        # checkpoint round trip through the tolerant loader, then strict hand-over to the test-phase twin
        path = str(tmp_path / 'ck.pth')
        torch.save(net_cv[0].state_dict(), path)                     # keys carry DataParallel's 'module.' prefix
        net_cv[1].module.load_weights(path)
        for (k, a), (_, b) in zip(net_cv[0].module.state_dict().items(), net_cv[1].module.state_dict().items()):
            assert torch.equal(a, b), k
        net_t = build_ssd('test', 300, 2, True, 4, 4, 1, True, True, True, 1, 4, True, False, 1).cuda()
        net_t.load_state_dict(net_cv[0].module.state_dict())          # strict
        net_t.eval()
        with torch.no_grad():
            det = net_t(images)
        assert det.shape == (4, 2, 200, 5) and torch.isfinite(det).all()
        # replicas over several devices are refused with a pointer to the one-process-per-GPU path, not silently mis-run
        with pytest.raises(GssdError):
            net_cv[0].module._replicate_for_data_parallel()
    finally:
        torch.set_default_tensor_type('torch.FloatTensor')


@pytest.mark.parametrize('name', ['gssdpp', 'vanilla'])
def test_graph_replay_equals_eager(dev, name):
    """From its third run on a launch plan replays itself from hipGraphs (static buffers, descriptors by value).  The replayed
    forward must equal the eager one bit for bit -- same kernels, same order -- including the state a training forward mutates
    (BatchNorm running statistics, spectral norm's u / v), and the event-bracketed variant (graph segments around eager launches)
    that bench.py's live roofline measurement uses."""
    if name == 'vanilla':
        from models.ssd import build_ssd as bs
        net = bs('train', 300, 2)
        x = synth.synth_images(2, seed=6, channels=3).to(dev)
    else:
        from models.ssd_multiphase_custom_group import build_ssd as bs
        net = bs('train', 300, 2, *NETS['gssdpp'][1])
        x = synth.synth_images(3, seed=6).to(dev)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
    import copy
    net.load_state_dict(sd)
    net = net.to(dev).train()
    twin = copy.deepcopy(net)
    os.environ['GSSD_NO_GRAPH'] = '0'
    outs = []
    with torch.no_grad():
        for it in range(5):
            outs.append(net(x))
    plan = net._engine._last_plan
    # runs 3..5 were graph replays: run 3 over the plan's static input buffer, runs 4-5 (the same input buffer again) zero-copy
    assert plan._runs == 5 and sorted(str(k[0]) if isinstance(k, tuple) and k else 'generic' for k in plan._graphs) == ['direct', 'generic']
    # the twin runs the same five training forwards eagerly
    import gssd.plan_common as E          # (plan_exec.run reads the switch through this module)
    E.USE_GRAPH = False
    try:
        with torch.no_grad():
            ref = [twin(x) for _ in range(5)]
    finally:
        E.USE_GRAPH = True
    assert getattr(twin._engine._last_plan, '_graphs', None) is None
    # split-K heads accumulate with fp32 atomics (order varies run to run): compare everything else exactly, loc / conf tightly
    for a, b in zip(outs, ref):
        assert rel(a[0], b[0]) < 1e-6 and rel(a[1], b[1]) < 1e-6
    for (k, va), (_, vb) in zip(net.state_dict().items(), twin.state_dict().items()):
        assert torch.equal(va, vb), k
    # the zero-copy graph reads the caller's buffer IN PLACE (new contents at the old address are seen), another buffer takes the generic graph
    with torch.no_grad():
        x.mul_(0.5)
        o_a, r_a = net(x), twin(x)
        x2 = (x * 1.7).clone()
        o_b, r_b = net(x2), twin(x2)
    assert rel(o_a[0], r_a[0]) < 1e-6 and rel(o_a[1], r_a[1]) < 1e-6 and rel(o_b[0], r_b[0]) < 1e-6 and rel(o_b[1], r_b[1]) < 1e-6
    assert float((o_a[0] - o_b[0]).abs().max()) > 0
    if name == 'gssdpp':
        class EL(list):
            only = {'dcn_x6<128x256>', 'dcn_fused<128x256>'}
        ev = EL()
        net.__dict__['_events'] = ev
        with torch.no_grad():
            o6 = net(x)
            r6 = twin(x)
        net.__dict__['_events'] = None
        torch.cuda.synchronize()
        assert len(ev) == 1 and ev[0][1].elapsed_time(ev[0][2]) > 0 and len(plan._graphs) == 3
        assert rel(o6[0], r6[0]) < 1e-6 and rel(o6[1], r6[1]) < 1e-6
        # round 6: event-record NODES inside the replayed graph (gssd_event_record_node; bench.py's measurement of kernels launched many times per
        # step): nothing is cut out of the graph, every launch of the named instances gets a (start, stop) pair that holds the last replay's times
        tags = sorted({st.tag[0] for st in plan.steps if st.tag is not None and st.tag[0].startswith('conv_igemm')})
        n_launch = sum(1 for st in plan.steps if st.tag is not None and st.tag[0] in tags[:2])

        class EN(list):
            only, nodes = None, set(tags[:2])
        for _ in range(2):
            ev = EN()
            net.__dict__['_events'] = ev
            with torch.no_grad():
                o7 = net(x)
                r7 = twin(x)
            torch.cuda.synchronize()
            ms = [e0.elapsed_time(e1) for _, e0, e1 in ev]
            assert len(ev) == n_launch >= 2 and all(0 < t < 50 for t in ms), (len(ev), n_launch, ms)
            assert rel(o7[0], r7[0]) < 1e-6 and rel(o7[1], r7[1]) < 1e-6
        net.__dict__['_events'] = None
        assert len(plan._graphs) == 4


@pytest.mark.parametrize('name', ['gssdpp', 'vanilla'])
def test_forward_is_run_to_run_deterministic(dev, name):
    """loc / conf carry identical bits from run to run (eval mode: nothing mutates between the calls): the heads' split-K slices are
    summed in a fixed order (GSSD_CONV_HEADS_SLICES + gssd_heads_reduce_f32), not by atomics; eager and hipGraph replay alike."""
    if name == 'vanilla':
        from models.ssd import build_ssd
        net = build_ssd('train', 300, 2)
        x = torch.from_numpy(np.random.default_rng(3).normal(size=(2, 3, 300, 300)).astype(np.float32)).to(dev)
    else:
        from models.ssd_multiphase_custom_group import build_ssd
        net = build_ssd('train', 300, 2, *NETS['gssdpp'][1])
        x = synth.synth_images(3, seed=77).to(dev)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
    net = net.to(dev).eval()
    outs = []
    with torch.no_grad():
        for _ in range(5):                                   # runs 1-2 eager, 3+ replayed
            loc, conf, _ = net(x)
            outs.append((loc.clone(), conf.clone()))
    for loc, conf in outs[1:]:
        assert torch.equal(loc, outs[0][0]) and torch.equal(conf, outs[0][1])


def test_cpu_input_fails_loudly():
    from models.ssd_multiphase_custom_group import build_ssd
    from gssd._lib import GssdError
    net = build_ssd('train', 300, 2, True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)
    with pytest.raises(GssdError):
        net(torch.zeros(2, 12, 300, 300))
