"""Kernel-level parity of csrc/conv_thin_x6.hip (round 6): the patch-staged direct 3x3 conv with three-plane bf16 operands that takes conv1_2
(4 x 16 -> 16 channels), conv2_1 (16 -> 32) and conv2_2 (32 -> 32) of the fp32 mode.  Every launch form against a float64 convolution on the
CPU: plain + bias + batch sums (one array / 8 replicas), the fused producer BatchNorm + ReLU with scales of both signs, the pooled-raw
epilogue (bit for bit the pooled image of the plain launch, identical batch sums), ragged tiles (maps that are no multiple of 8 x 16,
odd maps with ceil-mode pooling, non-square).  Gate: fp32-equivalent products of the DIRECT convolution -- closer to float64 than the fp32-MFMA
implicit GEMM on the same descriptor (e <= 1.5 e_fp32 + 1e-7, < 1e-6).  The round-5 kernels these launches used to take (conv_thin_wino.hip,
conv_wino.hip<32>, conv_thin.hip<16,32>) stay reachable with GSSD_THIN_X6=0: the last test re-runs their kernel-level tests in that mode."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, W, Cin, Cout          (4 groups, 3x3 / stride 1 / pad 1)
    (2, 83, 83, 64, 64),       # conv1_2 class, ragged 8 x 16 tiles (83 = 10 * 8 + 3 = 5 * 16 + 3), odd map
    (1, 160, 176, 64, 64),     # whole tiles, non-square, several tiles per persistent workgroup
    (3, 75, 75, 64, 128),      # conv2_1 class at the smallest map the kernel takes
    (2, 150, 150, 64, 128),    # conv2_1 at its own size
    (2, 78, 78, 128, 128),     # conv2_2 class: a workgroup owns a PAIR of groups
    (1, 150, 150, 128, 128),   # conv2_2 at its own size
    (9, 90, 76, 64, 64),       # more tiles than one sweep of 256 workgroups: the prefetch of the next tile crosses images
]


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.abs().max()), 1e-30))


@pytest.mark.parametrize('case', CASES)
def test_conv_thin_x6_forms_vs_float64(case):
    from gssd import ops, _lib
    lib = _lib.lib
    assert os.environ.get('GSSD_THIN_X6') is None
    B, H, W, Cin, Cout = case
    g = 4
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(sum(case))
    cin_g = Cin // g
    x = torch.from_numpy(rng.normal(0.1, 1.0, size=(B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, cin_g, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = ops.pack_weight(w.to(dev))
    kw = dict(B=B, H=H, W=W, in_stride=Cin, cin_g=cin_g, Cout=Cout, groups=g, k=3, stride=1, pad=1, bias=b.to(dev))
    n_px = B * H * W

    def launch(o, expect=1, **extra):
        d, _, _ = ops.make_conv_desc(xd, wp, o, **{**kw, **extra})
        assert lib.gssd_conv_thin_x6_takes(C.byref(d)) == expect
        _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
        torch.cuda.synchronize()

    def stat_err(stats, R, r64):
        s = stats.view(max(R, 1), 2 * Cout).sum(0).cpu()
        return max(float((s[:Cout] - r64.sum((0, 1, 2))).abs().max() / (n_px * float(r64.abs().max()))),
                   float((s[Cout:] - (r64 * r64).sum((0, 1, 2))).abs().max() / (n_px * float(r64.abs().max()) ** 2)))

    # ---- plain + bias + batch sums -------------------------------------------------------------------------------------------
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1, 1, g).permute(0, 2, 3, 1).contiguous()
    outs = {}
    for R in (0, 8):
        y = torch.full((B, H, W, Cout), float('nan'), device=dev)
        stats = torch.zeros(max(R, 1) * 2 * Cout, dtype=torch.float64, device=dev)
        launch(y, stats=stats, stats_rep=R)
        assert torch.isfinite(y).all()
        outs[R] = (y, stats)
        assert stat_err(stats, R, ref) < 2e-7
    assert torch.equal(outs[0][0], outs[8][0])
    y_plain, stats_plain = outs[0]
    # the fp32-MFMA implicit GEMM on the same descriptor (a resid that is all zeros keeps the launch away from every thin / Winograd kernel)
    y32 = torch.empty(B, H, W, Cout, device=dev)
    launch(y32, expect=0, resid=torch.zeros(B, H, W, Cout, device=dev))
    e6, e32 = rel(y_plain.cpu(), ref), rel(y32.cpu(), ref)
    print(f'{case} plain: thin_x6 vs float64 {e6:.2e}; fp32-MFMA implicit GEMM {e32:.2e}')
    assert e6 < 1e-6 and e6 <= 1.5 * e32 + 1e-7

    # ---- the same launch flagged GSSD_CONV_F16_OK (a forward launch on a materialised activation map): two fp16 planes, three MFMAs per product -----
    y_f16 = torch.full((B, H, W, Cout), float('nan'), device=dev)
    stats_f16 = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    launch(y_f16, stats=stats_f16, flags=_lib.CONV_F16_OK)
    ef = rel(y_f16.cpu(), ref)
    print(f'{case} plain, fp16 planes: {ef:.2e}')
    assert ef < 1e-6 and ef <= 1.5 * e32 + 1e-7 and stat_err(stats_f16, 0, ref) < 2e-7
    yp = torch.full((B, (H + 1) // 2, (W + 1) // 2, Cout), float('nan'), device=dev)
    launch(yp, flags=_lib.CONV_POOL2 | _lib.CONV_F16_OK, pool_sign=torch.ones(Cout, device=dev))
    assert torch.equal(yp, F.max_pool2d(y_f16.permute(0, 3, 1, 2), 2, 2, 0, ceil_mode=True).permute(0, 2, 3, 1))

    # ---- fused producer BatchNorm + ReLU, scales of both signs; zero padding AFTER the transform -------------------------------------
    scv = torch.from_numpy(rng.uniform(0.2, 1.5, size=Cin).astype(np.float32)) * torch.from_numpy(rng.choice([-1.0, 1.0], size=Cin).astype(np.float32))
    shv = torch.from_numpy(rng.normal(size=Cin).astype(np.float32))
    pdv = torch.where(scv > 0, torch.full_like(scv, -3.0e38), torch.full_like(scv, 3.0e38))
    act64 = torch.relu(torch.addcmul(shv.double().view(1, -1, 1, 1), x.double(), scv.double().view(1, -1, 1, 1)))
    refx = F.conv2d(act64, w.double(), b.double(), 1, 1, 1, g).permute(0, 2, 3, 1).contiguous()
    xf = dict(in_scale=scv.to(dev), in_shift=shv.to(dev), in_pad=pdv.to(dev))
    y_xf = torch.full((B, H, W, Cout), float('nan'), device=dev)
    stats_xf = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    launch(y_xf, stats=stats_xf, **xf)
    e = rel(y_xf.cpu(), refx)
    print(f'{case} fused producer BatchNorm + ReLU: {e:.2e}')
    assert e < 1e-6 and stat_err(stats_xf, 0, refx) < 2e-7
    # ... and as a train-mode forward launches it: flagged GSSD_CONV_F16_OK (fp16 planes); its pooled form = the pooled image of this launch
    y_xf16 = torch.full((B, H, W, Cout), float('nan'), device=dev)
    stats_xf16 = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    launch(y_xf16, stats=stats_xf16, flags=_lib.CONV_F16_OK, **xf)
    e = rel(y_xf16.cpu(), refx)
    print(f'{case} fused producer BatchNorm + ReLU, fp16 planes: {e:.2e}')
    assert e < 1e-6 and stat_err(stats_xf16, 0, refx) < 2e-7
    yp = torch.full((B, (H + 1) // 2, (W + 1) // 2, Cout), float('nan'), device=dev)
    launch(yp, flags=_lib.CONV_POOL2 | _lib.CONV_F16_OK, pool_sign=torch.ones(Cout, device=dev), **xf)
    assert torch.equal(yp, F.max_pool2d(y_xf16.permute(0, 3, 1, 2), 2, 2, 0, ceil_mode=True).permute(0, 2, 3, 1))

    # ---- pooled-raw epilogue: max / min by the sign of the consumer BatchNorm's weight, batch sums of the FULL map -------------------------
    gamma = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    gamma[5] = 0.0
    Hp, Wp = (H + 1) // 2, (W + 1) // 2

    def pool_by_sign(raw_nhwc):
        r_ = raw_nhwc.permute(0, 3, 1, 2)
        mx, mn = F.max_pool2d(r_, 2, 2, 0, ceil_mode=True), -F.max_pool2d(-r_, 2, 2, 0, ceil_mode=True)
        return torch.where(gamma.to(r_.dtype).view(1, -1, 1, 1) >= 0, mx, mn).permute(0, 2, 3, 1).contiguous()
    for name, extra, full, full_stats, r64 in (('pool', {}, y_plain, stats_plain, ref), ('pool + producer', xf, y_xf, stats_xf, refx)):
        yp = torch.full((B, Hp, Wp, Cout), float('nan'), device=dev)
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
        launch(yp, stats=stats, flags=_lib.CONV_POOL2, pool_sign=gamma.to(dev), **extra)
        assert torch.equal(yp.cpu(), pool_by_sign(full.cpu())), name            # bit for bit the pooled image of the plain launch
        assert rel(stats, full_stats) < 1e-13, name                             # the same fp32 additions in the same order
        assert rel(yp.cpu(), pool_by_sign(r64)) < 1e-6, name


@pytest.mark.parametrize('case', [(2, 75, 75), (1, 83, 90)])
def test_conv_thin_x6_conv3_1_class(case):
    """conv3_1's shape class (4 x 32 -> 64 channels, round 6): taken only with GSSD_CONV_F16_OK (its three-plane bf16 instance spills) -- plain, with the
    fused producer BatchNorm + ReLU, batch sums, pooled epilogue -- against float64 and the fp32 Winograd / implicit-GEMM kernels."""
    from gssd import ops, _lib
    lib = _lib.lib
    B, H, W = case
    Cin, Cout, g = 128, 256, 4
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(sum(case))
    x = torch.from_numpy(rng.normal(0.1, 1.0, size=(B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, Cin // g, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = ops.pack_weight(w.to(dev))
    kw = dict(B=B, H=H, W=W, in_stride=Cin, cin_g=Cin // g, Cout=Cout, groups=g, k=3, stride=1, pad=1, bias=b.to(dev))
    n_px = B * H * W

    def launch(o, expect, **extra):
        d, _, _ = ops.make_conv_desc(xd, wp, o, **{**kw, **extra})
        assert lib.gssd_conv_thin_x6_takes(C.byref(d)) == expect
        _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
        torch.cuda.synchronize()
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1, 1, g).permute(0, 2, 3, 1).contiguous()
    y32 = torch.empty(B, H, W, Cout, device=dev)
    launch(y32, 0)                                                     # unflagged: the fp32-MFMA implicit GEMM
    y = torch.full((B, H, W, Cout), float('nan'), device=dev)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    launch(y, 1, stats=stats, flags=_lib.CONV_F16_OK)
    e6, e32 = rel(y.cpu(), ref), rel(y32.cpu(), ref)
    print(f'{case} conv3_1 class: thin_x6 {e6:.2e}; implicit GEMM {e32:.2e}')
    assert torch.isfinite(y).all() and e6 < 1e-6 and e6 <= 1.5 * e32 + 1e-7
    s = stats.cpu()
    assert float((s[:Cout] - ref.sum((0, 1, 2))).abs().max() / (n_px * float(ref.abs().max()))) < 2e-7
    assert float((s[Cout:] - (ref * ref).sum((0, 1, 2))).abs().max() / (n_px * float(ref.abs().max()) ** 2)) < 2e-7
    scv = torch.from_numpy(rng.uniform(0.2, 1.5, size=Cin).astype(np.float32)) * torch.from_numpy(rng.choice([-1.0, 1.0], size=Cin).astype(np.float32))
    shv = torch.from_numpy(rng.normal(size=Cin).astype(np.float32))
    pdv = torch.where(scv > 0, torch.full_like(scv, -3.0e38), torch.full_like(scv, 3.0e38))
    act64 = torch.relu(torch.addcmul(shv.double().view(1, -1, 1, 1), x.double(), scv.double().view(1, -1, 1, 1)))
    refx = F.conv2d(act64, w.double(), b.double(), 1, 1, 1, g).permute(0, 2, 3, 1).contiguous()
    xf = dict(in_scale=scv.to(dev), in_shift=shv.to(dev), in_pad=pdv.to(dev))
    yx = torch.full((B, H, W, Cout), float('nan'), device=dev)
    launch(yx, 1, flags=_lib.CONV_F16_OK, **xf)
    assert rel(yx.cpu(), refx) < 1e-6
    yp = torch.full((B, (H + 1) // 2, (W + 1) // 2, Cout), float('nan'), device=dev)
    launch(yp, 1, flags=_lib.CONV_POOL2 | _lib.CONV_F16_OK, pool_sign=torch.ones(Cout, device=dev), **xf)
    assert torch.equal(yp, F.max_pool2d(yx.permute(0, 3, 1, 2), 2, 2, 0, ceil_mode=True).permute(0, 2, 3, 1))


def test_round5_kernels_of_these_layers_still_pass_with_the_switch_off():
    """GSSD_THIN_X6=0 (the switch row of tests/test_gpu_switches.py runs a whole training step in that mode): the kernel-level tests of
    conv_thin_wino.hip / conv_wino.hip<32> / conv_thin.hip on the shapes this kernel now takes by default."""
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_kernels.py'), '-q', '-x', '-k',
                        'test_conv_winograd or test_conv_igemm or test_batch_sum_replicas'], capture_output=True, text=True, timeout=1200,
                       env=dict(os.environ, GSSD_THIN_X6='0'), cwd=ROOT)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
