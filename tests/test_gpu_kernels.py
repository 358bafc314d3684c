"""GPU parity tests, kernel level: the conv family (implicit GEMM, Winograd, thin / patch-staged, three-plane bf16 forms), BatchNorm / pool /
L2Norm / spectral-norm passes, the deformable conv in all its forms, the attention cores and the backward building blocks, each against
the CPU oracle, float64 or the reference-generated fixtures.

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


CONV_CASES = [
    # B, H, Cin, Cout, k, s, p, d, groups
    (2, 37, 16, 64, 3, 1, 1, 1, 4),      # conv1_1-like: 4 ch / group (3 real + 1 pad)
    (2, 30, 64, 64, 3, 1, 1, 1, 4),      # cout_g = 16 tile
    (2, 21, 128, 128, 3, 1, 1, 1, 4),    # cout_g = 32
    (2, 19, 256, 256, 3, 1, 1, 1, 4),    # cout_g = 64
    (2, 19, 512, 512, 3, 1, 1, 1, 4),    # cout_g = 128
    (2, 19, 512, 1024, 3, 1, 6, 6, 4),   # conv6: dilation 6
    (3, 19, 1024, 1024, 1, 1, 0, 1, 4),  # conv7: grouped 1x1
    (2, 19, 256, 512, 3, 2, 1, 1, 4),    # extras stride 2
    (2, 5, 128, 256, 3, 1, 0, 1, 4),     # extras valid 3x3 (5 -> 3)
    (5, 3, 128, 256, 3, 1, 0, 1, 4),     # 3 -> 1
    (2, 10, 512, 512, 1, 1, 0, 1, 1),    # dense 1x1 fuse
    (2, 38, 512, 108, 3, 1, 1, 1, 1),    # DCN offset conv shape (Cout not a tile multiple)
    (2, 83, 16, 64, 3, 1, 1, 1, 4),      # thin patch-staged kernel <4,16>, ragged 8x16 tiles
    (2, 80, 64, 64, 3, 1, 1, 1, 4),      # thin <16,16>
    (3, 75, 64, 128, 3, 1, 1, 1, 4),     # thin <16,32>, exact 5x25 tiles
    (1, 150, 64, 128, 3, 1, 1, 1, 4),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_igemm(dev, ops, case):
    B, H, Cin, Cout, k, s, p, d, g = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = torch.from_numpy(rng.normal(size=(B, Cin, H, H)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, Cin // g, k, k)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    ref = torch.nn.functional.conv2d(x, w, b, s, p, d, g)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    y = ops.conv2d_nhwc(nhwc(x).to(dev), w.to(dev), b.to(dev), s, p, d, g, stats=stats)
    assert rel(nchw(y), ref) < TOL
    # fused batch statistics (what BatchNorm consumes)
    n = ref.numel() / Cout
    assert rel(stats[:Cout] / n, ref.double().mean(dim=(0, 2, 3))) < 1e-5
    assert rel(stats[Cout:] / n, (ref.double() ** 2).mean(dim=(0, 2, 3))) < 1e-5


@pytest.mark.parametrize('Cin,Cout,H,k,st,pd', [(64, 64, 80, 3, 1, 1), (64, 128, 75, 3, 1, 1), (16, 64, 77, 3, 1, 1),
                                               (128, 128, 40, 3, 1, 1), (512, 512, 19, 3, 1, 1), (1024, 1024, 19, 1, 1, 0),
                                               (256, 512, 19, 3, 2, 1), (64, 256, 21, 3, 1, 1)])
def test_conv_fused_input_bn_relu(dev, ops, Cin, Cout, H, k, st, pd):
    """A consumer conv applying its producer's BatchNorm + ReLU on the fly (conv1_1 -> conv1_2 in the engine): equals
    conv2d(relu(bn(x))) with zero padding applied AFTER the transform."""
    rng = np.random.default_rng(Cin + H)
    B, g = 2, 4
    x = torch.from_numpy(rng.normal(0.2, 1.0, size=(B, Cin, H, H)).astype(np.float32))
    gm = torch.from_numpy(rng.uniform(-1.5, 1.5, size=Cin).astype(np.float32))
    bt = torch.from_numpy(rng.normal(size=Cin).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, Cin // g, k, k)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    rm, rv = torch.zeros(Cin), torch.ones(Cin)
    ref = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.batch_norm(x, rm.clone(), rv.clone(), gm, bt, True, 0.1,
                                                                              1e-5)), w, b, st, pd, 1, g)
    stats = torch.stack([x.double().sum(dim=(0, 2, 3)), (x.double() ** 2).sum(dim=(0, 2, 3))]).reshape(-1).to(dev)
    sc, sh, pdv = (torch.empty(Cin, device=dev) for _ in range(3))
    rmd, rvd = rm.to(dev), rv.to(dev)
    ops.bn_finalize(stats, B * H * H, gm.to(dev), bt.to(dev), rmd, rvd, True, sc, sh, pdv)
    wp = ops.pack_weight(w.to(dev))
    Ho = ref.shape[2]
    out = torch.empty(B, Ho, Ho, Cout, device=dev)
    d, _, _ = ops.make_conv_desc(nhwc(x).to(dev), wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // g, Cout=Cout, groups=g,
                                 k=k, stride=st, pad=pd, bias=b.to(dev), in_scale=sc, in_shift=sh, in_pad=pdv)
    ops.run_conv(d)
    assert rel(nchw(out), ref) < TOL
    rm_ref, rv_ref = rm.clone(), rv.clone()
    torch.nn.functional.batch_norm(x, rm_ref, rv_ref, gm, bt, True, 0.1, 1e-5)
    assert rel(rmd, rm_ref) < 1e-5 and rel(rvd, rv_ref) < 1e-5


@pytest.mark.parametrize('B,H,Cin,Cout,k,pd,dl', [(4, 38, 512, 512, 3, 1, 1), (12, 19, 512, 1024, 3, 6, 6), (4, 37, 1024, 512, 1, 0, 1)])
def test_conv_wgrad_fused_input(dev, ops, B, H, Cin, Cout, k, pd, dl):
    """Weight gradient of a conv that applies its producer's BatchNorm + ReLU on the fly (csrc/wgrad_slot.hip XF path for the wide
    layers): d/dw of conv2d(relu(x * scale + shift)) with zero padding AFTER the transform."""
    rng = np.random.default_rng(B * 1000 + H)
    g = 4
    x = torch.from_numpy(rng.normal(0.1, 1.0, size=(B, Cin, H, H)).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(-1.5, 1.5, size=Cin).astype(np.float32))
    sh = torch.from_numpy(rng.normal(size=Cin).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, Cin // g, k, k)).astype(np.float32)).requires_grad_()
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    y = torch.nn.functional.conv2d(a, w, None, 1, pd, dl, g)
    dy = torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32))
    y.backward(dy)
    # the pad value is what the transform maps to 0 (gssd_bn_finalize_f32 writes it): -shift / scale, or anything <= that for scale > 0
    pad = torch.where(sc != 0, -sh / sc, torch.zeros_like(sc))
    pad = torch.where((pad * sc + sh) > 0, torch.nextafter(pad, -torch.sign(sc) * torch.full_like(pad, float('inf'))), pad)
    assert float(torch.relu(pad * sc + sh).abs().max()) == 0.0
    keep = [nhwc(x).to(dev), sc.to(dev), sh.to(dev), pad.to(dev)]
    desc, _, _ = ops.make_conv_desc(keep[0], None, None, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // g, Cout=Cout, groups=g, k=k, stride=1,
                                    pad=pd, dil=dl, in_scale=keep[1], in_shift=keep[2], in_pad=keep[3])
    dw = ops.conv_wgrad(desc, nhwc(dy).to(dev), Cout, Cin // g, k)
    assert rel(dw, w.grad) < TOL


WINO_CASES = [
    # B, H, Cin, Cout, groups          (3x3 / stride 1 / pad 1)
    (2, 38, 512, 512, 4),      # conv4_2: cout_g 128 -> two 64-channel blocks, 8 chunks, even map
    (3, 19, 512, 512, 4),      # conv5_x: odd map (ragged last tile row / column), tile list crosses images
    (2, 75, 128, 256, 4),      # conv3_1: cin_g 32 -> 2 chunks, odd map
    (2, 37, 128, 128, 4),      # conv2_2 shape class: cout_g 32 -> persistent 32-channel variant, several items per workgroup
    (5, 9, 64, 32, 1),         # dense, one group, one chunk; fewer tiles than one wave in places
    (1, 150, 128, 128, 4),     # many items per persistent workgroup
    (2, 38, 256, 108, 1),      # DCN offset / mask conv: 108 output channels padded to 128 inside U
    (2, 13, 64, 24, 1),        # padded to one 32-channel block
    (2, 83, 64, 64, 4),        # conv1_2 class: patch-staged Winograd (conv_thin_wino.hip), ragged 8 x 16 tiles
    (1, 160, 64, 64, 4),       # ... several tiles per persistent workgroup
]


@pytest.mark.parametrize('case', WINO_CASES)
def test_conv_winograd(dev, ops, case):
    """Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip) against CPU conv2d: plain, with fused batch statistics, with the
    producer's BatchNorm + ReLU applied on the fly, and as a data gradient accumulating into an existing gradient."""
    B, H, Cin, Cout, g = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = torch.from_numpy(rng.normal(0.1, 1.0, size=(B, Cin, H, H)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, Cin // g, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    assert ops.winograd_eligible(3, 1, 1, 1, Cin // g, Cout // g, g)
    ref = torch.nn.functional.conv2d(x, w, b, 1, 1, 1, g)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    xd = nhwc(x).to(dev)
    y = ops.conv2d_nhwc(xd, w.to(dev), b.to(dev), 1, 1, 1, g, stats=stats, winograd=True)
    assert rel(nchw(y), ref) < 2e-5
    n = ref.numel() / Cout
    assert rel(stats[:Cout] / n, ref.double().mean(dim=(0, 2, 3))) < 1e-5
    assert rel(stats[Cout:] / n, (ref.double() ** 2).mean(dim=(0, 2, 3))) < 1e-5
    # the direct implicit GEMM on the same descriptor agrees to fp32 rounding
    y2 = ops.conv2d_nhwc(xd, w.to(dev), b.to(dev), 1, 1, 1, g)
    assert rel(y, y2) < 2e-5
    # fused producer BatchNorm + ReLU, zero padding AFTER the transform
    gm = torch.from_numpy(rng.uniform(-1.5, 1.5, size=Cin).astype(np.float32))
    bt = torch.from_numpy(rng.normal(size=Cin).astype(np.float32))
    refx = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.batch_norm(x, None, None, gm, bt, True, 0.0, 1e-5)), w, b, 1, 1,
                                      1, g)
    st_in = torch.stack([x.double().sum(dim=(0, 2, 3)), (x.double() ** 2).sum(dim=(0, 2, 3))]).reshape(-1).to(dev)
    sc, sh, pdv = (torch.empty(Cin, device=dev) for _ in range(3))
    ops.bn_finalize(st_in, B * H * H, gm.to(dev), bt.to(dev), torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), True, sc,
                    sh, pdv)
    yx = ops.conv2d_nhwc(xd, w.to(dev), b.to(dev), 1, 1, 1, g, winograd=True, in_scale=sc, in_shift=sh, in_pad=pdv)
    assert rel(nchw(yx), refx) < 2e-5
    # data gradient through the same kernel: dX = existing + conv(dY, flipped weights)
    xg = x.clone().requires_grad_()
    yg = torch.nn.functional.conv2d(xg, w, None, 1, 1, 1, g)
    dy = torch.from_numpy(rng.normal(size=tuple(yg.shape)).astype(np.float32))
    yg.backward(dy)
    if ops.winograd_eligible(3, 1, 1, 1, Cout // g, Cin // g, g):
        wd = ops.pack_weight_dgrad(w.to(dev), g)
        ud = ops.winograd_weight(wd, g, Cout // g)
        existing = torch.from_numpy(rng.normal(size=(B, H, H, Cin)).astype(np.float32)).to(dev)
        dx = torch.empty(B, H, H, Cin, device=dev)
        dd, _, _ = ops.make_conv_desc(nhwc(dy).to(dev), wd, dx, B=B, H=H, W=H, in_stride=Cout, cin_g=Cout // g, Cout=Cin, groups=g,
                                      k=3, pad=1, resid=existing, wgt_wino=ud)
        ops.run_conv(dd)
        assert rel(nchw(dx - existing), xg.grad) < 2e-5


BWD_CASES = [
    # B, H, Cin, Cout, k, s, p, d, groups
    (2, 30, 64, 64, 3, 1, 1, 1, 4),      # cout_g 16 (scalar dY path)
    (2, 21, 64, 128, 3, 1, 1, 1, 4),     # cout_g 32
    (2, 19, 128, 256, 3, 1, 1, 1, 4),    # cout_g 64 (b128 path, 64 x 256 tile)
    (2, 19, 512, 512, 3, 1, 1, 1, 4),    # cout_g 128
    (2, 19, 512, 1024, 3, 1, 6, 6, 4),   # dilation 6
    (2, 19, 1024, 1024, 1, 1, 0, 1, 4),  # grouped 1x1
    (2, 19, 256, 512, 3, 2, 1, 1, 4),    # stride 2 (wgrad only)
    (2, 10, 512, 512, 1, 1, 0, 1, 1),    # dense 1x1
    (2, 10, 512, 36, 3, 1, 1, 1, 1),     # head
    (3, 33, 16, 64, 3, 1, 1, 1, 4),      # conv1_1: 4 (3 real) input channels per group
    (2, 83, 16, 64, 3, 1, 1, 1, 4),      # thin patch-staged wgrad <4>, ragged tiles
    (2, 80, 64, 64, 3, 1, 1, 1, 4),      # thin wgrad <16>
    (8, 38, 512, 512, 1, 1, 0, 1, 1),    # large dense 1x1: slot-scheduled TN wgrad (csrc/wgrad_slot.hip) + NT dgrad (gemm_slot.hip)
    (5, 37, 480, 120, 1, 1, 0, 1, 1),    # ... ragged: 6845 pixels (reduction tail), 120 of 128 rows, 480 of 512 columns
    (4, 38, 512, 512, 3, 1, 1, 1, 4),    # conv4_x: slot-scheduled TN wgrad with taps and groups (one tap per 128 columns)
    (12, 19, 512, 1024, 3, 1, 6, 6, 4),  # conv6: dilation 6, cout_g 256 (two row tiles per group)
    (16, 38, 512, 512, 3, 2, 1, 1, 4),   # stride 2 (wgrad only)
    (4, 38, 512, 108, 3, 1, 1, 1, 1),    # DCN offset conv: 108 of 128 rows, K = 4608
    (12, 19, 1024, 1024, 1, 1, 0, 1, 4), # conv7: grouped 1x1
]


@pytest.mark.parametrize('case', BWD_CASES)
def test_conv_backward(dev, ops, case):
    """wgrad kernel and dgrad-as-forward-conv against CPU autograd."""
    B, H, Cin, Cout, k, s, p, d, g = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = torch.from_numpy(rng.normal(size=(B, Cin, H, H)).astype(np.float32)).requires_grad_()
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, Cin // g, k, k)).astype(np.float32)).requires_grad_()
    y = torch.nn.functional.conv2d(x, w, None, s, p, d, g)
    dy = torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32))
    y.backward(dy)
    xd, dyd = nhwc(x.detach()).to(dev), nhwc(dy).to(dev)
    desc, _, _ = ops.make_conv_desc(xd, None, None, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // g, Cout=Cout, groups=g, k=k,
                                    stride=s, pad=p, dil=d)
    dw = ops.conv_wgrad(desc, dyd, Cout, Cin // g, k)
    assert rel(dw, w.grad) < TOL
    if s == 1:
        wd = ops.pack_weight_dgrad(w.detach().to(dev), g)
        Ho = y.shape[2]
        dx = torch.empty(B, H, H, Cin, device=dev)
        pd = d * (k - 1) - p
        dd, _, _ = ops.make_conv_desc(dyd, wd, dx, B=B, H=Ho, W=Ho, in_stride=Cout, cin_g=Cout // g, Cout=Cin, groups=g, k=k,
                                      pad=pd, dil=d)
        ops.run_conv(dd)
        assert rel(nchw(dx), x.grad) < TOL


@pytest.mark.parametrize('H,pool', [(30, None), (30, (2, 2, 0, False)), (75, (2, 2, 0, True)), (19, (3, 1, 1, False))])
def test_bn_relu_pool_backward(dev, ops, H, pool):
    rng = np.random.default_rng(H + 1)
    B, Cc = 3, 64
    x = torch.from_numpy(rng.normal(0.3, 1.5, size=(B, Cc, H, H)).astype(np.float32)).requires_grad_()
    gm = torch.from_numpy(rng.uniform(-1.5, 1.5, size=Cc).astype(np.float32)).requires_grad_()
    bt = torch.from_numpy(rng.normal(size=Cc).astype(np.float32)).requires_grad_()
    y = torch.relu(torch.nn.functional.batch_norm(x, None, None, gm, bt, True, 0.0, 1e-5))
    if pool:
        y = torch.nn.functional.max_pool2d(y, pool[0], pool[1], pool[2], ceil_mode=pool[3])
    dy = torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32))
    y.backward(dy)
    xd = nhwc(x.detach()).to(dev)
    stats = torch.stack([x.detach().double().sum(dim=(0, 2, 3)), (x.detach().double() ** 2).sum(dim=(0, 2, 3))]).reshape(-1).to(dev)
    sc, sh, pd = (torch.empty(Cc, device=dev) for _ in range(3))
    ops.bn_finalize(stats, B * H * H, gm.detach().to(dev), bt.detach().to(dev), torch.zeros(Cc, device=dev),
                    torch.ones(Cc, device=dev), True, sc, sh, pd)
    draw, dg, db, cs = ops.bn_backward(nhwc(dy).to(dev), xd, stats, B * H * H, gm.detach().to(dev), sc, sh,
                                       pool[:3] if pool else None, True, want_colsum=True)
    assert rel(nchw(draw), x.grad) < 2e-4
    assert rel(dg, gm.grad) < 1e-4 and rel(db, bt.grad) < 1e-4
    assert float(cs.abs().max()) < 1e-2 * float(x.grad.abs().sum(dim=(0, 2, 3)).max())      # sum of d(raw) vanishes


@pytest.mark.parametrize('bf16', [False, True])
def test_batch_sum_replicas(dev, ops, bf16):
    """gssd_conv_desc::stats_rep (round 4): the BatchNorm batch sums of a conv spread over R replicas (workgroup id mod R) add up to the
    single-array sums, and the consumers -- gssd_bn_finalize_*, gssd_bn_relu_pool_*, the BatchNorm backward -- fold them: same scale /
    shift / running statistics / outputs / gradients as with one array.  Shapes: a thin trunk layer (persistent kernel, where the
    serialised atomics cost 60 us per launch), a Winograd / flat-window layer and a generic one."""
    from gssd import _lib
    g = torch.Generator().manual_seed(11)
    for (B, H, Cin, Cout, groups) in ((4, 80, 64, 128, 4), (4, 40, 256, 256, 4), (3, 20, 64, 96, 1)):     # (even maps: every pixel is in a pool window)
        x = torch.randn(B, H, H, Cin, generator=g)
        w = torch.randn(Cout, Cin // groups, 3, 3, generator=g) * 0.1
        bias = torch.randn(Cout, generator=g)
        gm, bt = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
        res = {}
        for R in (0, 8):
            stats = torch.zeros(max(R, 1) * 2 * Cout, dtype=torch.float64, device=dev)
            if bf16:
                xd, wp = x.to(dev).to(torch.bfloat16), ops.pack_weight_bf16(w.to(dev))
                out = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
                fn = _lib.lib.gssd_conv2d_nhwc_bf16
            else:
                xd, wp = x.to(dev), ops.pack_weight(w.to(dev))
                out = torch.empty(B, H, H, Cout, device=dev)
                fn = _lib.lib.gssd_conv2d_nhwc_f32
            U = ops.winograd_weight(wp, groups, Cin // groups) if (not bf16 and ops.winograd_eligible(3, 1, 1, 1, Cin // groups, Cout // groups, groups)) else None
            d, _, _ = ops.make_conv_desc(xd, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // groups, Cout=Cout, groups=groups, k=3, pad=1,
                                         bias=bias.to(dev), stats=stats, stats_rep=R, wgt_wino=U)
            _lib.check(fn(ctypes.byref(d), torch.cuda.current_stream().cuda_stream))
            sc, sh = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
            rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
            n = B * H * H
            if bf16:
                pdv = torch.empty(Cout, device=dev, dtype=torch.bfloat16)
                _lib.check(_lib.lib.gssd_bn_finalize_bf16(stats.data_ptr(), float(n), gm.to(dev).data_ptr(), bt.to(dev).data_ptr(), rm.data_ptr(),
                                                          rv.data_ptr(), 0.1, 1e-5, 1, Cout, sc.data_ptr(), sh.data_ptr(), pdv.data_ptr(), R,
                                                          torch.cuda.current_stream().cuda_stream))
                act = None
            else:
                pdv = torch.empty(Cout, device=dev)
                ops.bn_finalize(stats, n, gm.to(dev), bt.to(dev), rm, rv, True, sc, sh, pdv, stats_rep=R)
                act = torch.empty(B, H // 2, H // 2, Cout, device=dev)
                rm2, rv2 = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
                ops.bn_relu_pool(out, act, stats, n, gm.to(dev), bt.to(dev), rm2, rv2, True, True, (2, 2, 0), stats_rep=R)
                dy = torch.randn(B, H // 2, H // 2, Cout, generator=torch.Generator().manual_seed(5)).to(dev)
                draw, dg, db, _ = ops.bn_backward(dy, out, stats, n, gm.to(dev), sc, sh, pool=(2, 2, 0), stats_rep=R)
                assert torch.equal(rm, rm2) and torch.equal(rv, rv2)
                act = (act, draw, dg, db)
            folded = stats.view(max(R, 1), 2 * Cout).sum(0)
            if R:
                assert int((stats.view(R, 2 * Cout)[:, :Cout].abs().sum(1) > 0).sum()) > 1, 'sums landed in one replica only'
            res[R] = (out.float(), folded, sc, sh, rm, rv, act)
        a, b = res[0], res[8]
        assert torch.equal(a[0], b[0])
        assert rel(b[1], a[1]) < 1e-12                    # same fp32 partial sums, another order of the fp64 additions
        for i in (2, 3, 4, 5):
            assert rel(b[i], a[i]) < 1e-6
        if a[6] is not None:
            for u, v in zip(a[6], b[6]):
                assert rel(v, u) < 1e-5


def test_l2norm_gather_upsample_backward(dev, ops):
    from gssd._lib import lib, check
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.normal(size=(2, 64, 9, 9)).astype(np.float32)).requires_grad_()
    w = torch.from_numpy(rng.uniform(15, 25, size=64).astype(np.float32)).requires_grad_()
    y = O.l2norm(x, w)
    dy = torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32))
    y.backward(dy)
    dx, dw = ops.l2norm_backward(nhwc(x.detach()).to(dev), w.detach().to(dev), nhwc(dy).to(dev))
    assert rel(nchw(dx), x.grad) < 1e-5 and rel(dw, w.grad) < 1e-5
    # heads gather
    B, P, A, HW, off, nc = 2, 300, 6, 25, 100, 2
    dloc = torch.from_numpy(rng.normal(size=(B, P, 4)).astype(np.float32))
    dconf = torch.from_numpy(rng.normal(size=(B, P, nc)).astype(np.float32))
    out = torch.empty(B, HW, A * (4 + nc), device=dev)
    dl, dc = dloc.to(dev), dconf.to(dev)          # keep the device tensors alive across the raw-pointer call
    check(lib.gssd_heads_gather_f32(dl.data_ptr(), dc.data_ptr(), out.data_ptr(), B, HW, A, nc, P, off, None))
    ref = torch.cat([dloc[:, off:off + HW * A].reshape(B, HW, A * 4), dconf[:, off:off + HW * A].reshape(B, HW, A * nc)], 2)
    assert torch.equal(out.cpu(), ref)
    # stride-2 conv dgrad = zero insertion + stride-1 conv with flipped weights
    for H in (19, 10):
        xs = torch.from_numpy(rng.normal(size=(2, 64, H, H)).astype(np.float32)).requires_grad_()
        ws = torch.from_numpy(rng.normal(0, 0.1, size=(128, 16, 3, 3)).astype(np.float32))
        ys = torch.nn.functional.conv2d(xs, ws, None, 2, 1, 1, 4)
        dys = torch.from_numpy(rng.normal(size=tuple(ys.shape)).astype(np.float32))
        ys.backward(dys)
        Ho = ys.shape[2]
        u = torch.empty(2, H, H, 128, device=dev)
        dyd = nhwc(dys).to(dev)
        check(lib.gssd_upsample_insert_f32(dyd.data_ptr(), u.data_ptr(), 2, Ho, Ho, H, H, 128, 2, None))
        wd = ops.pack_weight_dgrad(ws.to(dev), 4)
        dxs = torch.empty(2, H, H, 64, device=dev)
        dd, _, _ = ops.make_conv_desc(u, wd, dxs, B=2, H=H, W=H, in_stride=128, cin_g=32, Cout=64, groups=4, k=3, pad=1)
        ops.run_conv(dd)
        assert rel(nchw(dxs), xs.grad) < TOL


def test_conv_heads_layout(dev, ops):
    """loc|conf heads write straight into the concatenated [B,P,4] / [B,P,C] buffers in SSD prior order
    (models/ssd_multiphase_custom_group.py:375-380)."""
    from gssd import _lib
    import ctypes
    rng = np.random.default_rng(3)
    B, H, Cin, A, nc = 2, 5, 64, 6, 2
    x = torch.from_numpy(rng.normal(size=(B, Cin, H, H)).astype(np.float32))
    wl = torch.from_numpy(rng.normal(0, 0.1, size=(A * 4, Cin, 3, 3)).astype(np.float32))
    wc = torch.from_numpy(rng.normal(0, 0.1, size=(A * nc, Cin, 3, 3)).astype(np.float32))
    bl = torch.from_numpy(rng.normal(size=(A * 4,)).astype(np.float32))
    bc = torch.from_numpy(rng.normal(size=(A * nc,)).astype(np.float32))
    P, off = 400, 100                      # pretend this source starts at prior 100 of 400
    wp = torch.empty(A * 4 + A * nc, 9 * Cin, device=dev)
    ops.pack_weight(wl.to(dev), wp, 0)
    ops.pack_weight(wc.to(dev), wp, A * 4)
    bias = torch.cat([bl, bc]).to(dev)
    loc = torch.full((B, P, 4), -7.0, device=dev)
    conf = torch.full((B, P, nc), -7.0, device=dev)
    d, _, _ = ops.make_conv_desc(nhwc(x).to(dev), wp, loc, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin, Cout=A * (4 + nc),
                                 k=3, pad=1, bias=bias, out_mode=_lib.OUT_HEADS, out_b=conf, split_n=A * 4,
                                 out_batch_stride=P * 4, outb_batch_stride=P * nc, out_off=off * 4, outb_off=off * nc)
    ops.run_conv(d)
    rl = torch.nn.functional.conv2d(x, wl, bl, padding=1).permute(0, 2, 3, 1).reshape(B, -1, 4)
    rc = torch.nn.functional.conv2d(x, wc, bc, padding=1).permute(0, 2, 3, 1).reshape(B, -1, nc)
    n = H * H * A
    assert rel(loc[:, off:off + n], rl) < TOL and rel(conf[:, off:off + n], rc) < TOL
    assert (loc[:, :off] == -7).all() and (loc[:, off + n:] == -7).all() and (conf[:, off + n:] == -7).all()


@pytest.mark.parametrize('H,pool', [(30, None), (30, (2, 2, 0, False)), (75, (2, 2, 0, True)), (19, (3, 1, 1, False))])
@pytest.mark.parametrize('training', [True, False])
def test_bn_relu_pool(dev, ops, H, pool, training):
    rng = np.random.default_rng(H)
    B, Cc = 3, 64
    x = torch.from_numpy(rng.normal(0.3, 1.5, size=(B, Cc, H, H)).astype(np.float32))
    gm = torch.from_numpy(rng.uniform(-1.5, 1.5, size=Cc).astype(np.float32))     # negative gammas: max after affine
    bt = torch.from_numpy(rng.normal(size=Cc).astype(np.float32))
    rm = torch.from_numpy(rng.normal(size=Cc).astype(np.float32))
    rv = torch.from_numpy(rng.uniform(0.5, 2, size=Cc).astype(np.float32))
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = torch.relu(torch.nn.functional.batch_norm(x, rm_ref, rv_ref, gm, bt, training, 0.1, 1e-5))
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, pool[0], pool[1], pool[2], ceil_mode=pool[3])
    xd = nhwc(x).to(dev)
    stats = torch.stack([x.double().sum(dim=(0, 2, 3)), (x.double() ** 2).sum(dim=(0, 2, 3))]).reshape(-1).to(dev)
    Ho = ref.shape[2]
    out = torch.empty(B, Ho, Ho, Cc, device=dev)
    rmd, rvd = rm.to(dev), rv.to(dev)
    ops.bn_relu_pool(xd, out, stats, B * H * H, gm.to(dev), bt.to(dev), rmd, rvd, training, True,
                     pool[:3] if pool else None)
    assert rel(nchw(out), ref) < TOL
    assert rel(rmd, rm_ref) < 1e-5 and rel(rvd, rv_ref) < 1e-5


def test_l2norm_softmax_slicecat(dev, ops, golden):
    g = golden('ops')
    y = ops.l2norm(nhwc(torch.from_numpy(g['l2_x'])).to(dev), torch.from_numpy(g['l2_w']).to(dev))
    assert rel(nchw(y), g['l2_y']) < 1e-5                      # reference fixture
    rng = np.random.default_rng(0)
    for n in (1, 9, 25, 361, 1444):
        npad = (n + 3) // 4 * 4
        x = torch.from_numpy(rng.normal(0, 3, size=(7, npad)).astype(np.float32))
        ref = torch.softmax(x[:, :n], -1)
        xd = x.to(dev)
        ops.softmax_rows_(xd, n)
        assert rel(xd[:, :n], ref) < 1e-5 and (xd[:, n:] == 0).all()
    a = torch.from_numpy(rng.normal(size=(2, 512, 6, 6)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(2, 512, 6, 6)).astype(np.float32))
    ref = O.slice_and_cat(a, b, 4)
    out = ops.slice_and_cat(nhwc(a).to(dev), nhwc(b).to(dev), 4)
    assert torch.equal(nchw(out).cpu(), ref)


@pytest.mark.parametrize('do_iter', [True, False])
def test_spectral_norm(dev, ops, do_iter):
    rng = np.random.default_rng(5)
    items, refs = [], []
    # (the last three: three row slices with idle threads; shapes the 16-byte form does not take)
    for (r, c) in ((64, 512), (256, 512), (512, 256), (128, 1024), (512, 1024), (32, 256), (36, 1028), (30, 52), (33, 50)):
        w = torch.from_numpy(rng.normal(0, 0.05, size=(r, c, 1, 1)).astype(np.float32))
        u = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=r).astype(np.float32)), dim=0)
        v = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=c).astype(np.float32)), dim=0)
        wsn, un, vn = O.spectral_weight(w, u, v, do_iter)
        refs.append((float((w.flatten()[0] / wsn.flatten()[0])), un, vn))
        items.append((w.to(dev), u.to(dev), v.to(dev), torch.zeros(r, device=dev)))
    tab = ops.sn_items_tensor(items, dev)
    ops.spectral_norm(tab, len(items), do_iter)
    for (w, u, v, s), (sigma, un, vn) in zip(items, refs):
        assert rel(1.0 / s, np.full(s.shape[0], sigma)) < (1e-5 if do_iter else 1e-4)   # random u,v: u.Wv cancels
        assert rel(u, un) < 1e-5 and rel(v, vn) < 1e-5


def test_dcn_im2col_and_identities(dev, ops):
    """DCN sampling vs the oracle restatement (parity unpinned upstream) + the zero-offset identity the
    reference's zero-initialised conv_offset_mask guarantees (layers/dcn_v2_custom.py:75-77)."""
    rng = np.random.default_rng(8)
    B, Cc, H, dg, Cout = 2, 64, 9, 4, 32
    x = torch.from_numpy(rng.normal(size=(B, Cc, H, H)).astype(np.float32))
    om = torch.from_numpy(rng.normal(0, 1.5, size=(B, 27 * dg, H, H)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, Cc, 3, 3)).astype(np.float32))
    bias = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    o1, o2, m = torch.chunk(om, 3, dim=1)
    ref = O.dcn_v2_conv(x, torch.cat((o1, o2), 1), torch.sigmoid(m), w, bias, 1, 1, 1, dg)
    xd, omd = nhwc(x).to(dev), nhwc(om).to(dev)
    cols = torch.empty(B * H * H, 9 * Cc, device=dev)
    ops.dcn_im2col(xd, omd, cols, dg)
    y = ops.conv2d_nhwc(cols.view(B, H, H, 9 * Cc), w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cc, 1, 1).contiguous().to(dev),
                        bias.to(dev))
    assert rel(nchw(y), ref) < TOL
    omd.zero_()
    ops.dcn_im2col(xd, omd, cols, dg)
    y0 = ops.conv2d_nhwc(cols.view(B, H, H, 9 * Cc), w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cc, 1, 1).contiguous().to(dev),
                         bias.to(dev))
    ident = 0.5 * torch.nn.functional.conv2d(x, w, None, 1, 1) + bias.view(1, -1, 1, 1)
    assert rel(nchw(y0), ident) < TOL


@pytest.mark.parametrize('B,Cc,H,dg,Cout', [(2, 128, 9, 4, 32), (3, 128, 13, 1, 300), (1, 64, 5, 2, 256), (5, 256, 11, 4, 512)])
def test_dcn_fused_forward(dev, ops, B, Cc, H, dg, Cout):
    """The fused deformable conv kernel (sampling + contraction, no column buffer) vs the oracle restatement of DCNv2 (parity
    unpinned upstream), on ragged tiles (M not a multiple of 128, Cout not a multiple of 256, tiles crossing images) with
    offsets that leave the image, plus the two identities the reference's wrapper guarantees."""
    rng = np.random.default_rng(18)
    x = torch.from_numpy(rng.normal(size=(B, Cc, H, H)).astype(np.float32))
    om = torch.from_numpy(rng.normal(0, 2.5, size=(B, 27 * dg, H, H)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, Cc, 3, 3)).astype(np.float32))
    bias = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    o1, o2, m = torch.chunk(om, 3, dim=1)
    ref = O.dcn_v2_conv(x, torch.cat((o1, o2), 1), torch.sigmoid(m), w, bias, 1, 1, 1, dg)
    xd, omd, wd, bd = nhwc(x).to(dev), nhwc(om).to(dev), w.to(dev), bias.to(dev)
    y = ops.dcn_forward(xd, omd, wd, bd, dg)
    assert rel(nchw(y), ref) < TOL
    # zero offsets and mask logits (the zero-initialised conv_offset_mask, dcn_v2_custom.py:75-77): 0.5 * conv + b
    y0 = ops.dcn_forward(xd, torch.zeros_like(omd), wd, bd, dg)
    ident = 0.5 * torch.nn.functional.conv2d(x, w, None, 1, 1) + bias.view(1, -1, 1, 1)
    assert rel(nchw(y0), ident) < TOL
    # integer offsets (dy, dx) = (1, -2) on every tap with mask logit +30 (sigmoid = 1): a conv over the shifted, zero-padded input
    omi = torch.zeros(B, 27 * dg, H, H)
    omi[:, 0:18 * dg:2] = 1.0
    omi[:, 1:18 * dg:2] = -2.0
    omi[:, 18 * dg:] = 30.0
    yi = ops.dcn_forward(xd, nhwc(omi).to(dev), wd, bd, dg)
    xs = torch.zeros_like(x)
    xs[:, :, :H - 1, 2:] = x[:, :, 1:, :H - 2]                  # xs[y, x] = x[y + 1, x - 2]
    shifted = torch.nn.functional.conv2d(xs, w, None, 1, 1)
    # taps that fall outside the ORIGINAL image after the shift are zero in DCN; compare on the interior where both agree
    assert rel(nchw(yi)[:, :, 1:H - 2, 3:H - 1], (shifted + bias.view(1, -1, 1, 1))[:, :, 1:H - 2, 3:H - 1]) < TOL


@pytest.mark.parametrize('B,Cc,H,dg,Cout,std', [(2, 128, 9, 4, 32, 2.5), (1, 64, 7, 1, 64, 5.0), (3, 128, 13, 2, 96, 1.0)])
def test_dcn_fused_vs_scalar_restatement(dev, ops, B, Cc, H, dg, Cout, std):
    """The fused kernel against the SECOND, structurally independent restatement of DCNv2 (oracle/csrc/dcn_scalar.c: scalar loop nest
    in the published algorithm's order, float64 accumulation) -- offsets of several pixels, samples crossing every border -- and the
    properties the algorithm must have whatever the convention: linearity in the mask, integer-offset translation equivariance, the
    (row, column) offset order of utils/show_offset.py:28-32.  DCN stays "parity unpinned" (dcn_v2 is not vendored)."""
    from oracle import dcn_scalar
    rng = np.random.default_rng(28)
    x = rng.normal(size=(B, Cc, H, H)).astype(np.float32)
    om = rng.normal(0, std, size=(B, 27 * dg, H, H)).astype(np.float32)
    w = rng.normal(0, 0.1, size=(Cout, Cc, 3, 3)).astype(np.float32)
    bias = rng.normal(size=(Cout,)).astype(np.float32)
    sig = lambda v: (1.0 / (1.0 + np.exp(-v.astype(np.float64)))).astype(np.float32)
    ref = dcn_scalar.dcn_v2_conv(x, om[:, :18 * dg], sig(om[:, 18 * dg:]), w, bias, 1, 1, 1, dg)
    xd, wd, bd = nhwc(torch.from_numpy(x)).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(bias).to(dev)
    run = lambda o: nchw(ops.dcn_forward(xd, nhwc(torch.from_numpy(o)).to(dev), wd, bd, dg)).cpu().numpy()
    y = run(om)
    assert rel(y, ref) < TOL
    # linearity in the mask: logits chosen so that sigmoid = 0.2 / 0.6 / 0.8 -> y(0.8) - b = (y(0.2) - b) + (y(0.6) - b)
    logit = lambda p: float(np.log(p / (1 - p)))
    ys = []
    for p_ in (0.2, 0.6, 0.8):
        o = om.copy()
        o[:, 18 * dg:] = logit(p_)
        ys.append(run(o) - bias.reshape(1, -1, 1, 1))
    assert rel(ys[2], ys[0] + ys[1]) < 3e-5
    # offset order: +1 on channel 2k of every group moves tap k's sample one ROW down (show_offset.py:28-32), +1 on 2k+1 one COLUMN right
    xt = torch.from_numpy(x)
    for ch, shift in ((0, (1, 0)), (1, (0, 1))):
        o = np.zeros_like(om)
        o[:, 18 * dg:] = 30.0                                             # mask = 1
        for g in range(dg):
            o[:, g * 18 + ch:(g + 1) * 18:2] = 1.0
        xs = torch.zeros_like(xt)
        if shift == (1, 0):
            xs[:, :, :H - 1, :] = xt[:, :, 1:, :]                         # xs[y, x] = x[y + 1, x]
        else:
            xs[:, :, :, :H - 1] = xt[:, :, :, 1:]
        want = (torch.nn.functional.conv2d(xs, torch.from_numpy(w), None, 1, 1) + torch.from_numpy(bias).view(1, -1, 1, 1)).numpy()
        got = run(o)
        # rows / columns whose shifted taps would read the zero padding of the ORIGINAL map differ by construction: compare the interior
        assert rel(got[:, :, 1:H - 2, 1:H - 2], want[:, :, 1:H - 2, 1:H - 2]) < TOL, ch
    # translation equivariance under integer offsets: shifting the input by (2, -1) and adding (-2, +1) to every offset samples the
    # same values wherever both sample sets stay inside the map
    o2 = np.round(om).astype(np.float32)
    o2[:, 18 * dg:] = om[:, 18 * dg:]
    xs = torch.zeros_like(xt)
    xs[:, :, 2:, :H - 1] = xt[:, :, :H - 2, 1:]                           # xs[y, x] = x[y - 2, x + 1]
    o3 = o2.copy()
    for g in range(dg):
        o3[:, g * 18:(g + 1) * 18:2] += 2.0
        o3[:, g * 18 + 1:(g + 1) * 18:2] -= 1.0
    a = dcn_scalar.dcn_v2_conv(x, o2[:, :18 * dg], sig(o2[:, 18 * dg:]), w, bias, 1, 1, 1, dg)
    bq = nchw(ops.dcn_forward(nhwc(xs).to(dev), nhwc(torch.from_numpy(o3)).to(dev), wd, bd, dg)).cpu().numpy()
    # a sample (Y, X) of the unshifted problem becomes (Y + 2, X - 1) in the shifted one and reads the same value unless it is
    # inside the map before and outside after the shift: output pixels without such a sample must agree
    hh, ww = np.meshgrid(np.arange(H), np.arange(H), indexing='ij')
    ok = np.ones((B, H, H), bool)
    for g in range(dg):
        for t in range(9):
            Y = hh[None] - 1 + t // 3 + o2[:, g * 18 + 2 * t]
            X = ww[None] - 1 + t % 3 + o2[:, g * 18 + 2 * t + 1]
            inside = (Y >= 0) & (Y <= H - 1) & (X >= 0) & (X <= H - 1)
            ok &= ~(inside & ((Y + 2 > H - 1) | (X - 1 < 0)))
    assert ok.mean() > 0.02
    sel = np.broadcast_to(ok[:, None], a.shape)
    assert np.abs(a - bq)[sel].max() / np.abs(a).max() < TOL


@pytest.mark.parametrize('Cin,Cout,H,W', [(64, 64, 84, 84), (128, 128, 78, 78), (256, 256, 75, 75), (256, 256, 21, 37)])
@pytest.mark.parametrize('xf', [False, True])
def test_conv_winograd_pooled_epilogue(dev, ops, Cin, Cout, H, W, xf):
    """GSSD_CONV_POOL2 on the fp32 Winograd trunk kernels (conv1_2: conv_thin_wino, conv2_2: conv_wino<32>, conv3_3: conv_wino<64> with
    its ceil-mode 75 -> 38 pool; a non-square map): the launch stores max- / min-pooled raw outputs by the sign of the BatchNorm weight
    with the batch sums of the full map -- exactly the pooled image of what the plain launch stores -- and the deferred BatchNorm +
    ReLU of that map equals BatchNorm + ReLU + max-pool of the full map bit for bit."""
    import ctypes as C
    from gssd import _lib
    F = torch.nn.functional
    rng = np.random.default_rng(Cin + H)
    B, g = 2, 4
    x = torch.from_numpy(rng.normal(0.2, 1.0, size=(B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.08, size=(Cout, Cin // g, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    gamma = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    gamma[5] = 0.0
    act, sc, sh, pdv = x, None, None, None
    if xf:
        scv = torch.from_numpy(rng.uniform(0.2, 1.5, size=Cin).astype(np.float32)) * torch.from_numpy(rng.choice([-1.0, 1.0], size=Cin).astype(np.float32))
        shv = torch.from_numpy(rng.normal(size=Cin).astype(np.float32))
        act = torch.relu(x * scv.view(1, -1, 1, 1) + shv.view(1, -1, 1, 1))
        sc, sh = scv.to(dev), shv.to(dev)
        pdv = torch.where(sc > 0, torch.full_like(sc, -3.0e38), torch.full_like(sc, 3.0e38))
    ref = F.conv2d(act, w, b, 1, 1, 1, g)
    wp = ops.pack_weight(w.to(dev))
    U = ops.winograd_weight(wp, g, Cin // g)
    Hp, Wp = (H + 1) // 2, (W + 1) // 2
    out = torch.full((B, Hp, Wp, Cout), float('nan'), device=dev)
    full = torch.empty(B, H, W, Cout, device=dev)
    stats, stats_full = torch.zeros(2 * Cout, dtype=torch.float64, device=dev), torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    gd_ = gamma.to(dev)
    kw = dict(B=B, H=H, W=W, in_stride=Cin, cin_g=Cin // g, Cout=Cout, groups=g, k=3, stride=1, pad=1, bias=b.to(dev), wgt_wino=U,
              in_scale=sc, in_shift=sh, in_pad=pdv)
    d, _, _ = ops.make_conv_desc(nhwc(x).to(dev), wp, out, stats=stats, flags=_lib.CONV_POOL2, pool_sign=gd_, **kw)
    d0, _, _ = ops.make_conv_desc(nhwc(x).to(dev), wp, full, stats=stats_full, **kw)
    st_ = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib.gssd_conv2d_nhwc_f32(C.byref(d), st_))
    _lib.check(_lib.lib.gssd_conv2d_nhwc_f32(C.byref(d0), st_))

    def pool_by_sign(raw):
        mx, mn = F.max_pool2d(raw, 2, 2, 0, ceil_mode=True), -F.max_pool2d(-raw, 2, 2, 0, ceil_mode=True)
        return torch.where(gamma.view(1, -1, 1, 1) >= 0, mx, mn)
    want = pool_by_sign(nchw(full.cpu()))
    assert torch.equal(nchw(out.cpu()), want)                               # the pooled image of the plain launch's output, bit for bit
    assert rel(stats, stats_full) < 1e-13                                  # the same fp32 additions in the same order (fp64 atomics: last bits)
    assert rel(nchw(out), pool_by_sign(ref)) < TOL                          # (Winograd F(2x2, 3x3) vs direct: 1e-4 of the tensor's scale)
    scale, shift = gamma * 0.7, torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    f = lambda t: torch.relu(torch.addcmul(shift.view(1, -1, 1, 1), t, scale.view(1, -1, 1, 1)))
    assert torch.equal(f(want), F.max_pool2d(f(nchw(full.cpu())), 2, 2, 0, ceil_mode=True))
    # a shape no pooled epilogue exists for is refused loudly, not computed unpooled (without Winograd weights only csrc/conv_thin_x6.hip's
    # shape classes -- conv1_2, conv2_2 -- have one)
    d1, _, _ = ops.make_conv_desc(nhwc(x).to(dev), wp, out, flags=_lib.CONV_POOL2, pool_sign=gd_, **{**kw, 'wgt_wino': None})
    if _lib.lib.gssd_conv_thin_x6_takes(C.byref(d1)) == 0:
        assert _lib.lib.gssd_conv2d_nhwc_f32(C.byref(d1), st_) == -1


def test_sa_backward_building_blocks(dev):
    """gssd_bgemm_f32 (all four transpose forms, ragged sizes, batched), the row softmax backward, the spectral-norm chain rule and
    the small helpers of csrc/sa_backward.hip against torch-CPU."""
    from gssd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(4)
    Bt, M, N, K = 3, 77, 50, 133
    for ta in (0, 1):
        for tb in (0, 1):
            a = torch.from_numpy(rng.normal(size=(Bt, K, M) if ta else (Bt, M, K)).astype(np.float32))
            b = torch.from_numpy(rng.normal(size=(Bt, N, K) if tb else (Bt, K, N)).astype(np.float32))
            ref = torch.matmul(a.transpose(1, 2) if ta else a, b.transpose(1, 2) if tb else b)
            ad, bd = a.to(dev), b.to(dev)
            c = torch.full((Bt, M, N + 3), 7.0, device=dev)          # ldc > N: the slack columns must stay untouched
            check(lib.gssd_bgemm_f32(ad.data_ptr(), bd.data_ptr(), c.data_ptr(), M, N, K, a.shape[2], b.shape[2], N + 3, ta, tb,
                                     a[0].numel(), b[0].numel(), M * (N + 3), Bt, 0.5, 0, st))
            assert rel(c[:, :, :N], 0.5 * ref) < 1e-5 and float((c[:, :, N:] - 7.0).abs().max()) == 0
            check(lib.gssd_bgemm_f32(ad.data_ptr(), bd.data_ptr(), c.data_ptr(), M, N, K, a.shape[2], b.shape[2], N + 3, ta, tb,
                                     a[0].numel(), b[0].numel(), M * (N + 3), Bt, 1.0, 1, st))
            assert rel(c[:, :, :N], 1.5 * ref) < 1e-5
    # softmax backward over rows
    R, n, stride = 37, 101, 104
    logits = torch.from_numpy(rng.normal(0, 3, size=(R, n)).astype(np.float32)).requires_grad_()
    attn = torch.softmax(logits, -1)
    g = torch.from_numpy(rng.normal(size=(R, n)).astype(np.float32))
    attn.backward(g)
    A = torch.zeros(R, stride)
    A[:, :n] = attn.detach()
    dA = torch.full((R, stride), 3.0)
    dA[:, :n] = g
    Ad, dAd = A.to(dev), dA.to(dev)
    check(lib.gssd_softmax_bwd_rows_f32(Ad.data_ptr(), dAd.data_ptr(), R, n, stride, st))
    assert rel(dAd[:, :n], logits.grad) < 1e-5 and float(dAd[:, n:].abs().max()) == 0
    # spectral-norm chain rule: W_eff = W / (u^T W v)
    rows, cols = 24, 40
    W = torch.from_numpy(rng.normal(size=(rows, cols)).astype(np.float32)).requires_grad_()
    u = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=rows).astype(np.float32)), dim=0)
    v = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=cols).astype(np.float32)), dim=0)
    G = torch.from_numpy(rng.normal(size=(rows, cols + 5)).astype(np.float32))
    sigma = torch.dot(u, torch.mv(W, v))
    ((W / sigma) * (0.7 * G[:, :cols])).sum().backward()
    Gd, Wd, ud, vd = G.to(dev), W.detach().to(dev), u.to(dev), v.to(dev)
    isg = (1.0 / sigma.detach()).reshape(1).to(dev)
    sc = torch.tensor([0.7], device=dev)
    out = torch.empty(rows, cols, device=dev)
    scratch = torch.zeros(1, dtype=torch.float64, device=dev)
    check(lib.gssd_sn_weight_grad_f32(Gd.data_ptr(), cols + 5, Wd.data_ptr(), ud.data_ptr(), vd.data_ptr(), isg.data_ptr(), sc.data_ptr(),
                                      scratch.data_ptr(), out.data_ptr(), rows, cols, st))
    assert rel(out, W.grad) < 1e-5
    # scaled transpose, dot, axpby, scaled cast, sigma gradient
    al = torch.from_numpy(rng.uniform(0.5, 2, size=rows).astype(np.float32)).to(dev)
    wt = torch.empty(cols, rows, device=dev)
    check(lib.gssd_scaled_transpose_f32(Wd.data_ptr(), al.data_ptr(), wt.data_ptr(), rows, cols, st))
    assert rel(wt, (Wd * al.view(-1, 1)).t()) < 1e-6
    d64 = torch.zeros(1, dtype=torch.float64, device=dev)
    check(lib.gssd_dot_f32(Wd.data_ptr(), Gd[:, :cols].contiguous().data_ptr(), rows * cols, d64.data_ptr(), st))
    assert rel(d64, (W.detach().double() * G[:, :cols].double()).sum()) < 1e-9
    z = torch.empty_like(Wd)
    Gc = Gd[:, :cols].contiguous()
    check(lib.gssd_axpby_f32(Wd.data_ptr(), Gc.data_ptr(), z.data_ptr(), rows * cols, 2.0, -1.0, st))
    assert rel(z, 2 * Wd - Gc) < 1e-6
    cs = torch.from_numpy(rng.normal(size=rows)).to(dev)
    y = torch.empty(rows, device=dev)
    check(lib.gssd_scale_cast_f64_f32(cs.data_ptr(), sc.data_ptr(), y.data_ptr(), rows, st))
    assert rel(y, 0.7 * cs) < 1e-6
    ds = torch.empty(1, device=dev)
    check(lib.gssd_sa_sigma_grad_f32(d64.data_ptr(), cs.data_ptr(), al.data_ptr(), rows, ds.data_ptr(), st))
    assert rel(ds, d64 + (al.double() * cs).sum()) < 1e-6


def test_dcn_col2im_backward(dev, ops):
    """Sampling backward (d x by atomics, d offset, d mask logit) vs autograd through the oracle's DCN restatement."""
    rng = np.random.default_rng(18)
    B, Cc, H, dg = 2, 128, 11, 2
    x = torch.from_numpy(rng.normal(size=(B, Cc, H, H)).astype(np.float32)).requires_grad_()
    om = torch.from_numpy(rng.normal(0, 2.5, size=(B, 27 * dg, H, H)).astype(np.float32)).requires_grad_()
    # identity "weight": the conv output IS the column matrix (channel c*9 + tap), so d(out) = d(cols)
    w = torch.eye(Cc * 9).view(Cc * 9, Cc, 3, 3)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    cols_ref = O.dcn_v2_conv(x, torch.cat((o1, o2), 1), torch.sigmoid(m), w, torch.zeros(Cc * 9), 1, 1, 1, dg)
    gcols = torch.from_numpy(rng.normal(size=(B, Cc, 9, H, H)).astype(np.float32))       # [b][c][tap][h][w]
    cols_ref.backward(gcols.view(B, Cc * 9, H, H))
    xd, omd = nhwc(x.detach()).to(dev), nhwc(om.detach()).to(dev)
    dcols = gcols.permute(0, 3, 4, 2, 1).reshape(B * H * H, 9 * Cc).contiguous().to(dev)   # [pixel][tap*C + c]
    dx = torch.zeros_like(xd)
    dom = torch.zeros_like(omd)
    ops.dcn_col2im(xd, omd, dcols, dx, dom, dg)
    assert rel(nchw(dx), x.grad) < TOL
    assert rel(nchw(dom), om.grad) < TOL


@pytest.mark.parametrize('N,D,C2', [(1444, 64, 256), (1100, 64, 256), (300, 64, 256)])
def test_self_attn_core_x6(dev, N, D, C2):
    """csrc/flash_attn_x6.hip (the fp32 mode's attention core on the bf16 matrix cores, three-plane operands) against float64
    softmax(theta phi^T) g (layers/self_attn.py:68-80) and against the fp32-MFMA core on the same inputs: fp32-equivalent means it
    must be as close to float64 as the fp32 kernel is (and both within 3e-5 of the output scale at logits of +-200); the rows' log-sum-exp (what the
    training step's backward reads) likewise.  Logits of magnitude ~30 (the detector's are ~50): a bf16-rounded logit would be off by
    0.1; N = 300 leaves a ragged last key block and a ragged query tile."""
    from gssd import _lib
    lib = _lib.lib
    B = 2
    rng = np.random.default_rng(N + D)
    Np = (N + 3) // 4 * 4
    tp = torch.from_numpy(rng.normal(0, 1.9, size=(B, N, 2 * D)).astype(np.float32))
    gT = torch.zeros(B, C2, Np)
    gT[:, :, :N] = torch.from_numpy(rng.normal(0, 1.0, size=(B, C2, N)).astype(np.float32))
    th, ph = tp[:, :, :D].double(), tp[:, :, D:].double()
    logits = th @ ph.transpose(1, 2)
    ref = torch.softmax(logits, -1) @ gT[:, :, :N].double().transpose(1, 2)
    ref_lse = torch.logsumexp(logits, -1)
    tpd, gTd = tp.to(dev), gT.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    out6, lse6 = torch.empty(B, N, C2, device=dev), torch.empty(B, N, device=dev)
    ws = torch.empty(int(lib.gssd_self_attn_core_x6_ws_bytes(B, N, D, C2)) // 4, device=dev)
    assert lib.gssd_self_attn_core_x6_supported(D, C2) == 1 and lib.gssd_self_attn_core_x6_supported(128, 512) == 0
    _lib.check(lib.gssd_self_attn_core_x6_f32(tpd.data_ptr(), gTd.data_ptr(), out6.data_ptr(), B, N, Np, D, C2, ws.data_ptr(), lse6.data_ptr(), st))
    out4, lse4 = torch.empty(B, N, C2, device=dev), torch.empty(B, N, device=dev)
    _lib.check(lib.gssd_self_attn_core_kv_f32(tpd.data_ptr(), tpd[0, 0, D:].data_ptr(), gTd.data_ptr(), out4.data_ptr(), B, N, N, Np, D, C2, 2 * D,
                                              0, lse4.data_ptr(), st))
    torch.cuda.synchronize()
    e6, e4 = rel(out6, ref), rel(out4, ref)
    l6, l4 = float((lse6.cpu().double() - ref_lse).abs().max()), float((lse4.cpu().double() - ref_lse).abs().max())
    print(f'attention core N={N} D={D}: vs float64 three-plane {e6:.2e} fp32-MFMA {e4:.2e}; lse abs err {l6:.2e} / {l4:.2e}; max |logit| {float(logits.abs().max()):.1f}')
    assert e6 < 3e-5 and e6 <= 1.5 * e4 + 1e-6 and l6 <= 1.5 * l4 + 1e-5


def test_self_attn_op(dev, golden):
    """Self_Attn on its own (layers/self_attn.py:46-89) against the reference fixtures: out, sigma*o, the ATTENTION MAP, and
    (train) the spectral-norm u / v after-state; eval mode leaves u / v untouched."""
    from gssd.engine import SelfAttnOp
    from gssd.modules import Self_Attn
    g = golden('ops')
    for mode in ('eval', 'train'):
        sa = Self_Attn(64)
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in sa.state_dict().items()}, seed=21)   # as make_golden.py
        sa.load_state_dict(sd)
        sa = sa.to(dev)
        x = torch.from_numpy(g[f'sa_{mode}_x'])
        op = SelfAttnOp(sa, 2, 6, mode == 'train', dev)
        out, out2, attn = op.run(nhwc(x).to(dev))
        assert rel(nchw(out), g[f'sa_{mode}_out']) < TOL
        assert rel(nchw(out2), g[f'sa_{mode}_ag']) < TOL
        assert attn.shape == (2, 36, 36)
        assert rel(attn, g[f'sa_{mode}_attn']) < TOL
        after = sa.state_dict()
        for k in ('theta', 'phi', 'g', 'attn'):
            for uv in ('weight_u', 'weight_v'):
                key = f'snconv1x1_{k}.{uv}'
                want = g[f'sa_train_after.{key}'] if mode == 'train' else sd[key].numpy()
                assert rel(after[key], want) < TOL, (mode, key)
        # the oracle agrees on the same inputs (map included)
        upd = {}
        o_out, o_ag, o_attn = O.self_attn(x, {f'p.{k}': v for k, v in sd.items()}, 'p', mode == 'train', updates=upd)
        assert rel(nchw(out), o_out) < TOL and rel(attn, o_attn) < TOL


@pytest.mark.parametrize('case', ['bn_relu', 'gate_resid', 'split_t', 'ragged'])
def test_gemm_slot(dev, ops, case):
    """csrc/gemm_slot.hip (large plain 1x1 convs / GEMMs: 128 x 256 slot-scheduled MFMA stream) through gssd_conv2d_nhwc_f32: every
    epilogue it carries, checked against fp32 matmul on the CPU, after asserting the dispatcher really takes the slot kernel."""
    from gssd import _lib
    import ctypes as C
    rng = np.random.default_rng(77)
    t = lambda *s, sc=1.0: torch.from_numpy(rng.normal(0, sc, size=s).astype(np.float32))
    if case == 'bn_relu':                     # alpha + bias + ReLU + BatchNorm statistics, M a tile multiple + tail
        B, H, K, N = 10, 38, 512, 512
        x, w, b, al = t(B, H, H, K), t(N, K, sc=0.05), t(N), t(N).abs() + 0.5
        out = torch.empty(B, H, H, N, device=dev)
        stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
        keep = [x.to(dev), w.to(dev), b.to(dev), al.to(dev)]
        d, _, _ = ops.make_conv_desc(keep[0], keep[1], out, B=B, H=H, W=H, in_stride=K, cin_g=K, Cout=N, bias=keep[2], alpha=keep[3],
                                     relu=True, stats=stats)
        assert _lib.lib.gssd_gemm_slot_takes(C.byref(d)) == 1
        ops.run_conv(d)
        pre = (x.view(-1, K) @ w.t()) * al + b
        assert rel(out.view(-1, N), pre.clamp_min(0)) < TOL
        assert rel(stats[:N], pre.double().sum(0)) < 1e-5 and rel(stats[N:], (pre.double() ** 2).sum(0)) < 1e-5
    elif case == 'gate_resid':                # the attention output conv: sigma gate, second output, residual
        B, H, K, N = 10, 38, 256, 512
        x, w, b, res, g = t(B, H, H, K), t(N, K, sc=0.05), t(N), t(B, H, H, N), torch.tensor([0.37])
        out, out2 = torch.empty(B, H, H, N, device=dev), torch.empty(B, H, H, N, device=dev)
        keep = [x.to(dev), w.to(dev), b.to(dev), res.to(dev), g.to(dev)]
        d, _, _ = ops.make_conv_desc(keep[0], keep[1], out, B=B, H=H, W=H, in_stride=K, cin_g=K, Cout=N, bias=keep[2], gate=keep[4],
                                     resid=keep[3], out2=out2)
        assert _lib.lib.gssd_gemm_slot_takes(C.byref(d)) == 1
        ops.run_conv(d)
        o2 = ((x.view(-1, K) @ w.t()) + b) * 0.37
        assert rel(out2.view(-1, N), o2) < TOL and rel(out.view(-1, N), o2 + res.view(-1, N)) < TOL
    elif case == 'split_t':                   # merged theta | phi | g projection: per image, g written transposed
        B, H, K, C4, C2 = 34, 37, 512, 256, 256          # 1369 tokens: the transposed rows carry 3 pad columns
        Nn = H * H
        Np = (Nn + 3) // 4 * 4
        x, w, b = t(B, Nn, K), t(C4 + C2, K, sc=0.05), t(C4 + C2)
        tp = torch.empty(B, Nn, C4, device=dev)
        gT = torch.full((B, C2, Np), float("nan"), device=dev)
        keep = [x.to(dev), w.to(dev), b.to(dev)]
        d, _, _ = ops.make_conv_desc(keep[0], keep[1], tp, B=B, H=H, W=H, in_stride=K, cin_g=K, Cout=C4 + C2, bias=keep[2],
                                     out_mode=_lib.OUT_SPLIT_T, out_b=gT, split_n=C4, out_stride=C4, out_b_stride=Np, m_per_image=True,
                                     in_batch_stride=Nn * K, out_batch_stride=Nn * C4, outb_batch_stride=C2 * Np)
        assert _lib.lib.gssd_gemm_slot_takes(C.byref(d)) == 1
        ops.run_conv(d)
        ref = x @ w.t() + b
        assert rel(tp, ref[:, :, :C4]) < TOL
        assert rel(gT[:, :, :Nn], ref[:, :, C4:].transpose(1, 2)) < TOL
        assert float(gT[:, :, Nn:].abs().max()) == 0.0
    else:                                     # ragged: M tail, 480 of 512 columns, K = 96, strided input / output channel windows
        B, H, K, N = 11, 37, 96, 480
        xs, os_ = K + 32, N + 64
        x, w = t(B, H, H, xs), t(N, K, sc=0.05)
        out = torch.full((B, H, H, os_), 7.0, device=dev)
        keep = [x.to(dev), w.to(dev)]
        d, _, _ = ops.make_conv_desc(keep[0], keep[1], out, B=B, H=H, W=H, in_stride=xs, in_ch_off=16, cin_g=K, Cout=N, out_stride=os_,
                                     out_ch_off=32)
        assert _lib.lib.gssd_gemm_slot_takes(C.byref(d)) == 1
        ops.run_conv(d)
        ref = x[..., 16:16 + K].reshape(-1, K) @ w.t()
        assert rel(out.view(-1, os_)[:, 32:32 + N], ref) < TOL
        assert float((out.view(-1, os_)[:, :32] - 7.0).abs().max()) == 0.0 and float((out.view(-1, os_)[:, 32 + N:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize('B,Cc,Cout,dg', [(32, 1024, 512, 4), (24, 1024, 512, 4), (12, 1024, 512, 4), (16, 256, 640, 1), (20, 128, 300, 2)])
def test_dcn_fused_streamk(dev, ops, B, Cc, Cout, dg):
    """gssd_dcn_forward_f32 at the GSSD++ shape (38 x 38, 1024 -> 512, 4 deformable groups) in its stream-K form against the
    one-tile-per-workgroup form, at batch 32 (722 tiles for 256 CUs: two whole tiles per workgroup, the remaining 26 - 27 tiles of an XCD
    cut into 0.82-tile spans), 24 (542 tiles: remaining tiles cut into ~8 pieces each) and 12 (272 tiles: ~16 pieces each), and on two
    other shapes (three N tiles -- the tile order that does not divide the 8 XCDs --, a partial last N tile, one / two deformable groups,
    M not a multiple of 128): the same products, cut tiles add their pieces' partial sums in a fixed chain (fp32 rounding only); stream-K runs agree bit for bit and the
    per-tile flags are back to zero afterwards (a later launch would hang or read stale partials otherwise)."""
    from gssd._lib import lib, check
    H = 38
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, H, Cc, generator=g).to(dev)
    om = (torch.randn(B, H, H, 27 * dg, generator=g) * 0.8).to(dev)
    w = (torch.randn(Cout, Cc, 3, 3, generator=g) * 0.01).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    wp = ops.dcn_pack_weight(w, dg)
    st = torch.cuda.current_stream().cuda_stream

    def run(mode, xx=x, oo=om):
        prev = lib.gssd_dcn_streamk(mode)
        n = xx.shape[0]
        out = torch.full((n, H, H, Cout), float('nan'), device=dev)
        try:
            check(lib.gssd_dcn_forward_f32(xx.data_ptr(), oo.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr(), n, H, H, Cc, dg,
                                           27 * dg, Cout, st))
            torch.cuda.synchronize()
        finally:
            lib.gssd_dcn_streamk(prev)
        return out
    plain, sk1, sk2, sk3 = run(0), run(1), run(1), run(1)
    assert torch.isfinite(sk1).all()
    assert rel(sk1, plain) < 3e-6
    assert torch.equal(sk1, sk2) and torch.equal(sk1, sk3)
    frac = float((sk1 != plain).float().mean())
    print(f'stream-K B={B}: {100 * frac:.1f} % of the outputs differ from the unsplit form (cut tiles), max rel {rel(sk1, plain):.1e}')
    assert 0.0 < frac < 0.7
    # a shape whose tile count is below the CU count keeps the one-tile form whatever the setting
    a, b = run(0, x[:2].contiguous(), om[:2].contiguous()), run(1, x[:2].contiguous(), om[:2].contiguous())
    assert torch.equal(a, b)
    # usable, no wait timed out (bit 2 = "id & 7 is not the XCD here" is information: the hand-over is placement independent)
    assert lib.gssd_dcn_streamk_status(None) & ~2 == 0
    # the flag / slab regions belong to OUTPUT buffers: releasing one frees exactly its region, NULL frees the rest, a launch afterwards
    # makes its own again (regions used to accumulate for the life of the process)
    # (one region per output ADDRESS: the caching allocator may hand sk1 / sk2 / sk3 the same or different blocks)
    assert lib.gssd_dcn_streamk_release(sk3.data_ptr()) == 1 and lib.gssd_dcn_streamk_release(sk3.data_ptr()) == 0
    assert lib.gssd_dcn_streamk_release(None) >= 0 and lib.gssd_dcn_streamk_release(None) == 0
    assert torch.equal(run(1), sk1)


def test_dcn_streamk_is_placement_independent(dev, ops):
    """VERDICT r3 item 5 / ADVICE r3.  Round 3's stream-K hand-over of csrc/dcn_fused.hip passed partial sums through plain stores and was
    valid only while workgroup id & 7 decides the XCD.  Round 4 first added a run-time check of HW_REG_XCC_ID in every stream-K workgroup --
    and it FIRED on this pool as soon as several launches were in flight on different streams (this test's 12-launch case, and the
    captured forward of test_full_size_properties[gssdpp]).  The hand-over is now placement independent (sc1 slabs + drained flag,
    cdna_hip_programming.md Guideline 16); this test prints the idle-device placement probe (information), runs 12 launches on 4 streams
    at once -- each output has its own flags and slabs -- and demands bit-equality with a serial launch, and checks that
    gssd_dcn_streamk_reset leaves a usable state and that no wait timed out."""
    import ctypes
    from gssd._lib import lib, check
    xm = ctypes.c_uint(0)
    st = lib.gssd_dcn_streamk_status(ctypes.cast(ctypes.pointer(xm), ctypes.c_void_p))
    ids = [(xm.value >> (4 * r)) & 15 for r in range(8)]
    print(f'stream-K: status {st}; idle-device placement probe: XCC id of workgroup id & 7 = 0..7: {ids}')
    assert st & ~2 == 0, f'status {st}: stream-K is unusable on this device'
    B, H, Cc, Cout, dg = 12, 38, 1024, 512, 4
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, H, Cc, generator=g).to(dev)
    om = (torch.randn(B, H, H, 27 * dg, generator=g) * 0.8).to(dev)
    w = (torch.randn(Cout, Cc, 3, 3, generator=g) * 0.01).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    wp = ops.dcn_pack_weight(w, dg)
    prev = lib.gssd_dcn_streamk(1)
    try:
        def launch(out, stream):
            check(lib.gssd_dcn_forward_f32(x.data_ptr(), om.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, H, Cc, dg,
                                           27 * dg, Cout, stream.cuda_stream))
        ref = torch.empty(B, H, H, Cout, device=dev)
        launch(ref, torch.cuda.current_stream())
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
        outs = [torch.full((B, H, H, Cout), float('nan'), device=dev) for _ in range(12)]
        for i, o in enumerate(outs):                       # 12 launches in flight on 4 streams, 12 different outputs
            launch(o, streams[i % 4])
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, ref)
        check(lib.gssd_dcn_streamk_reset(torch.cuda.current_stream().cuda_stream))
        again = torch.empty_like(ref)
        launch(again, torch.cuda.current_stream())
        torch.cuda.synchronize()
        assert torch.equal(again, ref)
    finally:
        lib.gssd_dcn_streamk(prev)
    assert lib.gssd_dcn_streamk_status(None) & ~2 == 0


@pytest.mark.parametrize('f16ok', [False, True], ids=['bf16_planes', 'fp16_planes'])
def test_dcn_x6_matches_fused(dev, ops, f16ok):
    """csrc/dcn_x6.hip: the fp32 deformable conv with every operand as the exact sum of three bf16 planes and six bf16 MFMAs per product --
    or, flagged GSSD_CONV_F16_OK by the caller (a train-mode forward), of fp16 planes and three MFMAs -- fp32-equivalent: it must agree with the
    fp32-MFMA kernel to fp32 summation-order noise and with the float64 evaluation as well as that kernel does.  Shapes with a ragged last
    pixel tile and a masked channel tile."""
    rng = np.random.default_rng(77)
    for (B, Cc, H, dg, Cout) in ((2, 128, 13, 4, 136), (1, 256, 19, 1, 512), (3, 128, 11, 2, 264), (1, 192, 9, 2, 40),
                                  (1, 32, 7, 1, 8)):     # 1, 8, 2, 3, 1 channel blocks per tap; the last: nine chunks in all
        x = torch.from_numpy(rng.normal(size=(B, H, H, Cc)).astype(np.float32)).to(dev)
        om = torch.from_numpy(rng.normal(0, 1.5, size=(B, H, H, 27 * dg)).astype(np.float32)).to(dev)
        w = torch.from_numpy(rng.normal(0, 0.05, size=(Cout, Cc, 3, 3)).astype(np.float32)).to(dev)
        b = torch.from_numpy(rng.normal(size=Cout).astype(np.float32)).to(dev)
        ref = ops.dcn_forward(x, om, w, b, dg)
        got = ops.dcn_forward_x6(x, om, w, b, dg, f16ok=f16ok)
        o1, o2, m = torch.chunk(nchw(om.cpu()).double(), 3, dim=1)
        r64 = O.dcn_v2_conv(nchw(x.cpu()).double(), torch.cat((o1, o2), 1), torch.sigmoid(m), w.cpu().double(), b.cpu().double(), 1, 1, 1, dg)
        e_x6, e_f = rel(nchw(got), r64), rel(nchw(ref), r64)
        print(f'dcn x6 vs float64 {e_x6:.2e}; fp32-MFMA kernel vs float64 {e_f:.2e}; x6 vs fp32-MFMA {rel(got, ref):.2e}')
        assert e_x6 < 2e-6 + 2 * e_f and rel(got, ref) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 19, 64, 64, 1, 3, 1, 1, 1), (3, 21, 128, 256, 4, 3, 1, 1, 1), (2, 19, 128, 512, 4, 3, 6, 6, 1),
                                   (2, 17, 256, 216, 1, 3, 1, 1, 1), (2, 20, 256, 256, 4, 1, 0, 1, 1), (2, 23, 64, 96, 1, 3, 1, 1, 2),
                                   (1, 9, 32, 40, 1, 5, 2, 1, 1), (2, 19, 32, 64, 1, 1, 0, 1, 1), (1, 11, 32, 128, 1, 1, 0, 1, 1),
                                   (1, 13, 64, 136, 1, 1, 0, 1, 1)])          # (the last three: K loops of one and two chunks)
def test_conv_x6_matches_float64(shape):
    """csrc/conv_x6.hip: fp32 conv with three-plane bf16 operands (six MFMAs per product) is at least as close to a float64 convolution as
    the fp32-MFMA kernel, for every tile width, with the fused input transform, bias, ReLU and the batch sums."""
    import torch.nn.functional as F
    from gssd import ops
    B, H, Cin, Cout, g, k, pad, dil, stride = shape
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, H + 3, Cin, generator=gen).to(dev)
    w = (torch.randn(Cout, Cin // g, k, k, generator=gen) * 0.1).to(dev)
    b = torch.randn(Cout, generator=gen).to(dev)
    sc, sh = (torch.rand(Cin, generator=gen) + 0.5).to(dev), (torch.randn(Cin, generator=gen) * 0.3).to(dev)
    pdv = -sh / sc - 1.0
    # bf16 planes, six MFMAs (eval-mode forwards, every data gradient) and, flagged GSSD_CONV_F16_OK, fp16 planes, three MFMAs (a train-mode
    # forward's launches) -- each plain and with the fused input transform
    from gssd import _lib
    for xf, f16ok in ((False, False), (False, True), (True, False), (True, True)):
        kw = dict(in_scale=sc, in_shift=sh, in_pad=pdv) if xf else {}
        if f16ok:
            kw['flags'] = _lib.CONV_F16_OK
        xin = F.relu(x.double() * sc.double() + sh.double()) if xf else x.double()
        ref = F.conv2d(xin.permute(0, 3, 1, 2), w.double(), b.double(), stride, pad, dil, g).permute(0, 2, 3, 1)
        st6 = torch.zeros(2 * Cout, device=dev, dtype=torch.float64)
        y6 = ops.conv2d_nhwc(x, w, b, stride, pad, dil, g, stats=st6, x6=True, **kw)
        y32 = ops.conv2d_nhwc(x, w, b, stride, pad, dil, g, **kw)
        e6 = float((y6.double() - ref).abs().max() / ref.abs().max())
        e32 = float((y32.double() - ref).abs().max() / ref.abs().max())
        print(f'conv_x6 {shape} xf {xf} f16ok {f16ok}: x6 {e6:.2e} fp32 {e32:.2e}')
        assert e6 < 2e-6 and e6 <= 1.5 * e32 + 1e-7, (xf, f16ok, e6, e32)
        n_px = ref.shape[0] * ref.shape[1] * ref.shape[2]           # fp32 partial sums per tile: errors relative to sum |v|, not to the sum
        assert float((st6[:Cout] - ref.sum((0, 1, 2))).abs().max()) < 2e-7 * n_px * float(ref.abs().max())
        assert float((st6[Cout:] - (ref * ref).sum((0, 1, 2))).abs().max()) < 2e-7 * n_px * float(ref.abs().max()) ** 2
        yr = ops.conv2d_nhwc(x, w, b, stride, pad, dil, g, relu=True, x6=True, **kw)
        torch.testing.assert_close(yr, torch.relu(y6), rtol=0, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,Cc', [(3, 20, 512), (2, 19, 1024)])
def test_conv_x6_epilogues_match_igemm(B, H, Cc):
    """csrc/conv_x6.hip with the Self_Attn epilogues -- per-channel scale + gate + second output + residual, and the merged projection's
    split-transposed store (flat and per-image descriptors) -- against conv_igemm / gemm_slot on the same descriptors."""
    from gssd import ops, _lib
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(B * 1000 + H)
    N, C4, C2 = H * H, Cc // 4, Cc // 2
    Np = ops.round_up(N, 4)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)
    x, ag = rnd(B, H, H, Cc), rnd(B, H, H, C2)
    w_o, w_p = rnd(Cc, C2) * 0.05, rnd(C4 + C2, Cc) * 0.05
    bias_o, bias_p, alpha_o, alpha_p = rnd(Cc), rnd(C4 + C2), torch.rand(Cc, generator=gen).to(dev) + 0.5, torch.rand(C4 + C2, generator=gen).to(dev) + 0.5
    gate = torch.tensor([0.37], device=dev)
    M = B * N
    res = {}
    for tag in ('ref', 'x6'):
        x6o = ops.x6_weight(w_o, 1, C2, 1, ops.x6_tile(Cc, 1, M)) if tag == 'x6' else None
        x6p = ops.x6_weight(w_p, 1, Cc, 1, ops.x6_tile(C4 + C2, 1, M)) if tag == 'x6' else None
        out, out2 = torch.empty(B, H, H, Cc, device=dev), torch.empty(B, H, H, Cc, device=dev)
        d5, _, _ = ops.make_conv_desc(ag, w_o, out, B=B, H=H, W=H, in_stride=C2, cin_g=C2, Cout=Cc, bias=bias_o, alpha=alpha_o, gate=gate, resid=x,
                                      out2=out2, wgt_x6=x6o)
        assert _lib.lib.gssd_conv_x6_takes(ctypes.byref(d5)) == (1 if tag == 'x6' else 0)
        ops.run_conv(d5)
        outs = [out, out2]
        for flat in ((True, False) if N % 4 == 0 else (False,)):
            tp, gT = torch.empty(B, N, C4, device=dev), torch.zeros(B, C2, Np, device=dev)
            d1, _, _ = ops.make_conv_desc(x, w_p, tp, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C4 + C2, bias=bias_p, alpha=alpha_p, wgt_x6=x6p,
                                          out_mode=_lib.OUT_SPLIT_T, out_b=gT, split_n=C4, out_stride=C4, out_b_stride=Np, m_per_image=not flat,
                                          in_batch_stride=N * Cc, out_batch_stride=N * C4, outb_batch_stride=C2 * Np, flags=_lib.CONV_OUT_F32)
            assert _lib.lib.gssd_conv_x6_takes(ctypes.byref(d1)) == (1 if tag == 'x6' else 0)
            ops.run_conv(d1)
            outs += [tp, gT]
        res[tag] = outs
    for a, b in zip(res['ref'], res['x6']):
        assert torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 19, 64, 108), (1, 38, 128, 24), (3, 9, 32, 128), (1, 17, 96, 36), (2, 8, 256, 4)])
def test_conv_patch_x6_matches_float64(shape):
    """csrc/conv_patch_x6.hip (round 6): the patch-staged direct 3x3 conv for many input channels and few outputs (the DCN offset conv, fp16 planes,
    GSSD_CONV_F16_OK launches) is as close to a float64 convolution as the fp32-MFMA implicit GEMM -- ragged tiles in both directions (W = H + 3),
    one to eight 32-channel chunks, output channel counts that end inside a 16-channel tile and inside a wave's 32, bias, ReLU."""
    import torch.nn.functional as F
    from gssd import ops
    B, H, Cin, Cout = shape
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, H + 3, Cin, generator=gen).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05).to(dev)
    b = torch.randn(Cout, generator=gen).to(dev)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), 1, 1).permute(0, 2, 3, 1)
    yp = ops.conv2d_nhwc(x, w, b, 1, 1, 1, 1, patch=True)
    y32 = ops.conv2d_nhwc(x, w, b, 1, 1, 1, 1)
    ep = float((yp.double() - ref).abs().max() / ref.abs().max())
    e32 = float((y32.double() - ref).abs().max() / ref.abs().max())
    print(f'conv_patch_x6 {shape}: {ep:.2e}; fp32-MFMA implicit GEMM {e32:.2e}')
    assert torch.isfinite(yp).all() and ep < 2e-6 and ep <= 1.5 * e32 + 1e-7
    yr = ops.conv2d_nhwc(x, w, b, 1, 1, 1, 1, patch=True, relu=True)
    torch.testing.assert_close(yr, torch.relu(yp), rtol=0, atol=0)


@pytest.mark.gpu
def test_conv_wino_x6_heads_epilogue():
    """csrc/conv_wino_x6.hip with GSSD_OUT_HEADS (round 6: the 38 x 38 multibox head of a train-mode forward): the merged loc | conf conv writes its
    channels [0, split_n) into loc and the rest into conf, each at the source's offset inside an image's priors -- against the implicit GEMM's heads
    epilogue on the same descriptor (and float64 on a slice).  >= 8 192 Winograd tiles and the GSSD_CONV_F16_OK flag are the host rule."""
    import ctypes
    import torch.nn.functional as F
    from gssd import ops, _lib
    lib = _lib.lib
    dev = torch.device('cuda:0')
    B, H, Cin, A, nc = 6, 75, 64, 4, 2
    nloc, nconf = 4 * A, nc * A
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(B, H, H, Cin, generator=gen).to(dev)
    w = (torch.randn(nloc + nconf, Cin, 3, 3, generator=gen) * 0.05).to(dev)
    b = torch.randn(nloc + nconf, generator=gen).to(dev)
    wp = ops.pack_weight(w)
    U = ops.winograd_weight(wp, 1, Cin)
    P = H * H * A + 40                                   # this source's priors sit behind 10 others' (40 = 10 x A)
    res = {}
    for tag in ('wino', 'igemm'):
        loc = torch.full((B, P, 4), float('nan'), device=dev)
        conf = torch.full((B, P, nc), float('nan'), device=dev)
        d, _, _ = ops.make_conv_desc(x, wp, loc, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin, Cout=nloc + nconf, k=3, pad=1, bias=b,
                                     out_mode=_lib.OUT_HEADS, out_b=conf, split_n=nloc, out_batch_stride=P * 4, outb_batch_stride=P * nc,
                                     out_off=10 * 4, outb_off=10 * nc, wgt_wino=U if tag == 'wino' else None,
                                     flags=_lib.CONV_OUT_F32 | (_lib.CONV_F16_OK if tag == 'wino' else 0))
        assert lib.gssd_conv_wino_x6_takes(ctypes.byref(d)) == (1 if tag == 'wino' else 0)
        ops.run_conv(d)
        torch.cuda.synchronize()
        res[tag] = (loc, conf)
    for a, c in zip(res['wino'], res['igemm']):
        assert torch.equal(torch.isnan(a), torch.isnan(c))                 # the same elements written (the other sources' priors untouched)
        m = ~torch.isnan(c)
        assert float((a[m] - c[m]).abs().max() / c[m].abs().max()) < 2e-5
    ref = F.conv2d(x[:1].double().permute(0, 3, 1, 2), w.double(), b.double(), 1, 1).permute(0, 2, 3, 1)      # [1, H, H, 24]
    loc0 = res['wino'][0][0, 10:10 + H * H * A].reshape(H, H, nloc).double().cpu()
    conf0 = res['wino'][1][0, 10:10 + H * H * A].reshape(H, H, nconf).double().cpu()
    assert float((loc0 - ref[0, :, :, :nloc].cpu()).abs().max() / ref.abs().max()) < 2e-5
    assert float((conf0 - ref[0, :, :, nloc:].cpu()).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.gpu
def test_x6_kernels_bf16_planes():
    """GSSD_X6_F16=0: the x6 kernels' forward launches on the three bf16 planes (round 5's form, what the data gradients always run) -- the same
    kernel tests in a child process."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', os.path.join(root, 'tests', 'test_gpu_kernels.py'), os.path.join(root, 'tests', 'test_gpu_thin_x6.py'),
                        '-k', 'test_conv_x6_matches_float64 or test_dcn_x6_matches_fused or test_conv_x6_epilogues or test_conv_thin_x6_forms'],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, GSSD_X6_F16='0'))
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_dcn_x6_one_part_per_tile():
    """GSSD_DCN_X6_SPLITK=0: launches of fewer tiles than CUs (every kernel test's) split the K loop over two workgroups per tile that meet in
    the output by atomic adds; the one-part form (what batch 32 runs) against the same references, in a child process."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSSD_DCN_X6_SPLITK='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', os.path.join(root, 'tests', 'test_gpu_kernels.py'), '-k', 'test_dcn_x6_matches_fused'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'fuzz_x6.py'), '8', '5', 'dcn'], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'dcn_x6: 8 shapes' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_x6_kernels_random_shapes():
    """scripts/fuzz_x6.py: dcn_x6 against dcn_fused and conv_x6 against the fp32-MFMA implicit GEMM on random shapes (non-square maps, every
    tile tail, 1 .. 4 channel blocks per tap, strides / dilations / 1x1 .. 5x5, with and without the fused input transform): the rotated
    K loops, their half buffers and counted waits must not depend on the shape."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'fuzz_x6.py'), '12', '11'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'conv_x6: 12 shapes' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

