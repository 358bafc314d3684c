"""Randomised shape sweep of the round-5 K loops: dcn_x6 against dcn_fused, conv_x6 against the fp32-MFMA implicit GEMM (both fp32-equivalent:
agreement to fp32 summation-order noise), and -- leg `wino`, run with GSSD_WINO_X6=2 so that the kernel takes every shape it can --
conv_wino_x6 against the implicit GEMM.  usage (GPU box): python scripts/fuzz_x6.py [n_cases] [seed] [legs: dcn,conv | wino]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import numpy as np
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
legs = (sys.argv[3] if len(sys.argv) > 3 else 'dcn,conv').split(',')
worst = 0.0
for it in range(n_cases if 'dcn' in legs else 0):
    dg = int(rng.choice([1, 2, 4]))
    cpc = int(rng.choice([1, 2, 3, 4]))
    Cc = 32 * cpc * dg
    H, W = int(rng.integers(3, 23)), int(rng.integers(3, 23))
    B = int(rng.integers(1, 4))
    Cout = int(rng.choice([8, 24, 40, 128, 136, 256, 264, 520]))
    x = torch.from_numpy(rng.normal(size=(B, H, W, Cc)).astype(np.float32)).to(dev)
    om = torch.from_numpy(rng.normal(0, float(rng.choice([0.3, 1.5, 4.0])), size=(B, H, W, 27 * dg)).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.normal(0, 0.05, size=(Cout, Cc, 3, 3)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.normal(size=Cout).astype(np.float32)).to(dev)
    f16 = bool(rng.random() < 0.5)              # GSSD_CONV_F16_OK: the fp16-plane form (a train-mode forward's) or the bf16 planes
    if H != W:
        # ops.dcn_forward takes square maps only through its convenience wrapper: use the C ABI for both
        from gssd._lib import lib, check
        s = torch.cuda.current_stream().cuda_stream
        wp = ops.dcn_pack_weight(w, dg)
        n6 = int(lib.gssd_dcn_packed_weight_elems_x6(Cout, Cc))
        wp6 = torch.empty(n6, device=dev, dtype=torch.bfloat16)
        check(lib.gssd_dcn_pack_weight_x6(w.contiguous().data_ptr(), wp6.data_ptr(), Cout, Cc, dg, s))
        ref = torch.full((B, H, W, Cout), float('nan'), device=dev)
        got = torch.full((B, H, W, Cout), float('nan'), device=dev)
        check(lib.gssd_dcn_forward_f32(x.data_ptr(), om.data_ptr(), wp.data_ptr(), b.data_ptr(), ref.data_ptr(), B, H, W, Cc, dg, 27 * dg, Cout, s))
        check(lib.gssd_dcn_forward_x6_ex(x.data_ptr(), om.data_ptr(), wp6.data_ptr(), b.data_ptr(), got.data_ptr(), B, H, W, Cc, dg, 27 * dg, Cout,
                                         _lib.CONV_F16_OK if f16 else 0, s))
    else:
        ref = ops.dcn_forward(x, om, w, b, dg)
        got = ops.dcn_forward_x6(x, om, w, b, dg, f16ok=f16)
    torch.cuda.synchronize()
    e = float((got - ref).abs().max() / ref.abs().max())
    worst = max(worst, e)
    ok = torch.isfinite(got).all() and e < 2e-5
    print(f'dcn  B {B} {H}x{W} C {Cc} dg {dg} Cout {Cout} f16 {f16}: rel {e:.2e}' + ('' if ok else '   <-- FAIL'), flush=True)
    assert ok
if 'dcn' in legs:
    print(f'dcn_x6: {n_cases} shapes, worst rel {worst:.2e}')
worst = 0.0
for it in range(n_cases if 'conv' in legs else 0):
    g = int(rng.choice([1, 1, 4]))
    cin_g = 32 * int(rng.integers(1, 5))
    cout_g = int(rng.choice([32, 40, 64, 72, 128, 136, 256]))
    k = int(rng.choice([1, 3, 3, 5]))
    dil = int(rng.choice([1, 1, 2])) if k == 3 else 1
    pad = dil * (k // 2) if rng.random() < 0.8 else 0
    stride = int(rng.choice([1, 1, 2]))
    H, W = int(rng.integers(max(3, k * dil), 26)), int(rng.integers(max(3, k * dil), 26))
    B = int(rng.integers(1, 4))
    x = torch.from_numpy(rng.normal(size=(B, H, W, g * cin_g)).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.normal(0, 0.1, size=(g * cout_g, cin_g, k, k)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.normal(size=g * cout_g).astype(np.float32)).to(dev)
    xf = rng.random() < 0.5
    kw = {}
    if xf:
        sc = torch.from_numpy((rng.random(g * cin_g) + 0.5).astype(np.float32)).to(dev)
        sh = torch.from_numpy((rng.normal(size=g * cin_g) * 0.3).astype(np.float32)).to(dev)
        kw = dict(in_scale=sc, in_shift=sh, in_pad=-sh / sc - 1.0)
    f16ok = (not xf) and rng.random() < 0.5                    # plain launches: bf16 planes (data gradients) or, flagged, fp16 planes (forward)
    y6 = ops.conv2d_nhwc(x, w, b, stride, pad, dil, g, x6=True, **kw, **(dict(flags=_lib.CONV_F16_OK) if f16ok else {}))
    y32 = ops.conv2d_nhwc(x, w, b, stride, pad, dil, g, **kw)
    torch.cuda.synchronize()
    e = float((y6 - y32).abs().max() / y32.abs().max())
    worst = max(worst, e)
    ok = torch.isfinite(y6).all() and e < 2e-5
    print(f'conv B {B} {H}x{W} cin_g {cin_g} cout_g {cout_g} g {g} k {k} pad {pad} dil {dil} stride {stride} xf {xf} f16 {xf or f16ok}: rel {e:.2e}' + ('' if ok else '   <-- FAIL'), flush=True)
    assert ok
if 'conv' in legs:
    print(f'conv_x6: {n_cases} shapes, worst rel {worst:.2e}')
if 'wino' in legs:
    import ctypes as C
    from gssd import _lib
    assert os.environ.get('GSSD_WINO_X6') == '2', 'run the wino leg with GSSD_WINO_X6=2'
    worst = 0.0
    for it in range(n_cases):
        g = int(rng.choice([1, 1, 4]))
        cin_g = 16 * int(rng.integers(1, 9))
        cout_g = int(rng.choice([24, 32, 40, 64, 72, 96, 108, 128, 136])) if g == 1 else int(rng.choice([32, 64, 96, 128]))
        H, W = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        B = int(rng.integers(1, 5))
        x = torch.from_numpy(rng.normal(0.1, 1.0, size=(B, H, W, g * cin_g)).astype(np.float32)).to(dev)
        w = torch.from_numpy(rng.normal(0, 0.1, size=(g * cout_g, cin_g, 3, 3)).astype(np.float32)).to(dev)
        b = torch.from_numpy(rng.normal(size=g * cout_g).astype(np.float32)).to(dev)
        xf, rs = rng.random() < 0.5, rng.random() < 0.4
        kw = {}
        if xf:
            sc = torch.from_numpy(((rng.random(g * cin_g) + 0.5) * rng.choice([-1.0, 1.0], size=g * cin_g)).astype(np.float32)).to(dev)
            sh = torch.from_numpy((rng.normal(size=g * cin_g) * 0.3).astype(np.float32)).to(dev)
            kw = dict(in_scale=sc, in_shift=sh, in_pad=torch.where(sc > 0, torch.full_like(sc, -3.0e38), torch.full_like(sc, 3.0e38)))
        if rs:
            kw['resid'] = torch.from_numpy(rng.normal(size=(B, H, W, g * cout_g)).astype(np.float32)).to(dev)
        keep = []
        d = ops.conv2d_nhwc(x, w, b, 1, 1, 1, g, winograd=True, _keep=keep, **kw)
        assert _lib.lib.gssd_conv_wino_x6_takes(C.byref(d)) == 1, (cin_g, cout_g, g)
        ops.run_conv(d)
        yw = keep[-1]
        y32 = ops.conv2d_nhwc(x, w, b, 1, 1, 1, g, **kw)
        torch.cuda.synchronize()
        e = float((yw - y32).abs().max() / y32.abs().max())
        worst = max(worst, e)
        ok = torch.isfinite(yw).all() and e < 2e-5
        print(f'wino B {B} {H}x{W} cin_g {cin_g} cout_g {cout_g} g {g} xf {xf} resid {rs}: rel {e:.2e}' + ('' if ok else '   <-- FAIL'), flush=True)
        assert ok
    print(f'conv_wino_x6: {n_cases} shapes, worst rel {worst:.2e}')
