"""Subprocess of tests/test_gpu_wino_x6.py: every Winograd launch form of gssd_conv2d_nhwc_f32 on the shapes given below, under the
GSSD_WINO_X6 mode the parent put into the environment (read once per process: 2 = csrc/conv_wino_x6.hip takes every shape it can, 0 = the
fp32-MFMA kernel csrc/conv_wino.hip keeps all of them), each against a float64 convolution on the CPU.  Forms: plain + bias + batch sums
(one array and 8 replicas), the producer's BatchNorm + ReLU fused into the patch loads (padding vector anywhere -> per-element select; padding
vector directly behind the map, the engine's layout -> address select), the pooled-raw epilogue (GSSD_CONV_POOL2), and the data gradient
form (flipped weights, resid = an existing gradient: bwd_ops.py:91-96).  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')):
    sys.path.insert(0, p)
import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.nn.functional as F         # noqa: E402

CASES = [
    # B, H, W, Cin, Cout, groups          (3x3 / stride 1 / pad 1)
    (2, 38, 38, 512, 512, 4),      # conv4_2: two 64-channel blocks per group, 4 chunks
    (3, 19, 19, 512, 512, 4),      # conv5_x: odd map (ragged last tile row / column), the tile list crosses images
    (2, 75, 75, 128, 256, 4),      # conv3_1: cin_g 32 = one chunk, odd map
    (2, 37, 37, 128, 128, 4),      # conv2_2 class: 32-channel blocks
    (2, 41, 29, 64, 128, 4),       # conv2_1 class: cin_g 16 = half a chunk of padding slots, non-square
    (5, 9, 9, 64, 32, 1),          # dense, one 32-channel block; fewer tiles than one item
    (1, 150, 150, 128, 128, 4),    # many items per persistent workgroup
    (2, 38, 38, 256, 108, 1),      # DCN offset / mask conv: 108 channels = two 64-blocks with 20 padding rows
    (2, 13, 17, 64, 24, 1),        # 24 channels padded to one 32-channel block
    (1, 22, 30, 96, 72, 1),        # cin_g 96 = three chunks; 72 channels = a whole block + 8 rows
    (3, 20, 20, 256, 256, 4),      # even map: every 2x2 pool window whole
]


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.abs().max()), 1e-30))


def main():
    from gssd import ops, _lib
    lib = _lib.lib
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    only = [int(a) for a in sys.argv[1:]]
    res = []
    for ci, (B, H, W, Cin, Cout, g) in enumerate(CASES):
        if only and ci not in only:
            continue
        rng = np.random.default_rng(1000 + ci)
        cin_g, cout_g = Cin // g, Cout // g
        x = torch.from_numpy(rng.normal(0.1, 1.0, size=(B, Cin, H, W)).astype(np.float32))
        w = torch.from_numpy(rng.normal(0, 0.1, size=(Cout, cin_g, 3, 3)).astype(np.float32))
        b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
        sidx = torch.from_numpy(rng.integers(0, B * H * W * Cout, 512))
        out = dict(case=[B, H, W, Cin, Cout, g], forms={})
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        wp = ops.pack_weight(w.to(dev))
        U = ops.winograd_weight(wp, g, cin_g)
        kw = dict(B=B, H=H, W=W, in_stride=Cin, cin_g=cin_g, Cout=Cout, groups=g, k=3, stride=1, pad=1, bias=b.to(dev), wgt_wino=U)

        def launch(inp, o, **extra):
            # WX6_TEST_F16OK=1: every forward launch carries GSSD_CONV_F16_OK (the fp16-plane form of the kernel; the data gradients never do)
            extra['flags'] = extra.get('flags', 0) | (_lib.CONV_F16_OK if os.environ.get('WX6_TEST_F16OK') == '1' else 0)
            d, _, _ = ops.make_conv_desc(inp, wp, o, **{**kw, **extra})
            takes = int(lib.gssd_conv_wino_x6_takes(C.byref(d)))
            _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
            torch.cuda.synchronize()
            return takes

        def record(name, got, ref64, takes, **more):
            got = got.cpu()
            assert torch.isfinite(got).all(), (name, out['case'])
            out['forms'][name] = dict(takes=takes, err=rel(got, ref64), sample=got.reshape(-1)[sidx % got.numel()].tolist(),
                                      ref_max=float(ref64.abs().max()), **more)

        # ---- plain + bias + batch sums (one array, then 8 replicas) ----
        ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1, 1, g).permute(0, 2, 3, 1).contiguous()
        n_px = B * H * W
        for R in (0, 8):
            y = torch.full((B, H, W, Cout), float('nan'), device=dev)
            stats = torch.zeros(max(R, 1) * 2 * Cout, dtype=torch.float64, device=dev)
            takes = launch(xd, y, stats=stats, stats_rep=R)
            s = stats.view(max(R, 1), 2 * Cout).sum(0).cpu()
            record('plain' if R == 0 else 'plain_rep8', y, ref, takes,
                   stat_err=[float((s[:Cout] - ref.sum((0, 1, 2))).abs().max() / (n_px * float(ref.abs().max()))),
                             float((s[Cout:] - (ref * ref).sum((0, 1, 2))).abs().max() / (n_px * float(ref.abs().max()) ** 2))])
        y_plain = y

        # ---- fused producer BatchNorm + ReLU: scales of both signs, padding = a value the transform maps to 0 ----
        scv = torch.from_numpy(rng.uniform(0.2, 1.5, size=Cin).astype(np.float32)) * torch.from_numpy(rng.choice([-1.0, 1.0], size=Cin).astype(np.float32))
        shv = torch.from_numpy(rng.normal(size=Cin).astype(np.float32))
        pdv = torch.where(scv > 0, torch.full_like(scv, -3.0e38), torch.full_like(scv, 3.0e38))
        act64 = torch.relu(torch.addcmul(shv.double().view(1, -1, 1, 1), x.double(), scv.double().view(1, -1, 1, 1)))
        refx = F.conv2d(act64, w.double(), b.double(), 1, 1, 1, g).permute(0, 2, 3, 1).contiguous()
        sc, sh, pd_sep = scv.to(dev), shv.to(dev), pdv.to(dev)
        y = torch.full((B, H, W, Cout), float('nan'), device=dev)
        takes = launch(xd, y, in_scale=sc, in_shift=sh, in_pad=pd_sep)
        record('xf_select', y, refx, takes)
        y_sel = y
        buf = torch.empty(B * H * W * Cin + Cin, device=dev)          # the engine's layout: the padding vector directly behind the dense map
        buf[:B * H * W * Cin] = xd.reshape(-1)
        buf[B * H * W * Cin:] = pd_sep
        y = torch.full((B, H, W, Cout), float('nan'), device=dev)
        takes = launch(buf[:B * H * W * Cin].view(B, H, W, Cin), y, in_scale=sc, in_shift=sh, in_pad=buf[B * H * W * Cin:])
        record('xf_address', y, refx, takes, equal_to_select=bool(torch.equal(y, y_sel)))

        # ---- pooled-raw epilogue: max / min by the sign of the next BatchNorm's weight, batch sums of the FULL map ----
        gamma = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
        gamma[min(5, Cout - 1)] = 0.0
        Hp, Wp = (H + 1) // 2, (W + 1) // 2

        def pool_by_sign(raw_nhwc):
            r = raw_nhwc.permute(0, 3, 1, 2)
            mx, mn = F.max_pool2d(r, 2, 2, 0, ceil_mode=True), -F.max_pool2d(-r, 2, 2, 0, ceil_mode=True)
            return torch.where(gamma.to(r.dtype).view(1, -1, 1, 1) >= 0, mx, mn).permute(0, 2, 3, 1).contiguous()
        for xf in (False, True):
            yp = torch.full((B, Hp, Wp, Cout), float('nan'), device=dev)
            stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
            extra = dict(in_scale=sc, in_shift=sh, in_pad=pd_sep) if xf else {}
            takes = launch(xd, yp, stats=stats, flags=_lib.CONV_POOL2, pool_sign=gamma.to(dev), **extra)
            full = (y_sel if xf else y_plain).cpu()
            r64 = refx if xf else ref
            s = stats.cpu()
            record('pool_xf' if xf else 'pool', yp, pool_by_sign(r64), takes,
                   pooled_image_of_plain=bool(torch.equal(yp.cpu(), pool_by_sign(full))),
                   stat_err=[float((s[:Cout] - r64.sum((0, 1, 2))).abs().max() / (n_px * float(r64.abs().max()))),
                             float((s[Cout:] - (r64 * r64).sum((0, 1, 2))).abs().max() / (n_px * float(r64.abs().max()) ** 2))])

        # ---- data gradient: dX = existing + conv(dY, flipped weights) (bwd_ops.py:91-96), when the transposed shape is a Winograd shape ----
        if ops.winograd_eligible(3, 1, 1, 1, cout_g, cin_g, g):
            dy = torch.from_numpy(rng.normal(size=(B, Cout, H, W)).astype(np.float32))
            existing = torch.from_numpy(rng.normal(size=(B, H, W, Cin)).astype(np.float32))
            dref = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1, 0, g).permute(0, 2, 3, 1).contiguous()
            wd = ops.pack_weight_dgrad(w.to(dev), g)
            ud = ops.winograd_weight(wd, g, cout_g)
            dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
            for with_resid in (True, False):
                dx = torch.full((B, H, W, Cin), float('nan'), device=dev)
                ex = existing.to(dev) if with_resid else None
                dd, _, _ = ops.make_conv_desc(dyd, wd, dx, B=B, H=H, W=W, in_stride=Cout, cin_g=cout_g, Cout=Cin, groups=g, k=3, pad=1, resid=ex,
                                              wgt_wino=ud)
                takes = int(lib.gssd_conv_wino_x6_takes(C.byref(dd)))
                ops.run_conv(dd)
                torch.cuda.synchronize()
                record('dgrad_resid' if with_resid else 'dgrad', dx, dref + existing.double() if with_resid else dref, takes)
        res.append(out)
    print('WINOX6JSON ' + json.dumps(dict(mode=os.environ.get('GSSD_WINO_X6', ''), f16ok=os.environ.get('WX6_TEST_F16OK', ''), results=res)))


if __name__ == '__main__':
    main()
