"""conv1_2 / conv2_1 / conv2_2 at batch 32 (fused producer BatchNorm + ReLU, batch sums, the pooled epilogue where the net has one): the kernel
gssd_conv2d_nhwc_f32 dispatches to under the current GSSD_THIN_X6 (run once with =0 and once without for the A/B).
usage (GPU box): python scripts/bench_thin_x6.py [batch]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(3)
for name, H, Cin, Cout, pool in (('conv1_2', 300, 64, 64, True), ('conv2_1', 150, 64, 128, False), ('conv2_2', 150, 128, 128, True), ('conv3_1', 75, 128, 256, False)):
    x = torch.randn(B, H, H, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin // 4, 3, 3, generator=g) * 0.1).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(dev), (torch.randn(Cin, generator=g) * 0.3).to(dev)
    buf = torch.empty(B * H * H * Cin + Cin, device=dev)          # the engine's layout: the padding vector directly behind the map
    buf[:B * H * H * Cin] = x.reshape(-1)
    buf[B * H * H * Cin:] = -sh / sc - 1.0
    wp = ops.pack_weight(w)
    U = ops.winograd_weight(wp, 4, Cin // 4)
    Ho = (H + 1) // 2 if pool else H
    out = torch.empty(B, Ho, Ho, Cout, device=dev)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    d, _, _ = ops.make_conv_desc(buf[:B * H * H * Cin].view(B, H, H, Cin), wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // 4, Cout=Cout, groups=4, k=3, pad=1,
                                 bias=b, stats=stats, wgt_wino=U, in_scale=sc, in_shift=sh, in_pad=buf[B * H * H * Cin:],
                                 flags=(_lib.CONV_POOL2 if pool else 0) | (0 if os.environ.get('F16OK') == '0' else _lib.CONV_F16_OK), pool_sign=torch.ones(Cout, device=dev) if pool else None)
    takes = _lib.lib.gssd_conv_thin_x6_takes(C.byref(d))
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(_lib.lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    n = 20
    for _ in range(n):
        _lib.check(_lib.lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * H * Cout * 9 * (Cin // 4)
    by = 4.0 * (B * H * H * Cin + B * Ho * Ho * Cout)
    raw = C.CDLL(_lib.LIB_PATH)
    if hasattr(raw, 'gssd_thin_x6_timing_read'):
        t8 = (C.c_ulonglong * 8)()
        raw.gssd_thin_x6_timing_read(t8)
        nwg = max(1, t8[7])
        tot = sum(t8[k] for k in range(5)) or 1
        names = ['wait loads', 'transform+write', 'barrier 1', 'MFMA+epilogue', 'barrier 2']
        tiles = B * ((H + 7) // 8) * ((H + 15) // 16) * (1 if Cin // 4 == 16 and Cout // 4 == 16 else 2 if Cin // 4 == 16 else 4)      # work items of a launch
        per_item = [t8[k] / (23.0 * tiles) * 10.0 for k in range(5)]          # ns per work item (100 MHz stamps, 3 + 20 launches)
        print('   per-phase time of a workgroup (wave 1), ns per tile: ' + '  '.join(f'{n} {per_item[k]:.0f}' for k, n in enumerate(names)) +
              f'   sum {sum(per_item):.0f} ns;  workgroups per launch {nwg / 23.0:.0f}')
    print(f'{name} B={B} thin_x6={takes}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s direct  {by / us / 1e3:7.1f} GB/s', flush=True)
