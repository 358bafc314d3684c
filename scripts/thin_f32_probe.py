"""fp32 thin trunk layers (conv1_1 conv_thin<4,16>, conv1_2 conv_thin_wino, conv2_1 conv_thin<16,32>) standalone at B = 32: time with
batch sums in 32 replicas / one array / no batch sums at all -- is the end-of-launch atomics tail of the persistent kernels visible here?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import ctypes as C
import torch
from gssd import ops, _lib
lib = _lib.lib
dev = torch.device('cuda:0')
B = 32
for (name, H, Cin, Cout, xf, pool, wino) in (('conv1_1', 300, 16, 64, False, False, False), ('conv1_2', 300, 64, 64, True, False, True),
                                             ('conv1_2/pool', 300, 64, 64, True, True, True), ('conv2_1', 150, 64, 128, True, False, False),
                                             ('conv2_2/pool', 150, 128, 128, True, True, True), ('conv3_1', 75, 128, 256, True, False, True)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin // 4, 3, 3, device=dev) * 0.1
    wp = ops.pack_weight(w)
    U = ops.winograd_weight(wp, 4, Cin // 4) if wino else None
    b = torch.zeros(Cout, device=dev)
    Ho = (H + 1) // 2 if pool else H
    out = torch.empty(B, Ho, Ho, Cout, device=dev)
    sc, sh = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    pdv = torch.zeros(Cin, device=dev)
    sign = torch.randn(Cout, device=dev)
    res = []
    for R in (32, 0, None):
        stats = None if R is None else torch.zeros(max(R, 1) * 2 * Cout, dtype=torch.float64, device=dev)
        d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // 4, Cout=Cout, groups=4, k=3, stride=1, pad=1,
                                     bias=b, stats=stats, stats_rep=R or 0, in_scale=sc if xf else None, in_shift=sh if xf else None,
                                     in_pad=pdv if xf else None, wgt_wino=U, flags=_lib.CONV_POOL2 if pool else 0,
                                     pool_sign=sign if pool else None)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    byts = 4.0 * B * (H * H * Cin + Ho * Ho * Cout)
    print(f'{name:14s} 32 replicas {res[0]:7.1f} us | one array {res[1]:7.1f} us | no batch sums {res[2]:7.1f} us   ({byts / res[0] / 1e3:.0f} GB/s of compulsory bytes)')
