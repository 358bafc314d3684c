"""Latency of the small-map conv launches (the tail of a step: <= 10 x 10 maps at batch 32), timed as 50 launches replayed from one
hipGraph (back-to-back dependent launches like inside the forward plan; eager timing is host-bound at these sizes).
usage: python3 scripts/bench_small_conv.py [f32|bf16]      (GSSD_NO_SMALL_TILES=1 for the 128-row tiles)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import ctypes as C
import torch
from gssd import ops, _lib
lib = _lib.lib
B = 32
bf16 = len(sys.argv) > 1 and sys.argv[1] == 'bf16'
# name, H, Cin, Cout, k, stride, pad, groups
LAYERS = [('sa_tpg 10x10', 10, 512, 384, 1, 1, 0, 1), ('sa_out 10x10', 10, 256, 512, 1, 1, 0, 1), ('fuse 10x10', 10, 512, 512, 1, 1, 0, 1),
          ('ext1 3x3s2 19->10', 19, 256, 512, 3, 2, 1, 4), ('ext2 1x1 10x10', 10, 512, 128, 1, 1, 0, 4), ('ext3 3x3s2 10->5', 10, 128, 256, 3, 2, 1, 4),
          ('sa_tpg 5x5', 5, 256, 192, 1, 1, 0, 1), ('sa_out 5x5', 5, 128, 256, 1, 1, 0, 1), ('ext4 1x1 5x5', 5, 256, 128, 1, 1, 0, 4),
          ('ext5 3x3 5->3', 5, 128, 256, 3, 1, 0, 4), ('sa_tpg 3x3', 3, 256, 192, 1, 1, 0, 1), ('ext7 3x3 3->1', 3, 128, 256, 3, 1, 0, 4),
          ('sa_tpg 1x1', 1, 256, 192, 1, 1, 0, 1), ('sa_out 1x1', 1, 128, 256, 1, 1, 0, 1), ('fuse 1x1', 1, 256, 256, 1, 1, 0, 1)]
dev = torch.device('cuda:0')
tot = 0.0
for (name, H, Cin, Cout, k, s, p, g) in LAYERS:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    Ho = (H + 2 * p - (k - 1) - 1) // s + 1
    stats = torch.zeros(2 * Cout, device=dev, dtype=torch.float64)
    if bf16:
        x = x.to(torch.bfloat16)
        wp = ops.pack_weight_bf16(w)
        out = torch.empty(B, Ho, Ho, Cout, device=dev, dtype=torch.bfloat16)
        fn = lib.gssd_conv2d_nhwc_bf16
    else:
        wp = ops.pack_weight(w)
        out = torch.empty(B, Ho, Ho, Cout, device=dev)
        fn = lib.gssd_conv2d_nhwc_f32
    d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // g, Cout=Cout, groups=g, k=k, stride=s, pad=p,
                                 bias=bias, stats=stats)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(fn(C.byref(d), st))
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cs = torch.cuda.current_stream().cuda_stream
        for _ in range(50):
            _lib.check(fn(C.byref(d), cs))
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    tot += us
    print(f'{name:22s} M {B * Ho * Ho:6d} N {Cout:4d} K {k * k * Cin // g:5d}   {us:7.1f} us per launch')
print(f'sum {tot:.1f} us')
