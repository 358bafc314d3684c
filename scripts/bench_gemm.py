"""Time gssd_conv2d_nhwc_f32 on plain 1x1 shapes: python scripts/bench_gemm.py  (GSSD_NO_GEMM_SLOT=1 for the generic kernel)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import ctypes as C
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
shapes = [(46208, 512, k) for k in (256, 512, 1024, 4096)] + [(46208, 1024, 512), (46208, 256, 1024), (46208, 384, 512), (11552, 1024, 512),
                                                              (11552, 768, 1024), (11552, 1024, 2048), (3200, 512, 1024), (46208, 9216, 512)]
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    out = torch.empty(M, N, device=dev)
    d, _, _ = ops.make_conv_desc(x, w, out, B=1, H=M, W=1, in_stride=K, cin_g=K, Cout=N)
    takes = _lib.lib.gssd_gemm_slot_takes(C.byref(d))
    for _ in range(3): ops.run_conv(d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.run_conv(d)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'M {M} N {N} K {K} slot {takes}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:6.1f} TF')
