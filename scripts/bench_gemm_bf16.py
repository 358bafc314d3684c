"""Time gssd_conv2d_nhwc_bf16 on the GEMM-shaped launches of the bf16 mode (1x1 convs: the fuse convs, the Self_Attn projections and output
convs at 38 x 38 and 19 x 19, conv7): python scripts/bench_gemm_bf16.py   (GSSD_BF16_BIG_TILES=N selects the tile sweep of csrc/conv_bf16.hip)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
shapes = [(46208, 512, 512), (46208, 512, 1024), (46208, 512, 256), (46208, 384, 512), (46208, 256, 1024), (11552, 1024, 1024), (11552, 768, 1024), (11552, 1024, 512)]
tot = 0.0
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(2 * N, device=dev, dtype=torch.float64)
    d, _, _ = ops.make_conv_desc(x, w, out, B=1, H=M, W=1, in_stride=K, cin_g=K, Cout=N, stats=stats)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): _lib.check(_lib.lib.gssd_conv2d_nhwc_bf16(__import__('ctypes').byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _lib.check(_lib.lib.gssd_conv2d_nhwc_bf16(__import__('ctypes').byref(d), st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tot += ms
    ref = (x[:4096].float() @ w.float().t())
    err = float((out[:4096].float() - ref).abs().max() / ref.abs().max())
    print(f'M {M} N {N} K {K}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:6.1f} TF   (err {err:.1e})', flush=True)
print(f'sum {tot * 1e3:.1f} us')
