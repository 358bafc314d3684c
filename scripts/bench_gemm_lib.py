"""How fast is the library fp32 GEMM (hipBLASLt behind torch.matmul) on the DCN contraction and the 1x1 fuse shapes?"""
import torch
dev = torch.device('cuda:0')
torch.backends.cuda.matmul.allow_tf32 = False
for (M, K, N, name) in ((46208, 9216, 512, 'dcn main'), (46208, 512, 9216, 'dcn dcols'), (46208, 512, 512, 'fuse_11'),
                        (11552, 1024, 1024, 'fuse_21'), (46208, 256, 512, 'sa o-conv')):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{name:10s} M={M} K={K} N={N}: {ms:.3f} ms  {2.0 * M * K * N / ms / 1e9:.1f} TFLOP/s')
