"""gssd_self_attn_core_f32 alone (B = 32, N = 1444, theta/phi 64, g 256): timing + a target for scripts/pmc_kernel.sh."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import torch
from gssd import _lib
lib = _lib.lib
dev = torch.device('cuda:0')
B, N, D, C2 = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1444, 64, 256
Np = (N + 3) // 4 * 4
tp = torch.randn(B, N, 2 * D, device=dev)
gT = torch.randn(B, C2, Np, device=dev)
out = torch.empty(B, N, C2, device=dev)
st = torch.cuda.current_stream().cuda_stream
run = lambda: _lib.check(lib.gssd_self_attn_core_f32(tp.data_ptr(), gT.data_ptr(), out.data_ptr(), B, N, Np, D, C2, 0, st))
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'flash N {N}: {ms * 1e3:.1f} us  {2.0 * B * N * N * (D + C2) / ms / 1e9:.1f} TF')
