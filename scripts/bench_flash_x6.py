"""flash_attn_x6 (the fp32 mode's attention core on the bf16 matrix cores) at the GSSD++ shape: B = 32, N = 38 x 38, D = 64, C2 = 256; us per call
incl. the split passes.  A/B builds through GSSD_LIB_PATH (scripts/flash_x6_ab.sh)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import _lib
lib = _lib.lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, N, D, C2 = 32, 1444, 64, 256
Np = (N + 3) // 4 * 4
tp = (torch.randn(B, N, 2 * D, device=dev) * 1.9)
gT = torch.randn(B, C2, Np, device=dev)
out, lse = torch.empty(B, N, C2, device=dev), torch.empty(B, N, device=dev)
ws = torch.empty(int(lib.gssd_self_attn_core_x6_ws_bytes(B, N, D, C2)) // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(): _lib.check(lib.gssd_self_attn_core_x6_f32(tp.data_ptr(), gT.data_ptr(), out.data_ptr(), B, N, Np, D, C2, ws.data_ptr(), lse.data_ptr(), st))
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f'flash_attn_x6 B={B} N={N}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call incl. split passes; checksum {float(out.double().sum()):.6f}')
