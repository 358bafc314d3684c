"""Fused deformable conv kernel on the GSSD++ shape (B=32, 38x38, 1024 -> 512, 4 deformable groups), timed alone."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops
from gssd._lib import lib, check
dev = torch.device('cuda:0')
B, H, C, Cout, dg = int(os.environ.get('B', 32)), 38, 1024, 512, 4
torch.manual_seed(0)
x = torch.randn(B, H, H, C, device=dev)
om = torch.randn(B, H, H, 27 * dg, device=dev) * float(os.environ.get('OMS', 0.8))
w = torch.randn(Cout, C, 3, 3, device=dev) * 0.01
bias = torch.randn(Cout, device=dev)
wp = ops.dcn_pack_weight(w, dg)
out = torch.empty(B, H, H, Cout, device=dev)
s = torch.cuda.current_stream().cuda_stream
def run():
    check(lib.gssd_dcn_forward_f32(x.data_ptr(), om.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, H, C, dg, 27 * dg, Cout, s))
for _ in range(3): run()
torch.cuda.synchronize()
n = int(os.environ.get('N', 20))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * B * H * H * Cout * 9 * C
print(f'dcn_fused B={B}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  ({fl / ms / 1e9 / 157.3:.3f} of fp32 peak)')
