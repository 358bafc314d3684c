"""Micro-benchmark of gssd_dcn_col2im_f32 at the GSSD++ shape for several offset magnitudes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops
dev = torch.device('cuda:0')
B, H, C, dg = 32, 38, 1024, 4
x = torch.randn(B, H, H, C, device=dev)
dcols = torch.randn(B * H * H, 9 * C, device=dev)
for std in (0.0, 1.0, 3.0, 10.0):
    om = torch.randn(B, H, H, 27 * dg, device=dev) * std
    dx = torch.zeros_like(x); dom = torch.zeros_like(om)
    for _ in range(2): ops.dcn_col2im(x, om, dcols, dx, dom, dg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.dcn_col2im(x, om, dcols, dx, dom, dg)
    e1.record(); torch.cuda.synchronize()
    print(f'offset std {std}: {e0.elapsed_time(e1) / 5:.3f} ms')
if len(sys.argv) > 1:
    import bench  # noqa
