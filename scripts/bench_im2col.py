"""Times gssd_dcn_im2col_f32 / _bf16 at the GSSD++ shape (B = 32, 38 x 38, 1024 channels, 4 deformable groups)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import _lib
dev = torch.device('cuda:0')
B, H, C, dg, OMC = 32, 38, 1024, 4, 112
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, H, C, device=dev)
om = torch.randn(B, H, H, OMC, device=dev) * 0.5
for name, xx, cols, fn in (('f32', x, torch.empty(B * H * H, 9 * C, device=dev), _lib.lib.gssd_dcn_im2col_f32),
                           ('bf16', x.to(torch.bfloat16), torch.empty(B * H * H, 9 * C, device=dev, dtype=torch.bfloat16), _lib.lib.gssd_dcn_im2col_bf16)):
    for _ in range(3):
        _lib.check(fn(xx.data_ptr(), om.data_ptr(), cols.data_ptr(), B, H, H, C, dg, OMC, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(fn(xx.data_ptr(), om.data_ptr(), cols.data_ptr(), B, H, H, C, dg, OMC, st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'im2col {name}: {ms:.3f} ms = {cols.numel() * cols.element_size() / ms / 1e9:.2f} TB/s of column writes')
