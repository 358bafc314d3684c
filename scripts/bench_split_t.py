import os, sys
sys.path.insert(0, '/root/repo/grouped-ssd-pytorch_amd')
import torch, ctypes as C
from gssd import ops, _lib
lib = _lib.lib
dev = torch.device('cuda:0')
B, H, Cc = 32, 38, 512
N = H * H
C4, C2 = 128, 256
x = torch.randn(B, H, H, Cc, device=dev)
w = torch.randn(C4 + C2, Cc, device=dev) * 0.05
b = torch.randn(C4 + C2, device=dev)
bn = ops.x6_tile(C4 + C2, 1, B * N)
w6 = ops.x6_weight(w, 1, Cc, 1, bn)
tp = torch.empty(B, N, C4, device=dev); gT = torch.zeros(B, C2, N, device=dev)
full = torch.empty(B, N, C4 + C2, device=dev)
d_split, _, _ = ops.make_conv_desc(x, w, tp, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C4 + C2, bias=b, wgt_x6=w6, out_mode=_lib.OUT_SPLIT_T, out_b=gT,
                                   split_n=C4, out_stride=C4, out_b_stride=N, m_per_image=False, in_batch_stride=N * Cc, out_batch_stride=N * C4,
                                   outb_batch_stride=C2 * N, flags=_lib.CONV_OUT_F32)
d_nhwc, _, _ = ops.make_conv_desc(x, w, full, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C4 + C2, bias=b, wgt_x6=w6)
print('takes x6:', lib.gssd_conv_x6_takes(C.byref(d_split)), lib.gssd_conv_x6_takes(C.byref(d_nhwc)))
for name, d in (('split-T', d_split), ('NHWC', d_nhwc)):
    for _ in range(3): ops.run_conv(d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.run_conv(d)
    e1.record(); torch.cuda.synchronize()
    print(name, f'{e0.elapsed_time(e1) / 20 * 1e3:.1f} us')
print('max diff g', float((gT.permute(0, 2, 1) - full[..., C4:]).abs().max()), 'tp', float((tp - full[..., :C4]).abs().max()))
