"""Times gssd_conv2d_wgrad_bf16 on the trunk shapes of GSSD / GSSD++ at batch 32 beside the fp32 kernels that ran them before."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
from gssd import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
B = int(os.environ.get('B', 32))
SHAPES = [('conv1_2', 300, 64, 64), ('conv2_1', 150, 64, 128), ('conv2_2', 150, 128, 128), ('conv3_1', 75, 128, 256),
          ('conv3_2', 75, 256, 256), ('conv4_1', 38, 256, 512), ('conv4_2', 38, 512, 512), ('conv5_1', 19, 512, 512)]
st = torch.cuda.current_stream().cuda_stream
for name, H, Cin, Cout in SHAPES:
    cg = Cin // 4
    x = torch.randn(B, H, H, Cin, device=dev)
    dy = torch.randn(B, H, H, Cout, device=dev)
    xb, dyb = x.to(torch.bfloat16), dy.to(torch.bfloat16)
    sc, sh, pd = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev), torch.zeros(Cin, device=dev)
    dwp = torch.zeros(Cout, 9 * cg, device=dev)
    row = [name]
    for xf in (False, True):
        kw = dict(in_scale=sc, in_shift=sh, in_pad=pd) if xf else {}
        d16, _, _ = ops.make_conv_desc(xb, None, None, B=B, H=H, W=H, in_stride=Cin, cin_g=cg, Cout=Cout, groups=4, k=3, pad=1, **kw)
        d32, _, _ = ops.make_conv_desc(x, None, None, B=B, H=H, W=H, in_stride=Cin, cin_g=cg, Cout=Cout, groups=4, k=3, pad=1, **kw)
        for fn, d, g in ((_lib.lib.gssd_conv2d_wgrad_bf16, d16, dyb), (_lib.lib.gssd_conv2d_wgrad_f32, d32, dy)):
            for _ in range(3):
                _lib.check(fn(C.byref(d), g.data_ptr(), dwp.data_ptr(), st))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _lib.check(fn(C.byref(d), g.data_ptr(), dwp.data_ptr(), st))
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * B * H * H * Cout * 9 * cg
    print('%-8s bf16 %7.1f us (%6.1f TFLOP/s)  fp32 %7.1f us | deferred BN: bf16 %7.1f us  fp32 %7.1f us' %
          (row[0], row[1], fl / row[1] / 1e6, row[2], row[3], row[4]))
