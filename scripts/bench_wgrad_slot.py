"""Time gssd_conv2d_wgrad_f32 on plain 1x1 shapes (GSSD_NO_WGRAD_SLOT=1 for the generic kernel)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import ctypes as C
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
shapes = [(46208, 384, 512), (46208, 512, 256), (46208, 512, 512), (46208, 256, 1024), (46208, 512, 9216), (11552, 768, 1024),
          (11552, 1024, 512), (11552, 1024, 1024), (11552, 1024, 2048)]
for R, Cout, K in shapes:
    x = torch.randn(R, K, device=dev)
    dy = torch.randn(R, Cout, device=dev)
    dw = torch.zeros(Cout, K, device=dev)
    d, _, _ = ops.make_conv_desc(x, None, None, B=1, H=R, W=1, in_stride=K, cin_g=K, Cout=Cout)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _lib.check(_lib.lib.gssd_conv2d_wgrad_f32(C.byref(d), dy.data_ptr(), dw.data_ptr(), st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'R {R} Cout {Cout} K {K}: {ms * 1e3:8.1f} us  {2.0 * R * Cout * K / ms / 1e9:6.1f} TF')
