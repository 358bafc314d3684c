"""csrc/conv_patch_x6.hip against the kernels the DCN offset conv (1024 -> 108 channels, 38 x 38, B = 32) and the big multibox heads otherwise run:
python scripts/bench_patch_x6.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import ctypes as C
import torch
import torch.nn.functional as F
from gssd import ops, _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
B = 32
for name, H, Cin, Cout in (('dcn.om', 38, 1024, 108), ('head 38x38', 38, 512, 24), ('head 19x19', 19, 1024, 36)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
    b = torch.randn(Cout, device=dev)
    ref = F.conv2d(x[:1].double().permute(0, 3, 1, 2), w.double(), b.double(), 1, 1).permute(0, 2, 3, 1)
    line = f'{name:11s}'
    for tag, kw in (('patch_x6', dict(patch=True)), ('wino(_x6)', dict(winograd=True, flags=_lib.CONV_F16_OK | _lib.CONV_OUT_F32)), ('igemm', dict())):
        keep = []
        try:
            d = ops.conv2d_nhwc(x, w, b, 1, 1, 1, 1, _keep=keep, **kw)
        except _lib.GssdError as e:
            line += f'  {tag}: n/a'
            continue
        out = keep[0] if keep else None
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            _lib.check(_lib.lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _lib.check(_lib.lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
        e1.record()
        torch.cuda.synchronize()
        us = 100 * e0.elapsed_time(e1)
        line += f'  {tag}: {us:7.1f} us {2 * B * H * H * Cout * 9 * Cin / us / 1e6:6.1f} TF'
    print(line)
