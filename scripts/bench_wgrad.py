import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops
B = 32
LAYERS = [('conv1_1', 300, 16, 64, 3, 1, 1, 1, 4), ('conv1_2', 300, 64, 64, 3, 1, 1, 1, 4), ('conv2_1', 150, 64, 128, 3, 1, 1, 1, 4),
          ('conv2_2', 150, 128, 128, 3, 1, 1, 1, 4), ('conv3_1', 75, 128, 256, 3, 1, 1, 1, 4), ('conv3_2', 75, 256, 256, 3, 1, 1, 1, 4), ('conv4_2', 38, 512, 512, 3, 1, 1, 1, 4),
          ('conv5_1', 19, 512, 512, 3, 1, 1, 1, 4), ('conv6', 19, 512, 1024, 3, 1, 6, 6, 4), ('fuse_11', 38, 512, 512, 1, 1, 0, 1, 1),
          ('head1', 19, 1024, 36, 3, 1, 1, 1, 1)]
dev = torch.device('cuda:0')
for (name, H, Cin, Cout, k, s, p, d, g) in LAYERS:
    if sys.argv[1:] and not any(a in name for a in sys.argv[1:]): continue
    x = torch.randn(B, H, H, Cin, device=dev)
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    dy = torch.randn(B, Ho, Ho, Cout, device=dev)
    desc, _, _ = ops.make_conv_desc(x, None, None, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // g, Cout=Cout, groups=g, k=k, stride=s, pad=p, dil=d)
    for _ in range(2): ops.conv_wgrad(desc, dy, Cout, Cin // g, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10; e0.record()
    for _ in range(n): ops.conv_wgrad(desc, dy, Cout, Cin // g, k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * B * Ho * Ho * Cout * k * k * (Cin // g)
    print(f'{name:9s} wgrad {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s')
