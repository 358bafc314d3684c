"""Micro-benchmark of gssd_conv2d_nhwc_f32 on the GSSD layer shapes (B=32): us / TFLOP/s per layer.
usage: python scripts/bench_conv.py [name-substring ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops, _lib
B = int(os.environ.get('B', '32'))
# name, H, Cin, Cout, k, s, p, d, groups, heads_split
LAYERS = [
    ('conv1_1', 300, 16, 64, 3, 1, 1, 1, 4), ('conv1_2', 300, 64, 64, 3, 1, 1, 1, 4),
    ('conv2_1', 150, 64, 128, 3, 1, 1, 1, 4), ('conv2_2', 150, 128, 128, 3, 1, 1, 1, 4),
    ('conv3_1', 75, 128, 256, 3, 1, 1, 1, 4), ('conv3_2', 75, 256, 256, 3, 1, 1, 1, 4),
    ('conv4_1', 38, 256, 512, 3, 1, 1, 1, 4), ('conv4_2', 38, 512, 512, 3, 1, 1, 1, 4),
    ('conv5_1', 19, 512, 512, 3, 1, 1, 1, 4), ('conv6', 19, 512, 1024, 3, 1, 6, 6, 4),
    ('conv7', 19, 1024, 1024, 1, 1, 0, 1, 4), ('fuse_11', 38, 512, 512, 1, 1, 0, 1, 1),
    ('fuse_21', 19, 1024, 1024, 1, 1, 0, 1, 1), ('ext0', 19, 1024, 256, 1, 1, 0, 1, 4),
    ('ext1', 19, 256, 512, 3, 2, 1, 1, 4), ('head0', 38, 512, 24, 3, 1, 1, 1, 1),
    ('head1', 19, 1024, 36, 3, 1, 1, 1, 1), ('head2', 10, 512, 36, 3, 1, 1, 1, 1),
    ('sa_tp', 38, 512, 128, 1, 1, 0, 1, 1), ('dcn_om', 38, 1024, 108, 3, 1, 1, 1, 1),
    ('dcn_main', 38, 9216, 512, 1, 1, 0, 1, 1),
]
sel = sys.argv[1:]
dev = torch.device('cuda:0')
for (name, H, Cin, Cout, k, s, p, d, g) in LAYERS:
    if sel and not any(x in name for x in sel):
        continue
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    wp = ops.pack_weight(w)
    Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
    out = torch.empty(B, Ho, Ho, Cout, device=dev)
    stats = torch.zeros(2 * Cout, device=dev, dtype=torch.float64)
    desc, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // g, Cout=Cout, groups=g, k=k,
                                    stride=s, pad=p, dil=d, bias=bias, stats=None if name.startswith('head') else stats,
                                    split_k=ops.auto_split_k(B * Ho * Ho, Cout, g, k * k * (Cin // g)) if name.startswith('head') else 1)
    for _ in range(3):
        ops.run_conv(desc)
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.run_conv(desc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    cin_real = 3 if name == 'conv1_1' else Cin // g
    fl = 2.0 * B * Ho * Ho * Cout * k * k * cin_real
    print(f'{name:9s} {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  ideal {fl / 157.3e6:7.1f} us  eff {fl / us / 1e6 / 157.3:5.1%}')
