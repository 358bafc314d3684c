"""Direct implicit GEMM vs Winograd F(2x2,3x3) on the trunk's 3x3 layer shapes (B = 32, groups 4), with a parity check."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops, _lib
FLAGS = _lib.CONV_F16_OK if os.environ.get('F16OK') else 0      # bounded activations: the two-plane fp16 form of csrc/conv_wino_x6.hip
dev = torch.device('cuda:0')
B, G = int(os.environ.get('B', 32)), int(os.environ.get('G', 4))
shapes = [('conv1_2', 300, 16, 16), ('conv2_1', 150, 16, 32), ('conv2_2', 150, 32, 32), ('conv3_1', 75, 32, 64), ('conv3_2', 75, 64, 64),
          ('conv4_1', 38, 64, 128), ('conv4_2', 38, 128, 128), ('conv5_x', 19, 128, 128)]
if os.environ.get('SHAPES'):
    shapes = [tuple([t.split(':')[0]] + [int(v) for v in t.split(':')[1:]]) for t in os.environ['SHAPES'].split(',')]
only = os.environ.get('ONLY')
for name, H, cin_g, cout_g in shapes:
    if only and name != only:
        continue
    torch.manual_seed(0)
    x = torch.randn(B, H, H, G * cin_g, device=dev)
    w = torch.randn(G * cout_g, cin_g, 3, 3, device=dev) * (2.0 / (9 * cin_g)) ** 0.5
    bias = torch.randn(G * cout_g, device=dev)
    flops = 2.0 * B * H * H * G * cout_g * cin_g * 9
    res = {}
    for wino in (False, True):
        st = torch.zeros(2 * G * cout_g, dtype=torch.float64, device=dev)
        y = ops.conv2d_nhwc(x, w, bias, pad=1, groups=G, stats=st, winograd=wino)
        wp = ops.pack_weight(w)
        U = ops.winograd_weight(wp, G, cin_g) if wino else None
        out = torch.empty_like(y)
        d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=G * cin_g, cin_g=cin_g, Cout=G * cout_g, groups=G, k=3,
                                     pad=1, bias=bias, wgt_wino=U, flags=FLAGS)
        for _ in range(3): ops.run_conv(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.run_conv(d)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res[wino] = (y, st, ms)
    (yd, sd, md), (yw, sw, mw) = res[False], res[True]
    err = float((yd - yw).abs().max() / yd.abs().max())
    serr = float((sd - sw).abs().max() / sd.abs().max())
    print(f'{name}: direct {md:.3f} ms ({flops / md / 1e9:.1f} TF)  winograd {mw:.3f} ms ({flops / mw / 1e9:.1f} TF-equiv)  '
          f'rel err {err:.2e} stats {serr:.2e}', flush=True)
