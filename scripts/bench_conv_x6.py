"""csrc/conv_x6.hip (fp32 conv, fp32-equivalent products on the bf16 matrix cores) against the kernels gssd_conv2d_nhwc_f32 otherwise runs
(Winograd F(2x2,3x3) / fp32-MFMA implicit GEMM) on the GSSD++ trunk shapes at B = 32: time per launch and the error of both against a
float64 convolution of one image."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import torch
import torch.nn.functional as F
from gssd import ops, _lib
F16OK = 0 if os.environ.get('F16OK') == '0' else _lib.CONV_F16_OK      # as a train-mode forward flags its launches (F16OK=0: bf16 planes, an eval-mode forward / a data gradient)
dev = torch.device('cuda:0')
torch.manual_seed(0)
B = 32
#         name      H   Cin  Cout groups k pad dil
SHAPES = [('conv3_1', 75, 128, 256, 4, 3, 1, 1), ('conv3_2', 75, 256, 256, 4, 3, 1, 1), ('conv4_1', 38, 256, 512, 4, 3, 1, 1),
          ('conv4_2', 38, 512, 512, 4, 3, 1, 1), ('conv5_x', 19, 512, 512, 4, 3, 1, 1), ('dcn.om', 38, 512, 216, 1, 3, 1, 1),
          ('conv6', 19, 512, 1024, 4, 3, 6, 6), ('conv7', 19, 1024, 1024, 4, 1, 0, 1), ('1x1 512>512', 38, 512, 512, 1, 1, 0, 1)]
only = sys.argv[1:] or None
for name, H, Cin, Cout, g, k, pad, dil in SHAPES:
    if only and name not in only:
        continue
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin // g, k, k, device=dev) * (2.0 / (k * k * Cin // g)) ** 0.5
    b = torch.randn(Cout, device=dev)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
    pdv = -sh / sc - 1.0                  # a value the transform maps to 0
    ref = F.conv2d(F.relu(x[:1].double() * sc.double() + sh.double()).permute(0, 3, 1, 2), w.double(), b.double(), 1, pad, dil, g).permute(0, 2, 3, 1)
    res = {}
    wino = ops.winograd_eligible(k, 1, pad, dil, Cin // g, Cout // g, g)
    for tag, kw in (('default' + ('(wino)' if wino else ''), dict(winograd=wino)), ('x6', dict(x6=True))):
        stats = torch.zeros(2 * Cout, device=dev, dtype=torch.float64)
        keep = []
        d = ops.conv2d_nhwc(x, w, b, 1, pad, dil, g, stats=stats, in_scale=sc, in_shift=sh, in_pad=pdv, _keep=keep, flags=F16OK, **kw)
        out = keep[-1]
        for _ in range(3): ops.run_conv(d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.run_conv(d)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        err = float((out[:1].double() - ref).abs().max() / ref.abs().max())
        res[tag] = (ms, err)
    fl = 2.0 * B * H * H * Cout * k * k * Cin / g
    print(f'{name:12s} tile {ops.x6_tile(Cout // g, g, B * H * H):3d} ' + '  '.join(f'{t}: {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF err {err:.2e}' for t, (ms, err) in res.items()), flush=True)
