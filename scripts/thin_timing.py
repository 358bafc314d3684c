"""Per-phase shader-clock breakdown of conv_thin_bf16 (debug build with -DTHIN_TIMING, a -DTHIN_TIMING build of conv_thin_bf16.o linked into a copy of the library, loaded through GSSD_LIB_PATH): wave 1 of every workgroup accumulates the cycles between the phase boundaries of its tiles."""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import shutil
LIBD = os.path.join(ROOT, 'grouped-ssd-pytorch_amd', 'gssd', 'lib')
os.environ['GSSD_LIB_PATH'] = os.path.join(LIBD, 'libgssd_hip_tt.so')      # gssd/_lib.py loads this build; lib/libgssd_hip.so is never touched
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.lib
rd = C.CDLL(_lib.LIB_PATH).gssd_thin_timing_read
B = 32
for (H, Cin, Cout, xf, pool) in ((300, 32, 64, False, False), (300, 64, 64, True, True), (300, 64, 64, True, False), (150, 64, 128, True, False), (150, 128, 128, True, True)):
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin // 4, 3, 3, device=dev) * 0.1
    wp = ops.pack_weight_bf16(w)
    b = torch.zeros(Cout, device=dev)
    Ho = H // 2 if pool else H
    out = torch.empty(B, Ho, Ho, Cout, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(32 * 2 * Cout, dtype=torch.float64, device=dev)
    sign = torch.randn(Cout, device=dev)
    sc, sh = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    pdv = torch.zeros(Cin, device=dev, dtype=torch.bfloat16)
    d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // 4, Cout=Cout, groups=4, k=3, stride=1, pad=1,
                                 dil=1, bias=b, stats=stats, in_scale=sc if xf else None, in_shift=sh if xf else None,
                                 in_pad=pdv if xf else None, stats_rep=32, flags=_lib.CONV_POOL2 if pool else 0, pool_sign=sign if pool else None)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(lib.gssd_conv2d_nhwc_bf16(C.byref(d), st))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    rd(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        _lib.check(lib.gssd_conv2d_nhwc_bf16(C.byref(d), st))
    e1.record()
    torch.cuda.synchronize()
    rd(buf)
    t = list(buf)
    wgs = t[7] / n
    tot = sum(t[:7])
    names = ['dma issue', 'wait dma + barrier', 'transform', 'barrier', 'epilogue (bias, sums, pooling, stores)', 'barrier', 'fragment reads + mfma']
    print(f'H {H} Cin {Cin} Cout {Cout} xf {xf} pool {pool}: {e0.elapsed_time(e1) / n * 1e3:.1f} us/launch, {wgs:.0f} workgroups, '
          f'{tot / t[7]:.0f} cycles per workgroup in the tile loop')
    for k in range(7):
        print(f'    {names[k]:40s} {100.0 * t[k] / tot:5.1f} %   {t[k] / t[7]:10.0f} cycles per workgroup')
