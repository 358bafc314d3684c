"""Per-phase shader-clock breakdown of conv_wino (debug build: conv_wino.o compiled with -DWINO_TIMING and linked with the other
objects into lib/libgssd_hip_wt.so by `make -C grouped-ssd-pytorch_amd/gssd/csrc wino_timing`; this script loads it through GSSD_LIB_PATH).  Wave 0 of every workgroup accumulates the cycles between its phase
boundaries (the debug build also drains vmcnt before the transform to separate waiting from arithmetic)."""
import sys, os, shutil, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
LIBD = os.path.join(ROOT, 'grouped-ssd-pytorch_amd', 'gssd', 'lib')
os.environ['GSSD_LIB_PATH'] = os.path.join(LIBD, 'libgssd_hip_wt.so')      # gssd/_lib.py loads this build; lib/libgssd_hip.so is never touched
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.lib
rd = C.CDLL(_lib.LIB_PATH).gssd_wino_timing_read
B = 32
names = ['loop overhead / decode', 'wait patch + U (vmcnt 0)', 'permute + input transform', 'barrier', '256 MFMAs + next loads issued',
         'output transform + stores', 'stats + drain']
for (H, Cin, Cout, xf, st_) in ((150, 128, 128, True, True), (75, 128, 256, False, True), (75, 256, 256, True, True), (38, 256, 512, False, True),
                                (38, 512, 512, True, True), (38, 512, 512, False, False), (19, 512, 512, True, True)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin // 4, 3, 3, device=dev) * 0.1
    wp = ops.pack_weight(w)
    U = ops.winograd_weight(wp, 4, Cin // 4)
    b = torch.zeros(Cout, device=dev)
    out = torch.empty(B, H, H, Cout, device=dev)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    sc, sh, pdv = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), torch.zeros(Cin, device=dev)
    d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // 4, Cout=Cout, groups=4, k=3, stride=1, pad=1,
                                 dil=1, bias=b, stats=stats if st_ else None, in_scale=sc if xf else None, in_shift=sh if xf else None,
                                 in_pad=pdv if xf else None, wgt_wino=U)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    rd(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
    e1.record()
    torch.cuda.synchronize()
    rd(buf)
    t = list(buf)
    tot = sum(t[:7])
    us = e0.elapsed_time(e1) / n * 1e3
    gf = 2.0 * B * H * H * Cout * (Cin // 4) * 9 / 1e9
    print(f'H {H} Cin {Cin} Cout {Cout} xf {xf} stats {st_}: {us:.1f} us/launch ({gf / us * 1e3:.1f} TFLOP/s direct-conv), {t[7] / n:.0f} workgroups, '
          f'{tot / max(t[7], 1):.0f} cycles per workgroup')
    for k in range(7):
        print(f'    {names[k]:32s} {100.0 * t[k] / max(tot, 1):5.1f} %   {t[k] / max(t[7], 1):10.0f} cycles per workgroup')
