"""Per-phase shader-clock breakdown of conv_flat_bf16 (debug build: conv_flat_bf16.o compiled with -DFLAT_TIMING and linked with the
other objects into lib/libgssd_hip_ft.so by `make -C grouped-ssd-pytorch_amd/gssd/csrc flat_timing`; this script loads it through GSSD_LIB_PATH).  Wave 0 of every workgroup accumulates the cycles between its phase
boundaries."""
import sys, os, shutil, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
LIBD = os.path.join(ROOT, 'grouped-ssd-pytorch_amd', 'gssd', 'lib')
os.environ['GSSD_LIB_PATH'] = os.path.join(LIBD, 'libgssd_hip_ft.so')      # gssd/_lib.py loads this build; lib/libgssd_hip.so is never touched
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.lib
rd = C.CDLL(_lib.LIB_PATH).gssd_flat_timing_read
B = 32
names = ['issue window + ring', 'wait window + barrier', 'transform + barrier', '9 taps (MFMA)', 'epilogue stores issued', 'stats + drain']
for (H, Cin, Cout, xf, st_) in ((75, 128, 256, False, True), (75, 256, 256, True, True), (38, 256, 512, False, True), (38, 512, 512, True, True),
                                (38, 512, 512, True, False), (19, 512, 512, True, True)):
    for bm in (128, 256):
        x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
        w = torch.randn(Cout, Cin // 4, 3, 3, device=dev) * 0.1
        wp = ops.pack_weight_bf16(w)
        b = torch.zeros(Cout, device=dev)
        out = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
        sc, sh = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
        pdv = torch.zeros(Cin, device=dev, dtype=torch.bfloat16)
        d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // 4, Cout=Cout, groups=4, k=3, stride=1, pad=1,
                                     dil=1, bias=b, stats=stats if st_ else None, in_scale=sc if xf else None, in_shift=sh if xf else None,
                                     in_pad=pdv if xf else None)
        lib.gssd_conv_flat_bf16_tile(bm)
        took = lib.gssd_conv_flat_bf16_takes(C.byref(d))
        if took != bm:
            continue
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
        tot = sum(t[:6])
        print(f'H {H} Cin {Cin} Cout {Cout} xf {xf} stats {st_} bm {bm}: {e0.elapsed_time(e1) / n * 1e3:.1f} us/launch, {t[7] / n:.0f} workgroups, '
              f'{tot / t[7]:.0f} cycles per workgroup')
        for k in range(6):
            print(f'    {names[k]:24s} {100.0 * t[k] / tot:5.1f} %   {t[k] / t[7]:10.0f} cycles per workgroup')
