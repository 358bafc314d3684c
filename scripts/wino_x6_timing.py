"""Per-phase shader-clock breakdown of conv_wino_x6 (debug build with -DWX6_TIMING: `bash scripts/wino_x6_timing.sh build`, loaded through
GSSD_LIB_PATH).  Wave 0 of every workgroup accumulates the cycles between its phase boundaries; per STEP (one 32-channel chunk of one item =
four blocks) averages are printed."""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
os.environ['GSSD_LIB_PATH'] = os.path.join(ROOT, 'build_ko', 'libgssd_wx6_timing.so')
os.environ['GSSD_WINO_X6'] = '1'
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.lib
rd = C.CDLL(_lib.LIB_PATH).gssd_wino_x6_timing_read
B = 32
names = ['wait vmcnt(0) at block top', 'barrier', 'U DMA + patch loads issued', 'block body: vector work + MFMAs + fold', 'epilogue', 'prologue']
for (name, H, cin_g, cout_g) in (('conv3_1', 75, 32, 64), ('conv3_2', 75, 64, 64), ('conv4_2', 38, 128, 128), ('conv5_x', 19, 128, 128)):
    Cin, Cout = 4 * cin_g, 4 * cout_g
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, cin_g, 3, 3, device=dev) * 0.1
    wp = ops.pack_weight(w)
    U = ops.winograd_weight(wp, 4, cin_g)
    out = torch.empty(B, H, H, Cout, device=dev)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    sc, sh = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    xpad = torch.cat([x.reshape(-1), torch.zeros(Cin, device=dev)])          # the padding vector directly behind the map (the engine's layout)
    xin = xpad[:x.numel()].view(B, H, H, Cin)
    d, _, _ = ops.make_conv_desc(xin, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=cin_g, Cout=Cout, groups=4, k=3, stride=1, pad=1, dil=1,
                                 bias=torch.zeros(Cout, device=dev), stats=stats, in_scale=sc, in_shift=sh, in_pad=xpad[x.numel():], wgt_wino=U)
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
    steps = max(t[6], 1)
    print(f'{name}: {e0.elapsed_time(e1) / n * 1e3:.1f} us/launch, {t[7] / n:.0f} workgroups, {t[6] / max(t[7], 1):.1f} steps per workgroup, '
          f'{sum(t[:5]) / steps:.0f} cycles per step')
    for k in range(6):
        print(f'    {names[k]:42s} {t[k] / steps:9.0f} cycles per step')
