"""A/B of two builds of libgssd_hip.so on the same box, alternating (the clock of a box drifts by several per cent between runs and
boxes): python3 scripts/ab_lib.py libgssd_hip_old.so libgssd_hip_new.so [rounds].  Each round starts one worker process per library
(the worker loads its library through GSSD_LIB_PATH) that times the fp32 Winograd trunk
shapes with HIP events; the table shows the mean over rounds."""
import sys, os, subprocess, json, shutil
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
LIBD = os.path.join(ROOT, 'grouped-ssd-pytorch_amd', 'gssd', 'lib')
SHAPES = ((300, 64, 64, True), (150, 128, 128, True), (75, 128, 256, False), (75, 256, 256, True), (38, 256, 512, False), (38, 512, 512, True), (19, 512, 512, True))


def worker(libname):
    os.environ['GSSD_LIB_PATH'] = os.path.join(LIBD, libname)      # gssd/_lib.py loads this build; lib/libgssd_hip.so is never touched
    sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
    import ctypes as C
    import torch
    from gssd import ops, _lib
    dev = torch.device('cuda:0')
    lib = _lib.lib
    B, res = 32, []
    for (H, Cin, Cout, xf) in SHAPES:
        # the map with its padding vector directly behind it (the layout gssd/engine.py builds for deferred BatchNorms)
        xfull = torch.randn(B * H * H * Cin + Cin, device=dev)
        xfull[B * H * H * Cin:] = 0
        x = xfull[:B * H * H * Cin].view(B, H, H, Cin)
        w = torch.randn(Cout, Cin // 4, 3, 3, device=dev) * 0.1
        wp = ops.pack_weight(w)
        U = ops.winograd_weight(wp, 4, Cin // 4)
        b = torch.zeros(Cout, device=dev)
        out = torch.empty(B, H, H, Cout, device=dev)
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
        sc, sh, pdv = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), xfull[B * H * H * Cin:]
        d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // 4, Cout=Cout, groups=4, k=3, stride=1, pad=1,
                                     dil=1, bias=b, stats=stats, in_scale=sc if xf else None, in_shift=sh if xf else None,
                                     in_pad=pdv if xf else None, wgt_wino=U)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(10):
            _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 100
        for _ in range(n):
            _lib.check(lib.gssd_conv2d_nhwc_f32(C.byref(d), st))
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    print('AB_RESULT ' + json.dumps(res))


if __name__ == '__main__':
    if sys.argv[1] == '--worker':
        worker(sys.argv[2])
        sys.exit(0)
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    acc = {l: [] for l in libs}
    for _ in range(rounds):
        for l in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', l], capture_output=True, text=True).stdout
            line = [ln for ln in out.splitlines() if ln.startswith('AB_RESULT ')]
            acc[l].append(json.loads(line[0][10:]))
    print('shape (H, Cin, Cout, fused input BN)      ' + '   '.join(f'{l:>22s}' for l in libs))
    for i, sh in enumerate(SHAPES):
        print(f'{str(sh):40s}  ' + '   '.join(f'{sum(r[i] for r in acc[l]) / rounds:19.1f} us' for l in libs))
    print('sum                                       ' + '   '.join(f'{sum(sum(r) for r in acc[l]) / rounds:19.1f} us' for l in libs))
