"""Knock-out timing of the thin bf16 trunk kernels (csrc/conv_thin_bf16.hip built with -DTHIN_KO=k by `make thin_ko`): one child process
per library (GSSD_LIB_PATH), the four thin layers at B = 32 in their nograd-plan forms (conv1_2 / conv2_2 pooled).  Results of the
knock-out builds are wrong on purpose; only the times matter: which part of the tile loop costs what."""
import sys, os, subprocess, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
LIBD = os.path.join(ROOT, 'grouped-ssd-pytorch_amd', 'gssd', 'lib')
KO = {0: 'production', 1: 'no producer transform', 2: 'no fragment reads', 4: 'no MFMAs', 6: 'no fragment reads, no MFMAs', 7: 'no transform / reads / MFMAs',
      8: 'trivial epilogue', 79: '15 + linear tile order', 143: '15 + no swizzle', 256: 'no batch-sum atomics', 512: 'no weight prologue', 271: '15 + no batch-sum atomics', 15: 'only DMA + trivial epilogue + stores', 16: 'no stores', 24: 'trivial epilogue, no stores', 32: 'no patch DMA'}
LAYERS = (('conv1_1', 300, 32, 64, False, False), ('conv1_2/pool', 300, 64, 64, True, True), ('conv1_2', 300, 64, 64, True, False),
          ('conv2_1', 150, 64, 128, True, False), ('conv2_2/pool', 150, 128, 128, True, True), ('conv2_2', 150, 128, 128, True, False))


def worker():
    sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
    import ctypes as C
    import torch
    from gssd import ops, _lib
    lib = _lib.lib
    dev = torch.device('cuda:0')
    B = 32
    res = {}
    for (name, H, Cin, Cout, xf, pool) in LAYERS:
        x = (torch.ones if os.environ.get('KO_ONES') else torch.randn)(B, H, H, Cin, device=dev).to(torch.bfloat16)
        w = torch.randn(Cout, Cin // 4, 3, 3, device=dev) * 0.1
        wp = ops.pack_weight_bf16(w)
        b = torch.zeros(Cout, device=dev)
        Ho = H // 2 if pool else H
        out = torch.empty(B, Ho, Ho, Cout, device=dev, dtype=torch.bfloat16)
        R = int(os.environ.get('KO_REP', '32'))
        stats = torch.zeros(max(R, 1) * 2 * Cout, dtype=torch.float64, device=dev)
        sc, sh = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
        pdv = torch.zeros(Cin, device=dev, dtype=torch.bfloat16)
        sign = torch.randn(Cout, device=dev)
        d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // 4, Cout=Cout, groups=4, k=3, stride=1, pad=1,
                                     dil=1, bias=b, stats=stats, in_scale=sc if xf else None, in_shift=sh if xf else None,
                                     in_pad=pdv if xf else None, flags=_lib.CONV_POOL2 if pool else 0, pool_sign=sign if pool else None, stats_rep=R)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            _lib.check(lib.gssd_conv2d_nhwc_bf16(C.byref(d), st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n):
            _lib.check(lib.gssd_conv2d_nhwc_bf16(C.byref(d), st))
        e1.record()
        torch.cuda.synchronize()
        byts = 2.0 * B * (H * H * Cin + Ho * Ho * Cout)
        us = e0.elapsed_time(e1) / n * 1e3
        res[name] = (round(us, 1), round(byts / us / 1e3))
    print('RESULT ' + json.dumps(res))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'worker':
        worker()
        sys.exit(0)
    kos = [int(v) for v in sys.argv[1:]] or sorted(KO)
    print(f'{"build":44s}' + ''.join(f'{l[0]:>20s}' for l in LAYERS) + '    (us, GB/s of compulsory bytes)')
    for k in kos:
        env = dict(os.environ)
        if k:
            env['GSSD_LIB_PATH'] = os.path.join(LIBD, f'libgssd_hip_ko{k}.so')
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'worker'], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
        if not line:
            print(k, 'failed', r.stderr[-300:])
            continue
        res = json.loads(line[0][7:])
        print(f'{("KO " + str(k) + ": " + KO.get(k, "")):44s}' + ''.join(f'{res[l[0]][0]:12.1f} {res[l[0]][1]:6d} ' for l in LAYERS))
