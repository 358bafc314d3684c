"""Kernel-level parity of csrc/conv_wino_x6.hip: the three-plane Winograd kernel takes launches by
default only at >= 8 192 Winograd tiles, whole 64-channel blocks and >= 32 channels per group -- shapes no other kernel-level test reaches.
GSSD_WINO_X6=2 makes it take every shape it can; the switch is read once per process, hence a worker subprocess per mode
(tests/wino_x6_worker.py).  Every launch form (plain + batch sums, replicas, fused producer BatchNorm + ReLU with both padding paths,
pooled-raw epilogue, data gradient with and without an existing gradient) is held against a float64 convolution AND against the fp32-MFMA
Winograd kernel (GSSD_WINO_X6=0) on the same inputs: the three-plane products are fp32-equivalent, so the kernel has to be as close to
float64 as the fp32 kernel is (gate: e <= 1.5 e_fp32 + 1e-7, the rule of test_conv_x6_matches_float64).  A second test runs the DEFAULT
host rule in this process at shapes above its size gate (conv3_1 at B = 6, conv4_2 at B = 23) and a third the random-shape sweep of
scripts/fuzz_x6.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu
_cache = {}


def run_worker(mode, f16ok=False):
    key = (mode, f16ok)
    if key not in _cache:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'wino_x6_worker.py')], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, GSSD_WINO_X6=mode, WX6_TEST_F16OK='1' if f16ok else '0'))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('WINOX6JSON ')][-1]
        _cache[key] = json.loads(line[len('WINOX6JSON '):])
        assert _cache[key]['mode'] == mode
    return _cache[key]


def x6_can_take(case):
    """csrc/conv_wino_x6.hip::gssd_wino_x6_plane_elems: cin_g a multiple of 16 and at least 24 output channels per group."""
    B, H, W, Cin, Cout, g = case
    return (Cin // g) % 16 == 0 and Cout // g >= 24


@pytest.mark.parametrize('f16ok', [False, True], ids=['bf16_planes', 'fp16_planes'])
def test_conv_winograd_x6_every_form_vs_float64_and_fp32_kernel(f16ok):
    """f16ok: every forward launch carries GSSD_CONV_F16_OK -- the kernel's two-plane fp16 form (round 6: what a train-mode forward runs); the data
    gradients keep the three bf16 planes either way."""
    x6, f32 = run_worker('2', f16ok), run_worker('0')
    assert len(x6['results']) == len(f32['results']) >= 11
    lines = []
    for a, b in zip(x6['results'], f32['results']):
        case = a['case']
        assert case == b['case']
        B, H, W, Cin, Cout, g = case
        assert set(a['forms']) == set(b['forms']) and {'plain', 'plain_rep8', 'xf_select', 'xf_address', 'pool', 'pool_xf'} <= set(a['forms'])
        for name, fa in a['forms'].items():
            fb = b['forms'][name]
            dgrad = name.startswith('dgrad')
            takes_case = (B, H, W, Cout, Cin, g) if dgrad else tuple(case)
            assert fb['takes'] == 0, (case, name)                                       # GSSD_WINO_X6=0: never
            assert fa['takes'] == (1 if x6_can_take(takes_case) else 0), (case, name)   # GSSD_WINO_X6=2: every shape it can
            lines.append(f'{case} {name}: x6 {fa["err"]:.2e} fp32 {fb["err"]:.2e} takes {fa["takes"]}')
            assert fa['err'] < 2e-5 and fb['err'] < 2e-5, lines[-1]                     # Winograd F(2x2,3x3) vs direct float64 (test_conv_winograd's gate)
            assert fa['err'] <= 1.5 * fb['err'] + 1e-7, lines[-1]
            d = np.abs(np.asarray(fa['sample']) - np.asarray(fb['sample'])).max() / fa['ref_max']
            assert d < 4e-6, (lines[-1], d)                                             # same transforms: the products' last bits only
            if 'stat_err' in fa:
                assert max(fa['stat_err']) < 2e-7 and max(fb['stat_err']) < 2e-7, (lines[-1], fa['stat_err'])
            if 'pooled_image_of_plain' in fa:
                assert fa['pooled_image_of_plain'] and fb['pooled_image_of_plain'], lines[-1]     # bit for bit the pooled image of the plain launch
            if 'equal_to_select' in fa:
                assert fa['equal_to_select'] and fb['equal_to_select'], lines[-1]                # the two padding paths: same bits
    print('\n'.join(lines))
    taken = sum(f['takes'] for r in x6['results'] for f in r['forms'].values())
    assert taken >= 70, taken
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'wino_x6_forms_fp16.txt' if f16ok else 'wino_x6_forms.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')


@pytest.mark.parametrize('case', [(6, 75, 128, 256, 4), (6, 75, 256, 256, 4), (23, 38, 512, 512, 4), (24, 38, 1024, 108, 1)])
def test_conv_winograd_x6_default_host_rule(case):
    """The default mode in this process (no switch): shapes above the kernel's size gate -- what a batch-32 step launches -- are taken
    by conv_wino_x6 and agree with float64 (forward with fused producer transform + batch sums; the data gradient into an existing
    gradient, which at batch 32 is how conv3_1 .. conv4_3 get their d(x))."""
    import ctypes as C
    import torch.nn.functional as F
    from gssd import ops, _lib
    assert os.environ.get('GSSD_WINO_X6') is None
    B, H, Cin, Cout, g = case
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(sum(case))
    cin_g, cout_g = Cin // g, Cout // g
    x = torch.from_numpy(rng.normal(0.1, 1.0, size=(B, Cin, H, H)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.05, size=(Cout, cin_g, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    scv = torch.from_numpy(rng.uniform(0.5, 1.5, size=Cin).astype(np.float32))
    shv = torch.from_numpy(rng.normal(0, 0.3, size=Cin).astype(np.float32))
    act = torch.relu(torch.addcmul(shv.double().view(1, -1, 1, 1), x.double(), scv.double().view(1, -1, 1, 1)))
    ref = F.conv2d(act, w.double(), b.double(), 1, 1, 1, g).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = ops.pack_weight(w.to(dev))
    U = ops.winograd_weight(wp, g, cin_g)
    y = torch.full((B, H, H, Cout), float('nan'), device=dev)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    sc, sh = scv.to(dev), shv.to(dev)
    d, _, _ = ops.make_conv_desc(xd, wp, y, B=B, H=H, W=H, in_stride=Cin, cin_g=cin_g, Cout=Cout, groups=g, k=3, pad=1, bias=b.to(dev), wgt_wino=U,
                                 stats=stats, in_scale=sc, in_shift=sh, in_pad=(-sh / sc - 1.0), flags=_lib.CONV_F16_OK)      # (as a train-mode forward flags it)
    assert _lib.lib.gssd_conv_wino_x6_takes(C.byref(d)) == 1
    ops.run_conv(d)
    e = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    n_px = B * H * H
    s_err = float((stats[:Cout].cpu() - ref.sum((0, 1, 2))).abs().max() / (n_px * float(ref.abs().max())))
    print(f'{case}: conv_wino_x6 (default rule) vs float64 {e:.2e}, batch sums {s_err:.1e}')
    assert e < 4e-6 and s_err < 2e-7
    if ops.winograd_eligible(3, 1, 1, 1, cout_g, cin_g, g):
        dy = torch.from_numpy(rng.normal(size=(B, Cout, H, H)).astype(np.float32))
        existing = torch.from_numpy(rng.normal(size=(B, H, H, Cin)).astype(np.float32))
        dref = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1, 0, g).permute(0, 2, 3, 1) + existing.double()
        wd = ops.pack_weight_dgrad(w.to(dev), g)
        ud = ops.winograd_weight(wd, g, cout_g)
        dx = torch.full((B, H, H, Cin), float('nan'), device=dev)
        dd, _, _ = ops.make_conv_desc(dy.permute(0, 2, 3, 1).contiguous().to(dev), wd, dx, B=B, H=H, W=H, in_stride=Cout, cin_g=cout_g, Cout=Cin,
                                      groups=g, k=3, pad=1, resid=existing.to(dev), wgt_wino=ud)
        # the transposed shape under the same host rule (csrc/conv_wino_x6.hip::gssd_wino_x6_wanted): conv3_1's data gradient (256 -> 128 channels:
        # 32-channel blocks) stays on conv_wino.hip, conv4_2's (512 -> 512) is a conv_wino_x6 launch
        want = 1 if (cin_g % 64 == 0 and cout_g >= 32) else 0
        assert _lib.lib.gssd_conv_wino_x6_takes(C.byref(dd)) == want
        ops.run_conv(dd)
        e = float((dx.cpu().double() - dref).abs().max() / dref.abs().max())
        print(f'{case}: data gradient + existing ({"conv_wino_x6" if want else "conv_wino"}) vs float64 {e:.2e}')
        assert e < 4e-6


def test_conv_winograd_x6_random_shapes():
    """scripts/fuzz_x6.py wino: conv_wino_x6 (GSSD_WINO_X6=2) against the implicit GEMM on random shapes -- every tile tail, 1 .. 4 chunks,
    16-channel groups, padded output blocks, non-square maps, with / without the fused producer transform and the residual."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'fuzz_x6.py'), '16', '23', 'wino'], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GSSD_WINO_X6='2'))
    assert r.returncode == 0 and 'conv_wino_x6: 16 shapes' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
