"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/gssd_hip.h
declares (no compute calls: there is no GPU here), and the host logic (PriorBox, module tree / state-dict
keys, pooling arithmetic, target packing) matches the reference-generated fixtures."""
import copy
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_match_header():
    from gssd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'gssd_hip.h')).read()
    declared = set(re.findall(r'\b(gssd_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'gssd_conv_desc', 'gssd_sn_item', 'gssd_stream_t', 'gssd_plan_op'}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert _lib.lib.gssd_abi_version() == 8 and _lib.lib.gssd_build_arch() == b'gfx950'
    # struct layout agrees with the header's field order / C packing rules
    assert ctypes.sizeof(_lib.ConvDesc) == 17 * 8 + 27 * 4 + 4 + 6 * 8 == _lib.lib.gssd_conv_desc_size()   # (17 pointers since wgt_patch; 27 ints + 4 bytes of padding)
    assert ctypes.sizeof(_lib.SnItem) == 4 * 8 + 2 * 4


def test_plan_runner_table_and_encoding():
    """csrc/plan_run.hip: the runner's function table is exactly the header's `int gssd_*(..., gssd_stream_t stream)`
    entry points, each with the parameter count of the binding; argument words are encoded as the header says; bad ops are refused with
    their index before anything touches a device."""
    import struct
    from gssd import _lib, planrun
    lib = _lib.lib
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'gssd_hip.h')).read(), flags=re.S)
    want = {}
    for name, args in re.findall(r'\bint\s+(gssd_\w+)\s*\(([^;{]*?)\)\s*;', hdr, flags=re.S):
        a = [x.strip() for x in args.replace('\n', ' ').split(',')]
        if re.search(r'gssd_stream_t\s+\w+$', a[-1]):
            want[name] = len(a) - 1
    have = {lib.gssd_plan_fn_name(i).decode(): lib.gssd_plan_fn_nargs(i) for i in range(lib.gssd_plan_fn_count())}
    assert have == want and len(have) >= 85
    for name, n in have.items():
        assert len(_lib.SIGNATURES[name][1]) == n + 1 and planrun.fn_info(getattr(lib, name))[0] == lib.gssd_plan_fn_index(name.encode())
    assert lib.gssd_plan_fn_index(b'gssd_abi_version') == -1 and lib.gssd_plan_fn_nargs(10 ** 6) == -1
    assert max(have.values()) <= 24 and ctypes.sizeof(_lib.PlanOp) == 16 + 24 * 8 == lib.gssd_plan_op_size()
    # words: integers sign-extended, float / double as their bits, pointers by value, byref(struct) -> its address
    assert planrun.word(_lib.c_i, -1) == 2 ** 64 - 1 and planrun.word(_lib.c_i64, 1 << 40) == 1 << 40
    assert planrun.word(_lib.c_f, 0.5) == struct.unpack('<I', struct.pack('<f', 0.5))[0]
    assert planrun.word(_lib.c_d, -2.25) == struct.unpack('<Q', struct.pack('<d', -2.25))[0]
    d = _lib.ConvDesc()
    assert planrun.word(ctypes.POINTER(_lib.ConvDesc), ctypes.byref(d)) == ctypes.addressof(d) and planrun.word(_lib.c_fp, None) == 0
    # a program: a launch that the entry point itself refuses (null descriptor fields) -> its code and its index come back
    prog = planrun.Program([('wait', planrun.MAIN, planrun.MAIN), ('launch', lib.gssd_conv2d_nhwc_f32, (ctypes.byref(d),), planrun.MAIN)])
    with pytest.raises(_lib.GssdError, match=r'plan op 1 \(gssd_conv2d_nhwc_f32\)'):
        prog.run(0)
    ops_ = (_lib.PlanOp * 1)()
    ops_[0].kind, ops_[0].fn, ops_[0].stream = _lib.PLAN_LAUNCH, 10 ** 6, 0
    failed = ctypes.c_int(-7)
    streams = (_lib.c_fp * 1)()
    assert lib.gssd_plan_run(ops_, 1, streams, 1, ctypes.byref(failed)) == -1 and failed.value == 0
    ops_[0].fn, ops_[0].nargs = lib.gssd_plan_fn_index(b'gssd_conv2d_nhwc_f32'), 3                       # wrong argument count
    assert lib.gssd_plan_run(ops_, 1, streams, 1, ctypes.byref(failed)) == -1 and b'argument count' in lib.gssd_last_error()
    ops_[0].stream = 5
    assert lib.gssd_plan_run(ops_, 1, streams, 1, ctypes.byref(failed)) == -1 and b'stream index' in lib.gssd_last_error()


def test_argument_validation_without_gpu():
    """Entry points reject bad descriptors before touching the device."""
    from gssd import _lib
    d = _lib.ConvDesc()
    assert _lib.lib.gssd_conv2d_nhwc_f32(ctypes.byref(d), None) == -1
    assert b'invalid argument' in _lib.lib.gssd_last_error()
    assert _lib.lib.gssd_detect(None, None, None, 1, 8732, 2, 200, 0.01, 0.45, 0.1, 0.2, 0, 0, None, None, None, None) == -1


def test_conv_x6_host_rules():
    """Host side of csrc/conv_x6.hip (no device work): the tile rule, the size of the three-plane weight form, and which descriptors
    gssd_conv2d_nhwc_f32 hands to the kernel (gssd_conv_x6_takes)."""
    from gssd import _lib, ops
    lib = _lib.lib
    assert [lib.gssd_conv_x6_tile(c, 4, 46208) for c in (32, 64, 96, 128, 216, 256, 1024)] == [64, 64, 128, 128, 128, 128, 128]
    # (3 bf16 + 3 fp16 planes, round 6) x groups x ceil(cout_g / tile) x tile x taps x cin_g 16-bit elements; -1 for shapes the kernel does not take
    assert lib.gssd_conv_x6_weight_elems(1024, 4, 128, 9, 128) == 6 * 4 * 2 * 128 * 9 * 128
    assert lib.gssd_conv_x6_weight_elems(216, 1, 512, 9, 128) == 6 * 1 * 2 * 128 * 9 * 512          # 216 -> two 128-column tiles, zero rows
    assert lib.gssd_conv_x6_weight_elems(512, 1, 48, 1, 128) == -1                                    # cin_g % 32 != 0
    assert lib.gssd_conv_x6_weight_elems(512, 1, 64, 1, 96) == -1                                     # not a tile width
    assert ops.x6_wanted(1, 512, 512, 1, 46208) and ops.x6_wanted(3, 128, 256, 4, 11552)
    assert not ops.x6_wanted(3, 128, 128, 4, 46208, winograd=True) and not ops.x6_wanted(1, 512, 64, 1, 46208) and not ops.x6_wanted(1, 512, 512, 1, 3200)
    # Winograd shapes: only forward launches on maps too small for conv_wino_x6 (conv5_x at batch 32), and only on the fp16 planes
    assert ops.x6_wanted(3, 128, 128, 4, 11552, winograd=True, forward=True) == ops.X6_F16
    assert not ops.x6_wanted(3, 128, 128, 4, 11552, winograd=True) and not ops.x6_wanted(3, 128, 128, 4, 46208, winograd=True, forward=True)
    buf = torch.zeros(64, dtype=torch.float32)             # any 16-byte aligned host address: takes() only inspects the descriptor

    def desc(**kw):
        args = dict(B=2, H=19, W=19, in_stride=256, cin_g=256, Cout=512)
        args.update(kw)
        d, _, _ = ops.make_conv_desc(buf, buf, buf, wgt_x6=args.pop('x6', buf), **args)
        return lib.gssd_conv_x6_takes(ctypes.byref(d))
    assert desc() == 1
    assert desc(x6=None) == 0                               # no three-plane weights: the fp32-MFMA kernels
    assert desc(alpha=buf, gate=buf, resid=buf, out2=buf, relu=True, stats=torch.zeros(8, dtype=torch.float64)) == 1
    assert desc(out2=buf) == 0                              # a second output only exists behind a gate
    assert desc(split_k=2) == 0 and desc(out_mode=_lib.OUT_TRANSPOSED, m_per_image=True, out_stride=364) == 0
    assert desc(cin_g=48, in_stride=48) == 0 and desc(Cout=24) == 0
    # the merged Self_Attn projection: split-transposed store, whole tiles on either side of split_n, flat or contiguous per-image batches
    sp = dict(Cout=384, out_mode=_lib.OUT_SPLIT_T, out_b=buf, split_n=128, out_stride=128, out_b_stride=364, flags=_lib.CONV_OUT_F32,
              in_batch_stride=361 * 256, out_batch_stride=361 * 128, outb_batch_stride=256 * 364)
    assert desc(**sp) == 1 and desc(m_per_image=True, **sp) == 1
    assert desc(**dict(sp, split_n=64, out_stride=64, out_batch_stride=361 * 64)) == 0
    assert desc(m_per_image=True, **dict(sp, in_batch_stride=400 * 256)) == 0


def test_priorbox_bit_exact(golden):
    from layers.functions import PriorBox
    from data import v2, v2_512
    g = golden('priors')
    p = PriorBox(v2).forward()
    assert p.dtype == torch.float32 and np.array_equal(p.numpy(), g['v2'])
    p5 = PriorBox(v2_512).forward().numpy()
    assert p5.shape == (24564, 4) and np.array_equal(p5[g['v2_512_sample_idx']], g['v2_512_sample'])


@pytest.mark.parametrize('name,args', [
    ('gssd', (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)),
    ('gssd_sa', (True, 4, 4, 1, True, True, True, 0, 1, False, False, 1)),
    ('gssdpp', (True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)),
])
def test_module_tree_matches_reference(golden, name, args):
    from models.ssd_multiphase_custom_group import build_ssd, weights_init
    g = golden('e2e')
    net = build_ssd('train', 300, 2, *args)
    sd = net.state_dict()
    assert sorted(sd.keys()) == [str(k) for k in g[f'{name}.keys']]
    assert [str(tuple(sd[k].shape)) for k in sorted(sd.keys())] == [str(s) for s in g[f'{name}.shapes']]
    # the surface the driver touches (train_lesion_multiphase_v2.py:587-615)
    net.extras.apply(weights_init)
    net.loc.apply(weights_init)
    net.conf.apply(weights_init)
    twin = copy.deepcopy(net)
    assert twin._engine is not net._engine and twin._engine.net is twin
    test_net = build_ssd('test', 300, 2, *args)
    test_net.load_state_dict(net.state_dict())                 # strict, train -> test phase
    if name == 'gssdpp':
        assert any(n.startswith('dcn_list') for n, _ in net.named_parameters())
    assert net.priors.shape == (8732, 4)


def test_vanilla_module_tree(golden):
    from models.ssd import build_ssd
    g = golden('e2e')
    net = build_ssd('train', 300, 2)
    sd = net.state_dict()
    assert sorted(sd.keys()) == [str(k) for k in g['ssd.keys']]
    assert [str(tuple(sd[k].shape)) for k in sorted(sd.keys())] == [str(s) for s in g['ssd.shapes']]
    assert build_ssd('train', 512, 2) is None and build_ssd('x', 300, 2) is None


def test_build_ssd_rejects_like_reference(capsys):
    from models.ssd_multiphase_custom_group import build_ssd
    assert build_ssd('val', 300, 2, True) is None
    assert build_ssd('train', 512, 2, True) is None
    assert 'only SSD300' in capsys.readouterr().out
    with pytest.raises(AssertionError):
        build_ssd('train', 300, 2, True, 4, 4, 1, True, False, False, 1, 4, True, False, 1)   # cat_sab needs sab


def test_pool_and_pack_helpers():
    from gssd import ops
    for n, k, s, p, ceil in [(300, 2, 2, 0, False), (75, 2, 2, 0, True), (19, 3, 1, 1, False), (38, 2, 2, 0, False),
                             (5, 2, 2, 0, True), (7, 3, 2, 1, True)]:
        ref = torch.nn.functional.max_pool2d(torch.zeros(1, 1, n, n), k, s, p, ceil_mode=ceil).shape[-1]
        assert ops.pool_out_size(n, k, s, p, ceil) == ref
    tg = [torch.rand(2, 5), torch.rand(0, 5), torch.rand(3, 5)]
    flat, off = ops.pack_targets(tg, 'cpu')
    assert off.tolist() == [0, 2, 2, 5] and torch.equal(flat[2:], tg[2])
    assert ops.packed_k(3, 3, 3) == (4, 36)


def test_synth_is_deterministic():
    from gssd import synth
    a, b = synth.synth_images(2, seed=4), synth.synth_images(2, seed=4)
    assert torch.equal(a, b) and float(a.min()) == 0.0 and float(a.max()) == 1.0
    t = synth.synth_targets(4, seed=1)
    assert all(x.shape[1] == 5 and 1 <= x.shape[0] <= 3 for x in t)


def test_resample_coefficients_match_pillow_restatement():
    """Host half of the input stage (no GPU): the C library's fixed-point tables equal the oracle's (pinned on Pillow)."""
    from gssd import input_stage as IS
    from oracle import input_oracle as IO
    for a, b in ((512, 300), (64, 37), (20, 33), (300, 300), (513, 127)):
        for f in ('bicubic', 'bilinear'):
            bo, ko, ks = IO.resample_coeffs(a, b, f)
            bh, kh = IS.resample_tables(a, b, f)
            assert kh.shape[1] == ks and np.array_equal(bo, bh) and np.array_equal(ko, kh), (a, b, f)
    with pytest.raises(Exception):
        IS.resample_tables(0, 300)
    with pytest.raises(Exception):
        IS.DeviceInputStage(300)(torch.zeros(1, 4, 8, 8, 3, dtype=torch.uint8))      # CPU tensor: no fallback


def _run_bench(*args, env=None):
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **(env or {})))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r.returncode, [json.loads(ln) for ln in lines]


def test_bench_self_launch_refuses_missing_gpus_in_one_line():
    """`python bench.py --gpus N` on a host with fewer GPUs: one JSON error line, non-zero exit, nothing spawned."""
    ndev = torch.cuda.device_count()
    rc, lines = _run_bench('--gpus', str(ndev + 2))
    assert rc == 2 and len(lines) == 1 and lines[0]['devices_visible'] == ndev and 'error' in lines[0]


def test_bench_self_launch_two_ranks_rendezvous():
    """The self-launcher (no torch.distributed.run): two child ranks find each other on 127.0.0.1, rank 0 alone prints the line,
    the parent returns their code; a rank that dies ends the job with ITS code instead of hanging the others."""
    rc, lines = _run_bench('--gpus', '2', '--launch-probe')
    assert rc == 0 and len(lines) == 1
    assert lines[0]['ranks'] == [0.0, 1.0] and lines[0]['rccl_ranks'] == 2 and lines[0]['launcher'] == 'self'
    rc, lines = _run_bench('--gpus', '2', '--launch-probe', env={'GSSD_PROBE_FAIL_RANK': '1'})
    assert rc == 7 and lines == []


def test_bench_stdout_line_is_short_and_parses(tmp_path):
    """the driver keeps a bounded tail of stdout -- the ONE JSON line must stay under 8 KB (r04's 21.6 KB line left
    BENCH_r04.parsed = null).  Replays a full record of a real run (the committed round-4 detail, per-rank lists widened to 8 ranks)
    through bench.py's own line builder and through the CLI, and checks the contract's fields + roofline + cpu_baseline survive."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r04_h_gssdpp_b32_bench.json')))
    assert len(json.dumps(full)) > 2 * bench.LINE_LIMIT                      # the record itself is the long form
    for world in (1, 8):
        rec = json.loads(json.dumps(full))
        rec['n_gpus'] = rec['rccl_ranks'] = world
        rec['per_rank_ms_per_step'] = rec['per_rank_ms_per_step'] * world
        rec['host_enqueue_ms_per_step'] = rec['host_enqueue_ms_per_step'] * world
        rec['full_step']['host_enqueue_ms_per_step'] = rec['full_step']['host_enqueue_ms_per_step'] * world
        line = bench.compact_line(rec, 'gpurun_out/bench_detail.json')
        txt = json.dumps(line)
        assert len(txt) < bench.LINE_LIMIT == 8192, len(txt)
        back = json.loads(txt)
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                  'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
            assert k in back, k
        assert 'workload' in back['config'] and 'model' not in back['config']
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in back['roofline'], k
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in back['cpu_baseline'], k
        assert 'kernels' not in back and 'trunk' not in back
    p = tmp_path / 'detail.json'
    p.write_text(json.dumps(full))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--replay-detail', str(p)], capture_output=True, text=True, timeout=300)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and len(r.stdout.strip().splitlines()) == 1 and len(last) < 8192
    assert json.loads(last)['value'] == full['value']


def test_fp16_planes_are_flagged_only_for_train_mode_fp32_plans():
    """gssd/plan_ops.py::PlanOpsMixin.f16_ok: GSSD_CONV_F16_OK goes on forward launches of a train-mode, fp32-mode plan only (batch-statistics
    BatchNorm bounds the activations); eval mode, the bf16 storage mode and PixelLink's plan (a trunk without BatchNorm) never flag."""
    from gssd import _lib
    from gssd.plan_ops import PlanOpsMixin
    from gssd.pixellink import _PlanPixelLink

    import types

    class P(PlanOpsMixin):
        def __init__(self, training, bf16, batch_norm=True):
            self.training, self.bf16 = training, bf16
            self.eng = types.SimpleNamespace(net=types.SimpleNamespace(batch_norm=batch_norm))
    assert P(True, False).f16_ok == _lib.CONV_F16_OK == 32
    assert P(False, False).f16_ok == 0 and P(True, True).f16_ok == 0 and P(False, True).f16_ok == 0
    assert P(True, False, batch_norm=False).f16_ok == 0
    assert _PlanPixelLink.f16_ok == 0
    # the header documents the promise the flag makes
    hdr = open(os.path.join(ROOT, 'include', 'gssd_hip.h')).read()
    assert '#define GSSD_CONV_F16_OK 32' in hdr and 'NEVER inferred' in hdr and 'gssd_dcn_forward_x6_ex' in hdr
