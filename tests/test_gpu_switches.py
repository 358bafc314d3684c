"""Every GSSD_* kernel-selection / scheduling switch the product code still reads (profiles/README.md lists what each one is for) is an
untested combination unless something runs it: one GSSD++ training step at batch 4 per switch, in a subprocess (the switches are read
once per process), against the default configuration of the same process family.  The alternative paths compute the same function
with a different kernel or schedule: fp32 results agree to 1e-4 of the tensor's scale (gradients 2e-2 relative L2: ReLU / max-pool
decisions flip between fp32 summation orders, tests/test_gpu_training.py::test_backward_gradients), bf16 results within the bf16
contract of tests/test_gpu_bf16.py (a few per cent after the trunk, loss 1e-2)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
_cache = {}


def run(dtype, env, batch=4):
    key = (dtype, batch, tuple(sorted(env.items())))
    if key not in _cache:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'switch_worker.py'), dtype, str(batch)], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('SWITCHJSON ')][-1]
        _cache[key] = json.loads(line[len('SWITCHJSON '):])
    return _cache[key]


def l2rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def has(o, prefix):
    return any(k.startswith(prefix) for k in o['kernels'])


SWITCHES = [
    # dtype, environment (+ optional 'BATCH': a batch at which the switch really changes the kernel set), what must (not) be in the launched kernel set
    ('f32', {'GSSD_NO_GRAPH': '1'}, lambda o, base: o['graphs'] == 0 and base['graphs'] > 0),
    ('f32', {'GSSD_NO_BRANCH_STREAMS': '1'}, lambda o, base: True),
    ('f32', {'GSSD_BWD_STREAMS': '0'}, lambda o, base: True),
    ('f32', {'GSSD_BWD_GRAPH': '1'}, lambda o, base: o['bwd_graphs'] > 0 and base['bwd_graphs'] == 0),
    ('f32', {'GSSD_NO_WINOGRAD': '1'}, lambda o, base: not any(k.startswith(('conv_wino', 'conv_thin_wino')) for k in o['kernels'])
     and any(k.startswith('conv_wino') for k in base['kernels'])),
    ('f32', {'GSSD_NO_GEMM_SLOT': '1'}, lambda o, base: True),
    ('f32', {'GSSD_NO_SMALL_TILES': '1'}, lambda o, base: not any(k.startswith(('conv_igemm<32x', 'conv_igemm<64x')) for k in o['kernels'])
     and any(k.startswith(('conv_igemm<32x', 'conv_igemm<64x')) for k in base['kernels'])),
    ('bf16', {'GSSD_NO_SMALL_TILES': '1'}, lambda o, base: not any(k.startswith(('conv_bf16<32x', 'conv_bf16<64x')) for k in o['kernels'])
     and any(k.startswith(('conv_bf16<32x', 'conv_bf16<64x')) for k in base['kernels'])),
    ('f32', {'GSSD_NO_WGRAD_SLOT': '1'}, lambda o, base: True),
    ('f32', {'GSSD_GEMM_SLOT_SWAP': '0'}, lambda o, base: True),
    # (at this test's batch of 4 the deformable conv has 92 tiles for 256 CUs and keeps the one-tile form either way: the stream-K form
    # itself is tested at batch 32 by test_gpu_kernels.py::test_dcn_fused_streamk)
    ('f32', {'GSSD_DCN_STREAMK': '0'}, lambda o, base: True),
    # the deformable conv on the fp32 matrix cores (csrc/dcn_fused.hip) instead of the three-plane bf16 form (csrc/dcn_x6.hip)
    ('f32', {'GSSD_DCN_X6': '0'}, lambda o, base: any(k.startswith('dcn_fused') for k in o['kernels']) and any(k.startswith('dcn_x6') for k in base['kernels'])),
    # one workgroup per tile of the three-plane deformable conv (at batch 4 the default splits its K loop in two: csrc/dcn_x6.hip)
    ('f32', {'GSSD_DCN_X6_SPLITK': '0'}, lambda o, base: has(o, 'dcn_x6')),
    # the fp32-MFMA kernels (conv_igemm / gemm_slot) instead of the three-plane bf16 conv (csrc/conv_x6.hip) on the launches it takes
    ('f32', {'GSSD_CONV_X6': '0'}, lambda o, base: not any(k.startswith('conv_x6') for k in o['kernels']) and any(k.startswith('conv_x6') for k in base['kernels'])),
    ('bf16', {'GSSD_NO_CONV_FLAT': '1'}, lambda o, base: not any(k.startswith('conv_flat_bf16') for k in o['kernels'])
     and any(k.startswith('conv_flat_bf16') for k in base['kernels'])),
    ('bf16', {'GSSD_FLAT_BM': '128'}, lambda o, base: all(k.endswith(',128>') for k in o['kernels'] if k.startswith('conv_flat_bf16'))),
    ('bf16', {'GSSD_FLAT_BM': '256'}, lambda o, base: any(k.endswith(',256>') for k in o['kernels'] if k.startswith('conv_flat_bf16'))),
    ('bf16', {'GSSD_FLAT_PERSIST': '0'}, lambda o, base: True),
    # flash-style attention backward with its logits recomputed on the fp32 matrix cores instead of the three-product bf16 split
    ('bf16', {'GSSD_FLASH_BWD_X3': '0'}, lambda o, base: 'gssd_self_attn_flash_bwd_bf16' in o['bwd_fns']),
    # the round-3 form of the bf16 training step: the fp32 backward plan on fp32 copies of every stored map
    ('bf16', {'GSSD_BWD_BF16': '0'}, lambda o, base: 'gssd_conv2d_wgrad_bf16' in base['bwd_fns'] and 'gssd_bn_bwd_apply_mixed' in base['bwd_fns']
     and not any(f in o['bwd_fns'] for f in ('gssd_conv2d_wgrad_bf16', 'gssd_bn_bwd_apply_mixed', 'gssd_dcn_im2col_bf16'))),
    # ---- round 6: the size-gated three-plane kernels, at batch 24 where conv3_1 .. conv4_3 (+ the offset conv) and their data gradients run
    # csrc/conv_wino_x6.hip (>= 8 192 Winograd tiles) and conv6 / conv7 / the 19 x 19 dgrads run csrc/conv_x6.hip (M >= 4 096): a whole
    # training step of the batch-32 kernel mix against the same step on the fp32-MFMA kernels
    ('f32', {'GSSD_WINO_X6': '0', 'BATCH': 24}, lambda o, base: not has(o, 'conv_wino_x6') and has(base, 'conv_wino_x6<64>') and has(o, 'conv_wino<64>')),
    # (every shape it can take: conv5_x on the 19 x 19 maps -- 2 400 tiles -- move over from conv_wino.hip<64>; with the fp16 planes off, or the
    # train-mode forward runs them on conv_x6 since round 6)
    ('f32', {'GSSD_WINO_X6': '2', 'GSSD_X6_F16': '0', 'BATCH': 24}, lambda o, base: not has(o, 'conv_wino<64>') and has(o, 'conv_wino_x6<64>')),
    ('f32', {'GSSD_CONV_X6': '0', 'BATCH': 24}, lambda o, base: not has(o, 'conv_x6') and has(base, 'conv_x6')),
    ('f32', {'GSSD_DCN_X6': '0', 'BATCH': 24}, lambda o, base: has(o, 'dcn_fused') and has(base, 'dcn_x6')),
    ('f32', {'GSSD_WINO_X6': '0', 'GSSD_CONV_X6': '0', 'GSSD_DCN_X6': '0', 'GSSD_FLASH_X6': '0', 'BATCH': 24},
     lambda o, base: not any(has(o, k) for k in ('conv_wino_x6', 'conv_x6', 'dcn_x6', 'flash_attn_x6'))),
    # round 6: conv1_2 / conv2_1 / conv2_2 on the round-5 fp32-MFMA kernels instead of the patch-staged three-plane direct conv (csrc/conv_thin_x6.hip)
    ('f32', {'GSSD_THIN_X6': '0'}, lambda o, base: not has(o, 'conv_thin_x6') and has(base, 'conv_thin_x6<16,16>') and has(base, 'conv_thin_x6<32,32>')
     and has(o, 'conv_thin_wino')),
    # static tile shares instead of claimed tiles in the persistent thin trunk kernels of the fp32 mode (csrc/conv_thin_x6.hip)
    ('f32', {'GSSD_TX6_DYNAMIC': '0'}, lambda o, base: has(o, 'conv_thin_x6')),
    # the 38 x 38 multibox head on the implicit GEMM with reduction slices instead of conv_wino_x6's heads epilogue (>= 8 192 Winograd tiles: batch 24)
    ('f32', {'GSSD_HEADS_WINO': '0', 'BATCH': 24}, lambda o, base: has(o, 'conv_igemm<128x32>') and has(base, 'conv_wino_x6<32>')),
    # conv3_1 on the Winograd kernels instead of conv_thin_x6<32,64>
    ('f32', {'GSSD_THIN_X6_CONV31': '0'}, lambda o, base: has(base, 'conv_thin_x6<32,64>') and not has(o, 'conv_thin_x6<32,64>')),
    # the DCN offset conv on conv_wino_x6 / conv_wino instead of the patch-staged direct conv on fp16 planes (csrc/conv_patch_x6.hip)
    ('f32', {'GSSD_PATCH_X6': '0'}, lambda o, base: not has(o, 'conv_patch_x6') and has(base, 'conv_patch_x6')),
    # the x6 kernels' forward launches on three bf16 planes / six MFMAs (round 5's form) instead of the fp16 planes / three MFMAs; and the attention
    # core's opt-in fp16 form
    ('f32', {'GSSD_X6_F16': '0', 'BATCH': 24}, lambda o, base: has(o, 'conv_x6') and has(o, 'dcn_x6') and has(o, 'conv_wino_x6')),
    ('f32', {'GSSD_FLASH_X6_F16': '1'}, lambda o, base: has(o, 'flash_attn_x6')),
    # the attention cores of the 38 x 38 blocks on the fp32 matrix cores (csrc/flash_attn.hip) instead of the three-plane form
    ('f32', {'GSSD_FLASH_X6': '0'}, lambda o, base: not has(o, 'flash_attn_x6') and has(base, 'flash_attn_x6') and has(o, 'flash_attn<')),
    # the backward as one Python call per launch instead of one gssd_plan_run array per gradient segment (csrc/plan_run.hip)
    ('f32', {'GSSD_NO_PLAN_RUN': '1'}, lambda o, base: True),
    ('f32', {'GSSD_BRANCH0_LATE': '0'}, lambda o, base: True),
    # (conv2_1 belongs to conv_thin_x6 since round 6: the round-5 choice between its two fp32 kernels only exists with that one off)
    ('f32', {'GSSD_CONV21_WINO': '0', 'GSSD_THIN_X6': '0'}, lambda o, base: has(o, 'conv_thin<16,32>') and not has(base, 'conv_thin<16,32>')),
    ('bf16', {'GSSD_STATS_REP': '0'}, lambda o, base: True),
    # opt-in experiments that stay in the tree (DESIGN 9 "measured and rejected"): 256 x 128 bf16 tiles (>= 8 192 rows: batch 8), the loader /
    # matrix-wave form of the bf16 deformable conv
    ('bf16', {'GSSD_BF16_BIG_TILES': '1', 'BATCH': 8}, lambda o, base: has(o, 'conv_bf16<256x128>') and not has(base, 'conv_bf16<256x128>')),
    ('bf16', {'GSSD_DCN_BF16_V3': '1', 'BATCH': 8}, lambda o, base: True),
]


@pytest.mark.parametrize('dtype,env,check', SWITCHES, ids=[f"{d}-{'-'.join(f'{k}={v}' for k, v in e.items())}" for d, e, _ in SWITCHES])
def test_switch_path_agrees_with_default(dtype, env, check):
    env = dict(env)
    batch = env.pop('BATCH', 4)
    base = run(dtype, {}, batch)
    out = run(dtype, env, batch)
    assert check(out, base), (out['kernels'], out['graphs'])
    assert out['graph_replay'] <= (1e-6 if dtype == 'f32' else 0.0) * max(out['loc_max'], 1.0) + 1e-5     # eager == hipGraph replay
    f32 = dtype == 'f32'
    assert l2rel(out['loc'], base['loc']) < (1e-4 if f32 else 6e-2)
    for i in (0, 1):
        assert abs(out['loss'][i] - base['loss'][i]) <= (1e-4 if f32 else 1e-2) * abs(base['loss'][i])
    for k, v in out['gnorm'].items():
        tol = 2e-2 if f32 else 0.6        # bf16: two bf16 forwards' trunk gradients differ by tens of per cent (test_bf16_backward_gradients)
        assert abs(v - base['gnorm'][k]) <= tol * base['gnorm'][k], (k, v, base['gnorm'][k])
        if f32:
            assert l2rel(out['gsample'][k], base['gsample'][k]) < 5e-2, k
