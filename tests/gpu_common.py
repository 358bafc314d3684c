"""Shared pieces of the GPU parity test modules (tests/test_gpu_*.py): path setup, the tolerance of north_star, fixtures, layout helpers,
the net configurations and the canonical comparison of Detect outputs."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import gssd_oracle as O          # noqa: E402
from gssd import synth                       # noqa: E402

TOL = 1e-4


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from gssd import ops as _ops
    return _ops


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------------------
# end to end
# --------------------------------------------------------------------------------------------------
NETS = {
    'gssd': (dict(), (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)),
    'gssd_sa': (dict(use_self_attention=True, use_self_attention_base=True),
                (True, 4, 4, 1, True, True, True, 0, 1, False, False, 1)),
    'gssdpp': (dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                    dcn_cat_sab=True), (True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)),
}


def same_detections(det, ref, atol):
    """Rows agree as a set: scores saturate so exact fp32 ties exist, and the reference's visiting order among
    ties is an accident of torch's unstable sort.  Every reference row must have its own partner within atol."""
    if np.allclose(det, ref, rtol=0, atol=atol):
        return True
    for b in range(ref.shape[0]):
        for c in range(ref.shape[1]):
            d, r = det[b, c], ref[b, c]
            if (d[:, 0] > 0).sum() != (r[:, 0] > 0).sum():
                return False
            used = np.zeros(d.shape[0], bool)
            for row in r[r[:, 0] > 0]:
                err = np.abs(d - row).max(1)
                err[used] = np.inf
                j = int(err.argmin())
                if err[j] > atol:
                    return False
                used[j] = True
    return True


def _stage_errors(net, taps):
    """Cumulative HIP-vs-oracle error at every stage both sides expose (plan records vs the oracle's taps): where it jumps is the
    layer that eats the 1e-4 budget."""
    plan = net._engine._last_plan
    out = []
    for kind, r in plan.rec:
        if kind == 'convbn' and r['name'].startswith('vgg') and r.get('xf') is None:
            i = int(r['name'].split('.')[1])
            key = f'vgg.{i + 3}' if r['pool'] else f'vgg.{i + 2}'
            if key in taps and tuple(taps[key].shape) == tuple(nchw(r['out']).shape):
                out.append((r['name'] + ' (act)', rel(nchw(r['out']), taps[key])))
        elif kind == 'dcn' and 'dcn0.out' in taps:
            out.append(('dcn0', rel(nchw(r['out']), taps['dcn0.out'])))
        elif kind == 'l2norm' and 'l2norm' in taps:
            out.append(('l2norm', rel(nchw(r['out']), taps['l2norm'])))
    for i, (t, H, Cc) in enumerate(plan.sources):
        if f'source{i}' in taps:
            out.append((f'source{i}', rel(nchw(t), taps[f'source{i}'])))
    return out


FLOAT64_ARBITER_ALLOWED = {'fs2pp'}      # test_constructor_flags: the only config whose B = 2 graph may need the float64 arbiter
FLAG_NETS = {       # tests/golden/make_golden_flags.py: (oracle flags, build_ssd positional args, parameters whose gradients are compared)
    'nofuse': (dict(use_fuseconv=False, use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                    dcn_cat_sab=True), (True, 4, 4, 1, False, True, True, 1, 4, True, False, 1),
               ['vgg.0.weight', 'vgg.31.weight', 'extras.2.weight', 'loc.0.weight', 'conf.3.bias', 'L2Norm.weight',
                'self_attn_list.1.snconv1x1_g.weight_orig', 'dcn_list.0.weight']),
    'nobn': (dict(batch_norm=False, use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                  dcn_cat_sab=True), (False, 4, 4, 1, True, True, True, 1, 4, True, False, 1),
             ['vgg.0.weight', 'vgg.0.bias', 'vgg.21.weight', 'vgg.28.bias', 'vgg.33.weight', 'extras.3.weight', 'fuse_11.weight',
              'fuse_41.bias', 'loc.0.weight', 'conf.3.bias', 'L2Norm.weight', 'self_attn_base_list.0.snconv1x1_attn.weight_orig',
              'dcn_list.0.conv_offset_mask.weight']),
    'nobn_plain': (dict(batch_norm=False, use_fuseconv=False), (False, 4, 4, 1, False, False, False, 0, 1, False, False, 1),
                   ['vgg.0.weight', 'vgg.10.bias', 'vgg.31.weight', 'extras.0.weight', 'extras.7.bias', 'loc.1.weight',
                    'conf.5.weight', 'L2Norm.weight']),
    # (a single pooled key -- maps of 3 x 3 and below at factor 2 -- makes attn_g the same vector at every token, so g's BIAS only
    # adds a constant in front of the fuse BatchNorm: its gradient is mathematically zero there; g's weight is not)
    'mpf2': (dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True,
                  max_pool_factor=2), (True, 4, 4, 1, True, True, True, 1, 4, True, False, 2),
             ['vgg.0.weight', 'vgg.40.weight', 'fuse_21.weight', 'loc.0.weight', 'self_attn_list.0.snconv1x1_phi.weight_orig',
              'self_attn_base_list.1.snconv1x1_g.weight_orig', 'self_attn_list.3.snconv1x1_theta.weight_orig',
              'self_attn_list.4.snconv1x1_g.weight_orig', 'self_attn_base_list.2.snconv1x1_g.bias', 'dcn_list.0.weight']),
    'mpf3_sa': (dict(use_self_attention=True, use_self_attention_base=True, max_pool_factor=3),
                (True, 4, 4, 1, True, True, True, 0, 1, False, False, 3),
                ['vgg.0.weight', 'fuse_11.weight', 'loc.2.weight', 'self_attn_list.0.snconv1x1_phi.weight_orig',
                 'self_attn_base_list.2.snconv1x1_g.weight_orig', 'self_attn_list.2.snconv1x1_phi.weight_orig',
                 'self_attn_list.3.snconv1x1_g.weight_orig', 'self_attn_base_list.1.snconv1x1_g.bias']),
    'fs2': (dict(feature_scale=2), (True, 4, 4, 2, True, False, False, 0, 1, False, False, 1),
            ['vgg.0.weight', 'vgg.24.weight', 'vgg.44.weight', 'extras.4.weight', 'fuse_31.weight', 'loc.0.weight', 'conf.4.bias']),
    # round 4: --feature_scale 2 together with Self_Attn / DCN (VERDICT r3 "missing" 5): a (256, 1024) attention block on the 2048-channel
    # map (two launches of 512 g channels), 3072-float spectral-norm vectors, a 2048-channel deformable conv
    'fs2pp': (dict(feature_scale=2, use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True),
              (True, 4, 4, 2, True, True, True, 1, 4, True, False, 1),
              ['vgg.0.weight', 'vgg.24.weight', 'vgg.44.weight', 'extras.4.weight', 'fuse_31.weight', 'loc.0.weight', 'dcn_list.0.weight',
               'self_attn_list.1.snconv1x1_g.weight_orig', 'self_attn_base_list.0.snconv1x1_theta.weight_orig']),
    # two DCN layers (1024 -> 512, 512 -> 512), one deformable group, detach_sab (no gradient flows back into the SAB's attn_g copy)
    'dcn2_detach': (dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=2, groups_dcn=1, dcn_cat_sab=True),
                    (True, 4, 4, 1, True, True, True, 2, 1, True, True, 1),
                    ['vgg.0.weight', 'vgg.30.weight', 'dcn_list.0.weight', 'dcn_list.1.weight', 'dcn_list.1.conv_offset_mask.weight',
                     'dcn_list.0.conv_offset_mask.bias', 'self_attn_base_list.0.snconv1x1_g.weight_orig',
                     'self_attn_base_list.0.snconv1x1_attn.weight_orig', 'loc.0.weight']),
    'dcn_nocat': (dict(num_dcn_layers=1, groups_dcn=4), (True, 4, 4, 1, True, False, False, 1, 4, False, False, 1),
                  ['vgg.0.weight', 'vgg.30.weight', 'dcn_list.0.weight', 'dcn_list.0.bias', 'dcn_list.0.conv_offset_mask.weight',
                   'fuse_11.weight', 'loc.0.weight']),
    # round 3: --groups_vgg / --groups_extra 1 and 2 (train_lesion_multiphase_v2.py:47-48): the generic kernels, same plan code
    'g1': (dict(groups_vgg=1, groups_extra=1), (True, 1, 1, 1, True, False, False, 0, 1, False, False, 1),
           ['vgg.0.weight', 'vgg.14.weight', 'vgg.31.weight', 'vgg.44.weight', 'extras.2.weight', 'extras.8.weight', 'fuse_21.weight',
            'loc.0.weight', 'conf.3.bias']),
    'g2pp': (dict(groups_vgg=2, groups_extra=2, use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                  dcn_cat_sab=True), (True, 2, 2, 1, True, True, True, 1, 4, True, False, 1),
             ['vgg.0.weight', 'vgg.24.weight', 'vgg.40.weight', 'extras.4.weight', 'fuse_11.weight', 'loc.0.weight', 'dcn_list.0.weight',
              'dcn_list.0.conv_offset_mask.weight', 'self_attn_base_list.0.snconv1x1_g.weight_orig',
              'self_attn_list.1.snconv1x1_theta.weight_orig']),
    'g4e1': (dict(groups_extra=1), (True, 4, 1, 1, True, False, False, 0, 1, False, False, 1),
             ['vgg.0.weight', 'vgg.44.weight', 'extras.0.weight', 'extras.6.weight', 'fuse_41.weight', 'loc.4.weight', 'conf.4.bias']),
}
