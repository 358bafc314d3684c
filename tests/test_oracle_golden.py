"""Pin the CPU oracle against fixtures generated from the imported reference
(tests/golden/make_golden.py).  CPU-only; this is what makes oracle-vs-HIP parity meaningful."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from oracle import gssd_oracle as O
from gssd import synth


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_priors_bit_exact(golden):
    g = golden('priors')
    p = O.prior_box(O.V2)
    assert p.shape == (8732, 4) and p.dtype == np.float32
    assert np.array_equal(p, g['v2'])
    assert hashlib.sha256(p.tobytes()).hexdigest().startswith('a962243ddb360b31')
    p512 = O.prior_box(O.V2_512)
    assert p512.shape == (24564, 4)
    assert sha(p512) == g['v2_512_sha'].tobytes()
    assert np.array_equal(p512[g['v2_512_sample_idx']], g['v2_512_sample'])


def test_match_indices_bit_exact(golden):
    g = golden('match')
    pri = O.prior_box()
    for i in range(int(g['n'])):
        t = g[f't{i}']
        loc, conf, _ = O.match(0.5, t[:, :-1], pri, (0.1, 0.2), t[:, -1])
        assert np.array_equal(conf.astype(np.int8), g[f'conf{i}']), f'case {i}'
        pos = conf > 0
        ref = g[f'locpos{i}']
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(loc[pos]), fin)
        # log() may differ in the last ulp between torch (SLEEF) and numpy
        assert np.allclose(loc[pos][fin], ref[fin], rtol=2e-6, atol=2e-6), f'case {i}'
    # SURVEY.md 8(c) known answers
    loc, conf, _ = O.match(0.5, g['t0'][:, :-1], pri, (0.1, 0.2), g['t0'][:, -1])
    assert np.nonzero(conf)[0].tolist() == [6257, 6263, 6371, 6377, 6485, 6491, 8068, 8069, 8074, 8075, 8077,
                                             8134, 8137]
    assert np.allclose(loc[6257], [0.36760753, 2.2056446, 0.48936203, 1.927772], rtol=1e-6)


def test_multibox_loss(golden):
    g = golden('loss')
    m = golden('match')
    pri = O.prior_box()
    P = pri.shape[0]
    ll, lc = O.multibox_loss(np.zeros((2, P, 4), np.float32), np.zeros((2, P, 2), np.float32), pri,
                             [m['t0'], m['t1']])
    assert abs(ll - g['zero_loss'][0]) < 1e-6 and abs(lc - g['zero_loss'][1]) < 1e-6
    assert abs(lc - 4 * np.log(2)) < 1e-6
    for ci in range(3):
        rng = np.random.default_rng(int(g[f'seed{ci}']))
        loc = rng.normal(0, 1.0, size=(4, P, 4)).astype(np.float32)
        conf = rng.normal(0, 2.0, size=(4, P, 2)).astype(np.float32)
        tg = [g[f'tg{ci}_{b}'] for b in range(4)]
        ll, lc, d = O.multibox_loss(loc, conf, pri, tg, details=True)
        assert rel(ll, g[f'loss{ci}'][0]) < 1e-5 and rel(lc, g[f'loss{ci}'][1]) < 1e-5
        assert np.array_equal(np.packbits(d['pos']), g[f'pos{ci}'])
        neg_ref = np.unpackbits(g[f'neg{ci}'])[:4 * P].reshape(4, P).astype(bool)
        # mining ranks are integer work: identical to the reference except among scores within 2 ulp of the cut-off
        from helpers import assert_mined_negatives_contract
        assert_mined_negatives_contract(d['neg'], neg_ref, d['loss_c_all'], d['num_pos'])


def test_detect_and_nms(golden):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    from make_golden import closed_form_detect_inputs
    g = golden('detect')
    pri = O.prior_box()
    loc, logit = closed_form_detect_inputs(pri.shape[0], 2)
    # Detect's contract takes the already-softmaxed scores (detection_pytorch_ver_1point5.py:33): same inputs
    out, keeps = O.detect(2, 0, 200, 0.01, 0.45, loc, g['conf_sm'], pri, return_keep=True)
    assert np.array_equal(keeps[(0, 1)], g['keep0']) and np.array_equal(keeps[(1, 1)], g['keep1'])   # bit-exact
    assert keeps[(0, 1)].shape[0] == 190
    assert np.allclose(out, g['out'], rtol=0, atol=2e-6)
    assert np.all(out[:, 0] == 0)
    rng = np.random.default_rng(5)
    loc_r = rng.normal(0, 0.5, size=(2, pri.shape[0], 4)).astype(np.float32)
    rng.normal(0, 1.5, size=(2, pri.shape[0], 2))
    out_r = O.detect(2, 0, 200, 0.01, 0.45, loc_r, g['conf_sm_rand'], pri)
    assert np.abs(O.softmax_scores(logit) - g['conf_sm']).max() <= 1.2e-7      # K15: 1 ulp vs torch
    assert np.array_equal(out_r[..., 0] > 0, g['out_rand'][..., 0] > 0)
    assert np.allclose(out_r, g['out_rand'], rtol=0, atol=3e-6)
    with pytest.raises(ValueError):
        O.detect(2, 0, 200, 0.01, 0.0, loc, logit, pri)


def test_small_ops(golden):
    g = golden('ops')
    y = O.l2norm(torch.from_numpy(g['l2_x']), torch.from_numpy(g['l2_w'])).numpy()
    assert rel(y, g['l2_y']) < 1e-6
    from torch import nn
    shapes = {}
    C = 64
    for n, (co, ci) in dict(theta=(C // 8, C), phi=(C // 8, C), g=(C // 2, C), attn=(C, C // 2)).items():
        p = f'snconv1x1_{n}'
        shapes[p + '.bias'] = (co,)
        shapes[p + '.weight_orig'] = (co, ci, 1, 1)
        shapes[p + '.weight_u'] = (co,)
        shapes[p + '.weight_v'] = (ci,)
    shapes['sigma'] = (1,)
    for mode in ('eval', 'train'):
        sd = {'sa.' + k: v for k, v in synth.synth_state_dict(shapes, seed=21).items()}
        upd = {}
        out, ag, attn = O.self_attn(torch.from_numpy(g[f'sa_{mode}_x']), sd, 'sa', mode == 'train', 1, upd)
        assert rel(out.numpy(), g[f'sa_{mode}_out']) < 1e-5
        assert rel(ag.numpy(), g[f'sa_{mode}_ag']) < 1e-5
        assert rel(attn.numpy(), g[f'sa_{mode}_attn']) < 1e-5
        if mode == 'train':
            for k, v in upd.items():
                assert rel(v.numpy(), g['sa_train_after.' + k[3:]]) < 1e-5


def test_dcn_known_answers():
    """DCN parity is unpinned (no reference arithmetic in-tree); these are the identities the
    reference's own wrapper guarantees (dcn_v2_custom.py:75-77 zero-init)."""
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(size=(2, 16, 9, 9)).astype(np.float32))
    w = torch.from_numpy(rng.normal(size=(8, 16, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(8,)).astype(np.float32))
    dg = 4
    off = torch.zeros(2, dg * 18, 9, 9)
    mask = torch.full((2, dg * 9, 9, 9), 0.5)
    y = O.dcn_v2_conv(x, off, mask, w, b, 1, 1, 1, dg)
    ref = 0.5 * torch.nn.functional.conv2d(x, w, None, 1, 1) + b.view(1, -1, 1, 1)
    assert rel(y.numpy(), ref.numpy()) < 1e-5
    # integer offset (+1 row, -1 col on every tap) == conv over the shifted, zero-padded map
    off[:, 0::2] = 1.0
    off[:, 1::2] = -1.0
    y = O.dcn_v2_conv(x, off, torch.ones_like(mask), w, b, 1, 1, 1, dg)
    xs = torch.zeros_like(x)
    xs[:, :, :-1, 1:] = x[:, :, 1:, :-1]
    ref = torch.nn.functional.conv2d(xs, w, b, 1, 1)
    # border taps differ (the shifted map loses one row/col of real data): compare the interior
    assert rel(y[:, :, 2:-2, 2:-2].numpy(), ref[:, :, 2:-2, 2:-2].numpy()) < 1e-5


@pytest.mark.parametrize('name,flags', [
    ('gssd', dict()),
    ('gssd_sa', dict(use_self_attention=True, use_self_attention_base=True)),
    ('gssdpp', dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                    dcn_cat_sab=True)),
])
def test_end_to_end_vs_reference(golden, name, flags):
    g = golden('e2e')
    keys = [str(k) for k in g[f'{name}.keys']]
    shapes = {k: eval(s) for k, s in zip(keys, g[f'{name}.shapes'])}
    sd = synth.synth_state_dict(shapes, seed=1111)
    x = synth.synth_images(4, seed=5)
    torch.set_num_threads(8)
    with torch.no_grad():
        loc, conf, upd = O.gssd_forward(sd, x, **flags)
    assert loc.shape == (4, 8732, 4) and conf.shape == (4, 8732, 2)
    l, c = loc.numpy().reshape(-1), conf.numpy().reshape(-1)
    assert np.abs(l[g[f'{name}.loc_idx']] - g[f'{name}.loc_val']).max() / g[f'{name}.loc_absmax'] < 1e-5
    assert np.abs(c[g[f'{name}.conf_idx']] - g[f'{name}.conf_val']).max() / g[f'{name}.conf_absmax'] < 1e-5
    ll, lc = O.multibox_loss(loc.numpy(), conf.numpy(), O.prior_box(), [t.numpy() for t in synth.synth_targets(4, 5)])
    assert rel(ll, g[f'{name}.loss'][0]) < 1e-4 and rel(lc, g[f'{name}.loss'][1]) < 1e-4
    for k in ('vgg.1.running_mean', 'vgg.1.running_var', 'bn_fuse_11.running_mean', 'extras.15.running_var'):
        assert rel(upd[k].numpy(), g[f'{name}.after.{k}']) < 1e-5, k
    if name != 'gssd':
        k = 'self_attn_list.0.snconv1x1_theta.weight_u'
        assert rel(upd[k].numpy(), g[f'{name}.after.{k}']) < 1e-5
    # test phase twin: one more training forward with BN momentum 1.0 (running stats := batch stats, see
    # make_golden.py), then the eval-mode forward + Detect
    sd2 = dict(sd)
    sd2.update(upd)
    O.BN_MOMENTUM[0] = 1.0
    try:
        with torch.no_grad():
            _, _, upd2 = O.gssd_forward(sd2, x, **flags)
    finally:
        O.BN_MOMENTUM[0] = 0.1
    sd2.update(upd2)
    with torch.no_grad():
        loc_e, conf_e, _ = O.gssd_forward(sd2, x, training=False, **flags)
    det = O.detect(2, 0, 200, 0.01, 0.45, loc_e.numpy(), O.softmax_scores(conf_e.numpy()), O.prior_box())
    assert np.array_equal(det[..., 0] > 0, g[f'{name}.det'][..., 0] > 0)
    # exact fp32 score ties are visited in an order that is an accident of torch's unstable sort: compare the rows
    # as a set (canonical order), values to 2e-5
    assert same_detections(det, g[f'{name}.det'], 2e-5)


@pytest.mark.parametrize('name,flags', [
    ('gssd', dict()),
    ('gssdpp', dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True)),
])
def test_end_to_end_b32_vs_reference(golden, name, flags):
    """The BENCHMARKED batch (B = 32, BASELINE configs[1] / configs[2]) against the imported reference's own outputs
    (tests/golden/make_golden_b32.py: models/ssd_multiphase_custom_group.py:217-400 + multibox_loss.py:46-120): the oracle that
    tests/test_gpu_parity.py::test_full_size_parity compares the HIP path with is pinned at this size too, incl. the 1 x 1 map's priors."""
    g = golden('e2e_b32')
    B = int(g['batch'])
    shapes = {k: eval(s) for k, s in zip([str(k) for k in golden('e2e')[f'{name}.keys']], golden('e2e')[f'{name}.shapes'])}
    sd = synth.synth_state_dict(shapes, seed=int(g['wseed']))
    x = synth.synth_images(B, seed=int(g['xseed']))
    torch.set_num_threads(8)
    with torch.no_grad():
        loc, conf, upd = O.gssd_forward(sd, x, **flags)
    l, c = loc.numpy(), conf.numpy()
    assert np.abs(l.reshape(-1)[g[f'{name}.loc_idx']] - g[f'{name}.loc_val']).max() / g[f'{name}.loc_absmax'] < 1e-5
    assert np.abs(c.reshape(-1)[g[f'{name}.conf_idx']] - g[f'{name}.conf_val']).max() / g[f'{name}.conf_absmax'] < 1e-5
    assert np.abs(l[:, 8728:] - g[f'{name}.loc_1x1']).max() / g[f'{name}.loc_absmax'] < 1e-5
    assert np.abs(c[:, 8728:] - g[f'{name}.conf_1x1']).max() / g[f'{name}.conf_absmax'] < 1e-5
    ll, lc = O.multibox_loss(l, c, O.prior_box(), [t.numpy() for t in synth.synth_targets(B, int(g['xseed']))])[:2]
    assert rel(ll, g[f'{name}.loss'][0]) < 1e-5 and rel(lc, g[f'{name}.loss'][1]) < 1e-5
    for k in ('vgg.1.running_mean', 'vgg.41.running_var', 'bn_fuse_11.running_mean', 'extras.15.running_var'):
        assert rel(upd[k].numpy(), g[f'{name}.after.{k}']) < 1e-5, k


def test_self_attn_gradients_vs_reference(golden):
    """Gradient semantics of the Self_Attn restatement against the imported reference's own autograd (tests/golden/make_golden_grad.py:
    layers/self_attn.py:46-89 through layers/spectral_norm.py:39-89, float64): d(x) and every parameter gradient, incl. the spectral-normed
    weights -- the power iteration is outside the graph (torch.no_grad(), :74-81), sigma = u^T W v is differentiated through W only."""
    g = golden('grad')
    shapes = {k[5:]: g[k].shape for k in g.files if k.startswith('grad.')}
    from gssd.modules import Self_Attn
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in Self_Attn(64).state_dict().items()}, seed=21)
    sd64 = {k: (v.double().requires_grad_() if k in shapes else (v.double() if v.is_floating_point() else v)) for k, v in sd.items()}
    pre = {'self_attn_list.0.' + k: v for k, v in sd64.items()}
    x = torch.from_numpy(g['x']).requires_grad_()
    out = O.self_attn(x, pre, 'self_attn_list.0', True)[0]
    (out * torch.from_numpy(g['r'])).sum().backward()
    assert rel(x.grad.numpy(), g['dx']) < 1e-10
    for k in shapes:
        if np.abs(g['grad.' + k]).max() < 1e-9:                  # phi's bias: softmax is shift invariant along the keys
            assert sd64[k].grad.abs().max() < 1e-9
            continue
        assert rel(sd64[k].grad.numpy(), g['grad.' + k]) < 1e-10, k


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


FLAG_NETS = {       # tests/golden/make_golden_flags.py: the driver-reachable non-default constructor flags
    'nofuse': dict(use_fuseconv=False, use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                   dcn_cat_sab=True),
    'nobn': dict(batch_norm=False, use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                 dcn_cat_sab=True),
    'nobn_plain': dict(batch_norm=False, use_fuseconv=False),
    'mpf2': dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True,
                 max_pool_factor=2),
    'mpf3_sa': dict(use_self_attention=True, use_self_attention_base=True, max_pool_factor=3),
    'fs2': dict(feature_scale=2),
    'dcn2_detach': dict(use_self_attention=True, use_self_attention_base=True, num_dcn_layers=2, groups_dcn=1, dcn_cat_sab=True),
    'dcn_nocat': dict(num_dcn_layers=1, groups_dcn=4),
    'g1': dict(groups_vgg=1, groups_extra=1),
    'g2pp': dict(groups_vgg=2, groups_extra=2, use_self_attention=True, use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4,
                 dcn_cat_sab=True),
    'g4e1': dict(groups_extra=1),
}


@pytest.mark.parametrize('name', ['nofuse', 'nobn', 'nobn_plain', 'mpf2', 'mpf3_sa', 'fs2', 'dcn2_detach', 'dcn_nocat', 'g1', 'g2pp', 'g4e1'])
def test_constructor_flags_vs_reference(golden, name):
    """--use_fuseconv False / --batch_norm False / --max_pool_factor / --feature_scale / --groups_vgg, --groups_extra 1, 2
    (train_lesion_multiphase_v2.py:47-77):
    the oracle graph against what the imported reference computed (flags.npz)."""
    g = golden('flags')
    keys = [str(k) for k in g[f'{name}.keys']]
    shapes = {k: eval(s) for k, s in zip(keys, g[f'{name}.shapes'])}
    sd = synth.synth_state_dict(shapes, seed=1111)
    x = synth.synth_images(2, seed=5)
    torch.set_num_threads(8)
    with torch.no_grad():
        loc, conf, upd = O.gssd_forward(sd, x, **FLAG_NETS[name])
    l, c = loc.numpy().reshape(-1), conf.numpy().reshape(-1)
    assert np.abs(l[g[f'{name}.loc_idx']] - g[f'{name}.loc_val']).max() / g[f'{name}.loc_absmax'] < 1e-5
    assert np.abs(c[g[f'{name}.conf_idx']] - g[f'{name}.conf_val']).max() / g[f'{name}.conf_absmax'] < 1e-5
    ll, lc = O.multibox_loss(loc.numpy(), conf.numpy(), O.prior_box(), [t.numpy() for t in synth.synth_targets(2, 5)])
    assert rel(ll, g[f'{name}.loss'][0]) < 1e-4 and rel(lc, g[f'{name}.loss'][1]) < 1e-4
    for k in [k for k in g.files if k.startswith(f'{name}.after.')]:
        assert rel(upd[k[len(name) + 7:]].numpy(), g[k]) < 1e-5, k


def test_vanilla_ssd_config0(golden):
    """BASELINE.json configs[0]: vanilla VGG-SSD300, 1 phase, batch 2, CPU forward + MultiBoxLoss."""
    g = golden('e2e')
    keys = [str(k) for k in g['ssd.keys']]
    shapes = {k: eval(s) for k, s in zip(keys, g['ssd.shapes'])}
    sd = synth.synth_state_dict(shapes, seed=1111)
    x = synth.synth_images(2, seed=6, channels=3)
    with torch.no_grad():
        loc, conf = O.vanilla_ssd_forward(sd, x)
    l, c = loc.numpy().reshape(-1), conf.numpy().reshape(-1)
    assert rel(l[g['ssd.loc_idx']], g['ssd.loc_val']) < 1e-5
    assert rel(c[g['ssd.conf_idx']], g['ssd.conf_val']) < 1e-5
    ll, lc = O.multibox_loss(loc.numpy(), conf.numpy(), O.prior_box(), [t.numpy() for t in synth.synth_targets(4, 5)[:2]])
    assert rel(ll, g['ssd.loss'][0]) < 1e-5 and rel(lc, g['ssd.loss'][1]) < 1e-5
    assert bool(g['ssd.grad_finite'])


# --------------------------------------------------------------------------------------------------
# input stage: Pillow resample restatement vs the reference's base_transform_fast / ResizeFast
# --------------------------------------------------------------------------------------------------
def test_input_stage_golden(golden):
    import hashlib
    from oracle import input_oracle as IO
    from gssd import synth
    g = golden('input')
    mean = (49., 49., 49.)
    assert np.array_equal(IO.base_transform(g['small_in'], 37, mean, True), g['small_out_norm'])
    assert np.array_equal(IO.base_transform(g['small_in'], 37, mean, False), g['small_out_raw'])
    assert np.array_equal(IO.base_transform(g['up_in'], 33, mean, True), g['up_out_norm'])          # up-scaling branch
    big = synth.synth_study_u8(777, 4, 512)
    assert hashlib.sha256(big.tobytes()).digest() == g['big_in_sha'].tobytes()
    out = IO.base_transform(big, 300, mean, True)
    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest() == g['big_out_sha'].tobytes()
    assert np.array_equal(out.reshape(-1)[g['big_sample_idx']], g['big_sample'])
    # ResizeFast: (x * 255).astype(uint8) -> resize -> / 255
    xf = g['resizefast_in']
    rz = np.stack([IO.pil_resize_u8((xf[i] * 255).astype(np.uint8), 37).astype(np.float32) / 255. for i in range(4)])
    assert np.array_equal(rz, g['resizefast_out'])
    assert IO.to_network_input(out).shape == (12, 300, 300)


# --------------------------------------------------------------------------------------------------
# AP / IoBB evaluator vs the reference's own test_net on the same synthetic detections
# --------------------------------------------------------------------------------------------------
def eval_case(g, case):
    det = g[f'c{case}_det']
    S = float(g[f'c{case}_size'])
    off = np.concatenate([[0], np.cumsum(g[f'c{case}_gt_n'])])
    gts = [g[f'c{case}_gt'][off[i]:off[i + 1]] for i in range(det.shape[0])]
    scales = np.full((det.shape[0], 4), S, np.float32)
    return det, scales, gts


def test_evaluator_golden(golden):
    from oracle import eval_oracle as EO
    g = golden('eval')
    for case in (0, 1):
        det, scales, gts = eval_case(g, case)
        for use07 in (True, False):
            ap, iobb = EO.evaluate(det, scales, gts, 0.05, (0.1, 0.5), (0.1, 0.5), use07)
            assert np.array_equal(np.array(ap), g[f'c{case}_ap_{int(use07)}']), (case, use07)
            assert np.array_equal(np.array(iobb), g[f'c{case}_iobb_{int(use07)}']), (case, use07)
