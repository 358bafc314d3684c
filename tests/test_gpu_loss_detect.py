"""GPU parity tests of the index work and the callers either side of the network: match / MultiBoxLoss / Detect / NMS (bit-exact index
contracts), the device input stage, the AP / IoBB evaluator and the test_net drop-in.

Tolerances (BASELINE.json north_star): integer / index outputs bit-exact; fp32 activations and losses <= 1e-4 relative
(max-abs-diff / max-abs-ref per tensor).  Everything goes through the C ABI (ctypes -> libgssd_hip.so).
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_common import *                      # noqa: E402,F401,F403  (fixtures dev / ops, rel, TOL, nhwc / nchw, NETS, FLAG_NETS, same_detections)
from gpu_common import O, synth, ROOT, _stage_errors     # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_match_bit_exact(dev, ops, golden):
    g = golden('match')
    pri = torch.from_numpy(O.prior_box()).to(dev)
    n = int(g['n'])
    targets = [torch.from_numpy(g[f't{i}']) for i in range(n)]
    tg, ngt = ops.pack_targets(targets, dev)
    loc_t, conf_t = ops.match_batch(tg, ngt, pri)
    for i in range(n):
        ct = conf_t[i].cpu().numpy()
        assert np.array_equal(ct.astype(np.int8), g[f'conf{i}']), f'case {i}'          # vs the reference
        lo, co, _ = O.match(0.5, g[f't{i}'][:, :-1], O.prior_box(), (0.1, 0.2), g[f't{i}'][:, -1])
        assert np.array_equal(ct, co)                                                # vs the oracle
        pos = ct > 0
        ref = g[f'locpos{i}']
        fin = np.isfinite(ref)
        got = loc_t[i].cpu().numpy()[pos]
        assert np.allclose(got[fin], ref[fin], rtol=2e-6, atol=2e-6)


def test_box_utils_api(dev, golden):
    from layers import box_utils
    g = golden('match')
    pri = torch.from_numpy(O.prior_box()).to(dev)
    t = torch.from_numpy(g['t1']).to(dev)
    loc_t = torch.zeros(2, 8732, 4, device=dev)
    conf_t = torch.zeros(2, 8732, dtype=torch.long, device=dev)
    box_utils.match(0.5, t[:, :-1], pri, [0.1, 0.2], t[:, -1], loc_t, conf_t, 1)
    assert np.array_equal(conf_t[1].cpu().numpy().astype(np.int8), g['conf1']) and (conf_t[0] == 0).all()
    # nms() on raw boxes
    rng = np.random.default_rng(2)
    c = rng.uniform(0.2, 0.8, size=(300, 2))
    wh = rng.uniform(0.05, 0.3, size=(300, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = rng.uniform(0.02, 1, size=300).astype(np.float32)
    keep_ref = O.nms(boxes, scores, 0.45, 200)
    keep, cnt = box_utils.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.45, 200)
    assert cnt == keep_ref.shape[0] and np.array_equal(keep[:cnt].cpu().numpy(), keep_ref)


def test_multibox_loss(dev, ops, golden):
    from layers.modules import MultiBoxLoss
    g = golden('loss')
    m = golden('match')
    pri_np = O.prior_box()
    pri = torch.from_numpy(pri_np).to(dev)
    P = pri_np.shape[0]
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    ll, lc = crit((torch.zeros(2, P, 4, device=dev), torch.zeros(2, P, 2, device=dev), pri),
                  [torch.from_numpy(m['t0']), torch.from_numpy(m['t1'])])
    assert abs(ll.item() - g['zero_loss'][0]) < 1e-5 and abs(lc.item() - g['zero_loss'][1]) < 1e-5
    for ci in range(3):
        rng = np.random.default_rng(int(g[f'seed{ci}']))
        loc = rng.normal(0, 1.0, size=(4, P, 4)).astype(np.float32)
        conf = rng.normal(0, 2.0, size=(4, P, 2)).astype(np.float32)
        tg = [torch.from_numpy(g[f'tg{ci}_{b}']) for b in range(4)]
        loc_d = torch.from_numpy(loc).to(dev).requires_grad_()
        conf_d = torch.from_numpy(conf).to(dev).requires_grad_()
        ll, lc = crit((loc_d, conf_d, pri), tg)
        assert rel(ll, g[f'loss{ci}'][0]) < TOL and rel(lc, g[f'loss{ci}'][1]) < TOL
        (ll + lc).backward()
        idx = np.random.default_rng(3).choice(4 * P * 4, size=512, replace=False)
        assert np.allclose(loc_d.grad.cpu().numpy().reshape(-1)[idx], g[f'gloc_sample{ci}'], rtol=1e-4, atol=1e-7)
        idx = np.random.default_rng(3).choice(4 * P * 2, size=512, replace=False)
        assert np.allclose(conf_d.grad.cpu().numpy().reshape(-1)[idx], g[f'gconf_sample{ci}'], rtol=1e-4, atol=1e-7)
        # masks: positives bit-exact; mined negatives: exact count, identical to the reference's set except among priors
        # whose score is within 2 ulp of the cut-off (tests/helpers.py states the contract)
        tgp, ngt = ops.pack_targets(tg, dev)
        st = ops.multibox_loss_forward(loc_d.detach(), conf_d.detach(), pri, tgp, ngt, want_scores=True)
        sel = st['sel'].cpu().numpy()
        assert np.array_equal(np.packbits((sel & 1).astype(bool)), g[f'pos{ci}'])
        neg_ref = np.unpackbits(g[f'neg{ci}'])[:4 * P].reshape(4, P).astype(bool)
        neg = (sel & 2).astype(bool)
        from helpers import assert_mined_negatives_contract
        lca_o = O.multibox_loss(loc, conf, pri_np, [t.numpy() for t in tg], details=True)[2]['loss_c_all']
        assert_mined_negatives_contract(neg, neg_ref, lca_o, (sel & 1).sum(1))
        assert np.abs(st['loss_c_all'].cpu().numpy() - lca_o).max() <= 4 * np.spacing(np.float32(np.abs(lca_o).max()))
        # the selection logic itself is exact: oracle ranking of the kernel's own scores gives the same set
        lca = st['loss_c_all'].cpu().numpy()
        order = np.argsort(-lca, axis=1, kind='stable')
        rank = np.argsort(order, axis=1, kind='stable')
        npos = (sel & 1).sum(1, keepdims=True)
        assert np.array_equal(neg, rank < np.minimum(3 * npos, P - 1))


def test_detect_bit_exact(dev, ops, golden):
    from layers.functions import Detect
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    from make_golden import closed_form_detect_inputs
    g = golden('detect')
    pri_np = O.prior_box()
    pri = torch.from_numpy(pri_np).to(dev)
    loc, _ = closed_form_detect_inputs(pri_np.shape[0], 2)
    conf_sm = g['conf_sm']
    out, keep, cnt = ops.detect(torch.from_numpy(loc).to(dev), torch.from_numpy(conf_sm).to(dev), pri, 2, want_keep=True)
    ref_out, ref_keep = O.detect(2, 0, 200, 0.01, 0.45, loc, conf_sm, pri_np, return_keep=True)
    for b in range(2):
        k = keep[b, 1, :int(cnt[b, 1])].cpu().numpy()
        assert np.array_equal(k, g[f'keep{b}'])                 # vs the reference (exact ties included)
        assert np.array_equal(k, ref_keep[(b, 1)])              # vs the oracle
    assert np.array_equal(out.cpu().numpy(), ref_out)           # bit-exact vs the oracle (same exp recipe)
    assert np.allclose(out.cpu().numpy(), g['out'], rtol=0, atol=2e-6)
    # the autograd.Function API of the reference
    out2 = Detect.apply(2, 0, 200, 0.01, 0.45, torch.from_numpy(loc).to(dev), torch.from_numpy(conf_sm).to(dev), pri)
    assert torch.equal(out2, out)
    with pytest.raises(ValueError):
        Detect.apply(2, 0, 200, 0.01, 0.0, torch.from_numpy(loc).to(dev), torch.from_numpy(conf_sm).to(dev), pri)
    # random logits, > 200 candidates
    rng = np.random.default_rng(5)
    loc_r = rng.normal(0, 0.5, size=(2, pri_np.shape[0], 4)).astype(np.float32)
    sm_r = g['conf_sm_rand']
    out_r = ops.detect(torch.from_numpy(loc_r).to(dev), torch.from_numpy(sm_r).to(dev), pri, 2).cpu().numpy()
    assert np.array_equal(out_r, O.detect(2, 0, 200, 0.01, 0.45, loc_r, sm_r, pri_np))
    assert np.allclose(out_r, g['out_rand'], rtol=0, atol=3e-6)
    # edge: nothing above threshold -> all zeros
    z = ops.detect(torch.from_numpy(loc_r).to(dev), torch.zeros(2, pri_np.shape[0], 2, device=dev), pri, 2)
    assert (z == 0).all()


def test_input_stage_bit_exact(dev, golden):
    import hashlib
    from gssd.input_stage import DeviceInputStage
    from oracle import input_oracle as IO
    from data import BaseTransform
    g = golden('input')
    mean = (49., 49., 49.)
    for key_in, size, norm, key_out in (('small_in', 37, True, 'small_out_norm'), ('small_in', 37, False, 'small_out_raw'),
                                        ('up_in', 33, True, 'up_out_norm')):
        raw = torch.from_numpy(g[key_in]).unsqueeze(0).to(dev)
        x = DeviceInputStage(size, mean, norm)(raw)
        ref = IO.to_network_input(g[key_out])                                   # the reference's own output
        assert x.shape == (1, 12, size, size)
        assert np.array_equal(x[0].cpu().numpy(), ref), key_out
    # drop-in BaseTransform: same call, same shape as the reference's numpy result
    xb, _, _ = BaseTransform(37, np.array(mean), use_normalize=True)(g['small_in'])
    assert xb.is_cuda and np.array_equal(xb.cpu().numpy(), g['small_out_norm'])
    # the real geometry, a batch of different studies; bilinear too (vs the oracle)
    raws = np.stack([synth.synth_study_u8(777 + i, 4, 512) for i in range(3)])
    x = DeviceInputStage(300, mean, True)(torch.from_numpy(raws).to(dev)).cpu().numpy()
    out0 = np.transpose(x[0].reshape(4, 3, 300, 300), (0, 2, 3, 1))
    assert hashlib.sha256(np.ascontiguousarray(out0).tobytes()).digest() == g['big_out_sha'].tobytes()
    for i in (1, 2):
        assert np.array_equal(x[i], IO.to_network_input(IO.base_transform(raws[i], 300, mean, True)))
    xl = DeviceInputStage(300, mean, True, filt='bilinear')(torch.from_numpy(raws[:1]).to(dev)).cpu().numpy()
    assert np.array_equal(xl[0], IO.to_network_input(IO.base_transform(raws[0], 300, mean, True, filt='bilinear')))
    # other geometries of the LDS-staged kernels (whole 32-bit words per row): a small one, and a 5.12 x reduction whose vertical
    # window (73 input rows for 10 output rows) does not fit the staged column, so every tap is fetched in place
    for src, size in ((64, 40), (512, 100)):
        rs = np.stack([synth.synth_study_u8(31 + i, 4, src) for i in range(2)])
        xs = DeviceInputStage(size, mean, True)(torch.from_numpy(rs).to(dev)).cpu().numpy()
        for i in range(2):
            assert np.array_equal(xs[i], IO.to_network_input(IO.base_transform(rs[i], size, mean, True))), (src, size, i)


def test_evaluator_vs_reference_and_oracle(dev, golden):
    from gssd.evaluator import DeviceEvaluator
    from oracle import eval_oracle as EO
    from test_oracle_golden import eval_case
    g = golden('eval')
    for case in (0, 1):
        det, scales, gts = eval_case(g, case)
        for use07 in (True, False):
            ev = DeviceEvaluator(0.05, (0.1, 0.5), (0.1, 0.5), use07)
            confs, flags = [], []
            for s in range(0, det.shape[0], 16):                              # two batches: accumulation across add_batch
                c, f = ev.add_batch(torch.from_numpy(det[s:s + 16]).to(dev), scales[s:s + 16], gts[s:s + 16])
                confs.append(c.cpu().numpy())
                flags.append(f.cpu().numpy())
            ap, iobb = ev.result()
            ref_ap, ref_iobb = g[f'c{case}_ap_{int(use07)}'], g[f'c{case}_iobb_{int(use07)}']
            if use07:
                assert np.array_equal(np.array(ap), ref_ap) and np.array_equal(np.array(iobb), ref_iobb), (case, ap, ref_ap)
            else:                                                             # np.sum's pairwise order is not reproduced
                assert np.allclose(ap, ref_ap, rtol=1e-13, atol=0) and np.allclose(iobb, ref_iobb, rtol=1e-13, atol=0)
            # TP / FP flags against the oracle's greedy loop (integer work: exact)
            _, _, dt = EO.evaluate(det, scales, gts, 0.05, (0.1, 0.5), (0.1, 0.5), use07, details=True)
            conf = np.concatenate(confs)
            fl = np.concatenate(flags, axis=1)
            order = np.argsort(-conf, kind='stable')[:len(dt['conf'])]
            assert np.array_equal(conf[order].astype(np.float64), dt['conf'])
            assert np.array_equal(fl[:, order] == 1, dt['tp'] == 1) and np.array_equal(fl[:, order] == 2, dt['fp'] == 1)
    # no detections at all -> zeros, like the reference's early exit
    ev = DeviceEvaluator(0.05, (0.5,), (0.1,), True)
    ev.add_batch(torch.zeros(2, 2, 200, 5, device=dev), np.full((2, 4), 512., np.float32), [np.zeros((1, 4)), np.zeros((0, 4))])
    assert ev.result() == ([0.0], [0.0])


def test_test_net_dropin(dev):
    """``test_ap_iobb.test_net`` with the reference's calling convention on a tiny synthetic 'dataset'."""
    import test_ap_iobb as T
    from data import BaseTransform
    from models.ssd_multiphase_custom_group import build_ssd
    net = build_ssd('test', 300, 2, *NETS['gssd'][1])
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=3))
    net = net.to(dev).eval()

    class Set:
        name = 'lesion_test_ap_synth'
        def __len__(self): return 3
        def pull_image(self, i): return synth.synth_study_u8(40 + i, 4, 128)
        def pull_anno(self, i): return np.array([[20., 30., 60., 80., 0.]])
    ap, iobb = T.test_net(net, True, Set(), BaseTransform(300, (49., 49., 49.), use_normalize=True), 300, thresh=0.05,
                          mode='v2', use_07_metric=True, ap_list=[0.1, 0.5], iobb_list=[0.1, 0.5], batch_size=2)
    assert len(ap) == 2 and len(iobb) == 2 and all(0. <= v <= 1. for v in ap + iobb)
    rec, prec = np.array([0.2, 0.4, 0.4, 0.8]), np.array([1.0, 1.0, 0.66, 0.5])
    from oracle import eval_oracle as EO
    for m in (True, False):
        assert abs(T.voc_ap(rec, prec, m) - EO.voc_ap(rec, prec, m)) < 1e-15
