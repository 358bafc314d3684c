"""CPU tests of the PixelLink++ row: the oracle restatement (oracle/pixellink_oracle.py) against the fixtures
generated from the imported reference (tests/golden/make_pixellink_golden.py), and the drop-in module's state-dict surface."""
import numpy as np
import pytest
import torch

from oracle import pixellink_oracle as PO
from gssd import synth

FULL = dict(cascade_fuse=True, use_fuseconv=True, batch_norm=True, use_self_attention=True, use_self_attention_base=True,
            num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True, detach_sab=False)
PLAIN = dict(cascade_fuse=False, use_fuseconv=True, batch_norm=True, use_self_attention=False, use_self_attention_base=False,
             num_dcn_layers=0, groups_dcn=1, dcn_cat_sab=False, detach_sab=False)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def build(kw):
    from pixel_link.model import PixelLink
    net = PixelLink(**kw)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=2222))
    return net


@pytest.mark.parametrize('tag,kw', [('full', FULL), ('plain', PLAIN)])
def test_module_surface_matches_reference(golden, tag, kw):
    """Same state-dict keys (incl. the modules_except_dcn aliases) and shapes as the reference module: its checkpoints load."""
    g = golden('pixellink')
    net = build(kw)
    mine = [f'{k}:{"x".join(map(str, v.shape))}' for k, v in net.state_dict().items()]
    assert sorted(mine) == sorted(g[f'model_{tag}_keys'].tolist())


@pytest.mark.parametrize('tag,kw', [('full', FULL), ('plain', PLAIN)])
def test_oracle_forward_vs_reference(golden, tag, kw):
    g = golden('pixellink')
    net = build(kw)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = synth.synth_images(1, seed=300)
    with torch.no_grad():
        o1, o2, upd = PO.pixellink_forward(sd, x, training=True, **kw)
    assert rel(o1, g[f'model_{tag}_out1']) < 2e-5 and rel(o2, g[f'model_{tag}_out2']) < 2e-5
    assert rel(upd['bn_fuse3.running_mean'], g[f'model_{tag}_bn_fuse3_rm']) < 1e-5
    if kw['use_self_attention']:
        assert rel(upd['self_attn_list.0.snconv1x1_theta.weight_u'], g[f'model_{tag}_sa0_u']) < 1e-5


def test_oracle_loss_vs_reference(golden):
    g = golden('pixellink')
    t = lambda k: torch.from_numpy(g[k])
    pp, pn, lp, ln, negw = PO.pixel_link_loss(t('loss_out1'), t('loss_out2'), t('loss_pix'), t('loss_neg'), t('loss_posw'), t('loss_link'))
    assert rel([pp, pn, lp, ln], g['loss_vals']) < 1e-6
    assert np.array_equal(negw.numpy().astype(np.uint8), g['loss_neg_weight'])


def test_oracle_decode_vs_reference(golden):
    g = golden('pixellink')
    lab = PO.decode_links(torch.from_numpy(g['dec_out1']), torch.from_numpy(g['dec_out2']))
    assert int(lab.max()) < 256                      # the reference stores labels as uint8
    assert np.array_equal(lab, g['dec_labels'])


# --------------------------------------------------------------------------------------------------
# the box tail (pixel_link/box_geometry.py = the definition of postprocess.py:124-160 without OpenCV)
# --------------------------------------------------------------------------------------------------
def _brute_min_area(points, n_angles=20000):
    """Independent check: the enclosing rectangle's area over a fine sweep of orientations (the optimum lies on a hull edge direction; the
    sweep gets within O(1 / n_angles) of it)."""
    p = np.asarray(points, np.float64)
    th = np.linspace(0.0, np.pi / 2, n_angles, endpoint=False)
    u = np.stack([np.cos(th), np.sin(th)], 1)
    v = np.stack([-np.sin(th), np.cos(th)], 1)
    a, b = p @ u.T, p @ v.T
    return float(((a.max(0) - a.min(0)) * (b.max(0) - b.min(0))).min())


def test_box_geometry_closed_form_cases():
    from pixel_link import box_geometry as G
    # a 3 x 5 block of mask pixels at (row 10.., col 20..) on a 75 x 75 map: every mask pixel covers a 4 x 4 block of the 300 x 300 image
    lab = np.zeros((75, 75), np.int32)
    lab[10:13, 20:25] = 1
    up = G.upscale_nearest(lab, (300, 300))
    ys, xs = np.nonzero(up)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (40, 51, 80, 99) and up.sum() == 12 * 20
    r = G.min_area_rect(np.stack([xs, ys], 1))
    assert sorted(r['size']) == [11.0, 19.0] and abs(r['area'] - 209.0) < 1e-9 and r['center'] == (89.5, 45.5)
    c = G.rect_corners_int(r['corners'], (300, 300))
    assert (c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max()) == (80, 40, 99, 51)
    # a diamond |x - 50| + |y - 60| <= 12: the minimum rectangle is the rotated square through its four tips (area 2 r^2 = 288, not the
    # axis-aligned 24 x 24 = 576); its integer corners are the tips themselves
    yy, xx = np.mgrid[0:120, 0:120]
    pts = np.stack([xx[np.abs(xx - 50) + np.abs(yy - 60) <= 12], yy[np.abs(xx - 50) + np.abs(yy - 60) <= 12]], 1)
    r = G.min_area_rect(pts)
    assert abs(r['area'] - 288.0) < 1e-9 and abs(r['size'][0] - 12 * 2 ** 0.5) < 1e-9 and abs(r['size'][1] - 12 * 2 ** 0.5) < 1e-9
    c = G.rect_corners_int(r['corners'], (120, 120))
    assert sorted(map(tuple, c.tolist())) == sorted([(38, 60), (62, 60), (50, 48), (50, 72)])
    # degenerate sets: one pixel, a straight line of pixels (zero height: the reference's min_height filter drops it)
    assert G.min_area_rect(np.array([[7, 9]]))['size'] == (0.0, 0.0)
    line = G.min_area_rect(np.stack([np.arange(10), 2 * np.arange(10)], 1))
    assert line['area'] < 1e-12 and min(line['size']) < 1e-12 and abs(max(line['size']) - 9 * 5 ** 0.5) < 1e-9
    # corners are clamped into the image like rect_to_xys (:58-69)
    c = G.rect_corners_int(np.array([[-3.2, 5.0], [310.7, 5.0], [310.7, 299.9], [-3.2, 299.9]]), (300, 300))
    assert c.tolist() == [[0, 5], [299, 5], [299, 299], [0, 299]]


def test_box_geometry_rotating_calipers_vs_brute_force():
    from pixel_link import box_geometry as G
    rng = np.random.default_rng(3)
    for _ in range(25):
        n = int(rng.integers(3, 60))
        pts = rng.integers(0, 80, size=(n, 2))
        r = G.min_area_rect(pts)
        b = _brute_min_area(pts)
        assert r['area'] <= b + 1e-9 and b - r['area'] < 2e-3 * max(b, 1.0), (r['area'], b)     # the sweep can only be slightly worse
        # the rectangle really encloses every point
        c = r['corners']
        u = (c[1] - c[0]) / max(np.linalg.norm(c[1] - c[0]), 1e-30)
        v = np.array([-u[1], u[0]])
        a, bb = (pts - c[0]) @ u, (pts - c[0]) @ v
        assert a.min() > -1e-9 and a.max() < r['size'][0] + 1e-9 and bb.min() > -1e-9 and bb.max() < r['size'][1] + 1e-9


def test_box_geometry_resizes_and_component_boxes():
    from pixel_link import box_geometry as G
    # bilinear (cv2.INTER_LINEAR, half-pixel centres, border replicated): a linear ramp stays linear in the interior, constants stay constant
    ramp = np.tile(np.arange(75, dtype=np.float32), (75, 1))
    up = G.upscale_bilinear(ramp, (300, 300))
    xs = (np.arange(300) + 0.5) * 0.25 - 0.5
    assert np.allclose(up[17], np.clip(xs, 0, 74), atol=1e-5) and np.allclose(G.upscale_bilinear(np.full((75, 75), 0.3, np.float32), (300, 300)), 0.3)
    # two components on one map: an axis-aligned block (kept) and a single pixel (a 4 x 4 block after up-scaling: 3 x 3 rectangle, kept since
    # min_height = 1, min_area = 3), plus a one-pixel-wide column that the filters keep too (3 x 39): boxes = the blocks' pixel bounds
    lab = np.zeros((75, 75), np.int32)
    lab[10:13, 20:25] = 1
    lab[40, 40] = 2
    lab[50:60, 7] = 3
    prob = np.full((75, 75), 0.25, np.float32)
    prob[10:13, 20:25] = 0.9
    boxes, scores = G.component_boxes(lab, prob, (300, 300), 1, 3)
    assert boxes == [[80, 40, 99, 51], [160, 160, 163, 163], [28, 200, 31, 239]]
    assert 0.6 < scores[0] <= 0.9 and abs(scores[1] - 0.25) < 1e-6      # (the block's border pixels blend with the 0.25 background)
    # the reference's filters: with min_height 5 the single pixel and the thin column disappear
    boxes, _ = G.component_boxes(lab, prob, (300, 300), 5, 3)
    assert boxes == [[80, 40, 99, 51]]
