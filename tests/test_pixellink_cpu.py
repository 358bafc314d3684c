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
