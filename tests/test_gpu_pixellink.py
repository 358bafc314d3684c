"""GPU parity tests of the PixelLink++ row through the C ABI: the HIP path against the fixtures generated from
the imported reference (tests/golden/pixellink.npz) and against the CPU oracle on seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import pixellink_oracle as PO      # noqa: E402
from gssd import synth                         # noqa: E402
from test_pixellink_cpu import FULL, PLAIN, build, rel   # noqa: E402

TOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


@pytest.mark.parametrize('Hs,Hd', [(19, 19), (19, 38), (38, 75), (19, 75), (7, 1)])
def test_interp_add(dev, Hs, Hd):
    """gssd_interp_add_f32 vs F.interpolate(mode='bilinear', align_corners=True) (+ the lateral add), NHWC 18 channels."""
    from gssd import _lib
    g = torch.Generator().manual_seed(Hs * 100 + Hd)
    B, C = 3, 18
    src = torch.randn(B, C, Hs, Hs, generator=g)
    add = torch.randn(B, C, Hd, Hd, generator=g)
    ref = F.interpolate(src, size=(Hd, Hd), mode='bilinear', align_corners=True)
    s_d, a_d = src.permute(0, 2, 3, 1).contiguous().to(dev), add.permute(0, 2, 3, 1).contiguous().to(dev)
    out, out2 = torch.empty(B, Hd, Hd, C, device=dev), torch.empty(B, Hd, Hd, C, device=dev)
    _lib.check(_lib.lib.gssd_interp_add_f32(s_d.data_ptr(), a_d.data_ptr(), out.data_ptr(), out2.data_ptr(), B, Hs, Hs, Hd, Hd, C,
                                            torch.cuda.current_stream().cuda_stream))
    # fp32, same formula; hipcc contracts the two inner products into FMAs, ATen does not: a few ulp of the largest element
    assert rel(out.permute(0, 3, 1, 2).cpu(), ref) < 5e-6
    assert rel(out2.permute(0, 3, 1, 2).cpu(), ref + add) < 5e-6
    if Hs == Hd:
        assert torch.equal(out.cpu(), s_d.cpu())           # same size: the identity, bit for bit


@pytest.mark.parametrize('tag,kw', [('full', FULL), ('plain', PLAIN)])
def test_model_vs_reference_fixture(dev, golden, tag, kw):
    """PixelLink.forward on the HIP plan (B = 1, train mode) vs the imported reference's outputs and buffer updates."""
    g = golden('pixellink')
    net = build(kw).to(dev).train()
    x = synth.synth_images(1, seed=300).to(dev)
    with torch.no_grad():
        o1, o2 = net(x)
    assert tuple(o1.shape) == (1, 2, 75, 75) and tuple(o2.shape) == (1, 16, 75, 75)
    assert rel(o1.cpu(), g[f'model_{tag}_out1']) < TOL and rel(o2.cpu(), g[f'model_{tag}_out2']) < TOL
    sd = net.state_dict()
    assert rel(sd['bn_fuse3.running_mean'].cpu(), g[f'model_{tag}_bn_fuse3_rm']) < TOL
    assert int(sd['bn_fuse3.num_batches_tracked']) == 1
    if kw['use_self_attention']:
        assert rel(sd['self_attn_list.0.snconv1x1_theta.weight_u'].cpu(), g[f'model_{tag}_sa0_u']) < TOL


@pytest.mark.parametrize('training,tag', [(True, 'full'), (False, 'plain')])
def test_model_vs_oracle_batch(dev, training, tag):
    """B = 3 vs the CPU oracle: train mode (batch statistics over several images) on the full graph; eval mode (running statistics,
    no power iteration) on the convolution-only graph -- with the synthetic running statistics the un-normalised eval activations of
    the full graph reach 1e12 and saturate every attention softmax to one-hot, so an elementwise comparison there would only measure
    which of two near-equal logits wins; the full graph's eval mode is checked for determinism and finiteness instead."""
    kw = FULL if tag == 'full' else PLAIN
    net = build(kw)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = synth.synth_images(3, seed=301)
    with torch.no_grad():
        r1, r2, _ = PO.pixellink_forward(sd, x, training=training, **kw)
    net = net.to(dev)
    net.train(training)
    with torch.no_grad():
        o1, o2 = net(x.to(dev))
    assert rel(o1.cpu(), r1) < TOL and rel(o2.cpu(), r2) < TOL
    if not training:
        with torch.no_grad():
            o1b, o2b = net(x.to(dev))
        assert torch.equal(o1, o1b) and torch.equal(o2, o2b)
        full = build(FULL).to(dev).eval()
        with torch.no_grad():
            f1, f2 = full(x.to(dev))
            g1, g2 = full(x.to(dev))
        assert torch.isfinite(f1).all() and torch.isfinite(f2).all() and torch.equal(f1, g1) and torch.equal(f2, g2)


def test_forward_needs_no_grad(dev):
    from gssd import _lib
    net = build(PLAIN).to(dev)
    with pytest.raises(_lib.GssdError):
        net(synth.synth_images(1, seed=1).to(dev))
    with pytest.raises(_lib.GssdError):
        with torch.no_grad():
            net(synth.synth_images(1, seed=1))           # CPU input: no fallback


def test_loss_vs_reference_fixture(dev, golden):
    """PixelLinkLoss (pixel_loss + link_loss) on the HIP kernel vs the imported reference criterion: the four losses, neg_area, and the
    mined-negative mask -- equal except for candidates whose background probability is within 2 ulp of the cut-off (the kernel's
    expf / divide round differently from ATen's vectorised softmax; the reference's own <= makes exact ties all-selected)."""
    from pixel_link.criterion import PixelLinkLoss
    g = golden('pixellink')
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    crit = PixelLinkLoss()
    pp, pn = crit.pixel_loss(t('loss_out1'), t('loss_pix'), t('loss_neg'), t('loss_posw'), link=(t('loss_out2'), t('loss_link')))
    lp, ln = crit.link_loss(t('loss_out2'), t('loss_link'))
    got = np.array([float(pp), float(pn), float(lp), float(ln)])
    assert rel(got, g['loss_vals']) < 1e-5
    assert np.array_equal(crit.neg_area.cpu().numpy(), g['loss_neg_area'])
    mine, ref = crit.neg_pixel_weight.cpu().numpy().astype(bool), g['loss_neg_weight'].astype(bool)
    diff = mine != ref
    if diff.any():
        p0 = torch.softmax(torch.from_numpy(g['loss_out1']), 1)[:, 0].numpy()
        for b in range(p0.shape[0]):
            cand = np.sort(p0[b][g['loss_neg'][b] == 1])
            thr = cand[int(g['loss_neg_area'][b]) - 1]
            assert np.all(np.abs(p0[b][diff[b]] - thr) <= 2 * np.spacing(np.float32(thr)))
    assert diff.sum() <= 4
    # the two-launch calling pattern of the reference driver (pixel_loss, then link_loss) gives the same numbers
    crit2 = PixelLinkLoss()
    crit2.pixel_loss(t('loss_out1'), t('loss_pix'), t('loss_neg'), t('loss_posw'))
    lp2, ln2 = crit2.link_loss(t('loss_out2'), t('loss_link'))
    assert float(lp2) == float(lp) and float(ln2) == float(ln)


def test_decode_vs_reference_fixture(dev, golden):
    """Link decoding: the device label maps equal postprocess.func's (bit-exact, components numbered in raster order of their first
    pixel); the per-component statistics equal numpy reductions over those label maps."""
    from pixel_link import postprocess
    g = golden('pixellink')
    o1, o2 = torch.from_numpy(g['dec_out1']).to(dev), torch.from_numpy(g['dec_out2']).to(dev)
    labels, comps, ncomp = postprocess.decode(o1, o2)
    lab = labels.cpu().numpy()
    assert np.array_equal(lab, g['dec_labels'])
    score = torch.softmax(torch.from_numpy(g['dec_out1']), 1)[:, 1].numpy()
    comps = comps.cpu().numpy()
    for b in range(lab.shape[0]):
        n = int(lab[b].max())
        assert int(ncomp[b]) == n
        for c in range(1, n + 1):
            ys, xs = np.where(lab[b] == c)
            cnt, x0, y0, x1, y1, ssum = comps[b, c - 1]
            assert (cnt, x0, y0, x1, y1) == (len(ys), xs.min(), ys.min(), xs.max(), ys.max())
            assert abs(ssum - score[b][ys, xs].sum()) <= 1e-4 * max(1.0, ssum)
    boxes = postprocess.mask_to_box(o1, o2)
    assert len(boxes) == lab.shape[0] and all(len(d) <= int(lab[b].max()) for b, d in enumerate(boxes))
    # edge cases: nothing positive; everything positive and fully linked -> one component
    z1 = torch.zeros(1, 2, 20, 20, device=dev)
    z1[:, 0] = 5.0
    l0, _, n0 = postprocess.decode(z1, torch.zeros(1, 16, 20, 20, device=dev))
    assert int(n0[0]) == 0 and int(l0.abs().max()) == 0
    a1 = torch.zeros(1, 2, 20, 20, device=dev)
    a1[:, 1] = 5.0
    a2 = torch.zeros(1, 16, 20, 20, device=dev)
    a2[:, 1::2] = 5.0
    l1, c1, n1 = postprocess.decode(a1, a2)
    assert int(n1[0]) == 1 and int(l1.min()) == 1 and int(l1.max()) == 1 and float(c1[0, 0, 0]) == 400.0


def test_full_size_properties(dev):
    """B = 32 (the bench batch), size-independent properties: eval-mode outputs of an image do not depend on its batch neighbours
    (plain graph: convolutions + BatchNorm on running statistics only); PixelLinkLoss on the device predictions equals the oracle
    criterion on the same predictions; the decoded label maps equal the oracle's decoding of the same predictions."""
    from pixel_link.criterion import PixelLinkLoss
    from pixel_link import postprocess
    B = 32
    net = build(PLAIN).to(dev).eval()
    x = synth.synth_images(B, seed=310).to(dev)
    with torch.no_grad():
        o1, o2 = net(x)
        s1, s2 = net(x[5:7].contiguous())
    assert torch.isfinite(o1).all() and torch.isfinite(o2).all()
    assert rel(s1.cpu(), o1[5:7].cpu()) < 1e-5 and rel(s2.cpu(), o2[5:7].cpu()) < 1e-5
    # masks: a box of positives per image, every other pixel a negative candidate
    rng = np.random.default_rng(3)
    pix = np.zeros((B, 75, 75), np.int64)
    for b in range(B):
        y, xx = rng.integers(5, 55, size=2)
        pix[b, y:y + 10 + b % 5, xx:xx + 8] = 1
    neg = (pix == 0).astype(np.uint8)
    posw = pix.astype(np.float32) * rng.uniform(0.5, 1.5, size=pix.shape).astype(np.float32)
    link = np.repeat(pix[:, None], 8, 1) * (rng.random((B, 8, 75, 75)) > 0.3)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    # scale the (untrained, huge) logits into a numerically meaningful range before the loss / decoding comparison
    p1 = (o1 / o1.abs().max() * 6).contiguous()
    p2 = (o2 / o2.abs().max() * 6).contiguous()
    crit = PixelLinkLoss()
    pp, pn = crit.pixel_loss(p1, t(pix), t(neg), t(posw), link=(p2, t(link.astype(np.int64))))
    lp, ln = crit.link_loss(p2, t(link.astype(np.int64)))
    rp, rn, rlp, rln, _ = PO.pixel_link_loss(p1.cpu(), p2.cpu(), torch.from_numpy(pix), torch.from_numpy(neg), torch.from_numpy(posw),
                                             torch.from_numpy(link.astype(np.int64)))
    assert rel([float(pp), float(pn), float(lp), float(ln)], [rp, rn, rlp, rln]) < 1e-5
    labels, _, ncomp = postprocess.decode(p1, p2)
    ref = PO.decode_links(p1.cpu(), p2.cpu())
    assert np.array_equal(labels.cpu().numpy(), ref) and int(ncomp.sum()) == int(sum(r.max() for r in ref))


@pytest.mark.parametrize('kw', [
    dict(cascade_fuse=False, use_fuseconv=True, batch_norm=False, use_self_attention=True, use_self_attention_base=False,
         num_dcn_layers=2, groups_dcn=1, dcn_cat_sab=False, detach_sab=False),
    dict(cascade_fuse=True, use_fuseconv=False, batch_norm=False, use_self_attention=False, use_self_attention_base=True,
         num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True, detach_sab=True),
], ids=['sa_2dcn_nobn', 'sab_detach_nofuse'])
def test_flag_combinations_vs_oracle(dev, kw):
    """Constructor flags off the bench configuration (model.py:20-188): Self_Attn without the base blocks, two stacked DCN layers
    without the concatenation, fuse conv without BatchNorm, no fuse conv at all, detach_sab (a forward no-op) -- vs the CPU oracle."""
    net = build(kw)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = synth.synth_images(2, seed=305)
    with torch.no_grad():
        r1, r2, _ = PO.pixellink_forward(sd, x, training=True, **kw)
    net = net.to(dev).train()
    with torch.no_grad():
        o1, o2 = net(x.to(dev))
    assert rel(o1.cpu(), r1) < TOL and rel(o2.cpu(), r2) < TOL
