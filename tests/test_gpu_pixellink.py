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


def test_forward_error_behaviour(dev):
    from gssd import _lib
    net = build(PLAIN).to(dev)
    with pytest.raises(_lib.GssdError):
        with torch.no_grad():
            net(synth.synth_images(1, seed=1))           # CPU input: no fallback
    net.eval()
    with pytest.raises(_lib.GssdError):
        net(synth.synth_images(1, seed=1).to(dev))       # grad-enabled forward in eval mode: the backward differentiates batch statistics
    net.train()
    o1, o2 = net(synth.synth_images(1, seed=1).to(dev))
    assert o1.requires_grad and o2.requires_grad


@pytest.mark.parametrize('Hs,Hd', [(19, 19), (19, 38), (38, 75), (19, 75)])
def test_interp_add_backward(dev, Hs, Hd):
    """gssd_interp_add_bwd_f32 (channel stride 20) vs autograd through F.interpolate + the lateral add."""
    from gssd import _lib
    g = torch.Generator().manual_seed(Hs * 7 + Hd)
    B, C, LD = 2, 18, 20
    src = torch.randn(B, C, Hs, Hs, generator=g, dtype=torch.float64, requires_grad=True)
    add = torch.randn(B, C, Hd, Hd, generator=g, dtype=torch.float64, requires_grad=True)
    go, go2 = torch.randn(B, C, Hd, Hd, generator=g, dtype=torch.float64), torch.randn(B, C, Hd, Hd, generator=g, dtype=torch.float64)
    up = F.interpolate(src, size=(Hd, Hd), mode='bilinear', align_corners=True)
    ((up * go).sum() + ((up + add) * go2).sum()).backward()

    def pad(t):                                            # NCHW fp64 -> NHWC fp32 with channel stride LD
        out = torch.zeros(B, t.shape[2], t.shape[3], LD)
        out[..., :C] = t.permute(0, 2, 3, 1).float()
        return out.to(dev)
    d_o, d_o2 = pad(go), pad(go2)
    d_src = torch.zeros(B, Hs, Hs, LD, device=dev)
    d_add = torch.ones(B, Hd, Hd, LD, device=dev)          # holds another contribution already: accumulated into
    _lib.check(_lib.lib.gssd_interp_add_bwd_f32(d_o.data_ptr(), d_o2.data_ptr(), d_src.data_ptr(), d_add.data_ptr(), B, Hs, Hs, Hd, Hd, C,
                                                LD, torch.cuda.current_stream().cuda_stream))
    assert rel(d_src[..., :C].permute(0, 3, 1, 2).cpu(), src.grad) < 2e-6
    assert rel(d_add[..., :C].permute(0, 3, 1, 2).cpu() - 1.0, add.grad) < 2e-6
    assert float(d_src[..., C:].abs().max()) == 0.0 and float((d_add[..., C:] - 1.0).abs().max()) == 0.0


def _loss_inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    o1, o2 = torch.randn(B, 2, 75, 75, generator=g), torch.randn(B, 16, 75, 75, generator=g)
    pix = (torch.rand(B, 75, 75, generator=g) < 0.08).long()
    neg = ((torch.rand(B, 75, 75, generator=g) < 0.9) & (pix == 0)).to(torch.uint8)
    posw = torch.rand(B, 75, 75, generator=g) * pix.float()
    link = (torch.rand(B, 8, 75, 75, generator=g) < 0.5).long() * pix[:, None]
    return o1, o2, pix, neg, posw, link


def test_loss_backward_vs_oracle_autograd(dev):
    """d(PixelLinkLoss) on the HIP kernel vs autograd through the oracle restatement (the mined mask, areas and link weight sums are
    constants in both), for a weighted sum of the four returned means; both calling patterns of the reference driver."""
    from pixel_link.criterion import PixelLinkLoss
    o1, o2, pix, neg, posw, link = _loss_inputs(3, 77)
    wts = (1.0, 0.7, 2.0, 0.3)
    a1, a2 = o1.clone().requires_grad_(), o2.clone().requires_grad_()
    terms = PO.pixel_link_loss(a1, a2, pix, neg, posw, link, as_tensors=True)[:4]
    sum(w * t for w, t in zip(wts, terms)).backward()
    for fused in (True, False):
        h1, h2 = o1.to(dev).requires_grad_(), o2.to(dev).requires_grad_()
        crit = PixelLinkLoss()
        pp, pn = crit.pixel_loss(h1, pix.to(dev), neg.to(dev), posw.to(dev), link=(h2, link.to(dev)) if fused else None)
        lp, ln = crit.link_loss(h2, link.to(dev))
        (wts[0] * pp + wts[1] * pn + wts[2] * lp + wts[3] * ln).backward()
        assert rel([float(pp), float(pn), float(lp), float(ln)], [float(t) for t in terms]) < 1e-5
        assert rel(h1.grad.cpu(), a1.grad) < 1e-5 and rel(h2.grad.cpu(), a2.grad) < 1e-5


@pytest.mark.parametrize('tag', ['full', 'plain'])
def test_backward_gradients_vs_oracle(dev, tag):
    """Training step of the PixelLink++ row: HIP forward -> PixelLinkLoss -> loss.backward() (HIP loss backward + HIP backward plan)
    against CPU autograd through the oracle graph and the oracle loss, B = 2, relative L2 error per parameter tensor.  The bounds are
    those of the detector's gradient test (tests/test_gpu_training.py::test_backward_gradients): ReLU / max-pool decisions at |z| ~ 1e-6
    flip between two fp32 implementations and move single entries by ~1 %; parameters with no decision downstream match to 1e-4."""
    from pixel_link.criterion import PixelLinkLoss
    kw = FULL if tag == 'full' else PLAIN
    net = build(kw)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    B = 2
    x = synth.synth_images(B, seed=311)
    _, _, pix, neg, posw, link = _loss_inputs(B, 5)
    wts = (1.0, 1.0, 0.5, 0.5)
    skip = ('running_mean', 'running_var', 'weight_u', 'weight_v', 'num_batches_tracked')
    sdg = {k: (v.clone().requires_grad_() if (v.is_floating_point() and not k.endswith(skip)) else v) for k, v in sd.items()}
    r1, r2, _ = PO.pixellink_forward(sdg, x, training=True, **kw)
    terms = PO.pixel_link_loss(r1, r2, pix, neg, posw, link, as_tensors=True)[:4]
    sum(w * t for w, t in zip(wts, terms)).backward()
    net = net.to(dev).train()
    o1, o2 = net(x.to(dev))
    crit = PixelLinkLoss()
    pp, pn = crit.pixel_loss(o1, pix.to(dev), neg.to(dev), posw.to(dev), link=(o2, link.to(dev)))
    lp, ln = crit.link_loss(o2, link.to(dev))
    (wts[0] * pp + wts[1] * pn + wts[2] * lp + wts[3] * ln).backward()
    assert rel([float(pp), float(pn), float(lp), float(ln)], [float(t) for t in terms]) < 1e-4
    named = dict(net.named_parameters(remove_duplicate=False))     # (modules_except_dcn aliases most modules: keep every name)

    def l2rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / max(float(b.norm()), 1e-30))
    keys = ['conv1_1.weight', 'conv2_2.weight', 'conv3_3.bias', 'conv4_2.weight', 'conv5_3.weight', 'conv6.weight', 'conv7.bias',
            'fuse2.weight', 'bn_fuse3.weight', 'bn_fuse5.bias', 'out2_1.weight', 'out3_2.weight', 'out5_2.bias', 'final_1.weight',
            'final_2.weight', 'final_2.bias']
    if tag == 'full':
        keys += ['self_attn_list.0.snconv1x1_theta.weight_orig', 'self_attn_base_list.0.sigma', 'self_attn_base_list.3.snconv1x1_g.bias',
                 'dcn_list.0.weight', 'dcn_list.0.conv_offset_mask.weight']
    missing = [k for k in keys if named[k].grad is None]
    assert not missing, missing
    errs = {k: l2rel(named[k].grad, sdg[k].grad) for k in keys}
    print(tag, 'PixelLink++ gradient L2-relative errors vs CPU autograd', {k: f'{v:.1e}' for k, v in errs.items()})
    assert errs['final_1.weight'] < 1e-4 and errs['final_2.weight'] < 1e-4 and errs['final_2.bias'] < 1e-4
    assert max(v for k, v in errs.items() if not k.endswith('sigma')) < 2e-2, errs
    assert all(v < 6e-2 for k, v in errs.items() if k.endswith('sigma')), errs
    # every parameter the oracle graph differentiates received a gradient, and nothing else did
    for k, p in named.items():
        if not k.startswith('modules_except_dcn.') and p.grad is None:
            assert sdg[k].grad is None or float(sdg[k].grad.abs().max()) == 0.0, k
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    # a second step through the same plan (buffers re-zeroed, gradients accumulate into .grad like autograd's)
    g0 = named['final_2.weight'].grad.clone()
    o1, o2 = net(x.to(dev))
    pp, pn = crit.pixel_loss(o1, pix.to(dev), neg.to(dev), posw.to(dev), link=(o2, link.to(dev)))
    lp, ln = crit.link_loss(o2, link.to(dev))
    (wts[0] * pp + wts[1] * pn + wts[2] * lp + wts[3] * ln).backward()
    # plain graph: the same step again, exactly twice the gradient (split-K atomics aside); full graph: the spectral-norm u / v moved one
    # power iteration on between the two forwards (sigma changes by several per cent on the synthetic weights)
    assert l2rel(named['final_2.weight'].grad, 2 * g0) < (5e-3 if tag == 'plain' else 0.3)


def test_training_steps_reduce_loss(dev):
    """Five SGD steps of the full PixelLink++ graph on one batch (HIP forward, loss, backward; torch.optim.SGD on the parameters):
    the loss falls, every parameter with a non-zero gradient moved, nothing is NaN."""
    from pixel_link.criterion import PixelLinkLoss
    net = build(FULL).to(dev).train()
    B = 2
    x = synth.synth_images(B, seed=312).to(dev)
    _, _, pix, neg, posw, link = _loss_inputs(B, 6)
    pix, neg, posw, link = pix.to(dev), neg.to(dev), posw.to(dev), link.to(dev)
    crit = PixelLinkLoss()
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    losses, opt = [], None
    for _ in range(5):
        if opt is not None:
            opt.zero_grad(set_to_none=True)
        o1, o2 = net(x)
        pp, pn = crit.pixel_loss(o1, pix, neg, posw, link=(o2, link))
        lp, ln = crit.link_loss(o2, link)
        loss = pp + pn + lp + ln
        loss.backward()
        if opt is None:
            # step size from the first gradient: a first-order decrease of 2 % of the loss per step (the synthetic weights put the
            # activations at 1e2 .. 1e4, a fixed learning rate would be a guess)
            g2 = sum(float((p.grad.double() ** 2).sum()) for p in net.parameters() if p.grad is not None)
            opt = torch.optim.SGD(net.parameters(), lr=0.02 * float(loss) / g2)
        opt.step()
        losses.append(float(loss))
    print('PixelLink++ training losses', [round(v, 4) for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    # every weight moved.  Biases whose gradient is zero by construction stay bit-identical (rounding noise times a 1e-7 step): a conv
    # bias in front of a train-mode BatchNorm (the fuse convs; in the side-branch Self_Attn blocks also the g and output-conv biases,
    # which reach the fuse BatchNorm as per-channel constants) and the phi bias of every Self_Attn block (it shifts all logits of a query
    # row alike); a sigma gate (one scalar, gradient = a cancelling sum over the map) may move by less than its ulp
    still = sorted(k for k, v in net.named_parameters() if torch.equal(v.detach(), before[k]))
    print('unmoved', still)
    assert all(k.endswith(('.bias', '.sigma')) for k in still), still
    assert sum(k.endswith('.sigma') for k in still) <= 2 and len(still) < 0.2 * len(before)


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
    # the box tail (pixel_link/box_geometry.py: min-area rectangles of the up-scaled components) on the REFERENCE's label maps gives the same boxes
    from pixel_link import box_geometry as G, pixel_link_config as cfg
    for b in range(lab.shape[0]):
        bx, sc = G.component_boxes(g['dec_labels'][b], score[b], (300, 300), cfg.min_height, cfg.min_area)
        assert [d[1:] for d in boxes[b]] == bx and np.allclose([d[0] for d in boxes[b]], sc, atol=1e-6)
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
