"""CPU oracle for the GSSD / GSSD++ detection hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the algorithm of the reference path so that the HIP
kernels can be checked against it.  Nothing in the product package
(``grouped-ssd-pytorch_amd/``) may import it; only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the baseline.

Pinning: the reference has no tests or golden vectors of its own, so the
oracle is pinned by fixtures generated from the *imported reference* in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); ``tests/test_oracle_golden.py``
checks every function here against them.  ONE exception, stated loudly:

    ** DCN (modulated deformable conv) -- PARITY UNPINNED. **  The arithmetic lives in the
    third-party CUDA extension ``dcn_v2`` (CharlesShang/DCNv2, version not pinned by the
    reference, not vendored: ``ssd_liverdet/layers/dcn_v2_custom.py:13``).  ``dcn_v2_conv``
    below restates that project's published algorithm (modulated_deformable_im2col + GEMM);
    it is anchored only by the reference's call site (``dcn_v2_custom.py:79-89``), the offset
    channel order documented in ``ssd_liverdet/utils/show_offset.py:28-32`` and two
    known-answer identities (zero offsets -> 0.5*conv+b; integer offsets -> shifted taps).

Index / integer work (priors, matching, mining ranks, NMS) is numpy float32 arithmetic
written in the reference's operation order so results are bit-comparable; the network
forward is a functional torch-CPU fp32 restatement (the path is floating point).
Every function cites the reference file:line it follows (paths relative to
``/root/reference/ssd_liverdet/``).
"""
from itertools import product
from math import sqrt

import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32

# ----------------------------------------------------------------------------------------
# constants: data/config.py:114-134 (v2) and :91-110 (v2_512)
# ----------------------------------------------------------------------------------------
V2 = {
    'feature_maps': [38, 19, 10, 5, 3, 1], 'min_dim': 300, 'steps': [8, 16, 32, 64, 100, 300],
    'min_sizes': [30, 60, 111, 162, 213, 264], 'max_sizes': [60, 111, 162, 213, 264, 315],
    'aspect_ratios': [[2], [2, 3], [2, 3], [2, 3], [2], [2]], 'variance': [0.1, 0.2],
    'clip': True, 'name': 'v2',
}
V2_512 = {
    'feature_maps': [64, 32, 16, 8, 4, 2, 1], 'min_dim': 512,
    'steps': [8, 16, 32, 64, 128, 256, 512],
    'min_sizes': [20, 51, 133, 215, 296, 378, 460], 'max_sizes': [51, 133, 215, 296, 378, 460, 542],
    'aspect_ratios': [[2], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]], 'variance': [0.1, 0.2],
    'clip': True, 'name': 'v2_512',
}
VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'C', 512, 512, 512, 'M', 512, 512, 512]
EXTRAS_CFG = [256, 'S', 512, 128, 'S', 256, 128, 256, 128, 256]
MBOX = [4, 6, 6, 6, 4, 4]


# ----------------------------------------------------------------------------------------
# PriorBox -- layers/functions/prior_box.py:32-56 (v2 / v2_512 branch) and :169-172
# ----------------------------------------------------------------------------------------
def prior_box(cfg=V2):
    mean = []
    for k, f in enumerate(cfg['feature_maps']):
        for i, j in product(range(f), repeat=2):
            f_k = cfg['min_dim'] / cfg['steps'][k]
            cx = (j + 0.5) / f_k
            cy = (i + 0.5) / f_k
            s_k = cfg['min_sizes'][k] / cfg['min_dim']
            mean += [cx, cy, s_k, s_k]
            s_k_prime = sqrt(s_k * (cfg['max_sizes'][k] / cfg['min_dim']))
            mean += [cx, cy, s_k_prime, s_k_prime]
            for ar in cfg['aspect_ratios'][k]:
                mean += [cx, cy, s_k * sqrt(ar), s_k / sqrt(ar)]
                mean += [cx, cy, s_k / sqrt(ar), s_k * sqrt(ar)]
    out = np.asarray(mean, dtype=np.float64).astype(f32).reshape(-1, 4)
    if cfg['clip']:
        out = np.clip(out, f32(0), f32(1))
    return out


# ----------------------------------------------------------------------------------------
# box_utils.py
# ----------------------------------------------------------------------------------------
def point_form(boxes):  # box_utils.py:4-13
    boxes = boxes.astype(f32)
    return np.concatenate([boxes[:, :2] - boxes[:, 2:] / f32(2),
                           boxes[:, :2] + boxes[:, 2:] / f32(2)], axis=1)


def intersect(box_a, box_b):  # box_utils.py:28-46
    max_xy = np.minimum(box_a[:, None, 2:], box_b[None, :, 2:])
    min_xy = np.maximum(box_a[:, None, :2], box_b[None, :, :2])
    inter = np.maximum(max_xy - min_xy, f32(0))
    return inter[:, :, 0] * inter[:, :, 1]


def jaccard(box_a, box_b):  # box_utils.py:49-67
    box_a = box_a.astype(f32)
    box_b = box_b.astype(f32)
    inter = intersect(box_a, box_b)
    area_a = ((box_a[:, 2] - box_a[:, 0]) * (box_a[:, 3] - box_a[:, 1]))[:, None]
    area_b = ((box_b[:, 2] - box_b[:, 0]) * (box_b[:, 3] - box_b[:, 1]))[None, :]
    union = area_a + area_b - inter
    with np.errstate(invalid='ignore', divide='ignore'):
        return inter / union


def encode(matched, priors, variances):  # box_utils.py:114-135
    g_cxcy = (matched[:, :2] + matched[:, 2:]) / f32(2) - priors[:, :2]
    g_cxcy = g_cxcy / (f32(variances[0]) * priors[:, 2:])
    g_wh = (matched[:, 2:] - matched[:, :2]) / priors[:, 2:]
    with np.errstate(invalid='ignore', divide='ignore'):
        g_wh = np.log(g_wh.astype(f32)) / f32(variances[1])
    return np.concatenate([g_cxcy, g_wh], axis=1).astype(f32)


def match(threshold, truths, priors, variances, labels):
    """box_utils.py:70-111.  Returns (loc_t[P,4] f32, conf_t[P] int64, best_truth_idx[P] int64)."""
    truths = np.asarray(truths, dtype=f32)
    priors = np.asarray(priors, dtype=f32)
    labels = np.asarray(labels, dtype=f32)
    overlaps = jaccard(truths, point_form(priors))          # [n_gt, P]
    best_prior_idx = overlaps.argmax(axis=1)                # first max on ties (:90)
    best_truth_overlap = overlaps.max(axis=0).copy()        # (:92)
    best_truth_idx = overlaps.argmax(axis=0).astype(np.int64)
    best_truth_overlap[best_prior_idx] = f32(2)             # index_fill_ (:97)
    for j in range(best_prior_idx.shape[0]):                # later GT wins (:100-101)
        best_truth_idx[best_prior_idx[j]] = j
    matches = truths[best_truth_idx]
    conf = (labels[best_truth_idx] + f32(1)).astype(np.int64)
    conf[best_truth_overlap < f32(threshold)] = 0           # strict < (:104)
    loc = encode(matches, priors, variances)
    return loc, conf, best_truth_idx


def decode(loc, priors, variances):  # box_utils.py:139-157
    """cx,cy / w,h arithmetic in fp32 in the reference's order.  exp() is evaluated in double
    and rounded once (correctly-rounded fp32) -- torch's SLEEF expf may differ by 1 ulp, which
    the golden tests bound; the HIP kernel uses the same double-then-round recipe."""
    loc = loc.astype(f32)
    priors = priors.astype(f32)
    cxcy = priors[:, :2] + (loc[:, :2] * f32(variances[0])) * priors[:, 2:]
    wh = priors[:, 2:] * np.exp((loc[:, 2:] * f32(variances[1])).astype(np.float64)).astype(f32)
    x1y1 = cxcy - wh / f32(2)
    x2y2 = wh + x1y1
    return np.concatenate([x1y1, x2y2], axis=1).astype(f32)


def nms(boxes, scores, overlap=0.5, top_k=200):
    """box_utils.py:174-238.  Returns keep[int64, len=count] -- indices into ``boxes`` in pick order.

    Candidates = the ``top_k`` highest scores, visited in descending score order (:196-198: ascending
    sort, last ``top_k``, pop from the end).  TIES: torch's CPU sort is not stable for long inputs, so
    the reference's order among exactly-equal scores is an accident of the sort kernel; with
    torch 2.10 it visits the LOWER index first (pinned by tests/golden/detect.npz, whose closed-form
    scores contain exact ties).  The oracle and the HIP kernel define ties that way."""
    boxes = boxes.astype(f32)
    scores = scores.astype(f32)
    if boxes.size == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1) * (y2 - y1)
    idx = np.argsort(-scores, kind='stable')[:top_k][::-1]      # ascending; ties: lower index last
    keep = []
    while idx.size > 0:
        i = idx[-1]
        keep.append(int(i))
        if idx.size == 1:
            break
        idx = idx[:-1]
        xx1 = np.maximum(x1[idx], x1[i])
        yy1 = np.maximum(y1[idx], y1[i])
        xx2 = np.minimum(x2[idx], x2[i])
        yy2 = np.minimum(y2[idx], y2[i])
        w = np.maximum(xx2 - xx1, f32(0))
        h = np.maximum(yy2 - yy1, f32(0))
        inter = w * h
        union = (area[idx] - inter) + area[i]
        with np.errstate(invalid='ignore', divide='ignore'):
            iou = inter / union
        idx = idx[iou <= f32(overlap)]
    return np.asarray(keep, dtype=np.int64)


def softmax_scores(conf):
    """models/ssd_multiphase_custom_group.py:388 -- softmax over the class axis, evaluated with
    double exp and one rounding (same recipe as the HIP Detect kernel)."""
    conf = conf.astype(f32)
    m = conf.max(axis=-1, keepdims=True)
    e = np.exp((conf - m).astype(np.float64)).astype(f32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=f32)).astype(f32)


def detect(num_classes, bkg_label, top_k, conf_thresh, nms_thresh, loc_data, conf_data, prior_data,
           variance=(0.1, 0.2), return_keep=False):
    """layers/functions/detection_pytorch_ver_1point5.py:32-89 (lines 85-88 act on a copy: no-op)."""
    if nms_thresh <= 0:
        raise ValueError('nms_threshold must be non negative.')
    num = loc_data.shape[0]
    out = np.zeros((num, num_classes, top_k, 5), dtype=f32)
    keeps = {}
    for i in range(num):
        boxes_all = decode(loc_data[i], prior_data, variance)
        for cl in range(1, num_classes):
            sc = conf_data[i, :, cl].astype(f32)
            mask = sc > f32(conf_thresh)
            if not mask.any():
                keeps[(i, cl)] = np.zeros((0,), dtype=np.int64)
                continue
            sel = np.nonzero(mask)[0]
            keep = nms(boxes_all[sel], sc[sel], nms_thresh, top_k)
            n = keep.shape[0]
            out[i, cl, :n, 0] = sc[sel][keep]
            out[i, cl, :n, 1:] = boxes_all[sel][keep]
            keeps[(i, cl)] = sel[keep]          # prior indices, in pick order
    return (out, keeps) if return_keep else out


# ----------------------------------------------------------------------------------------
# MultiBoxLoss -- layers/modules/multibox_loss.py:46-120
# ----------------------------------------------------------------------------------------
def multibox_loss(loc_data, conf_data, priors, targets, threshold=0.5, negpos_ratio=3,
                  variance=(0.1, 0.2), details=False):
    loc_data = np.asarray(loc_data, dtype=f32)
    conf_data = np.asarray(conf_data, dtype=f32)
    num, P = loc_data.shape[0], loc_data.shape[1]
    priors = np.asarray(priors, dtype=f32)[:P]                           # :60
    loc_t = np.zeros((num, P, 4), dtype=f32)
    conf_t = np.zeros((num, P), dtype=np.int64)
    for idx in range(num):                                               # :67-72
        t = np.asarray(targets[idx], dtype=f32)
        loc_t[idx], conf_t[idx], _ = match(threshold, t[:, :-1], priors, variance, t[:, -1])
    pos = conf_t > 0                                                     # :80
    d = (loc_data[pos] - loc_t[pos]).astype(f32)                         # smooth-L1, sum (:85-88)
    ad = np.abs(d)
    sl1 = np.where(ad < f32(1), f32(0.5) * d * d, ad - f32(0.5))
    loss_l = sl1.sum(dtype=np.float64)
    x_max = conf_data.max()                                              # global max (box_utils.py:167)
    lse = np.log(np.exp(conf_data - x_max).sum(axis=2, dtype=f32)) + x_max
    gathered = np.take_along_axis(conf_data, conf_t[:, :, None], axis=2)[:, :, 0]
    loss_c_all = (lse - gathered).astype(f32)                            # :93
    loss_c_all[pos] = 0                                                  # :98
    order = np.argsort(-loss_c_all, axis=1, kind='stable')               # :101 (descending, stable)
    rank = np.argsort(order, axis=1, kind='stable')                      # :102
    num_pos = pos.sum(axis=1, keepdims=True).astype(np.int64)
    num_neg = np.minimum(negpos_ratio * num_pos, P - 1)                  # :104
    neg = rank < num_neg                                                 # :105
    sel = pos | neg
    xs = conf_data[sel].astype(np.float64)                               # CE sum (:109-113)
    mx = xs.max(axis=1, keepdims=True)
    ce = (np.log(np.exp(xs - mx).sum(axis=1)) + mx[:, 0]) - xs[np.arange(xs.shape[0]), conf_t[sel]]
    loss_c = ce.sum()
    N = float(num_pos.sum())                                             # :117
    with np.errstate(invalid='ignore', divide='ignore'):
        ll, lc = np.float64(loss_l) / N, np.float64(loss_c) / N
    if details:
        return ll, lc, dict(loc_t=loc_t, conf_t=conf_t, pos=pos, neg=neg, loss_c_all=loss_c_all,
                            num_pos=num_pos[:, 0])
    return ll, lc


# ----------------------------------------------------------------------------------------
# operators of the model graph (torch CPU fp32)
# ----------------------------------------------------------------------------------------
def bf16_round(t):
    """Round-to-nearest-even to bfloat16 and back to fp32: the storage rounding of BASELINE.json configs[4] (bf16 weights /
    activations, fp32 accumulation).  The bf16 mode of this oracle is NOT a reference code path (the reference has no bf16
    mode): it restates the fp32 graph with a rounding at every point where the MI355X bf16 path stores a tensor in bf16."""
    return t.to(torch.bfloat16).to(torch.float32)


def bf16_round_ste(t):
    """bf16_round for gradient checks: the same forward value (in t's own dtype, so float64 graphs stay float64), identity backward
    (straight-through): the gradient of the bf16 storage mode is the fp32 graph's gradient evaluated at the rounded activations --
    autograd through ``.to(bfloat16)`` would round the GRADIENT at every storage point instead."""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def _ident(t):
    return t


def l2norm(x, weight, eps=1e-10):  # layers/modules/l2norm.py:19-23
    norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + eps
    return weight.view(1, -1, 1, 1) * (x / norm)


def spectral_weight(w_orig, u, v, training, eps=1e-12):
    """layers/spectral_norm.py:39-89.  Returns (W_sn, u_new, v_new); one power iteration when
    ``training`` (``:74-81``), none in eval (``:100-101``)."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    if training:
        # the power iteration runs under torch.no_grad() (:74-81): u and v are CONSTANTS of the graph, sigma = u^T W v is differentiated
        # through W only.  (Until round 5 this restatement let autograd see the iteration: forward values identical, but the float64
        # gradients of the spectral-normed weights that the gradient tests compare against carried extra terms of 1e-2 relative --
        # found by tests/test_gpu_grad.py, which holds every parameter tensor to a tight bound.)
        with torch.no_grad():
            wd = wm.detach()
            v = F.normalize(torch.mv(wd.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wd, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w_orig / sigma, u, v


def self_attn(x, sd, prefix, training, max_pool_factor=1, updates=None, q=None):
    """layers/self_attn.py:46-89.  Returns (out, sigma*attn_g, attn).  ``q`` (bf16 mode): the 1x1 weights are stored rounded
    (1/sigma is applied in fp32 afterwards), theta / phi and the logits stay fp32, g and the UNNORMALISED probabilities
    exp(s - max) are rounded (the value product runs on the bf16 matrix cores, the denominator sums the rounded probabilities),
    attn.g and both outputs are stored rounded."""
    B, ch, h, w = x.shape
    ws = {}
    for name in ('theta', 'phi', 'g', 'attn'):
        p = f'{prefix}.snconv1x1_{name}'
        W, u, v = spectral_weight(sd[p + '.weight_orig'], sd[p + '.weight_u'], sd[p + '.weight_v'], training)
        if updates is not None and training:
            updates[p + '.weight_u'], updates[p + '.weight_v'] = u, v
        if q is not None:
            wo = sd[p + '.weight_orig']
            sigma = torch.dot(u, torch.mv(wo.reshape(wo.shape[0], -1), v))     # from the fp32 master weight
            W = q(wo) / sigma
        ws[name] = (W, sd[p + '.bias'])
    pool = max(int(h // max_pool_factor), 1)
    theta = F.conv2d(x, *ws['theta']).view(B, ch // 8, h * w)
    phi = F.adaptive_avg_pool2d(F.conv2d(x, *ws['phi']), pool)
    phi = phi.view(B, ch // 8, -1)
    attn = torch.softmax(torch.bmm(theta.permute(0, 2, 1), phi), dim=-1)
    g = F.adaptive_avg_pool2d(F.conv2d(x, *ws['g']), pool).view(B, ch // 2, -1)
    if q is not None:
        s = torch.bmm(theta.permute(0, 2, 1), phi)
        p = q(torch.exp(s - s.max(dim=-1, keepdim=True).values))
        attn_g = q((torch.bmm(q(g), p.permute(0, 2, 1)) / p.sum(dim=-1).unsqueeze(1)).view(B, ch // 2, h, w))
    else:
        attn_g = torch.bmm(g, attn.permute(0, 2, 1)).view(B, ch // 2, h, w)
    attn_g = F.conv2d(attn_g, *ws['attn'])
    sig = sd[prefix + '.sigma']
    if q is not None:
        return q(x + sig * attn_g), q(sig * attn_g), attn
    return x + sig * attn_g, sig * attn_g, attn


def slice_and_cat(a, b, groups):  # models/ssd_multiphase_custom_group.py:185-192
    a = torch.split(a, a.size(1) // groups, dim=1)
    b = torch.split(b, b.size(1) // groups, dim=1)
    return torch.cat([torch.cat([a[i], b[i]], dim=1) for i in range(len(a))], dim=1)


def dcn_v2_conv(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1, deformable_groups=1, col_round=None, cells=None):
    """Modulated deformable convolution (DCNv2).  PARITY UNPINNED -- see the module docstring.

    Follows CharlesShang/DCNv2 ``modulated_deformable_im2col`` + GEMM as called from
    ``layers/dcn_v2_custom.py:84-89``: for output pixel (h, w), tap k = i*kw + j, deformable
    group d = c // (Cin/dg):  dy = offset[b, d*2*K + 2k], dx = offset[b, d*2*K + 2k + 1],
    m = mask[b, d*K + k];  sample at (h*s - p + i*dil + dy, w*s - p + j*dil + dx); the sample is
    0 unless -1 < y < H and -1 < x < W; bilinear with corners outside the map contributing 0.
    """
    B, Cin, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    K = kh * kw
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    dg = deformable_groups
    cpg = Cin // dg
    ys = (torch.arange(Ho, dtype=x.dtype) * stride - padding).view(1, 1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * stride - padding).view(1, 1, 1, Wo)
    off = offset.view(B, dg, K, 2, Ho, Wo)
    msk = mask.view(B, dg, K, Ho, Wo)
    xg = x.view(B, dg, cpg, H * W)
    cols = x.new_zeros(B, dg, cpg, K, Ho * Wo)
    for k in range(K):
        i, j = divmod(k, kw)
        py = ys + i * dilation + off[:, :, k, 0]            # [B, dg, Ho, Wo]
        px = xs + j * dilation + off[:, :, k, 1]
        valid = (py > -1) & (px > -1) & (py < H) & (px < W)
        y0 = torch.floor(py)
        x0 = torch.floor(px)
        if cells is not None:
            # externally imposed sampling cells (Decisions.dcn_cells: floor() and the in-range gate are the discontinuous decisions of the
            # bilinear sampler): the interpolation weights stay functions of THIS graph's offsets
            valid, y0, x0 = cells['valid'][k], cells['y0'][k].to(x.dtype), cells['x0'][k].to(x.dtype)
        ly, lx = py - y0, px - x0
        hy, hx = 1 - ly, 1 - lx
        val = x.new_zeros(B, dg, cpg, Ho * Wo)
        for (yy, xx, wgt) in ((y0, x0, hy * hx), (y0, x0 + 1, hy * lx),
                              (y0 + 1, x0, ly * hx), (y0 + 1, x0 + 1, ly * lx)):
            inside = valid & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
            lin = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().view(B, dg, 1, Ho * Wo)
            v = torch.gather(xg, 3, lin.expand(B, dg, cpg, Ho * Wo))
            val = val + v * (wgt * inside.to(x.dtype)).view(B, dg, 1, Ho * Wo)
        cols[:, :, :, k] = val * msk[:, :, k].reshape(B, dg, 1, Ho * Wo)
    cols = cols.view(B, Cin * K, Ho * Wo)                   # (c, k) flattened like weight.view
    if col_round is not None:                               # bf16 mode: the sampled, modulated columns are the MFMA's bf16 operand
        cols = col_round(cols)
    out = torch.matmul(weight.view(Cout, Cin * K), cols) + bias.view(1, Cout, 1)
    return out.view(B, Cout, Ho, Wo)


def dcn(x, sd, prefix, deformable_groups, q=None, decide=None):
    """layers/dcn_v2_custom.py:79-89: offset/mask conv -> chunk(3) -> cat(o1,o2), sigmoid(mask).  ``q`` (bf16 mode): rounded
    weights, fp32 offsets / mask, rounded sampled columns, rounded output."""
    qq = q or _ident
    om = F.conv2d(x, qq(sd[prefix + '.conv_offset_mask.weight']), sd[prefix + '.conv_offset_mask.bias'],
                  stride=1, padding=1)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    out = dcn_v2_conv(x, offset, torch.sigmoid(m), qq(sd[prefix + '.weight']), sd[prefix + '.bias'],
                      1, 1, 1, deformable_groups, col_round=q, cells=None if decide is None else decide.cells(prefix))
    return qq(out), offset


# ----------------------------------------------------------------------------------------
# the model graph -- models/ssd_multiphase_custom_group.py
# ----------------------------------------------------------------------------------------
def vgg_layers(batch_norm=True, groups=4, in_ch=12, fs=1):
    """Layer table of ``vgg()`` (:434-460): list of (kind, module_index, attrs)."""
    layers, idx, cin = [], 0, in_ch
    for v in VGG_CFG:
        if v == 'M' or v == 'C':
            layers.append(('pool', idx, dict(k=2, s=2, p=0, ceil=(v == 'C'))))
            idx += 1
        else:
            layers.append(('conv', idx, dict(cin=cin, cout=v * fs, k=3, s=1, p=1, d=1, groups=groups)))
            idx += 1
            if batch_norm:
                layers.append(('bn', idx, dict(c=v * fs)))
                idx += 1
            layers.append(('relu', idx, {}))
            idx += 1
            cin = v * fs
    layers.append(('pool', idx, dict(k=3, s=1, p=1, ceil=False)))
    idx += 1
    for (cout, k, p, d) in ((1024 * fs, 3, 6, 6), (1024 * fs, 1, 0, 1)):
        layers.append(('conv', idx, dict(cin=cin, cout=cout, k=k, s=1, p=p, d=d, groups=groups)))
        idx += 1
        if batch_norm:
            layers.append(('bn', idx, dict(c=cout)))
            idx += 1
        layers.append(('relu', idx, {}))
        idx += 1
        cin = cout
    return layers


def extras_layers(batch_norm=True, groups=4, in_ch=1024, fs=1):
    """Layer table of ``add_extras()`` (:463-490)."""
    layers, idx, cin, flag = [], 0, in_ch, False
    cfg = EXTRAS_CFG
    for k, v in enumerate(cfg):
        if cin != 'S':
            if v == 'S':
                layers.append(('conv', idx, dict(cin=cin, cout=cfg[k + 1] * fs, k=(1, 3)[flag], s=2, p=1, d=1,
                                                 groups=groups)))
            else:
                layers.append(('conv', idx, dict(cin=cin, cout=v * fs, k=(1, 3)[flag], s=1, p=0, d=1,
                                                 groups=groups)))
            idx += 1
            if batch_norm:
                layers.append(('bn', idx, dict(c=layers[-1][2]['cout'])))
                idx += 1
            flag = not flag
        cin = v if v == 'S' else v * fs
    return layers


BN_MOMENTUM = [0.1]     # nn.BatchNorm2d default; tests set 1.0 to make running stats == batch stats


def _bn(x, sd, prefix, training, updates, q=None):
    rm, rv = sd[prefix + '.running_mean'].clone(), sd[prefix + '.running_var'].clone()
    if q is None:
        y = F.batch_norm(x, rm, rv, sd[prefix + '.weight'], sd[prefix + '.bias'], training, BN_MOMENTUM[0], 1e-5)
    else:
        # bf16 mode: the statistics come from the fp32 conv output, the normalisation reads its bf16-stored copy
        if training:
            mean = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)
            n = x.numel() / x.shape[1]
            mom = BN_MOMENTUM[0]
            rm.mul_(1 - mom).add_(mom * mean)
            rv.mul_(1 - mom).add_(mom * var * (n / max(n - 1, 1)))
        else:
            mean, var = rm, rv
        scale = sd[prefix + '.weight'] / torch.sqrt(var + 1e-5)
        shift = sd[prefix + '.bias'] - mean * scale
        y = q(x) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if training and updates is not None:
        updates[prefix + '.running_mean'], updates[prefix + '.running_var'] = rm, rv
    return y


class Decisions:
    """Externally imposed ReLU masks and max-pool arg-maxes (tests/test_gpu_grad.py: the float64 graph evaluated with the decisions the HIP
    path took, so that its gradients can be compared tensor by tensor without the discontinuities).  ``relu_mask[site]``: bool [B,C,H,W];
    ``pool_idx[site]``: int64 flat indices as ``F.max_pool2d(..., return_indices=True)`` returns them; ``dcn_cells['dcn_list.<i>']``: the
    bilinear sampler's cells (see ``cells``).  Sites are named by the layer that
    takes the decision: 'vgg.<idx of the ReLU / pool>', 'extras.<idx of the BatchNorm>', 'bn_fuse_<nn>'.  Every site the graph reaches
    must be present (a missing one raises: no silent fallback to the graph's own decision)."""

    def __init__(self):
        self.relu_mask, self.pool_idx, self.dcn_cells, self.used = {}, {}, {}, set()

    def cells(self, site):
        """``dcn_cells[site]`` = dict(y0, x0, valid): lists over the 9 taps of [B, dg, H, W] tensors (floor of the sampling position, the
        in-range gate) of one deformable conv layer."""
        self.used.add(site)
        return self.dcn_cells[site]

    def relu(self, site, x):
        self.used.add(site)
        return x * self.relu_mask[site].to(x.dtype)

    def pool(self, site, x):
        self.used.add(site)
        idx = self.pool_idx[site]
        return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)


def _relu(x, site, decide):
    return F.relu(x) if decide is None else decide.relu(site, x)


def _run_table(x, table, sd, prefix, training, updates, taps=None, q=None, decide=None):
    for kind, idx, a in table:
        name = f'{prefix}.{idx}'
        if kind == 'conv':
            w = sd[name + '.weight'] if q is None else q(sd[name + '.weight'])
            x = F.conv2d(x, w, sd[name + '.bias'], a['s'], a['p'], a['d'], a['groups'])
        elif kind == 'bn':
            x = _bn(x, sd, name, training, updates, q)
        elif kind == 'relu':
            x = _relu(x, name, decide)
            if q is not None:
                x = q(x)                   # the activation is stored (or re-created by the consumer) in bf16
        elif kind == 'pool':
            x = F.max_pool2d(x, a['k'], a['s'], a['p'], ceil_mode=a['ceil']) if decide is None else decide.pool(name, x)
        if taps is not None:
            taps[name] = x
    return x


def gssd_forward(sd, x, num_classes=2, batch_norm=True, groups_vgg=4, groups_extra=4, use_fuseconv=True,
                 use_self_attention=False, use_self_attention_base=False, num_dcn_layers=0, groups_dcn=1,
                 dcn_cat_sab=False, max_pool_factor=1, training=True, taps=None, bf16=False, feature_scale=1, decide=None):
    """``SSD.forward`` train-phase return (:217-400): (loc[B,P,4], conf[B,P,C], updates).

    ``sd`` is a state dict with the reference's keys; ``updates`` holds the buffers a training
    forward mutates (BN running stats, spectral-norm u/v).  ``taps`` (dict) collects named
    intermediate activations for op-level parity tests.  ``decide`` (a Decisions object): take every ReLU / max-pool decision from it."""
    assert batch_norm or not bf16, 'bf16 storage mode is defined for the BatchNorm graph only'
    # BASELINE.json configs[4]: rounding at every bf16 storage point (see bf16_round); bf16='ste' = straight-through rounding for autograd
    q = (bf16_round_ste if bf16 == 'ste' else bf16_round) if bf16 else None
    qq = q or _ident
    x = qq(x)
    updates = {}
    tp = taps if taps is not None else {}
    vt = vgg_layers(batch_norm, groups_vgg, fs=feature_scale)
    split = 33 if batch_norm else 23                                        # :254-257
    sa_i = sab_i = 0
    sources = []
    x = _run_table(x, [l for l in vt if l[1] < split], sd, 'vgg', training, updates, taps, q, decide)
    if use_self_attention_base:                                             # :261-265
        x, attn_g, amap = self_attn(x, sd, f'self_attn_base_list.{sab_i}', training, max_pool_factor, updates, q)
        tp[f'sab{sab_i}.attn'] = amap
        sab_i += 1
        tp['sab0.out'], tp['sab0.attn_g'] = x, attn_g
    if dcn_cat_sab:                                                         # :267-271
        x = slice_and_cat(x, attn_g, groups_vgg)
    for i in range(num_dcn_layers):                                         # :273-278
        x, offset = dcn(x, sd, f'dcn_list.{i}', groups_dcn, q, decide)
        tp[f'dcn{i}.out'], tp[f'dcn{i}.offset'] = x, offset
    s = qq(l2norm(x, sd['L2Norm.weight']))                                  # :281
    tp['l2norm'] = s

    def branch(s, fuse):
        nonlocal sa_i
        if use_self_attention:
            s, _, amap = self_attn(s, sd, f'self_attn_list.{sa_i}', training, max_pool_factor, updates, q)
            tp[f'sa{sa_i}.attn'] = amap
            sa_i += 1
        if use_fuseconv:
            s = F.conv2d(s, qq(sd[f'fuse_{fuse}.weight']), sd[f'fuse_{fuse}.bias'])
            s = qq(_relu(_bn(s, sd, f'bn_fuse_{fuse}', training, updates, q), f'bn_fuse_{fuse}', decide)) if batch_norm else F.relu(s)   # :284-290
        return s
    sources.append(branch(s, '11'))                                         # :284-297
    x = _run_table(x, [l for l in vt if l[1] >= split], sd, 'vgg', training, updates, taps, q, decide)   # :300-301
    if use_self_attention_base:                                             # :303-307
        x, attn_g, amap = self_attn(x, sd, f'self_attn_base_list.{sab_i}', training, max_pool_factor, updates, q)
        tp[f'sab{sab_i}.attn'] = amap
        sab_i += 1
    sources.append(branch(x, '21'))                                         # :309-325
    et = extras_layers(batch_norm, groups_extra, 1024 * feature_scale, feature_scale)
    fuse_names = ['31', '41', '51', '61']
    conv_i = 0
    for kind, idx, a in et:                                                 # :329-372 (no BN: ReLU after every conv, a source
        x = _run_table(x, [(kind, idx, a)], sd, 'extras', training, updates, taps, q, decide)   # after every second one)
        if idx % 2 == 1 or not batch_norm:
            x = qq(_relu(x, f'extras.{idx}', decide))
        if (idx % 4 == 3) if batch_norm else (idx % 2 == 1):
            if use_self_attention_base:
                x, attn_g, amap = self_attn(x, sd, f'self_attn_base_list.{sab_i}', training, max_pool_factor,
                                            updates, q)
                tp[f'sab{sab_i}.attn'] = amap
                sab_i += 1
            sources.append(branch(x, fuse_names[conv_i]))
            conv_i += 1
    for i, s in enumerate(sources):
        tp[f'source{i}'] = s
    loc, conf = [], []
    for i, s in enumerate(sources):                                         # :375-380
        loc.append(F.conv2d(s, qq(sd[f'loc.{i}.weight']), sd[f'loc.{i}.bias'], padding=1).permute(0, 2, 3, 1))
        conf.append(F.conv2d(s, qq(sd[f'conf.{i}.weight']), sd[f'conf.{i}.bias'], padding=1).permute(0, 2, 3, 1))
    B = x.shape[0]
    loc = torch.cat([o.reshape(B, -1) for o in loc], 1).view(B, -1, 4)
    conf = torch.cat([o.reshape(B, -1) for o in conf], 1).view(B, -1, num_classes)
    return loc, conf, updates


def vanilla_ssd_forward(sd, x, num_classes=2):
    """models/ssd.py:48-108 train phase (BASELINE.json configs[0]: dense VGG-SSD300, 3-ch, no BN)."""
    table = [l for l in vgg_layers(False, 1, 3)]
    x = _run_table(x, [l for l in table if l[1] < 23], sd, 'vgg', True, None)
    sources = [l2norm(x, sd['L2Norm.weight'])]
    x = _run_table(x, [l for l in table if l[1] >= 23], sd, 'vgg', True, None)
    sources.append(x)
    for kind, idx, a in extras_layers(False, 1):
        x = F.relu(_run_table(x, [(kind, idx, a)], sd, 'extras', True, None))
        if idx % 2 == 1:
            sources.append(x)
    B = x.shape[0]
    loc = [F.conv2d(s, sd[f'loc.{i}.weight'], sd[f'loc.{i}.bias'], padding=1).permute(0, 2, 3, 1).reshape(B, -1)
           for i, s in enumerate(sources)]
    conf = [F.conv2d(s, sd[f'conf.{i}.weight'], sd[f'conf.{i}.bias'], padding=1).permute(0, 2, 3, 1).reshape(B, -1)
            for i, s in enumerate(sources)]
    return torch.cat(loc, 1).view(B, -1, 4), torch.cat(conf, 1).view(B, -1, num_classes)
