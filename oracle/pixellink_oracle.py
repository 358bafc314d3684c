"""CPU oracle for the PixelLink++ row.  TEST INFRASTRUCTURE ONLY -- same rules as gssd_oracle.py: nothing in
the product package may import it.

Restates, with torch-CPU fp32 / numpy, ``ssd_liverdet/pixel_link``:
  * ``pixellink_forward``  -- ``PixelLink.forward`` (model.py:189-412) for ``config.version == "4s"`` (pixel_link_config.py:1), built
    from the pinned operators of gssd_oracle (Self_Attn, slice_and_cat, BatchNorm) and its DCN restatement (PARITY UNPINNED there);
  * ``pixel_link_loss``    -- ``PixelLinkLoss.pixel_loss`` + ``link_loss`` (criterion.py:24-104);
  * ``decode_links``       -- thresholds of ``mask_to_box`` (postprocess.py:104-121) + ``func`` (:178-234): labelled components.

Pinning (tests/golden/make_pixellink_golden.py -> tests/golden/pixellink.npz, checked by tests/test_oracle_golden.py): the forward
against the imported reference model (cv2 = empty import stub, not on a numeric path; dcn_v2 = the oracle's DCN restatement, so the
graph wiring is pinned and the DCN arithmetic is not), the loss against the imported ``criterion.PixelLinkLoss``, the decoding against
the imported ``postprocess.func``.  The cv2 half of the post-process (resize / findContours / minAreaRect / boxPoints,
postprocess.py:124-160) is NOT restated: cv2 is not in this image, so nothing could pin it -- the product emits the labelled
components, their bounding boxes and mean scores and stops there.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import gssd_oracle as O

GROUPS = 4          # pixel_link_config.py:36 vgg_groups
NEG_POS_RATIO = 3   # pixel_link_config.py:22
PIXEL_THR, LINK_THR = 0.2, 0.8   # pixel_link_config.py:27-28

# (name, cin, cout, dilation) per stage; pools between stages (model.py:35-71)
_STAGES = [
    [('conv1_1', 12, 64, 1), ('conv1_2', 64, 64, 1)],
    [('conv2_1', 64, 128, 1), ('conv2_2', 128, 128, 1)],
    [('conv3_1', 128, 256, 1), ('conv3_2', 256, 256, 1), ('conv3_3', 256, 256, 1)],
    [('conv4_1', 256, 512, 1), ('conv4_2', 512, 512, 1), ('conv4_3', 512, 512, 1)],
    [('conv5_1', 512, 512, 1), ('conv5_2', 512, 512, 1), ('conv5_3', 512, 512, 1)],
]


def _conv_relu(x, sd, name, pad=1, dil=1, groups=GROUPS):
    return F.relu(F.conv2d(x, sd[name + '.weight'], sd[name + '.bias'], 1, pad, dil, groups))


def pixellink_forward(sd, x, cascade_fuse=True, use_fuseconv=True, batch_norm=True, use_self_attention=True,
                      use_self_attention_base=True, num_dcn_layers=1, groups_dcn=4, dcn_cat_sab=True, detach_sab=False,
                      max_pool_factor=1, training=True, taps=None):
    """model.py:189-412 (version "4s", dilation=True).  Returns (out_1 [B,2,75,75], out_2 [B,16,75,75], updates)."""
    updates = {}
    tp = taps if taps is not None else {}
    sab = [0]
    sa = [0]

    def sa_base(x):
        if not use_self_attention_base:
            return x, None
        out, ag, _ = O.self_attn(x, sd, f'self_attn_base_list.{sab[0]}', training, max_pool_factor, updates)
        sab[0] += 1
        return out, ag

    def stage_out(s, k):                                                   # [SA] -> fuse (+BN) -> the two 1x1 heads
        if use_self_attention:
            s, _, _ = O.self_attn(s, sd, f'self_attn_list.{sa[0]}', training, max_pool_factor, updates)
            sa[0] += 1
        if use_fuseconv:
            s = F.conv2d(s, sd[f'fuse{k}.weight'], sd[f'fuse{k}.bias'])
            if batch_norm:
                s = O._bn(s, sd, f'bn_fuse{k}', training, updates)
        tp[f's{k}'] = s
        return (F.conv2d(s, sd[f'out{k}_1.weight'], sd[f'out{k}_1.bias']),
                F.conv2d(s, sd[f'out{k}_2.weight'], sd[f'out{k}_2.bias']))

    for (n, _, _, _) in _STAGES[0]:
        x = _conv_relu(x, sd, n)
    x = F.max_pool2d(x, 2, ceil_mode=True)                                 # pool1
    for (n, _, _, _) in _STAGES[1]:
        x = _conv_relu(x, sd, n)
    x = F.max_pool2d(x, 2, ceil_mode=True)                                 # pool2 (applied at :236)
    for (n, _, _, _) in _STAGES[2]:
        x = _conv_relu(x, sd, n)
    x, attn_g = sa_base(x)                                                 # :238-241
    if num_dcn_layers > 0:                                                 # :242-249
        if dcn_cat_sab:
            x = O.slice_and_cat(x, attn_g.detach() if detach_sab else attn_g, GROUPS)
        for i in range(num_dcn_layers):
            x, off = O.dcn(x, sd, f'dcn_list.{i}', groups_dcn)
            tp[f'dcn{i}.out'], tp[f'dcn{i}.offset'] = x, off
    l2 = stage_out(x, 2)
    x = F.max_pool2d(x, 2, ceil_mode=True)                                 # pool3
    for (n, _, _, _) in _STAGES[3]:
        x = _conv_relu(x, sd, n)
    x, _ = sa_base(x)
    l3 = stage_out(x, 3)
    x = F.max_pool2d(x, 2, ceil_mode=True)                                 # pool4
    for (n, _, _, _) in _STAGES[4]:
        x = _conv_relu(x, sd, n)
    x, _ = sa_base(x)
    l4 = stage_out(x, 4)
    x = F.max_pool2d(x, 3, 1, 1, ceil_mode=True)                           # pool5 (:62)
    x = _conv_relu(x, sd, 'conv6', pad=6, dil=6)
    x = _conv_relu(x, sd, 'conv7', pad=0)
    x, _ = sa_base(x)
    l5 = stage_out(x, 5)

    def up(t, like):
        return F.interpolate(t, size=like.shape[2:], mode='bilinear', align_corners=True)

    outs = []
    for j, fin in ((0, 'final_1'), (1, 'final_2')):
        u1 = up(l5[j] + l4[j], l3[j])                                      # :342 / :368
        u2 = up(u1 + l3[j], l2[j])                                         # :344
        logit = u2 + l2[j]                                                 # :355
        if cascade_fuse:
            feats = [up(l5[j], logit), up(l5[j] + l4[j], logit), up(u1 + l3[j], logit), logit]     # :356-359
            outs.append(F.conv2d(torch.cat(feats, 1), sd[fin + '.weight'], sd[fin + '.bias']))
        else:
            outs.append(F.conv2d(logit, sd[fin + '.weight'], sd[fin + '.bias']))                  # :399,411
    return outs[0], outs[1], updates


def pixel_link_loss(out_1, out_2, pixel_masks, neg_pixel_masks, pixel_pos_weights, link_masks, neg_pos_ratio=NEG_POS_RATIO,
                    as_tensors=False):
    """criterion.py:24-104.  Returns (pixel_pos, pixel_neg, link_pos, link_neg) as python floats + the mined-negative mask;
    ``as_tensors``: the four means as fp64 tensors that carry the autograd graph (the mined mask, areas and weight sums enter as
    constants, as in the reference: topk indices and comparisons have no gradient)."""
    out_1, out_2 = out_1.float(), out_2.float()
    B = out_1.shape[0]
    p0 = torch.softmax(out_1, dim=1)[:, 0].detach()
    ce = F.cross_entropy(out_1, pixel_masks, reduction='none')
    area = pixel_masks.view(B, -1).sum(1)
    neg_w = torch.zeros_like(pixel_pos_weights, dtype=torch.bool)
    neg_area = torch.zeros(B, dtype=torch.long)
    for i in range(B):
        cand = p0[i][neg_pixel_masks[i] == 1].view(-1)
        r = int(area[i]) * neg_pos_ratio
        if r == 0:
            r = 10000
        k = min(r, cand.numel())
        neg_area[i] = k
        thr = torch.sort(cand).values[k - 1]                               # = -topk(-cand, k)[-1]
        neg_w[i] = (p0[i] <= thr) & (neg_pixel_masks[i] == 1)
    den = area.double() + neg_area.double()
    pos = ((pixel_pos_weights * ce).view(B, -1).double().sum(1) / den).mean()
    neg = ((neg_w.float() * ce).view(B, -1).double().sum(1) / den).mean()
    lce = torch.stack([F.cross_entropy(out_2[:, 2 * n:2 * n + 2], link_masks[:, n], reduction='none') for n in range(8)], 1)
    w = pixel_pos_weights.unsqueeze(1).expand(-1, 8, -1, -1)
    wp, wn = (link_masks == 1).float() * w, (link_masks == 0).float() * w
    swp, swn = wp.view(B, -1).double().sum(1), wn.view(B, -1).double().sum(1)
    lp = torch.where(swp == 0, torch.zeros_like(swp), (wp * lce).view(B, -1).double().sum(1) / swp.clamp_min(1e-300))
    ln = torch.where(swn == 0, torch.zeros_like(swn), (wn * lce).view(B, -1).double().sum(1) / swn.clamp_min(1e-300))
    if as_tensors:
        return pos, neg, lp.mean(), ln.mean(), neg_w
    return float(pos), float(neg), float(lp.mean()), float(ln.mean()), neg_w


_NEIGH = [(-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1)]     # postprocess.py:166-176


def decode_links(out_1, out_2, pixel_thr=PIXEL_THR, link_thr=LINK_THR):
    """postprocess.py:104-121 + func (:178-234) for every image: int32 label maps [B,H,W] (0 = background; components numbered in
    raster order of their first pixel -- what root_map's insertion order yields)."""
    B, _, H, W = out_1.shape
    pix = torch.softmax(out_1.float(), 1)[:, 1] > pixel_thr
    links = torch.stack([torch.softmax(out_2[:, 2 * n:2 * n + 2].float(), 1)[:, 1] > link_thr for n in range(8)], 1) & pix[:, None]
    pix, links = pix.numpy(), links.numpy()
    res = np.zeros((B, H, W), np.int32)
    for b in range(B):
        parent = {}
        pts = list(zip(*np.where(pix[b])))
        for p in pts:
            parent[p] = -1

        def find(p):
            while parent[p] != -1:
                p = parent[p]
            return p
        for (y, x) in pts:
            for n, (dy, dx) in enumerate(_NEIGH):
                yy, xx = y + dy, x + dx
                if yy < 0 or xx < 0 or yy >= H or xx >= W:
                    continue
                if pix[b, yy, xx] and links[b, n, y, x]:
                    ra, rb = find((y, x)), find((yy, xx))
                    if ra != rb:
                        parent[rb] = ra
        ids = {}
        for p in pts:
            r = find(p)
            if r not in ids:
                ids[r] = len(ids) + 1
            res[b][p] = ids[r]
    return res
