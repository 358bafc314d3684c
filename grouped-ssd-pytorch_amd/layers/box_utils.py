"""box_utils (drop-in for ssd_liverdet/layers/box_utils.py).

``match`` and ``nms`` run the HIP kernels (the same ones MultiBoxLoss / Detect use); the one-line
coordinate helpers are plain tensor expressions kept for API compatibility -- they are not on the
product's hot path (the kernels do this arithmetic internally)."""
import torch

from gssd import ops


def _cxcywh(t):
    return t[..., 0], t[..., 1], t[..., 2], t[..., 3]


def point_form(boxes):
    """(cx, cy, w, h) -> (xmin, ymin, xmax, ymax)   [reference box_utils.py:4-13]"""
    cx, cy, w, h = _cxcywh(boxes)
    hw, hh = w / 2, h / 2
    return torch.stack((cx - hw, cy - hh, cx + hw, cy + hh), dim=-1)


def center_size(boxes):
    """(xmin, ymin, xmax, ymax) -> (cx, cy, w, h)   [reference box_utils.py:16-25]"""
    x1, y1, x2, y2 = _cxcywh(boxes)
    return torch.stack(((x2 + x1) / 2, (y2 + y1) / 2, x2 - x1, y2 - y1), dim=-1)


def intersect(box_a, box_b):
    """Pairwise intersection areas [A, B]   [reference box_utils.py:28-46]"""
    lo = torch.maximum(box_a[:, None, :2], box_b[None, :, :2])
    hi = torch.minimum(box_a[:, None, 2:], box_b[None, :, 2:])
    wh = (hi - lo).clamp_min(0)
    return wh[..., 0] * wh[..., 1]


def jaccard(box_a, box_b):
    """Pairwise IoU [A, B]   [reference box_utils.py:49-67]"""
    inter = intersect(box_a, box_b)
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])       # noqa: E731
    return inter / (area(box_a)[:, None] + area(box_b)[None, :] - inter)


def encode(matched, priors, variances):
    """Regression targets of matched ground truth w.r.t. priors   [reference box_utils.py:114-135]"""
    p_c, p_wh = priors[:, :2], priors[:, 2:]
    centre = ((matched[:, :2] + matched[:, 2:]) / 2 - p_c) / (variances[0] * p_wh)
    size = torch.log((matched[:, 2:] - matched[:, :2]) / p_wh) / variances[1]
    return torch.cat((centre, size), dim=1)


def decode(loc, priors, variances):
    """Boxes (x1, y1, x2, y2) from regression outputs   [reference box_utils.py:139-157]"""
    centre = priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:]
    size = priors[:, 2:] * torch.exp(loc[:, 2:] * variances[1])
    top_left = centre - size / 2
    return torch.cat((top_left, size + top_left), dim=1)


def log_sum_exp(x):
    """log sum_c exp(x) with the GLOBAL maximum as shift   [reference box_utils.py:160-168]"""
    shift = x.detach().max()
    return (x - shift).exp().sum(dim=1, keepdim=True).log() + shift


def match(threshold, truths, priors, variances, labels, loc_t, conf_t, idx):
    """box_utils.py:70-111 -- writes row ``idx`` of the caller's ``loc_t`` / ``conf_t`` in place."""
    dev = priors.device
    tg = torch.cat([truths.to(dev).float(), labels.to(dev).float().unsqueeze(1)], 1).contiguous()
    n = torch.tensor([0, truths.size(0)], dtype=torch.int32, device=dev)
    lt, ct = ops.match_batch(tg, n, priors.contiguous().float(), float(threshold),
                             (float(variances[0]), float(variances[1])))
    loc_t[idx] = lt[0].to(loc_t.device)
    conf_t[idx] = ct[0].to(conf_t.device)


def nms(boxes, scores, overlap=0.5, top_k=200):
    """box_utils.py:174-238 -- returns ``(keep, count)``; ``keep`` is zero padded to ``scores.size(0)``.
    Runs the Detect kernel with identity decoding (loc = 0, prior = box in centre form is NOT used: the
    boxes are passed through as priors with zero-size offsets)."""
    keep = scores.new_zeros(scores.size(0)).long()
    if boxes.numel() == 0:
        return keep
    n = boxes.size(0)
    # decode(loc=0, prior=(cx,cy,w,h)) reproduces (x1,y1,x2,y2) only up to rounding, so feed the kernel the boxes
    # through its own decode path with exact arithmetic: cx - w/2 etc. is not exact in fp32 -> use the dedicated
    # raw-box entry of ops.detect instead
    out, kidx, cnt = ops.detect_boxes(boxes.contiguous().float(), scores.contiguous().float(), float(overlap), top_k)
    c = int(cnt.item())
    keep[:c] = kidx[:c].long()
    return keep, c
