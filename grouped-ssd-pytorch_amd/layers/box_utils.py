"""box_utils (drop-in for ssd_liverdet/layers/box_utils.py).

``match`` and ``nms`` run the HIP kernels (the same ones MultiBoxLoss / Detect use); the one-line
coordinate helpers are plain tensor expressions kept for API compatibility -- they are not on the
product's hot path (the kernels do this arithmetic internally)."""
import torch

from gssd import ops


def point_form(boxes):      # box_utils.py:4-13
    return torch.cat((boxes[:, :2] - boxes[:, 2:] / 2, boxes[:, :2] + boxes[:, 2:] / 2), 1)


def center_size(boxes):     # box_utils.py:16-25
    return torch.cat(((boxes[:, 2:] + boxes[:, :2]) / 2, boxes[:, 2:] - boxes[:, :2]), 1)


def intersect(box_a, box_b):  # box_utils.py:28-46
    A, B = box_a.size(0), box_b.size(0)
    max_xy = torch.min(box_a[:, 2:].unsqueeze(1).expand(A, B, 2), box_b[:, 2:].unsqueeze(0).expand(A, B, 2))
    min_xy = torch.max(box_a[:, :2].unsqueeze(1).expand(A, B, 2), box_b[:, :2].unsqueeze(0).expand(A, B, 2))
    inter = torch.clamp((max_xy - min_xy), min=0)
    return inter[:, :, 0] * inter[:, :, 1]


def jaccard(box_a, box_b):    # box_utils.py:49-67
    inter = intersect(box_a, box_b)
    area_a = ((box_a[:, 2] - box_a[:, 0]) * (box_a[:, 3] - box_a[:, 1])).unsqueeze(1).expand_as(inter)
    area_b = ((box_b[:, 2] - box_b[:, 0]) * (box_b[:, 3] - box_b[:, 1])).unsqueeze(0).expand_as(inter)
    return inter / (area_a + area_b - inter)


def encode(matched, priors, variances):   # box_utils.py:114-135
    g_cxcy = (matched[:, :2] + matched[:, 2:]) / 2 - priors[:, :2]
    g_cxcy = g_cxcy / (variances[0] * priors[:, 2:])
    g_wh = torch.log((matched[:, 2:] - matched[:, :2]) / priors[:, 2:]) / variances[1]
    return torch.cat([g_cxcy, g_wh], 1)


def decode(loc, priors, variances):       # box_utils.py:139-157
    boxes = torch.cat((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:],
                       priors[:, 2:] * torch.exp(loc[:, 2:] * variances[1])), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def log_sum_exp(x):                       # box_utils.py:160-168
    x_max = x.data.max()
    return torch.log(torch.sum(torch.exp(x - x_max), 1, keepdim=True)) + x_max


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
