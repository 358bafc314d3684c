"""CPU oracle for the AP / IoBB evaluator -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.

Restates ssd_liverdet/test_ap_iobb.py: ``make_pred`` :126-148 (consumption of the Detect output: class-1 rows with
score > 0, coordinates scaled to pixels with an fp32 multiply, ``score > thresh`` filter, one global descending sort),
``test_net`` :255-328 (greedy TP / FP assignment in float64, cumulative precision / recall) and ``voc_ap`` :10-41.

One definition where the reference is not deterministic: ``np.argsort(-confidence)`` is an unstable sort, so detections
with exactly equal confidence are ordered arbitrarily there; here (and in the HIP evaluator) ties keep their original
order (image index, then Detect's row order).

Pinned: tests/golden/eval.npz holds the reference's own ``test_net`` results for seeded synthetic detections
(tests/golden/make_golden_eval.py; tests/test_oracle_golden.py::test_evaluator_golden).
"""
import numpy as np


def voc_ap(rec, prec, use_07_metric=True):  # test_ap_iobb.py:10-41
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def collect(det, scales, thresh):
    """make_pred :126-148 for a batch of Detect outputs ``det[N, 2, top_k, 5]`` (fp32) and per-image
    ``scales[N, 4]`` = (W, H, W, H) fp32.  Returns image_ids (int), confidence (float64), BB (float64 [nd, 4]) in the
    globally sorted order, plus the sort permutation over the flattened valid rows."""
    ids, conf, bb = [], [], []
    for n in range(det.shape[0]):
        d = det[n, 1]
        d = d[d[:, 0] > 0.]
        coords = (d[:, 1:] * scales[n].astype(np.float32)).astype(np.float32)
        keep = d[:, 0].astype(np.float64) > thresh
        ids.extend([n] * int(keep.sum()))
        conf.extend(d[keep, 0].astype(np.float64))
        bb.extend(coords[keep].astype(np.float64))
    conf = np.asarray(conf, np.float64)
    bb = np.asarray(bb, np.float64).reshape(-1, 4)
    order = np.argsort(-conf, kind='stable')
    return [ids[i] for i in order], conf[order], bb[order]


def evaluate(det, scales, gts, thresh=0.05, ap_list=(0.5,), iobb_list=(0.1,), use_07_metric=True, details=False):
    """test_net :231-328.  ``gts``: list of N float64 arrays [n_i, 4] in pixels (mode 'v2': every box counts)."""
    image_ids, conf, BB = collect(det, scales, thresh)
    npos = sum(g.shape[0] for g in gts)
    nd = len(image_ids)
    if nd == 0:
        return [0.] * len(ap_list), [0.] * len(iobb_list)
    thr = [('iou', t) for t in ap_list] + [('iobb', t) for t in iobb_list]
    tp = np.zeros((len(thr), nd))
    fp = np.zeros((len(thr), nd))
    flags = [[np.zeros(g.shape[0], bool) for g in gts] for _ in thr]
    for d in range(nd):
        G = gts[image_ids[d]].astype(float)
        bb = BB[d]
        if G.size == 0:
            continue                                     # the reference leaves both tp and fp at 0 here (:257-258)
        iw = np.maximum(np.minimum(G[:, 2], bb[2]) - np.maximum(G[:, 0], bb[0]), 0.)
        ih = np.maximum(np.minimum(G[:, 3], bb[3]) - np.maximum(G[:, 1], bb[1]), 0.)
        inters = iw * ih
        area = (bb[2] - bb[0]) * (bb[3] - bb[1])
        ov = {'iou': inters / (area + (G[:, 2] - G[:, 0]) * (G[:, 3] - G[:, 1]) - inters), 'iobb': inters / area}
        for m, (kind, t) in enumerate(thr):
            j = int(np.argmax(ov[kind]))
            if ov[kind][j] > t and not flags[m][image_ids[d]][j]:
                tp[m, d] = 1.
                flags[m][image_ids[d]][j] = True
            else:
                fp[m, d] = 1.
    out = []
    for m in range(len(thr)):
        f, t_ = np.cumsum(fp[m]), np.cumsum(tp[m])
        rec = t_ / float(npos)
        prec = t_ / np.maximum(t_ + f, np.finfo(np.float64).eps)
        out.append(voc_ap(rec, prec, use_07_metric))
    res = out[:len(ap_list)], out[len(ap_list):]
    return res + (dict(image_ids=image_ids, conf=conf, BB=BB, tp=tp, fp=fp, npos=npos),) if details else res
