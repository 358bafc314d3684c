"""Drop-in for the reference's evaluation entry point (ssd_liverdet/test_ap_iobb.py:231-328 ``test_net``, :10-41
``voc_ap``): same call, but the images are transformed, run through the test-phase network and scored in batches on the
MI355X (``gssd.input_stage`` -> HIP engine -> ``gssd.evaluator``).  ``visualize=True`` (:122, :157-179) runs the visualize plan
(``net(x, visualize=True)`` -> detections, DCN offsets, base / fusion attention maps) and writes the reference's numpy dumps per
image (``<idx>_x.npy``, ``_annotation.npy``, ``_all_offset.npy``, ``_all_fusion_attention.npy``, ``_all_base_attention.npy``); its two
cv2 JPEG overlays are not written (no cv2 in the image: nothing to pin them against).  The PixelLink branch is out of scope."""
import os

import numpy as np
import torch

from gssd.evaluator import DeviceEvaluator


def voc_ap(rec, prec, use_07_metric=True):
    """VOC AP from recall / precision arrays (11-point VOC07 rule, or area under the monotone precision envelope)."""
    rec, prec = np.asarray(rec, np.float64), np.asarray(prec, np.float64)
    if use_07_metric:
        total = 0.
        for thr in np.arange(0., 1.1, 0.1):
            hit = rec >= thr
            total = total + (np.max(prec[hit]) if hit.any() else 0) / 11.
        return total
    r = np.concatenate(([0.], rec, [1.]))
    p = np.concatenate(([0.], prec, [0.]))
    p = np.maximum.accumulate(p[::-1])[::-1]
    step = np.where(r[1:] != r[:-1])[0]
    return np.sum((r[step + 1] - r[step]) * p[step + 1])


def test_net(net, cuda, testset, transform, imsize=300, thresh=0.05, mode='v1', use_07_metric=True, ap_list=[0.5],
             iobb_list=[0.1], writer=None, iteration=None, visualize=False, output_path=None, model_name=None,
             use_pixel_link=False, batch_size=32):
    if use_pixel_link:
        raise NotImplementedError('PixelLink evaluation (mask_to_box) is not part of the hot path')
    vis_dir = None
    if visualize:
        assert output_path is not None and model_name is not None          # (test_ap_iobb.py:158)
        vis_dir = os.path.join(output_path, 'visualize', model_name, testset.name)
        os.makedirs(vis_dir, exist_ok=True)
    ev = DeviceEvaluator(thresh, ap_list, iobb_list, use_07_metric)
    n = len(testset)
    for start in range(0, n, batch_size):
        idxs = range(start, min(n, start + batch_size))
        imgs = [testset.pull_image(i) for i in idxs]
        annos = [np.asarray(testset.pull_anno(i)) for i in idxs]
        xs = [torch.as_tensor(transform(im)[0]) for im in imgs]                 # [4, s, s, 3] each (device or host)
        x = torch.stack(xs).cuda().float().permute(0, 1, 4, 2, 3)
        x = x.reshape(x.shape[0], -1, x.shape[3], x.shape[4]).contiguous()      # [B, 12, s, s]
        with torch.no_grad():
            if visualize:
                y, all_offset, all_attnb, all_attn = net(x, visualize=True)
            else:
                y = net(x)
        if visualize:
            for j, i in enumerate(idxs):                                         # the reference's per-image dumps (:163-179)
                np.save(os.path.join(vis_dir, f'{i}_x.npy'), x[j:j + 1].cpu().numpy())
                np.save(os.path.join(vis_dir, f'{i}_annotation.npy'), annos[j][None])
                np.save(os.path.join(vis_dir, f'{i}_all_offset.npy'), np.array([o[j:j + 1].cpu().numpy() for o in all_offset], dtype=object),
                        allow_pickle=True)
                np.save(os.path.join(vis_dir, f'{i}_all_fusion_attention.npy'),
                        {str(k): a[j:j + 1].cpu().numpy() for k, a in enumerate(all_attn)}, allow_pickle=True)
                np.save(os.path.join(vis_dir, f'{i}_all_base_attention.npy'),
                        {str(k): a[j:j + 1].cpu().numpy() for k, a in enumerate(all_attnb)}, allow_pickle=True)
        scales = [[im.shape[2], im.shape[1], im.shape[2], im.shape[1]] for im in imgs]
        if mode == 'v1':
            gts = [a[2:3, :-1] for a in annos]                                  # portal-phase box only (:206)
        else:
            gts = [a[:, :-1] for a in annos]
        ev.add_batch(y, scales, gts)
    return ev.result()
