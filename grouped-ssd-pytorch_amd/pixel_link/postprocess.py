"""Link decoding on the device (ssd_liverdet/pixel_link/postprocess.py).

``decode`` = the thresholds of ``mask_to_box`` (:104-121) + ``func`` (:178-234): labelled connected components per image, with their
pixel count, bounding box and mean positive-class score.  ``mask_to_box`` returns per image ``[[score, x0, y0, x1, y1], ...]``
in image pixels.  The reference goes on through cv2 (nearest resize to 300 x 300, findContours, minAreaRect, boxPoints; :124-160); cv2 is not
part of this image, so that tail is DEFINED in pixel_link/box_geometry.py (round 6): minimum-area enclosing rectangle of each component by
convex hull + rotating calipers, the reference's filters, integer corners, axis-aligned bounds, bilinear score map -- pinned by closed-form
cases and a brute-force cross-check (tests/test_pixellink_cpu.py), not by OpenCV outputs (there are none to pin it to)."""
import torch

import pixel_link.pixel_link_config as config
from gssd import _lib

lib = _lib.lib
STAT_COMPS_MAX = 1024        # csrc/pixellink.hip PL_STAT_COMPS: components with statistics (the label map itself is unlimited)


def decode(pixel_mask, link_mask, pixel_thres=None, link_thres=None, max_components=512):
    """-> (labels int32 [B,H,W], comps fp32 [B,max_components,6] = count, min x, min y, max x, max y, score sum; ncomp int32 [B])."""
    if not pixel_mask.is_cuda:
        raise _lib.GssdError('pixel_link.postprocess: inputs must live on the MI355X; there is no CPU fallback')
    B, _, H, W = pixel_mask.shape
    dev = pixel_mask.device
    o1, o2 = pixel_mask.detach().float().contiguous(), link_mask.detach().float().contiguous()
    labels = torch.empty(B, H, W, device=dev, dtype=torch.int32)
    comps = torch.empty(B, max_components, 6, device=dev, dtype=torch.float32)
    ncomp = torch.empty(B, device=dev, dtype=torch.int32)
    _lib.check(lib.gssd_pixellink_decode_f32(o1.data_ptr(), o2.data_ptr(), labels.data_ptr(), comps.data_ptr(), ncomp.data_ptr(), B, H, W,
                                             float(config.pixel_conf_threshold if pixel_thres is None else pixel_thres),
                                             float(config.link_conf_threshold if link_thres is None else link_thres),
                                             max_components, torch.cuda.current_stream().cuda_stream))
    return labels, comps, ncomp


def mask_to_box(pixel_mask, link_mask, neighbors=8, img_shape=(300, 300), pixel_thres=None):
    """postprocess.py:84-161: per image ``[[score, min_x, min_y, max_x, max_y], ...]``.  Components on the device (``decode``), then the
    reference's host tail as pixel_link/box_geometry.py defines it: nearest-neighbour up-scaling of the label map, the minimum-area enclosing
    rectangle of each component, its min_height / min_area filters, integer corners clamped into the image and their axis-aligned bounds; the score
    is the mean of the bilinearly up-scaled positive-class probability over the component's pixels."""
    assert neighbors == 8
    import numpy as np
    from pixel_link import box_geometry as G
    labels, _, _ = decode(pixel_mask, link_mask, pixel_thres)
    lab = labels.cpu().numpy()
    logit = pixel_mask.detach().float().cpu().numpy()
    prob = (1.0 / (1.0 + np.exp((logit[:, 0] - logit[:, 1]).astype(np.float64)))).astype(np.float32)       # Softmax2d, positive class
    out = []
    for b in range(lab.shape[0]):
        boxes, scores = G.component_boxes(lab[b], prob[b], (img_shape[1], img_shape[0]), config.min_height, config.min_area)
        out.append([[s] + bx for s, bx in zip(scores, boxes)])
    return out
