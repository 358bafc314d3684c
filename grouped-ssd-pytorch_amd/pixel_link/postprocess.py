"""Link decoding on the device (ssd_liverdet/pixel_link/postprocess.py).

``decode`` = the thresholds of ``mask_to_box`` (:104-121) + ``func`` (:178-234): labelled connected components per image, with their
pixel count, bounding box and mean positive-class score.  ``mask_to_box`` here returns per image ``[[score, x0, y0, x1, y1], ...]``
in image pixels from those component boxes.  The reference goes on through cv2 (nearest resize to 300 x 300, findContours,
minAreaRect, boxPoints; :124-160); cv2 is not part of this image, so that detour is not reproduced: the boxes are the components'
own axis-aligned bounds under the same nearest-neighbour 75 -> 300 up-scaling (a mask pixel covers a 4 x 4 block), the score is the
mean over the component's mask pixels, and the reference's min_area / min_height filters act on those boxes."""
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
    assert neighbors == 8
    B, _, H, W = pixel_mask.shape
    _, comps, ncomp = decode(pixel_mask, link_mask, pixel_thres)
    nmax = int(ncomp.max())
    if nmax > comps.shape[1]:
        # more components than the default statistics table holds (a noisy 75 x 75 map can have > 512): the reference's func()
        # handles every component, so decode again with the kernel's largest table and refuse to truncate beyond that
        if nmax > STAT_COMPS_MAX:
            raise _lib.GssdError(f'pixel_link.postprocess.mask_to_box: {nmax} components in one image; the device keeps statistics '
                                 f'for at most {STAT_COMPS_MAX} (the label map from decode() is complete, the box table is not)')
        _, comps, ncomp = decode(pixel_mask, link_mask, pixel_thres, max_components=STAT_COMPS_MAX)
    comps, ncomp = comps.cpu(), ncomp.cpu()
    sx, sy = img_shape[0] / W, img_shape[1] / H
    out = []
    for b in range(B):
        dets = []
        for c in comps[b, :min(int(ncomp[b]), comps.shape[1])].tolist():
            n, x0, y0, x1, y1, ssum = c
            bx0, by0 = int(x0 * sx), int(y0 * sy)
            bx1, by1 = min(int((x1 + 1) * sx) - 1, img_shape[0] - 1), min(int((y1 + 1) * sy) - 1, img_shape[1] - 1)
            w, h = bx1 - bx0 + 1, by1 - by0 + 1
            if min(w, h) < config.min_height or w * h < config.min_area:
                continue
            dets.append([ssum / n, bx0, by0, bx1, by1])
        out.append(dets)
    return out
