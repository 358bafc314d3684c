"""``PixelLinkLoss`` (drop-in for ssd_liverdet/pixel_link/criterion.py:5-104): one HIP launch per call pair, same return values.
``pixel_loss`` must be called before ``link_loss`` (the reference stores ``pos_pixel_weight`` on the object the same way, :32,68)."""
import torch

import pixel_link.pixel_link_config as config
from gssd import _lib

lib = _lib.lib


class PixelLinkLoss(object):
    def __init__(self):
        self.pos_pixel_weight = None
        self.neg_pixel_weight = None
        self.area = None
        self.neg_area = None
        self._pending = None

    def _run(self, out_1, out_2, target, neg_pixel_masks, pos_weight, link_target):
        if not out_1.is_cuda:
            raise _lib.GssdError('PixelLinkLoss: inputs must live on the MI355X; there is no CPU fallback')
        B, _, H, W = out_1.shape
        dev = out_1.device
        o1, o2 = out_1.detach().float().contiguous(), out_2.detach().float().contiguous()
        tgt = target.to(dev, torch.int64).contiguous()
        neg = (neg_pixel_masks.to(dev) == 1).to(torch.uint8).contiguous()
        pw = pos_weight.to(dev, torch.float32).contiguous()
        lt = link_target.to(dev, torch.int64).contiguous()
        res = torch.empty(B, 6, device=dev, dtype=torch.float64)
        nw = torch.empty(B, H, W, device=dev, dtype=torch.float32)
        _lib.check(lib.gssd_pixellink_loss_f32(o1.data_ptr(), o2.data_ptr(), tgt.data_ptr(), neg.data_ptr(), pw.data_ptr(), lt.data_ptr(),
                                               res.data_ptr(), nw.data_ptr(), B, H, W, int(config.neg_pos_ratio),
                                               torch.cuda.current_stream().cuda_stream))
        return res, nw

    def pixel_loss(self, input, target, neg_pixel_masks, pos_weight, link=None):
        """criterion.py:24-64 -> [mean pos term, mean neg term].  ``link`` = (out_2, link_masks) lets one launch serve the following
        ``link_loss`` call too; without it the link half runs on zeros and link_loss launches again."""
        B, _, H, W = input.shape
        if link is None:
            o2 = torch.zeros(B, 16, H, W, device=input.device)
            lt = torch.zeros(B, 8, H, W, device=input.device, dtype=torch.int64)
        else:
            o2, lt = link
        res, nw = self._run(input, o2, target, neg_pixel_masks, pos_weight, lt)
        self.pos_pixel_weight = pos_weight
        self.neg_pixel_weight = nw.to(torch.uint8)
        self.area, self.neg_area = res[:, 4].float(), res[:, 5].to(torch.int)
        self._pending = (input, target, neg_pixel_masks, res if link is not None else None)
        return [res[:, 0].mean().float(), res[:, 1].mean().float()]

    def link_loss(self, input, target, neighbors=8):
        """criterion.py:66-104 -> (mean pos-link term, mean neg-link term)."""
        assert input.size(1) == 16 and neighbors == 8
        if self._pending is None:
            raise _lib.GssdError('PixelLinkLoss.link_loss: call pixel_loss first (it sets pos_pixel_weight, criterion.py:32)')
        o1, tgt, neg, res = self._pending
        if res is None:
            res, _ = self._run(o1, input, tgt, neg, self.pos_pixel_weight, target)
        return res[:, 2].mean().float(), res[:, 3].mean().float()
