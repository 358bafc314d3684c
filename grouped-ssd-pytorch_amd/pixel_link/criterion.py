"""``PixelLinkLoss`` (drop-in for ssd_liverdet/pixel_link/criterion.py:5-104): one HIP launch per call pair, same return values.
``pixel_loss`` must be called before ``link_loss`` (the reference stores ``pos_pixel_weight`` on the object the same way, :32,68).
With score maps that require grad the four returned means carry a grad_fn (``_LossFn``): their backward is one more HIP launch
(gssd_pixellink_loss_bwd_f32) -- the mined-negative mask, the areas and the link weight sums are constants of the step, as under the
reference's autograd (topk indices / comparisons carry no gradient)."""
import torch

import pixel_link.pixel_link_config as config
from gssd import _lib

lib = _lib.lib


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, crit, out_1, out_2, target, neg_pixel_masks, pos_weight, link_target):
        res, nw, saved = crit._run(out_1, out_2, target, neg_pixel_masks, pos_weight, link_target, keep=True)
        ctx.saved, ctx.res, ctx.nw = saved, res, nw
        ctx.shape = (tuple(out_1.shape), tuple(out_2.shape))
        crit._last = (res, nw)
        m = res[:, :4].mean(0).float()
        return m[0], m[1], m[2], m[3]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        o1, o2, tgt, pw, lt = ctx.saved
        res, nw = ctx.res, ctx.nw
        B, _, H, W = ctx.shape[0]
        dev = o1.device
        up = torch.stack([g if g is not None else torch.zeros((), device=dev) for g in (g0, g1, g2, g3)]).float().contiguous()
        d1, d2 = torch.empty(ctx.shape[0], device=dev), torch.empty(ctx.shape[1], device=dev)
        _lib.check(lib.gssd_pixellink_loss_bwd_f32(o1.data_ptr(), o2.data_ptr(), tgt.data_ptr(), nw.data_ptr(), pw.data_ptr(), lt.data_ptr(),
                                                   res.data_ptr(), up.data_ptr(), d1.data_ptr(), d2.data_ptr(), B, H, W,
                                                   torch.cuda.current_stream().cuda_stream))
        return None, d1, d2, None, None, None, None


class PixelLinkLoss(object):
    def __init__(self):
        self.pos_pixel_weight = None
        self.neg_pixel_weight = None
        self.area = None
        self.neg_area = None
        self._pending = None

    def _run(self, out_1, out_2, target, neg_pixel_masks, pos_weight, link_target, keep=False):
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
        if keep:
            return res, nw, (o1, o2, tgt, pw, lt)
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
        diff = torch.is_grad_enabled() and (input.requires_grad or (link is not None and o2.requires_grad))
        if diff:
            p_pos, p_neg, l_pos, l_neg = _LossFn.apply(self, input, o2, target, neg_pixel_masks, pos_weight, lt)
            res, nw = self._last
        else:
            res, nw = self._run(input, o2, target, neg_pixel_masks, pos_weight, lt)
            p_pos, p_neg = res[:, 0].mean().float(), res[:, 1].mean().float()
            l_pos, l_neg = res[:, 2].mean().float(), res[:, 3].mean().float()
        self.pos_pixel_weight = pos_weight
        self.neg_pixel_weight = nw.to(torch.uint8)
        self.area, self.neg_area = res[:, 4].float(), res[:, 5].to(torch.int)
        self._pending = (input, target, neg_pixel_masks, (l_pos, l_neg, o2) if link is not None else None)
        return [p_pos, p_neg]

    def link_loss(self, input, target, neighbors=8):
        """criterion.py:66-104 -> (mean pos-link term, mean neg-link term)."""
        assert input.size(1) == 16 and neighbors == 8
        if self._pending is None:
            raise _lib.GssdError('PixelLinkLoss.link_loss: call pixel_loss first (it sets pos_pixel_weight, criterion.py:32)')
        o1, tgt, neg, done = self._pending
        if done is not None and done[2] is input:
            return done[0], done[1]
        # no (or another) link tensor was given to pixel_loss: one more launch; the pixel half of its result is not used (its upstream
        # gradient is zero in the backward)
        if torch.is_grad_enabled() and input.requires_grad:
            _, _, l_pos, l_neg = _LossFn.apply(self, o1.detach(), input, tgt, neg, self.pos_pixel_weight, target)
            return l_pos, l_neg
        res, _ = self._run(o1, input, tgt, neg, self.pos_pixel_weight, target)
        return res[:, 2].mean().float(), res[:, 3].mean().float()
