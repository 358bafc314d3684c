"""Detect (drop-in for ssd_liverdet/layers/functions/detection_pytorch_ver_1point5.py:32-89).

``Detect.apply(num_classes, bkg_label, top_k, conf_thresh, nms_thresh, loc_data, conf_data, prior_data)``
-> ``[B, num_classes, top_k, 5]`` rows ``(score, x1, y1, x2, y2)`` per class in NMS pick order, zero padded,
class 0 all zero.  One HIP launch for the whole batch (gssd_detect) instead of the reference's Python loop
over images x classes x up-to-200 NMS iterations."""
import torch
from torch.autograd import Function

from gssd import ops
from data.config import v2 as cfg


class Detect(Function):
    @staticmethod
    def forward(ctx, num_classes, bkg_label, top_k, conf_thresh, nms_thresh, loc_data, conf_data, prior_data,
                conf_is_logits=False):
        if nms_thresh <= 0:
            raise ValueError('nms_threshold must be non negative.')
        if bkg_label != 0:
            raise NotImplementedError('background label must be class 0')
        num = loc_data.size(0)
        num_priors = prior_data.size(0)
        conf = conf_data.reshape(num, num_priors, num_classes)
        out = ops.detect(loc_data.reshape(num, num_priors, 4), conf, prior_data, num_classes, top_k, conf_thresh,
                         nms_thresh, tuple(cfg['variance']), conf_is_logits)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, *grads):
        return (None,) * 9
