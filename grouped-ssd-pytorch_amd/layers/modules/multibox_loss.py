"""MultiBoxLoss (drop-in for ssd_liverdet/layers/modules/multibox_loss.py:8-120).

Same constructor and ``forward(predictions, targets)`` contract: ``predictions`` is the train-phase
tuple ``(loc [B,P,4], conf [B,P,C], priors [>=P,4])``, ``targets`` a python list of ``[n_i,5]`` tensors;
returns ``(loss_l, loss_c)`` 0-dim tensors with gradients to ``loc`` / ``conf``.

The reference's per-image Python matching loop with a D2H copy per image (:67-75) and the two full
sorts of [B, 8732] (:101-102) become three HIP launches for the whole batch (gssd_match_batch,
gssd_hnm_loss, gssd_loss_finalize); nothing synchronises with the host."""
import torch
import torch.nn as nn

from gssd import ops
from data.config import v2 as cfg


class _MultiBoxLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc, conf, priors, tg, n_gt, threshold, negpos_ratio, variance, global_n=False):
        st = ops.multibox_loss_forward(loc.detach(), conf.detach(), priors, tg, n_gt, threshold, negpos_ratio,
                                       variance, global_n=global_n)
        ctx.st = st
        losses = st['losses']
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_l, g_c):
        dloc, dconf = ops.multibox_loss_backward(ctx.st, g_l, g_c)
        return dloc, dconf, None, None, None, None, None, None, None


class MultiBoxLoss(nn.Module):
    def __init__(self, num_classes, overlap_thresh, prior_for_matching, bkg_label, neg_mining, neg_pos, neg_overlap,
                 encode_target, use_gpu=True):
        super().__init__()
        self.use_gpu = use_gpu
        self.num_classes = num_classes
        self.threshold = overlap_thresh
        self.background_label = bkg_label
        self.encode_target = encode_target
        self.use_prior_for_matching = prior_for_matching
        self.do_neg_mining = neg_mining
        self.negpos_ratio = neg_pos
        self.neg_overlap = neg_overlap
        self.variance = cfg['variance']
        # extension (not in the reference): one process per GPU keeps the reference's per-replica normaliser N by default; True
        # all-reduces N over the ranks so that the data-parallel mean of the losses / gradients equals the single (world x B)-image batch
        self.global_normalizer = False

    def forward(self, predictions, targets):
        loc_data, conf_data, priors = predictions
        P = loc_data.size(1)
        priors = priors[:P, :]                       # DataParallel concatenates priors (:60)
        if priors.device != loc_data.device:
            priors = priors.to(loc_data.device)
        tg, n_gt = ops.pack_targets(targets, loc_data.device)
        return _MultiBoxLossFn.apply(loc_data, conf_data, priors.detach().contiguous(), tg, n_gt, float(self.threshold),
                                     int(self.negpos_ratio), (float(self.variance[0]), float(self.variance[1])),
                                     bool(self.global_normalizer))
