from .l2norm import L2Norm
from .multibox_loss import MultiBoxLoss

__all__ = ['L2Norm', 'MultiBoxLoss']
