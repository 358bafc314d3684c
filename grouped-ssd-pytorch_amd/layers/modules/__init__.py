"""Loss-side modules of the drop-in package (names as the training script imports them)."""
from . import l2norm as _l2norm
from . import multibox_loss as _multibox_loss

L2Norm = _l2norm.L2Norm
MultiBoxLoss = _multibox_loss.MultiBoxLoss

__all__ = ('L2Norm', 'MultiBoxLoss')
