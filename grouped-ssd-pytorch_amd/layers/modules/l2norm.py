"""L2Norm (drop-in for ssd_liverdet/layers/modules/l2norm.py)."""
from gssd.modules import L2Norm

__all__ = ['L2Norm']
