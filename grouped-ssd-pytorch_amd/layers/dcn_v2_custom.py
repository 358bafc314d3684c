"""DCN (drop-in name for ssd_liverdet/layers/dcn_v2_custom.py:58-89)."""
from gssd.modules import DCN

__all__ = ['DCN']
