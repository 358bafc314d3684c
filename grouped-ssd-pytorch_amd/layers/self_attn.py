"""Self_Attn (drop-in name for ssd_liverdet/layers/self_attn.py:29-89)."""
from gssd.modules import Self_Attn, SNConv1x1

__all__ = ['Self_Attn', 'SNConv1x1']
