"""``data`` of the drop-in: only the constants the hot path imports (``from data import v2``).
The dataset / augmentation code of the reference (ssd_liverdet/data/*.py) is out of scope."""
from .config import v2, v2_512

__all__ = ['v2', 'v2_512']
