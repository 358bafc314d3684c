"""Inference-side operators of the drop-in package: prior generation and the decode + NMS op."""
from . import detection as _detection
from . import prior_box as _prior_box

Detect = _detection.Detect
PriorBox = _prior_box.PriorBox

__all__ = ('Detect', 'PriorBox')
