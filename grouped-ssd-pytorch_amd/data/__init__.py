"""``data`` of the drop-in: the constants the hot path imports (``from data import v2``) and the deterministic input
transform (``BaseTransform`` / ``base_transform_fast``, ssd_liverdet/data/__init__.py:33-70) as a device stage.
The datasets, collate functions and random augmentations of the reference (ssd_liverdet/data/*.py,
utils/augmentations.py) are out of scope."""
import numpy as np
import torch

from .config import v2, v2_512

__all__ = ['v2', 'v2_512', 'BaseTransform', 'base_transform_fast']


def base_transform_fast(image, size, mean, use_normalize=False, p_only=False):
    """data/__init__.py:33-54 on the MI355X.  ``image``: one study slice ``[4, S, S, 3]`` uint8 (numpy or torch), or a batch
    ``[B, 4, S, S, 3]``.  Returns a CUDA fp32 tensor shaped like the reference's result (``[4, size, size, 3]``, or
    ``[B, 4, size, size, 3]``): a permuted view of the stage's ``[B, 12, size, size]`` network input."""
    from gssd.input_stage import DeviceInputStage
    if p_only:
        raise NotImplementedError('p_only (portal phase repeated 4x) is not part of the hot path')
    t = torch.as_tensor(np.ascontiguousarray(image)) if not isinstance(image, torch.Tensor) else image
    single = t.dim() == 4
    if single:
        t = t.unsqueeze(0)
    if not t.is_cuda:
        t = t.cuda()
    stage = _stage_cache.get((size, tuple(np.asarray(mean, np.float32).reshape(-1).tolist()), bool(use_normalize)))
    if stage is None:
        stage = DeviceInputStage(size, mean, use_normalize)
        _stage_cache[(size, tuple(np.asarray(mean, np.float32).reshape(-1).tolist()), bool(use_normalize))] = stage
    x = stage(t)                                               # [B, 12, size, size]
    if use_normalize:
        stage.check_not_flat()                                 # the reference's assert (data/__init__.py:49)
    B = x.shape[0]
    x = x.view(B, t.shape[1], 3, size, size).permute(0, 1, 3, 4, 2)
    return x[0] if single else x


_stage_cache = {}


class BaseTransform:
    """data/__init__.py:57-70: ``transform(image, boxes, labels) -> (x, boxes, labels)``."""

    def __init__(self, size, mean, use_normalize=False, p_only=False):
        self.size = size
        self.mean = np.array(mean, dtype=np.float32)
        self.use_normalize = use_normalize
        self.p_only = p_only

    def __call__(self, image, boxes=None, labels=None):
        return base_transform_fast(image, self.size, self.mean, self.use_normalize, self.p_only), boxes, labels
