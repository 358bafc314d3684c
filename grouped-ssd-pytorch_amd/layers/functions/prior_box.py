"""PriorBox (drop-in for ssd_liverdet/layers/functions/prior_box.py:32-56,116-138,169-172).

Default boxes (cx, cy, w, h) for the 'v2' / 'v2_512' layouts, computed in float64 on the host
exactly like the reference's Python doubles, cast to fp32, then clamped to [0, 1].  One-off host
work at model construction; the result is bit-identical to the reference
(tests/golden/priors.npz)."""
from math import sqrt

import numpy as np
import torch


class PriorBox(object):
    def __init__(self, cfg):
        self.image_size = cfg['min_dim']
        self.num_priors = len(cfg['aspect_ratios'])
        self.variance = cfg['variance'] or [0.1]
        self.feature_maps = cfg['feature_maps']
        self.min_sizes = cfg['min_sizes']
        self.max_sizes = cfg['max_sizes']
        self.steps = cfg['steps']
        self.aspect_ratios = cfg['aspect_ratios']
        self.clip = cfg['clip']
        self.version = cfg['name']
        if self.version not in ('v2', 'v2_512'):
            raise NotImplementedError(f"prior layout '{self.version}' is not on the GSSD path")
        for v in self.variance:
            if v <= 0:
                raise ValueError('Variances must be greater than 0')

    def forward(self):
        chunks = []
        for k, f in enumerate(self.feature_maps):
            f_k = self.image_size / self.steps[k]
            s_k = self.min_sizes[k] / self.image_size
            s_kp = sqrt(s_k * (self.max_sizes[k] / self.image_size))
            wh = [(s_k, s_k), (s_kp, s_kp)]
            for ar in self.aspect_ratios[k]:
                wh += [(s_k * sqrt(ar), s_k / sqrt(ar)), (s_k / sqrt(ar), s_k * sqrt(ar))]
            wh = np.asarray(wh, dtype=np.float64)                       # [A, 2]
            c = (np.arange(f, dtype=np.float64) + 0.5) / f_k
            cy, cx = np.meshgrid(c, c, indexing='ij')                    # row i -> cy, col j -> cx
            A = wh.shape[0]
            boxes = np.empty((f, f, A, 4), dtype=np.float64)
            boxes[..., 0] = cx[:, :, None]
            boxes[..., 1] = cy[:, :, None]
            boxes[..., 2] = wh[None, None, :, 0]
            boxes[..., 3] = wh[None, None, :, 1]
            chunks.append(boxes.reshape(-1, 4))
        out = torch.from_numpy(np.concatenate(chunks, 0).astype(np.float32))
        if self.clip:
            out.clamp_(max=1, min=0)
        return out
