"""Synthetic study slices, lesion boxes and weights for tests and bench.py.

This is the generator SURVEY.md section 8(d) specifies: there is no dataset and no
checkpoint on the GPU box, so every run (tests, smoke, bench) draws its inputs
from numpy ``default_rng`` seeds.  Nothing here touches the GPU or the oracle; the
same arrays are handed to both sides of a parity check.

Layout notes (reference):
  * network input is ``[B, 12, 300, 300]`` fp32 in [0, 1]; channel = phase*3 + slice
    (``ssd_liverdet/train_lesion_multiphase_v2.py:198``)
  * targets are a python list of ``[n_i, 5]`` tensors ``(xmin, ymin, xmax, ymax, label)``
    normalised to [0, 1], label 0.0 = lesion (``ssd_liverdet/data/data_custom_v2.py:260-263``)
"""
import math

import numpy as np
import torch


def synth_images(batch, seed=0, size=300, channels=12):
    """``[B, C, S, S]`` fp32 study slices, min-max normalised per image to [0, 1]
    (what ``Normalize`` does, ``ssd_liverdet/utils/augmentations.py:518-524``)."""
    rng = np.random.default_rng(seed)
    x = rng.random((batch, channels, size, size), dtype=np.float32)
    # low-frequency structure so the maps are not pure white noise
    yy, xx = np.meshgrid(np.linspace(0, 1, size, dtype=np.float32),
                         np.linspace(0, 1, size, dtype=np.float32), indexing="ij")
    for b in range(batch):
        for c in range(channels):
            fx, fy, ph = rng.uniform(1, 6), rng.uniform(1, 6), rng.uniform(0, 6.28)
            x[b, c] = 0.5 * x[b, c] + 0.5 * (0.5 + 0.5 * np.sin(6.28 * (fx * xx + fy * yy) + ph))
        lo, hi = x[b].min(), x[b].max()
        x[b] = (x[b] - lo) / (hi - lo)
    return torch.from_numpy(x)


def synth_study_u8(seed, phases=4, size=512):
    """One raw study slice as the reference's loaders hand it to ``BaseTransform``: uint8 ``[phases, S, S, 3]`` (three
    adjacent z-slices per contrast phase).  Smooth structure + noise + a few saturated blobs (they exercise the clip of
    the resampler)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, size), np.linspace(0, 1, size), indexing='ij')
    out = np.empty((phases, size, size, 3), np.uint8)
    for p in range(phases):
        for c in range(3):
            f = 120 + 90 * np.sin(6.28 * (rng.uniform(1, 5) * xx + rng.uniform(1, 5) * yy) + rng.uniform(0, 6.28))
            f += rng.normal(0, 25, size=(size, size))
            f[(xx - rng.uniform(.2, .8)) ** 2 + (yy - rng.uniform(.2, .8)) ** 2 < 0.004] = 255
            f[(xx - rng.uniform(.2, .8)) ** 2 + (yy - rng.uniform(.2, .8)) ** 2 < 0.004] = 0
            out[p, :, :, c] = np.clip(f, 0, 255).astype(np.uint8)
    return out


def synth_detections(n_images, seed=0, size=512, top_k=200):
    """Synthetic Detect outputs for the evaluator: ``det[N, 2, top_k, 5]`` fp32 (class-1 rows = score, x1, y1, x2, y2
    normalised; descending score; zero rows after the last detection, like ``Detect``) and per-image ground truth
    ``[n_i, 4]`` float64 pixel boxes.  A mix of jittered true positives, duplicates and random false positives."""
    rng = np.random.default_rng(seed)
    det = np.zeros((n_images, 2, top_k, 5), np.float32)
    gts = []
    for n in range(n_images):
        k = int(rng.integers(0 if n % 9 == 8 else 1, 4))
        cxy = rng.uniform(0.2, 0.8, size=(k, 2))
        wh = rng.uniform(0.05, 0.3, size=(k, 2))
        g = np.concatenate([cxy - wh / 2, cxy + wh / 2], 1).clip(0, 1)
        gts.append(np.round(g * size).astype(np.float64))
        rows = []
        for j in range(k):
            for _ in range(int(rng.integers(0, 4))):                  # 0-3 jittered hits per lesion
                jit = g[j] + rng.normal(0, 0.03, 4)
                rows.append([rng.uniform(0.2, 1.0), *jit])
        for _ in range(int(rng.integers(0, 12))):                     # random false positives
            c, w = rng.uniform(0.1, 0.9, 2), rng.uniform(0.03, 0.3, 2)
            rows.append([rng.uniform(0.011, 0.7), *(c - w / 2), *(c + w / 2)])
        if rows:
            r = np.asarray(rows, np.float32)
            r = r[np.argsort(-r[:, 0], kind='stable')][:top_k]
            det[n, 1, :r.shape[0]] = r
    return det, gts


def synth_targets(batch, seed=0, max_boxes=3):
    """List of ``[n, 5]`` fp32 tensors: 1..max_boxes small lesion boxes per image."""
    rng = np.random.default_rng(seed + 7919)
    out = []
    for _ in range(batch):
        n = int(rng.integers(1, max_boxes + 1))
        cxy = rng.uniform(0.2, 0.8, size=(n, 2))
        wh = rng.uniform(0.05, 0.25, size=(n, 2))
        box = np.concatenate([cxy - wh / 2, cxy + wh / 2], axis=1).clip(0.0, 1.0)
        t = np.concatenate([box, np.zeros((n, 1))], axis=1).astype(np.float32)
        out.append(torch.from_numpy(t))
    return out


def synth_state_dict(shapes, seed=1111):
    """Fill a state dict by walking its keys in sorted order with one numpy generator.

    ``shapes`` maps state-dict key -> shape (or tensor).  Stable across machines because it
    does not depend on torch's RNG.  Every learnable quantity is randomised (biases, BN
    affine, sigma gates, DCN offset conv) so no branch of the path is trivially zero.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for key in sorted(shapes.keys()):
        shp = shapes[key]
        shp = tuple(shp.shape) if hasattr(shp, "shape") else tuple(shp)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[key] = torch.zeros(shp, dtype=torch.long)
            continue
        if leaf == "sigma":
            v = np.full(shp, 0.5, dtype=np.float32)
        elif leaf == "running_mean":
            v = rng.normal(0.0, 0.1, size=shp)
        elif leaf == "running_var":
            v = rng.uniform(0.5, 1.5, size=shp)
        elif leaf in ("weight_u", "weight_v"):
            v = rng.normal(0.0, 1.0, size=shp)
            v = v / max(np.linalg.norm(v), 1e-12)
        elif key == "L2Norm.weight":
            v = 20.0 + rng.normal(0.0, 0.5, size=shp)
        elif "conv_offset_mask" in key:
            v = rng.normal(0.0, 0.01 if leaf == "weight" else 0.05, size=shp)
        elif len(shp) == 4:  # conv weight / weight_orig, OIHW
            fan_in = shp[1] * shp[2] * shp[3]
            v = rng.normal(0.0, math.sqrt(2.0 / fan_in), size=shp)
        elif leaf == "weight" and len(shp) == 1:  # BatchNorm gamma
            v = rng.uniform(0.5, 1.5, size=shp)
        elif leaf == "bias":
            v = rng.normal(0.0, 0.05, size=shp)
        else:
            v = rng.normal(0.0, 0.1, size=shp)
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shp))
    # fuse_31..61 are registered twice (also as fuse_list1.N / bn_fuse_list1.N,
    # models/ssd_multiphase_custom_group.py:135-139): a real checkpoint holds the same values under
    # both names, so make the aliases agree
    for i, n in enumerate(('31', '41', '51', '61')):
        for alias, real in ((f'fuse_list1.{i}.', f'fuse_{n}.'), (f'bn_fuse_list1.{i}.', f'bn_fuse_{n}.')):
            for key in list(sd.keys()):
                if key.startswith(alias) and real + key[len(alias):] in sd:
                    sd[key] = sd[real + key[len(alias):]].clone()
    return sd
