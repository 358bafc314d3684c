"""CPU oracle for the input stage -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the product
path (``gssd/input_stage.py`` -> HIP) never does.

What it restates
----------------
``base_transform_fast`` (ssd_liverdet/data/__init__.py:33-54) on a 4-phase study slice ``image[4, S, S, 3]`` uint8:
per phase ``Image.fromarray(image[idx]).resize((size, size))`` -> float32 -> ``-= mean`` -> (``use_normalize``) min-max
over the whole 4-phase array.  The training-side ``ResizeFast`` (utils/augmentations.py:506-516) is the same resize applied
to ``(image * 255).astype(uint8)`` followed by ``/ 255``; the driver then views ``[B, 4, 3, H, W]`` as ``[B, 12, H, W]``
(train_lesion_multiphase_v2.py:198; the dataset's ``permute(0, 3, 1, 2)``), i.e. channel ``c = phase * 3 + slice``.

The resize arithmetic is NOT in /root/reference: it lives in the third-party dependency **Pillow** (``PIL.Image.resize``;
the reference pins no version -- no requirements / lock file).  ``Image.resize`` defaults to BICUBIC from Pillow 7.0 on
(NEAREST before); this oracle follows Pillow >= 7 and takes the filter as a parameter.  The published algorithm
(src/libImaging/Resample.c: ``precompute_coeffs``, ``normalize_coeffs_8bpc``, ``ImagingResampleHorizontal_8bpc`` /
``Vertical_8bpc``) is restated below: separable, antialiased (filter support scaled by the down-scale factor), double
coefficients normalised per output pixel, quantised to 22-bit fixed point, integer accumulation starting from 1 << 21,
arithmetic shift, clip to [0, 255]; horizontal pass first, then vertical, uint8 in between.

Pinned: bit-exact against Pillow 12.2.0 in the build container (tests/golden/make_golden.py -> tests/golden/input.npz;
tests/test_oracle_golden.py::test_input_stage_golden).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


FILTERS = {'bilinear': (_bilinear, 1.0), 'bicubic': (_bicubic, 2.0)}


def resample_coeffs(in_size, out_size, filt='bicubic'):
    """Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the full-image box (in0 = 0, in1 = in_size).
    Returns (bounds int32 [out, 2] = (xmin, count), kk int32 [out, ksize], ksize)."""
    f, fsupport = FILTERS[filt]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img, bounds, kk, axis):
    """One 8bpc pass along ``axis`` of ``img[H, W, C]`` uint8."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for o in range(bounds.shape[0]):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.tensordot(kk[o, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_u8(img, size, filt='bicubic'):
    """``np.asarray(Image.fromarray(img).resize((size, size), filt))`` for ``img[H, W, C]`` uint8."""
    H, W = img.shape[:2]
    out = img
    if W != size:
        b, k, _ = resample_coeffs(W, size, filt)
        out = _pass(out, b, k, 1)                     # horizontal first (Resample.c ImagingResampleInner)
    if H != size:
        b, k, _ = resample_coeffs(H, size, filt)
        out = _pass(out, b, k, 0)
    return out


def base_transform(image, size, mean, use_normalize=False, filt='bicubic'):
    """data/__init__.py:33-54 for ``image[4, S, S, 3]`` uint8 -> float32 ``[4, size, size, 3]``."""
    mean = np.asarray(mean, np.float32)
    x = np.zeros((image.shape[0], size, size, image.shape[3]), np.float32)
    for idx in range(image.shape[0]):
        ph = pil_resize_u8(image[idx], size, filt).astype(np.float32)
        ph -= mean
        x[idx] = ph
    if use_normalize:
        lo, hi = x.min(), x.max()
        assert lo != hi, 'all-black image detected during Normalizing. check preprocessing'
        x = (x - lo) / (hi - lo)
    return x


def to_network_input(x):
    """``[4, H, W, 3]`` -> ``[12, H, W]`` with c = phase * 3 + slice (dataset permute + train...v2.py:198 view)."""
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2))).reshape(-1, x.shape[1], x.shape[2])
