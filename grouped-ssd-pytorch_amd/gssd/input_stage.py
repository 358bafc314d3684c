"""Device-side input stage: raw uint8 4-phase study slices -> the network's ``[B, 12, S, S]`` fp32 input.

Mirrors ``base_transform_fast`` / ``BaseTransform`` (ssd_liverdet/data/__init__.py:33-70): per phase
``Image.fromarray(u8).resize((size, size))`` (Pillow, 8-bit fixed-point resampler), ``-= mean``, optional min-max normalise
over the whole study, then the dataset's ``permute`` + the driver's ``[B, 4, 3, H, W] -> [B, 12, H, W]`` view
(train_lesion_multiphase_v2.py:198).  Three HIP launches (horizontal pass, vertical pass + extrema, finish); the only host
arithmetic is the coefficient table (double precision, like Pillow), computed once per geometry by the C library.
"""
import numpy as np
import torch

from . import _lib
from ._lib import lib, check

FILTERS = {'bilinear': 2, 'bicubic': 3}


def resample_tables(in_size, out_size, filt='bicubic'):
    """Host coefficient tables (numpy int32): bounds [out, 2], kk [out, ksize]."""
    f = FILTERS[filt]
    ksize = lib.gssd_resample_ksize(in_size, out_size, f)
    if ksize <= 0:
        raise _lib.GssdError(f'bad resample geometry {in_size} -> {out_size} ({filt})')
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    check(lib.gssd_resample_coeffs(in_size, out_size, f, bounds.ctypes.data, kk.ctypes.data))
    return bounds, kk


class DeviceInputStage:
    """``stage(raw)``: ``raw`` uint8 ``[B, phases, S, S, 3]`` on the GPU -> fp32 ``[B, phases * 3, size, size]``."""

    def __init__(self, size=300, mean=(49., 49., 49.), use_normalize=True, filt='bicubic'):
        self.size, self.use_normalize, self.filt = int(size), bool(use_normalize), filt
        m = np.asarray(mean, np.float32).reshape(-1)
        self.mean = tuple(float(v) for v in (m if m.size == 3 else np.repeat(m, 3)[:3]))
        if filt not in FILTERS:
            raise ValueError(f'unknown filter {filt!r}')
        self._tables = {}
        self._work = {}

    def _table(self, in_size, dev):
        key = (in_size, dev)
        if key not in self._tables:
            b, k = resample_tables(in_size, self.size, self.filt)
            self._tables[key] = (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), k.shape[1])
        return self._tables[key]

    def __call__(self, raw, out=None):
        if not (isinstance(raw, torch.Tensor) and raw.is_cuda):
            raise _lib.GssdError('input stage: raw study slices must be a uint8 tensor on the MI355X (no CPU fallback)')
        if raw.dtype != torch.uint8 or raw.dim() != 5 or raw.shape[-1] != 3:
            raise _lib.GssdError(f'expected uint8 [B, phases, S, S, 3], got {raw.dtype} {tuple(raw.shape)}')
        raw = raw.contiguous()
        B, P, H, W, Cc = raw.shape
        S, dev = self.size, raw.device
        key = (B, P, H, W, dev)
        if key not in self._work:
            self._work[key] = (torch.empty(B * P, H, S, Cc, dtype=torch.uint8, device=dev),
                               torch.empty(B * P, S, S, Cc, dtype=torch.uint8, device=dev),
                               torch.zeros(B, Cc, 2, dtype=torch.int32, device=dev))
        tmp, small, mm = self._work[key]
        stream = torch.cuda.current_stream().cuda_stream
        src = raw
        if W != S:
            bx, kx, ksx = self._table(W, dev)
            check(lib.gssd_resize_u8_horizontal(raw.data_ptr(), tmp.data_ptr(), bx.data_ptr(), kx.data_ptr(), ksx, B * P, H, W, S,
                                                Cc, stream))
            src = tmp
        mm.zero_()
        by, ky, ksy = self._table(H, dev)              # H == S: a one-tap identity table, still folds the extrema
        check(lib.gssd_resize_u8_vertical(src.data_ptr(), small.data_ptr(), by.data_ptr(), ky.data_ptr(), ksy, B * P, H, S, S, Cc,
                                          P, mm.data_ptr(), stream))
        if out is None:
            out = torch.empty(B, P * Cc, S, S, dtype=torch.float32, device=dev)
        check(lib.gssd_input_finish_f32(small.data_ptr(), mm.data_ptr(), *self.mean, out.data_ptr(), B, P, S, Cc,
                                        int(self.use_normalize), stream))
        self.last_minmax = mm
        return out

    def check_not_flat(self):
        """The reference asserts ``x_min != x_max`` (data/__init__.py:49); on the device that costs a sync, so it is opt-in."""
        mm = self.last_minmax.cpu().numpy()
        lo = (255 - mm[:, :, 0]).astype(np.float32) - np.asarray(self.mean, np.float32)
        hi = mm[:, :, 1].astype(np.float32) - np.asarray(self.mean, np.float32)
        assert (lo.min(1) != hi.max(1)).all(), 'all-black image detected during Normalizing. check preprocessing'
