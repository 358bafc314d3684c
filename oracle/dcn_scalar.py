"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import oracle/).

ctypes binding of oracle/csrc/dcn_scalar.c: the scalar, per-output-element restatement of DCNv2 that cross-checks
oracle/gssd_oracle.py::dcn_v2_conv (vectorised torch) and the HIP kernels.  PARITY UNPINNED upstream (see the C file's header)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_build', 'libdcn_scalar.so')
_lib = None


def build():
    subprocess.run(['make', '-C', HERE, '-s'], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.dcn_scalar_forward.restype = C.c_int
        _lib.dcn_scalar_forward.argtypes = [C.c_void_p] * 6 + [C.c_int] * 11
    return _lib


def dcn_v2_conv(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1, deformable_groups=1):
    """Same signature as gssd_oracle.dcn_v2_conv (numpy or torch NCHW float32 in, numpy float32 out)."""
    a = [np.ascontiguousarray(np.asarray(t, dtype=np.float32)) for t in (x, offset, mask, weight)]
    b = None if bias is None else np.ascontiguousarray(np.asarray(bias, dtype=np.float32))
    B, Cc, H, W = a[0].shape
    Cout, _, kh, kw = a[3].shape
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    assert a[1].shape == (B, 2 * kh * kw * deformable_groups, Ho, Wo) and a[2].shape == (B, kh * kw * deformable_groups, Ho, Wo)
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    rc = lib().dcn_scalar_forward(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                                  None if b is None else b.ctypes.data, out.ctypes.data, B, Cc, H, W, Cout, kh, kw, stride, padding,
                                  dilation, deformable_groups)
    if rc != 0:
        raise ValueError('dcn_scalar_forward: channels do not divide into the deformable groups')
    return out
