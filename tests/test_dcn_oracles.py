"""DCNv2 is "parity unpinned" (layers/dcn_v2_custom.py:13 imports an un-vendored CUDA extension; nothing in the reference holds a
known answer).  The most that can exist here: two structurally independent CPU restatements -- oracle/gssd_oracle.py::dcn_v2_conv
(vectorised torch gathers, fp32 matmul) and oracle/csrc/dcn_scalar.c (scalar loop nest per output element in the order of the
published algorithm, float64 accumulation) -- agreeing on the cases that separate the conventions, plus the anchors the reference
DOES hold: the call site's channel convention (dcn_v2_custom.py:79-89), the zero-init identity (:75-77) and the (row, column)
offset order of utils/show_offset.py:28-32."""
import numpy as np
import pytest
import torch

from oracle import dcn_scalar
from oracle import gssd_oracle as O


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def case(seed, B, Cc, H, W, Cout, dg, stride=1, pad=1, dil=1, off_std=2.5, k=3):
    rng = np.random.default_rng(seed)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    x = rng.normal(size=(B, Cc, H, W)).astype(np.float32)
    off = rng.normal(0, off_std, size=(B, 2 * k * k * dg, Ho, Wo)).astype(np.float32)
    msk = rng.uniform(0, 1, size=(B, k * k * dg, Ho, Wo)).astype(np.float32)
    w = rng.normal(0, 0.2, size=(Cout, Cc, k, k)).astype(np.float32)
    b = rng.normal(size=(Cout,)).astype(np.float32)
    return x, off, msk, w, b


@pytest.mark.parametrize('B,Cc,H,W,Cout,dg,stride,pad,dil,std', [
    (2, 8, 7, 7, 5, 1, 1, 1, 1, 2.5),        # offsets well beyond one pixel, samples leaving the map on every side
    (1, 8, 6, 9, 4, 2, 1, 1, 1, 4.0),        # H != W (a row / column mix-up cannot hide), two deformable groups
    (2, 16, 9, 8, 6, 4, 1, 1, 1, 1.0),       # the detector's dg = 4
    (1, 8, 11, 10, 3, 2, 2, 1, 1, 2.0),      # stride 2
    (1, 4, 10, 12, 3, 1, 1, 2, 2, 2.0),      # dilation 2, pad 2
    (1, 6, 9, 9, 2, 3, 2, 2, 2, 3.0),        # stride 2 + dilation 2, three groups
    (1, 4, 5, 5, 2, 1, 1, 0, 1, 0.7),        # no padding
])
def test_two_restatements_agree(B, Cc, H, W, Cout, dg, stride, pad, dil, std):
    x, off, msk, w, b = case(B * 100 + H, B, Cc, H, W, Cout, dg, stride, pad, dil, std)
    a = O.dcn_v2_conv(torch.from_numpy(x), torch.from_numpy(off), torch.from_numpy(msk), torch.from_numpy(w), torch.from_numpy(b),
                      stride, pad, dil, dg).numpy()
    s = dcn_scalar.dcn_v2_conv(x, off, msk, w, b, stride, pad, dil, dg)
    assert a.shape == s.shape and rel(a, s) < 2e-6


def test_border_samples_exactly_on_the_gates():
    """Sample positions exactly at -1, just inside -1, exactly H-1, just below H and exactly H: the `-1 < y < H` gate and the
    per-corner inside tests are where restatements usually differ."""
    H = W = 5
    x = np.arange(1, H * W + 1, dtype=np.float32).reshape(1, 1, H, W)
    w = np.zeros((1, 1, 3, 3), np.float32)
    w[0, 0, 1, 1] = 1.0                               # only the centre tap: out(h, w) = sample at (h + dy, w + dx)
    msk = np.ones((1, 9, H, W), np.float32)
    for dy, dx, expect in [(-1.0, 0.0, 0.0),                   # from pixel (0, 0): y = -1 exactly -> outside the open interval
                           (-0.75, 0.0, 0.25 * 1.0),           # y = -0.75: only the h_high = 0 row contributes, weight lh = 0.25
                           (4.0, 0.0, 21.0),                   # y = H-1 exactly: h_high = H is outside, weight on h_low = 1
                           (4.5, 0.0, 0.5 * 21.0),             # y = 4.5 < H: inside the gate, lower corner only
                           (5.0, 0.0, 0.0),                    # y = H: gate closed
                           (0.0, -0.5, 0.5 * 1.0), (0.0, 4.5, 0.5 * 5.0)]:
        off = np.zeros((1, 18, H, W), np.float32)
        off[0, 8], off[0, 9] = dy, dx                 # centre tap k = 4: rows 2k, 2k + 1
        s = dcn_scalar.dcn_v2_conv(x, off, msk, w, None, 1, 1, 1, 1)
        a = O.dcn_v2_conv(torch.from_numpy(x), torch.from_numpy(off), torch.from_numpy(msk), torch.from_numpy(w), torch.zeros(1),
                          1, 1, 1, 1).numpy()
        assert s[0, 0, 0, 0] == np.float32(expect) and a[0, 0, 0, 0] == np.float32(expect), (dy, dx, s[0, 0, 0, 0], a[0, 0, 0, 0])


def test_offset_order_anchor_show_offset():
    """utils/show_offset.py:28-32 reads the layer's offset tensor as offset[2*idx] = row (y) shift, offset[2*idx + 1] = column (x)
    shift of tap idx = 3*i + j.  A +1 ROW shift of one tap must therefore move that tap's sample DOWN one row in both restatements."""
    H, W = 6, 7
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 1, H, W)).astype(np.float32)
    for t in range(9):
        i, j = divmod(t, 3)
        w = np.zeros((1, 1, 3, 3), np.float32)
        w[0, 0, i, j] = 1.0
        off = np.zeros((1, 18, H, W), np.float32)
        off[0, 2 * t] = 1.0                                            # row shift of tap t
        msk = np.ones((1, 9, H, W), np.float32)
        want = np.zeros((H, W), np.float32)
        for h in range(H):
            for ww in range(W):
                y, xx = h - 1 + i + 1, ww - 1 + j
                if 0 <= y < H and 0 <= xx < W:
                    want[h, ww] = x[0, 0, y, xx]
        s = dcn_scalar.dcn_v2_conv(x, off, msk, w, None)[0, 0]
        a = O.dcn_v2_conv(torch.from_numpy(x), torch.from_numpy(off), torch.from_numpy(msk), torch.from_numpy(w), torch.zeros(1))[0, 0]
        assert np.array_equal(s, want) and np.array_equal(a.numpy(), want), t


def test_wrapper_identities_and_linearity():
    """dcn_v2_custom.py:75-77 zero-initialises conv_offset_mask: offsets 0, mask sigmoid(0) = 0.5  =>  0.5 * conv2d(x, W, pad 1) + b.
    The op is linear in the mask and, with integer offsets, equal to a conv over shifted taps."""
    x, off, msk, w, b = case(5, 2, 8, 8, 9, 6, 2)
    z = np.zeros_like(off)
    half = np.full_like(msk, 0.5)
    ident = (0.5 * torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, 1, 1) + torch.from_numpy(b).view(1, -1, 1, 1)).numpy()
    assert rel(dcn_scalar.dcn_v2_conv(x, z, half, w, b, 1, 1, 1, 2), ident) < 2e-6
    y1 = dcn_scalar.dcn_v2_conv(x, off, msk, w, None, 1, 1, 1, 2)
    y2 = dcn_scalar.dcn_v2_conv(x, off, 0.25 * msk, w, None, 1, 1, 1, 2)
    assert rel(y2, 0.25 * y1) < 2e-6
    m2 = np.random.default_rng(9).uniform(0, 1, size=msk.shape).astype(np.float32)
    y3 = dcn_scalar.dcn_v2_conv(x, off, msk + m2, w, None, 1, 1, 1, 2)
    assert rel(y3, y1 + dcn_scalar.dcn_v2_conv(x, off, m2, w, None, 1, 1, 1, 2)) < 2e-6


def test_wrapper_channel_convention():
    """dcn_v2_custom.py:81-82: o1, o2, mask = chunk(conv_offset_mask(x), 3); offset = cat(o1, o2) -- i.e. the first 18*dg channels
    are the offsets in (group, tap, y|x) order and the last 9*dg the mask logits.  oracle.dcn() must feed dcn_v2_conv exactly that."""
    rng = np.random.default_rng(2)
    dg, Cc, Cout, H = 2, 8, 4, 6
    sd = {'d.conv_offset_mask.weight': torch.from_numpy(rng.normal(0, 0.1, size=(27 * dg, Cc, 3, 3)).astype(np.float32)),
          'd.conv_offset_mask.bias': torch.from_numpy(rng.normal(0, 0.5, size=(27 * dg,)).astype(np.float32)),
          'd.weight': torch.from_numpy(rng.normal(0, 0.2, size=(Cout, Cc, 3, 3)).astype(np.float32)),
          'd.bias': torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))}
    x = torch.from_numpy(rng.normal(size=(2, Cc, H, H)).astype(np.float32))
    out, offset = O.dcn(x, sd, 'd', dg)
    om = torch.nn.functional.conv2d(x, sd['d.conv_offset_mask.weight'], sd['d.conv_offset_mask.bias'], 1, 1).numpy()
    assert np.array_equal(offset.numpy(), om[:, :18 * dg])
    s = dcn_scalar.dcn_v2_conv(x.numpy(), om[:, :18 * dg], 1.0 / (1.0 + np.exp(-om[:, 18 * dg:].astype(np.float64))),
                               sd['d.weight'].numpy(), sd['d.bias'].numpy(), 1, 1, 1, dg)
    assert rel(out.numpy(), s) < 5e-6
