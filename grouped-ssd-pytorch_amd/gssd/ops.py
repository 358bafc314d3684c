"""Tensor-level wrappers over the C ABI: each takes torch CUDA(ROCm) tensors, hands raw device
pointers + the current HIP stream to libgssd_hip.so and returns torch tensors.  PyTorch is only the
allocator / stream owner here; every computation happens in the HIP kernels.  No CPU fallback:
a non-CUDA tensor raises."""
import ctypes as C

import os
import torch

from . import _lib
from ._lib import ConvDesc, SnItem, check, lib

BK = 32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.GssdError('the GSSD HIP path needs tensors on the MI355X (got a CPU tensor); '
                                 'there is no CPU fallback')


def round_up(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------------
# packing
# ------------------------------------------------------------------------------------------------
def pack_input(x_nchw, groups, cpg_out):
    _need_cuda(x_nchw)
    x = x_nchw.contiguous().float()
    B, Cc, H, W = x.shape
    y = torch.empty(B, H, W, groups * cpg_out, device=x.device, dtype=torch.float32)
    check(lib.gssd_pack_input_nhwc(_p(x), _p(y), B, Cc, H, W, groups, cpg_out, _stream()))
    return y


def unpack_nhwc(x_nhwc, C_=None):
    _need_cuda(x_nhwc)
    B, H, W, S = x_nhwc.shape
    C_ = C_ or S
    y = torch.empty(B, C_, H, W, device=x_nhwc.device, dtype=torch.float32)
    check(lib.gssd_unpack_nhwc_to_nchw(_p(x_nhwc), _p(y), B, C_, H, W, S, _stream()))
    return y


def packed_k(cin_g, kh, kw):
    cin_pad = round_up(cin_g, 4)
    return cin_pad, kh * kw * cin_pad


class PackRecorder:
    """While installed (``ops.recorder = PackRecorder()``), pack_weight / copy_into append table items instead of launching:
    gssd.engine builds ONE device table out of every weight-refresh job and replays it with a single launch per optimizer step."""

    class Unstable(Exception):
        """a source that is not the parameter's own fp32 storage (its pointer would not survive to the next refresh)"""

    def __init__(self):
        self.items, self.keep = [], []

    def add(self, src, dst, Cout, cin_g, taps, cin_pad, K):
        self.items.append((src.data_ptr(), dst.data_ptr(), Cout, cin_g, taps, cin_pad, K))
        self.keep += [src, dst]


recorder = None


def _stable_f32(t):
    # only a parameter's own storage may enter the table: anything derived from it (a padded / cast / transposed temporary) would be
    # packed again from its stale copy at the next refresh
    d = t.detach()
    if not isinstance(t, torch.nn.Parameter) or d.dtype != torch.float32 or not d.is_contiguous():
        raise PackRecorder.Unstable()
    return d


def pack_weight(w_oihw, out=None, row_offset=0):
    """OIHW -> [Cout][K] K-major rows (k = tap*cin_g_pad + c).  ``out``/``row_offset`` let several
    weights share one packed matrix (loc+conf heads)."""
    _need_cuda(w_oihw)
    w = _stable_f32(w_oihw) if recorder is not None else w_oihw.detach().contiguous().float()
    Cout, cin_g, KH, KW = w.shape
    cin_pad, K = packed_k(cin_g, KH, KW)
    if out is None:
        out = torch.empty(Cout, K, device=w.device, dtype=torch.float32)
    assert out.shape[1] == K
    dst = out[row_offset:row_offset + Cout]
    if recorder is not None:
        recorder.add(w, dst, Cout, cin_g, KH * KW, cin_pad, K)
        return out
    check(lib.gssd_pack_conv_weight(_p(w), _p(dst), Cout, cin_g, KH, KW, cin_pad, K, _stream()))
    return out


def copy_into(dst, src):
    """dst.copy_(src) for fp32 contiguous tensors of equal size -- as a table item while a PackRecorder is installed."""
    if recorder is not None and dst.dtype == torch.float32 and dst.is_contiguous() and dst.numel() == src.numel():
        s = _stable_f32(src)
        n = s.numel()
        recorder.add(s, dst, 1, n, 1, n, n)
        return dst
    if recorder is not None:
        raise PackRecorder.Unstable()
    return dst.copy_(src.detach().reshape(dst.shape) if src.numel() == dst.numel() else src.detach())


def pack_table(rec, device):
    """Device table (gssd_pack_item[n]) of a PackRecorder's items."""
    import numpy as np
    dt = np.dtype([('w', np.uint64), ('wp', np.uint64), ('Cout', np.int32), ('cin_g', np.int32), ('taps', np.int32),
                   ('cin_pad', np.int32), ('K', np.int32), ('reserved', np.int32)])
    arr = np.zeros(len(rec.items), dt)
    for i, it in enumerate(rec.items):
        arr[i] = it + (0,)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


def run_pack_table(table, n):
    check(lib.gssd_pack_conv_weights_batched(_p(table), n, _stream()))


def pack_weight_bf16(w_oihw, out=None, row_offset=0, cin_pad=None):
    """OIHW fp32 -> bf16 [Cout][K] K-major rows (k = tap*cin_pad + c, cin_pad a multiple of 8) for gssd_conv2d_nhwc_bf16."""
    _need_cuda(w_oihw)
    w = w_oihw.detach().contiguous().float()
    Cout, cin_g, KH, KW = w.shape
    cin_pad = cin_pad or round_up(cin_g, 8)
    K = KH * KW * cin_pad
    if out is None:
        out = torch.empty(Cout, K, device=w.device, dtype=torch.bfloat16)
    assert out.shape[1] == K and out.dtype == torch.bfloat16
    dst = out[row_offset:row_offset + Cout]
    check(lib.gssd_pack_conv_weight_bf16(_p(w), _p(dst), Cout, cin_g, KH, KW, cin_pad, K, _stream()))
    return out


def cast_bf16(x, out=None):
    """fp32 -> bf16 (round to nearest even) on the device."""
    _need_cuda(x)
    x = x.detach().contiguous().float()
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib.gssd_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# conv
# ------------------------------------------------------------------------------------------------
def make_conv_desc(inp, wgt, out, *, B, H, W, in_stride, cin_g, Cout, groups=1, k=1, stride=1, pad=0, dil=1,
                   bias=None, in_ch_off=0, out_stride=None, out_ch_off=0, out_mode=_lib.OUT_NHWC, relu=False,
                   stats=None, alpha=None, gate=None, resid=None, out2=None, out_b=None, split_n=0,
                   m_per_image=False, in_batch_stride=0, wgt_batch_stride=0, out_batch_stride=0,
                   outb_batch_stride=0, out_off=0, outb_off=0, wgt_row_stride=None, split_k=1, in_scale=None,
                   in_shift=None, in_pad=None, wgt_wino=None, out_b_stride=0, flags=0, pool_sign=None, stats_rep=0, wgt_x6=None, wgt_patch=None):
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    K = k * k * cin_g
    d = ConvDesc()
    d.in_, d.wgt, d.bias, d.out, d.out_b = _p(inp), _p(wgt), _p(bias), _p(out), _p(out_b)
    d.alpha, d.gate, d.resid, d.out2, d.stats = _p(alpha), _p(gate), _p(resid), _p(out2), _p(stats)
    d.B, d.H, d.W, d.in_stride, d.in_ch_off, d.Ho, d.Wo = B, H, W, in_stride, in_ch_off, Ho, Wo
    d.Cout, d.groups, d.cin_g, d.KH, d.KW, d.stride, d.pad, d.dil = Cout, groups, cin_g, k, k, stride, pad, dil
    d.K = K
    d.wgt_row_stride = wgt_row_stride if wgt_row_stride is not None else K
    d.out_stride = out_stride if out_stride is not None else Cout
    d.out_ch_off, d.out_mode, d.relu, d.m_per_image, d.split_n = out_ch_off, out_mode, int(relu), int(m_per_image), split_n
    d.split_k = split_k
    d.in_scale, d.in_shift, d.in_pad = _p(in_scale), _p(in_shift), _p(in_pad)
    d.wgt_wino = _p(wgt_wino)
    d.wgt_x6 = _p(wgt_x6)
    d.wgt_patch = _p(wgt_patch)
    d.stats_rep = int(stats_rep)
    d.pool_sign = _p(pool_sign)
    d.out_b_stride = out_b_stride
    d.flags = flags
    d.in_batch_stride, d.wgt_batch_stride = in_batch_stride, wgt_batch_stride
    d.out_batch_stride, d.outb_batch_stride, d.out_off, d.outb_off = out_batch_stride, outb_batch_stride, out_off, outb_off
    return d, Ho, Wo


def auto_split_k(M, Cout, groups, K, target_blocks=1024, max_split=48):
    """Split-K factor for small-grid / long-K launches (mirrors the tile choice of csrc/conv_igemm.hip)."""
    cout_g = Cout // groups
    bn = 128 if cout_g > 64 else 64 if cout_g > 32 else 32 if cout_g > 16 else 16
    blocks = -(-M // 128) * groups * (-(-cout_g // bn))
    chunks = -(-K // BK)
    return int(max(1, min(max_split, chunks // 4, target_blocks // blocks)))


def run_conv(desc):
    check(lib.gssd_conv2d_nhwc_f32(C.byref(desc), _stream()))


def winograd_eligible(k, stride, pad, dil, cin_g, cout_g, groups=4):
    """Shapes csrc/conv_wino.hip takes (mirrors gssd_try_conv_wino)."""
    if not (k == 3 and stride == 1 and pad == 1 and dil == 1 and cin_g % 16 == 0):
        return False
    return cout_g % 32 == 0 or (groups == 1 and cout_g >= 24) or (groups == 4 and cin_g == 16 and cout_g == 16)


def x6_tile(cout_g, groups, M):
    """N tile (64 / 128 / 256) csrc/conv_x6.hip uses for this launch; the packed weights depend on it."""
    return int(lib.gssd_conv_x6_tile(cout_g, groups, M))


X6_F16 = os.environ.get('GSSD_X6_F16', '1') != '0'        # csrc/conv_x6.hip & co.: forward launches on fp16 planes, three MFMAs per product


def x6_wanted(k, cin_g, cout_g, groups, M, winograd=False, forward=False):
    """Launches the engine hands to csrc/conv_x6.hip (fp32 mode): at least 128 output channels per group and 256 k, M = B Ho Wo >= 4096 (19 x 19
    maps at batch 32, 38 x 38 at batch 4) -- measured with scripts/bench_conv_x6.py: conv6 306 -> 174 us, conv7 84 -> 56, the fuse convs
    208 -> 166, the Self_Attn output convs 143-175 -> ~90.  Not the Winograd shapes (the same speed there with bf16 planes; the large ones belong to
    csrc/conv_wino_x6.hip) -- except, round 6, FORWARD launches on maps too small for conv_wino_x6 (conv5_x at batch 32, M = 11 552): the fp16-plane
    form runs them in 76 us against fp32 Winograd's 100 (their data gradients: 97 against 101, left where they were)."""
    ok = cin_g % 32 == 0 and cout_g >= 128 and cout_g % 8 == 0 and k * k * cin_g >= 256 and M >= 4096
    if winograd:
        return ok and forward and X6_F16 and M < 16384
    return ok


def x6_weight(w_packed, groups, cin_g, taps, bn, out=None):
    """Packed K-major fp32 weights [Cout][taps*cin_g] -> the three bf16 planes of csrc/conv_x6.hip (uint16 tensor)."""
    Cout = w_packed.shape[0]
    if out is None:
        n = int(lib.gssd_conv_x6_weight_elems(Cout, groups, cin_g, taps, bn))
        if n <= 0:
            raise _lib.GssdError(f'not a conv_x6 shape: Cout {Cout}, groups {groups}, cin_g {cin_g}, tile {bn}')
        out = torch.empty(n, device=w_packed.device, dtype=torch.int16)
    check(lib.gssd_conv_x6_pack_weight(_p(w_packed), _p(out), Cout, groups, cin_g, taps, w_packed.stride(0), bn, _stream()))
    return out


def patch_x6_weight(w_packed, cin, out=None):
    """Packed K-major fp32 weights [Cout][9*cin] -> the two fp16 planes of csrc/conv_patch_x6.hip (int16 tensor)."""
    Cout = w_packed.shape[0]
    if out is None:
        n = int(lib.gssd_conv_patch_x6_weight_elems(Cout, cin))
        if n <= 0:
            raise _lib.GssdError(f'not a conv_patch_x6 shape: Cout {Cout}, cin {cin}')
        out = torch.empty(n, device=w_packed.device, dtype=torch.int16)
    check(lib.gssd_conv_patch_x6_pack_weight(_p(w_packed), _p(out), Cout, cin, w_packed.stride(0), _stream()))
    return out


def winograd_weight(w_packed, groups, cin_g, out=None):
    """Packed K-major 3x3 weights [Cout][9*cin_g] -> U[g][16][cout_pad][cin_g] (G g G^T)."""
    Cout = w_packed.shape[0]
    if out is None:
        n = int(lib.gssd_winograd_weight_elems(Cout, groups, cin_g))
        if n <= 0:
            raise _lib.GssdError(f'not a Winograd shape: Cout {Cout}, groups {groups}, cin_g {cin_g}')
        out = torch.empty(n, device=w_packed.device, dtype=torch.float32)
    check(lib.gssd_winograd_weight_f32(_p(w_packed), _p(out), Cout, groups, cin_g, w_packed.stride(0), _stream()))
    return out


def conv2d_nhwc(x, w_oihw, bias=None, stride=1, pad=0, dil=1, groups=1, relu=False, stats=None, winograd=False, x6=False, patch=False, _keep=None, **kw):
    """Convenience one-shot conv for tests: x NHWC [B,H,W,Cin], weight OIHW; returns NHWC."""
    _need_cuda(x, w_oihw)
    B, H, W, Cin = x.shape
    Cout, cin_g, k, _ = w_oihw.shape
    assert cin_g % 4 == 0 and Cin == cin_g * groups
    wp = pack_weight(w_oihw)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    out = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    U = winograd_weight(wp, groups, cin_g) if winograd else None
    X6 = x6_weight(wp, groups, cin_g, k * k, x6_tile(Cout // groups, groups, B * Ho * Wo)) if x6 else None
    if patch:
        kw['wgt_patch'] = patch_x6_weight(wp, cin_g)
        kw['flags'] = kw.get('flags', 0) | _lib.CONV_F16_OK | _lib.CONV_OUT_F32
    d, _, _ = make_conv_desc(x, wp, out, B=B, H=H, W=W, in_stride=Cin, cin_g=cin_g, Cout=Cout, groups=groups, k=k,
                             stride=stride, pad=pad, dil=dil, bias=bias, relu=relu, stats=stats, wgt_wino=U, wgt_x6=X6, **kw)
    if patch and lib.gssd_conv_patch_x6_takes(C.byref(d)) != 1:
        raise _lib.GssdError('conv2d_nhwc(patch=True): csrc/conv_patch_x6.hip does not take this descriptor')
    if x6 and lib.gssd_conv_x6_takes(C.byref(d)) != 1:
        raise _lib.GssdError('conv2d_nhwc(x6=True): csrc/conv_x6.hip does not take this descriptor')
    if _keep is not None:
        _keep.extend([d, wp, U, X6, kw.get('wgt_patch'), out])           # (the output last: callers read keep[-1])
        return d
    run_conv(d)
    return out


def conv_wgrad(desc, dy, Cout, cin_g_real, k, cin_g_pad=None, out=None, accumulate=False):
    """OIHW weight gradient of the conv described by ``desc`` for the dense NHWC output gradient ``dy``."""
    cin_g_pad = cin_g_pad or desc.cin_g
    K = k * k * cin_g_pad
    dwp = torch.zeros(Cout, K, device=dy.device, dtype=torch.float32)
    check(lib.gssd_conv2d_wgrad_f32(C.byref(desc), _p(dy), _p(dwp), _stream()))
    if out is None:
        out = torch.empty(Cout, cin_g_real, k, k, device=dy.device, dtype=torch.float32)
    check(lib.gssd_unpack_conv_weight_grad(_p(dwp), _p(out), Cout, cin_g_real, k, k, cin_g_pad, K, int(accumulate), _stream()))
    return out


def pack_weight_dgrad(w_oihw, groups, out=None):
    """OIHW -> [Cin_total][taps*cout_g] rows for the data-gradient conv (flipped taps, co <-> ci)."""
    w = w_oihw.detach().contiguous().float()
    Cout, cin_g, KH, KW = w.shape
    Kd = KH * KW * (Cout // groups)
    if out is None:
        out = torch.empty(groups * cin_g, Kd, device=w.device, dtype=torch.float32)
    check(lib.gssd_pack_conv_weight_dgrad(_p(w), _p(out), Cout, groups, cin_g, KH, KW, _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# elementwise
# ------------------------------------------------------------------------------------------------
def bn_relu_pool(raw, out, stats, count, gamma, beta, rmean, rvar, training, relu=True, pool=None, momentum=0.1,
                 eps=1e-5, stats_rep=0):
    """raw/out NHWC.  pool = (k, s, p) or None."""
    B, H, W, Cc = raw.shape
    _, Ho, Wo, _ = out.shape
    pk, ps, pp = pool if pool else (0, 1, 0)
    check(lib.gssd_bn_relu_pool_f32(_p(raw), _p(out), B, H, W, Cc, Ho, Wo, pk, ps, pp, _p(stats), float(count),
                                    _p(gamma), _p(beta), _p(rmean), _p(rvar), momentum, eps, int(training), int(relu),
                                    int(stats_rep), _stream()))
    return out


def bn_finalize(stats, count, gamma, beta, rmean, rvar, training, scale, shift, pad, momentum=0.1, eps=1e-5, stats_rep=0):
    check(lib.gssd_bn_finalize_f32(_p(stats), float(count), _p(gamma), _p(beta), _p(rmean), _p(rvar), momentum, eps,
                                   int(training), gamma.numel(), _p(scale), _p(shift), _p(pad), int(stats_rep), _stream()))


def pool_out_size(n, k, s, p, ceil):
    if ceil:
        o = -(-(n + 2 * p - k) // s) + 1
        if (o - 1) * s >= n + p:
            o -= 1
        return o
    return (n + 2 * p - k) // s + 1


def l2norm(x, weight, eps=1e-10, out=None):
    _need_cuda(x, weight)
    Cc = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    check(lib.gssd_l2norm_f32(_p(x), _p(weight), _p(out), x.numel() // Cc, Cc, eps, _stream()))
    return out


def softmax_rows_(x, n):
    rows = x.numel() // x.shape[-1]
    check(lib.gssd_softmax_rows_f32(_p(x), rows, n, x.shape[-1], _stream()))
    return x


def slice_and_cat(a, b, groups, out=None):
    _need_cuda(a, b)
    Ca, Cb = a.shape[-1], b.shape[-1]
    out = torch.empty(*a.shape[:-1], Ca + Cb, device=a.device, dtype=torch.float32) if out is None else out
    check(lib.gssd_slice_and_cat_f32(_p(a), _p(b), _p(out), a.numel() // Ca, Ca, Cb, groups, _stream()))
    return out


def sn_items_tensor(items, device):
    """items: list of (w, u, v, inv_sigma_slice).  Returns a uint8 device tensor holding gssd_sn_item[]."""
    arr = (SnItem * len(items))()
    for i, (w, u, v, s) in enumerate(items):
        rows = w.shape[0]
        cols = w.numel() // rows
        if rows + cols > 4096:
            raise _lib.GssdError(f'spectral norm: rows + cols = {rows + cols} exceeds the kernel\'s 4096-float LDS vectors')
        arr[i].w, arr[i].u, arr[i].v, arr[i].inv_sigma, arr[i].rows, arr[i].cols = _p(w), _p(u), _p(v), _p(s), rows, cols
    raw = bytes(arr)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


def spectral_norm(items_dev, n, do_power_iteration, eps=1e-12):
    check(lib.gssd_spectral_norm_f32(_p(items_dev), n, int(do_power_iteration), eps, _stream()))


def dcn_im2col(x, om, cols, dg):
    B, H, W, Cc = x.shape
    check(lib.gssd_dcn_im2col_f32(_p(x), _p(om), _p(cols), B, H, W, Cc, dg, om.shape[-1], _stream()))
    return cols


def dcn_pack_weight(w_oihw, dg, out=None):
    """Dense OIHW [Cout][C][3][3] -> the chunk-major, LDS-image layout gssd_dcn_forward_f32 streams by LDS-DMA."""
    _need_cuda(w_oihw)
    w = w_oihw.detach().contiguous().float()
    Cout, Cc = w.shape[0], w.shape[1]
    if out is None:
        n = int(lib.gssd_dcn_packed_weight_elems(Cout, Cc))
        if n <= 0:
            raise _lib.GssdError(f'deformable conv: unsupported shape C {Cc}, Cout {Cout}')
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    check(lib.gssd_dcn_pack_weight_f32(_p(w), _p(out), Cout, Cc, dg, _stream()))
    return out


def dcn_forward(x, om, w_oihw, bias, dg, w_packed=None):
    """Fused modulated deformable 3x3 conv (layers/dcn_v2_custom.py:84-89): x NHWC [B,H,W,C], om NHWC [B,H,W,27*dg] (the raw
    conv_offset_mask output) -> NHWC [B,H,W,Cout]."""
    _need_cuda(x, om, w_oihw)
    B, H, W, Cc = x.shape
    Cout = w_oihw.shape[0]
    wp = w_packed if w_packed is not None else dcn_pack_weight(w_oihw, dg)
    out = torch.empty(B, H, W, Cout, device=x.device, dtype=torch.float32)
    check(lib.gssd_dcn_forward_f32(_p(x), _p(om), _p(wp), _p(bias), _p(out), B, H, W, Cc, dg, om.shape[-1], Cout, _stream()))
    return out


def dcn_forward_x6(x, om, w_oihw, bias, dg, f16ok=False):
    """The same op through csrc/dcn_x6.hip (fp32 in / out, three-plane bf16 split on the bf16 matrix cores; ``f16ok``: the caller's promise that x
    and the weights lie inside fp16's range -- GSSD_CONV_F16_OK -- fp16 planes, three MFMAs per product)."""
    _need_cuda(x, om, w_oihw)
    B, H, W, Cc = x.shape
    w = w_oihw.detach().contiguous().float()
    Cout = w.shape[0]
    n = int(lib.gssd_dcn_packed_weight_elems_x6(Cout, Cc))
    if n <= 0:
        raise _lib.GssdError(f'deformable conv (x6): unsupported shape C {Cc}, Cout {Cout}')
    wp = torch.empty(n, device=w.device, dtype=torch.bfloat16)
    check(lib.gssd_dcn_pack_weight_x6(_p(w), _p(wp), Cout, Cc, dg, _stream()))
    out = torch.empty(B, H, W, Cout, device=x.device, dtype=torch.float32)
    check(lib.gssd_dcn_forward_x6_ex(_p(x), _p(om), _p(wp), _p(bias), _p(out), B, H, W, Cc, dg, om.shape[-1], Cout,
                                     _lib.CONV_F16_OK if f16ok else 0, _stream()))
    return out


def dcn_col2im(x, om, dcols, dx, dom, dg):
    """Backward of :func:`dcn_im2col`: ADDS d(x) into ``dx`` and d(om) (offsets + mask logits) into ``dom``."""
    B, H, W, Cc = x.shape
    check(lib.gssd_dcn_col2im_f32(_p(x), _p(om), _p(dcols), _p(dx), _p(dom), B, H, W, Cc, dg, om.shape[-1], _stream()))
    return dx, dom


def softmax_lastdim(x):
    _need_cuda(x)
    x = x.contiguous().float()
    y = torch.empty_like(x)
    check(lib.gssd_softmax_lastdim_f32(_p(x), _p(y), x.numel() // x.shape[-1], x.shape[-1], _stream()))
    return y


# ------------------------------------------------------------------------------------------------
# loss / detect
# ------------------------------------------------------------------------------------------------
def pack_targets(targets, device):
    """Python list of [n_i, 5] tensors -> (flat [sum n_i, 5] float32, offsets [B+1] int32) on ``device``.
    Shapes are host metadata, so nothing synchronises: GPU-resident targets are concatenated on the device
    (the reference does B D2H copies instead, multibox_loss.py:67-75)."""
    ns = [int(t.shape[0]) for t in targets]
    if max(ns) > 64:
        raise _lib.GssdError('more than 64 ground-truth boxes in one image')
    off = [0]
    for n in ns:
        off.append(off[-1] + n)
    if all(t.is_cuda for t in targets):
        flat = torch.cat([t.detach().reshape(-1, 5) for t in targets], 0).to(device=device, dtype=torch.float32)
    else:
        flat = torch.cat([t.detach().to('cpu', torch.float32).reshape(-1, 5) for t in targets], 0).to(device)
    if flat.shape[0] == 0:
        flat = torch.zeros(1, 5, device=device)
    return flat.contiguous(), torch.tensor(off, dtype=torch.int32).to(device, non_blocking=True)


def match_batch(tg, gt_off, priors, threshold=0.5, variance=(0.1, 0.2)):
    _need_cuda(tg, gt_off, priors)
    B = gt_off.shape[0] - 1
    P = priors.shape[0]
    loc_t = torch.empty(B, P, 4, device=tg.device, dtype=torch.float32)
    conf_t = torch.empty(B, P, device=tg.device, dtype=torch.int64)
    check(lib.gssd_match_batch(_p(tg), _p(gt_off), _p(priors), B, P, threshold, variance[0], variance[1], _p(loc_t),
                               _p(conf_t), _stream()))
    return loc_t, conf_t


def multibox_loss_forward(loc, conf, priors, tg, n_gt, threshold=0.5, negpos_ratio=3, variance=(0.1, 0.2),
                          want_scores=False, global_n=False):
    """Returns dict(losses[2], loc_t, conf_t, sel, n_total, loss_c_all?).  ``global_n``: the normaliser N of multibox_loss.py:117 over every
    rank's images (one all-reduce of one double) instead of this rank's -- SURVEY.md 8e's exact equivalence to the single big batch."""
    _need_cuda(loc, conf, priors)
    loc = loc.contiguous()
    conf = conf.contiguous()
    B, P, _ = loc.shape
    Cc = conf.shape[-1]
    priors = priors[:P].contiguous()
    loc_t, conf_t = match_batch(tg, n_gt, priors, threshold, variance)
    dev = loc.device
    NX = 128
    xmax = torch.empty(NX, device=dev, dtype=torch.float32)
    check(lib.gssd_reduce_max_f32(_p(conf), conf.numel(), _p(xmax), NX, _stream()))
    sel = torch.empty(B, P, device=dev, dtype=torch.uint8)
    partial = torch.empty(B, 4, device=dev, dtype=torch.float64)
    lca = torch.empty(B, P, device=dev, dtype=torch.float32) if want_scores else None
    check(lib.gssd_hnm_loss(_p(loc), _p(conf), _p(loc_t), _p(conf_t), _p(xmax), NX, B, P, Cc, int(negpos_ratio), _p(sel),
                            _p(partial), _p(lca), _stream()))
    losses = torch.empty(2, device=dev, dtype=torch.float32)
    n_total = torch.empty(1, device=dev, dtype=torch.float64)
    check(lib.gssd_loss_finalize(_p(partial), B, _p(losses), _p(n_total), _stream()))
    if global_n:
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size() > 1:
            n_glob = n_total.clone()
            dist.all_reduce(n_glob, op=dist.ReduceOp.SUM)
            check(lib.gssd_loss_finalize_global(_p(partial), B, _p(n_glob), dist.get_world_size(), _p(losses), _p(n_total), _stream()))
    return dict(losses=losses, loc_t=loc_t, conf_t=conf_t, sel=sel, n_total=n_total, loss_c_all=lca, partial=partial,
                loc=loc, conf=conf)


def multibox_loss_backward(ctx, grad_l, grad_c):
    loc, conf = ctx['loc'], ctx['conf']
    B, P, _ = loc.shape
    Cc = conf.shape[-1]
    dloc = torch.empty_like(loc)
    dconf = torch.empty_like(conf)
    gl = grad_l.reshape(1).float().contiguous() if grad_l is not None else None
    gc = grad_c.reshape(1).float().contiguous() if grad_c is not None else None
    check(lib.gssd_loss_backward(_p(loc), _p(conf), _p(ctx['loc_t']), _p(ctx['conf_t']), _p(ctx['sel']),
                                 _p(ctx['n_total']), _p(gl), _p(gc), B, P, Cc, _p(dloc), _p(dconf), _stream()))
    if grad_l is None:
        dloc.zero_()
    if grad_c is None:
        dconf.zero_()
    return dloc, dconf


def detect(loc, conf, priors, num_classes, top_k=200, conf_thresh=0.01, nms_thresh=0.45, variance=(0.1, 0.2),
           conf_is_logits=False, want_keep=False, loc_is_boxes=False):
    _need_cuda(loc, conf, priors)
    if nms_thresh <= 0:
        raise ValueError('nms_threshold must be non negative.')     # detection_pytorch_ver_1point5.py:39-40
    loc = loc.contiguous().float()
    conf = conf.contiguous().float()
    priors = priors.contiguous().float()
    B, P, _ = loc.shape
    out = torch.empty(B, num_classes, top_k, 5, device=loc.device, dtype=torch.float32)
    keep = torch.empty(B, num_classes, top_k, device=loc.device, dtype=torch.int32) if want_keep else None
    cnt = torch.empty(B, num_classes, device=loc.device, dtype=torch.int32) if want_keep else None
    check(lib.gssd_detect(_p(loc), _p(conf), _p(priors), B, P, num_classes, top_k, conf_thresh, nms_thresh,
                          variance[0], variance[1], int(conf_is_logits), int(loc_is_boxes), _p(out), _p(keep), _p(cnt),
                          _stream()))
    return (out, keep, cnt) if want_keep else out


def detect_boxes(boxes, scores, overlap, top_k):
    """Greedy NMS over raw (x1,y1,x2,y2) boxes with positive scores: (rows[top_k,5], keep[top_k] int32, count)."""
    _need_cuda(boxes, scores)
    n = boxes.shape[0]
    conf = torch.stack([torch.zeros_like(scores), scores], 1).unsqueeze(0).contiguous()
    out, keep, cnt = detect(boxes.unsqueeze(0), conf, boxes, 2, top_k=top_k, conf_thresh=0.0, nms_thresh=overlap,
                            want_keep=True, loc_is_boxes=True)
    return out[0, 1], keep[0, 1], cnt[0, 1]


# ------------------------------------------------------------------------------------------------
# backward helpers (tests and the engine's backward plan use the same entry points)
# ------------------------------------------------------------------------------------------------
def bn_backward(dout, raw, fwd_stats, count, gamma, scale, shift, pool=None, relu=True, eps=1e-5, want_colsum=False, stats_rep=0):
    """d(conv output) of conv -> BN(train) -> ReLU -> (pool).  Returns (draw NHWC, dgamma, dbeta, colsum or None)."""
    B, H, W, Cc = raw.shape
    _, Ho, Wo, _ = dout.shape
    pk, ps, pp = pool if pool else (0, 1, 0)
    dev = raw.device
    dz = torch.zeros_like(raw) if (pool and ps < pk) else torch.empty_like(raw)
    sums = torch.zeros(2 * Cc, device=dev, dtype=torch.float64)
    check(lib.gssd_bn_bwd_reduce_f32(_p(dout), _p(raw), _p(scale), _p(shift), _p(dz), _p(sums), B, H, W, Cc, Ho, Wo, pk, ps, pp,
                                     int(relu), _stream()))
    ca, cb, cc, dg, db = (torch.empty(Cc, device=dev) for _ in range(5))
    check(lib.gssd_bn_bwd_finalize_f32(_p(fwd_stats), float(count), _p(sums), _p(gamma), eps, Cc, _p(ca), _p(cb), _p(cc), _p(dg),
                                       _p(db), int(stats_rep), _stream()))
    cs = torch.zeros(Cc, device=dev, dtype=torch.float64) if want_colsum else None
    check(lib.gssd_bn_bwd_apply_f32(_p(dz), _p(raw), _p(ca), _p(cb), _p(cc), B * H * W, Cc, _p(cs), _stream()))
    return dz, dg, db, cs


def l2norm_backward(x, weight, dy, dx_add=None, eps=1e-10):
    Cc = x.shape[-1]
    dx = torch.empty_like(x)
    dw = torch.zeros(Cc, device=x.device, dtype=torch.float64)
    check(lib.gssd_l2norm_bwd_f32(_p(x), _p(weight), _p(dy), _p(dx), _p(dx_add), _p(dw), x.numel() // Cc, Cc, eps, _stream()))
    return dx, dw
