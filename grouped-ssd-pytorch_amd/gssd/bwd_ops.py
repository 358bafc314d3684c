"""Per-record handlers of the HIP backward plan: one method per forward record kind (conv + BatchNorm, heads, pools, L2Norm, Self_Attn,
slice_and_cat, the deformable conv, the PixelLink++ tail) plus the weight- / data-gradient emitters they share.  Mixin of
backward.BackwardPlan; SURVEY.md 8f row 1."""
import ctypes as C
import os
import torch
from . import _lib, ops
from ._lib import lib


class _PaddedWeight:
    """Stands in for a conv module where only ``.weight`` is read (the dgrad packer)."""

    def __init__(self, w):
        self.weight = w


def conv_weight_ptr(conv, cin_g_expected):
    """OIHW weight pointer for the dgrad packer (conv1_1 never needs a dgrad, so no channel padding arises)."""
    assert conv.weight.shape[1] == cin_g_expected
    return conv.weight.data_ptr()


def _pack_dgrad_from_packed(hw, wd, Cout, Cs, Cp=None):
    """Merged head weights [Cout][9][Cs] (forward packing) -> dgrad rows [Cs][9 flipped][Cout] (tiny: plain tensor ops); ``Cp``: the
    rows' channel count widened to Cp (the pad stays zero)."""
    if Cp is None or Cp == Cout:
        wd.copy_(hw.view(Cout, 9, Cs).flip(1).permute(2, 1, 0).reshape(Cs, 9 * Cout))
    else:
        wd.view(Cs, 9, Cp)[:, :, :Cout].copy_(hw.view(Cout, 9, Cs).flip(1).permute(2, 1, 0))


class BackwardOpsMixin:
    def _x6w(self, w, Cout, groups, cin_g, k, M, winograd=False):
        """Three-plane bf16 form of the K-major fp32 weight rows ``w`` (re-derived inside the plan, right behind the launch that writes
        ``w``) for the fp32 GEMMs csrc/conv_x6.hip takes (ops.x6_wanted); None otherwise."""
        from .engine import USE_CONV_X6
        if not USE_CONV_X6 or self.bf16_ops or not ops.x6_wanted(k, cin_g, Cout // groups, groups, M, winograd=winograd):
            return None
        bn = ops.x6_tile(Cout // groups, groups, M)
        t = torch.empty(int(lib.gssd_conv_x6_weight_elems(Cout, groups, cin_g, k * k, bn)), device=self.dev, dtype=torch.int16)
        self.keep.append(t)
        self._add(lib.gssd_conv_x6_pack_weight, (w.data_ptr(), t.data_ptr(), Cout, groups, cin_g, k * k, w.stride(0), bn))
        return t

    # gradient contribution of a conv to its input: dX (+)= conv(dY, flipped weights)
    def _dgrad(self, r, dy, x_in, conv, groups, Cin, H, Ho, Cout, k, stride, pad, dil):
        B = self.B
        wd = self._buf(Cin, k * k * (Cout // groups))
        self._add(lib.gssd_pack_conv_weight_dgrad, (conv_weight_ptr(conv, Cin // groups), wd.data_ptr(), Cout, groups,
                                                    Cin // groups, k, k), keep=conv)
        src, Hs = dy, Ho
        if stride != 1:
            u = self._buf(B, H, H, Cout)
            self._add(lib.gssd_upsample_insert_f32, (dy.data_ptr(), u.data_ptr(), B, Ho, Ho, H, H, Cout, stride))
            src, Hs = u, H
            pd = k - 1 - pad
        else:
            pd = dil * (k - 1) - pad
        existing = self._grad_of(x_in)
        g = existing if existing is not None else self._buf(B, H, H, Cin)
        lowp = False
        if self.bf16_ops and existing is None and stride == 1 and x_in.data_ptr() in self._bn_outs16():
            # the trunk's fast bf16 conv kernels store bf16: usable where this data gradient is the ONE contribution to the producer's
            # d(out) and that map's one reader is the producer's BatchNorm backward (which takes a bf16 d(out))
            g16 = torch.empty(B, H, H, Cin, device=self.dev, dtype=torch.bfloat16)
            w16 = torch.empty(Cin, k * k * (Cout // groups), device=self.dev, dtype=torch.bfloat16)
            d, Hout, _ = ops.make_conv_desc(src, w16, g16, B=B, H=Hs, W=Hs, in_stride=Cout, cin_g=Cout // groups, Cout=Cin, groups=groups,
                                            k=k, pad=pd, dil=dil)
            thin = (groups == 4 and k == 3 and pd == 1 and dil == 1 and H * H >= 75 * 75
                    and (Cout // groups, Cin // groups) in ((16, 16), (32, 32)))                 # csrc/conv_thin_bf16.hip: conv1_2, conv2_2
            lowp = Hout == H and (thin or bool(lib.gssd_conv_flat_bf16_takes(C.byref(d))))       # csrc/conv_flat_bf16.hip: conv3_2 .. conv5_3
        if lowp:
            s16 = self._cast16(src)
            self.keep += [g16, w16]
            self._add(lib.gssd_cast_f32_bf16, (wd.data_ptr(), w16.data_ptr(), wd.numel()))
            d, Hout, _ = ops.make_conv_desc(s16, w16, g16, B=B, H=Hs, W=Hs, in_stride=Cout, cin_g=Cout // groups, Cout=Cin, groups=groups,
                                            k=k, pad=pd, dil=dil)
            self._add(lib.gssd_conv2d_nhwc_bf16, (C.byref(d),), keep=(d, s16, w16, g16))
            self.__dict__.setdefault('_g16', {})[x_in.data_ptr()] = g16
            return
        if self.bf16_ops and (Cout // groups) % 8 == 0 and Cout % 8 == 0:
            # bf16 storage mode: d(input) on the bf16 matrix cores -- d(output) and the flipped / transposed weight rounded to bf16 once,
            # fp32 accumulation, fp32 gradient map (an existing contribution is added in fp32)
            self._nt_bf16(src, wd, g, B=B, H=Hs, in_stride=Cout, cin_g=Cout // groups, Cout=Cin, groups=groups, k=k, pad=pd, dil=dil,
                          resid=existing, expect_H=H)
            self.gbuf[x_in.data_ptr()] = g
            return
        ud = None
        from .engine import USE_WINOGRAD
        if USE_WINOGRAD and ops.winograd_eligible(k, 1, pd, dil, Cout // groups, Cin // groups, groups):
            ud = self._buf(int(lib.gssd_winograd_weight_elems(Cin, groups, Cout // groups)))
            self._add(lib.gssd_winograd_weight_f32, (wd.data_ptr(), ud.data_ptr(), Cin, groups, Cout // groups, wd.stride(0)))
        x6d = self._x6w(wd, Cin, groups, Cout // groups, k, B * H * H, winograd=ud is not None)
        d, Hout, _ = ops.make_conv_desc(src, wd, g, B=B, H=Hs, W=Hs, in_stride=Cout, cin_g=Cout // groups, Cout=Cin,
                                        groups=groups, k=k, pad=pd, dil=dil, resid=existing, wgt_wino=ud, wgt_x6=x6d)
        assert Hout == H, (Hout, H)
        self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d),), keep=d)
        self.gbuf[x_in.data_ptr()] = g

    def _cast16(self, t):
        """bf16 copy of a FINAL fp32 gradient map (one cast launch where it is first asked for; the data-gradient conv and the weight
        gradient of a layer share it)."""
        c = self.__dict__.setdefault('_c16', {})
        k = t.data_ptr()
        if k not in c:
            t16 = torch.empty(t.shape, device=self.dev, dtype=torch.bfloat16)
            self.keep.append((t, t16))
            self._add(lib.gssd_cast_f32_bf16, (t.data_ptr(), t16.data_ptr(), t.numel()), leaf=False)
            c[k] = t16
        return c[k]

    def _nt_bf16(self, src, w, out, *, B, H, in_stride, cin_g, Cout, groups=1, k=1, pad=0, dil=1, resid=None, expect_H=None, gate=None):
        """out (fp32 NHWC) [+= resid] = conv(src, w) with both operands rounded to bf16 for the launch (fp32 accumulate): two cast
        launches + gssd_conv2d_nhwc_bf16 with GSSD_CONV_OUT_F32 (| GSSD_CONV_RESID_F32)."""
        s16 = self._cast16(src)
        w16 = torch.empty(w.shape, device=self.dev, dtype=torch.bfloat16)
        self.keep.append(w16)
        self._add(lib.gssd_cast_f32_bf16, (w.data_ptr(), w16.data_ptr(), w.numel()))
        d, Hout, _ = ops.make_conv_desc(s16, w16, out, B=B, H=H, W=H, in_stride=in_stride, cin_g=cin_g, Cout=Cout, groups=groups, k=k,
                                        pad=pad, dil=dil, resid=resid, gate=gate, wgt_row_stride=w.stride(0),
                                        flags=_lib.CONV_OUT_F32 | (_lib.CONV_RESID_F32 if resid is not None else 0))
        assert expect_H is None or Hout == expect_H, (Hout, expect_H)
        self._add(lib.gssd_conv2d_nhwc_bf16, (C.byref(d),), keep=(d, s16, w16))

    def _wgrad_nt_bf16(self, x, ld_x, cin, dy, ld_dy, cout, M, dwp, ld_w, groups=1):
        """dwp[n][k] (fp32, zero-filled each run) += sum_m dy[m][n] x[m][k] on the bf16 matrix cores: both operands transposed + rounded
        to bf16 (gssd_transpose_cast_f32_bf16: the reduction index m becomes the contiguous one), then one split-K NT launch of
        gssd_conv2d_nhwc_bf16 per group with the transposed d(output) as its "image" of ``cout`` one-pixel rows."""
        Mp = -(-M // 64) * 64
        dyT = torch.zeros(cout, Mp, device=self.dev, dtype=torch.bfloat16)          # columns [M, Mp) stay zero
        xT = torch.zeros(cin, Mp, device=self.dev, dtype=torch.bfloat16)
        self.keep += [dyT, xT]
        self._add(lib.gssd_transpose_cast_f32_bf16, (dy.data_ptr() if torch.is_tensor(dy) else dy, dyT.data_ptr(), M, cout, ld_dy, Mp),
                  leaf=True)
        self._add(lib.gssd_transpose_cast_f32_bf16, (x.data_ptr() if torch.is_tensor(x) else x, xT.data_ptr(), M, cin, ld_x, Mp), leaf=True)
        cg, ng = cin // groups, cout // groups
        tiles = -(-ng // 128) * -(-cg // (128 if cg > 64 else 64))
        split = int(max(1, min(64, Mp // 256, 2048 // (tiles * groups))))
        for g in range(groups):
            d, _, _ = ops.make_conv_desc(dyT[g * ng:], xT[g * cg:], dwp[g * ng:], B=1, H=1, W=ng, in_stride=Mp, cin_g=Mp, Cout=cg,
                                         wgt_row_stride=Mp, out_stride=ld_w, split_k=split, flags=_lib.CONV_OUT_F32)
            self._add(lib.gssd_conv2d_nhwc_bf16, (C.byref(d),), keep=(d, dyT, xT), leaf=True)

    def _wgrad_1x1_bf16(self, x16, ld_x, cin, dy16, cout, M, dwp, groups=1, in_xf=None, leaf=None):
        """dwp[cout][cin / groups] (fp32, zero-filled each run) += dy^T x over M rows, both operands bf16 in their natural [row][channel]
        layout: csrc/conv_wgrad_bf16.hip as a 1x1 "conv" over an (M / 16) x 16 pixel map.  False when the shape is not one of its."""
        if M % 16:
            return False
        d, _, _ = ops.make_conv_desc(x16, None, None, B=1, H=M // 16, W=16, in_stride=ld_x, cin_g=cin // groups, Cout=cout, groups=groups,
                                     in_scale=in_xf[0] if in_xf else None, in_shift=in_xf[1] if in_xf else None)
        if not lib.gssd_conv2d_wgrad_bf16_supported(C.byref(d)):
            return False
        self._add(lib.gssd_conv2d_wgrad_bf16, (C.byref(d), dy16.data_ptr(), dwp.data_ptr()), keep=(d, x16, dy16), leaf=leaf)
        return True

    def _wgrad(self, fdesc, dy, conv, cin_g_real, cin_g_pad, k, Cout, row0=0, param=None):
        """packed dW (zeroed each run) -> OIHW grad of ``param`` (rows [row0, row0 + param.shape[0]) of the packed matrix)."""
        K = k * k * cin_g_pad
        dwp = self._buf(Cout, K, zero_each_run=True)
        self._add(lib.gssd_conv2d_wgrad_f32, (C.byref(fdesc), dy.data_ptr(), dwp.data_ptr()), keep=fdesc)
        return dwp, K

    def _unpack(self, dwp, K, row0, param, cin_g_real, cin_g_pad, k):
        g = self._pgrad(param)
        n = param.shape[0]
        self._add(lib.gssd_unpack_conv_weight_grad, (dwp[row0:row0 + n].data_ptr(), g.data_ptr(), n, cin_g_real, k, k, cin_g_pad,
                                                     K, 0))

    def _bias_from_colsum(self, cs64, param, off=0):
        g = self._pgrad(param)
        self._add(lib.gssd_cast_f64_f32, (cs64[off:off + param.numel()].data_ptr(), g.data_ptr(), param.numel(), 0))

    # ------------------------------------------------------------------------------------------------
    def _head(self, r):
        B, H, Cs, A, nc = self.B, r['H'], r['C'], r['A'], self.plan.nc
        Cout = A * (4 + nc)
        dyh = self._buf(B, H, H, Cout)
        self._add(lib.gssd_heads_gather_f32, (self.dloc.data_ptr(), self.dconf.data_ptr(), dyh.data_ptr(), B, H * H, A, nc,
                                              self.plan.P, r['off']))
        Cp = ops.round_up(Cout, 8)
        d16 = None
        if self.bf16_ops and r.get('src16') is not None:
            d16, _, _ = ops.make_conv_desc(r['src16'], None, None, B=B, H=H, W=H, in_stride=Cs, cin_g=Cs, Cout=Cp, k=3, pad=1)
            if not lib.gssd_conv2d_wgrad_bf16_supported(C.byref(d16)):
                d16 = None
        if d16 is not None:
            # bf16 storage mode: the merged head gradient widened to a multiple of 8 channels (bf16, zero pad), both GEMMs on bf16
            dyh16 = torch.empty(B, H, H, Cp, device=self.dev, dtype=torch.bfloat16)
            self.keep.append(dyh16)
            self._add(lib.gssd_cast_rows_f32_bf16, (dyh.data_ptr(), dyh16.data_ptr(), B * H * H, Cout, Cout, Cp))
            self.__dict__.setdefault('_c16', {})[dyh.data_ptr()] = dyh16
            K = 9 * Cs
            dwp = self._buf(Cp, K, zero_each_run=True)
            self._add(lib.gssd_conv2d_wgrad_bf16, (C.byref(d16), dyh16.data_ptr(), dwp.data_ptr()), keep=d16, leaf=True)
        else:
            self._need(r['src'])
            fdesc, _, _ = ops.make_conv_desc(r['src'], None, None, B=B, H=H, W=H, in_stride=Cs, cin_g=Cs, Cout=Cout, k=3, pad=1)
            dwp, K = self._wgrad(fdesc, dyh, None, Cs, Cs, 3, Cout)
        self._unpack(dwp, K, 0, r['loc'].weight, Cs, Cs, 3)
        self._unpack(dwp, K, A * 4, r['conf'].weight, Cs, Cs, 3)
        cs = self._buf(Cout, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dyh.data_ptr(), B * H * H, Cout, Cout, cs.data_ptr()))
        self._bias_from_colsum(cs, r['loc'].bias, 0)
        self._bias_from_colsum(cs, r['conf'].bias, A * 4)
        # d(source): the merged head weight [Cout][9*Cs] viewed as one conv
        hw = self.plan.eng._packed[f"heads.{r['i']}.w"]          # packed forward rows [Cout][9*Cs] (k = tap*Cs + c)
        existing = self._grad_of(r['src'])
        g = existing if existing is not None else self._buf(B, H, H, Cs)
        if d16 is not None:
            wd = torch.zeros(Cs, 9 * Cp, device=self.dev)          # pad columns stay zero
            self.keep.append(wd)
            self._add(_pack_dgrad_from_packed, (hw, wd, Cout, Cs, Cp))
            self._nt_bf16(dyh, wd, g, B=B, H=H, in_stride=Cp, cin_g=Cp, Cout=Cs, k=3, pad=1, resid=existing, expect_H=H)
        else:
            wd = self._buf(Cs, 9 * Cout)
            self._add(_pack_dgrad_from_packed, (hw, wd, Cout, Cs))
            d, _, _ = ops.make_conv_desc(dyh, wd, g, B=B, H=H, W=H, in_stride=Cout, cin_g=Cout, Cout=Cs, k=3, pad=1, resid=existing)
            self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d),), keep=d)
        self.gbuf[r['src'].data_ptr()] = g

    def _convbn(self, r, need_dgrad=True):
        B, H, Ho, Hp, Cin, Cout, groups = self.B, r['H'], r['Ho'], r['Hp'], r['Cin'], r['Cout'], r['groups']
        conv, bn, raw = r['conv'], r['bn'], r['raw']
        dout16 = self.__dict__.get('_g16', {}).pop(r['out'].data_ptr(), None)      # d(out) as a bf16 map (see _dgrad): this is its reader
        dout = dout16 if dout16 is not None else self._grad_of(r['out'])
        if dout is None:
            raise _lib.GssdError(f"no gradient reaches {r['name']}")
        # scale / shift of this layer's BatchNorm (deferred layers already hold them from the forward)
        if r['xf'] is not None:
            sc, sh = r['xf'][0], r['xf'][1]
        else:
            sc, sh, pd_ = self._buf(Cout), self._buf(Cout), self._buf(Cout)
            self._add(lib.gssd_bn_finalize_f32, (r['stats'].data_ptr(), float(B * Ho * Ho), bn.weight.data_ptr(),
                                                 bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                 float(bn.momentum), float(bn.eps), 2, Cout, sc.data_ptr(), sh.data_ptr(),
                                                 pd_.data_ptr(), r.get('stats_rep', 0)))
        pool = r['pool']
        pk, ps, pp = (pool[0], pool[1], pool[2]) if pool else (0, 1, 0)
        dz = self._buf(B, Ho, Ho, Cout, zero_each_run=bool(pool and ps < pk))
        sums = self._buf(2 * Cout, dtype=torch.float64, zero_each_run=True)
        # bf16 storage mode: which precision do the consumers of d(pre-activation) read?  (weight gradient: csrc/conv_wgrad_bf16.hip for the
        # grouped 3x3 trunk shapes; data gradient: gssd_conv2d_nhwc_bf16 when the channel counts allow 16-byte rows)
        cin_g_pad = Cin // groups
        cin_g_real = conv.weight.shape[1]
        d16 = None
        if self.bf16_ops and r.get('x16') is not None and cin_g_real == cin_g_pad:
            ix = r['in_xf']
            flat = r['k'] == 1 and r['stride'] == 1 and r['pad'] == 0 and (B * H * H) % 16 == 0    # 1x1: the map as (M / 16) x 16 pixels
            d16, _, _ = ops.make_conv_desc(r['x16'], None, None, B=1 if flat else B, H=B * H * H // 16 if flat else H, W=16 if flat else H,
                                           in_stride=r['Cin16'], cin_g=r['Cin16'] // groups, Cout=Cout,
                                           groups=groups, k=r['k'], stride=r['stride'], pad=r['pad'], dil=r['dil'],
                                           in_scale=ix[0] if ix else None, in_shift=ix[1] if ix else None)
            if not lib.gssd_conv2d_wgrad_bf16_supported(C.byref(d16)):
                d16 = None
        dg16 = bool(need_dgrad and self.bf16_ops and (Cout // groups) % 8 == 0 and Cout % 8 == 0 and r['stride'] == 1)   # (stride 2: the
        # zero-insertion pass reads the fp32 map)
        raw16 = r.get('raw16') if self.bf16_ops else None
        mixed = raw16 is not None                                 # BatchNorm backward reads the forward's bf16 map itself
        want16 = mixed and (d16 is not None or dg16)
        want32 = (d16 is None) or (need_dgrad and not dg16) or not mixed
        dz16 = torch.empty(B, Ho, Ho, Cout, device=self.dev, dtype=torch.bfloat16) if want16 else None
        if want16:
            self.keep.append(dz16)
            self.__dict__.setdefault('_c16', {})[dz.data_ptr()] = dz16      # _cast16(dz) finds it: no cast launch
        if not mixed:
            self._need(raw)
        # without pooling the reduce pass only sums (dz = NULL) and the apply pass re-derives dz from d(out): 5 instead of 6 HBM passes
        # mixed + a non-overlapping pool: the routed (un-pooled) gradient travels from the reduce pass to the apply pass as a bf16 map
        dzp16 = torch.empty(B, Ho, Ho, Cout, device=self.dev, dtype=torch.bfloat16) if (mixed and pool and ps >= pk) else None
        if mixed:
            self._add(lib.gssd_bn_bwd_reduce_mixed,
                      (dout.data_ptr(), int(dout16 is not None), raw16.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                       dz.data_ptr() if (pool and dzp16 is None) else 0, dzp16.data_ptr() if dzp16 is not None else 0, sums.data_ptr(),
                       B, Ho, Ho, Cout, Hp, Hp, pk, ps, pp, int(r['relu'])), keep=dzp16)
        else:
            self._add(lib.gssd_bn_bwd_reduce_f32,
                      (dout.data_ptr(), raw.data_ptr(), sc.data_ptr(), sh.data_ptr(), dz.data_ptr() if pool else 0, sums.data_ptr(), B, Ho, Ho,
                       Cout, Hp, Hp, pk, ps, pp, int(r['relu'])))
        ca, cb, cc = self._buf(Cout), self._buf(Cout), self._buf(Cout)
        self._add(lib.gssd_bn_bwd_finalize_f32, (r['stats'].data_ptr(), float(B * Ho * Ho), sums.data_ptr(), bn.weight.data_ptr(),
                                                 float(bn.eps), Cout, ca.data_ptr(), cb.data_ptr(), cc.data_ptr(),
                                                 self._pgrad(bn.weight).data_ptr(), self._pgrad(bn.bias).data_ptr(), r.get('stats_rep', 0)))
        cs = self._buf(Cout, dtype=torch.float64, zero_each_run=True)
        if mixed:
            if dzp16 is not None:          # d = the bf16 map the reduce pass routed through the pool (already masked)
                src, src16, msk = dzp16.data_ptr(), 1, False
            elif pool:                     # d = dz as the reduce pass wrote it (overlapping windows: fp32 atomics)
                src, src16, msk = 0, 0, False
            else:                          # d = d(out) o [pre-activation > 0], re-derived here
                src, src16, msk = dout.data_ptr(), int(dout16 is not None), True
            self._add(lib.gssd_bn_bwd_apply_mixed, (src, src16, dz.data_ptr(), dz16.data_ptr() if want16 else 0, raw16.data_ptr(),
                                                    sc.data_ptr() if msk else 0, sh.data_ptr() if msk else 0, int(r['relu']) if msk else 0,
                                                    ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), B * Ho * Ho, Cout, cs.data_ptr(),
                                                    int(want32)), keep=(raw16, dz16))
        elif pool:
            self._add(lib.gssd_bn_bwd_apply_f32, (dz.data_ptr(), raw.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(),
                                                  B * Ho * Ho, Cout, cs.data_ptr()))
        else:
            self._add(lib.gssd_bn_bwd_apply_masked_f32, (dout.data_ptr(), raw.data_ptr(), sc.data_ptr(), sh.data_ptr(), int(r['relu']),
                                                         ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), dz.data_ptr(), B * Ho * Ho, Cout,
                                                         cs.data_ptr()))
        self._bias_from_colsum(cs, conv.bias)
        # weight gradient (the forward descriptor carries the input geometry and the fused input transform)
        if d16 is not None:
            # bf16 storage mode, grouped 3x3 trunk layers: the STORED bf16 input and the bf16-rounded d(pre-activation) on the bf16
            # matrix cores (fp32 accumulation, fp32 gradient); 3 - 6 x the fp32 kernels
            K = r['k'] * r['k'] * cin_g_pad
            dwp = self._buf(Cout, K, zero_each_run=True)
            dz16 = self._cast16(dz)
            self._add(lib.gssd_conv2d_wgrad_bf16, (C.byref(d16), dz16.data_ptr(), dwp.data_ptr()), keep=(d16, dz16), leaf=True)
        else:
            self._need(r['x_in'])
            dwp, K = self._wgrad(r['desc'], dz, conv, cin_g_real, cin_g_pad, r['k'], Cout)
        self._unpack(dwp, K, 0, conv.weight, cin_g_real, cin_g_pad, r['k'])
        if need_dgrad:
            self._dgrad(r, dz, r['x_in'], conv, groups, Cin, H, Ho, Cout, r['k'], r['stride'], r['pad'], r['dil'])

    def _bn_outs16(self):
        """Outputs of conv + BatchNorm layers whose backward reads bf16 maps (gssd_bn_bwd_*_mixed takes a bf16 d(out) too)."""
        s = self.__dict__.get('_bn_outs_set')
        if s is None:
            s = set()
            if self.bf16_ops:
                # consumers per stored map: only a map with ONE reader (the next conv) may get a bf16 gradient
                n = {}
                for kind, q in self.plan.rec:
                    for key in ('x_in', 'src', 'a', 'b'):
                        t = q.get(key)
                        if torch.is_tensor(t):
                            n[t.data_ptr()] = n.get(t.data_ptr(), 0) + 1
                s = {q['out'].data_ptr() for kind, q in self.plan.rec
                     if kind == 'convbn' and q.get('raw16') is not None and n.get(q['out'].data_ptr(), 0) == 1}
            self._bn_outs_set = s
        return s

    def _need(self, t):
        """A step reads the fp32 CONTENT of a stored map: schedule its cast if the bf16 shadow plan shadows it lazily."""
        n = getattr(self.plan, 'need', None)
        if n is not None:
            n(t)

    def _convrelu(self, r, need_dgrad=True):
        """conv + ReLU without BatchNorm (vanilla SSD, models/ssd.py:104-118; the grouped batch_norm=False graph): dz = d(out) * [out > 0] (the mask of the
        stored post-ReLU output equals the pre-activation's), bias gradient = column sums, then wgrad / dgrad."""
        B, H, Ho, Cin, Cout, conv, out = self.B, r['H'], r['Ho'], r['Cin'], r['Cout'], r['conv'], r['out']
        dout = self._grad_of(out)
        if dout is None:
            raise _lib.GssdError(f"no gradient reaches {r['name']}")
        if r.get('relu', True):
            dz = self._buf(B, Ho, Ho, Cout)
            self._add(lib.gssd_bn_bwd_reduce_f32, (dout.data_ptr(), out.data_ptr(), 0, 0, dz.data_ptr(), 0, B, Ho, Ho, Cout, Ho, Ho,
                                                   0, 1, 0, 1))
        else:
            dz = dout                    # conv + bias only (PixelLink++ fuse conv without BatchNorm): no mask
        cs = self._buf(Cout, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dz.data_ptr(), B * Ho * Ho, Cout, Cout, cs.data_ptr()))
        self._bias_from_colsum(cs, conv.bias)
        groups = r.get('groups', 1)
        cin_g_real, cin_g_pad = conv.weight.shape[1], Cin // groups
        dwp, K = self._wgrad(r['desc'], dz, conv, cin_g_real, cin_g_pad, r['k'], Cout)
        self._unpack(dwp, K, 0, conv.weight, cin_g_real, cin_g_pad, r['k'])
        if need_dgrad:
            self._dgrad(r, dz, r['x_in'], conv, groups, Cin, H, Ho, Cout, r['k'], r['stride'], r['pad'], r['dil'])

    def _pool(self, r):
        B, H, Cc, Hp = self.B, r['H'], r['C'], r['Hp']
        dout = self._grad_of(r['out'])
        existing = self._grad_of(r['x_in'])
        g = self._buf(B, H, H, Cc, zero_each_run=(r['s'] < r['k']))
        self._add(lib.gssd_bn_bwd_reduce_f32, (dout.data_ptr(), r['x_in'].data_ptr(), 0, 0, g.data_ptr(), 0, B, H, H, Cc, Hp, Hp,
                                               r['k'], r['s'], r['p'], 0))
        if existing is not None:         # (a hoisted branch hanging off the same activation -- L2Norm on conv4_3 -- wrote first)
            self._add(lib.gssd_axpby_f32, (existing.data_ptr(), g.data_ptr(), existing.data_ptr(), B * H * H * Cc, 1.0, 1.0))
            g = existing
        self.gbuf[r['x_in'].data_ptr()] = g

    def _l2norm(self, r):
        B, H, Cc, mod = self.B, r['H'], r['C'], r['mod']
        dy = self._grad_of(r['out'])
        existing = self._grad_of(r['x_in'])
        g = existing if existing is not None else self._buf(B, H, H, Cc)
        dw = self._buf(Cc, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_l2norm_bwd_f32, (r['x_in'].data_ptr(), mod.weight.data_ptr(), dy.data_ptr(), g.data_ptr(),
                                            existing.data_ptr() if existing is not None else 0, dw.data_ptr(), B * H * H, Cc,
                                            float(mod.eps)))
        self._bias_from_colsum(dw, mod.weight)
        self.gbuf[r['x_in'].data_ptr()] = g

    def _relupool(self, r):
        """ReLU and / or max-pool pass of the PixelLink++ trunk (identity-affine bn_relu_pool launch): first-maximum routing, ReLU mask."""
        B, H, Cc, Hp = self.B, r['H'], r['C'], r['Hp']
        dout = self._grad_of(r['out'])
        if dout is None:
            raise _lib.GssdError('no gradient reaches a ReLU / pool pass of the PixelLink++ trunk')
        existing = self._grad_of(r['x_in'])
        g = self._buf(B, H, H, Cc, zero_each_run=bool(r['k'] and r['s'] < r['k']))
        self._add(lib.gssd_bn_bwd_reduce_f32, (dout.data_ptr(), r['x_in'].data_ptr(), 0, 0, g.data_ptr(), 0, B, H, H, Cc, Hp, Hp,
                                               r['k'], r['s'], r['p'], int(r['relu'])))
        if existing is not None:
            self._add(lib.gssd_axpby_f32, (existing.data_ptr(), g.data_ptr(), existing.data_ptr(), B * H * H * Cc, 1.0, 1.0))
            g = existing
        self.gbuf[r['x_in'].data_ptr()] = g

    def _pl_grad(self, t, H):
        """Gradient map (channel stride PL_LD, zeroed every run: the cascade accumulates into it) of an 18-channel score map."""
        g = self.gbuf.get(t.data_ptr())
        if g is None:
            g = self._buf(self.B, H, H, self.PL_LD, zero_each_run=True)
            self.gbuf[t.data_ptr()] = g
        return g

    def _plfinal(self, r):
        """final_1 / final_2 (model.py:360,386 / 396,411): d(features), weight and bias gradients; the roots d(out_1), d(out_2)."""
        B, H, feats, f1, f2 = self.B, r['H'], r['feats'], r['final_1'], r['final_2']
        nf = len(feats)
        gs = [self._pl_grad(f, H) for f in feats]
        dw1 = self._buf(2 * 2 * nf + 2, dtype=torch.float64, zero_each_run=True)
        dw2 = self._buf(16 * 16 * nf + 16, dtype=torch.float64, zero_each_run=True)
        w1, w2 = f1.weight.detach().view(2, -1), f2.weight.detach().view(16, -1)
        fp = [f.data_ptr() for f in feats] + [0] * (4 - nf)
        gp = [g.data_ptr() for g in gs] + [0] * (4 - nf)
        self._add(lib.gssd_pixellink_final_bwd_f32, (self.d_out1.data_ptr(), self.d_out2.data_ptr(), *fp, nf, w1.data_ptr(), w2.data_ptr(),
                                                     *gp, 0, dw1.data_ptr(), dw2.data_ptr(), B, H * H, self.PL_LD), keep=(w1, w2))
        for dw, mod, nw in ((dw1, f1, 4 * nf), (dw2, f2, 256 * nf)):
            self._add(lib.gssd_cast_f64_f32, (dw.data_ptr(), self._pgrad(mod.weight).data_ptr(), nw, 0))
            self._add(lib.gssd_cast_f64_f32, (dw[nw:].data_ptr(), self._pgrad(mod.bias).data_ptr(), mod.bias.numel(), 0))

    def _interp(self, r):
        """out = interp(src) [, out2 = out + addend]: d(src) += interp^T(d(out) + d(out2)), d(addend) += d(out2)."""
        B, Hs, Hd = self.B, r['Hs'], r['Hd']
        go = self.gbuf.get(r['out'].data_ptr())
        go2 = self.gbuf.get(r['out2'].data_ptr()) if r['out2'] is not None else None
        if go is None and go2 is None:
            return
        gsrc = self._pl_grad(r['src'], Hs)
        gadd = self._pl_grad(r['addend'], Hd) if (r['addend'] is not None and go2 is not None) else None
        self._add(lib.gssd_interp_add_bwd_f32, (go.data_ptr() if go is not None else 0, go2.data_ptr() if go2 is not None else 0,
                                                gsrc.data_ptr(), gadd.data_ptr() if gadd is not None else 0, B, Hs, Hs, Hd, Hd, 18,
                                                self.PL_LD))

    def _plhead(self, r):
        """The merged 1x1 score heads out{k}_1 | out{k}_2 (18 channels, run on PL_LD = 20 in the backward: rows 18, 19 are zero)."""
        B, H, Cs, LD = self.B, r['H'], r['C'], self.PL_LD
        o1, o2, src = r['o1'], r['o2'], r['src']
        dy = self.gbuf.get(r['out'].data_ptr())
        if dy is None:
            raise _lib.GssdError(f"no gradient reaches the score heads of stage {r['k']}")
        fdesc, _, _ = ops.make_conv_desc(src, None, None, B=B, H=H, W=H, in_stride=Cs, cin_g=Cs, Cout=LD, k=1)
        dwp, K = self._wgrad(fdesc, dy, None, Cs, Cs, 1, LD)
        self._unpack(dwp, K, 0, o1.weight, Cs, Cs, 1)
        self._unpack(dwp, K, 2, o2.weight, Cs, Cs, 1)
        cs = self._buf(LD, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dy.data_ptr(), B * H * H, LD, LD, cs.data_ptr()))
        self._bias_from_colsum(cs, o1.bias, 0)
        self._bias_from_colsum(cs, o2.bias, 2)
        # d(src) = dy . W  (1x1 conv over dy with the transposed merged weight [Cs][LD], refreshed every run)
        wd = torch.zeros(Cs, LD, device=self.dev)
        self.keep.append(wd)

        def refresh(wd=wd, o1=o1, o2=o2, Cs=Cs):
            wd[:, :2].copy_(o1.weight.detach().view(2, Cs).t())
            wd[:, 2:18].copy_(o2.weight.detach().view(16, Cs).t())
        self.steps.append((refresh, None))
        existing = self._grad_of(src)
        g = existing if existing is not None else self._buf(B, H, H, Cs)
        d, _, _ = ops.make_conv_desc(dy, wd, g, B=B, H=H, W=H, in_stride=LD, cin_g=LD, Cout=Cs, k=1, resid=existing)
        self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d),), keep=d)
        self.gbuf[src.data_ptr()] = g

    # ---- GSSD++ blocks ---------------------------------------------------------------------------------------------------------
    def _reserve(self, t, shape):
        """Gradient buffer of forward tensor ``t`` (created on first use); returns (buffer, existed_before)."""
        buf = self.gbuf.get(t.data_ptr())
        if buf is not None:
            return buf, True
        buf = self._buf(*shape)
        self.gbuf[t.data_ptr()] = buf
        return buf, False

    def _sa(self, r):
        """Self_Attn (layers/self_attn.py:46-89), all HIP (csrc/sa_backward.hip + the conv kernels).  With T = d(out) + d(out2),
        s = sigma, W_eff = W / sigma_sn (alpha = 1 / sigma_sn per conv; u, v constants of the step):
            d(ag)' = T W_o^T alpha_o              (1x1 conv over T with the scaled, transposed o weights; everything below is
                                                   linear in d(ag) = s d(ag)', so s is applied where the chain ends)
            d sigma = <d(ag)', ag> + <b_o, colsum T>;   d b_o = s colsum T;   dW_o = SN(s T^T ag)
            A = softmax(theta phi^T)              (the forward is flash-style and keeps no map: re-materialised here)
            dA = d(ag)' g;  dS = A (dA - rowsum(A dA));  d theta = dS phi;  d phi = dS^T theta;  d g = A^T d(ag)'   (batched GEMMs)
            dW_{theta,phi,g} = SN(s [d theta | d phi | d g]^T x);  biases = s colsum;  dx = d(out) + s [d theta | d phi | d g] W_tpg alpha
        SN(G) = G / sigma_sn - <G, W> / sigma_sn^2 u v^T."""
        sa, x, out, out2 = r['mod'], r['x_in'], r['out'], r['out2']
        g_out = self._grad_of(out)
        g_out2 = self._grad_of(out2) if out2 is not None else None
        gx, existed = self._reserve(x, x.shape)
        B, N, Np, Cc, H = self.B, r['N'], r['Np'], r['C'], r['H']
        C8, C2, C4 = Cc // 8, Cc // 2, Cc // 4
        CT = C4 + C2
        M = B * N
        tp, gT, ag = r['tp'], r['gT'], r['ag']
        a_tpg, a_o = r['inv_sigma']
        cv = {k: getattr(sa, 'snconv1x1_' + k) for k in ('theta', 'phi', 'g', 'attn')}
        name = r['name']
        w_tpg = self.plan.eng._packed[name + '.tpg.w']                       # fp32 [C4 + C2][C]: theta | phi | g rows
        w_o = cv['attn'].weight_orig.detach().view(Cc, C2)
        sig = sa.sigma
        mk = ops.make_conv_desc
        fn = lib.gssd_conv2d_nhwc_f32
        # T
        T = g_out
        if g_out2 is not None:
            T = self._buf(B, H, H, Cc)
            self._add(lib.gssd_axpby_f32, (g_out.data_ptr(), g_out2.data_ptr(), T.data_ptr(), M * Cc, 1.0, 1.0))
        # d(ag)' = T . (W_o^T alpha_o)
        wd_o = self._buf(C2, Cc)
        self._add(lib.gssd_scaled_transpose_f32, (w_o.data_ptr(), a_o.data_ptr(), wd_o.data_ptr(), Cc, C2), keep=w_o)
        dag = self._buf(B, N, C2)
        if self.bf16_ops:             # the block's four dense GEMMs over the tokens on the bf16 matrix cores (fp32 accumulation)
            self._nt_bf16(T, wd_o, dag, B=B, H=H, in_stride=Cc, cin_g=Cc, Cout=C2)
        else:
            d_dag, _, _ = mk(T, wd_o, dag, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C2, wgt_x6=self._x6w(wd_o, C2, 1, Cc, 1, B * N))
            self._add(fn, (C.byref(d_dag),), keep=d_dag)
        # sigma, o bias, o weight
        dot = self._buf(1, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_dot_f32, (dag.data_ptr(), ag.data_ptr(), M * C2, dot.data_ptr()))
        csT = self._buf(Cc, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (T.data_ptr(), M, Cc, Cc, csT.data_ptr()))
        self._add(lib.gssd_sa_sigma_grad_f32, (dot.data_ptr(), csT.data_ptr(), cv['attn'].bias.data_ptr(), Cc, self._pgrad(sig).data_ptr()))
        self._add(lib.gssd_scale_cast_f64_f32, (csT.data_ptr(), sig.data_ptr(), self._pgrad(cv['attn'].bias).data_ptr(), Cc))
        d_o, _, _ = mk(ag, None, None, B=B, H=H, W=H, in_stride=C2, cin_g=C2, Cout=Cc)
        dwo = self._buf(Cc, C2, zero_each_run=True)
        sndot = self._buf(4, dtype=torch.float64, zero_each_run=True)      # <dW_eff, W> of the block's four convs
        if not (self.bf16_ops and r.get('ag16') is not None and self._wgrad_1x1_bf16(r['ag16'], C2, C2, self._cast16(T), Cc, M, dwo)):
            self._add(lib.gssd_conv2d_wgrad_f32, (C.byref(d_o), T.data_ptr(), dwo.data_ptr()), keep=d_o)
        self._add(lib.gssd_sn_weight_grad_f32, (dwo.data_ptr(), C2, cv['attn'].weight_orig.data_ptr(), cv['attn'].weight_u.data_ptr(),
                                                cv['attn'].weight_v.data_ptr(), a_o.data_ptr(), sig.data_ptr(), sndot[3:].data_ptr(),
                                                self._pgrad(cv['attn'].weight_orig).data_ptr(), Cc, C2))
        # attention map A (no stored copy: the forward is flash-style).  Keys / values: phi / g of the same tokens, or their P x P
        # average-pooled copies (max_pool_factor > 1) -- then the key-side gradients come out per cell and are un-pooled below
        Nk, Nkp, pooled = r['Nk'], r['Nkp'], r['kp'] is not None
        keys, krow, vals = (r['kp'], C8, r['gTp']) if pooled else (tp[0, 0, C8:], C4, gT)
        lse = r.get('lse')
        flash = self.bf16_ops and r.get('g16') is not None and lse is not None and not pooled
        dtpg = self._buf(B, N, CT)
        if flash:
            # bf16 storage mode: d theta | d phi | d g in two launches of the flash-style kernel, no [N][N] map
            Dv = self._buf(B, N)
            self._add(lib.gssd_rowdot_f32, (dag.data_ptr(), ag.data_ptr(), Dv.data_ptr(), M, C2))
            # theta | phi as a two-term bf16 split: the logits are recomputed on the bf16 matrix cores to 2^-16 (GSSD_FLASH_BWD_X3=0: fp32 MFMA)
            tph = torch.empty(tp.shape, device=self.dev, dtype=torch.bfloat16)
            tpl = torch.empty(tp.shape, device=self.dev, dtype=torch.bfloat16)
            self._add(lib.gssd_cast_split_f32_bf16, (tp.data_ptr(), tph.data_ptr(), tpl.data_ptr(), tp.numel()))
            x3 = os.environ.get('GSSD_FLASH_BWD_X3', '1') != '0'
            self._add(lib.gssd_self_attn_flash_bwd_bf16, (tp.data_ptr(), tph.data_ptr(), tpl.data_ptr() if x3 else 0, r['g16'].data_ptr(),
                                                          self._cast16(dag).data_ptr(), lse.data_ptr(), Dv.data_ptr(), dtpg.data_ptr(),
                                                          B, N, C8, C2), keep=(r['g16'], lse, tph, tpl))
        else:
            self._sa_explicit(r, dtpg, dag, ag, tp, keys, krow, vals, lse, pooled)
        self._sa_tail(r, dtpg, g_out, gx, existed, x, a_tpg, w_tpg, cv, sig, sndot)

    def _sa_explicit(self, r, dtpg, dag, ag, tp, keys, krow, vals, lse, pooled):
        B, N, Np, Cc, H = self.B, r['N'], r['Np'], r['C'], r['H']
        C8, C2, C4 = Cc // 8, Cc // 2, Cc // 4
        CT = C4 + C2
        M = B * N
        Nk, Nkp = r['Nk'], r['Nkp']
        mk = ops.make_conv_desc
        fn = lib.gssd_conv2d_nhwc_f32
        A = self._buf(B, N, Nkp)
        dA = self._buf(B, N, Nkp)
        if lse is not None:
            # A = exp(theta . keys^T - lse) and dS = A o (d(ag)' . values - D), D_i = <d(ag)'_i, ag_i> = rowsum(A o dA): both in the
            # epilogue of the GEMM that produces the logits / dA -- no pass over the [N, Nk] maps for softmax or its backward
            self._add(lib.gssd_bgemm_ex_f32, (tp.data_ptr(), keys.data_ptr(), A.data_ptr(), N, Nk, C8, C4, krow, Nkp, 0, 1, N * C4, Nk * krow,
                                              N * Nkp, B, 1.0, 1, lse.data_ptr(), 0))
            Dv = self._buf(B, N)
            self._add(lib.gssd_rowdot_f32, (dag.data_ptr(), ag.data_ptr(), Dv.data_ptr(), M, C2))
            self._add(lib.gssd_bgemm_ex_f32, (dag.data_ptr(), vals.data_ptr(), dA.data_ptr(), N, Nk, C2, C2, Nkp, Nkp, 0, 0, N * C2, C2 * Nkp,
                                              N * Nkp, B, 1.0, 2, Dv.data_ptr(), A.data_ptr()))
        else:
            d_qk, _, _ = mk(tp, keys, A, B=B, H=H, W=H, in_stride=C4, cin_g=C8, Cout=Nk, out_stride=Nkp, m_per_image=True,
                            in_batch_stride=N * C4, wgt_batch_stride=Nk * krow, out_batch_stride=N * Nkp, wgt_row_stride=krow)
            self._add(fn, (C.byref(d_qk),), keep=d_qk)
            self._add(lib.gssd_softmax_rows_f32, (A.data_ptr(), B * N, Nk, Nkp))
            # dA = d(ag)' . g ;  dS in place
            self._add(lib.gssd_bgemm_f32, (dag.data_ptr(), vals.data_ptr(), dA.data_ptr(), N, Nk, C2, C2, Nkp, Nkp, 0, 0, N * C2, C2 * Nkp,
                                           N * Nkp, B, 1.0, 0))
            self._add(lib.gssd_softmax_bwd_rows_f32, (A.data_ptr(), dA.data_ptr(), B * N, Nk, Nkp))
        # [d theta | d phi | d g] token-major, one buffer (the gradient of the merged projection's output)
        self._add(lib.gssd_bgemm_f32, (dA.data_ptr(), keys.data_ptr(), dtpg.data_ptr(), N, C8, Nk, Nkp, krow, CT, 0, 0, N * Nkp, Nk * krow,
                                       N * CT, B, 1.0, 0))
        if pooled:
            CW = C8 + C2
            dkg = self._buf(B, Nk, CW)                                        # d(pooled phi) | d(pooled g) per cell
            self._add(lib.gssd_bgemm_f32, (dA.data_ptr(), tp.data_ptr(), dkg.data_ptr(), Nk, C8, N, Nkp, C4, CW, 1, 0, N * Nkp, N * C4,
                                           Nk * CW, B, 1.0, 0))
            self._add(lib.gssd_bgemm_f32, (A.data_ptr(), dag.data_ptr(), dkg[0, 0, C8:].data_ptr(), Nk, C2, N, Nkp, C2, CW, 1, 0, N * Nkp,
                                           N * C2, Nk * CW, B, 1.0, 0))
            self._add(lib.gssd_sa_unpool_f32, (dkg.data_ptr(), dtpg[0, 0, C8:].data_ptr(), B, H, r['P'], CW, CT))
        else:
            self._add(lib.gssd_bgemm_f32, (dA.data_ptr(), tp.data_ptr(), dtpg[0, 0, C8:].data_ptr(), N, C8, N, Np, C4, CT, 1, 0, N * Np,
                                           N * C4, N * CT, B, 1.0, 0))
            self._add(lib.gssd_bgemm_f32, (A.data_ptr(), dag.data_ptr(), dtpg[0, 0, C4:].data_ptr(), N, C2, N, Np, C2, CT, 1, 0, N * Np,
                                           N * C2, N * CT, B, 1.0, 0))

    def _sa_tail(self, r, dtpg, g_out, gx, existed, x, a_tpg, w_tpg, cv, sig, sndot):
        B, N, Cc, H = self.B, r['N'], r['C'], r['H']
        C8, C2, C4 = Cc // 8, Cc // 2, Cc // 4
        CT = C4 + C2
        M = B * N
        mk = ops.make_conv_desc
        fn = lib.gssd_conv2d_nhwc_f32
        # projection weights / biases
        d_p, _, _ = mk(x, None, None, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=CT)
        dwp = self._buf(CT, Cc, zero_each_run=True)
        if not (self.bf16_ops and r.get('x16') is not None and self._wgrad_1x1_bf16(r['x16'], Cc, Cc, self._cast16(dtpg), CT, M, dwp)):
            self._need(x)
            self._add(lib.gssd_conv2d_wgrad_f32, (C.byref(d_p), dtpg.data_ptr(), dwp.data_ptr()), keep=d_p)
        csP = self._buf(CT, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dtpg.data_ptr(), M, CT, CT, csP.data_ptr()))
        for si, (key, row0, rows) in enumerate((('theta', 0, C8), ('phi', C8, C8), ('g', C4, C2))):
            m_ = cv[key]
            self._add(lib.gssd_sn_weight_grad_f32, (dwp[row0:].data_ptr(), Cc, m_.weight_orig.data_ptr(), m_.weight_u.data_ptr(),
                                                    m_.weight_v.data_ptr(), a_tpg[row0:].data_ptr(), sig.data_ptr(),
                                                    sndot[si:].data_ptr(), self._pgrad(m_.weight_orig).data_ptr(), rows, Cc))
            self._add(lib.gssd_scale_cast_f64_f32, (csP[row0:].data_ptr(), sig.data_ptr(), self._pgrad(m_.bias).data_ptr(), rows))
        # dx = d(out) (+ what was already there) + sigma * dtpg . (W_tpg alpha)
        wd_p = self._buf(Cc, CT)
        self._add(lib.gssd_scaled_transpose_f32, (w_tpg.data_ptr(), a_tpg.data_ptr(), wd_p.data_ptr(), CT, Cc), keep=w_tpg)
        resid = g_out
        if existed:
            resid = self._buf(B, H, H, Cc)
            self._add(lib.gssd_axpby_f32, (g_out.data_ptr(), gx.data_ptr(), resid.data_ptr(), M * Cc, 1.0, 1.0))
        if self.bf16_ops:
            self._nt_bf16(dtpg, wd_p, gx, B=B, H=H, in_stride=CT, cin_g=CT, Cout=Cc, gate=sig.detach(), resid=resid)
            self.keep.append(sig)
        else:
            d_dx, _, _ = mk(dtpg, wd_p, gx, B=B, H=H, W=H, in_stride=CT, cin_g=CT, Cout=Cc, gate=sig.detach(), resid=resid,
                            wgt_x6=self._x6w(wd_p, Cc, 1, CT, 1, B * H * H))
            self._add(fn, (C.byref(d_dx),), keep=(d_dx, sig))

    def _slice_cat(self, r):
        a, b, out, groups, Ca, Cb = r['a'], r['b'], r['out'], r['groups'], r['Ca'], r['Cb']
        g_out = self._grad_of(out)
        ga, a_existed = self._reserve(a, a.shape)
        gb, b_existed = (None, False) if r['detach_b'] else self._reserve(b, b.shape)
        ca, cb = Ca // groups, Cb // groups

        def step():
            v = g_out.view(*g_out.shape[:-1], groups, ca + cb)
            da = v[..., :ca].reshape(a.shape)
            ga.add_(da) if a_existed else ga.copy_(da)
            if gb is not None:
                db = v[..., ca:].reshape(b.shape)
                gb.add_(db) if b_existed else gb.copy_(db)
        self.steps.append((step, None))

    def _dcn(self, r):
        """Modulated deformable conv (layers/dcn_v2_custom.py:79-89): the 1x1 GEMM over the sampled columns, the sampling
        itself (gssd_dcn_col2im_f32) and the offset/mask conv, all HIP."""
        B, H, Cin, Cout, dg, m = self.B, r['H'], r['Cin'], r['Cout'], r['dg'], r['mod']
        x, om = r['x_in'], r['om']
        dy = self._grad_of(r['out'])
        Kc = 9 * Cin
        # the fused forward keeps no column matrix: rebuild it here for the weight gradient
        OMC = r['omc']                           # channel stride of the offset / mask rows (27 * dg rounded up to 4; 8 in bf16 mode)
        M = B * H * H
        cols16 = None
        if self.bf16_ops and r.get('x16') is not None and M % 16 == 0:
            # bf16 storage mode: the columns in bf16 from the bf16 map the forward sampled; the weight gradient reads them as they are
            cols16 = torch.empty(M, Kc, device=self.dev, dtype=torch.bfloat16)
            self.keep.append(cols16)
            self._add(lib.gssd_dcn_im2col_bf16, (r['x16'].data_ptr(), om.data_ptr(), cols16.data_ptr(), B, H, H, Cin, dg, OMC), leaf=True)
        else:
            cols = self._buf(B * H * H, Kc)
            self._add(lib.gssd_dcn_im2col_f32, (x.data_ptr(), om.data_ptr(), cols.data_ptr(), B, H, H, Cin, dg, OMC))
        w_main = self._buf(Cout, Kc)
        self._add(lib.gssd_pack_conv_weight, (m.weight.data_ptr(), w_main.data_ptr(), Cout, Cin, 3, 3, Cin, Kc), keep=m)
        # main weight / bias: dW[Cout][9*Cin] = dY^T . cols, d(cols) = dY . W  -- the slot-scheduled TN / NT GEMMs (csrc/wgrad_slot.hip, csrc/gemm_slot.hip; no vendor library)
        dwp = self._buf(Cout, Kc, zero_each_run=True)
        if cols16 is not None and self._wgrad_1x1_bf16(cols16, Kc, Kc, self._cast16(dy), Cout, M, dwp, leaf=True):
            pass                                                                             # 436 GFLOP: 3.3 ms as an fp32 TN GEMM
        elif cols16 is not None:
            raise _lib.GssdError('deformable conv: no bf16 weight-gradient kernel for this shape')
        elif self.bf16_ops:
            self._wgrad_nt_bf16(cols, Kc, Kc, dy, Cout, Cout, B * H * H, dwp, Kc)
        else:
            d_c, _, _ = ops.make_conv_desc(cols, None, None, B=B, H=H, W=H, in_stride=Kc, cin_g=Kc, Cout=Cout)
            self._add(lib.gssd_conv2d_wgrad_f32, (C.byref(d_c), dy.data_ptr(), dwp.data_ptr()), keep=d_c)
        self._unpack(dwp, Kc, 0, m.weight, Cin, Cin, 3)
        cs = self._buf(Cout, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dy.data_ptr(), B * H * H, Cout, Cout, cs.data_ptr()))
        self._bias_from_colsum(cs, m.bias)
        wt = self._buf(Kc, Cout)
        self._add(lib.gssd_scaled_transpose_f32, (w_main.data_ptr(), 0, wt.data_ptr(), Cout, Kc), keep=w_main)
        dcols = self._buf(B * H * H, Kc)
        if self.bf16_ops:
            self._nt_bf16(dy, wt, dcols, B=B, H=H, in_stride=Cout, cin_g=Cout, Cout=Kc)      # 436 GFLOP: 3.4 ms in fp32
        else:
            d_dc, _, _ = ops.make_conv_desc(dy, wt, dcols, B=B, H=H, W=H, in_stride=Cout, cin_g=Cout, Cout=Kc,
                                            wgt_x6=self._x6w(wt, Kc, 1, Cout, 1, B * H * H))
            self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d_dc),), keep=(d_dc, wt))
        # sampling backward: d(x) by atomics, d(offset / mask logits) per pixel
        gx = self._grad_of(x)
        if gx is None:
            gx = self._buf(B, H, H, Cin, zero_each_run=True)
            self.gbuf[x.data_ptr()] = gx
        dom = self._buf(B, H, H, OMC, zero_each_run=True)
        self._add(lib.gssd_dcn_col2im_f32, (x.data_ptr(), om.data_ptr(), dcols.data_ptr(), gx.data_ptr(), dom.data_ptr(), B, H, H,
                                            Cin, dg, OMC))
        # offset / mask conv (its gradients on the padded channel count: the pad channel of d(om) is zero, its weight row is dropped)
        cm = m.conv_offset_mask
        d_w = r['d_om']
        if OMC != 27 * dg:
            d_w, _, _ = ops.make_conv_desc(x, None, None, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin, Cout=OMC, k=3, pad=1)
        dwo = self._buf(OMC, Kc, zero_each_run=True)
        d16 = None
        if self.bf16_ops and r.get('x16') is not None and OMC % 8 == 0:
            d16, _, _ = ops.make_conv_desc(r['x16'], None, None, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin, Cout=OMC, k=3, pad=1)
            if not lib.gssd_conv2d_wgrad_bf16_supported(C.byref(d16)):
                d16 = None
        if d16 is not None:                # dense 3x3, 512 -> 112: four 128-channel input blocks of csrc/conv_wgrad_bf16.hip
            dom16 = self._cast16(dom)
            self._add(lib.gssd_conv2d_wgrad_bf16, (C.byref(d16), dom16.data_ptr(), dwo.data_ptr()), keep=(d16, dom16), leaf=True)
        else:
            self._add(lib.gssd_conv2d_wgrad_f32, (C.byref(d_w), dom.data_ptr(), dwo.data_ptr()), keep=d_w)
        self._unpack(dwo, Kc, 0, cm.weight, Cin, Cin, 3)
        cs2 = self._buf(OMC, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dom.data_ptr(), B * H * H, OMC, OMC, cs2.data_ptr()))
        self._bias_from_colsum(cs2, cm.bias)
        if OMC == 27 * dg:
            self._dgrad(r, dom, x, cm, 1, Cin, H, H, OMC, 3, 1, 1, 1)
        else:
            # the data-gradient packer reads an OIHW weight with OMC output channels: a zero-padded copy, refreshed every run
            wpad = torch.zeros(OMC, Cin, 3, 3, device=self.dev)
            self.keep.append(wpad)

            def refresh(wpad=wpad, cm=cm, n=27 * dg):
                wpad[:n].copy_(cm.weight.detach())
            self.steps.append((refresh, None))
            self._dgrad(r, dom, x, _PaddedWeight(wpad), 1, Cin, H, H, OMC, 3, 1, 1, 1)
