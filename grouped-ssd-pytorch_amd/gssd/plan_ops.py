"""Per-op emitters of the forward launch plan: conv + BatchNorm (+ ReLU / pool), stand-alone pools, the merged loc | conf heads, spectral
norm, Self_Attn (layers/self_attn.py:46-89) and the deformable conv (layers/dcn_v2_custom.py:79-89).  Each appends its launches to the
plan and a record for the backward.  Mixin of engine._Plan."""
import ctypes as C
import os
import torch
from . import _lib, ops
from ._lib import lib
from .plan_common import DCN_X6, HEAD_OFF, MBOX, SN_STREAM, USE_CONV_X6, USE_FLASH_X6, USE_HEADS_WINO, USE_PATCH_X6, USE_WINOGRAD


class PlanOpsMixin:
    @property
    def f16_ok(self):
        """_lib.CONV_F16_OK for the forward launches of an fp32-mode, TRAIN-mode network, else 0: train-mode BatchNorm (batch statistics) bounds every
        activation map by |gamma| sqrt(n) + |beta|, far inside fp16's range, so the split-operand kernels may use their fp16 planes (three MFMAs per
        product).  Eval mode normalises with running statistics, which bound nothing, and so does a graph built with batch_norm=False (tests/test_gpu_pixellink.py's synthetic eval graph reaches
        1e12): those launches keep the bf16 planes (include/gssd_hip.h: GSSD_CONV_F16_OK)."""
        net = getattr(getattr(self, 'eng', None), 'net', None)
        bn = bool(getattr(net, 'batch_norm', False))          # (a graph without BatchNorm -- batch_norm=False, the vanilla SSD -- bounds nothing either)
        return _lib.CONV_F16_OK if (bn and not getattr(self, 'bf16', False) and getattr(self, 'training', False)) else 0

    def _conv_act(self, name, conv, x, H, Cin, groups):
        """(grouped) conv + bias + ReLU in ONE launch (ReLU in the conv epilogue): the batch_norm=False layers."""
        B = self.B
        k, s, p, dl = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
        Cout = conv.out_channels
        wp = self._packed_conv(name, conv)
        Ho = (H + 2 * p - dl * (k - 1) - 1) // s + 1
        out = self._buf(B, Ho, Ho, Cout)
        d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin // groups, Cout=Cout, groups=groups, k=k,
                                     stride=s, pad=p, dil=dl, bias=conv.bias.detach(), relu=True)
        self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d),), keep=d)
        self.rec.append(('convrelu', dict(name=name, conv=conv, x_in=x, out=out, H=H, Cin=Cin, Ho=Ho, Cout=Cout, desc=d, k=k,
                                          stride=s, pad=p, dil=dl, groups=groups)))
        return out, Ho, Cout

    def _head(self, i, s, Hs, Cs):
        """loc[i] / conf[i] (models/...group.py:375-380) as ONE merged 3x3 conv writing straight into the concatenated fp32
        loc [B,8732,4] / conf [B,8732,C] at this source's prior offset."""
        eng, net, B, dev, f32 = self.eng, self.eng.net, self.B, self.dev, torch.float32
        off = HEAD_OFF[i]
        A = MBOX[i]
        nloc, nconf = A * 4, A * self.nc
        lw, cw = net.loc[i], net.conf[i]
        cin_pad, K = ops.packed_k(Cs, 3, 3)

        def build_w(out, lw=lw, cw=cw, nloc=nloc, nconf=nconf, K=K):
            if out is None:
                out = torch.empty(nloc + nconf, K, device=dev, dtype=self.adt)
            pk = ops.pack_weight_bf16 if self.bf16 else ops.pack_weight
            pk(lw.weight, out, 0)
            pk(cw.weight, out, nloc)
            return out

        def build_b(out, lw=lw, cw=cw, nloc=nloc):
            if out is None:
                out = torch.empty(nloc + cw.bias.numel(), device=dev, dtype=f32)
            ops.copy_into(out[:nloc], lw.bias)
            ops.copy_into(out[nloc:], cw.bias)
            return out
        wp = eng._pack(f'heads.{i}.w', build_w)
        bp = eng._pack(f'heads.{i}.b', build_b)
        split = (ops.auto_split_k(B * Hs * Hs, nloc + nconf, 1, K, target_blocks=256, max_split=8)
                 if self.bf16 else ops.auto_split_k(B * Hs * Hs, nloc + nconf, 1, K))
        U = None
        if (self.f16_ok and USE_WINOGRAD and USE_HEADS_WINO and nloc % 4 == 0 and nconf % 4 == 0 and B * ((Hs + 1) // 2) ** 2 >= 8192
                and ops.winograd_eligible(3, 1, 1, 1, Cs, nloc + nconf, 1)):
            # train-mode fp32 forward, large map (38 x 38 at batch >= 23): Winograd on fp16 planes with the heads' two-destination epilogue
            # (csrc/conv_wino_x6.hip: 95 us against 173 us for the implicit GEMM and its two reduction slices); one slice
            def build_u(out, key=f'heads.{i}.w', cin=Cs):
                return ops.winograd_weight(eng._packed[key], 1, cin, out)
            U = eng._pack(f'heads.{i}.U', build_u)
            split = 1
        d, _, _ = ops.make_conv_desc(s, wp, None, B=B, H=Hs, W=Hs, in_stride=Cs, cin_g=Cs, Cout=nloc + nconf, k=3,
                                     pad=1, bias=bp, out_mode=_lib.OUT_HEADS, out_b=None, split_n=nloc,
                                     out_batch_stride=self.P * 4, outb_batch_stride=self.P * self.nc,
                                     out_off=off * 4, outb_off=off * self.nc, split_k=split, wgt_wino=U,
                                     flags=_lib.CONV_OUT_F32 | (self.f16_ok if U is not None else 0))
        self.head_descs.append(d)
        self._add(self.conv_fn, (C.byref(d),), keep=d)
        self.rec.append(('head', dict(i=i, src=s, H=Hs, C=Cs, A=A, off=off, loc=lw, conf=cw, K=K)))

    def _setup_spectral_norm(self, lists):
        """layers/spectral_norm.py:74-89 for every Self_Attn conv of ``lists`` = [(list name, ModuleList)]: ONE launch that
        (training) runs the power iteration in place and writes 1/sigma per output channel (the convs' ``alpha`` vectors)."""
        self.sn_items = []
        self.sa_state = {}
        for lst_name, lst in lists:
            for i, sa in enumerate(lst):
                Cc = sa.in_channels
                a_tpg = self._buf(Cc // 4 + Cc // 2)         # 1/sigma per output channel of the merged theta|phi|g projection
                a_o = self._buf(Cc)
                self.sn_items += [
                    (sa.snconv1x1_theta.weight_orig, sa.snconv1x1_theta.weight_u, sa.snconv1x1_theta.weight_v, a_tpg[:Cc // 8]),
                    (sa.snconv1x1_phi.weight_orig, sa.snconv1x1_phi.weight_u, sa.snconv1x1_phi.weight_v, a_tpg[Cc // 8:Cc // 4]),
                    (sa.snconv1x1_g.weight_orig, sa.snconv1x1_g.weight_u, sa.snconv1x1_g.weight_v, a_tpg[Cc // 4:]),
                    (sa.snconv1x1_attn.weight_orig, sa.snconv1x1_attn.weight_u, sa.snconv1x1_attn.weight_v, a_o),
                ]
                self.sa_state[(lst_name, i)] = (a_tpg, a_o)
        if self.sn_items:
            self.sn_dev = ops.sn_items_tensor([(w.detach(), u, v, s) for (w, u, v, s) in self.sn_items], self.dev)
            # one workgroup per matrix (48 of 256 CUs, ~180 us): on its own stream beside conv1_1 .. conv4_3 inside the graph; the
            # first Self_Attn launch joins it (every later one forks from the trunk after that point)
            prev, self._sid = getattr(self, '_sid', 0), SN_STREAM
            self._add(lib.gssd_spectral_norm_f32, (self.sn_dev.data_ptr(), len(self.sn_items), int(self.training), 1e-12))
            self._sid = prev
            self._sn_unjoined = True

    def _packed_conv(self, name, conv):
        eng = self.eng

        def build(out, conv=conv):
            if self.bf16:
                return ops.pack_weight_bf16(conv.weight, out)
            return ops.pack_weight(conv.weight, out)
        return eng._pack(name + '.w', build)

    def _conv_bn(self, name, conv, bn, x, H, Cin, groups, relu=True, pool=None, in_xf=None, defer_bn=False):
        """conv (raw output + fp64 batch sums) -> BN + ReLU (+ max-pool).  ``in_xf`` = (scale, shift, pad) of a producer
        whose BN + ReLU this conv applies on the fly; ``defer_bn`` leaves this layer's own BN + ReLU to its consumer and
        returns (raw, H, C, (scale, shift, pad))."""
        B = self.B
        self._layer = name
        k, s, p, dl = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
        Cout = conv.out_channels
        cin_g = Cin // groups
        if defer_bn and Cout // groups > 512:
            defer_bn = False        # the consumer (same group count) would read more than 512 channels per group: the conv kernels' fused
            #                         input transform keeps at most 512 scale / shift pairs (ungrouped conv6 -> conv7 at groups_vgg = 1)
        wp = self._packed_conv(name, conv)
        U = None
        if not self.bf16 and USE_WINOGRAD and ops.winograd_eligible(k, s, p, dl, cin_g, Cout // groups, groups):
            def build_u(out, key=name + '.w', groups=groups, cin_g=cin_g):
                return ops.winograd_weight(self.eng._packed[key], groups, cin_g, out)
            U = self.eng._pack(name + '.U', build_u)          # registered after '.w', so refreshed after it
        Ho = (H + 2 * p - dl * (k - 1) - 1) // s + 1
        X6 = None
        if not self.bf16 and USE_CONV_X6 and ops.x6_wanted(k, cin_g, Cout // groups, groups, B * Ho * Ho, winograd=U is not None, forward=bool(self.f16_ok)):
            def build_x6(out, key=name + '.w', groups=groups, cin_g=cin_g, taps=k * k, bn=ops.x6_tile(Cout // groups, groups, B * Ho * Ho)):
                return ops.x6_weight(self.eng._packed[key], groups, cin_g, taps, bn, out)
            X6 = self.eng._pack(name + f'.x6@{ops.x6_tile(Cout // groups, groups, B * Ho * Ho)}', build_x6)       # the tile is part of the packed layout: part of the key       # (after '.w' as well)
        st = self.eng_stat(bn)
        srep = getattr(self, 'stat_rep', {}).get(id(bn), 0) if self.training else 0
        # Pooled trunk layers of a no-backward forward (conv1_2, conv2_2, conv3_3): max-pooling commutes with the monotone BatchNorm +
        # ReLU, and the direction of monotonicity is the sign of the BatchNorm weight, known before the launch.  The conv's epilogue
        # writes max- (gamma >= 0) or min- (gamma < 0) pooled RAW outputs, a quarter of the map, with the batch sums of the full map;
        # the separate BatchNorm + ReLU + pool pass disappears and the next conv applies the deferred BatchNorm + ReLU to the pooled
        # raw map on read: bit-identical activations (include/gssd_hip.h: GSSD_CONV_POOL2), the full-resolution raw map is never
        # written or re-read (conv1_2 in bf16: 369 MB written + 369 MB re-read + 92 MB written become 92 MB written).
        cout_g = Cout // groups
        pooled = (getattr(self, 'nograd', False) and relu and pool is not None and pool[:3] == (2, 2, 0) and (pool[3] or Ho % 2 == 0) and k == 3 and s == 1
                  and p == 1 and dl == 1 and groups == 4 and
                  ((U is not None) if not self.bf16 else ((cin_g, cout_g) in ((16, 16), (32, 32)) and Ho % 2 == 0 and Ho * Ho >= 75 * 75)))
        if pooled:
            Hp = ops.pool_out_size(Ho, 2, 2, 0, pool[3])
            raw, pd = self._abuf_tail(Cout, B, Hp, Hp, Cout)
            d, _, _ = ops.make_conv_desc(x, wp, raw, B=B, H=H, W=H, in_stride=Cin, cin_g=cin_g, Cout=Cout, groups=groups, k=k,
                                         stride=s, pad=p, dil=dl, bias=conv.bias.detach(), wgt_wino=U,
                                         stats=st if self.training else None,
                                         in_scale=in_xf[0] if in_xf else None, in_shift=in_xf[1] if in_xf else None,
                                         in_pad=in_xf[2] if in_xf else None, flags=_lib.CONV_POOL2 | self.f16_ok, pool_sign=bn.weight.detach(), stats_rep=srep)
            self._add(self.conv_fn, (C.byref(d),), keep=d)
            sc, sh = self._buf(Cout), self._buf(Cout)
            self._add(lib.gssd_bn_finalize_bf16 if self.bf16 else lib.gssd_bn_finalize_f32,
                      (st.data_ptr(), float(B * Ho * Ho), bn.weight.data_ptr(), bn.bias.data_ptr(),
                       bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps),
                       int(self.training), Cout, sc.data_ptr(), sh.data_ptr(), pd.data_ptr(), srep))
            self.rec.append(('convbn', dict(name=name, conv=conv, bn=bn, x_in=x, in_xf=in_xf, H=H, Cin=Cin, groups=groups, raw=raw, Ho=Ho,
                                            Cout=Cout, desc=d, stats=st, stats_rep=srep, pool=pool, relu=relu, k=k, stride=s, pad=p, dil=dl, out=raw,
                                            Hp=Hp, xf=(sc, sh, pd), pooled=True)))
            self._layer = None
            return raw, Hp, Cout, (sc, sh, pd)
        raw, pd_tail = self._abuf_tail(Cout, B, Ho, Ho, Cout) if defer_bn else (self._abuf(B, Ho, Ho, Cout), None)
        d, _, _ = ops.make_conv_desc(x, wp, raw, B=B, H=H, W=H, in_stride=Cin, cin_g=cin_g, Cout=Cout, groups=groups, k=k,
                                     stride=s, pad=p, dil=dl, bias=conv.bias.detach(), wgt_wino=U, wgt_x6=X6,
                                     stats=st if self.training else None,
                                     in_scale=in_xf[0] if in_xf else None, in_shift=in_xf[1] if in_xf else None,
                                     in_pad=in_xf[2] if in_xf else None, stats_rep=srep,
                                     flags=self.f16_ok)      # fp32 mode, train-mode network: forward launches on activation maps may use the x6 kernels' fp16 planes
        self._add(self.conv_fn, (C.byref(d),), keep=d)
        rec = dict(name=name, conv=conv, bn=bn, x_in=x, in_xf=in_xf, H=H, Cin=Cin, groups=groups, raw=raw, Ho=Ho, Cout=Cout,
                   desc=d, stats=st, stats_rep=srep, pool=pool, relu=relu, k=k, stride=s, pad=p, dil=dl)
        self.rec.append(('convbn', rec))
        if defer_bn:
            assert pool is None and relu
            sc, sh, pd = self._buf(Cout), self._buf(Cout), pd_tail
            self._add(lib.gssd_bn_finalize_bf16 if self.bf16 else lib.gssd_bn_finalize_f32,
                      (st.data_ptr(), float(B * Ho * Ho), bn.weight.data_ptr(), bn.bias.data_ptr(),
                       bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps),
                       int(self.training), Cout, sc.data_ptr(), sh.data_ptr(), pd.data_ptr(), srep))
            rec.update(out=raw, Hp=Ho, xf=(sc, sh, pd))
            self._layer = None
            return raw, Ho, Cout, (sc, sh, pd)
        if pool:
            pk, ps, pp, ceil = pool
            Hp = ops.pool_out_size(Ho, pk, ps, pp, ceil)
        else:
            pk, ps, pp, Hp = 0, 1, 0, Ho
        act = self._abuf(B, Hp, Hp, Cout)
        self._add(lib.gssd_bn_relu_pool_bf16 if self.bf16 else lib.gssd_bn_relu_pool_f32,
                  (raw.data_ptr(), act.data_ptr(), B, Ho, Ho, Cout, Hp, Hp, pk, ps, pp, st.data_ptr(), float(B * Ho * Ho),
                   bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                   float(bn.momentum), float(bn.eps), int(self.training), int(relu), srep),
                  tag=('bn_relu_pool_bf16' if self.bf16 else 'bn_relu_pool', 0.0,
                       (2.0 if self.bf16 else 4.0) * B * Cout * (Ho * Ho + Hp * Hp)))
        rec.update(out=act, Hp=Hp, xf=None)
        self._layer = None
        return act, Hp, Cout, None

    def eng_stat(self, bn):
        return self.stat_of[id(bn)]

    def _pool_only(self, x, H, Cc, k, s, p, ceil=False):
        B = self.B
        Hp = ops.pool_out_size(H, k, s, p, ceil)
        out = self._abuf(B, Hp, Hp, Cc)
        self._add(lib.gssd_bn_relu_pool_bf16 if self.bf16 else lib.gssd_bn_relu_pool_f32,
                  (x.data_ptr(), out.data_ptr(), B, H, H, Cc, Hp, Hp, k, s, p, 0, 1.0, 0, 0, 0, 0, 0.1, 1e-5, 0, 0, 0),
                  tag=('bn_relu_pool_bf16' if self.bf16 else 'bn_relu_pool', 0.0, (2.0 if self.bf16 else 4.0) * B * Cc * (H * H + Hp * Hp)))
        self.rec.append(('pool', dict(x_in=x, out=out, H=H, C=Cc, k=k, s=s, p=p, Hp=Hp)))
        return out, Hp

    def _self_attn(self, lst_name, idx, x, H, Cc, need_out2, want_map=False):
        """layers/self_attn.py:46-89 as three launches: ONE pass over x for the theta | phi | g projections (K9; g written
        transposed), the flash-style core theta^T phi -> softmax -> . g (K10, csrc/flash_attn.hip: the [N, N] map never exists),
        and the o conv with the sigma-gated residual epilogue.  ``want_map`` (visualize=True, op-level tests) additionally
        materialises the attention map with two extra launches; the output path does not read it."""
        eng, B = self.eng, self.B
        sa = getattr(eng.net, lst_name)[idx]
        a_tpg, a_o = self.sa_state[(lst_name, idx)]
        if self.__dict__.pop('_sn_unjoined', False):
            self._pending_wait = SN_STREAM          # the next launch added (this block's projection) waits for the 1/sigma vectors
        N = H * H
        Np = ops.round_up(N, 4)
        C8, C2, C4 = Cc // 8, Cc // 2, Cc // 4
        dev, f32 = self.dev, torch.float32
        name = f'{lst_name}.{idx}'

        def build_w(out):
            if out is None:
                out = torch.empty(C4 + C2, Cc, device=dev, dtype=self.adt)
            ops.copy_into(out[:C8], sa.snconv1x1_theta.weight_orig)                     # (rounds to bf16 in bf16 mode)
            ops.copy_into(out[C8:C4], sa.snconv1x1_phi.weight_orig)
            ops.copy_into(out[C4:], sa.snconv1x1_g.weight_orig)
            return out

        def build_wo(out):
            if out is None:
                out = torch.empty(Cc, C2, device=dev, dtype=self.adt)
            ops.copy_into(out, sa.snconv1x1_attn.weight_orig)
            return out

        def build_b(out):
            if out is None:
                out = torch.empty(C4 + C2, device=dev, dtype=f32)
            ops.copy_into(out[:C8], sa.snconv1x1_theta.bias)
            ops.copy_into(out[C8:C4], sa.snconv1x1_phi.bias)
            ops.copy_into(out[C4:], sa.snconv1x1_g.bias)
            return out
        w_tpg = eng._pack(name + '.tpg.w', build_w)
        b_tpg = eng._pack(name + '.tpg.b', build_b)
        # the o conv's weight is already K-major rows; bf16 mode keeps a rounded copy
        w_o = eng._pack(name + '.o.w', build_wo) if self.bf16 else sa.snconv1x1_attn.weight_orig.detach().view(Cc, C2)
        tp = self._buf(B, N, C4)               # theta | phi stay fp32 in both modes: the logits and the softmax are fp32
        if self.bf16:                          # g^T bf16, rows in the key order of the bf16-value core (csrc/flash_attn.hip)
            Np = ops.round_up(N, 32)
            gT = self._abuf(B, C2, Np)
        else:
            gT = self._buf(B, C2, Np)
        ag = self._abuf(B, N, C2)
        out = self._abuf(B, H, H, Cc)
        out2 = self._abuf(B, H, H, Cc) if need_out2 else None
        mk = ops.make_conv_desc
        # fp32, N % 4 == 0 (38 x 38): all images as ONE M range -- 361 full row tiles instead of 12 per image with a ragged last one,
        # and the plain-GEMM dispatch (slot stream) instead of the per-image one
        flat = not self.bf16 and N % 4 == 0 and Np == N
        x6_tpg = None
        if (not self.bf16 and USE_CONV_X6 and ops.x6_wanted(1, Cc, C4 + C2, 1, B * N) and C4 % ops.x6_tile(C4 + C2, 1, B * N) == 0):
            def build_x6p(out, key=name + '.tpg.w', bn=ops.x6_tile(C4 + C2, 1, B * N)):
                return ops.x6_weight(eng._packed[key], 1, Cc, 1, bn, out)
            x6_tpg = eng._pack(name + f'.tpg.x6@{ops.x6_tile(C4 + C2, 1, B * N)}', build_x6p)
            gT.zero_()                         # csrc/conv_x6.hip never writes the row tails [N, Np) of g^T (conv_igemm zero-fills them)
        d1, _, _ = mk(x, w_tpg, tp, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C4 + C2, bias=b_tpg, alpha=a_tpg, wgt_x6=x6_tpg,
                      out_mode=_lib.OUT_SPLIT_T, out_b=gT, split_n=C4, out_stride=C4, out_b_stride=Np, m_per_image=not flat,
                      in_batch_stride=N * Cc, out_batch_stride=N * C4, outb_batch_stride=C2 * Np,
                      flags=_lib.CONV_OUT_F32 | (_lib.CONV_OUTB_BF16_PERM32 if self.bf16 else self.f16_ok))
        x6_o = None
        if not self.bf16 and USE_CONV_X6 and ops.x6_wanted(1, C2, Cc, 1, B * N):
            def build_x6o(out, bn=ops.x6_tile(Cc, 1, B * N)):
                return ops.x6_weight(sa.snconv1x1_attn.weight_orig.detach().view(Cc, C2), 1, C2, 1, bn, out)
            x6_o = eng._pack(name + f'.o.x6@{ops.x6_tile(Cc, 1, B * N)}', build_x6o)
        d5, _, _ = mk(ag, w_o, out, B=B, H=H, W=H, in_stride=C2, cin_g=C2, Cout=Cc, bias=sa.snconv1x1_attn.bias.detach(),
                      alpha=a_o, gate=sa.sigma.detach(), resid=x, out2=out2, wgt_x6=x6_o, flags=self.f16_ok)
        fn = self.conv_fn
        if C4 % 64 == 0:
            self._add(fn, (C.byref(d1),), keep=(d1, w_tpg, b_tpg))
        else:
            if self.bf16:
                raise _lib.GssdError('bf16 mode: Self_Attn needs >= 64 theta|phi channels (in_channels >= 256)')
            # narrow blocks (fewer than 64 theta|phi channels: not on the detector's path, op-level tests only): the merged
            # launch's column split needs whole 64-channel tiles, so theta|phi and g go out as two launches over the same weights
            d1a, _, _ = mk(x, w_tpg, tp, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C4, bias=b_tpg, alpha=a_tpg)
            d1b, _, _ = mk(x, w_tpg[C4:], gT, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C2, bias=b_tpg[C4:], alpha=a_tpg[C4:],
                           out_mode=_lib.OUT_TRANSPOSED, out_stride=Np, m_per_image=True, in_batch_stride=N * Cc,
                           out_batch_stride=C2 * Np)
            self._add(fn, (C.byref(d1a),), keep=(d1a, w_tpg, b_tpg))
            self._add(fn, (C.byref(d1b),), keep=d1b)
        # training keeps the rows' log-sum-exp: the backward rebuilds the probabilities from it in a GEMM epilogue
        lse = self._buf(B, N) if self.training else None
        # max_pool_factor > 1 (layers/self_attn.py:57-59, 67, 76): keys / values average-pooled to a P x P grid before the core
        P = max(H // int(sa.max_pool_factor), 1)
        pooled = P != H
        Nk, Nkp, kp, gTp = N, Np, None, None
        if pooled:
            if self.bf16:
                raise _lib.GssdError('bf16 storage mode is built for max_pool_factor = 1 (BASELINE.json configs[4])')
            Nk, Nkp = P * P, ops.round_up(P * P, 4)
            kp, gTp = self._buf(B, Nk, C8), self._buf(B, C2, Nkp)
            self._add(lib.gssd_sa_pool_kv_f32, (tp.data_ptr(), gT.data_ptr(), kp.data_ptr(), gTp.data_ptr(), B, H, P, C8, C2, Np, Nkp))
            self._add(lib.gssd_self_attn_core_kv_f32, (tp.data_ptr(), kp.data_ptr(), gTp.data_ptr(), ag.data_ptr(), B, N, Nk, Nkp, C8, C2,
                                                       C8, 0, lse.data_ptr() if lse is not None else 0),
                      tag=(f'flash_attn<{C8},{C2}>', 2.0 * B * N * Nk * (C8 + C2), 4.0 * B * (N * C8 + Nk * C8 + C2 * Nkp + N * C2)))
        elif self.bf16:
            self._add(lib.gssd_self_attn_core_bf16v, (tp.data_ptr(), gT.data_ptr(), ag.data_ptr(), B, N, Np, C8, C2,
                                                      lse.data_ptr() if lse is not None else 0),
                      tag=(f'flash_attn_bf16v<{C8},{C2}>', 2.0 * B * N * N * (C8 + C2), B * (4.0 * N * C4 + 2.0 * C2 * Np + 2.0 * N * C2)))
        elif USE_FLASH_X6 and N >= 1024 and C8 == 64 and lib.gssd_self_attn_core_x6_supported(C8, C2):
            # both products of the core on the bf16 matrix cores over three-plane operands (csrc/flash_attn_x6.hip): fp32-equivalent results at
            # 6 / 16 of the fp32 instruction's matrix-pipe time; the planes of theta | phi and g^T live in a scratch buffer of the plan
            ws = self._buf(int(lib.gssd_self_attn_core_x6_ws_bytes(B, N, C8, C2)) // 4)
            self._add(lib.gssd_self_attn_core_x6_f32, (tp.data_ptr(), gT.data_ptr(), ag.data_ptr(), B, N, Np, C8, C2, ws.data_ptr(),
                                                       lse.data_ptr() if lse is not None else 0),
                      tag=(f'flash_attn_x6<{C8},{C2}>', 2.0 * B * N * N * (C8 + C2), 4.0 * B * (N * C4 + C2 * Np + N * C2)))
        else:
            self._add(lib.gssd_self_attn_core_kv_f32, (tp.data_ptr(), tp[0, 0, C8:].data_ptr(), gT.data_ptr(), ag.data_ptr(), B, N, N, Np,
                                                       C8, C2, C4, 0, lse.data_ptr() if lse is not None else 0),
                      tag=(f'flash_attn<{C8},{C2}>', 2.0 * B * N * N * (C8 + C2), 4.0 * B * (N * C4 + C2 * Np + N * C2)))
        S = None
        if want_map:
            # attn[b,i,j] = softmax_j(sum_c theta[b,i,c] * phi[b,j,c])   (no 1/sqrt(d) scaling, self_attn.py:71-72)
            S = self._buf(B, N, Nkp)
            keys, krow = (kp, C8) if pooled else (tp[0, 0, C8:], C4)
            d3, _, _ = mk(tp, keys, S, B=B, H=H, W=H, in_stride=C4, cin_g=C8, Cout=Nk, out_stride=Nkp, m_per_image=True,
                          in_batch_stride=N * C4, wgt_batch_stride=Nk * krow, out_batch_stride=N * Nkp, wgt_row_stride=krow)
            self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d3),), keep=d3)        # fp32 operands in both modes
            self._add(lib.gssd_softmax_rows_f32, (S.data_ptr(), B * N, Nk, Nkp))
        self._add(fn, (C.byref(d5),), keep=d5)
        self.attn_maps = getattr(self, 'attn_maps', {})
        self.attn_maps[(lst_name, idx)] = (S, Nk, Nkp)
        self.rec.append(('sa', dict(mod=sa, name=name, x_in=x, out=out, out2=out2, H=H, C=Cc, tp=tp, gT=gT, ag=ag, N=N, Np=Np,
                                    inv_sigma=(a_tpg, a_o), P=P, Nk=Nk, Nkp=Nkp, kp=kp, gTp=gTp, lse=lse)))
        return out, out2

    def _dcn(self, li, x, H, Cin):
        """layers/dcn_v2_custom.py:79-89: offset/mask conv, then ONE fused kernel for the modulated bilinear sampling and the
        9*Cin-deep contraction (csrc/dcn_fused.hip) -- no column buffer."""
        eng, B = self.eng, self.B
        m = eng.net.dcn_list[li]
        dg, Cout = m.deformable_groups, m.out_channels
        w_om = self._packed_conv(f'dcn_list.{li}.om', m.conv_offset_mask)

        def build_w(out, m=m, Cin=Cin, dg=dg):
            elems = lib.gssd_dcn_packed_weight_elems_bf16 if self.bf16 else lib.gssd_dcn_packed_weight_elems_x6 if DCN_X6 else lib.gssd_dcn_packed_weight_elems
            pack = lib.gssd_dcn_pack_weight_bf16 if self.bf16 else lib.gssd_dcn_pack_weight_x6 if DCN_X6 else lib.gssd_dcn_pack_weight_f32
            if out is None:
                n = int(elems(m.out_channels, Cin))
                if n <= 0:
                    raise _lib.GssdError(f'deformable conv: unsupported shape Cin {Cin}, Cout {m.out_channels}')
                out = torch.empty(n, device=self.dev, dtype=torch.bfloat16 if (DCN_X6 and not self.bf16) else self.adt)
            _lib.check(pack(m.weight.detach().contiguous().data_ptr(), out.data_ptr(), m.out_channels, Cin, dg,
                            torch.cuda.current_stream().cuda_stream))
            return out
        w_main = eng._pack(f'dcn_list.{li}.wt' + ('.x6' if (DCN_X6 and not self.bf16) else ''), build_w)
        # offsets / mask logits stay fp32 in both modes; rows padded to a multiple of 4 channels (27 * dg is one only for dg = 4, 8, ..):
        # the weight-gradient and data-gradient kernels of the offset conv want 16-byte aligned channel vectors
        # (bf16 mode: a multiple of 8 -- the training step's bf16 data / weight gradients of the offset conv read 16-byte bf16 rows)
        OMC = ops.round_up(27 * dg, 8 if self.bf16 else 4)
        om = self._buf(B, H, H, OMC)
        if OMC != 27 * dg:
            om.zero_()
        out = self._abuf(B, H, H, Cout)
        u_om = None
        if not self.bf16 and USE_WINOGRAD and ops.winograd_eligible(3, 1, 1, 1, Cin, 27 * dg, 1):
            def build_u(out, key=f'dcn_list.{li}.om.w', cin=Cin):
                return ops.winograd_weight(eng._packed[key], 1, cin, out)
            u_om = eng._pack(f'dcn_list.{li}.om.U', build_u)
        p_om = None
        if self.f16_ok and USE_PATCH_X6 and lib.gssd_conv_patch_x6_weight_elems(27 * dg, Cin) > 0:
            # train-mode fp32 forward: the patch-staged direct conv on fp16 planes (csrc/conv_patch_x6.hip: 1024 -> 108 channels, 483 -> 398 us)
            def build_p(out, key=f'dcn_list.{li}.om.w', cin=Cin):
                return ops.patch_x6_weight(eng._packed[key], cin, out)
            p_om = eng._pack(f'dcn_list.{li}.om.px6', build_p)
        d1, _, _ = ops.make_conv_desc(x, w_om, om, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin, Cout=27 * dg, k=3, pad=1, out_stride=OMC,
                                      bias=m.conv_offset_mask.bias.detach(), wgt_wino=u_om, wgt_patch=p_om,
                                      flags=_lib.CONV_OUT_F32 | self.f16_ok)
        self._add(self.conv_fn, (C.byref(d1),), keep=d1)
        M = B * H * H
        esz = 2.0 if self.bf16 else 4.0
        self._add(lib.gssd_dcn_forward_bf16 if self.bf16 else lib.gssd_dcn_forward_x6_ex if DCN_X6 else lib.gssd_dcn_forward_f32,
                  (x.data_ptr(), om.data_ptr(), w_main.data_ptr(), m.bias.data_ptr(), out.data_ptr(), B, H, H, Cin, dg, OMC, Cout) +
                  ((self.f16_ok,) if DCN_X6 and not self.bf16 else ()),
                  keep=w_main, tag=('dcn_bf16<128x256>' if self.bf16 else 'dcn_x6<128x256>' if DCN_X6 else 'dcn_fused<128x256>', 2.0 * M * Cout * 9 * Cin,
                                    esz * (M * (Cin + Cout) + Cout * 9 * Cin) + 4.0 * M * 27 * dg))
        if not self.bf16 and not DCN_X6:
            # the stream-K form keeps a flag / slab region per output buffer: released with the plan (engine._Plan.__del__)
            self.__dict__.setdefault('_sk_outs', []).append(out.data_ptr())
        self.offsets = getattr(self, 'offsets', [])
        self.offsets.append((om, H, dg))
        self.rec.append(('dcn', dict(mod=m, x_in=x, out=out, H=H, Cin=Cin, Cout=Cout, om=om, d_om=d1, dg=dg, li=li, omc=OMC)))
        return out, Cout
