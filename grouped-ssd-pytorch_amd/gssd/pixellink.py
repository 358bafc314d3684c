"""Launch plan of PixelLink++ (ssd_liverdet/pixel_link/model.py:189-413, config version "4s") on the GSSD++ kernels.

Nothing here is a new convolution: the grouped VGG trunk (conv + bias, ReLU applied by the consumer's fused input transform or by
the pool pass), the Self_Attn blocks (merged projection + flash core + gated output conv), the slice_and_cat + fused deformable
conv at the 75 x 75 stage, the 1x1 fuse conv + BatchNorm and the 1x1 score heads (pixel 2 + link 16 channels as ONE 18-channel
conv per stage) are launches of the kernels the detector uses.  New are the upsample-add cascade and the final 1x1 convs
(csrc/pixellink.hip).  fp32.  Every launch leaves a record (``rec``); a grad-enabled forward is followed by the HIP backward plan
of gssd/backward.py::PixelLinkBackwardPlan (SURVEY.md 8f row 4 as a TRAINING row).
"""
import ctypes as C

import torch

from . import _lib, ops
from .engine import GssdEngine, _Plan, _RecList, USE_WINOGRAD, USE_CONV_X6

lib = _lib.lib

_TRUNK = (('conv1_1', 'conv1_2'), ('conv2_1', 'conv2_2'), ('conv3_1', 'conv3_2', 'conv3_3'), ('conv4_1', 'conv4_2', 'conv4_3'),
          ('conv5_1', 'conv5_2', 'conv5_3'))


class PixelLinkEngine(GssdEngine):
    def _build(self, B, training, dev, want_maps=False, nograd=False):
        return _PlanPixelLink(self, B, training, dev)

    def forward(self, x, training, events=None):
        out_1, out_2, _ = self.forward_plan(x, training, events)
        return out_1, out_2


class _PlanPixelLink(_Plan):
    # PixelLink++'s VGG trunk is conv + ReLU without BatchNorm (model.py:40-77): nothing bounds its activation maps, so its launches never carry
    # GSSD_CONV_F16_OK (the split-operand kernels keep their bf16 planes)
    f16_ok = 0

    def backward_plan(self):
        if self._bwd is None:
            from .backward import PixelLinkBackwardPlan
            self._bwd = PixelLinkBackwardPlan(self)
        return self._bwd

    def __init__(self, eng, B, training, dev):   # noqa: _Plan.__init__ builds the detector graph; not called on purpose
        self.eng, self.B, self.training, self.dev = eng, B, training, dev
        self.want_maps = False
        self.bf16, self.adt, self.conv_fn, self.cpad = False, torch.float32, lib.gssd_conv2d_nhwc_f32, 4
        net = eng.net
        self.steps, self.bufs, self.head_descs = [], [], []
        self.rec = _RecList(self)         # forward graph records, walked in reverse by gssd/backward.py::PixelLinkBackwardPlan
        self.P, self.nc = 0, 0            # (no prior boxes: the roots of the backward are d(out_1), d(out_2))
        g = net.vgg_groups
        # ---- batch-stat arena (the fuse BatchNorms) ----------------------------------------------------
        uniq, seen = [], set()
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d) and id(m) not in seen:
                seen.add(id(m))
                uniq.append(m)
        self.stats = torch.zeros(max(sum(2 * m.num_features for m in uniq), 2), device=dev, dtype=torch.float64)
        self.stat_of, off = {}, 0
        for m in uniq:
            self.stat_of[id(m)] = self.stats[off:off + 2 * m.num_features]
            off += 2 * m.num_features
        self.nbt = [m.num_batches_tracked for m in uniq]
        self._setup_spectral_norm([(n, getattr(net, n)) for n in ('self_attn_base_list', 'self_attn_list')
                                   if getattr(net, n, None) is not None])
        self._relu_xf = {}
        # ---- trunk ----------------------------------------------------------------------------------------
        x16 = self._buf(B, 300, 300, 4 * g)
        self._pack_step = len(self.steps)
        self._add(lib.gssd_pack_input_nhwc, [0, x16.data_ptr(), B, 12, 300, 300, g, 4])
        cur, H, Cc = x16, 300, 4 * g
        self.sab_i = self.sa_i = 0
        self.l = {}
        for si, names in enumerate(_TRUNK):
            xf = None
            for name in names:
                cur, H, Cc = self._conv_raw(name, getattr(net, name), cur, H, Cc, g, xf)
                xf = self._identity_relu(Cc)                  # the next conv of the stage applies this layer's ReLU on read
            if si < 2:                                         # pool1 / pool2 directly behind the stage: ReLU + pool, one pass
                cur, H = self._relu_pool(cur, H, Cc, (2, 2, 0, True))
                continue
            cur, _ = self._relu_pool(cur, H, Cc, None)         # stage output: explicit ReLU (Self_Attn / heads read it)
            cur, Cc = self._stage(cur, H, Cc, si)              # [SA-base] -> [cat + DCN] ; side branch -> l{k}
            cur, H = self._relu_pool(cur, H, Cc, (2, 2, 0, True) if si < 4 else (3, 1, 1, True), relu=False)   # pool3 / 4 / 5
        cur, H, Cc = self._conv_raw('conv6', net.conv6, cur, H, Cc, g, None)
        cur, H, Cc = self._conv_raw('conv7', net.conv7, cur, H, Cc, g, self._identity_relu(Cc))
        cur, _ = self._relu_pool(cur, H, Cc, None)
        self._stage(cur, H, Cc, 5)
        # ---- cascade (model.py:341-411) on 18-channel maps --------------------------------------------------
        (l2, H2), (l3, H3), (l4, H4), (l5, H5) = self.l[2], self.l[3], self.l[4], self.l[5]
        assert H4 == H5
        t1 = self._interp(l5, H5, H4, addend=l4)[1]            # l5 + l4 (same size: the interpolation is the identity)
        t2 = self._interp(t1, H4, H3, addend=l3)[1]            # up(l5 + l4) + l3
        f2, logit = self._interp(t2, H3, H2, addend=l2)        # up(...) ; + l2
        if net.cascade_fuse:
            f0 = self._interp(l5, H5, H2)[0]
            f1 = self._interp(t1, H4, H2)[0]
            feats = [f0, f1, f2, logit]
        else:
            feats = [logit]
        self.H_out = H2
        self._final_step = len(self.steps)
        w1, b1 = net.final_1.weight.detach().view(2, -1), net.final_1.bias.detach()
        w2, b2 = net.final_2.weight.detach().view(16, -1), net.final_2.bias.detach()
        ptrs = [f.data_ptr() for f in feats] + [0] * (4 - len(feats))
        self._add(lib.gssd_pixellink_final_f32, ptrs + [len(feats), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 0, 0, B,
                                                        H2 * H2], keep=(w1, b1, w2, b2, feats))
        self.rec.append(('plfinal', dict(feats=feats, H=H2, final_1=net.final_1, final_2=net.final_2)))

    # ------------------------------------------------------------------------------------------------
    def _identity_relu(self, Cc):
        """(scale = 1, shift = 0, pad = 0): the fused input transform max(x * scale + shift, 0) of the conv kernels, used as a bare
        ReLU -- the producing conv stores conv + bias, its only consumer rectifies on read (exact: x * 1 + 0 == x)."""
        if Cc not in self._relu_xf:
            one = torch.ones(Cc, device=self.dev)
            zero = torch.zeros(Cc, device=self.dev)
            self._relu_xf[Cc] = (one, zero, torch.zeros(Cc, device=self.dev))
        return self._relu_xf[Cc]

    def _conv_raw(self, name, conv, x, H, Cin, groups, in_xf, relu_after=True):
        """conv + bias; ``relu_after``: the ReLU that follows in the graph is applied by the consumer (fused input transform) or by the
        next _relu_pool pass -- the backward masks this layer's output gradient with [raw > 0]."""
        B = self.B
        k, s, p, dl = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
        Cout = conv.out_channels
        cin_g = Cin // groups
        wp = self._packed_conv(name, conv)
        U = None
        if USE_WINOGRAD and ops.winograd_eligible(k, s, p, dl, cin_g, Cout // groups, groups):
            def build_u(out, key=name + '.w', groups=groups, cin_g=cin_g):
                return ops.winograd_weight(self.eng._packed[key], groups, cin_g, out)
            U = self.eng._pack(name + '.U', build_u)
        Ho = (H + 2 * p - dl * (k - 1) - 1) // s + 1
        raw = self._buf(B, Ho, Ho, Cout)
        X6 = None
        if not getattr(self, "bf16", False) and USE_CONV_X6 and ops.x6_wanted(k, cin_g, Cout // groups, groups, B * Ho * Ho, winograd=U is not None):      # conv6 / conv7
            def build_x6(out, key=name + '.w', groups=groups, cin_g=cin_g, taps=k * k, bn=ops.x6_tile(Cout // groups, groups, B * Ho * Ho)):
                return ops.x6_weight(self.eng._packed[key], groups, cin_g, taps, bn, out)
            X6 = self.eng._pack(name + f'.x6@{ops.x6_tile(Cout // groups, groups, B * Ho * Ho)}', build_x6)       # the tile is part of the packed layout: part of the key
        d, _, _ = ops.make_conv_desc(x, wp, raw, B=B, H=H, W=H, in_stride=Cin, cin_g=cin_g, Cout=Cout, groups=groups, k=k, stride=s,
                                     pad=p, dil=dl, bias=conv.bias.detach(), wgt_wino=U, wgt_x6=X6,
                                     in_scale=in_xf[0] if in_xf else None, in_shift=in_xf[1] if in_xf else None,
                                     in_pad=in_xf[2] if in_xf else None)
        self._add(self.conv_fn, (C.byref(d),), keep=(d, in_xf))
        self.rec.append(('convrelu', dict(name=name, conv=conv, x_in=x, out=raw, H=H, Cin=Cin, Ho=Ho, Cout=Cout, desc=d, k=k, stride=s,
                                          pad=p, dil=dl, groups=groups, relu=bool(relu_after))))
        return raw, Ho, Cout

    def _relu_pool(self, x, H, Cc, pool, relu=True):
        """ReLU and / or max-pool (ceil_mode) as the identity-affine BatchNorm pass."""
        B = self.B
        if pool:
            pk, ps, pp, ceil = pool
            Hp = ops.pool_out_size(H, pk, ps, pp, ceil)
        else:
            pk, ps, pp, Hp = 0, 1, 0, H
        out = self._buf(B, Hp, Hp, Cc)
        self._add(lib.gssd_bn_relu_pool_f32, (x.data_ptr(), out.data_ptr(), B, H, H, Cc, Hp, Hp, pk, ps, pp, 0, 1.0, 0, 0, 0, 0, 0.1, 1e-5,
                                              0, int(relu), 0))
        self.rec.append(('relupool', dict(x_in=x, out=out, H=H, C=Cc, k=pk, s=ps, p=pp, Hp=Hp, relu=bool(relu))))
        return out, Hp

    def _stage(self, x, H, Cc, si):
        """One output stage (model.py:238-262 and its three repeats): [SA-base] -> (75 x 75 stage only) [slice_and_cat] + DCN* ->
        x continues down the trunk;  side: [SA] -> fuse conv (+ BatchNorm, no ReLU) -> the 2 + 16 channel 1x1 heads."""
        net, B = self.eng.net, self.B
        k = si                             # stage index k = 2 (conv3_3), 3, 4, 5 (fc7)
        attn_g = None
        if net.use_self_attention_base:
            x, attn_g = self._self_attn('self_attn_base_list', self.sab_i, x, H, Cc, need_out2=bool(net.dcn_cat_sab and k == 2))
            self.sab_i += 1
        if k == 2 and net.use_dcn:
            xin, Cin = x, Cc
            if net.dcn_cat_sab:
                xc = self._buf(B, H, H, 2 * Cc)
                self._add(lib.gssd_slice_and_cat_f32, (x.data_ptr(), attn_g.data_ptr(), xc.data_ptr(), B * H * H, Cc, Cc, net.vgg_groups))
                self.rec.append(('slice_cat', dict(a=x, b=attn_g, out=xc, H=H, Ca=Cc, Cb=Cc, groups=net.vgg_groups,
                                                   detach_b=bool(net.detach_sab))))
                xin, Cin = xc, 2 * Cc
            for li in range(net.num_dcn_layers):
                xin, Cin = self._dcn(li, xin, H, Cin)
            x, Cc = xin, Cin
        s = x
        if net.use_self_attention:
            s, _ = self._self_attn('self_attn_list', self.sa_i, s, H, Cc, need_out2=False)
            self.sa_i += 1
        if net.use_fuseconv:
            conv = getattr(net, f'fuse{k}')
            if net.batch_norm:
                s, _, _, _ = self._conv_bn(f'fuse{k}', conv, getattr(net, f'bn_fuse{k}'), s, H, Cc, 1, relu=False)
            else:
                s, _, _ = self._conv_raw(f'fuse{k}', conv, s, H, Cc, 1, None, relu_after=False)
        o1, o2 = getattr(net, f'out{k}_1'), getattr(net, f'out{k}_2')

        def build_w(out, o1=o1, o2=o2, Cc=Cc):
            if out is None:
                out = torch.empty(18, Cc, device=self.dev)
            ops.copy_into(out[:2], o1.weight)
            ops.copy_into(out[2:], o2.weight)
            return out

        def build_b(out, o1=o1, o2=o2):
            if out is None:
                out = torch.empty(18, device=self.dev)
            ops.copy_into(out[:2], o1.bias)
            ops.copy_into(out[2:], o2.bias)
            return out
        w = self.eng._pack(f'out{k}.w', build_w)
        bb = self.eng._pack(f'out{k}.b', build_b)
        l = self._buf(B, H, H, 18)
        d, _, _ = ops.make_conv_desc(s, w, l, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=18, bias=bb)
        self._add(self.conv_fn, (C.byref(d),), keep=(d, w, bb))
        self.rec.append(('plhead', dict(src=s, out=l, H=H, C=Cc, k=k, o1=o1, o2=o2)))
        self.l[k] = (l, H)
        return x, Cc

    def _interp(self, src, Hs, Hd, addend=None):
        B = self.B
        out = self._buf(B, Hd, Hd, 18)
        out2 = self._buf(B, Hd, Hd, 18) if addend is not None else None
        self._add(lib.gssd_interp_add_f32, (src.data_ptr(), addend.data_ptr() if addend is not None else 0, out.data_ptr(),
                                            out2.data_ptr() if out2 is not None else 0, B, Hs, Hs, Hd, Hd, 18))
        self.rec.append(('interp', dict(src=src, addend=addend, out=out, out2=out2, Hs=Hs, Hd=Hd)))
        return out, out2

    # ------------------------------------------------------------------------------------------------
    def run(self, x, events=None):
        self.generation += 1
        x = x.contiguous().float()
        B, dev, Ho = self.B, self.dev, self.H_out
        out_1 = torch.empty(B, 2, Ho, Ho, device=dev)
        out_2 = torch.empty(B, 16, Ho, Ho, device=dev)
        self.steps[self._pack_step].args[0] = x.data_ptr()
        fin = self.steps[self._final_step].args
        fin[9], fin[10] = out_1.data_ptr(), out_2.data_ptr()
        if self.training:
            self.stats.zero_()
        stream = torch.cuda.current_stream().cuda_stream
        for st in self.steps:
            if events is not None and st.tag is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._launch(st, stream)
                e1.record()
                events.append((st.tag, e0, e1))
            else:
                self._launch(st, stream)
        if self.training and self.nbt:
            torch._foreach_add_(self.nbt, 1)
        self._x_keepalive = x
        return out_1, out_2
