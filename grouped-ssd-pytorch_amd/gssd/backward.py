"""HIP backward of the GSSD graph (grouped VGG trunk, extras, fuse convs, L2Norm, heads): SURVEY.md 8f row 1.

Built from the forward plan's records (``engine._Plan.rec``) and walked in reverse.  Per conv + BN + ReLU (+ pool) layer:

    d(out)  --bn_bwd_reduce-->  dz (pool's first-max routing, ReLU mask)  +  per-channel (sum dz, sum dz*raw)
            --bn_bwd_finalize-> dgamma, dbeta, coefficients of d(raw) = A*dz + B*raw + C
            --bn_bwd_apply----> d(raw) in place (+ column sums = conv bias gradient)
            --conv wgrad------> packed dW (split-K atomics) --unpack--> weight.grad (OIHW)
            --conv dgrad------> d(input): the forward conv kernel over d(raw) with flipped / transposed weights
                                (stride 2: zero insertion first); a second contribution is added through ``resid``

The consumer-side fused BN + ReLU of the forward (deferred BatchNorm) needs no activation buffer here either: wgrad
re-applies the transform to the raw input, and the producer's BN backward recomputes its ReLU mask from raw.
The deformable conv is HIP too (dcn.hip: col2im with atomics).  Self-attention blocks have no hand-written backward
kernels yet: inside this plan they run block-local ATen autograd (token-major matmul / softmax); everything else is HIP.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import lib


class BackwardPlan:
    def __init__(self, plan):
        self.plan = plan
        self.B, self.dev = plan.B, plan.dev
        self.steps = []
        self.keep = []
        self.grads = {}          # id(param) -> fp32 grad tensor
        self.zero_list = []      # tensors to zero before every run
        self.gbuf = {}           # data_ptr of a forward tensor -> gradient buffer w.r.t. it
        # every parameter gradient is a 16-byte aligned slice of ONE flat fp32 tensor: the autograd glue hands these views out
        # as ``param.grad`` without copying, and data-parallel training all-reduces ``self.flat`` in a single call
        offs, n = [], 0
        for p in plan.eng.net.parameters():
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(max(n, 4), device=self.dev, dtype=torch.float32)
        self._slices = {id(p): self.flat[o:o + p.numel()].view(p.shape) for p, o in zip(plan.eng.net.parameters(), offs)}
        self.dloc = torch.empty(self.B, plan.P, 4, device=self.dev)
        self.dconf = torch.empty(self.B, plan.P, plan.nc, device=self.dev)
        net = plan.eng.net
        first_in = plan.rec[0][1]['x_in'].data_ptr()
        for kind, r in reversed(plan.rec):
            if kind == 'head':
                self._head(r)
            elif kind == 'convbn':
                self._convbn(r, need_dgrad=(r['x_in'].data_ptr() != first_in))
            elif kind == 'convrelu':
                self._convrelu(r, need_dgrad=(r['x_in'].data_ptr() != first_in))
            elif kind == 'pool':
                self._pool(r)
            elif kind == 'l2norm':
                self._l2norm(r)
            elif kind == 'sa':
                self._sa(r)
            elif kind == 'slice_cat':
                self._slice_cat(r)
            elif kind == 'dcn':
                self._dcn(r)
            else:
                raise _lib.GssdError(f'no HIP backward for {kind}')
        self.param_order = [p for p in net.parameters()]

    # ------------------------------------------------------------------------------------------------
    def _add(self, fn, args, keep=None):
        self.steps.append((fn, args))
        if keep is not None:
            self.keep.append(keep)

    def _buf(self, *shape, dtype=torch.float32, zero_each_run=False):
        t = torch.empty(*shape, device=self.dev, dtype=dtype)
        self.keep.append(t)
        if zero_each_run:
            self.zero_list.append(t)
        return t

    def _pgrad(self, p):
        g = self.grads.get(id(p))
        if g is None:
            g = self._slices[id(p)]              # parameters the plan never writes keep no entry -> gradient None
            self.grads[id(p)] = g
        return g

    def _grad_of(self, t):
        return self.gbuf.get(t.data_ptr())

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
        ud = None
        from .engine import USE_WINOGRAD
        if USE_WINOGRAD and ops.winograd_eligible(k, 1, pd, dil, Cout // groups, Cin // groups, groups):
            ud = self._buf(int(lib.gssd_winograd_weight_elems(Cin, groups, Cout // groups)))
            self._add(lib.gssd_winograd_weight_f32, (wd.data_ptr(), ud.data_ptr(), Cin, groups, Cout // groups, wd.stride(0)))
        d, Hout, _ = ops.make_conv_desc(src, wd, g, B=B, H=Hs, W=Hs, in_stride=Cout, cin_g=Cout // groups, Cout=Cin,
                                        groups=groups, k=k, pad=pd, dil=dil, resid=existing, wgt_wino=ud)
        assert Hout == H, (Hout, H)
        self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d),), keep=d)
        self.gbuf[x_in.data_ptr()] = g

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
        fdesc, _, _ = ops.make_conv_desc(r['src'], None, None, B=B, H=H, W=H, in_stride=Cs, cin_g=Cs, Cout=Cout, k=3, pad=1)
        dwp, K = self._wgrad(fdesc, dyh, None, Cs, Cs, 3, Cout)
        self._unpack(dwp, K, 0, r['loc'].weight, Cs, Cs, 3)
        self._unpack(dwp, K, A * 4, r['conf'].weight, Cs, Cs, 3)
        cs = self._buf(Cout, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dyh.data_ptr(), B * H * H, Cout, Cout, cs.data_ptr()))
        self._bias_from_colsum(cs, r['loc'].bias, 0)
        self._bias_from_colsum(cs, r['conf'].bias, A * 4)
        # d(source): the merged head weight [Cout][9*Cs] viewed as one conv
        wd = self._buf(Cs, 9 * Cout)
        hw = self.plan.eng._packed[f"heads.{r['i']}.w"]          # packed forward rows [Cout][9*Cs] (k = tap*Cs + c)
        self._add(_pack_dgrad_from_packed, (hw, wd, Cout, Cs))
        existing = self._grad_of(r['src'])
        g = existing if existing is not None else self._buf(B, H, H, Cs)
        d, _, _ = ops.make_conv_desc(dyh, wd, g, B=B, H=H, W=H, in_stride=Cout, cin_g=Cout, Cout=Cs, k=3, pad=1, resid=existing)
        self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d),), keep=d)
        self.gbuf[r['src'].data_ptr()] = g

    def _convbn(self, r, need_dgrad=True):
        B, H, Ho, Hp, Cin, Cout, groups = self.B, r['H'], r['Ho'], r['Hp'], r['Cin'], r['Cout'], r['groups']
        conv, bn, raw = r['conv'], r['bn'], r['raw']
        dout = self._grad_of(r['out'])
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
                                                 pd_.data_ptr()))
        pool = r['pool']
        pk, ps, pp = (pool[0], pool[1], pool[2]) if pool else (0, 1, 0)
        dz = self._buf(B, Ho, Ho, Cout, zero_each_run=bool(pool and ps < pk))
        sums = self._buf(2 * Cout, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_bn_bwd_reduce_f32, (dout.data_ptr(), raw.data_ptr(), sc.data_ptr(), sh.data_ptr(), dz.data_ptr(),
                                               sums.data_ptr(), B, Ho, Ho, Cout, Hp, Hp, pk, ps, pp, int(r['relu'])))
        ca, cb, cc = self._buf(Cout), self._buf(Cout), self._buf(Cout)
        self._add(lib.gssd_bn_bwd_finalize_f32, (r['stats'].data_ptr(), float(B * Ho * Ho), sums.data_ptr(), bn.weight.data_ptr(),
                                                 float(bn.eps), Cout, ca.data_ptr(), cb.data_ptr(), cc.data_ptr(),
                                                 self._pgrad(bn.weight).data_ptr(), self._pgrad(bn.bias).data_ptr()))
        cs = self._buf(Cout, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_bn_bwd_apply_f32, (dz.data_ptr(), raw.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(),
                                              B * Ho * Ho, Cout, cs.data_ptr()))
        self._bias_from_colsum(cs, conv.bias)
        # weight gradient (the forward descriptor carries the input geometry and the fused input transform)
        cin_g_pad = Cin // groups
        cin_g_real = conv.weight.shape[1]
        dwp, K = self._wgrad(r['desc'], dz, conv, cin_g_real, cin_g_pad, r['k'], Cout)
        self._unpack(dwp, K, 0, conv.weight, cin_g_real, cin_g_pad, r['k'])
        if need_dgrad:
            self._dgrad(r, dz, r['x_in'], conv, groups, Cin, H, Ho, Cout, r['k'], r['stride'], r['pad'], r['dil'])

    def _convrelu(self, r, need_dgrad=True):
        """conv + ReLU of the vanilla SSD (models/ssd.py:104-118, no BatchNorm): dz = d(out) * [out > 0] (the mask of the
        stored post-ReLU output equals the pre-activation's), bias gradient = column sums, then wgrad / dgrad."""
        B, H, Ho, Cin, Cout, conv, out = self.B, r['H'], r['Ho'], r['Cin'], r['Cout'], r['conv'], r['out']
        dout = self._grad_of(out)
        if dout is None:
            raise _lib.GssdError(f"no gradient reaches {r['name']}")
        dz = self._buf(B, Ho, Ho, Cout)
        self._add(lib.gssd_bn_bwd_reduce_f32, (dout.data_ptr(), out.data_ptr(), 0, 0, dz.data_ptr(), 0, B, Ho, Ho, Cout, Ho, Ho,
                                               0, 1, 0, 1))
        cs = self._buf(Cout, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dz.data_ptr(), B * Ho * Ho, Cout, Cout, cs.data_ptr()))
        self._bias_from_colsum(cs, conv.bias)
        cin_real = conv.weight.shape[1]
        dwp, K = self._wgrad(r['desc'], dz, conv, cin_real, Cin, r['k'], Cout)
        self._unpack(dwp, K, 0, conv.weight, cin_real, Cin, r['k'])
        if need_dgrad:
            self._dgrad(r, dz, r['x_in'], conv, 1, Cin, H, Ho, Cout, r['k'], r['stride'], r['pad'], r['dil'])

    def _pool(self, r):
        B, H, Cc, Hp = self.B, r['H'], r['C'], r['Hp']
        dout = self._grad_of(r['out'])
        existing = self._grad_of(r['x_in'])
        assert existing is None, 'pool backward must be the first contribution to its input'
        g = self._buf(B, H, H, Cc, zero_each_run=(r['s'] < r['k']))
        self._add(lib.gssd_bn_bwd_reduce_f32, (dout.data_ptr(), r['x_in'].data_ptr(), 0, 0, g.data_ptr(), 0, B, H, H, Cc, Hp, Hp,
                                               r['k'], r['s'], r['p'], 0))
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
        """Self_Attn (layers/self_attn.py:46-89), interim: no hand-written kernels yet.  The block's backward is written out
        by hand over the forward plan's own buffers (theta|phi, g^T, the softmaxed attention map, attn.g) as a dozen batched
        ATen matmuls -- no autograd graph and no recomputation of the forward:
            d_o = sigma dT;  d(ag) = d_o W_o;  dA = d(ag) g^T;  dS = A (dA - rowsum(A dA));  d theta = dS phi;
            d phi = dS^T theta;  d g = A^T d(ag);  dx = d(out) + d(theta|phi) W_tp + d g W_g;  weight grads = d^T x,
        and for every spectrally normalised conv (W_eff = W / s, s = u^T W v, u and v constants of the step)
            dW = dW_eff / s - (sum dW_eff . W) / s^2 * u v^T."""
        sa, x, out, out2 = r['mod'], r['x_in'], r['out'], r['out2']
        g_out = self._grad_of(out)
        g_out2 = self._grad_of(out2) if out2 is not None else None
        gx, existed = self._reserve(x, x.shape)
        B, N, Cc = self.B, r['N'], r['C']
        C8, C2, C4 = Cc // 8, Cc // 2, Cc // 4
        tp, gT, ag = r['tp'], r['gT'], r['ag']
        a_tpg, a_o = r['inv_sigma']
        cv = {k: getattr(sa, 'snconv1x1_' + k) for k in ('theta', 'phi', 'g', 'attn')}
        pg = {k: (self._pgrad(m.weight_orig), self._pgrad(m.bias)) for k, m in cv.items()}
        pg_sigma = self._pgrad(sa.sigma)

        def sn_grad(conv, dw_eff, inv_s, dst):
            """dW_orig from dW_eff (both [rows, cols])."""
            w = conv.weight_orig.detach().view(dw_eff.shape)
            dot = (dw_eff * w).sum()
            dst.copy_((dw_eff * inv_s - torch.outer(conv.weight_u.detach(), conv.weight_v.detach()) * (dot * inv_s * inv_s))
                      .view(dst.shape))

        def step():
            X = x.view(B, N, Cc)
            dT = g_out.view(B, N, Cc)
            if g_out2 is not None:
                dT = dT + g_out2.view(B, N, Cc)
            sig = sa.sigma.detach()
            is_t, is_p, is_g, is_o = a_tpg[0], a_tpg[C8], a_tpg[C4], a_o[0]       # 1 / sigma_sn of each conv (device scalars)
            w_o = cv['attn'].weight_orig.detach().view(Cc, C2)
            w_g = cv['g'].weight_orig.detach().view(C2, Cc)
            w_t = cv['theta'].weight_orig.detach().view(C8, Cc)
            w_p = cv['phi'].weight_orig.detach().view(C8, Cc)
            theta, phi = tp[:, :, :C8], tp[:, :, C8:]
            A = torch.softmax(torch.bmm(theta, phi.transpose(1, 2)), dim=-1)     # the forward is flash-style: no stored map
            g_tok = gT[:, :, :N].transpose(1, 2)                                 # [B, N, C2] view
            # output conv and the gate
            o_raw = torch.matmul(ag, w_o.t()) * is_o + cv['attn'].bias.detach()
            pg_sigma.copy_((dT * o_raw).sum().view(pg_sigma.shape))
            d_o = dT * sig
            sn_grad(cv['attn'], torch.matmul(d_o.reshape(-1, Cc).t(), ag.reshape(-1, C2)), is_o, pg['attn'][0])
            pg['attn'][1].copy_(d_o.sum((0, 1)))
            dag = torch.matmul(d_o, w_o) * is_o                                  # [B, N, C2]
            # attention
            dA = torch.bmm(dag, g_tok.transpose(1, 2))                           # [B, N, N]
            dS = A * (dA - (A * dA).sum(-1, keepdim=True))
            d_theta = torch.bmm(dS, phi)
            d_phi = torch.bmm(dS.transpose(1, 2), theta)
            d_g = torch.bmm(A.transpose(1, 2), dag)                              # [B, N, C2] (token-major g)
            # the three input convs
            Xf = X.reshape(-1, Cc)
            sn_grad(cv['theta'], torch.matmul(d_theta.reshape(-1, C8).t(), Xf), is_t, pg['theta'][0])
            sn_grad(cv['phi'], torch.matmul(d_phi.reshape(-1, C8).t(), Xf), is_p, pg['phi'][0])
            sn_grad(cv['g'], torch.matmul(d_g.reshape(-1, C2).t(), Xf), is_g, pg['g'][0])
            pg['theta'][1].copy_(d_theta.sum((0, 1)))
            pg['phi'][1].copy_(d_phi.sum((0, 1)))
            pg['g'][1].copy_(d_g.sum((0, 1)))
            # the residual branch carries d(out) only: out2 = sigma * o has no direct x term
            dx = g_out.view(B, N, Cc) + torch.matmul(d_theta, w_t) * is_t + torch.matmul(d_phi, w_p) * is_p + torch.matmul(d_g, w_g) * is_g
            dxn = dx.view(gx.shape)
            gx.add_(dxn) if existed else gx.copy_(dxn)
        self.steps.append((step, None))

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
        cols = self._buf(B * H * H, Kc)
        self._add(lib.gssd_dcn_im2col_f32, (x.data_ptr(), om.data_ptr(), cols.data_ptr(), B, H, H, Cin, dg, 27 * dg))
        w_main = self._buf(Cout, Kc)
        self._add(lib.gssd_pack_conv_weight, (m.weight.data_ptr(), w_main.data_ptr(), Cout, Cin, 3, 3, Cin, Kc), keep=m)
        # main weight / bias: dW[Cout][9*Cin] = dY^T . cols, d(cols) = dY . W  -- plain GEMMs (rocBLAS)
        dwp = self._buf(Cout, Kc)
        self._add(lib.gssd_gemm_tn_f32, (dy.data_ptr(), cols.data_ptr(), dwp.data_ptr(), B * H * H, Cout, Kc, Cout, Kc, Kc, 0))
        self._unpack(dwp, Kc, 0, m.weight, Cin, Cin, 3)
        cs = self._buf(Cout, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dy.data_ptr(), B * H * H, Cout, Cout, cs.data_ptr()))
        self._bias_from_colsum(cs, m.bias)
        wt = self._buf(Kc, Cout)
        self.steps.append((lambda w=w_main, wt=wt: wt.copy_(w.t()), None))
        dcols = self._buf(B * H * H, Kc)
        self._add(lib.gssd_gemm_nt_f32, (dy.data_ptr(), wt.data_ptr(), dcols.data_ptr(), B * H * H, Kc, Cout, Cout, Cout, Kc, 0, 0),
                  keep=wt)
        # sampling backward: d(x) by atomics, d(offset / mask logits) per pixel
        gx = self._grad_of(x)
        if gx is None:
            gx = self._buf(B, H, H, Cin, zero_each_run=True)
            self.gbuf[x.data_ptr()] = gx
        dom = self._buf(B, H, H, 27 * dg, zero_each_run=True)
        self._add(lib.gssd_dcn_col2im_f32, (x.data_ptr(), om.data_ptr(), dcols.data_ptr(), gx.data_ptr(), dom.data_ptr(), B, H, H,
                                            Cin, dg, 27 * dg))
        # offset / mask conv
        cm = m.conv_offset_mask
        dwo = self._buf(27 * dg, Kc, zero_each_run=True)
        self._add(lib.gssd_conv2d_wgrad_f32, (C.byref(r['d_om']), dom.data_ptr(), dwo.data_ptr()), keep=r['d_om'])
        self._unpack(dwo, Kc, 0, cm.weight, Cin, Cin, 3)
        cs2 = self._buf(27 * dg, dtype=torch.float64, zero_each_run=True)
        self._add(lib.gssd_colsum_f32, (dom.data_ptr(), B * H * H, 27 * dg, 27 * dg, cs2.data_ptr()))
        self._bias_from_colsum(cs2, cm.bias)
        self._dgrad(r, dom, x, cm, 1, Cin, H, H, 27 * dg, 3, 1, 1, 1)

    # ------------------------------------------------------------------------------------------------
    def run(self, dloc, dconf):
        self.dloc.copy_(dloc)
        self.dconf.copy_(dconf)
        if self.zero_list:
            torch._foreach_zero_(self.zero_list)
        stream = torch.cuda.current_stream().cuda_stream
        for fn, args in self.steps:
            if args is None:                     # block-local ATen autograd callback
                fn()
                continue
            if fn is _pack_dgrad_from_packed:
                fn(*args)
                continue
            rc = fn(*args, stream)
            if rc != 0:
                _lib.check(rc)
        return [self.grads.get(id(p)) for p in self.param_order]


def conv_weight_ptr(conv, cin_g_expected):
    """OIHW weight pointer for the dgrad packer (conv1_1 never needs a dgrad, so no channel padding arises)."""
    assert conv.weight.shape[1] == cin_g_expected
    return conv.weight.data_ptr()


def _pack_dgrad_from_packed(hw, wd, Cout, Cs):
    """Merged head weights [Cout][9][Cs] (forward packing) -> dgrad rows [Cs][9 flipped][Cout] (tiny: plain tensor ops)."""
    wd.copy_(hw.view(Cout, 9, Cs).flip(1).permute(2, 1, 0).reshape(Cs, 9 * Cout))
