"""bf16 storage mode: what the mixed-precision backward reads of a bf16 forward plan -- fp32 shadows of the stored maps only where a step still
needs one, the bf16 maps themselves everywhere else (Bf16Shadow), and the engine view the backward's emitters use."""
import ctypes as C
import os
import torch
from . import _lib, ops
from ._lib import lib
from .bwd_common import BWD_BF16


class _EngView:
    """What BackwardPlan reads of the engine, with fp32 packed weights for a bf16 forward plan."""

    def __init__(self, eng, packed):
        self.net, self._packed = eng.net, packed


class Bf16Shadow:
    """fp32 view of a bf16-storage forward plan (BASELINE.json configs[4] as a TRAINING config,
    train_lesion_multiphase_v2.py:242-253): the backward of the bf16 mode is mixed precision -- the forward computed and STORED its
    activations in bf16 (bf16 MFMA, fp32 accumulate); the backward differentiates the fp32 graph AT the rounded activations
    (straight-through rounding; the oracle's ``bf16='ste'``) with fp32 master weights, fp32 gradients and fp32 accumulation.
    This class gives BackwardPlan the records of the forward plan with fp32 shadows of the stored maps (the plan keys its gradient
    buffers by them) and rebuilds, at the start of every backward (``pre`` steps, on the main stream before the branch streams fork),
    what the bf16 forward does not keep in a form the backward reads:
      * gssd_cast_bf16_f32 of a stored map -- only where a step reads the fp32 CONTENT (``need``; eager for pools, L2Norm, Self_Attn,
        DCN sampling, heads).  With GSSD_BWD_BF16 (default) the conv + BatchNorm layers read the bf16 maps themselves
        (gssd_bn_bwd_*_mixed, gssd_conv2d_wgrad_bf16: records carry ``x16`` / ``raw16`` / ``src16`` / ``ag16``), their shadows stay
        uncast; GSSD_BWD_BF16=0 casts every map (the round-3 form: the fp32 backward plan on fp32 copies);
      * the pad vectors of the deferred BatchNorms in fp32 (gssd_bn_finalize_f32, mode 2 = no second running-statistics update);
      * Self_Attn's value projection g^T in fp32 token order (the forward's copy is bf16 in the 32-key order of the bf16 attention
        core): one fp32 1x1 conv per block from the fp32 copy of the block's input (the rows' log-sum-exp is the forward's own);
      * fp32 packed copies of the weights the backward reads in packed form (merged head rows, merged theta | phi | g rows)."""

    def __init__(self, plan):
        self.real = plan
        self.B, self.dev, self.P, self.nc = plan.B, plan.dev, plan.P, plan.nc
        self.training = plan.training
        self._keep = []
        self._s = {}                 # data_ptr of a bf16 forward tensor -> its fp32 shadow
        self._xf = {}                # data_ptr of a deferred BatchNorm's scale vector -> (scale, shift, pad_fp32)
        packed = {}
        self.eng = _EngView(plan.eng, packed)
        eng, net, B, dev = plan.eng, plan.eng.net, plan.B, plan.dev
        f32 = torch.float32
        casts, later = [], []

        self._lazy = {}              # data_ptr of an fp32 shadow whose cast is not scheduled (yet) -> the cast step

        def S(t, lazy=False):
            """fp32 shadow of a stored bf16 map.  ``lazy``: the shadow exists (the backward plan keys its gradient buffers by it) but the
            cast runs only if some step asks for its CONTENT (``need``): the conv + BatchNorm layers read the bf16 maps themselves
            (gssd_bn_bwd_*_mixed, gssd_conv2d_wgrad_bf16), so most trunk maps are never copied."""
            if t is None or t.dtype == f32:
                return t
            k = t.data_ptr()
            if k not in self._s:
                sh = torch.empty(t.shape, device=dev, dtype=f32)
                self._s[k] = sh
                self._keep.append((t, sh))
                self._lazy[sh.data_ptr()] = (lib.gssd_cast_bf16_f32, (t.data_ptr(), sh.data_ptr(), t.numel()))
            sh = self._s[k]
            if not lazy:
                self.need(sh)
            return sh
        self._casts = casts

        def XF(xf, bn, stats, count, Cc, srep=0):
            if xf is None:
                return None
            if xf[0].data_ptr() not in self._xf:
                sc, sh = xf[0], xf[1]
                pd = torch.empty(Cc, device=dev, dtype=f32)
                sc2, sh2 = torch.empty(Cc, device=dev, dtype=f32), torch.empty(Cc, device=dev, dtype=f32)
                self._keep.append((xf, pd, sc2, sh2))
                later.append((lib.gssd_bn_finalize_f32, (stats.data_ptr(), float(count), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                                         bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum),
                                                         float(bn.eps), 2, Cc, sc2.data_ptr(), sh2.data_ptr(), pd.data_ptr(), srep)))
                self._xf[xf[0].data_ptr()] = (sc, sh, pd)  # the forward's own scale / shift (identical values), an fp32 pad
            return self._xf[xf[0].data_ptr()]

        rec = []
        first = True
        for kind, r in plan.rec:
            q = dict(r)
            if kind == 'convbn' and first and r['groups'] == 4 and r['Cin'] == 32 and r['conv'].weight.shape[1] == 3:
                # conv1_1: the bf16 forward packs 3 -> 8 channels per phase; its weight gradient wants the fp32 layout (3 -> 4, the
                # patch-staged thin wgrad kernel) -- 2.5 ms per step on the generic kernel otherwise.  Same (bf16-rounded) values,
                # channels 0 .. 3 of every phase: a strided copy (tensor bookkeeping, like slice_and_cat's gradient split)
                x8 = r['x_in']
                x4 = torch.empty(*x8.shape[:3], 16, device=dev, dtype=f32)
                self._s[x8.data_ptr()] = x4
                self._keep.append((x8, x4))

                def repack(x8=x8, x4=x4):
                    x4.view(*x4.shape[:3], 4, 4).copy_(x8.view(*x8.shape[:3], 4, 8)[..., :4])
                casts.append((repack, None))
                q['Cin'] = 16
            first = False
            if kind == 'convbn':
                # the stored bf16 input itself, for the weight gradient on the bf16 matrix cores (csrc/conv_wgrad_bf16.hip)
                q['x16'], q['Cin16'] = (r['x_in'], r['Cin']) if r['x_in'].dtype == torch.bfloat16 else (None, None)
                q['raw16'] = r['raw'] if r['raw'].dtype == torch.bfloat16 else None
                lz = BWD_BF16
                q['x_in'], q['raw'] = S(r['x_in'], lz), S(r['raw'], lz)
                q['out'] = q['raw'] if r['out'] is r['raw'] else S(r['out'], lz)
                q['xf'] = XF(r['xf'], r['bn'], r['stats'], B * r['Ho'] * r['Ho'], r['Cout'], r.get('stats_rep', 0))
            elif kind == 'head':
                q['src16'] = r['src'] if r['src'].dtype == torch.bfloat16 else None
                q['src'] = S(r['src'], BWD_BF16)          # (content read only by the fp32 weight-gradient fallback)
            elif kind in ('pool', 'l2norm'):
                q['x_in'], q['out'] = S(r['x_in']), S(r['out'])
            elif kind == 'slice_cat':
                q['a'], q['b'], q['out'] = S(r['a']), S(r['b']), S(r['out'])
            elif kind == 'dcn':
                q['x16'] = r['x_in'] if r['x_in'].dtype == torch.bfloat16 else None
                q['x_in'], q['out'] = S(r['x_in']), S(r['out'])
            elif kind == 'sa':
                bf = torch.bfloat16
                q['x16'], q['ag16'] = (r['x_in'] if r['x_in'].dtype == bf else None), (r['ag'] if r['ag'].dtype == bf else None)
                q['x_in'], q['out'], q['ag'] = S(r['x_in'], BWD_BF16), S(r['out']), S(r['ag'])
                q['out2'] = S(r['out2']) if r['out2'] is not None else None
            else:
                raise _lib.GssdError(f'bf16 backward: no fp32 view for a {kind} record')
            rec.append((kind, q))
        # second pass: what needs the shadows of OTHER records (a consumer's fused input transform is its producer's xf)
        for (kind, r), (_, q) in zip(plan.rec, rec):
            if kind == 'convbn':
                if r['in_xf'] is not None:
                    prod = next(rr for kk, rr in plan.rec if kk == 'convbn' and rr.get('xf') is not None and rr['xf'][0] is r['in_xf'][0])
                    q['in_xf'] = XF(r['in_xf'], prod['bn'], prod['stats'], B * prod['Ho'] * prod['Ho'], prod['Cout'], prod.get('stats_rep', 0))
                ix = q['in_xf']
                d, _, _ = ops.make_conv_desc(q['x_in'], None, q['raw'], B=B, H=r['H'], W=r['H'], in_stride=q['Cin'],
                                             cin_g=q['Cin'] // r['groups'], Cout=r['Cout'], groups=r['groups'], k=r['k'], stride=r['stride'],
                                             pad=r['pad'], dil=r['dil'], in_scale=ix[0] if ix else None, in_shift=ix[1] if ix else None,
                                             in_pad=ix[2] if ix else None)
                q['desc'] = d
            elif kind == 'dcn':
                OMC = r['omc']
                d, _, _ = ops.make_conv_desc(q['x_in'], None, None, B=B, H=r['H'], W=r['H'], in_stride=r['Cin'], cin_g=r['Cin'], Cout=OMC,
                                             k=3, pad=1)
                q['d_om'] = d
            elif kind == 'sa':
                sa, name, Cc, H, N = r['mod'], r['name'], r['C'], r['H'], r['N']
                C8, C2, C4 = Cc // 8, Cc // 2, Cc // 4
                Np4 = ops.round_up(N, 4)
                w32 = torch.empty(C4 + C2, Cc, device=dev, dtype=f32)
                packed[name + '.tpg.w'] = w32

                def refresh(w32=w32, sa=sa, C8=C8, C4=C4):
                    ops.copy_into(w32[:C8], sa.snconv1x1_theta.weight_orig)
                    ops.copy_into(w32[C8:C4], sa.snconv1x1_phi.weight_orig)
                    ops.copy_into(w32[C4:], sa.snconv1x1_g.weight_orig)
                later.append((refresh, None))
                a_tpg = r['inv_sigma'][0]
                if (BWD_BF16 and r['kp'] is None and r.get('lse') is not None and q.get('x16') is not None
                        and lib.gssd_self_attn_flash_bwd_supported(C8, C2)):
                    # flash-style attention backward (csrc/sa_flash_bwd.hip): g token-major in bf16 from the block's bf16 input and the
                    # forward's own bf16 weights -- no fp32 g^T, no [N][N] maps
                    g16 = torch.empty(B, N, C2, device=dev, dtype=torch.bfloat16)
                    wq, bq = plan.eng._packed[name + '.tpg.w@bf16'], plan.eng._packed[name + '.tpg.b@bf16']      # (Engine._pack's bf16-mode keys)
                    d, _, _ = ops.make_conv_desc(q['x16'], wq[C4:], g16, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C2, bias=bq[C4:],
                                                 alpha=a_tpg[C4:])
                    self._keep.append((d, g16, wq, bq))
                    later.append((lib.gssd_conv2d_nhwc_bf16, (C.byref(d),)))
                    q.update(g16=g16, gT=None, Np=Np4, Nkp=Np4)
                    continue
                self.need(q['x_in'])                      # the fp32 g^T conv below reads the block input's fp32 copy
                gT = torch.empty(B, C2, Np4, device=dev, dtype=f32)
                bg = sa.snconv1x1_g.bias.detach()
                d, _, _ = ops.make_conv_desc(q['x_in'], w32[C4:], gT, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C2, bias=bg,
                                             alpha=a_tpg[C4:], out_mode=_lib.OUT_TRANSPOSED, out_stride=Np4, m_per_image=True,
                                             in_batch_stride=N * Cc, out_batch_stride=C2 * Np4)
                self._keep.append((d, gT, w32, bg))
                later.append((lib.gssd_conv2d_nhwc_f32, (C.byref(d),)))
                q.update(gT=gT, Np=Np4, Nkp=Np4)          # (lse: the forward's own, fp32)
            elif kind == 'head':
                i, Cs, A = r['i'], r['C'], r['A']
                nloc = A * 4
                _, K = ops.packed_k(Cs, 3, 3)
                hw = torch.empty(nloc + A * plan.nc, K, device=dev, dtype=f32)
                packed[f'heads.{i}.w'] = hw

                def refresh_h(hw=hw, lw=r['loc'], cw=r['conf'], nloc=nloc):
                    ops.pack_weight(lw.weight, hw, 0)
                    ops.pack_weight(cw.weight, hw, nloc)
                later.append((refresh_h, None))
        self.rec = rec
        self._later = later

    @property
    def pre(self):
        return self._casts + self._later

    def need(self, t):
        """Schedule the cast of a lazily shadowed map (a step is about to read the fp32 CONTENT of ``t``); no-op for anything else."""
        step = self._lazy.pop(t.data_ptr(), None) if t is not None else None
        if step is not None:
            self._casts.append(step)

    def _side_stream(self, sid):
        return self.real._side_stream(sid)
