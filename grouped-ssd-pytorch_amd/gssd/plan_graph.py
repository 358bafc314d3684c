"""The WALK over the module tree (models/ssd_multiphase_custom_group.py:217-400 restated as launches): which layer follows which, where the
branches fork, where the heads reduce.  Mixin of engine._Plan; the per-op emitters it calls live in plan_ops.py."""
import ctypes as C
import os
import torch
from . import _lib, ops
from ._lib import lib
from .plan_common import ALL_STREAMS, FUSE_NAMES, SN_STREAM, VGG_CFG


class PlanGraphMixin:
    def _place_branch0(self):
        """Branch 0 (L2Norm -> [SA] -> fuse_11 -> head on the 38 x 38 map: ~1.2 ms of chip-filling launches in GSSD++) was registered right
        behind the block after conv4_3, so its stream forked there and its launches shared the CUs with conv5_x / conv6 / conv7 -- the
        critical path, which then ran 1.5 - 2.5 x slower than alone (profiles/r04b_critical_path_f32.txt).  Registered behind conv7
        instead, the branch forks there: the trunk's heavy layers run alone, and the branch fills the chip under the small-map tail
        (SA-base, extras), whose launches have 1 .. 100 workgroups.  GSSD_BRANCH0_LATE=0 keeps the registration order."""
        # (measured: GSSD++ 12.14 -> 12.10 ms fp32, 3.89 -> 3.86 ms bf16; plain GSSD, whose branch 0 is two small launches, 5.55 -> 5.63 ms)
        if (os.environ.get('GSSD_BRANCH0_LATE', '1') == '0' or getattr(self, '_mark_conv7', None) is None
                or not self.eng.net.use_self_attention):
            return
        idx = [i for i, st in enumerate(self.steps) if st.sid == 1]
        if not idx or idx[-1] - idx[0] + 1 != len(idx) or idx[-1] >= self._mark_conv7:
            return                                    # (not one contiguous block in front of the mark: leave the order alone)
        a, b, c = idx[0], idx[-1] + 1, self._mark_conv7
        block = self.steps[a:b]
        self.steps[a:c] = self.steps[b:c] + block     # indices < a and >= c are unchanged (_pack_step, _reduce_steps)

    def _place_sn_step(self):
        """The spectral-norm launch (its own stream inside the captured graph) was registered first, which makes it a ROOT node of the
        hipGraph beside the input pack -- and the round-3 timeline (profiles/r04_critical_path_*.txt) shows the runtime then runs the two
        roots one after the other: 0.56 ms of a 48-workgroup kernel in front of every step.  Registered behind conv1_1 it forks from the
        trunk there and runs beside conv1_2 .. conv4_3 (its 1/sigma vectors are first read by the Self_Attn block behind conv4_3)."""
        sn = next((i for i, st in enumerate(self.steps) if st.sid == SN_STREAM), None)
        if sn is None or sn > self._pack_step:
            return
        st = self.steps.pop(sn)                       # (sn == 0: registered before the pack step)
        self._pack_step -= 1
        self.steps.insert(self._pack_step + 2, st)    # behind pack_input and conv1_1; every later index is unchanged

    def _build_bn_graph(self, x16):
        """models/...group.py:254-372, batch_norm=True (the driver's graph, train_lesion_multiphase_v2.py:77)."""
        net = self.eng.net
        g = net.groups_vgg
        # ---- trunk -------------------------------------------------------------------------------------
        cur, H, Cc = x16, 300, self.cpad * g
        vi = 0
        cfg = list(VGG_CFG)
        i = 0
        x43 = None
        xf = None
        while i < len(cfg):
            v = cfg[i]
            assert v not in ('M', 'C')
            conv, bn = net.vgg[vi], net.vgg[vi + 1]
            pool = None
            nxt = cfg[i + 1] if i + 1 < len(cfg) else None
            is_conv4_3 = (vi == 30)
            last = (i == len(cfg) - 1)
            if nxt in ('M', 'C') and not is_conv4_3:
                pool = (2, 2, 0, nxt == 'C')
            if last:
                pool = (3, 1, 1, False)               # pool5
            # A conv whose only consumer is the next conv (no pool, not a multibox source) leaves its BatchNorm + ReLU to
            # that consumer, which applies scale/shift/ReLU on the fragments it reads: one HBM round trip less per layer
            # (737 MB for conv1_1).  Pooled layers and sources keep the separate BN + ReLU (+ pool) pass.
            defer = (pool is None and not is_conv4_3)
            cur, H, Cc, xf = self._conv_bn(f'vgg.{vi}', conv, bn, cur, H, Cc, g, relu=True, pool=pool, in_xf=xf,
                                           defer_bn=defer)
            vi += 3
            if nxt in ('M', 'C'):
                vi += 1
                i += 1
            i += 1
            if is_conv4_3:
                x43 = cur
                cur, H, Cc, src0 = self._after_conv4_3(cur, H, Cc)
        vi += 1   # pool5 module
        xf = None
        for li in range(2):                                 # conv6 (BN deferred into conv7), conv7
            conv, bn = net.vgg[vi], net.vgg[vi + 1]
            cur, H, Cc, xf = self._conv_bn(f'vgg.{vi}', conv, bn, cur, H, Cc, g, relu=True, in_xf=xf, defer_bn=(li == 0))
            vi += 3
        sources = [src0]
        self._mark_conv7 = len(self.steps)           # (everything up to conv7's BatchNorm pass is enqueued: _place_branch0)
        sab_i, sa_i = 1, 1
        if net.use_self_attention_base:
            cur, _ = self._self_attn('self_attn_base_list', sab_i, cur, H, Cc, need_out2=False, want_map=self.want_maps)
            sab_i += 1
        sources.append(self._branch(cur, H, Cc, sa_i, '21'))
        sa_i += 1
        # ---- extras --------------------------------------------------------------------------------------
        ge = net.groups_extra
        n_ex = len(net.extras)
        fi = 2
        xf = None
        for k in range(0, n_ex, 2):
            conv, bn = net.extras[k], net.extras[k + 1]
            cur, H, Cc, xf = self._conv_bn(f'extras.{k}', conv, bn, cur, H, Cc, ge, relu=True, in_xf=xf,
                                           defer_bn=((k + 1) % 4 != 3))
            if (k + 1) % 4 == 3:
                if net.use_self_attention_base:
                    cur, _ = self._self_attn('self_attn_base_list', sab_i, cur, H, Cc, need_out2=False, want_map=self.want_maps)
                    sab_i += 1
                sources.append(self._branch(cur, H, Cc, sa_i, FUSE_NAMES[fi]))
                sa_i += 1
                fi += 1
        self.sources = sources

    def _build_plain_graph(self, x16):
        """batch_norm=False (models/...group.py:254-256, 329-349; vgg() / add_extras() without BatchNorm, multibox sources [21, -2]):
        every conv carries bias + ReLU in its epilogue, pools are the identity-affine pool pass, fuse convs have no BatchNorm."""
        net = self.eng.net
        g, ge = net.groups_vgg, net.groups_extra
        cur, H, Cc = x16, 300, self.cpad * g
        mods = list(net.vgg)
        i = 0
        src0 = None
        while i < len(mods):
            m = mods[i]
            if isinstance(m, torch.nn.Conv2d):
                cur, H, Cc = self._conv_act(f'vgg.{i}', m, cur, H, Cc, g)
                i += 2                                   # conv + ReLU
                if i - 2 == 21:                          # conv4_3 (ReLU at 22, idx_until_conv4_3 = 23): the first source's block
                    cur, H, Cc, src0 = self._after_conv4_3(cur, H, Cc)      # ... which also runs pool4 (vgg[23])
                    i += 1
            else:
                cur, H = self._pool_only(cur, H, Cc, m.kernel_size, m.stride, m.padding, m.ceil_mode)
                i += 1
        sources = [src0]
        sab_i, sa_i = 1, 1
        if net.use_self_attention_base:
            cur, _ = self._self_attn('self_attn_base_list', sab_i, cur, H, Cc, need_out2=False, want_map=self.want_maps)
            sab_i += 1
        sources.append(self._branch(cur, H, Cc, sa_i, '21'))
        sa_i += 1
        fi = 2
        for k, m in enumerate(net.extras):
            cur, H, Cc = self._conv_act(f'extras.{k}', m, cur, H, Cc, ge)
            if k % 2 == 1:
                if net.use_self_attention_base:
                    cur, _ = self._self_attn('self_attn_base_list', sab_i, cur, H, Cc, need_out2=False, want_map=self.want_maps)
                    sab_i += 1
                sources.append(self._branch(cur, H, Cc, sa_i, FUSE_NAMES[fi]))
                sa_i += 1
                fi += 1
        self.sources = sources

    def _finish_heads(self):
        """Deterministic split-K for the heads: every reduction slice of a head conv writes its partial sums to its own copy of the
        outputs (GSSD_CONV_HEADS_SLICES); two launches then add the slices of every prior in order into loc / conf.  (With fp32
        atomics loc / conf -- and with them Detect's index output -- differed in their last bits from run to run.)"""
        B, dev = self.B, self.dev
        smax = max(d.split_k for d in self.head_descs)
        splits = torch.ones(self.P, dtype=torch.int8)
        off = 0
        for d in self.head_descs:
            A = d.split_n // 4
            n = d.Ho * d.Wo * A
            splits[off:off + n] = d.split_k
            off += n
        assert off == self.P
        self._head_splits = splits.to(dev)
        self._ws_loc = torch.empty(smax, B, self.P, 4, device=dev, dtype=torch.float32)
        self._ws_conf = torch.empty(smax, B, self.P, self.nc, device=dev, dtype=torch.float32)
        for d in self.head_descs:
            d.out, d.out_b = self._ws_loc.data_ptr(), self._ws_conf.data_ptr()
            d.flags |= _lib.CONV_HEADS_SLICES
        prev, self._sid = getattr(self, '_sid', 0), 0
        self._pending_wait = ALL_STREAMS                      # the heads run on the branch streams: join them all first
        self._reduce_steps = (len(self.steps), len(self.steps) + 1)
        self._add(lib.gssd_heads_reduce_f32, [self._ws_loc.data_ptr(), self._head_splits.data_ptr(), 0, B, self.P, 4])
        self._add(lib.gssd_heads_reduce_f32, [self._ws_conf.data_ptr(), self._head_splits.data_ptr(), 0, B, self.P, self.nc])
        self._sid = prev

    def _set_outputs(self, loc, conf):
        self.steps[self._reduce_steps[0]].args[2] = loc.data_ptr()
        self.steps[self._reduce_steps[1]].args[2] = conf.data_ptr()

    def _after_conv4_3(self, x, H, Cc):
        """models/...group.py:261-298: [SA-base] -> [slice_and_cat] -> [DCN]* -> L2Norm -> [SA] -> fuse_11; pool4."""
        net, B = self.eng.net, self.B
        attn_g = None
        if net.use_self_attention_base:
            x, attn_g = self._self_attn('self_attn_base_list', 0, x, H, Cc, need_out2=bool(net.dcn_cat_sab), want_map=self.want_maps)
        if net.use_dcn:
            xin, Cin = x, Cc
            if net.dcn_cat_sab:
                xc = self._abuf(B, H, H, 2 * Cc)
                esz = 2 if self.bf16 else 1          # a pure copy: bf16 pairs travel as one 4-byte word
                self._add(lib.gssd_slice_and_cat_f32, (x.data_ptr(), attn_g.data_ptr(), xc.data_ptr(), B * H * H, Cc // esz, Cc // esz,
                                                       net.groups_vgg))
                self.rec.append(('slice_cat', dict(a=x, b=attn_g, out=xc, H=H, Ca=Cc, Cb=Cc, groups=net.groups_vgg,
                                                   detach_b=bool(net.detach_sab))))
                xin, Cin = xc, 2 * Cc
            for li in range(net.num_dcn_layers):
                xin, Cin = self._dcn(li, xin, H, Cin)
            x, Cc = xin, Cin
        self.x_after_block = x
        s = self._abuf(B, H, H, Cc)
        self._sid = 1                              # L2Norm opens branch 0
        self._add(lib.gssd_l2norm_bf16 if self.bf16 else lib.gssd_l2norm_f32,
                  (x.data_ptr(), net.L2Norm.weight.data_ptr(), s.data_ptr(), B * H * H, Cc, float(net.L2Norm.eps)))
        self.rec.append(('l2norm', dict(x_in=x, out=s, H=H, C=Cc, mod=net.L2Norm)))
        self._sid = 0
        src0 = self._branch(s, H, Cc, 0, '11')
        self._layer = 'vgg.33'                     # pool4: a trunk pass (conv5_1 reads it)
        pooled, Hp = self._pool_only(x, H, Cc, 2, 2, 0)
        self._layer = None
        return pooled, Hp, Cc, src0

    def _branch(self, s, H, Cc, sa_i, fuse):
        """[SA] -> 1x1 fuse conv + BN + ReLU -> a multibox source (models/...group.py:284-297) -> its loc | conf head.  Nothing
        downstream of the trunk reads a branch, so branch i is tagged with stream id i + 1: captured as a hipGraph the six branches
        run beside the trunk's continuation (on the small maps a kernel has 1..100 workgroups for 256 CUs)."""
        net = self.eng.net
        prev, self._sid = getattr(self, '_sid', 0), sa_i + 1
        if net.use_self_attention:
            s, _ = self._self_attn('self_attn_list', sa_i, s, H, Cc, need_out2=False, want_map=self.want_maps)
        if net.use_fuseconv and net.batch_norm:
            conv, bn = getattr(net, f'fuse_{fuse}'), getattr(net, f'bn_fuse_{fuse}')
            s, H, Cc, _ = self._conv_bn(f'fuse_{fuse}', conv, bn, s, H, Cc, 1, relu=True)
        elif net.use_fuseconv:
            s, H, Cc = self._conv_act(f'fuse_{fuse}', getattr(net, f'fuse_{fuse}'), s, H, Cc, 1)
        self._head(sa_i, s, H, Cc)
        self._sid = prev
        return (s, H, Cc)
