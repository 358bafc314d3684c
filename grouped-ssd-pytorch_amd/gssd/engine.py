"""Forward engine of the GSSD / GSSD++ detector on MI355X.

``GssdEngine`` turns the module tree built by ``build_ssd`` into a flat launch plan: a list of
(C-ABI function, prebuilt argument tuple) pairs over preallocated NHWC buffers.  Running the plan
is a tight loop of ctypes calls on the current HIP stream -- no tensor ops, no allocation (≈2.5 ms of host time per
GSSD step against 6.5 ms on the GPU, scripts/host_overhead.py).

Dataflow restated from models/ssd_multiphase_custom_group.py:217-400:

  x[B,12,300,300] -> pack NHWC(16 ch: 3->4 per phase) -> grouped VGG (conv -> raw + batch stats; BN+ReLU(+pool))
  conv4_3 -> [SA-base0] -> [slice_and_cat] -> [DCN]* -> x ; s = L2Norm(x) -> [SA0] -> fuse_11+BN+ReLU = source0
  pool4 -> conv5_x -> pool5 -> conv6(dil 6) -> conv7 -> [SA-base1] ; -> [SA1] -> fuse_21 = source1
  extras (conv+BN+ReLU)x8, every second one: [SA-base] ; [SA] -> fuse -> source2..5
  heads: one merged (loc|conf) 3x3 conv per source writing straight into loc[B,8732,4] / conf[B,8732,C]

Weights are re-packed (OIHW -> K-major rows) only when a parameter's version counter changed.
"""
import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import lib
from .plan_common import (VGG_CFG, EXTRAS_CFG, MBOX, SRC_HW, HEAD_OFF, FUSE_NAMES, USE_CONV_X6, USE_WINOGRAD, DCN_X6, USE_GRAPH, USE_FLASH_X6, USE_BRANCH_STREAMS, SN_STREAM, ALL_STREAMS, Tag, _Step, conv_tag)      # noqa: F401 (re-exported: backward.py, pixellink.py, bench.py)
from .plan_exec import PlanExecMixin
from .plan_graph import PlanGraphMixin
from .plan_ops import PlanOpsMixin


class GssdEngine:
    def __init__(self, net):
        self.net = net
        self._plans = {}
        self._packed = {}        # name -> packed weight tensor
        self._pack_jobs = []     # (callable) refreshers
        self._pack_table = None  # (device table, items, kept tensors, jobs that stay individual launches): see _refresh_packed
        self._versions = None
        self._ptrs = None
        self._param_list = None
        self._ptr_list = None
        self._bn_list = None

    # ------------------------------------------------------------------------------------------
    def _state(self):
        """(version counters of the parameters, storage pointers of every parameter AND buffer).  The launch plans hold raw
        device pointers of all of them (BN gamma / beta / running stats, conv biases, L2Norm weight, spectral-norm u / v,
        sigma, ...): a changed pointer (``p.data = ...``, a replaced head, ``bn.running_mean = ...``) rebuilds the plans, a
        changed version only re-runs the weight pack jobs.  The module tree is walked once (nn.Module traversal costs
        ~0.5 ms per call on this net); a module that gains or loses parameters after the first forward must call
        invalidate()."""
        if self._param_list is None:
            self._param_list = list(self.net.parameters())
            self._ptr_list = self._param_list + list(self.net.buffers())
            self._bn_list = [m for m in self.net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        return tuple([p._version for p in self._param_list]), tuple([t.data_ptr() for t in self._ptr_list])

    def invalidate(self):
        for plans in self._plans.values():
            for pl in plans:
                pl.generation += 1        # a backward still holding one of these plans must not use it
        self._plans.clear()
        self._packed.clear()
        self._pack_jobs = []
        self._pack_table = None
        self._versions = None
        self._ptrs = None
        self._param_list = None
        self._ptr_list = None
        self._bn_list = None

    # ------------------------------------------------------------------------------------------
    MAX_PLANS_IN_FLIGHT = 4

    def forward_plan(self, x, training, events=None, want_maps=False, need_backward=True):
        """Run one forward; returns (loc, conf, plan).  A plan whose last grad-enabled forward still awaits its backward is busy
        (gssd/autograd.py) and is never reused: the call takes (or builds) another instance with its own buffers."""
        net = self.net
        if not x.is_cuda:
            raise _lib.GssdError('GSSD HIP engine: input must live on the MI355X (cuda/ROCm tensor); there is no '
                                 'CPU fallback')
        vers, ptrs = self._state()
        if self._ptrs is not None and ptrs != self._ptrs:
            self.invalidate()              # some parameter / buffer storage moved (.cuda(), p.data = ..., new head ...)
            vers, ptrs = self._state()
        p0 = self._param_list[0]
        if p0.device != x.device:
            raise _lib.GssdError(f'model is on {p0.device}, input on {x.device}')
        B = x.shape[0]
        cin = 3 if getattr(net, 'vanilla', False) else 12
        if tuple(x.shape[1:]) != (cin, 300, 300):
            raise _lib.GssdError(f'expected input [B,{cin},300,300], got {tuple(x.shape)}')
        bn_cfg = tuple((m.momentum, m.eps) for m in self._bn_list)
        # a forward no backward will follow (torch.no_grad(): evaluation, the fwd + loss metric) takes a plan whose pooled trunk layers
        # never write their full-resolution raw maps (GSSD_CONV_POOL2, _Plan._conv_bn); net.pooled_raw = False keeps one plan form
        nograd = (not need_backward) and bool(getattr(net, 'pooled_raw', True))
        key = (B, bool(training), x.device.index, hash(bn_cfg), bool(want_maps), getattr(net, 'compute_dtype', 'f32'), nograd)
        plans = self._plans.setdefault(key, [])
        plan = next((pl for pl in plans if not pl.busy), None)
        if plan is None:
            if len(plans) >= self.MAX_PLANS_IN_FLIGHT:
                raise _lib.GssdError(f'{len(plans)} forwards of batch {B} are still waiting for their backward; free their '
                                     f'outputs (or call backward) before running more')
            plan = self._build(B, bool(training), x.device, bool(want_maps), nograd)
            plans.append(plan)
            vers, ptrs = self._state()
        self._ptrs = ptrs
        if vers != self._versions:
            self._refresh_packed(x.device)
            self._versions = vers
        self._last_plan = plan
        loc, conf = plan.run(x, events)
        return loc, conf, plan

    def forward(self, x, training, events=None, want_maps=False):
        loc, conf, _ = self.forward_plan(x, training, events, want_maps, need_backward=False)
        return loc, conf

    # ------------------------------------------------------------------------------------------
    def _pack(self, name, build):
        """Register a packed weight: ``build(out_or_None) -> tensor`` fills/refreshes it in place.  The cache is per storage mode
        (a module switched between fp32 and bf16 keeps both sets of packed weights)."""
        if getattr(self.net, 'compute_dtype', 'f32') == 'bf16':
            name = name + '@bf16'
        if name not in self._packed:
            t = build(None)
            self._packed[name] = t
            self._pack_jobs.append(lambda: build(self._packed[name]))
            self._pack_table = None
        return self._packed[name]

    def _refresh_packed(self, dev):
        """Re-derive every packed weight from the (changed) parameters.  Jobs made only of ops.pack_weight / ops.copy_into from the
        parameters' own storage are recorded ONCE into a device table and from then on refreshed by a single launch
        (gssd_pack_conv_weights_batched: ~130 launches of 5-8 us after every optimizer step otherwise); the rest -- Winograd
        transforms of packed weights, the DCN layout, bf16 rounding -- run after it, in registration order."""
        if self._pack_table is None:
            rec, rest = ops.PackRecorder(), []
            for job in self._pack_jobs:
                n0, k0 = len(rec.items), len(rec.keep)
                ops.recorder = rec
                try:
                    job()
                except ops.PackRecorder.Unstable:
                    del rec.items[n0:], rec.keep[k0:]
                    rest.append(job)
                finally:
                    ops.recorder = None
                if len(rec.items) == n0 and job not in rest:
                    rest.append(job)               # recorded nothing: a job of other launches only
            table = ops.pack_table(rec, dev) if rec.items else None
            self._pack_table = (table, len(rec.items), rec.keep, rest)
        table, n, _, rest = self._pack_table
        if n:
            ops.run_pack_table(table, n)
        for job in rest:
            job()

    def _build(self, B, training, dev, want_maps=False, nograd=False):
        drain_sk_releases()
        if getattr(self.net, 'vanilla', False):
            return _PlanVanilla(self, B, training, dev)
        return _Plan(self, B, training, dev, want_maps, nograd)


_SK_PENDING = []          # (device index, output pointer) of dropped plans whose stream-K regions await their release


def drain_sk_releases():
    """Free the stream-K regions queued by _Plan.__del__: only when no stream of this thread is capturing, each on its own device."""
    if not _SK_PENDING or torch.cuda.is_current_stream_capturing():
        return 0
    n = 0
    while _SK_PENDING:
        di, ptr = _SK_PENDING.pop()
        with torch.cuda.device(di):
            n += max(0, int(lib.gssd_dcn_streamk_release(ptr)))
    return n


class _RecList(list):
    """The forward graph records; append() notes which branch (stream id of the launch plan, 0 = trunk) the record belongs to."""

    def __init__(self, plan):
        super().__init__()
        self._plan = plan

    def append(self, item):
        item[1].setdefault('sid', getattr(self._plan, '_sid', 0))
        super().append(item)


class _PlanBase:
    """State shared by the grouped and the vanilla launch plans."""
    generation = 0        # bumped by every run(): a backward checks it ran against the forward that produced it
    busy = False          # a grad-enabled forward awaits its backward (gssd/autograd.py::_Lease)
    _bwd = None

    def backward_plan(self):
        if getattr(self, 'nograd', False):
            raise _lib.GssdError('this forward plan was built for a forward without backward (its pooled layers kept no raw maps)')
        if self._bwd is None:
            from .backward import BackwardPlan, Bf16Shadow
            # bf16 storage mode: the fp32 backward plan over fp32 copies of the stored bf16 activations (mixed precision)
            self._bwd = BackwardPlan(Bf16Shadow(self) if getattr(self, 'bf16', False) else self)
        return self._bwd


class _Plan(_PlanBase, PlanGraphMixin, PlanOpsMixin, PlanExecMixin):
    """The launch plan of one (batch size, mode) of the detector.  The graph walk lives in plan_graph.PlanGraphMixin, the per-op emitters in
    plan_ops.PlanOpsMixin, eager / hipGraph execution in plan_exec.PlanExecMixin; this class keeps construction and the buffer / step helpers."""

    def __del__(self):
        # csrc/dcn_fused.hip's stream-K regions belong to this plan's deformable-conv outputs (GSSD_DCN_X6=0 only).  The release frees device
        # memory (it synchronises) and the collector may drop a plan at any time -- inside another plan's open hipGraph capture, or with another
        # device current -- so the pointers are only QUEUED here, with their device; drain_sk_releases() frees them at the next plan build
        # that finds no capture open.
        try:
            outs = self.__dict__.get('_sk_outs', ())
            if outs:
                _SK_PENDING.extend((self.dev.index if self.dev.index is not None else torch.cuda.current_device(), ptr) for ptr in outs)
        except Exception:        # noqa: BLE001 -- interpreter teardown
            pass

    def __init__(self, eng, B, training, dev, want_maps=False, nograd=False):
        self.eng, self.B, self.training, self.dev = eng, B, training, dev
        self.nograd = nograd                       # no backward will read this plan's activations
        self.want_maps = want_maps                 # visualize=True: also materialise the attention maps
        # BASELINE.json configs[4]: bf16 NHWC activations + bf16 packed weights + bf16 MFMA, fp32 accumulation / BatchNorm
        # statistics / softmax / offsets / loc + conf / loss / NMS (net.compute_dtype = 'bf16'; parameters stay fp32 masters)
        self.bf16 = getattr(eng.net, 'compute_dtype', 'f32') == 'bf16'
        self.adt = torch.bfloat16 if self.bf16 else torch.float32
        self.conv_fn = lib.gssd_conv2d_nhwc_bf16 if self.bf16 else lib.gssd_conv2d_nhwc_f32
        # channels per group of the packed input: 12 / groups_vgg real ones (3 per CT phase at the default 4 groups), zero-padded to whole
        # 16-byte pieces (fp32: 4 channels, bf16: 8)
        self.cpad = ops.round_up(12 // eng.net.groups_vgg, 8 if self.bf16 else 4)
        net = eng.net
        self.steps = []
        self.bufs = []
        self.head_descs = []
        self.rec = _RecList(self)   # forward graph records, consumed by gssd/backward.py (each tagged with its branch stream id)
        self.P = 8732
        self.nc = net.num_classes
        g = net.groups_vgg
        f32 = torch.float32

        def buf(*shape):
            t = torch.empty(*shape, device=dev, dtype=f32)
            self.bufs.append(t)
            return t
        abuf = self._abuf

        # ---- batch-stat arena ------------------------------------------------------------------
        bn_mods = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        seen, uniq = set(), []
        for m in bn_mods:
            if id(m) not in seen:
                seen.add(id(m))
                uniq.append(m)
        # The trunk's batch sums are kept in R replicas (include/gssd_hip.h: gssd_conv_desc::stats_rep): device-scope fp64 atomics on one
        # cache line are served serially, and the persistent trunk kernels flush every workgroup's sums at the END of the launch (conv2_1
        # in bf16: 131 k atomics on 16 lines = 60 of its 140 us, profiles/r04_thin_knockout.txt).  R * C <= 2048 (at most 32): every
        # trunk layer spreads its sums over 256 lines; the consumers add the replicas up in a fixed order.
        # bf16 storage mode only: the fp32 trunk kernels are compute-bound, their workgroups finish spread out and the tail is not
        # there to remove (scripts/thin_f32_probe.py: conv1_1 248 us with one array, 260 with 32 replicas, 224 without batch sums)
        trunk_bn = ({id(m) for m in net.vgg if isinstance(m, torch.nn.BatchNorm2d)}
                    if (self.bf16 and os.environ.get('GSSD_STATS_REP', '1') != '0') else set())
        self.stat_rep = {id(m): (max(1, min(32, 2048 // m.num_features)) if id(m) in trunk_bn else 1) for m in uniq}
        total = sum(2 * m.num_features * self.stat_rep[id(m)] for m in uniq)
        self.stats = torch.zeros(max(total, 2), device=dev, dtype=torch.float64)
        off = 0
        self.stat_of = {}
        for m in uniq:
            n = 2 * m.num_features * self.stat_rep[id(m)]
            self.stat_of[id(m)] = self.stats[off:off + n]
            off += n
        self.nbt = [m.num_batches_tracked for m in uniq]

        self._setup_spectral_norm([(n, getattr(net, n)) for n in ('self_attn_base_list', 'self_attn_list')
                                   if getattr(net, n, None) is not None])

        # ---- input pack ------------------------------------------------------------------------------
        self.x_in = None   # set per run
        x16 = abuf(B, 300, 300, self.cpad * g)
        self._pack_step = len(self.steps)
        if self.bf16:
            self._add(lib.gssd_pack_input_nhwc_bf16, [0, x16.data_ptr(), B, 12, 300, 300, g])
        else:
            self._add(lib.gssd_pack_input_nhwc, [0, x16.data_ptr(), B, 12, 300, 300, g, self.cpad])

        if not net.batch_norm:
            if self.bf16:
                raise _lib.GssdError('bf16 storage mode is built for the batch_norm=True graph (BASELINE.json configs[4])')
            self._build_plain_graph(x16)
        else:
            self._build_bn_graph(x16)
        assert len(self.head_descs) == 6
        self._finish_heads()
        self._place_sn_step()
        self._place_branch0()


    # ------------------------------------------------------------------------------------------------
    def _add(self, fn, args, keep=None, tag=None):
        d = None
        if tag is None and fn in (lib.gssd_conv2d_nhwc_f32, lib.gssd_conv2d_nhwc_bf16):
            d = keep[0] if isinstance(keep, tuple) else keep
            tag = conv_tag(d, 3 if (d.cin_g in (4, 8) and d.groups == 4 and d.H == 300) else None,
                           bf16=fn is lib.gssd_conv2d_nhwc_bf16)
        if tag is not None:
            tag = Tag(tag)
            tag.layer = getattr(self, '_layer', None)
            tag.desc = d
        self.steps.append(_Step(fn, args, keep, tag, getattr(self, '_sid', 0), self.__dict__.pop('_pending_wait', None)))

    def _abuf(self, *shape):
        """Activation buffer in the plan's storage type (fp32, or bf16 in configs[4] mode)."""
        t = torch.empty(*shape, device=self.dev, dtype=getattr(self, 'adt', torch.float32))
        self.bufs.append(t)
        return t

    def _buf(self, *shape):
        t = torch.empty(*shape, device=self.dev, dtype=torch.float32)
        self.bufs.append(t)
        return t

    def _abuf_tail(self, tail, *shape):
        """Activation buffer with ``tail`` more elements stored right behind it (one allocation): a raw conv output whose deferred
        BatchNorm's padding vector sits at ``map + numel`` -- the fp32 Winograd kernel then fetches the padding value of an out-of-image
        patch position through the load ADDRESS (32-bit offset from the map) instead of selecting it per loaded element
        (include/gssd_hip.h: in_pad).  Returns (map, tail view)."""
        if getattr(self, 'bf16', False):
            return self._abuf(*shape), self._abuf(tail)          # (only the fp32 Winograd kernel uses the layout)
        n = 1
        for v in shape:
            n *= v
        flat = torch.empty(n + tail, device=self.dev, dtype=getattr(self, 'adt', torch.float32))
        self.bufs.append(flat)
        return flat[:n].view(*shape), flat[n:]


class _PlanVanilla(_Plan):
    """models/ssd.py:48-108: dense VGG-SSD300 without BatchNorm (BASELINE.json configs[0]).  conv + ReLU is one launch
    (ReLU in the conv epilogue); pools are the identity-affine pool pass."""

    def __init__(self, eng, B, training, dev):   # noqa: super().__init__ builds the grouped graph; not called on purpose
        self.eng, self.B, self.training, self.dev = eng, B, training, dev
        net = eng.net
        self.steps, self.bufs, self.head_descs, self.rec = [], [], [], _RecList(self)
        self.P, self.nc = 8732, net.num_classes
        self.stats = torch.zeros(2, device=dev, dtype=torch.float64)
        self.nbt = []
        x4 = self._buf(B, 300, 300, 4)
        self._pack_step = 0
        self._add(lib.gssd_pack_input_nhwc, [0, x4.data_ptr(), B, 3, 300, 300, 1, 4])
        cur, H, Cc = x4, 300, 4
        sources = []
        mods = list(net.vgg)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, torch.nn.Conv2d):
                cur, H, Cc = self._conv_relu(f'vgg.{i}', m, cur, H, Cc)
                i += 2                                   # conv + ReLU
                if i - 2 == 21:                          # conv4_3 (+ReLU at 22): L2Norm source (ssd.py:73-77)
                    s = self._buf(B, H, H, Cc)
                    self._add(lib.gssd_l2norm_f32, (cur.data_ptr(), net.L2Norm.weight.data_ptr(), s.data_ptr(), B * H * H, Cc,
                                                    float(net.L2Norm.eps)))
                    self.rec.append(('l2norm', dict(x_in=cur, out=s, H=H, C=Cc, mod=net.L2Norm)))
                    sources.append((s, H, Cc))
            else:                                        # MaxPool2d
                k, st, pd = m.kernel_size, m.stride, m.padding
                Hp = ops.pool_out_size(H, k, st, pd, m.ceil_mode)
                out = self._buf(B, Hp, Hp, Cc)
                self._add(lib.gssd_bn_relu_pool_f32, (cur.data_ptr(), out.data_ptr(), B, H, H, Cc, Hp, Hp, k, st, pd, 0, 1.0, 0, 0,
                                                      0, 0, 0.1, 1e-5, 0, 0, 0))
                self.rec.append(('pool', dict(x_in=cur, out=out, H=H, C=Cc, k=k, s=st, p=pd, Hp=Hp)))
                cur, H = out, Hp
                i += 1
        sources.append((cur, H, Cc))                     # conv7
        for k, m in enumerate(net.extras):
            cur, H, Cc = self._conv_relu(f'extras.{k}', m, cur, H, Cc)
            if k % 2 == 1:
                sources.append((cur, H, Cc))
        self.sources = sources
        off = 0
        f32 = torch.float32
        for i, (s, Hs, Cs) in enumerate(sources):
            A = MBOX[i]
            nloc, nconf = A * 4, A * self.nc
            lw, cw = net.loc[i], net.conf[i]
            _, K = ops.packed_k(Cs, 3, 3)

            def build_w(out, lw=lw, cw=cw, nloc=nloc, nconf=nconf, K=K):
                if out is None:
                    out = torch.empty(nloc + nconf, K, device=dev, dtype=f32)
                ops.pack_weight(lw.weight, out, 0)
                ops.pack_weight(cw.weight, out, nloc)
                return out

            def build_b(out, lw=lw, cw=cw, nloc=nloc):
                if out is None:
                    out = torch.empty(nloc + cw.bias.numel(), device=dev, dtype=f32)
                ops.copy_into(out[:nloc], lw.bias)
                ops.copy_into(out[nloc:], cw.bias)
                return out
            wp = eng._pack(f'heads.{i}.w', build_w)
            bp = eng._pack(f'heads.{i}.b', build_b)
            d, _, _ = ops.make_conv_desc(s, wp, None, B=B, H=Hs, W=Hs, in_stride=Cs, cin_g=Cs, Cout=nloc + nconf, k=3,
                                         pad=1, bias=bp, out_mode=_lib.OUT_HEADS, out_b=None, split_n=nloc,
                                         out_batch_stride=self.P * 4, outb_batch_stride=self.P * self.nc,
                                         out_off=off * 4, outb_off=off * self.nc,
                                         split_k=ops.auto_split_k(B * Hs * Hs, nloc + nconf, 1, K))
            self.head_descs.append(d)
            self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d),), keep=d)
            self.rec.append(('head', dict(i=i, src=s, H=Hs, C=Cs, A=A, off=off, loc=lw, conf=cw, K=K)))
            off += Hs * Hs * A
        assert off == self.P, off
        self._finish_heads()

    def _conv_relu(self, name, conv, x, H, Cin):
        B = self.B
        k, s, p, dl = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
        Cout = conv.out_channels

        def build(out, conv=conv, Cin=Cin):
            w = conv.weight
            if w.shape[1] != Cin:                        # conv1_1: 3 input channels stored in 4 (zero pad)
                w = torch.nn.functional.pad(w.detach(), (0, 0, 0, 0, 0, Cin - w.shape[1]))
            return ops.pack_weight(w, out)
        wp = self.eng._pack(name + '.w', build)
        Ho = (H + 2 * p - dl * (k - 1) - 1) // s + 1
        out = self._buf(B, Ho, Ho, Cout)
        d, _, _ = ops.make_conv_desc(x, wp, out, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin, Cout=Cout, k=k, stride=s, pad=p,
                                     dil=dl, bias=conv.bias.detach(), relu=True)
        self._add(lib.gssd_conv2d_nhwc_f32, (C.byref(d),), keep=d)
        self.rec.append(('convrelu', dict(name=name, conv=conv, x_in=x, out=out, H=H, Cin=Cin, Ho=Ho, Cout=Cout, desc=d, k=k,
                                          stride=s, pad=p, dil=dl)))
        return out, Ho, Cout


class SelfAttnOp(_Plan):
    """ONE Self_Attn block (layers/self_attn.py:46-89) as its own launch plan -- the op-level entry the parity tests use:
    x NHWC [B,H,H,C] -> (x + sigma*o, sigma*o, attention map [B,N,N]).  ``training`` runs the spectral-norm power iteration
    first (and mutates weight_u / weight_v in place), like the reference's forward pre-hook."""

    def __init__(self, sa, B, H, training, dev):   # noqa: a stand-alone block; _Plan.__init__ builds the whole network
        holder = torch.nn.Module()
        holder.self_attn_list = torch.nn.ModuleList([sa])
        self.eng = GssdEngine(holder)
        self.B, self.training, self.dev = B, bool(training), dev
        self.bf16, self.adt, self.conv_fn = False, torch.float32, lib.gssd_conv2d_nhwc_f32
        self.steps, self.bufs, self.rec, self.head_descs = [], [], _RecList(self), []
        Cc = sa.in_channels
        self.H, self.C = H, Cc
        self.x = self._buf(B, H, H, Cc)
        self._setup_spectral_norm([('self_attn_list', holder.self_attn_list)])
        self.out, self.out2 = self._self_attn('self_attn_list', 0, self.x, H, Cc, need_out2=True, want_map=True)

    def run(self, x_nhwc):
        self.generation += 1
        self.x.copy_(x_nhwc)
        stream = torch.cuda.current_stream().cuda_stream
        for st in self.steps:
            rc = st.fn(*st.args, stream)
            if rc != 0:
                _lib.check(rc)
        S, N, Np = self.attn_maps[('self_attn_list', 0)]
        return self.out, self.out2, S[:, :, :N]
