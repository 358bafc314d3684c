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

VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'C', 512, 512, 512, 'M', 512, 512, 512]
EXTRAS_CFG = [256, 'S', 512, 128, 'S', 256, 128, 256, 128, 256]
MBOX = [4, 6, 6, 6, 4, 4]
SRC_HW = [38, 19, 10, 5, 3, 1]
HEAD_OFF = [sum(h * h * a for h, a in zip(SRC_HW[:i], MBOX[:i])) for i in range(6)]      # first prior of source i
FUSE_NAMES = ['11', '21', '31', '41', '51', '61']


# Winograd F(2x2,3x3) for the compute-bound 3x3 trunk layers (csrc/conv_wino.hip); GSSD_NO_WINOGRAD=1 keeps the direct
# implicit GEMM everywhere (ablation / cross-check).
USE_CONV_X6 = os.environ.get('GSSD_CONV_X6', '1') != '0'      # csrc/conv_x6.hip for the launches ops.x6_wanted names (fp32 mode)
USE_WINOGRAD = os.environ.get('GSSD_NO_WINOGRAD', '0') != '1'
# fp32 mode: the deformable conv on the bf16 matrix cores with three-plane (fp32-equivalent) operands, csrc/dcn_x6.hip;
# GSSD_DCN_X6=0: the fp32-MFMA kernel csrc/dcn_fused.hip
DCN_X6 = os.environ.get('GSSD_DCN_X6', '1') != '0'
# GSSD_NO_GRAPH=1 keeps every forward an eager list of launches (debugging / ablation)
USE_GRAPH = os.environ.get('GSSD_NO_GRAPH', '0') != '1'
# GSSD_FLASH_X6=0: the fp32-MFMA attention core (csrc/flash_attn.hip) keeps every launch of the fp32 mode (ablation / A-B)
USE_FLASH_X6 = os.environ.get('GSSD_FLASH_X6', '1') != '0'
# GSSD_NO_BRANCH_STREAMS=1 captures the plan as one serial chain (ablation)
USE_BRANCH_STREAMS = os.environ.get('GSSD_NO_BRANCH_STREAMS', '0') != '1'
SN_STREAM = 9               # stream id of the spectral-norm launch inside a captured graph
ALL_STREAMS = -1            # _Step.wait value: join every forked stream before this step

class Tag(tuple):
    """(kernel instance, algorithmic FLOPs, algorithmic bytes) of one launch; ``layer`` names the module it belongs to
    ('vgg.0' = conv1_1 ... 'vgg.40' = conv5_3) so bench.py can sum the trunk's launches -- convs AND their BatchNorm passes."""
    layer = None


class _Step:
    __slots__ = ('fn', 'args', 'keep', 'tag', 'sid', 'wait')

    def __init__(self, fn, args, keep=None, tag=None, sid=0, wait=None):
        # sid: stream id inside a captured graph (0 = trunk);  wait: a stream id whose work this step consumes (joined before it)
        self.fn, self.args, self.keep, self.tag, self.sid, self.wait = fn, args, keep, tag, sid, wait


def conv_tag(d, real_cin_g=None, bf16=False):
    """(kernel instance, algorithmic FLOPs, algorithmic bytes) of one gssd_conv2d launch; the instance name
    mirrors the tile selection in csrc/conv_igemm.hip so it can be matched against rocprofv3's kernel names."""
    cout_g = d.Cout // d.groups
    inst = '128x128' if cout_g > 64 else '128x64' if cout_g > 32 else '128x32' if cout_g > 16 else '128x16'
    if cout_g > 64:       # same wave-quantisation rule as gssd_conv2d_nhwc_f32
        mt = -(-(d.Ho * d.Wo * (1 if d.m_per_image else d.B)) // 128)
        z = d.B if d.m_per_image else d.split_k
        b128 = mt * d.groups * (-(-cout_g // 128)) * z
        b64 = mt * d.groups * (-(-cout_g // 64)) * z
        e128 = b128 / (-(-b128 // 512) * 512)
        e64 = 0.94 * b64 / (-(-b64 // 768) * 768)
        if e64 > e128 or d.K <= 256:
            inst = '128x64'
    # small maps: 32- / 64-row tiles with a three-stage K loop (csrc/conv_igemm.hip, csrc/conv_bf16.hip: the same host rule)
    Ms, Mtot = d.Ho * d.Wo * (1 if d.m_per_image else d.B), d.Ho * d.Wo * d.B
    if (cout_g > 32 and d.split_k == 1 and Mtot <= 4096 and os.environ.get('GSSD_NO_SMALL_TILES') is None
            and not (d.out_mode == _lib.OUT_SPLIT_T and d.split_n % 64 != 0)):
        inst = '32x64' if (Mtot <= 512 or (d.m_per_image and Ms <= 128)) else '64x64'
    name = ('conv_bf16<' if bf16 else 'conv_igemm<') + inst + '>'
    if not bf16 and d.wgt_x6 and lib.gssd_conv_x6_takes(C.byref(d)) == 1:
        M6 = d.B * d.Ho * d.Wo
        flops = 2.0 * M6 * d.Cout * d.KH * d.KW * d.cin_g
        return (f'conv_x6<{ops.x6_tile(cout_g, d.groups, M6)}>', flops, 4.0 * (d.B * d.H * d.W * d.cin_g * d.groups + M6 * d.Cout + d.Cout * d.KH * d.KW * d.cin_g))
    if bf16:
        if (d.groups == 4 and d.KH == 3 and d.stride == 1 and d.pad == 1 and d.dil == 1 and d.H * d.W >= 75 * 75
                and (d.cin_g, cout_g) in ((8, 16), (16, 16), (16, 32), (32, 32)) and not d.m_per_image and d.split_k == 1
                and d.flags in (0, _lib.CONV_POOL2)):
            name = f'conv_thin_bf16<{d.cin_g},{cout_g}>' + ('/pool2' if d.flags & _lib.CONV_POOL2 else '')   # gssd_try_conv_thin_bf16
    elif (d.groups == 4 and d.KH == 3 and d.stride == 1 and d.pad == 1 and d.dil == 1 and d.H * d.W >= 75 * 75
            and (d.cin_g, cout_g) in ((4, 16), (16, 16), (16, 32)) and not d.m_per_image and d.split_k == 1):
        name = f'conv_thin<{d.cin_g},{cout_g}>'          # gssd_try_conv_thin (csrc/conv_thin.hip)
        if d.wgt_wino and (d.cin_g, cout_g) == (16, 32) and os.environ.get('GSSD_CONV21_WINO', '1') != '0':
            name = 'conv_wino<32>'                       # conv2_1 with Winograd weights: handed on to gssd_try_conv_wino
        if d.wgt_wino and (d.cin_g, cout_g) == (16, 16) and not d.resid:
            name = 'conv_thin_wino<16,16>'               # gssd_try_conv_thin_wino (csrc/conv_thin_wino.hip)
    elif (d.wgt_wino and ops.winograd_eligible(d.KH, d.stride, d.pad, d.dil, d.cin_g, cout_g, d.groups) and not d.m_per_image
          and d.split_k <= 1 and not d.relu):
        name = f'conv_wino<{64 if (cout_g % 64 == 0 or (cout_g % 32 != 0 and cout_g > 32)) else 32}>'   # gssd_try_conv_wino
    if name.startswith('conv_wino<'):
        # one name per kernel SYMBOL, as rocprofv3 --stats groups them (template <tile, fused input transform, ..., pooled epilogue>)
        name += ('' if d.in_scale else '/plain') + ('/pool2' if d.flags & _lib.CONV_POOL2 else '')
    if bf16 and name.startswith('conv_bf16'):
        bm = lib.gssd_conv_flat_bf16_takes(C.byref(d))      # csrc/conv_flat_bf16.hip: the library's own host rule
        if bm:
            name = f'conv_flat_bf16<{d.cin_g},{min(cout_g, 128) if cout_g % 128 == 0 else 64},{bm}>'
    if not bf16 and name.startswith('conv_igemm') and lib.gssd_gemm_slot_takes(C.byref(d)) == 1:
        name = 'gemm_slot<128x128>'                      # gssd_try_gemm_slot (csrc/gemm_slot.hip): the library's own host rule
    M = d.B * d.Ho * d.Wo
    cin_g = real_cin_g if real_cin_g is not None else d.cin_g
    flops = 2.0 * M * d.Cout * d.KH * d.KW * cin_g
    esz = 2.0 if bf16 else 4.0
    out_elems = M * d.Cout // 4 if (d.flags & _lib.CONV_POOL2) else M * d.Cout       # pooled raw output: a quarter of the map
    byts = esz * (d.B * d.H * d.W * cin_g * d.groups + out_elems + d.Cout * d.KH * d.KW * cin_g)
    return (name, flops, byts)


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
        if getattr(self.net, 'vanilla', False):
            return _PlanVanilla(self, B, training, dev)
        return _Plan(self, B, training, dev, want_maps, nograd)


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


class _Plan(_PlanBase):
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
        d, _, _ = ops.make_conv_desc(s, wp, None, B=B, H=Hs, W=Hs, in_stride=Cs, cin_g=Cs, Cout=nloc + nconf, k=3,
                                     pad=1, bias=bp, out_mode=_lib.OUT_HEADS, out_b=None, split_n=nloc,
                                     out_batch_stride=self.P * 4, outb_batch_stride=self.P * self.nc,
                                     out_off=off * 4, outb_off=off * self.nc,
                                     split_k=(ops.auto_split_k(B * Hs * Hs, nloc + nconf, 1, K, target_blocks=256, max_split=8)
                                              if self.bf16 else ops.auto_split_k(B * Hs * Hs, nloc + nconf, 1, K)),
                                     flags=_lib.CONV_OUT_F32)
        self.head_descs.append(d)
        self._add(self.conv_fn, (C.byref(d),), keep=d)
        self.rec.append(('head', dict(i=i, src=s, H=Hs, C=Cs, A=A, off=off, loc=lw, conf=cw, K=K)))

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
        if not self.bf16 and USE_CONV_X6 and ops.x6_wanted(k, cin_g, Cout // groups, groups, B * Ho * Ho, winograd=U is not None):
            def build_x6(out, key=name + '.w', groups=groups, cin_g=cin_g, taps=k * k, bn=ops.x6_tile(Cout // groups, groups, B * Ho * Ho)):
                return ops.x6_weight(self.eng._packed[key], groups, cin_g, taps, bn, out)
            X6 = self.eng._pack(name + '.x6', build_x6)       # (after '.w' as well)
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
                                         in_pad=in_xf[2] if in_xf else None, flags=_lib.CONV_POOL2, pool_sign=bn.weight.detach(), stats_rep=srep)
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
                                     in_pad=in_xf[2] if in_xf else None, stats_rep=srep)
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
            x6_tpg = eng._pack(name + '.tpg.x6', build_x6p)
            gT.zero_()                         # csrc/conv_x6.hip never writes the row tails [N, Np) of g^T (conv_igemm zero-fills them)
        d1, _, _ = mk(x, w_tpg, tp, B=B, H=H, W=H, in_stride=Cc, cin_g=Cc, Cout=C4 + C2, bias=b_tpg, alpha=a_tpg, wgt_x6=x6_tpg,
                      out_mode=_lib.OUT_SPLIT_T, out_b=gT, split_n=C4, out_stride=C4, out_b_stride=Np, m_per_image=not flat,
                      in_batch_stride=N * Cc, out_batch_stride=N * C4, outb_batch_stride=C2 * Np,
                      flags=_lib.CONV_OUT_F32 | (_lib.CONV_OUTB_BF16_PERM32 if self.bf16 else 0))
        x6_o = None
        if not self.bf16 and USE_CONV_X6 and ops.x6_wanted(1, C2, Cc, 1, B * N):
            def build_x6o(out, bn=ops.x6_tile(Cc, 1, B * N)):
                return ops.x6_weight(sa.snconv1x1_attn.weight_orig.detach().view(Cc, C2), 1, C2, 1, bn, out)
            x6_o = eng._pack(name + '.o.x6', build_x6o)
        d5, _, _ = mk(ag, w_o, out, B=B, H=H, W=H, in_stride=C2, cin_g=C2, Cout=Cc, bias=sa.snconv1x1_attn.bias.detach(),
                      alpha=a_o, gate=sa.sigma.detach(), resid=x, out2=out2, wgt_x6=x6_o)
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
        d1, _, _ = ops.make_conv_desc(x, w_om, om, B=B, H=H, W=H, in_stride=Cin, cin_g=Cin, Cout=27 * dg, k=3, pad=1, out_stride=OMC,
                                      bias=m.conv_offset_mask.bias.detach(), wgt_wino=u_om, flags=_lib.CONV_OUT_F32)
        self._add(self.conv_fn, (C.byref(d1),), keep=d1)
        M = B * H * H
        esz = 2.0 if self.bf16 else 4.0
        self._add(lib.gssd_dcn_forward_bf16 if self.bf16 else lib.gssd_dcn_forward_x6 if DCN_X6 else lib.gssd_dcn_forward_f32,
                  (x.data_ptr(), om.data_ptr(), w_main.data_ptr(), m.bias.data_ptr(), out.data_ptr(), B, H, H, Cin, dg, OMC, Cout),
                  keep=w_main, tag=('dcn_bf16<128x256>' if self.bf16 else 'dcn_x6<128x256>' if DCN_X6 else 'dcn_fused<128x256>', 2.0 * M * Cout * 9 * Cin,
                                    esz * (M * (Cin + Cout) + Cout * 9 * Cin) + 4.0 * M * 27 * dg))
        self.offsets = getattr(self, 'offsets', [])
        self.offsets.append((om, H, dg))
        self.rec.append(('dcn', dict(mod=m, x_in=x, out=out, H=H, Cin=Cin, Cout=Cout, om=om, d_om=d1, dg=dg, li=li, omc=OMC)))
        return out, Cout

    # ------------------------------------------------------------------------------------------------
    def run(self, x, events=None):
        """``events``: optional list; when given, every tagged launch (or only the kernel instances named in ``events.only``) is
        bracketed by a pair of HIP events recorded on the launch stream and (tag, start, end) is appended (bench.py's live
        roofline measurement).

        From its third run on a plan replays itself from hipGraphs: the ~200-270 launches of a step are static (preallocated
        buffers, descriptors by value), so the host side of a step shrinks from a ctypes call per kernel (~2.5 ms) to a few graph
        launches.  Launches that must be bracketed by events stay eager and split the plan into graph segments around them."""
        self.generation += 1
        x = x.contiguous().float()
        only = getattr(events, 'only', None) if events is not None else None
        self._runs = getattr(self, '_runs', 0) + 1
        if USE_GRAPH and (events is None or only) and self._runs > 2:
            return self._run_graphs(x, events, only)
        return self._run_eager(x, events, only)

    def _launch(self, st, stream):
        rc = st.fn(*st.args, stream)
        if rc != 0:
            _lib.check(rc)

    def _run_eager(self, x, events, only):
        B, dev = self.B, self.dev
        # every element is written: the head convs store per-slice partial sums, _finish_heads' reduce launches add them in order
        loc = torch.empty(B, self.P, 4, device=dev, dtype=torch.float32)
        conf = torch.empty(B, self.P, self.nc, device=dev, dtype=torch.float32)
        self._set_outputs(loc, conf)
        self.steps[self._pack_step].args[0] = x.data_ptr()
        if self.training:
            self.stats.zero_()
        stream = torch.cuda.current_stream().cuda_stream
        for st in self.steps:
            if events is not None and st.tag is not None and (only is None or st.tag[0] in only):
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
        return loc, conf

    def _run_graphs(self, x, events, only):
        key = tuple(sorted(only)) if only else None
        cache = self.__dict__.setdefault('_graphs', {})
        if key not in cache:
            cache[key] = self._capture(x, only)
        segs = cache[key]
        self._gx.copy_(x)
        stream = torch.cuda.current_stream().cuda_stream
        for kind, obj in segs:
            if kind == 'graph':
                obj.replay()
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._launch(obj, stream)
                e1.record()
                events.append((obj.tag, e0, e1))
        return self._gloc.clone(), self._gconf.clone()

    def _side_stream(self, sid):
        pool = self.__dict__.setdefault('_side_streams', {})
        if sid not in pool:
            pool[sid] = torch.cuda.Stream(device=self.dev)
        return pool[sid]

    def _capture(self, x, only):
        """Capture the plan as hipGraph segments over static input / output buffers; the steps named in ``only`` stay eager."""
        B, dev = self.B, self.dev
        if getattr(self, '_gx', None) is None:
            self._gx = torch.empty_like(x)
            self._gloc = torch.zeros(B, self.P, 4, device=dev, dtype=torch.float32)
            self._gconf = torch.zeros(B, self.P, self.nc, device=dev, dtype=torch.float32)
        self._set_outputs(self._gloc, self._gconf)
        self.steps[self._pack_step].args[0] = self._gx.data_ptr()
        groups, cur = [], []
        for st in self.steps:
            if only and st.tag is not None and st.tag[0] in only:
                groups.append(('graph', cur))
                groups.append(('step', st))
                cur = []
            else:
                cur.append(st)
        groups.append(('graph', cur))
        torch.cuda.synchronize(dev)
        # the stream-K deformable conv keeps per-tile flags that every launch leaves at zero; a launch that was aborted would not:
        # start every captured plan from zeroed flags (include/gssd_hip.h: gssd_dcn_streamk_reset)
        _lib.check(lib.gssd_dcn_streamk_reset(torch.cuda.current_stream().cuda_stream))
        pool = torch.cuda.graph_pool_handle()
        segs, n_graph = [], sum(1 for k, _ in groups if k == 'graph')
        gi = 0
        for kind, obj in groups:
            if kind == 'step':
                segs.append(('step', obj))
                continue
            first, last = gi == 0, gi == n_graph - 1
            gi += 1
            if not obj and not first and not (last and self.training and self.nbt):
                continue
            g = torch.cuda.CUDAGraph()
            # (measured and rejected, round 4: capturing the trunk on a high-priority stream so that a branch's chip-filling launches
            # do not take CUs from the critical path's next kernel -- 12.21 -> 13.34 ms fp32, 3.95 -> 5.03 ms bf16)
            with torch.cuda.graph(g, pool=pool):
                if first:
                    if self.training:
                        self.stats.zero_()
                main = torch.cuda.current_stream()
                forked = {}
                for st in obj:
                    if st.sid == 0 or not USE_BRANCH_STREAMS:
                        if st.wait == ALL_STREAMS:
                            for side in forked.values():
                                main.wait_stream(side)
                        elif st.wait is not None and st.wait in forked:
                            main.wait_stream(forked[st.wait])
                        self._launch(st, main.cuda_stream)
                        continue
                    side = forked.get(st.sid)
                    if side is None:                       # fork: the branch starts behind everything the trunk has enqueued
                        side = self._side_stream(st.sid)
                        side.wait_stream(main)
                        forked[st.sid] = side
                    if st.wait is not None and st.wait in forked:
                        side.wait_stream(forked[st.wait])
                    self._launch(st, side.cuda_stream)
                for side in forked.values():               # join: a graph segment ends with every branch folded back
                    main.wait_stream(side)
                if last and self.training and self.nbt:
                    torch._foreach_add_(self.nbt, 1)
            segs.append(('graph', g))
        # the capture itself does not execute anything: the caller's replay is the run
        return segs


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
