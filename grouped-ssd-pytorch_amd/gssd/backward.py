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
The deformable conv (dcn.hip: col2im with atomics; the contraction's two gradients on the conv kernels) and the
self-attention blocks (csrc/sa_backward.hip) are HIP too: the plan is a list of C-ABI launches, no library GEMM, no ATen math.
"""
import ctypes as C

import os

import torch

from . import _lib, ops, planrun
from ._lib import lib
from .bwd_common import (BWD_STREAMS, USE_BWD_GRAPH, BWD_BF16, LEAF_SID, HOIST_FROM, N_LEAF, _leaf_fns)      # noqa: F401
from .bwd_ops import BackwardOpsMixin, _PaddedWeight, _pack_dgrad_from_packed, conv_weight_ptr      # noqa: F401
from .bwd_shadow import Bf16Shadow, _EngView      # noqa: F401


class BackwardPlan(BackwardOpsMixin):
    def __init__(self, plan):
        self.plan = plan
        self.B, self.dev = plan.B, plan.dev
        self.bf16_ops = BWD_BF16 and isinstance(plan, Bf16Shadow)      # NT GEMMs / data-gradient convs on the bf16 matrix cores
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
        self._offs = {id(p): (o, (p.numel() + 3) // 4 * 4) for p, o in zip(plan.eng.net.parameters(), offs)}
        self._last_write = {}        # id(param) -> index of the last step that writes its gradient
        self.segment_hook = None     # callable(k, flat[lo:hi]) invoked right after segment k's last writer was enqueued
        self.dloc = torch.empty(self.B, plan.P, 4, device=self.dev)
        self.dconf = torch.empty(self.B, plan.P, plan.nc, device=self.dev)
        net = plan.eng.net
        self.first_in = plan.rec[0][1]['x_in'].data_ptr()
        # The six branch blocks ([SA] -> fuse conv -> loc | conf head; block 0 also L2Norm) depend only on d(loc) / d(conf): their
        # backward is hoisted to the front and tagged with the branch's stream id, so the chains run beside each other and beside the
        # trunk's backward (on the 19 x 19 .. 1 x 1 maps a launch has 1 .. 100 workgroups); the trunk waits for branch k where it first
        # touches a gradient that branch writes.  (Any order of the contributions to a shared activation gradient is handled: whoever
        # comes second accumulates.)  Measured: blocks 1 .. 5 GSSD++ 56.2 -> 54.8 ms; block 0 as well GSSD 25.5 -> 23.3 ms (GSSD++
        # unchanged: its block 0 is the 1444-token attention, chip-filling launches).  The step list in build order is also a valid
        # single-stream order (run() uses it that way with GSSD_BWD_STREAMS=0; a gradient-segment hook joins the streams where it fires).
        self.step_sid = []           # stream id per step
        self.step_wait = {}          # step index -> [stream ids the step's stream waits for first]
        self._cur_sid = 0
        hoisted = sorted({r.get('sid', 0) for _, r in plan.rec if r.get('sid', 0) >= HOIST_FROM}, reverse=True) if BWD_STREAMS else []
        for sid in hoisted:
            self._cur_sid = sid
            for kind, r in reversed(plan.rec):
                if r.get('sid', 0) == sid:
                    self._one(kind, r)
        self._cur_sid = 0

        def inputs(kind, r):
            ts = [r['src']] if kind == 'head' else [r['a'], r['b']] if kind == 'slice_cat' else [r['x_in']]
            return {t.data_ptr() for t in ts}
        root = {}                    # branch -> the trunk activation it hangs off (input of its first forward record)
        for kind, r in plan.rec:
            if r.get('sid', 0) in hoisted and r['sid'] not in root:
                root[r['sid']] = inputs(kind, r)
        joined = set()

        def join(sid):
            if sid not in joined:
                joined.add(sid)
                self.step_wait.setdefault(len(self.steps), []).append(sid)
        for kind, r in reversed(plan.rec):
            sid = r.get('sid', 0)
            if sid in hoisted:
                join(sid)            # where the sequential order had this branch's steps
                continue
            for b, ptrs in root.items():
                if ptrs & inputs(kind, r):
                    join(b)          # a trunk layer reading the same activation accumulates into the gradient the branch writes
            self._one(kind, r)
        self.hoisted = hoisted
        self.param_order = [p for p in net.parameters()]

    def _one(self, kind, r):
        self._layer_no = getattr(self, '_layer_no', 0) + 1      # (the leaf launches of one layer stay on one leaf stream, in order)
        first_in = self.first_in
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
        elif kind == 'relupool':
            self._relupool(r)
        elif kind == 'plhead':
            self._plhead(r)
        elif kind == 'interp':
            self._interp(r)
        elif kind == 'plfinal':
            self._plfinal(r)
        else:
            raise _lib.GssdError(f'no HIP backward for {kind}')
        while len(self.step_sid) < len(self.steps):          # (handlers also append host-side steps directly)
            self.step_sid.append(self._cur_sid)

    # ------------------------------------------------------------------------------------------------
    def _add(self, fn, args, keep=None, leaf=None):
        self.steps.append((fn, args))
        # launches that only finish parameter gradients (weight gradients and their unpacking, bias column sums, spectral-norm and
        # sigma gradients, the DCN column matrix for its weight gradient) feed nothing downstream: in the trunk they go to the "leaf"
        # stream, off the d(activation) chain -- on the small maps the chain's launches no longer queue behind them, and the tails of
        # the big ones overlap
        leaf = BWD_STREAMS and self._cur_sid == 0 and (fn in _leaf_fns() if leaf is None else leaf)
        self.step_sid.append(LEAF_SID + self._layer_no % N_LEAF if leaf else self._cur_sid)
        if keep is not None:
            self.keep.append(keep)

    def _buf(self, *shape, dtype=torch.float32, zero_each_run=False):
        t = torch.empty(*shape, device=self.dev, dtype=dtype)
        self.keep.append(t)
        if zero_each_run:
            self.zero_list.append(t)
        return t

    def _pgrad(self, p):
        self._last_write[id(p)] = len(self.steps)        # (evaluated while the writing step's argument tuple is built)
        g = self.grads.get(id(p))
        if g is None:
            g = self._slices[id(p)]              # parameters the plan never writes keep no entry -> gradient None
            self.grads[id(p)] = g
        return g

    def _grad_of(self, t):
        if t.data_ptr() in self.__dict__.get('_g16', {}):
            raise _lib.GssdError('this gradient map exists in bf16 only (a thin trunk layer): its one reader is the BatchNorm backward')
        return self.gbuf.get(t.data_ptr())


    # ---- PixelLink++ tail (gssd/pixellink.py records; csrc/pixellink.hip) -------------------------------------------------------------
    PL_LD = 20          # channel stride of the gradient maps of the 18-channel score maps (whole 16-byte quads for the conv kernels)


    # ------------------------------------------------------------------------------------------------
    def segments(self, nseg=4):
        """Split the flat gradient buffer into ``nseg`` contiguous ranges of about equal size (parameter granularity) and find, for
        each, the step after which all of its gradients are final: [(lo, hi, ready_step)].  The backward walks the network in
        reverse, so the LAST range (heads, extras) is ready first -- its all-reduce can overlap the trunk's backward."""
        items = sorted(self._offs.items(), key=lambda kv: kv[1][0])          # (id, (offset, padded size))
        total = self.flat.numel()
        segs, lo, acc, ready = [], 0, 0, -1
        for i, (pid, (o, n)) in enumerate(items):
            acc += n
            ready = max(ready, self._last_write.get(pid, -1))
            if acc >= total * (len(segs) + 1) / nseg or i == len(items) - 1:
                segs.append((lo, o + n, ready))
                lo, ready = o + n, -1
        return segs

    def run(self, dloc, dconf):
        self.dloc.copy_(dloc)
        self.dconf.copy_(dconf)
        return self._execute()

    def _execute(self):
        """One backward over the static buffers.  From its third run on the plan replays itself from hipGraphs (round 4; VERDICT r3
        item 6): the ~500 launches of a training step's backward are static -- preallocated buffers, descriptors by value, the same
        stream forks every time -- so the host side shrinks from a ctypes call per kernel (27.6 ms per GSSD++ step, scripts/
        host_vs_gpu.py) to a few graph launches.  With a gradient-segment hook (data-parallel training: gssd/dist.py starts a range's
        all-reduce where its last writer has been enqueued) the step list is cut into one graph per segment and the hook runs
        between them, on the host, exactly where the eager executor called it.  Opt-in (GSSD_BWD_GRAPH=1): see USE_BWD_GRAPH."""
        hook = self.segment_hook
        fire = {}
        if hook is not None:
            if getattr(self, '_segs', None) is None:
                self._segs = self.segments()
            for k, (lo, hi, ready) in enumerate(self._segs):
                fire.setdefault(max(ready, 0), []).append(k)
        cuts = sorted(fire) + ([len(self.steps) - 1] if (len(self.steps) - 1) not in fire else [])
        ranges, lo = [], 0
        for c in cuts:                                       # [lo, hi] inclusive step ranges, a hook point (or the end) behind each
            ranges.append((lo, c))
            lo = c + 1
        self._nrun = getattr(self, '_nrun', 0) + 1
        use_graph = USE_BWD_GRAPH and self._nrun > 2 and not getattr(self, 'single_stream', False)
        if not use_graph:
            for i, (lo, hi) in enumerate(ranges):
                self._run_range(lo, hi, first=(i == 0), last=(i == len(ranges) - 1))
                self._fire(hook, fire, hi)
            return [self.grads.get(id(p)) for p in self.param_order]
        key = tuple(ranges)
        cache = self.__dict__.setdefault('_graphs', {})
        if key not in cache:
            torch.cuda.synchronize(self.dev)
            pool = torch.cuda.graph_pool_handle()
            gs = []
            for i, (lo, hi) in enumerate(ranges):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    self._run_range(lo, hi, first=(i == 0), last=(i == len(ranges) - 1))
                gs.append(g)
            cache[key] = gs                                  # (the capture executes nothing: the replay below is the run)
        for (lo, hi), g in zip(ranges, cache[key]):
            g.replay()
            self._fire(hook, fire, hi)
        return [self.grads.get(id(p)) for p in self.param_order]

    def _fire(self, hook, fire, si):
        if hook is not None and si in fire:
            for k in fire[si]:
                lo, hi, _ = self._segs[k]
                hook(k, self.flat[lo:hi])

    def _run_range(self, lo, hi, first, last):
        """Steps lo .. hi (inclusive) on the current stream and the plan's branch / leaf streams; every stream forked here is folded
        back into the current stream before returning (a range ends where a gradient segment is handed out, or at the end).

        The range is static, so its launches and stream forks / joins are recorded once (planrun.RecordSink) and replayed with one
        gssd_plan_run call per run of launches (round 5: ~500 ctypes calls + ~100 torch stream calls per training step before); the
        few steps that are torch code run on the host between the programs.  hipGraph capture and GSSD_NO_PLAN_RUN=1 execute the same
        control flow directly (planrun.EagerSink)."""
        if not planrun.USE_PLAN_RUN or torch.cuda.is_current_stream_capturing():
            return self._emit_range(planrun.EagerSink(), lo, hi, first, last)
        cache = self.__dict__.setdefault('_programs', {})
        key = (lo, hi, first, last, bool(getattr(self, 'single_stream', False)), len(self.zero_list))
        segs = cache.get(key)
        if segs is None:
            sink = planrun.RecordSink()
            self._emit_range(sink, lo, hi, first, last)
            segs = cache[key] = sink.finish()
        planrun.replay(segs)

    def _emit_range(self, sink, lo, hi, first, last):
        MAIN = planrun.MAIN
        if first:
            if self.zero_list:
                # one multi-tensor launch per dtype (a mixed fp32 / fp64 list takes _foreach_zero_'s slow path: ~180 fill launches a step)
                if getattr(self, '_zero_groups', None) is None or sum(len(g) for g in self._zero_groups) != len(self.zero_list):
                    by = {}
                    for t in self.zero_list:
                        by.setdefault(t.dtype, []).append(t)
                    self._zero_groups = list(by.values())
                groups = self._zero_groups
                sink.host(lambda: [torch._foreach_zero_(grp) for grp in groups])
            for fn, args in getattr(self.plan, 'pre', ()):      # bf16 forward plan: fp32 copies of what the forward stored (Bf16Shadow)
                self._emit_step(sink, fn, args, MAIN)
        if getattr(self, 'single_stream', False) or not (self.hoisted or any(x >= LEAF_SID for x in self.step_sid)):
            for si in range(lo, hi + 1):
                fn, args = self.steps[si]
                self._emit_step(sink, fn, args, MAIN)
            return
        # branch chains on their own streams (forked behind everything enqueued so far), joined where the trunk needs them
        sides = {}

        def side(sid):
            if sid not in sides:
                st = self.plan._side_stream(100 + sid)
                sink.wait(st, MAIN)
                sides[sid] = st
            return sides[sid]
        if first:
            for sid in self.hoisted:
                side(sid)
        leaves = [self.plan._side_stream(LEAF_SID + k) for k in range(N_LEAF)]
        dirty = [True] * N_LEAF                            # main has launches this leaf stream has not been ordered behind yet
        used_leaf = [False] * N_LEAF
        for si in range(lo, hi + 1):
            fn, args = self.steps[si]
            for w in self.step_wait.get(si, ()):
                if w in sides:                               # (a branch whose steps all ran in an earlier range is already joined)
                    sink.wait(MAIN, sides[w])
            sid = self.step_sid[si]
            if sid == 0:
                self._emit_step(sink, fn, args, MAIN)
                dirty = [True] * N_LEAF
            elif sid >= LEAF_SID:
                k = sid - LEAF_SID
                if dirty[k]:
                    sink.wait(leaves[k], MAIN)             # everything this launch reads was produced by earlier steps
                    dirty[k] = False
                used_leaf[k] = True
                self._emit_step(sink, fn, args, leaves[k])
            else:
                self._emit_step(sink, fn, args, side(sid))
        if last:
            for w in self.step_wait.get(len(self.steps), ()):
                if w in sides:
                    sink.wait(MAIN, sides[w])
        # fold every stream that may have written gradients back into the main stream (a hook's all-reduce is ordered behind it)
        for st in sides.values():
            sink.wait(MAIN, st)
        for k, lf in enumerate(leaves):
            if used_leaf[k]:
                sink.wait(MAIN, lf)

    @staticmethod
    def _emit_step(sink, fn, args, key):
        if args is None:                         # host-side tensor bookkeeping (channel split of slice_and_cat's gradient)
            sink.host(fn, key)
        elif fn is _pack_dgrad_from_packed:
            sink.host(lambda fn=fn, args=args: fn(*args), key)
        else:
            sink.launch(fn, args, key)


class PixelLinkBackwardPlan(BackwardPlan):
    """HIP backward of the PixelLink++ launch plan (gssd/pixellink.py): the same record walk, rooted at d(out_1) [B,2,H,W] and
    d(out_2) [B,16,H,W] (what PixelLinkLoss's backward hands over) instead of d(loc), d(conf)."""

    def __init__(self, plan):
        Ho = plan.H_out
        self.d_out1 = torch.zeros(plan.B, 2, Ho, Ho, device=plan.dev)
        self.d_out2 = torch.zeros(plan.B, 16, Ho, Ho, device=plan.dev)
        super().__init__(plan)

    def run(self, d_out1, d_out2):
        self.d_out1.copy_(d_out1)
        self.d_out2.copy_(d_out2)
        return self._execute()


