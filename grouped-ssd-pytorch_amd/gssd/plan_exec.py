"""Execution of a forward launch plan: eager launches (optionally bracketed by HIP events for bench.py) and hipGraph capture / replay with
the branch streams forked and joined inside the graph.  Mixin of engine._Plan."""
import ctypes as C
import os
import torch
from . import _lib, ops
from ._lib import lib
from . import plan_common
from .plan_common import ALL_STREAMS, USE_BRANCH_STREAMS


class NodeEvent:
    """A HIP event recorded through gssd_event_record_node: under stream capture an event-record node of the graph (torch refuses external
    events on ROCm; the HIP runtime has them)."""

    def __init__(self):
        h = C.c_void_p()
        _lib.check(lib.gssd_event_create(C.byref(h)))
        self.h = h

    def record(self, stream_ptr):
        _lib.check(lib.gssd_event_record_node(self.h, stream_ptr))

    def elapsed_time(self, other):
        ms = C.c_float()
        _lib.check(lib.gssd_event_elapsed_ms(self.h, other.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib.gssd_event_destroy(self.h)
        except Exception:
            pass


class PlanExecMixin:
    # ------------------------------------------------------------------------------------------------
    def run(self, x, events=None):
        # (``events.nodes``, round 6: kernel instances bracketed by event-record NODES inside the replayed hipGraph -- NodeEvent below --: nothing is
        # cut out of the graph, the launches keep their neighbours on the other branches; after a replay has finished, (tag, start, stop) of the
        # LAST replay are in ``events`` and ``start.elapsed_time(stop)`` reads them)
        """``events``: optional list; when given, every tagged launch (or only the kernel instances named in ``events.only``) is
        bracketed by a pair of HIP events recorded on the launch stream and (tag, start, end) is appended (bench.py's live
        roofline measurement).

        From its third run on a plan replays itself from hipGraphs: the ~200-270 launches of a step are static (preallocated
        buffers, descriptors by value), so the host side of a step shrinks from a ctypes call per kernel (~2.5 ms) to a few graph
        launches.  Launches that must be bracketed by events stay eager and split the plan into graph segments around them."""
        self.generation += 1
        x = x.contiguous().float()
        only = getattr(events, 'only', None) if events is not None else None
        nodes = getattr(events, 'nodes', None) if events is not None else None
        self._runs = getattr(self, '_runs', 0) + 1
        if plan_common.USE_GRAPH and (events is None or only or nodes) and self._runs > 2:
            return self._run_graphs(x, events, only, nodes)
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

    def _run_graphs(self, x, events, only, nodes=None):
        key = (tuple(sorted(only)) if only else None) if not nodes else ('nodes', tuple(sorted(nodes)), tuple(sorted(only)) if only else None)
        cache = self.__dict__.setdefault('_graphs', {})
        # Zero-copy input (round 6): a caller that hands over the SAME device buffer step after step (a training loop's pinned staging buffer, the
        # benchmark's resident batch) gets a graph whose first node reads that buffer in place -- the 138 MB x -> static-buffer copy in front of
        # every replay was 50 us of an 8.3 ms step.  A pointer seen twice in a row is captured (at most two such graphs per plan); any other
        # input takes the generic graph over the plan's static input buffer.
        ptr = x.data_ptr()
        dkey = ('direct', ptr, key)
        if dkey not in cache and self.__dict__.get('_last_in_ptr') == ptr and sum(1 for k in cache if isinstance(k, tuple) and k[:1] == ('direct',)) < 2:
            cache[dkey] = self._capture(x, only, in_ptr=ptr, nodes=nodes)
        self.__dict__['_last_in_ptr'] = ptr
        if dkey in cache:
            segs = cache[dkey]
            self._x_keepalive = x
        else:
            if key not in cache:
                cache[key] = self._capture(x, only, nodes=nodes)
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
        if nodes:
            events.extend(self.__dict__.get('_node_events', {}).get(id(segs), ()))
        return self._gloc.clone(), self._gconf.clone()

    def _side_stream(self, sid):
        pool = self.__dict__.setdefault('_side_streams', {})
        if sid not in pool:
            pool[sid] = torch.cuda.Stream(device=self.dev)
        return pool[sid]

    def _capture(self, x, only, in_ptr=None, nodes=None):
        """Capture the plan as hipGraph segments over static input / output buffers; the steps named in ``only`` stay eager.  ``in_ptr``: the graph
        reads the input at this address instead of the plan's static input buffer (zero-copy replay, _run_graphs)."""
        B, dev = self.B, self.dev
        if getattr(self, '_gx', None) is None:
            self._gx = torch.empty_like(x)
            self._gloc = torch.zeros(B, self.P, 4, device=dev, dtype=torch.float32)
            self._gconf = torch.zeros(B, self.P, self.nc, device=dev, dtype=torch.float32)
        self._set_outputs(self._gloc, self._gconf)
        self.steps[self._pack_step].args[0] = in_ptr if in_ptr is not None else self._gx.data_ptr()
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
        node_evs = []

        def launch(st, stream_ptr):
            if nodes and st.tag is not None and st.tag[0] in nodes:
                e0, e1 = NodeEvent(), NodeEvent()
                e0.record(stream_ptr)
                self._launch(st, stream_ptr)
                e1.record(stream_ptr)
                node_evs.append((st.tag, e0, e1))
            else:
                self._launch(st, stream_ptr)
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
                        launch(st, main.cuda_stream)
                        continue
                    side = forked.get(st.sid)
                    if side is None:                       # fork: the branch starts behind everything the trunk has enqueued
                        side = self._side_stream(st.sid)
                        side.wait_stream(main)
                        forked[st.sid] = side
                    if st.wait is not None and st.wait in forked:
                        side.wait_stream(forked[st.wait])
                    launch(st, side.cuda_stream)
                for side in forked.values():               # join: a graph segment ends with every branch folded back
                    main.wait_stream(side)
                if last and self.training and self.nbt:
                    torch._foreach_add_(self.nbt, 1)
            segs.append(('graph', g))
        # the capture itself does not execute anything: the caller's replay is the run
        if nodes:
            self.__dict__.setdefault('_node_events', {})[id(segs)] = node_evs
        return segs
