"""Host side of the launch-plan runner (csrc/plan_run.hip, include/gssd_hip.h: gssd_plan_run).

A plan's step list is static -- preallocated buffers, descriptors kept alive by the plan, the same stream forks and joins every step -- so
the Python loop that made one ctypes call per kernel (and two torch calls per fork / join) is replaced by: build the op array ONCE, then
one ``gssd_plan_run`` call per segment.  The plan code writes its control flow once against a *sink*: ``EagerSink`` executes it on the
spot (the reference behaviour; hipGraph capture and GSSD_NO_PLAN_RUN=1 use it), ``RecordSink`` records it into ``Program`` objects
(runs of LAUNCH / WAIT ops) and host callables (the few steps that are torch ops), which are cached and replayed.
"""
import ctypes as C
import os
import struct

import torch

from . import _lib

lib = _lib.lib
# GSSD_NO_PLAN_RUN=1: every launch from Python again (ablation; scripts/host_vs_gpu.py)
USE_PLAN_RUN = os.environ.get('GSSD_NO_PLAN_RUN', '0') != '1'
MAIN = 'main'           # stream key of the stream that is current when a program runs

_M64 = (1 << 64) - 1
_fn_cache = {}


def fn_info(fn):
    """(index in the runner's table, argtypes without the stream) of a ctypes entry point."""
    name = fn.__name__
    info = _fn_cache.get(name)
    if info is None:
        idx = lib.gssd_plan_fn_index(name.encode())
        if idx < 0:
            raise _lib.GssdError(f'{name} is not an entry point gssd_plan_run can launch (its last parameter is not the stream?)')
        argtypes = list(_lib.SIGNATURES[name][1][:-1])
        if lib.gssd_plan_fn_nargs(idx) != len(argtypes):
            raise _lib.GssdError(f'{name}: the binding declares {len(argtypes)} parameters, the library {lib.gssd_plan_fn_nargs(idx)}')
        info = _fn_cache[name] = (idx, argtypes)
    return info


def word(t, v):
    """One argument as the 64-bit word gssd_plan_run expects."""
    if t is _lib.c_f:
        return struct.unpack('<I', struct.pack('<f', float(v)))[0]
    if t is _lib.c_d:
        return struct.unpack('<Q', struct.pack('<d', float(v)))[0]
    if t in (_lib.c_i, _lib.c_i64, C.c_longlong, C.c_int64):
        return int(v) & _M64
    # pointers: integers (tensor.data_ptr()), None, ctypes.byref(struct) / pointer(struct) / a structure instance kept alive by the plan
    if v is None:
        return 0
    if isinstance(v, int):
        return v & _M64
    if hasattr(v, '_obj'):                                   # ctypes.byref(...)
        return C.addressof(v._obj)
    if isinstance(v, (C.Structure, C.Array)):
        return C.addressof(v)
    if hasattr(v, 'contents'):                               # ctypes.pointer(...)
        return C.addressof(v.contents)
    if hasattr(v, 'value'):                                  # c_void_p(...)
        return int(v.value or 0)
    raise TypeError(f'cannot pass {type(v).__name__} to gssd_plan_run')


class Program:
    """A run of LAUNCH / WAIT ops as one gssd_plan_op array.  ``streams[0]`` is the stream current at run time; the others are the plan's
    side streams (torch.cuda.Stream objects kept alive here)."""

    def __init__(self, ops):
        keys = [MAIN]
        for op in ops:
            for k in (op[3],) if op[0] == 'launch' else (op[1], op[2]):
                if k is not MAIN and all(k is not q for q in keys):
                    keys.append(k)
        self.stream_objs = keys
        sidx = lambda k: next(i for i, q in enumerate(keys) if q is k)
        self.n = len(ops)
        self.arr = (_lib.PlanOp * max(self.n, 1))()
        self.keep = []
        for i, op in enumerate(ops):
            o = self.arr[i]
            if op[0] == 'launch':
                _, fn, args, sk = op
                idx, argtypes = fn_info(fn)
                if len(args) != len(argtypes):
                    raise _lib.GssdError(f'{fn.__name__}: {len(args)} arguments for {len(argtypes)} parameters')
                o.kind, o.fn, o.stream, o.nargs = _lib.PLAN_LAUNCH, idx, sidx(sk), len(args)
                for k, (t, v) in enumerate(zip(argtypes, args)):
                    o.args[k] = word(t, v)
                self.keep.append(args)                       # descriptors / byref objects stay alive with the program
            else:
                _, dst, src = op
                o.kind, o.fn, o.stream, o.nargs = _lib.PLAN_WAIT, sidx(src), sidx(dst), 0
        self.streams = (_lib.c_fp * len(keys))()
        for i, k in enumerate(keys[1:], 1):
            self.streams[i] = k.cuda_stream
        self.failed = C.c_int(-1)

    def set_arg(self, op_index, k, t, v):
        self.arr[op_index].args[k] = word(t, v)

    def run(self, main_handle):
        self.streams[0] = main_handle
        rc = lib.gssd_plan_run(self.arr, self.n, self.streams, len(self.stream_objs), C.byref(self.failed))
        if rc != 0:
            i = self.failed.value
            name = lib.gssd_plan_fn_name(self.arr[i].fn).decode() if 0 <= i < self.n and self.arr[i].kind == _lib.PLAN_LAUNCH else 'wait'
            raise _lib.GssdError(f'libgssd_hip: error {rc} in plan op {i} ({name}): {lib.gssd_last_error().decode()}')


class EagerSink:
    """Executes the plan's control flow on the spot: one ctypes call per launch, torch stream waits."""

    def __init__(self):
        self.main = torch.cuda.current_stream()

    def _s(self, key):
        return self.main if key is MAIN else key

    def launch(self, fn, args, key):
        rc = fn(*args, self._s(key).cuda_stream)
        if rc != 0:
            _lib.check(rc)

    def wait(self, dst, src):
        self._s(dst).wait_stream(self._s(src))

    def host(self, call, key=MAIN):
        """A step that is torch code: runs with stream ``key`` current."""
        if key is MAIN:
            call()
        else:
            with torch.cuda.stream(key):
                call()


class RecordSink:
    """Records the same control flow: -> [('prog', Program) | ('host', callable, stream key)]."""

    def __init__(self):
        self.segs, self.cur = [], []

    def launch(self, fn, args, key):
        self.cur.append(('launch', fn, tuple(args), key))

    def wait(self, dst, src):
        if dst is not src:
            self.cur.append(('wait', dst, src))

    def host(self, call, key=MAIN):
        self._flush()
        self.segs.append(('host', call, key))

    def _flush(self):
        if self.cur:
            self.segs.append(('prog', Program(self.cur)))
            self.cur = []

    def finish(self):
        self._flush()
        return self.segs


def replay(segs):
    main = torch.cuda.current_stream()
    handle = main.cuda_stream
    for seg in segs:
        if seg[0] == 'prog':
            seg[1].run(handle)
        elif seg[2] is MAIN:
            seg[1]()
        else:
            with torch.cuda.stream(seg[2]):
                seg[1]()
