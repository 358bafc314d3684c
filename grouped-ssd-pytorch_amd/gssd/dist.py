"""One-process-per-GPU harness of the sharded path: rank r owns images [r*B, (r+1)*B) of the global
batch; the forward + loss path has NO data-path collective, so the only communication is the timing barrier and the
max-over-ranks of the elapsed time.  Backend 'nccl' (= RCCL over xGMI) on GPUs, 'gloo' in the CPU tests."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend, device=None):
    world, rank, _ = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        kw = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
        rdzv = os.environ.get('GSSD_DIST_INIT_FILE')
        if rdzv:
            # bench.py's self-launcher: a file-store rendezvous in a directory the parent owns -- no TCP port is picked ahead of the bind, so
            # nothing on the host can take it in between (the loopback-port race ADVICE r5 names; torch.distributed.run keeps env://)
            kw.update(init_method='file://' + rdzv, world_size=world, rank=rank)
        dist.init_process_group(backend, **kw)
    return world, rank


def world_size():
    """Ranks the process group actually has (1 when no group was initialised: no collective can run)."""
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_seed(base_seed, rank):
    """Each rank draws its own shard of the global synthetic batch."""
    return base_seed + rank


def barrier(device=None):
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
        if device is not None and device.type == 'cuda':
            torch.cuda.synchronize(device)


def max_over_ranks(seconds, device=None):
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else 'cpu')
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device=None):
    """Every rank's value, in rank order (a list of length world)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def aggregate_rate(world, per_rank_units, steps, seconds):
    """Whole-job throughput: units all ranks processed / max-over-ranks time."""
    return world * per_rank_units * steps / seconds


def _scale_(t, world):
    """t *= 1 / world in place.  Device tensors: one gssd_axpby_f32 launch on the current stream (the product path keeps its math in the
    HIP library); CPU tensors (the gloo tests): torch."""
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
        from . import _lib
        _lib.check(_lib.lib.gssd_axpby_f32(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.numel(), 1.0 / world, 0.0,
                                           torch.cuda.current_stream(t.device).cuda_stream))
    else:
        t.div_(world)
    return t


def allreduce_grads(params, world=None):
    """Data-parallel gradient averaging: ONE flat buffer (33 MB GSSD / 74 MB GSSD++ in fp32), one all-reduce (RCCL ring
    over xGMI on the GPUs), then scatter back -- replaces nn.DataParallel's reduce-to-device-0
    (train_lesion_multiphase_v2.py:593).  Returns the number of elements reduced."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    world = world or (dist.get_world_size() if dist.is_initialized() else 1)
    if world == 1:
        return sum(g.numel() for g in grads)
    base = grads[0]._base
    if base is not None and base.dim() == 1 and all(g._base is base for g in grads):
        # the HIP backward plan keeps all gradients as slices of one flat tensor: reduce it in place, no copies
        dist.all_reduce(base, op=dist.ReduceOp.SUM)
        _scale_(base, world)
        return base.numel()
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    _scale_(flat, world)
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)
    return flat.numel()


class OverlappedGradReducer:
    """Gradient averaging overlapped with the backward (BASELINE.json configs[3]).  The HIP backward plan finishes the gradients of
    the LAST layers first; its flat buffer is cut into ``nseg`` contiguous ranges and each range's all-reduce is started (async, on
    the communication stream RCCL owns) the moment the last kernel writing into it has been enqueued -- the ring all-reduce of the
    heads / extras / fuse ranges then runs under the trunk's backward, and only the first range (conv1_x .. conv3_x gradients,
    finished last) is exposed.  Usage per step:  ``red.arm(net)`` before ``loss.backward()``, ``red.finish()`` after it.

    Precondition of the overlapped form: every ``p.grad`` IS its slice of the plan's flat buffer after the backward, i.e. a plain
    ``loss.backward()`` with ``p.grad is None`` before it (``optimizer.zero_grad(set_to_none=True)``, torch's default), no tensor
    hooks on parameters, ``net.direct_grad_handout`` left on.  When that does not hold (gradient accumulation into existing
    ``p.grad``, ``zero_grad(set_to_none=False)``, hooks, ``torch.autograd.grad``) the autograd glue withholds the hook -- nothing is
    reduced under the backward -- and ``finish()`` falls back to ``allreduce_grads`` over ``p.grad``: still correct, not overlapped.
    The hook is removed from the engine in ``finish()``; a backward without ``arm()`` never sums across ranks."""

    def __init__(self, world=None, nseg=4):
        self.world = world or (dist.get_world_size() if dist.is_initialized() else 1)
        self.nseg = nseg
        self.works, self.ranges = [], []
        self.net = self.eng = None
        self.overlapped_last = False      # did the last finish() reduce under the backward (True) or fall back (False)?

    def arm(self, net):
        """Install the hook on every backward plan of ``net``'s engine that exists or gets built (cheap: one attribute)."""
        self.works, self.ranges = [], []
        self.net = net
        self.eng = net._engine if hasattr(net, '_engine') else net.module._engine
        self.eng.grad_segment_skipped = False
        self.eng.grad_segment_hook = self._hook if self.world > 1 else None

    def _hook(self, k, flat_slice):
        self.ranges.append(flat_slice)
        self.works.append(dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Wait for the ranges in flight, scale by 1 / world; returns the number of elements reduced."""
        eng = self.eng
        if eng is not None:
            eng.grad_segment_hook = None
        n = 0
        for w, t in zip(self.works, self.ranges):
            w.wait()
            _scale_(t, self.world)
            n += t.numel()
        self.overlapped_last = bool(self.works)
        skipped = eng is not None and getattr(eng, 'grad_segment_skipped', False)
        if self.works and skipped:
            raise RuntimeError('OverlappedGradReducer: one backward of this step reduced the flat gradient buffer in place and '
                               'another one accumulated into p.grad -- use gssd.dist.allreduce_grads for accumulated gradients')
        self.works, self.ranges = [], []
        if self.world > 1 and (skipped or not self.overlapped_last) and self.net is not None:
            n = allreduce_grads([p for p in self.net.parameters() if p.requires_grad], self.world)
        if eng is not None:
            eng.grad_segment_skipped = False
        return n


def broadcast_params(module, src=0):
    """Rank 0's weights and buffers to every rank before the first step."""
    if dist.is_initialized():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def finish():
    if dist.is_initialized():
        dist.destroy_process_group()
