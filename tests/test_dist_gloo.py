"""N > 1 path on CPU: two processes over gloo exercise the sharding / barrier / max-over-ranks harness bench.py uses
with RCCL on the GPUs (the forward + loss path itself has no collective)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from gssd import dist as gd, synth
    w, r = gd.init('gloo')
    assert (w, r) == (world, rank)
    x = synth.synth_images(2, seed=gd.shard_seed(100, rank))          # each rank owns different images
    tg = synth.synth_targets(2, seed=gd.shard_seed(100, rank))
    gd.barrier()
    t = gd.max_over_ranks(1.0 + rank)                                  # slowest rank defines the step time
    rate = gd.aggregate_rate(w, 32, 10, t)
    out.put((rank, float(x.sum()), len(tg), t, rate))
    # gradient averaging over ranks (flat-buffer all-reduce) and the initial weight broadcast
    lin = torch.nn.Linear(4, 3)
    gd.broadcast_params(lin)
    for i, q in enumerate(lin.parameters()):
        q.grad = torch.full_like(q, float(rank + 1 + i))
    n = gd.allreduce_grads(list(lin.parameters()))
    # gradients that are slices of one flat tensor (what the HIP backward plan hands out) are reduced in place, no copies
    flat = torch.zeros(16)
    lin2 = torch.nn.Linear(4, 3)
    lin2.weight.grad = flat[0:12].view(3, 4)
    lin2.bias.grad = flat[12:15].view(3)
    flat[:12] = float(rank + 1)
    flat[12:15] = float(10 * (rank + 1))
    n2 = gd.allreduce_grads(list(lin2.parameters()))
    same_storage = lin2.weight.grad.data_ptr() == flat.data_ptr()
    out.put((rank, n, [float(q.grad.mean()) for q in lin.parameters()], float(lin.weight.sum()),
             (n2, same_storage, float(lin2.weight.grad.mean()), float(lin2.bias.grad.mean()))))
    gd.barrier()
    gd.finish()


def test_two_rank_harness():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(2 * world)]
    res = sorted(r for r in got if len(r) == 5 and not isinstance(r[4], tuple))
    res2 = sorted(r for r in got if len(r) == 5 and isinstance(r[4], tuple))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] != res[1][1]                     # different shards
    assert res[0][3] == res[1][3] == 2.0              # max over ranks
    assert res[0][4] == res[1][4] == 2 * 32 * 10 / 2.0
    assert res2[0][1] == res2[1][1] == 15 and res2[0][2] == res2[1][2] == [1.5, 2.5]      # mean over ranks
    assert res2[0][3] == res2[1][3]                                                            # broadcast weights
    assert res2[0][4] == res2[1][4] == (16, True, 1.5, 15.0)                                   # flat-slice gradients, in place


def _worker_overlap(rank, world, port, out):
    """The segmented, overlapped reducer on a flat buffer laid out like BackwardPlan's (16-byte aligned parameter slices, ranges
    becoming ready last-to-first): driven exactly as gssd/backward.py drives it."""
    sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import types
    from gssd import dist as gd
    from gssd.backward import BackwardPlan
    gd.init('gloo')
    shapes = [(64, 3, 3, 3), (64,), (128, 64, 3, 3), (128,), (36, 128, 3, 3), (36,), (1,)]
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
    bp = BackwardPlan.__new__(BackwardPlan)                       # only the flat-buffer bookkeeping of the plan, no kernels
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    bp.flat = torch.zeros(n)
    bp._offs = {id(p): (o, (p.numel() + 3) // 4 * 4) for p, o in zip(params, offs)}
    bp._slices = {id(p): bp.flat[o:o + p.numel()].view(p.shape) for p, o in zip(params, offs)}
    bp._last_write = {id(p): len(params) - 1 - i for i, p in enumerate(params)}      # reverse order, like a backward
    segs = bp.segments(3)
    assert segs[0][0] == 0 and segs[-1][1] == n and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
    assert [s[2] for s in segs] == sorted([s[2] for s in segs], reverse=True)        # the last range is ready first
    for p in params:
        bp._slices[id(p)].fill_(float(rank + 1))
        p.grad = bp._slices[id(p)]
    net = types.SimpleNamespace(_engine=types.SimpleNamespace())
    red = gd.OverlappedGradReducer(world, nseg=3)
    red.arm(net)
    hook = net._engine.grad_segment_hook
    for k, (lo, hi, ready) in sorted(enumerate(segs), key=lambda kv: kv[1][2]):      # in the order the backward would fire them
        hook(k, bp.flat[lo:hi])
    nred = red.finish()
    assert net._engine.grad_segment_hook is None and red.overlapped_last      # finish() disarms the engine
    res = (rank, nred == n, min(float(p.grad.min()) for p in params), float(bp.flat.max()),
           [float(p.grad.mean()) for p in params][:3])
    # the fallback: the autograd glue withheld the hook (accumulation into foreign p.grad tensors): nothing was reduced under
    # the backward, finish() must average p.grad itself and leave the engine disarmed
    net.parameters = lambda: iter(params)
    for p in params:
        p.grad = torch.full_like(p, float(10 * (rank + 1)))
    red.arm(net)
    net._engine.grad_segment_skipped = True                # what GssdTrainFn.backward sets when p.grad is not the flat slice
    red.finish()
    assert not red.overlapped_last and net._engine.grad_segment_hook is None
    assert all(float(p.grad.min()) == float(p.grad.max()) == 5.0 * (world + 1) for p in params)
    out.put(res)
    gd.barrier()
    gd.finish()


@pytest.mark.parametrize('world', [2, 8])
def test_overlapped_reducer_ranks(world, tmp_path, monkeypatch):
    """world 8 = BASELINE configs[3]'s launch shape: the ranges fire last-to-first on every rank and each element is
    averaged exactly once; the ranks meet through the FileStore rendezvous bench.py's self-launcher uses (gssd/dist.py::init)."""
    port = _free_port()
    monkeypatch.setenv('GSSD_DIST_INIT_FILE', str(tmp_path / 'store'))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_overlap, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean = (world + 1) / 2.0                    # ranks fill their gradients with rank + 1
    for r in got:
        assert r[1] and r[2] == r[3] == mean and r[4] == [mean] * 3                  # every element averaged exactly once, in place
