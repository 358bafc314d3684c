"""Rank program of tests/test_gpu_multi.py (one process per GPU, RCCL): two training steps from identical state, once with the
overlapped segment reducer and once with the one-call flat all-reduce; both must leave the same averaged gradients on every rank,
and those must equal the mean of the per-rank gradients computed without any collective."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')):
    sys.path.insert(0, p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch                                                               # noqa: E402
import torch.distributed as dist                                           # noqa: E402


def main():
    from gssd import dist as gd, synth
    from layers.modules import MultiBoxLoss
    from models.ssd_multiphase_custom_group import build_ssd
    world, rank, local = gd.env_world()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    gd.init('nccl', dev)
    net = build_ssd('train', 300, 2, True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111 + rank))
    net = net.to(dev).train()
    gd.broadcast_params(net)                                               # rank 0's weights everywhere (seeds differed on purpose)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    x = synth.synth_images(4, seed=gd.shard_seed(7, rank)).to(dev)
    tg = [t.to(dev) for t in synth.synth_targets(4, seed=gd.shard_seed(7, rank))]
    params = [p for p in net.parameters() if p.requires_grad]

    def grads(mode):
        net.load_state_dict(state)                                         # same BN running stats / spectral-norm u, v each time
        for p in params:
            p.grad = None
        red = gd.OverlappedGradReducer(world)
        ll, lc = crit(net(x), tg)
        if mode == 'overlap':
            red.arm(net)
        (ll + lc).backward()
        if mode == 'overlap':
            red.finish()
            assert red.overlapped_last and net._engine.grad_segment_hook is None
        elif mode == 'flat':
            gd.allreduce_grads(params, world)
        torch.cuda.synchronize()
        return torch.cat([p.grad.reshape(-1).clone() for p in params])
    g_local = grads('none')
    g_mean = g_local.clone()
    dist.all_reduce(g_mean)
    g_mean /= world
    g_flat, g_over = grads('flat'), grads('overlap')
    # accumulation into existing gradients: the hook must be withheld and finish() must still average p.grad
    red = gd.OverlappedGradReducer(world)
    net.load_state_dict(state)
    for p in params:
        p.grad = torch.zeros_like(p)
    ll, lc = crit(net(x), tg)
    red.arm(net)
    (ll + lc).backward()
    red.finish()
    torch.cuda.synchronize()
    g_acc = torch.cat([p.grad.reshape(-1) for p in params])
    den = float(g_mean.abs().max())
    out = dict(rank=rank, world=world, flat=float((g_flat - g_mean).abs().max()) / den, over=float((g_over - g_mean).abs().max()) / den,
               acc=float((g_acc - g_mean).abs().max()) / den, acc_overlapped=bool(red.overlapped_last),
               differs_from_local=float((g_mean - g_local).abs().max()) / den)
    allout = [None] * world
    dist.all_gather_object(allout, out)
    if rank == 0:
        print(json.dumps(allout))
    gd.finish()


if __name__ == '__main__':
    main()
