"""Subprocess of tests/test_gpu_multi.py::test_global_normalizer_equals_single_batch: rank r of `world` (all on GPU 0, gloo collectives --
RCCL refuses two ranks on one device) runs MultiBoxLoss with global_normalizer = True on its own B images' predictions; rank 0 also runs the
single (world x B)-image batch with the reference's normaliser and compares: the data-parallel MEAN over ranks of the losses and of
d(loss)/d(loc, conf) must equal the one big batch (multibox_loss.py:117 with N over all images)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')):
    sys.path.insert(0, p)
import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.distributed as dist        # noqa: E402


def data(rank, B, dev):
    from gssd import synth
    rng = np.random.default_rng(50 + rank)
    loc = torch.from_numpy(rng.normal(0, 1.0, size=(B, 8732, 4)).astype(np.float32)).to(dev)
    conf = torch.from_numpy(rng.normal(0, 2.0, size=(B, 8732, 2)).astype(np.float32)).to(dev)
    tg = [t.to(dev) for t in synth.synth_targets(B, seed=50 + rank)]
    return loc, conf, tg


def main():
    from gssd import dist as gd
    from layers.modules import MultiBoxLoss
    from layers.functions import PriorBox
    from data import v2
    world, rank = gd.init('gloo')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    B = 3
    with torch.no_grad():
        pri = PriorBox(v2).forward().to(dev)
    crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
    crit.global_normalizer = True
    loc, conf, tg = data(rank, B, dev)
    loc.requires_grad_()
    conf.requires_grad_()
    ll, lc = crit((loc, conf, pri), tg)
    (ll + 2.0 * lc).backward()
    mine = torch.cat([ll.detach().reshape(1), lc.detach().reshape(1), loc.grad.reshape(-1), conf.grad.reshape(-1)]).cpu()
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    if rank == 0:
        parts = [data(r, B, dev) for r in range(world)]
        L = torch.cat([p[0] for p in parts]).requires_grad_()
        Cf = torch.cat([p[1] for p in parts]).requires_grad_()
        T = [t for p in parts for t in p[2]]
        ref = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
        rl, rc = ref((L, Cf, pri), T)
        (rl + 2.0 * rc).backward()
        n = B * 8732
        mean_l = float(sum(g[0] for g in got) / world)
        mean_c = float(sum(g[1] for g in got) / world)
        dl = torch.cat([g[2:2 + 4 * n] for g in got]) / world
        dc = torch.cat([g[2 + 4 * n:] for g in got]) / world
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        # and the default (per-rank N) is NOT the big batch unless the ranks happen to have equal positives
        loc0, conf0, tg0 = data(0, B, dev)
        crit.global_normalizer = False
        l0, c0 = crit((loc0, conf0, pri), tg0)
        print('GLOBALNJSON ' + json.dumps(dict(world=world, loss=[mean_l, mean_c], ref=[float(rl), float(rc)],
                                               dloc_rel=rel(dl, L.grad.reshape(-1).cpu()), dconf_rel=rel(dc, Cf.grad.reshape(-1).cpu()),
                                               rank0_local=[float(l0), float(c0)], rank0_global=[float(got[0][0]), float(got[0][1])])))
    gd.barrier(dev)
    gd.finish()


if __name__ == '__main__':
    main()
