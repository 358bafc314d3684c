"""Soak run of the training step at the benchmarked batch: N SGD steps of GSSD++ on ONE synthetic batch of 32 in fp32 and in the bf16
storage mode -- the loss must fall and stay finite, and the bf16 curve must track the fp32 one (same data, same initial weights)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
for p in (ROOT, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')):
    sys.path.insert(0, p)
import torch
from gssd import synth
from layers.modules import MultiBoxLoss
from models.ssd_multiphase_custom_group import build_ssd
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = 32
x = synth.synth_images(B, seed=5).to(dev)
tg = synth.synth_targets(B, seed=5)
crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
curves = {}
for mode in ('f32', 'bf16'):
    net = build_ssd('train', 300, 2, True, 4, 4, 1, True, True, True, 1, 4, True, False, 1)
    net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
    net = net.to(dev).train()
    net.compute_dtype = mode
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    losses = []
    for it in range(N):
        opt.zero_grad(set_to_none=True)
        loc, conf, pri = net(x)
        ll, lc = crit((loc, conf, pri), tg)
        (ll + lc).backward()
        opt.step()
        losses.append(float(ll.detach() + lc.detach()))
        assert losses[-1] == losses[-1] and losses[-1] < 1e6, (mode, it, losses[-1])
    curves[mode] = losses
    print(mode, 'loss every 10 steps:', [round(v, 3) for v in losses[::10]], 'last', round(losses[-1], 3))
assert curves['f32'][-1] < 0.5 * curves['f32'][0] and curves['bf16'][-1] < 0.5 * curves['bf16'][0]
print('ratio bf16 / f32 at the end:', round(curves['bf16'][-1] / curves['f32'][-1], 3))
