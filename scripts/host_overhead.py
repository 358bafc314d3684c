"""Host-side cost of one step (Python + ctypes launches) vs the GPU time: is the launch loop ahead of the GPU?"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')); sys.path.insert(0, ROOT)
import torch, bench
from gssd import synth
from layers.modules import MultiBoxLoss
from models.ssd_multiphase_custom_group import build_ssd
dev = torch.device('cuda:0')
net = build_ssd('train', 300, 2, *bench.CONFIGS['gssd'][0])
net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
net = net.to(dev).train()
crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
x = synth.synth_images(32, seed=100).to(dev)
tg = [t.to(dev) for t in synth.synth_targets(32, seed=100)]
def step():
    with torch.no_grad():
        return crit(net(x), tg)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host issue time {1e3 * (t1 - t0) / 50:.3f} ms/step; total {1e3 * (t2 - t0) / 50:.3f} ms/step (GPU-bound if issue < total)')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
