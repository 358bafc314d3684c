"""Is the forward + loss step host- or GPU-bound?  Wall time to ENQUEUE n steps vs wall time until they are done."""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd')); sys.path.insert(0, ROOT)
import torch, bench
from gssd import synth
from layers.modules import MultiBoxLoss
from models.ssd_multiphase_custom_group import build_ssd
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = torch.device('cuda:0')
net = build_ssd('train', 300, 2, *bench.CONFIGS['gssdpp'][0])
net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
net = net.to(dev).train(); net.compute_dtype = dtype
crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
x = synth.synth_images(32, seed=100).to(dev); tg = [t.to(dev) for t in synth.synth_targets(32, seed=100)]
def step():
    with torch.no_grad(): return crit(net(x), tg)
for _ in range(10): step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(dtype, 'enqueue ms/step', round((t1 - t0) / n * 1e3, 3), 'total ms/step', round((t2 - t0) / n * 1e3, 3))
t0 = time.perf_counter()
for _ in range(n):
    with torch.no_grad(): out = net(x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(dtype, 'forward only: enqueue', round((t1 - t0) / n * 1e3, 3), 'total', round((t2 - t0) / n * 1e3, 3))
out = net(x) if False else out
t0 = time.perf_counter()
for _ in range(n):
    with torch.no_grad(): crit(out, tg)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(dtype, 'loss only: enqueue', round((t1 - t0) / n * 1e3, 3), 'total', round((t2 - t0) / n * 1e3, 3))
if dtype == 'f32':
    # full training step: forward + loss + backward (multi-stream) + SGD
    opt = torch.optim.SGD(net.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
    def tstep():
        opt.zero_grad(set_to_none=True)
        ll, lc = crit(net(x), tg)
        (ll + lc).backward()
        opt.step()
    for _ in range(3): tstep()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n): tstep()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(dtype, 'training step: enqueue', round((t1 - t0) / n * 1e3, 3), 'total', round((t2 - t0) / n * 1e3, 3))
