"""Where does the HOST time of a training step go?  cProfile over n enqueue-only training steps (GSSD++, B = 32), top entries by own time,
plus wall-clock splits of the step's parts.  usage: python scripts/host_profile.py [f32|bf16]"""
import cProfile, pstats, io, sys, os, time
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
B = int(os.environ.get("B", 32))
x = synth.synth_images(B, seed=100).to(dev); tg = [t.to(dev) for t in synth.synth_targets(B, seed=100)]
opt = torch.optim.SGD(net.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
parts = {}
def tstep(timed=False):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    out = net(x); t.append(time.perf_counter())
    ll, lc = crit(out, tg); t.append(time.perf_counter())
    (ll + lc).backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if timed:
        for k, name in enumerate(('zero_grad', 'forward', 'loss', 'backward', 'sgd')):
            parts[name] = parts.get(name, 0.0) + t[k + 1] - t[k]
for _ in range(4): tstep()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): tstep(True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(dtype, 'training step: enqueue', round((t1 - t0) / n * 1e3, 3), 'total', round((t2 - t0) / n * 1e3, 3))
print('  host ms per step:', {k: round(v / n * 1e3, 3) for k, v in parts.items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(n): tstep()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:6000])
