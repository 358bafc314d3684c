import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
from gssd import synth
from gssd.autograd_shadow import shadow_forward
from layers.modules import MultiBoxLoss
from models.ssd_multiphase_custom_group import build_ssd
torch.backends.cudnn.benchmark = (sys.argv[1] == '1')
net = build_ssd('train', 300, 2, True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)
net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
net = net.cuda().train()
B = 32
x = synth.synth_images(B, seed=1).cuda(); tg = [t.cuda() for t in synth.synth_targets(B, seed=1)]
crit = MultiBoxLoss(2, 0.5, True, 0, True, 3, 0.5, False, True)
def step():
    for p in net.parameters(): p.grad = None
    ll, lc = crit(net(x), tg); (ll + lc).backward()
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print('step', i, f'{(time.perf_counter()-t0)*1e3:.1f} ms')
# split: shadow forward only / forward+backward through shadow
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(): shadow_forward(net, x)
    torch.cuda.synchronize(); print('shadow fwd no_grad', f'{(time.perf_counter()-t0)*1e3:.1f} ms')
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=70))
