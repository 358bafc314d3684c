"""Per-launch conv times of one forward (HIP events), in plan order: which layer costs what."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
sys.path.insert(0, ROOT)
import torch
import bench
from gssd import synth
from models.ssd_multiphase_custom_group import build_ssd
cfg = sys.argv[1] if len(sys.argv) > 1 else 'gssd'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32'
args = bench.CONFIGS[cfg][0]
dev = torch.device('cuda:0')
net = build_ssd('train', 300, 2, *args)
net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
net = net.to(dev).train()
net.compute_dtype = dtype
x = synth.synth_images(32, seed=100).to(dev)
with torch.no_grad():
    for _ in range(3): net(x)
    ev = []
    net.__dict__['_events'] = ev
    for _ in range(5): net(x)
    net.__dict__['_events'] = None
torch.cuda.synchronize()
n = len(ev) // 5
names = [r['name'] for k, r in net._engine._last_plan.rec if k == 'convbn']
print(n, 'conv launches per step;', len(names), 'conv+BN records')
tot = 0
for i in range(n):
    ms = sum(ev[i + k * n][1].elapsed_time(ev[i + k * n][2]) for k in range(5)) / 5
    (tag, fl, by) = ev[i][0]
    tot += ms
    d = getattr(ev[i][0], 'desc', None)
    geo = ''
    if d is not None and os.environ.get('LAYER_GEOMETRY'):
        epi = ''.join(c for c, on in (('a', d.alpha), ('g', d.gate), ('r', d.resid), ('2', d.out2), ('x', d.in_scale), ('s', d.stats), ('R', d.relu), ('i', d.m_per_image)) if on)
        geo = f'  {getattr(ev[i][0], "layer", None)} H{d.H} {d.cin_g * d.groups}>{d.Cout} g{d.groups} k{d.KH} s{d.stride} p{d.pad} d{d.dil} mode{d.out_mode} splitk{d.split_k} [{epi}]'
    print(f'{i:3d} {tag:24s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF  {by / ms / 1e6:7.0f} GB/s  {fl / 1e9:7.2f} GF  {by / 1e6:7.1f} MB{geo}')
print('total conv ms', tot)
