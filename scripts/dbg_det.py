import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
from gssd import synth
from models.ssd_multiphase_custom_group import build_ssd
g = np.load(os.path.join(ROOT, 'tests/golden/e2e.npz'))
args = (True, 4, 4, 1, True, True, True, 0, 1, False, False, 1); name = 'gssd_sa'
net = build_ssd('train', 300, 2, *args)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
net.load_state_dict(sd); net = net.cuda().train()
x = synth.synth_images(4, seed=5).cuda()
with torch.no_grad():
    net(x)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d): m.momentum = 1.0
    net(x)
    nt = build_ssd('test', 300, 2, *args); nt.load_state_dict(net.state_dict()); nt = nt.cuda().eval()
    det = nt(x).cpu().numpy()
ref = g[f'{name}.det']
print('kept', [(det[b,1,:,0]>0).sum() for b in range(4)], [(ref[b,1,:,0]>0).sum() for b in range(4)])
for b in range(4):
    d = det[b,1]; r = ref[b,1]
    n = (r[:,0]>0).sum()
    bad = np.nonzero(np.abs(d[:n]-r[:n]).max(1) > 5e-5)[0]
    print(b, 'rows differing in place:', bad[:20], 'max score diff', np.abs(d[:n,0]-r[:n,0]).max())
    if len(bad):
        i = bad[0]; print(d[max(i-1,0):i+3]); print(r[max(i-1,0):i+3])
