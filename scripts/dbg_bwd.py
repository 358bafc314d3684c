import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
from oracle import gssd_oracle as O
from gssd import synth
from gssd.autograd_shadow import shadow_forward
from models.ssd_multiphase_custom_group import build_ssd
args = (True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)
net = build_ssd('train', 300, 2, *args)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111)
net.load_state_dict(sd); net = net.cuda().train()
x = synth.synth_images(4, seed=9)
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
with torch.no_grad():
    loc, conf, _ = net(x.cuda())
    lo, co, _ = O.gssd_forward(sd, x)
    for tf32 in (True, False):
        torch.backends.cudnn.allow_tf32 = tf32; torch.backends.cuda.matmul.allow_tf32 = tf32
        ls, cs = shadow_forward(net, x.cuda())
        print('tf32', tf32, 'shadow vs oracle', rel(ls, lo), rel(cs, co), ' engine vs oracle', rel(loc, lo), rel(conf, co))
# gradient: shadow on GPU vs shadow on CPU (same code) for a mid layer
rng = np.random.default_rng(0)
r1 = torch.from_numpy(rng.normal(size=(4, 8732, 4)).astype(np.float32)); r1[:, 8728:] = 0
import copy
netc = copy.deepcopy(net).cpu()
for dev_net, dev in ((net, 'cuda'), (netc, 'cpu')):
    for p in dev_net.parameters(): p.grad = None
    l, c = shadow_forward(dev_net, x.to(dev))
    (l * r1.to(dev)).sum().backward()
g1 = dict(net.named_parameters()); g2 = dict(netc.named_parameters())
for k in ('vgg.0.weight', 'vgg.14.weight', 'vgg.31.weight', 'fuse_11.weight', 'loc.0.weight'):
    print(k, 'shadow gpu vs shadow cpu', rel(g1[k].grad, g2[k].grad))
# oracle autograd
sdg = {k: (v.clone().requires_grad_() if (v.is_floating_point() and not k.endswith(('running_mean', 'running_var', 'weight_u', 'weight_v'))) else v) for k, v in sd.items()}
lo, co, _ = O.gssd_forward(sdg, x)
(lo * r1).sum().backward()
for k in ('vgg.0.weight', 'vgg.14.weight', 'vgg.31.weight', 'fuse_11.weight', 'loc.0.weight'):
    print(k, 'shadow cpu vs oracle', rel(g2[k].grad, sdg[k].grad))
