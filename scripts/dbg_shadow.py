import sys, os, copy, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'grouped-ssd-pytorch_amd'))
from gssd import synth
from gssd.autograd_shadow import _conv, _bn
from models.ssd_multiphase_custom_group import build_ssd
net = build_ssd('train', 300, 2, True, 4, 4, 1, True, False, False, 0, 1, False, False, 1)
net.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1111))
netg = copy.deepcopy(net).cuda()
x = synth.synth_images(4, seed=9); xg = x.cuda()
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
with torch.no_grad():
    for i, (m, mg) in enumerate(zip(net.vgg, netg.vgg)):
        if isinstance(m, torch.nn.Conv2d): x, xg = _conv(m, x), _conv(mg, xg)
        elif isinstance(m, torch.nn.BatchNorm2d): x, xg = _bn(m, x), _bn(mg, xg)
        elif isinstance(m, torch.nn.ReLU): x, xg = F.relu(x), F.relu(xg)
        else: x, xg = m(x), mg(xg)
        print(i, type(m).__name__, f'{rel(xg, x):.2e}', 'then resync' if rel(xg, x) > 1e-4 else '')
        if rel(xg, x) > 1e-4: xg = x.cuda()
