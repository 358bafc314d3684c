"""Drop-in for ``ssd_liverdet/models/ssd.py`` (vanilla VGG-SSD300: dense convs, 3 input channels, no BatchNorm, no
fuse convs) -- BASELINE.json configs[0], SURVEY.md row a16.  Same constructor ``build_ssd(phase, size, num_classes)``,
module names and state-dict keys; ``forward`` runs the HIP engine (conv + ReLU epilogue, pool passes, L2Norm, merged
loc|conf heads)."""
import os

import torch
import torch.nn as nn

from gssd.engine import GssdEngine, VGG_CFG, EXTRAS_CFG, MBOX
from gssd.modules import L2Norm
from layers.functions.prior_box import PriorBox
from layers.functions.detection import Detect
from data.config import v2


class SSD(nn.Module):
    """Reference: models/ssd.py:9-118."""

    def __init__(self, phase, base, extras, head, num_classes):
        super().__init__()
        self.phase = phase
        self.num_classes = num_classes
        self.priorbox = PriorBox(v2)
        self.priors = self.priorbox.forward()
        self.size = 300
        self.vgg = nn.ModuleList(base)
        self.L2Norm = L2Norm(512, 20)
        self.extras = nn.ModuleList(extras)
        self.loc = nn.ModuleList(head[0])
        self.conf = nn.ModuleList(head[1])
        if phase == 'test':
            self.softmax = nn.Softmax(dim=-1)
            self.detect = Detect
        self.vanilla = True
        object.__setattr__(self, '_engine', GssdEngine(self))

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != '_engine':
                new.__dict__[k] = copy.deepcopy(v, memo)
        object.__setattr__(new, '_engine', GssdEngine(new))
        return new

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self.priors = fn(self.priors)
        self._engine.invalidate()
        return out

    def forward(self, x):
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from gssd.autograd import GssdTrainFn                   # HIP forward plan + HIP backward plan
            loc, conf = GssdTrainFn.apply(self, x, *tuple(self.parameters()))
        else:
            loc, conf = self._engine.forward(x, self.training)
        priors = self.priors if self.priors.device == x.device else self.priors.to(x.device)
        if self.phase == 'test':
            return self.detect.apply(self.num_classes, 0, 200, 0.01, 0.45, loc, conf, priors, True)
        return loc, conf, priors

    def load_weights(self, base_file):
        other, ext = os.path.splitext(base_file)
        if ext == '.pkl' or '.pth':
            print('Loading weights into state dict...')
            self.load_state_dict(torch.load(base_file, map_location=lambda storage, loc: storage))
            print('Finished!')
        else:
            print('Sorry only .pth and .pkl files supported.')


def vgg(cfg, i, batch_norm=False):
    layers, cin = [], i
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        elif v == 'C':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    layers += [nn.MaxPool2d(kernel_size=3, stride=1, padding=1),
               nn.Conv2d(512, 1024, kernel_size=3, padding=6, dilation=6), nn.ReLU(inplace=True),
               nn.Conv2d(1024, 1024, kernel_size=1), nn.ReLU(inplace=True)]
    return layers


def add_extras(cfg, i, batch_norm=False):
    layers, cin, flag = [], i, False
    for k, v in enumerate(cfg):
        if cin != 'S':
            if v == 'S':
                layers.append(nn.Conv2d(cin, cfg[k + 1], kernel_size=(1, 3)[flag], stride=2, padding=1))
            else:
                layers.append(nn.Conv2d(cin, v, kernel_size=(1, 3)[flag]))
            flag = not flag
        cin = v
    return layers


def multibox(vgg_layers, extra_layers, cfg, num_classes):
    loc_layers, conf_layers = [], []
    for k, v in enumerate([24, -2]):
        loc_layers.append(nn.Conv2d(vgg_layers[v].out_channels, cfg[k] * 4, kernel_size=3, padding=1))
        conf_layers.append(nn.Conv2d(vgg_layers[v].out_channels, cfg[k] * num_classes, kernel_size=3, padding=1))
    for k, v in enumerate(extra_layers[1::2], 2):
        loc_layers.append(nn.Conv2d(v.out_channels, cfg[k] * 4, kernel_size=3, padding=1))
        conf_layers.append(nn.Conv2d(v.out_channels, cfg[k] * num_classes, kernel_size=3, padding=1))
    return vgg_layers, extra_layers, (loc_layers, conf_layers)


def build_ssd(phase, size=300, num_classes=21):
    if phase != "test" and phase != "train":
        print("Error: Phase not recognized")
        return
    if size != 300:
        print("Error: Sorry only SSD300 is supported currently!")
        return
    return SSD(phase, *multibox(vgg(VGG_CFG, 3), add_extras(EXTRAS_CFG, 1024), MBOX, num_classes), num_classes)
