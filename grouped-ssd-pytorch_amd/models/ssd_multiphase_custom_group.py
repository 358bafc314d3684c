"""Drop-in for ``ssd_liverdet/models/ssd_multiphase_custom_group.py`` (GSSD / GSSD++) on MI355X.

Same constructor (``build_ssd``, 15 positional args as called from
train_lesion_multiphase_v2.py:142-145), same module attribute names and state-dict keys, same
forward contract (train tuple / Detect output / ``visualize=True`` extras) -- but ``forward`` runs
the hand-written HIP kernels through ``gssd.engine.GssdEngine`` instead of ATen/cuDNN ops.
The ``nn.Conv2d`` / ``nn.BatchNorm2d`` objects below are parameter containers only.
"""
import os

import torch
import torch.nn as nn

from gssd.engine import GssdEngine, VGG_CFG, EXTRAS_CFG, MBOX
from gssd.modules import DCN, L2Norm, Self_Attn
from gssd import ops
from layers.functions.prior_box import PriorBox
from layers.functions.detection import Detect
from data.config import v2


def xavier(param):
    nn.init.xavier_uniform_(param)


def weights_init(m):
    if isinstance(m, nn.Conv2d):
        xavier(m.weight.data)
        m.bias.data.zero_()


class SSD(nn.Module):
    """Reference: models/ssd_multiphase_custom_group.py:23-430."""

    def __init__(self, phase, base, extras, head, num_classes, batch_norm, groups_vgg, groups_extra, feature_scale,
                 use_fuseconv, use_self_attention, use_self_attention_base, num_dcn_layers, groups_dcn, dcn_cat_sab,
                 detach_sab, max_pool_factor):
        super().__init__()
        if groups_vgg not in (1, 2, 4) or groups_extra not in (1, 2, 4):
            raise ValueError('groups_vgg / groups_extra must divide the 12 input channels and every layer width: 1, 2 or 4 '
                             '(4 = one group per CT phase is what the tuned kernels are laid out for; 1 and 2 run the generic ones)')
        self.phase = phase
        self.num_classes = num_classes
        self.batch_norm = batch_norm
        self.priorbox = PriorBox(v2)
        self.priors = self.priorbox.forward()
        self.priors.requires_grad = False
        self.size = 300
        self.groups_vgg, self.groups_extra, self.feature_scale = groups_vgg, groups_extra, feature_scale
        self.use_fuseconv = use_fuseconv
        self.use_self_attention = use_self_attention
        self.use_self_attention_base = use_self_attention_base
        self.num_dcn_layers = num_dcn_layers

        self.vgg = nn.ModuleList(base)
        self.L2Norm = L2Norm(512 * feature_scale, 20)
        self.extras = nn.ModuleList(extras)
        self.loc = nn.ModuleList(head[0])
        self.conf = nn.ModuleList(head[1])
        if phase == 'test':
            self.softmax = nn.Softmax(dim=-1)
            self.detect = Detect      # new-style autograd.Function: used via .apply (reference :75,:384)

        fs = feature_scale
        if use_fuseconv:                                   # reference :79-139 (fuse convs only with use_fuseconv, their BN only with
            for name, ch in (('11', 512), ('21', 1024), ('31', 512), ('41', 256), ('51', 256), ('61', 256)):   # batch_norm)
                conv = nn.Conv2d(ch * fs, ch * fs, kernel_size=1)
                conv.apply(weights_init)
                setattr(self, f'fuse_{name}', conv)
                if batch_norm:
                    setattr(self, f'bn_fuse_{name}', nn.BatchNorm2d(ch * fs))
            # the same modules registered a second time, like the reference (:135-139) -> duplicate state-dict keys
            self.fuse_list1 = nn.ModuleList([self.fuse_31, self.fuse_41, self.fuse_51, self.fuse_61])
            if batch_norm:
                self.bn_fuse_list1 = nn.ModuleList([self.bn_fuse_31, self.bn_fuse_41, self.bn_fuse_51, self.bn_fuse_61])

        self.max_pool_factor = max_pool_factor
        chans = [c * fs for c in (512, 1024, 512, 256, 256, 256)]
        if use_self_attention:
            self.self_attn_list = nn.ModuleList([Self_Attn(c, max_pool_factor) for c in chans])
        if use_self_attention_base:
            self.self_attn_base_list = nn.ModuleList([Self_Attn(c, max_pool_factor) for c in chans])
        if num_dcn_layers > 0:
            self.use_dcn = True
            self.groups_dcn = groups_dcn
            self.dcn_cat_sab = dcn_cat_sab
            self.detach_sab = detach_sab
            if detach_sab:
                assert dcn_cat_sab is True, "deatch_sab requires --dcn_cat_sab=True"
            layers = []
            if dcn_cat_sab:
                assert use_self_attention_base is True, "dcn_cat_sab requires use_self_attention_base=True"
                layers.append(DCN(1024 * fs, 512 * fs, 3, 1, 1, deformable_groups=groups_dcn))
            else:
                layers.append(DCN(512 * fs, 512 * fs, 3, 1, 1, deformable_groups=groups_dcn))
            for _ in range(num_dcn_layers - 1):
                layers.append(DCN(512 * fs, 512 * fs, 3, 1, 1, deformable_groups=groups_dcn))
            self.dcn_list = nn.ModuleList(layers)
        else:
            self.use_dcn = False
            self.dcn_cat_sab = False
            self.detach_sab = False
        object.__setattr__(self, '_engine', GssdEngine(self))

    # nn.Module plumbing that must not drag the engine along
    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == '_engine':
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        object.__setattr__(new, '_engine', GssdEngine(new))
        return new

    def _replicate_for_data_parallel(self):
        # nn.DataParallel over ONE device calls the module directly (no replicas) and works; replicas on several devices are
        # parameter-less views driven from one thread per GPU -- this path runs one PROCESS per GPU instead (gssd.dist,
        # bench.py --gpus N: per-rank BatchNorm statistics exactly like DataParallel's replicas, RCCL gradient all-reduce)
        from gssd._lib import GssdError
        raise GssdError('multi-device nn.DataParallel replicas are not supported by the HIP engine: launch one process per GPU '
                        '(python -m torch.distributed.run --nproc-per-node N ..., gssd.dist.allreduce_grads / '
                        'OverlappedGradReducer average the gradients over RCCL)')

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self.priors = fn(self.priors)
        self._engine.invalidate()
        return out

    def slice_and_cat(self, a, b):
        """NCHW in / NCHW out (reference :185-192), HIP kernel underneath."""
        ah, bh = a.permute(0, 2, 3, 1).contiguous(), b.permute(0, 2, 3, 1).contiguous()
        return ops.slice_and_cat(ah, bh, self.groups_vgg).permute(0, 3, 1, 2)

    def forward(self, x, visualize=False):
        self.__dict__['_want_maps'] = bool(visualize)      # visualize=True also materialises the attention maps
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # HIP forward plan + HIP backward plan (gssd/autograd.py); an eval-mode forward carries no autograd graph
            from gssd.autograd import GssdTrainFn
            loc, conf = GssdTrainFn.apply(self, x, *tuple(self.parameters()))
        else:
            loc, conf = self._engine.forward(x, self.training, self.__dict__.get('_events'), bool(visualize))
        priors = self.priors if self.priors.device == x.device else self.priors.to(x.device)
        if self.phase == 'test':
            # softmax (reference :388) is fused into the Detect kernel (conf_is_logits)
            output = self.detect.apply(self.num_classes, 0, 200, 0.01, 0.45, loc, conf, priors, True)
        else:
            output = (loc, conf, priors)
        if visualize:
            return (output,) + self._engine_visuals(x.shape[0])
        return output

    def _engine_visuals(self, B):
        """all_offset / all_attnb / all_attn of the reference's visualize=True return (:397-398), as NCHW /
        [B,N,N] torch tensors."""
        plan = self._engine._last_plan
        offs = []
        for om, H, dg in getattr(plan, 'offsets', []):
            offs.append(ops.unpack_nhwc(om, 18 * dg))
        maps = getattr(plan, 'attn_maps', {})

        def collect(name):
            lst = []
            for (n, i), (S, N, Np) in sorted(maps.items(), key=lambda kv: kv[0][1]):
                if n == name:
                    lst.append(S[:, :, :N].clone())
            return lst
        return offs, collect('self_attn_base_list'), collect('self_attn_list')

    def load_weights(self, base_file):
        """Tolerant loader of the reference (:402-429): strips ``module.``, skips shape mismatches."""
        other, ext = os.path.splitext(base_file)
        if ext == '.pkl' or '.pth':
            print('Loading weights into state dict...')
            pre = torch.load(base_file, map_location=lambda storage, loc: storage)
            model_dict = self.state_dict()
            picked = {}
            for k, v in pre.items():
                name = k[7:] if k.startswith('module.') else k
                if name in model_dict and model_dict[name].shape == v.shape:
                    picked[name] = v
                elif name in model_dict:
                    print('WARNING: shape of pretrained {} {} does not match the current model {}. this weight will '
                          'be ignored.'.format(name, v.shape, model_dict[name].shape))
            model_dict.update(picked)
            self.load_state_dict(model_dict)
        else:
            print('Sorry only .pth and .pkl files supported.')


def vgg(cfg, i, batch_norm=False, feature_scale=1, groups_vgg=4):
    layers, cin = [], i
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        elif v == 'C':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True))
        else:
            layers.append(nn.Conv2d(cin, v * feature_scale, kernel_size=3, padding=1, groups=groups_vgg))
            if batch_norm:
                layers.append(nn.BatchNorm2d(v * feature_scale))
            layers.append(nn.ReLU(inplace=True))
            cin = v * feature_scale
    layers.append(nn.MaxPool2d(kernel_size=3, stride=1, padding=1))
    for conv in (nn.Conv2d(512 * feature_scale, 1024 * feature_scale, kernel_size=3, padding=6, dilation=6,
                           groups=groups_vgg),
                 nn.Conv2d(1024 * feature_scale, 1024 * feature_scale, kernel_size=1, groups=groups_vgg)):
        layers.append(conv)
        if batch_norm:
            layers.append(nn.BatchNorm2d(1024 * feature_scale))
        layers.append(nn.ReLU(inplace=True))
    return layers


def add_extras(cfg, i, batch_norm=False, feature_scale=1, groups_extra=4):
    layers, cin, flag = [], i, False
    for k, v in enumerate(cfg):
        if cin != 'S':
            if v == 'S':
                cout = cfg[k + 1] * feature_scale
                layers.append(nn.Conv2d(cin, cout, kernel_size=(1, 3)[flag], stride=2, padding=1, groups=groups_extra))
            else:
                cout = v * feature_scale
                layers.append(nn.Conv2d(cin, cout, kernel_size=(1, 3)[flag], groups=groups_extra))
            if batch_norm:
                layers.append(nn.BatchNorm2d(cout))
            flag = not flag
        cin = v if v == 'S' else v * feature_scale
    return layers


def multibox(vgg_layers, extra_layers, cfg, num_classes, batch_norm):
    loc_layers, conf_layers = [], []
    vgg_source = [30, -3] if batch_norm else [21, -2]
    for k, v in enumerate(vgg_source):
        loc_layers.append(nn.Conv2d(vgg_layers[v].out_channels, cfg[k] * 4, kernel_size=3, padding=1))
        conf_layers.append(nn.Conv2d(vgg_layers[v].out_channels, cfg[k] * num_classes, kernel_size=3, padding=1))
    for k, v in enumerate(extra_layers[2::4] if batch_norm else extra_layers[1::2], 2):
        loc_layers.append(nn.Conv2d(v.out_channels, cfg[k] * 4, kernel_size=3, padding=1))
        conf_layers.append(nn.Conv2d(v.out_channels, cfg[k] * num_classes, kernel_size=3, padding=1))
    return vgg_layers, extra_layers, (loc_layers, conf_layers)


base = {'300': VGG_CFG, '512': []}
extras = {'300': EXTRAS_CFG, '512': []}
mbox = {'300': MBOX, '512': []}


def build_ssd(phase, size=300, num_classes=21, batch_norm=False, groups_vgg=4, groups_extra=4, feature_scale=1,
              use_fuseconv=True, use_self_attention=False, use_self_attention_base=False, num_dcn_layers=0,
              groups_dcn=1, dcn_cat_sab=False, detach_sab=False, max_pool_factor=1):
    if phase != "test" and phase != "train":
        print("Error: Phase not recognized")
        return
    if size != 300:
        print("Error: Sorry only SSD300 is supported currently!")
        return
    return SSD(phase, *multibox(vgg(base[str(size)], 12, batch_norm, feature_scale, groups_vgg),
                                add_extras(extras[str(size)], 1024 * feature_scale, batch_norm, feature_scale,
                                           groups_extra),
                                mbox[str(size)], num_classes, batch_norm),
               num_classes, batch_norm, groups_vgg, groups_extra, feature_scale, use_fuseconv, use_self_attention,
               use_self_attention_base, num_dcn_layers, groups_dcn, dcn_cat_sab, detach_sab, max_pool_factor)
