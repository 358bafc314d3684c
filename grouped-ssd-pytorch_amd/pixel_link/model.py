"""``PixelLink`` (drop-in for ssd_liverdet/pixel_link/model.py:20-188: same constructor, attribute names and state-dict keys --
including the ``modules_except_dcn`` aliases -- so reference checkpoints load); ``forward`` runs the HIP launch plan of
gssd/pixellink.py (a grad-enabled call is differentiable: the HIP backward plan follows) and returns ``[out_1 [B,2,75,75], out_2 [B,16,75,75]]`` like model.py:413."""
import os

import torch
import torch.nn as nn

import pixel_link.pixel_link_config as config
from gssd import _lib
from gssd.modules import DCN, Self_Attn

_TRUNK = (('1_1', 12, 64), ('1_2', 64, 64), 'pool1', ('2_1', 64, 128), ('2_2', 128, 128), 'pool2', ('3_1', 128, 256),
          ('3_2', 256, 256), ('3_3', 256, 256), 'pool3', ('4_1', 256, 512), ('4_2', 512, 512), ('4_3', 512, 512), 'pool4',
          ('5_1', 512, 512), ('5_2', 512, 512), ('5_3', 512, 512), 'pool5')
_STAGE_CH = {2: 256, 3: 512, 4: 512, 5: 1024}       # channels of the four output stages (conv3_3, conv4_3, conv5_3, fc7)


def weights_init(m):                                  # model.py:14-17
    if isinstance(m, nn.Conv2d):
        nn.init.xavier_uniform_(m.weight.data)
        m.bias.data.zero_()


class PixelLink(nn.Module):
    def __init__(self, cascade_fuse, use_fuseconv, batch_norm, use_self_attention, use_self_attention_base, num_dcn_layers,
                 groups_dcn, dcn_cat_sab, detach_sab, max_pool_factor=1):
        super().__init__()
        if config.version != "4s" or config.feature_scale != 1 or not config.dilation:
            raise NotImplementedError('the HIP PixelLink++ path is built for pixel_link_config version "4s", feature_scale 1, '
                                      'dilation True (the reference defaults)')
        g = self.vgg_groups = config.vgg_groups
        self.scale = config.feature_scale
        self.cascade_fuse, self.use_fuseconv, self.batch_norm = cascade_fuse, use_fuseconv, batch_norm
        self.use_self_attention, self.use_self_attention_base = use_self_attention, use_self_attention_base
        self.num_dcn_layers, self.max_pool_factor = num_dcn_layers, max_pool_factor
        trunk = []
        for t in _TRUNK:
            if isinstance(t, str):
                pool = nn.MaxPool2d(kernel_size=[3, 3], stride=1, padding=1, ceil_mode=True) if t == 'pool5' else \
                    nn.MaxPool2d(2, ceil_mode=True)
                setattr(self, t, pool)
                trunk.append(pool)
            else:
                n, ci, co = t
                setattr(self, 'conv' + n, nn.Conv2d(ci, co, 3, stride=1, padding=1, groups=g))
                setattr(self, 'relu' + n, nn.ReLU())
                trunk += [getattr(self, 'conv' + n), getattr(self, 'relu' + n)]
        self.conv6 = nn.Conv2d(512, 1024, 3, stride=1, padding=6, dilation=6, groups=g)
        self.relu6 = nn.ReLU()
        self.conv7 = nn.Conv2d(1024, 1024, 1, stride=1, padding=0, groups=g)
        self.relu7 = nn.ReLU()
        self.modules_except_dcn = nn.ModuleList(trunk + [self.conv6, self.relu6, self.conv7, self.relu7])
        for k, c in _STAGE_CH.items():
            setattr(self, f'out{k}_1', nn.Conv2d(c, 2, 1))
            setattr(self, f'out{k}_2', nn.Conv2d(c, 16, 1))
            self.modules_except_dcn.extend([getattr(self, f'out{k}_1'), getattr(self, f'out{k}_2')])
        if use_fuseconv:
            for k, c in _STAGE_CH.items():
                setattr(self, f'fuse{k}', nn.Conv2d(c, c, kernel_size=1))
            self.modules_except_dcn.extend([getattr(self, f'fuse{k}') for k in _STAGE_CH])
            if batch_norm:
                for k, c in _STAGE_CH.items():
                    setattr(self, f'bn_fuse{k}', nn.BatchNorm2d(c))
                self.modules_except_dcn.extend([getattr(self, f'bn_fuse{k}') for k in _STAGE_CH])
        nf = 4 if cascade_fuse else 1
        self.final_1 = nn.Conv2d(2 * nf, 2, 1)
        self.final_2 = nn.Conv2d(16 * nf, 16, 1)
        self.modules_except_dcn.extend([self.final_1, self.final_2])
        chans = list(_STAGE_CH.values())
        if use_self_attention_base:
            self.self_attn_base_in_channel_list = chans
            self.self_attn_base_list = nn.ModuleList([Self_Attn(c, max_pool_factor=max_pool_factor) for c in chans])
        if use_self_attention:
            self.self_attn_in_channel_list = chans
            self.self_attn_list = nn.ModuleList([Self_Attn(c, max_pool_factor=max_pool_factor) for c in chans])
        self.use_dcn = num_dcn_layers > 0
        self.dcn_cat_sab = bool(dcn_cat_sab) and self.use_dcn
        self.detach_sab = bool(detach_sab) and self.use_dcn
        if self.use_dcn:
            self.groups_dcn = groups_dcn
            if self.detach_sab:
                assert self.dcn_cat_sab is True, "deatch_sab requires --dcn_cat_sab=True"
            if self.dcn_cat_sab:
                assert use_self_attention_base is True, "dcn_cat_sab requires use_self_attention_base=True"
            self.dcn_in_channel_list = [256]
            self.dcn_list = nn.ModuleList([DCN(256 * (2 if (self.dcn_cat_sab and j == 0) else 1), 256, kernel_size=3, stride=1,
                                               padding=1, deformable_groups=groups_dcn) for j in range(num_dcn_layers)])
        for m in self.modules():
            weights_init(m)
        for m in self.modules():                       # the reference's spectral-norm convs are nn.Conv2d: weights_init zeroes their bias
            if isinstance(m, Self_Attn):
                for c in (m.snconv1x1_theta, m.snconv1x1_phi, m.snconv1x1_g, m.snconv1x1_attn):
                    c.bias.data.zero_()
        self.__dict__['_engine'] = None

    def forward(self, x):
        if self.__dict__.get('_engine') is None:
            from gssd.pixellink import PixelLinkEngine
            self.__dict__['_engine'] = PixelLinkEngine(self)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from gssd.autograd import PixelLinkTrainFn             # HIP forward plan + HIP backward plan (gssd/backward.py)
            if not self.training:
                raise _lib.GssdError('a grad-enabled PixelLink++ forward needs train mode (the backward differentiates the batch-statistics '
                                     'BatchNorm); call .train() or torch.no_grad()')
            out_1, out_2 = PixelLinkTrainFn.apply(self, x, *self.parameters())
            return [out_1, out_2]
        out_1, out_2 = self.__dict__['_engine'].forward(x, self.training)
        return [out_1, out_2]

    def _replicate_for_data_parallel(self):
        raise _lib.GssdError('one process per GPU (torch.distributed), not nn.DataParallel replicas')

    def load_weights(self, base_file):                 # model.py:420-449: shape-filtered load, "module." prefixes dropped
        if os.path.splitext(base_file)[1] not in ('.pkl', '.pth'):
            print('Sorry only .pth and .pkl files supported.')
            return
        print('Loading weights into state dict...')
        own = self.state_dict()
        for k, v in torch.load(base_file, map_location='cpu').items():
            k = k[7:] if k.startswith('module.') else k
            if k in own:
                if v.shape == own[k].shape:
                    own[k] = v
                else:
                    print(f'WARNING: shape of pretrained {k} {tuple(v.shape)} does not match the current model '
                          f'{tuple(own[k].shape)}. this weight will be ignored.')
        self.load_state_dict(own)
