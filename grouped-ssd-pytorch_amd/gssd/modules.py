"""Parameter containers that mirror the reference's module tree (names, shapes, state-dict keys)
so checkpoints load strictly in both directions.  They hold weights only; the arithmetic runs in
the HIP engine (``gssd/engine.py``).

Key compatibility (probed against the reference):
  * ``Self_Attn``: ``snconv1x1_{theta,phi,g,attn}.{bias,weight_orig,weight_u,weight_v}``, ``sigma``
    (layers/self_attn.py:29-44 + layers/spectral_norm.py:126-139)
  * ``DCN``: ``weight``, ``bias``, ``conv_offset_mask.{weight,bias}`` (layers/dcn_v2_custom.py:18-77)
  * ``L2Norm``: ``weight`` (layers/modules/l2norm.py:7-17)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init


class L2Norm(nn.Module):
    """layers/modules/l2norm.py: learnable per-channel scale (init ``scale``), eps 1e-10 after sqrt."""

    def __init__(self, n_channels, scale):
        super().__init__()
        self.n_channels = n_channels
        self.gamma = scale or None
        self.eps = 1e-10
        self.weight = nn.Parameter(torch.empty(n_channels))
        init.constant_(self.weight, self.gamma)

    def forward(self, x):
        """NCHW in / NCHW out, HIP kernel underneath."""
        from . import ops
        xh = x.permute(0, 2, 3, 1).contiguous()
        return ops.l2norm(xh, self.weight.detach(), self.eps).permute(0, 3, 1, 2)


class SNConv1x1(nn.Module):
    """A spectral-normed 1x1 conv's state: ``weight_orig``, ``bias`` (parameters), ``weight_u``,
    ``weight_v`` (buffers, unit vectors) -- layers/spectral_norm.py:111-139."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        w = torch.empty(out_channels, in_channels, 1, 1)
        init.kaiming_uniform_(w, a=math.sqrt(5))            # nn.Conv2d default
        self.weight_orig = nn.Parameter(w)
        bound = 1.0 / math.sqrt(in_channels)
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        self.register_buffer('weight_u', F.normalize(torch.randn(out_channels), dim=0, eps=1e-12))
        self.register_buffer('weight_v', F.normalize(torch.randn(in_channels), dim=0, eps=1e-12))


class Self_Attn(nn.Module):
    """layers/self_attn.py:29-89 (state only; ``GssdEngine`` runs it)."""

    def __init__(self, in_channels, max_pool_factor=1):
        super().__init__()
        self.in_channels = in_channels
        self.snconv1x1_theta = SNConv1x1(in_channels, in_channels // 8)
        self.snconv1x1_phi = SNConv1x1(in_channels, in_channels // 8)
        self.snconv1x1_g = SNConv1x1(in_channels, in_channels // 2)
        self.snconv1x1_attn = SNConv1x1(in_channels // 2, in_channels)
        self.sigma = nn.Parameter(torch.zeros(1))
        self.max_pool_factor = max_pool_factor


class DCN(nn.Module):
    """layers/dcn_v2_custom.py:18-89 (state only)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        if (kernel_size, stride, padding, dilation) != (3, 1, 1, 1):
            raise NotImplementedError('the path only uses 3x3 / stride 1 / pad 1 DCN (models/...group.py:170-179)')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = (3, 3), (1, 1), (1, 1), (1, 1)
        self.deformable_groups = deformable_groups
        n = in_channels * 9
        stdv = 1.0 / math.sqrt(n)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3).uniform_(-stdv, stdv))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.conv_offset_mask = nn.Conv2d(in_channels, deformable_groups * 27, kernel_size=3, stride=1, padding=1,
                                          bias=True)
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()
