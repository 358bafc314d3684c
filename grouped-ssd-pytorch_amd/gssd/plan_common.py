"""Constants, switches and the small value types shared by the forward engine's modules (gssd/engine.py and its mixins plan_graph / plan_ops /
plan_exec): the layer tables of models/ssd_multiphase_custom_group.py:434-490, the GSSD_* environment switches, the step / tag records."""
import ctypes as C
import os
import torch
from . import _lib, ops
from ._lib import lib

VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'C', 512, 512, 512, 'M', 512, 512, 512]
EXTRAS_CFG = [256, 'S', 512, 128, 'S', 256, 128, 256, 128, 256]
MBOX = [4, 6, 6, 6, 4, 4]
SRC_HW = [38, 19, 10, 5, 3, 1]
HEAD_OFF = [sum(h * h * a for h, a in zip(SRC_HW[:i], MBOX[:i])) for i in range(6)]      # first prior of source i
FUSE_NAMES = ['11', '21', '31', '41', '51', '61']


# Winograd F(2x2,3x3) for the compute-bound 3x3 trunk layers (csrc/conv_wino.hip); GSSD_NO_WINOGRAD=1 keeps the direct
# implicit GEMM everywhere (ablation / cross-check).
USE_HEADS_WINO = os.environ.get('GSSD_HEADS_WINO', '1') != '0'  # the 38 x 38 multibox head of a train-mode fp32 forward on csrc/conv_wino_x6.hip
USE_PATCH_X6 = os.environ.get('GSSD_PATCH_X6', '1') != '0'    # csrc/conv_patch_x6.hip for the DCN offset conv of a train-mode fp32 forward
USE_CONV_X6 = os.environ.get('GSSD_CONV_X6', '1') != '0'      # csrc/conv_x6.hip for the launches ops.x6_wanted names (fp32 mode)
USE_WINOGRAD = os.environ.get('GSSD_NO_WINOGRAD', '0') != '1'
# fp32 mode: the deformable conv on the bf16 matrix cores with three-plane (fp32-equivalent) operands, csrc/dcn_x6.hip;
# GSSD_DCN_X6=0: the fp32-MFMA kernel csrc/dcn_fused.hip
DCN_X6 = os.environ.get('GSSD_DCN_X6', '1') != '0'
# GSSD_NO_GRAPH=1 keeps every forward an eager list of launches (debugging / ablation)
USE_GRAPH = os.environ.get('GSSD_NO_GRAPH', '0') != '1'
# GSSD_FLASH_X6=0: the fp32-MFMA attention core (csrc/flash_attn.hip) keeps every launch of the fp32 mode (ablation / A-B)
USE_FLASH_X6 = os.environ.get('GSSD_FLASH_X6', '1') != '0'
# GSSD_NO_BRANCH_STREAMS=1 captures the plan as one serial chain (ablation)
USE_BRANCH_STREAMS = os.environ.get('GSSD_NO_BRANCH_STREAMS', '0') != '1'
SN_STREAM = 9               # stream id of the spectral-norm launch inside a captured graph
ALL_STREAMS = -1            # _Step.wait value: join every forked stream before this step

class Tag(tuple):
    """(kernel instance, algorithmic FLOPs, algorithmic bytes) of one launch; ``layer`` names the module it belongs to
    ('vgg.0' = conv1_1 ... 'vgg.40' = conv5_3) so bench.py can sum the trunk's launches -- convs AND their BatchNorm passes."""
    layer = None


class _Step:
    __slots__ = ('fn', 'args', 'keep', 'tag', 'sid', 'wait')

    def __init__(self, fn, args, keep=None, tag=None, sid=0, wait=None):
        # sid: stream id inside a captured graph (0 = trunk);  wait: a stream id whose work this step consumes (joined before it)
        self.fn, self.args, self.keep, self.tag, self.sid, self.wait = fn, args, keep, tag, sid, wait


def conv_tag(d, real_cin_g=None, bf16=False):
    """(kernel instance, algorithmic FLOPs, algorithmic bytes) of one gssd_conv2d launch; the instance name
    mirrors the tile selection in csrc/conv_igemm.hip so it can be matched against rocprofv3's kernel names."""
    cout_g = d.Cout // d.groups
    inst = '128x128' if cout_g > 64 else '128x64' if cout_g > 32 else '128x32' if cout_g > 16 else '128x16'
    if cout_g > 64:       # same wave-quantisation rule as gssd_conv2d_nhwc_f32
        mt = -(-(d.Ho * d.Wo * (1 if d.m_per_image else d.B)) // 128)
        z = d.B if d.m_per_image else d.split_k
        b128 = mt * d.groups * (-(-cout_g // 128)) * z
        b64 = mt * d.groups * (-(-cout_g // 64)) * z
        e128 = b128 / (-(-b128 // 512) * 512)
        e64 = 0.94 * b64 / (-(-b64 // 768) * 768)
        if e64 > e128 or d.K <= 256:
            inst = '128x64'
    # small maps: 32- / 64-row tiles with a three-stage K loop (csrc/conv_igemm.hip, csrc/conv_bf16.hip: the same host rule)
    Ms, Mtot = d.Ho * d.Wo * (1 if d.m_per_image else d.B), d.Ho * d.Wo * d.B
    if (cout_g > 32 and d.split_k == 1 and Mtot <= 4096 and os.environ.get('GSSD_NO_SMALL_TILES') is None
            and not (d.out_mode == _lib.OUT_SPLIT_T and d.split_n % 64 != 0)):
        inst = '32x64' if (Mtot <= 512 or (d.m_per_image and Ms <= 128)) else '64x64'
    if (bf16 and os.environ.get('GSSD_BF16_BIG_TILES', '0') == '1' and not d.m_per_image and d.split_k == 1 and d.B * d.Ho * d.Wo >= 8192
            and cout_g % 128 == 0 and d.K % 64 == 0 and d.K >= 256
            and (d.out_mode == _lib.OUT_NHWC or (d.out_mode == _lib.OUT_SPLIT_T and d.split_n % 128 == 0))):
        inst = '256x128'                                 # csrc/conv_bf16.hip: opt-in experiment (measured slower, round 5)
    name = ('conv_bf16<' if bf16 else 'conv_igemm<') + inst + '>'
    if not bf16 and d.wgt_patch and lib.gssd_conv_patch_x6_takes(C.byref(d)) == 1:      # first in gssd_conv2d_nhwc_f32's dispatch order
        M6 = d.B * d.Ho * d.Wo
        return ('conv_patch_x6<128>', 2.0 * M6 * d.Cout * 9 * d.cin_g, 4.0 * (d.B * d.H * d.W * d.cin_g + M6 * d.Cout + d.Cout * 9 * d.cin_g))
    if not bf16 and d.wgt_x6 and lib.gssd_conv_x6_takes(C.byref(d)) == 1:
        M6 = d.B * d.Ho * d.Wo
        flops = 2.0 * M6 * d.Cout * d.KH * d.KW * d.cin_g
        return (f'conv_x6<{ops.x6_tile(cout_g, d.groups, M6)}>', flops, 4.0 * (d.B * d.H * d.W * d.cin_g * d.groups + M6 * d.Cout + d.Cout * d.KH * d.KW * d.cin_g))
    if bf16:
        if (d.groups == 4 and d.KH == 3 and d.stride == 1 and d.pad == 1 and d.dil == 1 and d.H * d.W >= 75 * 75
                and (d.cin_g, cout_g) in ((8, 16), (16, 16), (16, 32), (32, 32)) and not d.m_per_image and d.split_k == 1
                and d.flags in (0, _lib.CONV_POOL2)):
            name = f'conv_thin_bf16<{d.cin_g},{cout_g}>' + ('/pool2' if d.flags & _lib.CONV_POOL2 else '')   # gssd_try_conv_thin_bf16
    elif (d.groups == 4 and d.KH == 3 and d.stride == 1 and d.pad == 1 and d.dil == 1 and d.H * d.W >= 75 * 75
            and (d.cin_g, cout_g) in ((4, 16), (16, 16), (16, 32)) and not d.m_per_image and d.split_k == 1):
        name = f'conv_thin<{d.cin_g},{cout_g}>'          # gssd_try_conv_thin (csrc/conv_thin.hip)
        if d.wgt_wino and (d.cin_g, cout_g) == (16, 32) and os.environ.get('GSSD_CONV21_WINO', '1') != '0':
            name = 'conv_wino<32>'                       # conv2_1 with Winograd weights: handed on to gssd_try_conv_wino
        if d.wgt_wino and (d.cin_g, cout_g) == (16, 16) and not d.resid:
            name = 'conv_thin_wino<16,16>'               # gssd_try_conv_thin_wino (csrc/conv_thin_wino.hip)
    elif (d.wgt_wino and ops.winograd_eligible(d.KH, d.stride, d.pad, d.dil, d.cin_g, cout_g, d.groups) and not d.m_per_image
          and d.split_k <= 1 and not d.relu):
        name = f'conv_wino<{64 if (cout_g % 64 == 0 or (cout_g % 32 != 0 and cout_g > 32)) else 32}>'   # gssd_try_conv_wino
    igemm_name = ('conv_bf16<' if bf16 else 'conv_igemm<') + inst + '>'
    if name.startswith('conv_wino<') and lib.gssd_conv_wino_x6_takes(C.byref(d)) == 1:
        name = f'conv_wino_x6<{64 if cout_g > 32 else 32}>'      # csrc/conv_wino_x6.hip: the library's own host rule
    elif name.startswith('conv_wino<') and d.out_mode == _lib.OUT_HEADS:
        name = igemm_name                                        # (the fp32 Winograd kernel has no heads epilogue: gssd_try_conv_wino hands it on)
    if not bf16 and lib.gssd_conv_thin_x6_takes(C.byref(d)) == 1:
        name = f'conv_thin_x6<{d.cin_g},{cout_g}>'               # csrc/conv_thin_x6.hip: first in gssd_conv2d_nhwc_f32's dispatch order
    if name.startswith(('conv_wino<', 'conv_wino_x6<', 'conv_thin_x6<')):
        # one name per kernel SYMBOL, as rocprofv3 --stats groups them (template <tile, fused input transform, ..., pooled epilogue>)
        name += ('' if d.in_scale else '/plain') + ('/pool2' if d.flags & _lib.CONV_POOL2 else '')
    if bf16 and name.startswith('conv_bf16'):
        bm = lib.gssd_conv_flat_bf16_takes(C.byref(d))      # csrc/conv_flat_bf16.hip: the library's own host rule
        if bm:
            name = f'conv_flat_bf16<{d.cin_g},{min(cout_g, 128) if cout_g % 128 == 0 else 64},{bm}>'
    if not bf16 and name.startswith('conv_igemm') and lib.gssd_gemm_slot_takes(C.byref(d)) == 1:
        name = 'gemm_slot<128x128>'                      # gssd_try_gemm_slot (csrc/gemm_slot.hip): the library's own host rule
    M = d.B * d.Ho * d.Wo
    cin_g = real_cin_g if real_cin_g is not None else d.cin_g
    flops = 2.0 * M * d.Cout * d.KH * d.KW * cin_g
    esz = 2.0 if bf16 else 4.0
    out_elems = M * d.Cout // 4 if (d.flags & _lib.CONV_POOL2) else M * d.Cout       # pooled raw output: a quarter of the map
    byts = esz * (d.B * d.H * d.W * cin_g * d.groups + out_elems + d.Cout * d.KH * d.KW * cin_g)
    return (name, flops, byts)
