"""Switches and constants of the HIP backward plan (gssd/backward.py and its modules bwd_ops / bwd_shadow)."""
import ctypes as C
import os
import torch
from . import _lib, ops
from ._lib import lib

BWD_STREAMS = os.environ.get('GSSD_BWD_STREAMS', '1') != '0'
# GSSD_BWD_GRAPH=1: the backward plan replays from hipGraphs from its third run on (BackwardPlan._execute).  OFF by default: measured on
# one MI355X (round 4, GSSD++ B = 32, 16 timed steps) the graph replay of the ~500-node, 8-stream backward is SLOWER than the eager
# launches -- 52.9 against 48.9 ms per training step -- and at one rank the host is not the bottleneck (it enqueues a step in ~20 ms).
# It is the form to try when 8 ranks share one host's cores (gloo test with 8 ranks on one GPU: tests/test_gpu_multi.py).
USE_BWD_GRAPH = os.environ.get('GSSD_BWD_GRAPH', '0') == '1' and os.environ.get('GSSD_NO_GRAPH', '0') != '1'
# bf16 storage mode: the data-gradient convs / GEMMs (NT form: d(input) = d(output) * W) run on the bf16 matrix cores -- d(output) and the
# packed weight rounded to bf16 once per launch, fp32 accumulation, fp32 gradient maps (round 4).  GSSD_BWD_BF16=0
# keeps the fp32 kernels on fp32 copies.
BWD_BF16 = os.environ.get('GSSD_BWD_BF16', '1') != '0'
LEAF_SID = 1000
HOIST_FROM = 1       # first branch stream id whose backward is hoisted (1 = all six; hoisting block 0 as well: GSSD 25.5 -> 23.3 ms)
N_LEAF = 1           # leaf streams, taken in turn by the layers (measured: 1 -> 50.3 ms, 2 -> 50.8, 3 -> 51.2, 4 -> 52.1)


def _leaf_fns():
    return (lib.gssd_conv2d_wgrad_f32, lib.gssd_unpack_conv_weight_grad, lib.gssd_cast_f64_f32, lib.gssd_colsum_f32,
            lib.gssd_sn_weight_grad_f32, lib.gssd_scale_cast_f64_f32, lib.gssd_sa_sigma_grad_f32, lib.gssd_dot_f32,
            lib.gssd_dcn_im2col_f32)
