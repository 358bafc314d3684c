"""ctypes binding of libgssd_hip.so (the C ABI declared in include/gssd_hip.h).

The library is built in-tree by ``make -C grouped-ssd-pytorch_amd/gssd/csrc`` (or
``__graft_entry__.build()``).  There is NO fallback: if the shared object is missing or does
not export every symbol of the header, importing this module raises.
"""
import ctypes as C
import os

# torch must load ITS HIP runtime (libamdhip64 bundled under torch/lib) before libgssd_hip.so is opened: if the
# system copy under /opt/rocm is pulled in first the process ends up with two runtimes and kernels launched from this
# library fail with "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSSD_LIB_PATH: an alternate build of the SAME library (debug / timing / A-B builds of scripts/*_timing.py, scripts/ab_lib.py)
LIB_PATH = os.environ.get('GSSD_LIB_PATH') or os.path.join(_HERE, 'lib', 'libgssd_hip.so')

c_fp = C.c_void_p      # device pointers travel as integers (tensor.data_ptr())
c_i = C.c_int
c_f = C.c_float
c_d = C.c_double
c_i64 = C.c_int64


class ConvDesc(C.Structure):
    """struct gssd_conv_desc (include/gssd_hip.h)."""
    _fields_ = [
        ('in_', c_fp), ('wgt', c_fp), ('bias', c_fp), ('out', c_fp), ('out_b', c_fp), ('alpha', c_fp),
        ('gate', c_fp), ('resid', c_fp), ('out2', c_fp), ('in_scale', c_fp), ('in_shift', c_fp), ('in_pad', c_fp),
        ('stats', c_fp), ('wgt_wino', c_fp), ('pool_sign', c_fp), ('wgt_x6', c_fp), ('wgt_patch', c_fp),
        ('B', c_i), ('H', c_i), ('W', c_i), ('in_stride', c_i), ('in_ch_off', c_i), ('Ho', c_i), ('Wo', c_i),
        ('Cout', c_i), ('groups', c_i), ('cin_g', c_i), ('KH', c_i), ('KW', c_i), ('stride', c_i), ('pad', c_i),
        ('dil', c_i), ('K', c_i), ('wgt_row_stride', c_i), ('out_stride', c_i), ('out_ch_off', c_i),
        ('out_mode', c_i), ('relu', c_i), ('m_per_image', c_i), ('split_n', c_i), ('split_k', c_i),
        ('out_b_stride', c_i), ('flags', c_i), ('stats_rep', c_i),
        ('in_batch_stride', c_i64), ('wgt_batch_stride', c_i64), ('out_batch_stride', c_i64),
        ('outb_batch_stride', c_i64), ('out_off', c_i64), ('outb_off', c_i64),
    ]


class PlanOp(C.Structure):
    """struct gssd_plan_op (csrc/plan_run.hip): one LAUNCH / WAIT of a launch-plan segment."""
    _fields_ = [('kind', c_i), ('fn', c_i), ('stream', c_i), ('nargs', c_i), ('args', C.c_uint64 * 24)]


PLAN_LAUNCH, PLAN_WAIT = 0, 1


class SnItem(C.Structure):
    """struct gssd_sn_item."""
    _fields_ = [('w', c_fp), ('u', c_fp), ('v', c_fp), ('inv_sigma', c_fp), ('rows', c_i), ('cols', c_i)]


OUT_NHWC, OUT_TRANSPOSED, OUT_HEADS, OUT_SPLIT_T = 0, 1, 2, 3
CONV_OUT_F32, CONV_OUTB_BF16_PERM32, CONV_HEADS_SLICES, CONV_POOL2, CONV_RESID_F32, CONV_F16_OK = 1, 2, 4, 8, 16, 32

# name -> (restype, argtypes); mirrors include/gssd_hip.h one to one
SIGNATURES = {
    'gssd_abi_version': (c_i, []),
    'gssd_event_create': (c_i, [C.POINTER(C.c_void_p)]),
    'gssd_event_destroy': (c_i, [c_fp]),
    'gssd_event_record_node': (c_i, [c_fp, c_fp]),
    'gssd_event_elapsed_ms': (c_i, [c_fp, c_fp, C.POINTER(C.c_float)]),
    'gssd_conv_desc_size': (c_i, []),
    'gssd_last_error': (C.c_char_p, []),
    'gssd_build_arch': (C.c_char_p, []),
    'gssd_pack_input_nhwc': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_unpack_nhwc_to_nchw': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_pack_conv_weight': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_conv2d_nhwc_f32': (c_i, [C.POINTER(ConvDesc), c_fp]),
    'gssd_conv2d_nhwc_bf16': (c_i, [C.POINTER(ConvDesc), c_fp]),
    'gssd_conv_x6_tile': (c_i, [c_i, c_i, c_i64]),
    'gssd_conv_x6_weight_elems': (c_i64, [c_i, c_i, c_i, c_i, c_i]),
    'gssd_conv_x6_pack_weight': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_conv_x6_takes': (c_i, [C.POINTER(ConvDesc)]),
    'gssd_conv_wino_x6_takes': (c_i, [C.POINTER(ConvDesc)]),
    'gssd_conv_thin_x6_takes': (c_i, [C.POINTER(ConvDesc)]),
    'gssd_plan_fn_count': (c_i, []),
    'gssd_plan_fn_name': (C.c_char_p, [c_i]),
    'gssd_plan_fn_index': (c_i, [C.c_char_p]),
    'gssd_plan_fn_nargs': (c_i, [c_i]),
    'gssd_plan_op_size': (c_i, []),
    'gssd_plan_run': (c_i, [C.POINTER(PlanOp), c_i, C.POINTER(c_fp), c_i, C.POINTER(c_i)]),
    'gssd_pack_conv_weight_bf16': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_cast_f32_bf16': (c_i, [c_fp, c_fp, c_i64, c_fp]),
    'gssd_cast_rows_f32_bf16': (c_i, [c_fp, c_fp, c_i64, c_i, c_i, c_i, c_fp]),
    'gssd_transpose_cast_f32_bf16': (c_i, [c_fp, c_fp, c_i64, c_i, c_i64, c_i64, c_fp]),
    'gssd_cast_bf16_f32': (c_i, [c_fp, c_fp, c_i64, c_fp]),
    'gssd_pack_input_nhwc_bf16': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_bn_relu_pool_bf16': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_d, c_fp, c_fp,
                                     c_fp, c_fp, c_f, c_f, c_i, c_i, c_i, c_fp]),
    'gssd_bn_finalize_bf16': (c_i, [c_fp, c_d, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp]),
    'gssd_l2norm_bf16': (c_i, [c_fp, c_fp, c_fp, c_i64, c_i, c_f, c_fp]),
    'gssd_dcn_packed_weight_elems_bf16': (C.c_longlong, [c_i, c_i]),
    'gssd_dcn_pack_weight_bf16': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    'gssd_dcn_forward_bf16': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_winograd_weight_elems': (C.c_longlong, [c_i, c_i, c_i]),
    'gssd_winograd_weight_f32': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_conv2d_wgrad_f32': (c_i, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp]),
    'gssd_conv2d_wgrad_bf16': (c_i, [C.POINTER(ConvDesc), c_fp, c_fp, c_fp]),
    'gssd_conv2d_wgrad_bf16_supported': (c_i, [C.POINTER(ConvDesc)]),
    'gssd_unpack_conv_weight_grad': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_pack_conv_weight_dgrad': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_bn_relu_pool_f32': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_d, c_fp, c_fp,
                                    c_fp, c_fp, c_f, c_f, c_i, c_i, c_i, c_fp]),
    'gssd_bn_finalize_f32': (c_i, [c_fp, c_d, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp]),
    'gssd_bn_bwd_reduce_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_bn_bwd_finalize_f32': (c_i, [c_fp, c_d, c_fp, c_fp, c_f, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_fp]),
    'gssd_bn_bwd_apply_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i, c_fp, c_fp]),
    'gssd_bn_bwd_reduce_mixed': (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_bn_bwd_apply_mixed': (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_i64, c_i, c_fp, c_i, c_fp]),
    'gssd_bn_bwd_apply_masked_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_i64, c_i, c_fp, c_fp]),
    'gssd_colsum_f32': (c_i, [c_fp, c_i64, c_i, c_i, c_fp, c_fp]),
    'gssd_cast_f64_f32': (c_i, [c_fp, c_fp, c_i, c_i, c_fp]),
    'gssd_l2norm_bwd_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i, c_f, c_fp]),
    'gssd_heads_gather_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_upsample_insert_f32': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_l2norm_f32': (c_i, [c_fp, c_fp, c_fp, c_i64, c_i, c_f, c_fp]),
    'gssd_self_attn_core_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_self_attn_core_kv_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    'gssd_sa_pool_kv_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_sa_unpool_f32': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_pack_conv_weights_batched': (c_i, [c_fp, c_i, c_fp]),
    'gssd_gemm_slot_takes': (c_i, [c_fp]),
    'gssd_conv_flat_bf16_takes': (c_i, [c_fp]),
    'gssd_conv_flat_bf16_tile': (c_i, [c_i]),
    'gssd_heads_reduce_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    'gssd_interp_add_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_pixellink_final_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp]),
    'gssd_pixellink_loss_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_interp_add_bwd_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_pixellink_final_bwd_f32': (c_i, [c_fp] * 6 + [c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    'gssd_pixellink_loss_bwd_f32': (c_i, [c_fp] * 10 + [c_i, c_i, c_i, c_fp]),
    'gssd_pixellink_decode_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, C.c_float, C.c_float, c_i, c_fp]),
    'gssd_self_attn_flash_bwd_bf16': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_cast_split_f32_bf16': (c_i, [c_fp, c_fp, c_fp, c_i64, c_fp]),
    'gssd_self_attn_flash_bwd_supported': (c_i, [c_i, c_i]),
    'gssd_self_attn_core_x6_supported': (c_i, [c_i, c_i]),
    'gssd_self_attn_core_x6_ws_bytes': (C.c_longlong, [c_i, c_i, c_i, c_i]),
    'gssd_self_attn_core_x6_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    'gssd_self_attn_core_bf16v': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    'gssd_softmax_rows_f32': (c_i, [c_fp, c_i64, c_i, c_i, c_fp]),
    'gssd_slice_and_cat_f32': (c_i, [c_fp, c_fp, c_fp, c_i64, c_i, c_i, c_i, c_fp]),
    'gssd_spectral_norm_f32': (c_i, [c_fp, c_i, c_i, c_f, c_fp]),
    'gssd_dcn_im2col_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_dcn_im2col_bf16': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_dcn_packed_weight_elems_x6': (C.c_longlong, [c_i, c_i]),
    'gssd_dcn_pack_weight_x6': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    'gssd_dcn_forward_x6': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_conv_patch_x6_weight_elems': (C.c_longlong, [c_i, c_i]),
    'gssd_conv_patch_x6_pack_weight': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    'gssd_conv_patch_x6_takes': (c_i, [C.POINTER(ConvDesc)]),
    'gssd_dcn_forward_x6_ex': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_dcn_packed_weight_elems': (C.c_longlong, [c_i, c_i]),
    'gssd_dcn_pack_weight_f32': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    'gssd_dcn_streamk': (c_i, [c_i]),
    'gssd_dcn_streamk_status': (c_i, [c_fp]),
    'gssd_dcn_streamk_reset': (c_i, [c_fp]),
    'gssd_dcn_streamk_release': (c_i, [c_fp]),
    'gssd_dcn_forward_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_resample_ksize': (c_i, [c_i, c_i, c_i]),
    'gssd_resample_coeffs': (c_i, [c_i, c_i, c_i, c_fp, c_fp]),
    'gssd_resize_u8_horizontal': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_resize_u8_vertical': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    'gssd_input_finish_f32': (c_i, [c_fp, c_fp, c_f, c_f, c_f, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_eval_match': (c_i, [c_fp, C.c_longlong, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_d, c_fp, c_i, c_i, c_fp, c_fp, c_fp]),
    'gssd_eval_workspace_bytes': (C.c_longlong, [c_i]),
    'gssd_eval_ap': (c_i, [c_fp, c_fp, c_i, c_i, c_d, c_i, c_fp, C.c_longlong, c_fp, c_fp]),
    'gssd_bgemm_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.c_longlong, C.c_longlong, C.c_longlong, c_i, c_f,
                             c_i, c_fp]),
    'gssd_softmax_bwd_rows_f32': (c_i, [c_fp, c_fp, c_i64, c_i, c_i, c_fp]),
    'gssd_sn_weight_grad_f32': (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp]),
    'gssd_scaled_transpose_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_fp]),
    'gssd_dot_f32': (c_i, [c_fp, c_fp, c_i64, c_fp, c_fp]),
    'gssd_rowdot_f32': (c_i, [c_fp, c_fp, c_fp, c_i64, c_i, c_fp]),
    'gssd_bgemm_ex_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.c_longlong, C.c_longlong, C.c_longlong, c_i, c_f,
                                c_i, c_fp, c_fp, c_fp]),
    'gssd_axpby_f32': (c_i, [c_fp, c_fp, c_fp, c_i64, c_f, c_f, c_fp]),
    'gssd_scale_cast_f64_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_fp]),
    'gssd_sa_sigma_grad_f32': (c_i, [c_fp, c_fp, c_fp, c_i, c_fp, c_fp]),
    'gssd_dcn_col2im_f32': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    'gssd_match_batch': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_f, c_f, c_f, c_fp, c_fp, c_fp]),
    'gssd_reduce_max_f32': (c_i, [c_fp, c_i64, c_fp, c_i, c_fp]),
    'gssd_hnm_loss': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    'gssd_loss_finalize': (c_i, [c_fp, c_i, c_fp, c_fp, c_fp]),
    'gssd_loss_finalize_global': (c_i, [c_fp, c_i, c_fp, c_i, c_fp, c_fp, c_fp]),
    'gssd_loss_backward': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    'gssd_detect': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    'gssd_softmax_lastdim_f32': (c_i, [c_fp, c_fp, c_i64, c_i, c_fp]),
}


class GssdError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise GssdError(
            f'{LIB_PATH} is missing: the HIP kernels are not built.  Run `make -C '
            f'{os.path.join(_HERE, "csrc")}` (needs hipcc, --offload-arch=gfx950) or '
            f'`python -c "import __graft_entry__ as g; g.build()"`.  There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GssdError(f'{LIB_PATH} does not export {name}; rebuild it') from e
        fn.restype = res
        fn.argtypes = args
    if lib.gssd_plan_op_size() != C.sizeof(PlanOp):
        raise GssdError(f'{LIB_PATH}: struct gssd_plan_op is {lib.gssd_plan_op_size()} bytes in the library, {C.sizeof(PlanOp)} in this '
                        f'binding; rebuild the library')
    if lib.gssd_conv_desc_size() != C.sizeof(ConvDesc):
        raise GssdError(f'{LIB_PATH}: struct gssd_conv_desc is {lib.gssd_conv_desc_size()} bytes in the library, '
                        f'{C.sizeof(ConvDesc)} in this binding; rebuild the library')
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise GssdError(f'libgssd_hip: error {rc}: {lib.gssd_last_error().decode()}')
