/*
 * gssd_hip.h -- C ABI of libgssd_hip.so: the MI355X (gfx950) kernels behind the GSSD / GSSD++
 * detection hot path of L0SG/grouped-ssd-pytorch.
 *
 * The reference has no native code and no FFI of its own: its "kernel layer" is
 * whatever ATen/cuDNN op each Python line dispatches, plus the un-vendored `dcn_v2` CUDA
 * extension.  Each entry point below replaces one such implicit kernel family; the comment
 * on it cites the reference call site it stands in for (paths relative to
 * /root/reference/ssd_liverdet/).  The Python host code binds these with ctypes
 * (grouped-ssd-pytorch_amd/gssd/_lib.py); INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - plain C: raw device pointers, ints, floats; no torch / HIP C++ types in any signature.
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), allocates
 *     nothing, and returns 0 on success or a negative GSSD_E* code (gssd_last_error() gives text).
 *   - activations are NHWC fp32 ("pixel-major"): element (b, y, x, c) at ((b*H + y)*W + x)*stride + c.
 *     Channels are phase-major, so conv group g owns the contiguous slab [g*C/G, (g+1)*C/G).
 *   - re-entrant per stream: no global scratch; all workspaces are caller-provided.
 */
#ifndef GSSD_HIP_H
#define GSSD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSSD_OK 0
#define GSSD_EINVAL (-1)   /* bad argument / unsupported shape */
#define GSSD_ELAUNCH (-2)  /* hipLaunch / runtime error */

typedef void* gssd_stream_t; /* hipStream_t */

int gssd_abi_version(void);
/* sizeof(gssd_conv_desc) as the library was built: a binding checks its own struct layout against it */
int gssd_conv_desc_size(void);
const char* gssd_last_error(void);
/* name of the GPU architecture the library was built for ("gfx950") */
const char* gssd_build_arch(void);

/* Timing events that survive hipGraph capture (round 6, ABI 8): gssd_event_record_node() records `ev` on `stream`; under stream capture it becomes
 * an event-record NODE of the graph (hipEventRecordExternal), so a caller that replays its step from a hipGraph can still time single launches
 * INSIDE the replay, beside whatever runs on the graph's other branches -- bench.py's roofline measurement of kernels launched many times per
 * step.  After a replay has finished, gssd_event_elapsed_ms() gives the time between two of them in that replay. */
typedef void* gssd_event_t; /* hipEvent_t */
int gssd_event_create(gssd_event_t* ev);
int gssd_event_destroy(gssd_event_t ev);
int gssd_event_record_node(gssd_event_t ev, gssd_stream_t stream);
int gssd_event_elapsed_ms(gssd_event_t start, gssd_event_t stop, float* ms);

/* ------------------------------------------------------------------------------------------
 * Layout packing
 * ------------------------------------------------------------------------------------------ */

/* NCHW image batch -> NHWC with each group's channels padded from cpg_in to cpg_out (zeros).
 * Replaces: the implicit NCHW tensor handed to vgg[0] (train_lesion_multiphase_v2.py:198,
 * models/ssd_multiphase_custom_group.py:258-259). */
int gssd_pack_input_nhwc(const float* x_nchw, float* y_nhwc, int B, int C, int H, int W, int groups,
                         int cpg_out, gssd_stream_t stream);

/* NHWC -> NCHW (for handing activations back to torch callers, e.g. visualize=True outputs). */
int gssd_unpack_nhwc_to_nchw(const float* x_nhwc, float* y_nchw, int B, int C, int H, int W, int x_stride,
                             gssd_stream_t stream);

/* Conv weight OIHW [Cout][cin_g][KH][KW] -> K-major rows [Cout][Kpad], k = (kh*KW + kw)*cin_g_pad + c,
 * zero padded. */
int gssd_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int cin_g, int KH, int KW,
                          int cin_g_pad, int Kpad, gssd_stream_t stream);

/* One launch for a table of such packs (device array of items; a plain copy of n floats is {Cout 1, cin_g n, taps 1, cin_g_pad n,
 * Kpad n}): the refresh of every packed weight after an optimizer step. */
typedef struct gssd_pack_item {
    const float* w;  /* OIHW source */
    float* wp;       /* packed rows [Cout][Kpad] */
    int32_t Cout, cin_g, taps, cin_g_pad, Kpad, reserved;
} gssd_pack_item;
int gssd_pack_conv_weights_batched(const gssd_pack_item* items_dev, int n_items, gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32)
 * ------------------------------------------------------------------------------------------
 * out[m, n] = epilogue( sum_k A[m, k] * Wp[n, k] ),  m = output pixel, n = output channel,
 * k = (tap, input channel of the group).  One kernel family serves
 *   K1/K2  grouped 3x3 / dilated 3x3 / grouped 1x1 trunk convs (models/...group.py:444,451-452)
 *   K5     extras (:463-490)            K7  1x1 fuse convs (:77-139)
 *   K8     loc/conf heads + permute + cat (:375-380, out_mode = GSSD_OUT_HEADS)
 *   K9     Self_Attn theta/phi/g/o 1x1 convs (layers/self_attn.py:62-82)
 *   K10    the two attention bmm's (layers/self_attn.py:71,80) via per-image operands
 *   K13    DCN offset/mask conv (layers/dcn_v2_custom.py:80)   K14  DCN main contraction
 */
#define GSSD_OUT_NHWC 0
#define GSSD_OUT_TRANSPOSED 1 /* per image [n][m] with row stride out_stride (needs m_per_image) */
#define GSSD_OUT_HEADS 2      /* n < split_n -> out (loc), else -> out_b (conf), SSD prior order */
#define GSSD_OUT_SPLIT_T 3    /* merged projections (Self_Attn theta|phi|g, one pass over x): n < split_n -> out, NHWC rows of
                                 out_stride floats; n >= split_n -> out_b, per image TRANSPOSED [n - split_n][m] with rows of
                                 out_b_stride floats (zero padded), images outb_batch_stride apart; one group, split_n a multiple
                                 of 64; m_per_image, or (fp32 entry) all images in one M range when Ho*Wo % 4 == 0 */

typedef struct gssd_conv_desc {
    const float* in;    /* NHWC activations */
    const float* wgt;   /* packed rows [Cout][wgt_row_stride], K-major */
    const float* bias;  /* [Cout] or NULL */
    float* out;
    float* out_b;       /* GSSD_OUT_HEADS second buffer, else NULL */
    const float* alpha; /* per-output-channel scale [Cout] applied to the accumulator (1/sigma of spectral norm) or NULL */
    const float* gate;  /* device scalar: out = resid + gate*(acc*alpha + bias) (Self_Attn sigma) or NULL */
    const float* resid; /* NHWC, same geometry as out, or NULL */
    float* out2;        /* receives gate*(acc*alpha+bias) when gate != NULL and out2 != NULL */
    const float* in_scale; /* fused producer BatchNorm+ReLU: the input is read as max(x*in_scale[c] + in_shift[c], 0) */
    const float* in_shift; /* (per input channel, from gssd_bn_finalize_f32); NULL = plain input.  Zero padding applies */
    const float* in_pad;   /* AFTER the transform: out-of-image taps read in_pad[c], a value the transform maps to 0.  Layout hint (no
                            * change of meaning): when in_pad == in + B*H*W*in_stride (the vector stored directly behind a dense map) the
                            * fp32 Winograd kernel loads the padding value through the tap's address instead of selecting it per element */
    double* stats;      /* [2*Cout]: per-channel sum / sum of squares of the pre-activation output
                           accumulated with fp64 atomics (BatchNorm batch statistics), or NULL */
    const float* wgt_wino; /* optional Winograd F(2x2,3x3) form of `wgt` (gssd_winograd_weight_f32): 3x3 / stride 1 / pad 1
                              convs of a Winograd shape (see gssd_winograd_weight_f32) then take the Winograd kernel; NULL = never */
    const float* pool_sign; /* flags & GSSD_CONV_POOL2 only: per-output-channel BatchNorm weight (gamma) of the layer's own BatchNorm;
                               only its SIGN is read (see GSSD_CONV_POOL2) */
    const void* wgt_x6;    /* optional three-plane bf16 form of `wgt` (gssd_conv_x6_pack_weight, tile gssd_conv_x6_tile): plain-epilogue
                              convs with cin_g % 32 == 0 then run csrc/conv_x6.hip (fp32-equivalent products on the bf16 matrix cores,
                              see gssd_conv_x6_takes); NULL = never.  fp32 entry point only */
    const void* wgt_patch; /* optional two fp16 planes of `wgt` in the order of csrc/conv_patch_x6.hip (gssd_conv_patch_x6_pack_weight): dense 3x3 /
                              stride 1 / pad 1 convs with cin % 32 == 0 and <= 128 outputs (the DCN offset conv) that carry GSSD_CONV_F16_OK
                              run there (gssd_conv_patch_x6_takes); NULL = never.  fp32 entry point only.  Round 6, ABI 8 */
    int B, H, W;        /* input geometry */
    int in_stride;      /* floats between consecutive input pixels */
    int in_ch_off;      /* first input channel used */
    int Ho, Wo;
    int Cout, groups;
    int cin_g;          /* input channels per group (multiple of 4) */
    int KH, KW, stride, pad, dil;
    int K;              /* KH*KW*cin_g */
    int wgt_row_stride; /* floats between packed weight rows (>= K, multiple of 4) */
    int out_stride;     /* floats between output pixels (NHWC) or between channel rows (TRANSPOSED) */
    int out_ch_off;
    int out_mode;
    int relu;
    int m_per_image;    /* 1: grid.z = image, tiles do not cross images, *_batch_stride apply */
    int split_n;        /* GSSD_OUT_HEADS: channels [0, split_n) are loc, the rest conf */
    int split_k;        /* >= 1; > 1 slices K over grid.z and accumulates with fp32 atomics into a zero-filled
                           output (small-M / long-K launches such as the heads); plain epilogues only */
    int out_b_stride;   /* GSSD_OUT_SPLIT_T: floats between the transposed rows of out_b */
    int flags;          /* GSSD_CONV_* bits (bf16 entry point only) */
    int stats_rep;      /* 0 / 1: `stats` is one [2*Cout] array.  R > 1: `stats` holds R replicas of [2*Cout] doubles, R * 2 * Cout in all
                           (zero-filled by the caller); a workgroup adds its sums into replica (workgroup id mod R) and the consumers
                           -- gssd_bn_finalize_*, gssd_bn_relu_pool_*, gssd_bn_bwd_finalize_f32, which take the same R -- add the
                           replicas up in a fixed order.  Device-scope fp64 atomics on one cache line are served serially; the
                           persistent trunk kernels flush every workgroup's sums at the end of the launch (conv2_1 in bf16: 131 k
                           atomics on 16 lines = 60 of its 140 us).  The split of the sums over replicas depends on the grid, their
                           total does not (up to fp64 rounding of the order of addition) */
    int64_t in_batch_stride, wgt_batch_stride, out_batch_stride, outb_batch_stride;
    int64_t out_off, outb_off; /* GSSD_OUT_HEADS: float offset of this source inside one image's rows */
} gssd_conv_desc;

int gssd_conv2d_nhwc_f32(const gssd_conv_desc* d, gssd_stream_t stream);

/* fp32 convolution with fp32-equivalent products on the BF16 matrix cores (csrc/conv_x6.hip; nn.Conv2d of the reference's trunk,
 * models/ssd_multiphase_custom_group.py vgg() / add_extras()): every operand is the exact sum of three bf16 planes, six
 * v_mfma_f32_16x16x32_bf16 per product (all terms above 2^-24), fp32 accumulation.  gssd_conv_x6_tile: the N tile (64 / 128 / 256) the
 * launch of a (cout_g, groups, M = B*Ho*Wo) conv uses -- the packed weights depend on it.  gssd_conv_x6_weight_elems: bf16 elements of
 * the packed form (-1: not a shape the kernel takes).  gssd_conv_x6_pack_weight: K-major fp32 rows [Cout][row_stride] (gssd_pack_conv_weight)
 * -> the three planes.  gssd_conv_x6_takes: 1 when gssd_conv2d_nhwc_f32 runs the descriptor on this kernel (wgt_x6 set, NHWC output,
 * epilogue = bias / ReLU / batch sums only). */
int gssd_conv_x6_tile(int cout_g, int groups, long long M);
long long gssd_conv_x6_weight_elems(int Cout, int groups, int cin_g, int taps, int BN);
int gssd_conv_x6_pack_weight(const float* w_packed, void* w_x6, int Cout, int groups, int cin_g, int taps, int row_stride, int BN,
                             gssd_stream_t stream);
int gssd_conv_x6_takes(const gssd_conv_desc* d);

/* bf16 storage mode (BASELINE.json configs[4]: bf16 weights / activations, bf16 MFMA v_mfma_f32_16x16x32_bf16, fp32
 * accumulation).  Same descriptor; `in`, `wgt`, `resid`, `out`, `out2`, `in_pad` point to bf16 (uint16) data, element strides /
 * offsets are in bf16 elements and must be multiples of 8; bias / alpha / gate / in_scale / in_shift / stats stay fp32 / fp64.
 * flags & GSSD_CONV_OUT_F32: `out` (and `out_b`) are fp32 -- required for GSSD_OUT_HEADS (loc / conf stay fp32 for the loss and
 * NMS), GSSD_OUT_TRANSPOSED and GSSD_OUT_SPLIT_T (the Self_Attn projections feed the fp32 softmax core). */
#define GSSD_CONV_OUT_F32 1
/* GSSD_OUT_SPLIT_T only: the transposed second range (`out_b`) is bf16, its rows out_b_stride (a multiple of 32) elements long, and
 * inside every block of 32 tokens token 16a + 4b + c sits at position 8b + 4a + c (the key order of gssd_self_attn_core_bf16v) */
#define GSSD_CONV_OUTB_BF16_PERM32 2
/* GSSD_OUT_HEADS with split_k > 1: instead of fp32 atomics into zero-filled outputs, reduction slice k writes its partial sums
 * with plain stores to out + k * B * out_batch_stride (and out_b + k * B * outb_batch_stride); gssd_heads_reduce_f32 then adds
 * the slices in order: run-to-run identical loc / conf */
#define GSSD_CONV_HEADS_SLICES 4
/* Both entry points; trunk layers that are followed by BatchNorm + ReLU + a 2x2 / stride-2 max-pool (conv1_2, conv2_2, conv3_3) when
 * no backward will read the raw map: `out` is the POOLED raw map [B][ceil(Ho/2)][ceil(Wo/2)][Cout] -- per channel the maximum of the
 * 2x2 window's raw outputs where pool_sign[c] >= 0, the minimum where it is negative (ceil mode: windows cut by the border take
 * what exists).  Max-pooling commutes with a monotone map, and y -> max(y * scale + shift, 0) is non-decreasing for scale >= 0 and
 * non-increasing for scale < 0, sign(scale) = sign(gamma): the consumer's deferred BatchNorm + ReLU of this pooled map is BIT-IDENTICAL
 * to BatchNorm + ReLU + max-pool of the full map, which is then never written or re-read (the BatchNorm batch sums in `stats` are
 * still those of the full map).  Only the kernels that support it accept the flag (fp32: the Winograd trunk kernels; bf16: the thin
 * kernels with 16 / 32 output channels per group); everything else returns GSSD_EINVAL. */
#define GSSD_CONV_POOL2 8
/* bf16 entry point: `resid` is an fp32 map of the OUTPUT's geometry (with GSSD_CONV_OUT_F32: the backward of the bf16 storage mode
 * accumulates a data gradient computed on the bf16 matrix cores into an fp32 gradient map).  Default: `resid` has the input's type (bf16). */
#define GSSD_CONV_RESID_F32 16
/* fp32 mode, the CALLER's promise: every operand of this launch -- the input map (after the fused producer BatchNorm + ReLU, if any) and the
 * weights -- lies inside fp16's range (|v| < 65 504; full precision for 4e-3 <= |v|, below that the absolute error stays under 5e-10).  The
 * split-operand kernels (csrc/conv_x6.hip, conv_wino_x6.hip, conv_thin_x6.hip, dcn_x6.hip) then split into fp16 planes and issue three MFMAs per
 * product instead of six (x = h + l / 2^k to 2^-24 |x|).  It is NEVER inferred from the descriptor: a value beyond the range becomes inf.  The
 * engine sets it on forward launches of a train-mode network (batch-statistics BatchNorm bounds |y| by |gamma| sqrt(n) + |beta|), never in eval
 * mode (running statistics bound nothing) and never on a data-gradient launch (fp16 has no exponent range for gradients).  GSSD_X6_F16=0 (environment)
 * ignores it. */
#define GSSD_CONV_F16_OK 32
int gssd_conv2d_nhwc_bf16(const gssd_conv_desc* d, gssd_stream_t stream);
/* OIHW fp32 -> packed bf16 rows [Cout][Kpad] (cin_g_pad, Kpad multiples of 8); fp32 -> bf16 array cast (round to nearest even) */
int gssd_pack_conv_weight_bf16(const float* w_oihw, void* w_packed, int Cout, int cin_g, int KH, int KW, int cin_g_pad, int Kpad,
                               gssd_stream_t stream);
int gssd_cast_f32_bf16(const float* x, void* y, int64_t n, gssd_stream_t stream);
/* bf16 -> fp32 array cast (exact; x and y 16-byte aligned): the backward of the bf16 storage mode runs on fp32 copies of the
 * activations the forward stored (gssd/backward.py::Bf16Shadow) */
int gssd_cast_bf16_f32(const void* x, float* y, int64_t n, gssd_stream_t stream);
/* bf16 variants of the HBM-bound passes (fp32 arithmetic, one rounding on store): input pack (3 -> 8 channels per phase),
 * BatchNorm + ReLU + max-pool, BN finalize with a bf16 pad vector, L2Norm */
int gssd_pack_input_nhwc_bf16(const float* x_nchw, void* y_nhwc, int B, int C, int H, int W, int groups, gssd_stream_t stream);
int gssd_bn_relu_pool_bf16(const void* raw, void* out, int B, int H, int W, int C, int Ho, int Wo, int pool_k, int pool_s, int pool_p,
                           const double* stats, double count, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, int training, int relu, int stats_rep,
                           gssd_stream_t stream);
int gssd_bn_finalize_bf16(const double* stats, double count, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float momentum, float eps, int training, int C, float* scale, float* shift,
                          void* pad_bf16, int stats_rep, gssd_stream_t stream);
int gssd_l2norm_bf16(const void* x, const float* weight, void* out, int64_t pixels, int C, float eps, gssd_stream_t stream);

/* U[g][xi][co][ci] = (G g G^T)_xi of the packed K-major weights [Cout][tap*cin_g + ci] (row stride `row_stride`).
 * Winograd shapes: cin_g % 16 == 0 and cout_g % 32 == 0, or (one group) any cout_g >= 24 -- U's rows per group are then
 * padded with zeros to a multiple of the kernel's channel block; gssd_winograd_weight_elems (HOST) returns the float
 * count of the buffer the packer fills, or -1 for other shapes: the fp32 U followed -- for the shapes csrc/conv_wino_x6.hip takes
 * (cin_g % 16 == 0, cout_g >= 24) -- by U's three bf16 planes and, round 6, its two fp16 planes (h and (v - h) * 2048: the GSSD_CONV_F16_OK form) in that kernel's staging order (two 16-bit values per counted float).  The same packer serves the data gradient (weights packed by
 * gssd_pack_conv_weight_dgrad). */
long long gssd_winograd_weight_elems(int Cout, int groups, int cin_g);
int gssd_winograd_weight_f32(const float* w_packed, float* U, int Cout, int groups, int cin_g, int row_stride,
                             gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Convolution backward (replaces the cuDNN dgrad / wgrad kernels behind loss.backward(),
 * train_lesion_multiphase_v2.py:247-248)
 * ------------------------------------------------------------------------------------------
 * Weight gradient: dw_packed[n][k] += sum_m dy[m][n] * im2col(in)[m][k] for the forward described by `d`
 * (d->in / geometry / in_scale.. are used; d->wgt, d->out are ignored).  dy is the dense NHWC gradient
 * [B*Ho*Wo][Cout]; dw_packed [Cout][K] must be zero-filled (split-K partials are added with fp32 atomics).
 * gssd_unpack_conv_weight_grad turns it into the OIHW gradient (accumulate != 0: +=).
 * Data gradient: a stride-1 conv's dgrad is a forward gssd_conv2d_nhwc_f32 over dy with the weights packed by
 * gssd_pack_conv_weight_dgrad (rows = input channels, k' = flipped tap * cout_g + co) and pad' = dil*(k-1) - pad. */
int gssd_conv2d_wgrad_f32(const gssd_conv_desc* d, const float* dy, float* dw_packed, gssd_stream_t stream);
/* The same weight gradient on the bf16 matrix cores (training step of the bf16 storage mode): d->in and dy are bf16 (dy [B*Ho*Wo][Cout]),
 * dw_packed is fp32 and zero-filled; a deferred BatchNorm + ReLU on the input (d->in_scale / d->in_shift, fp32) is applied to the staged
 * input in fp32 and rounded to bf16 like the forward does (d->in_pad is not read: out-of-image pixels are zero).  Shapes: the grouped 3x3
 * stride-1 pad-1 trunk convolutions (16 -> 16 / 32 and 32 -> 32 channels per group with groups % 4 == 0; 32 / 64 -> multiples of 64;
 * 128 -> multiples of 32) -- gssd_conv2d_wgrad_bf16_supported() returns 1 for them; anything else is GSSD_EINVAL, not a fallback. */
int gssd_conv2d_wgrad_bf16(const gssd_conv_desc* d, const void* dy, float* dw_packed, gssd_stream_t stream);
int gssd_conv2d_wgrad_bf16_supported(const gssd_conv_desc* d);
int gssd_unpack_conv_weight_grad(const float* w_packed, float* w_oihw, int Cout, int cin_g, int KH, int KW,
                                 int cin_g_pad, int Kpad, int accumulate, gssd_stream_t stream);
int gssd_pack_conv_weight_dgrad(const float* w_oihw, float* w_packed, int Cout, int groups, int cin_g, int KH, int KW,
                                gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm (train / eval) + ReLU + max-pool, one pass
 * ------------------------------------------------------------------------------------------
 * Replaces nn.BatchNorm2d + nn.ReLU + nn.MaxPool2d (models/...group.py:439-441,446,450,455-456;
 * extras :477,484 with the ReLU of :354-355; fuse BNs :292,319,369).  training != 0: batch mean and
 * biased variance come from `stats` (filled by the conv) over `count` elements per channel, and
 * running_mean / running_var are updated in place (momentum, unbiased variance);
 * training == 0: running statistics are used.  pool_k == 0 means no pooling.  gamma == NULL means
 * identity affine (plain ReLU / max-pool pass, e.g. pool4 after the DCN block).  stats_rep (here and in gssd_bn_finalize_* /
 * gssd_bn_bwd_finalize_f32): the number of replicas `stats` holds (gssd_conv_desc::stats_rep of the conv that filled it; 0 / 1 = one).
 */
int gssd_bn_relu_pool_f32(const float* raw, float* out, int B, int H, int W, int C, int Ho, int Wo, int pool_k,
                          int pool_s, int pool_p, const double* stats, double count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          int training, int relu, int stats_rep, gssd_stream_t stream);

/* BatchNorm statistics -> per-channel affine, for consumers that apply BN+ReLU on the fly (gssd_conv_desc.in_scale):
 * scale = gamma / sqrt(var + eps), shift = beta - mean*scale, pad = -/+3e38 (mapped to <= 0 by the transform);
 * training != 0 uses the fp64 batch sums and updates running_mean / running_var like nn.BatchNorm2d. */
int gssd_bn_finalize_f32(const double* stats, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, int training, int C,
                         float* scale, float* shift, float* pad, int stats_rep, gssd_stream_t stream);

/* BatchNorm(train) + ReLU + max-pool backward, three launches (replaces autograd's native_batch_norm_backward /
 * threshold_backward / max_pool2d_with_indices_backward):
 *   reduce  : dz = d(out) routed through the pool's first-max and the ReLU mask of z = raw*scale + shift (scale == NULL:
 *             z = raw, i.e. a plain pool / ReLU backward); sums[2C] += (sum dz, sum dz*raw) (fp64, may be NULL).
 *             For overlapping pools (stride < kernel) dz must be zero-filled by the caller.
 *   finalize: per-channel constants of d(raw) = A*dz + B*raw + C from the forward batch sums, plus dgamma, dbeta.
 *   apply   : dz <- A*dz + B*raw + C in place; optional column sums of the result (the conv bias gradient). */
int gssd_bn_bwd_reduce_f32(const float* dout, const float* raw, const float* scale, const float* shift, float* dz,
                           double* sums, int B, int H, int W, int C, int Ho, int Wo, int pool_k, int pool_s, int pool_p,
                           int relu, gssd_stream_t stream);
int gssd_bn_bwd_finalize_f32(const double* fwd_stats, double count, const double* bwd_sums, const float* gamma, float eps,
                             int C, float* coef_a, float* coef_b, float* coef_c, float* dgamma, float* dbeta,
                             int stats_rep, gssd_stream_t stream);
int gssd_bn_bwd_apply_f32(float* dz, const float* raw, const float* coef_a, const float* coef_b, const float* coef_c,
                          int64_t pixels, int C, double* colsum, gssd_stream_t stream);
/* The same for a layer WITHOUT pooling, straight from d(out): dz = dout * [raw * scale + shift > 0] (relu) is re-derived here, so
 * gssd_bn_bwd_reduce_f32 may be called with dz = NULL (sums only) -- one HBM pass less per layer than reduce-writes / apply-reads. */
/* The same two passes for the training step of the bf16 storage mode: `raw_bf16` is the forward's own stored map (bf16), gradients stay
 * fp32.  gssd_bn_bwd_apply_mixed: d = dout * [relu mask] when dout != NULL, else d is read from dz (what the pooled reduce pass wrote);
 * the result a*d + b*raw + c goes to dz_bf16 (bf16, when != NULL: the operand of the bf16 data-gradient conv and weight gradient) and,
 * when store_f32 != 0, to dz.  dout_bf16 != 0: d(out) itself is a bf16 map (the two thin trunk layers whose consumer's data-gradient conv
 * runs on the patch-staged bf16 kernel, which stores bf16).
 * gssd_bn_bwd_reduce_mixed, dz_bf16 != NULL (non-overlapping pools): the routed gradient goes out as a bf16 map instead of dz; the apply
 * pass then takes it as its d(out) (dout = that map, dout_bf16 = 1, relu = 0, scale = shift = NULL). */
int gssd_bn_bwd_reduce_mixed(const void* dout, int dout_bf16, const void* raw_bf16, const float* scale, const float* shift, float* dz,
                             void* dz_bf16, double* sums, int B, int H, int W, int C, int Ho, int Wo, int pool_k, int pool_s, int pool_p,
                             int relu, gssd_stream_t stream);
int gssd_bn_bwd_apply_mixed(const void* dout, int dout_bf16, float* dz, void* dz_bf16, const void* raw_bf16, const float* scale,
                            const float* shift, int relu, const float* coef_a, const float* coef_b, const float* coef_c, int64_t pixels,
                            int C, double* colsum, int store_f32, gssd_stream_t stream);
int gssd_bn_bwd_apply_masked_f32(const float* dout, const float* raw, const float* scale, const float* shift, int relu,
                                 const float* coef_a, const float* coef_b, const float* coef_c, float* draw, int64_t pixels, int C,
                                 double* colsum, gssd_stream_t stream);
/* out[c] += sum_rows x[row*row_stride + c] (fp64): conv bias gradients. */
int gssd_colsum_f32(const float* x, int64_t rows, int C, int row_stride, double* out, gssd_stream_t stream);
int gssd_cast_f64_f32(const double* x, float* y, int n, int accumulate, gssd_stream_t stream);
/* L2Norm backward (layers/modules/l2norm.py:19-23): dx (+ dx_add if not NULL), dweight[C] += (fp64). */
int gssd_l2norm_bwd_f32(const float* x, const float* weight, const float* dy, float* dx, const float* dx_add,
                        double* dweight, int64_t pixels, int C, float eps, gssd_stream_t stream);
/* Gradient of one multibox source's merged (loc|conf) head output, gathered from d(loc)[B,P,4] / d(conf)[B,P,nc]. */
int gssd_heads_gather_f32(const float* dloc, const float* dconf, float* out, int B, int HW, int A, int nc, int P,
                          int prior_off, gssd_stream_t stream);
/* u[b, s*i, s*j, :] = dy[b, i, j, :], zero elsewhere: turns a strided conv's dgrad into a stride-1 conv. */
int gssd_upsample_insert_f32(const float* dy, float* u, int B, int Ho, int Wo, int H, int W, int C, int s,
                             gssd_stream_t stream);

/* x / (sqrt(sum_c x^2) + eps) * w_c per pixel.  Replaces layers/modules/l2norm.py:19-23. */
int gssd_l2norm_f32(const float* x, const float* weight, float* out, int64_t pixels, int C, float eps,
                    gssd_stream_t stream);

/* Self_Attn core, flash style (layers/self_attn.py:68-80: torch.bmm(theta^T, phi) -> Softmax(dim=-1) -> torch.bmm(g, attn^T)):
 * out[b][i][c] = sum_j softmax_j(sum_d tp[b][i][d] * tp[b][j][D + d]) * gT[b][c][j]; the [N, N] attention map is never written.
 * tp [B][N][2*D] token-major theta|phi, gT [B][C2][Np] (Np >= N, multiple of 4, pad columns zero), out [B][N][C2].
 * Built for (D, C2) = (64, 256), (128, 512), (32, 128) -- Self_Attn on 512 / 1024 / 256 channels. */
int gssd_self_attn_core_f32(const float* tp, const float* gT, void* out, int B, int N, int Np, int D, int C2, int out_bf16,
                            gssd_stream_t stream); /* out_bf16 != 0: `out` is bf16 (configs[4]); the softmax is fp32 either way */

/* The same core with the keys / values as their own tensors: kp [B][Nk][kstride] (first D floats of a row = the key), gT [B][C2][Nkp].
 * gssd_self_attn_core_f32 is the case kp = tp + D, Nk = N, kstride = 2 D; Self_Attn with max_pool_factor > 1 passes the pooled
 * phi / g of gssd_sa_pool_kv_f32 (layers/self_attn.py:57-59, 67, 76: Nk = max(H / factor, 1)^2 keys for N = H^2 queries). */
int gssd_self_attn_core_kv_f32(const float* tp, const float* kp, const float* gT, void* out, int B, int N, int Nk, int Nkp, int D,
                               int C2, int kstride, int out_bf16, float* lse, gssd_stream_t stream);
/* lse (optional, [B][N]): log-sum-exp of every query's logits -- what the training step keeps of the softmax for its backward */

/* F.adaptive_avg_pool2d(phi, P) / (g, P) of Self_Attn (layers/self_attn.py:67, 76) on the projection outputs: tp [B][H*H][2*C8]
 * (theta | phi) -> kp [B][P*P][C8]; gT [B][C2][Np] -> gTp [B][C2][Nkp] (pad columns zero).  Cell o of the P-grid covers rows
 * [floor(o H / P), ceil((o + 1) H / P)). */
int gssd_sa_pool_kv_f32(const float* tp, const float* gT, float* kp, float* gTp, int B, int H, int P, int C8, int C2, int Np, int Nkp,
                        gssd_stream_t stream);

/* Backward of that pooling: dkg [B][P*P][CW] (gradients of the pooled phi | g, token-major) -> dst[b][n][j] for j < CW, rows `ld`
 * floats apart = sum over the cells containing token n of dkg[cell][j] / |cell|. */
int gssd_sa_unpool_f32(const float* dkg, float* dst, int B, int H, int P, int CW, int ld, gssd_stream_t stream);

/* 1 when gssd_conv2d_nhwc_f32 runs this descriptor on the slot-scheduled 128 x 256 GEMM stream (csrc/gemm_slot.hip: large plain
 * 1x1 convolutions / GEMMs), 0 when it stays on the generic implicit-GEMM kernel.  Host logic only. */
int gssd_gemm_slot_takes(const gssd_conv_desc* d);

/* bf16 mode: pixels per workgroup (128 or 256) when gssd_conv2d_nhwc_bf16 runs this descriptor on the flat-window kernel
 * (csrc/conv_flat_bf16.hip: grouped 3x3 trunk layers with 32 .. 128 channels per group, conv3_1 .. conv6), 0 when another kernel
 * takes it.  Host logic only.  gssd_conv_flat_bf16_tile(0 | 128 | 256) overrides the per-shape choice (0 = automatic; ablation and
 * tests) and returns the previous setting. */
int gssd_conv_flat_bf16_takes(const gssd_conv_desc* d);
int gssd_conv_flat_bf16_tile(int pixels_per_workgroup);

/* bf16-storage variant (configs[4]): theta | phi fp32 and the logits on the fp32 matrix cores (a bf16 logit would move its
 * probability by tens of percent), g^T and the probabilities bf16 on v_mfma_f32_16x16x32_bf16 (80 % of the block's FLOPs), out bf16.
 * gT rows are Np32 (multiple of 32) bf16 long with the token order of GSSD_CONV_OUTB_BF16_PERM32.
 * lse (optional, [B][N] fp32): the rows' log-sum-exp, kept by training forwards for the backward's exp(s - lse) GEMM epilogue. */
int gssd_self_attn_core_bf16v(const float* tp, const void* gT_bf16, void* out_bf16, int B, int N, int Np32, int D, int C2,
                              float* lse, gssd_stream_t stream);

/* Flash-style backward of the attention core for the training step of the bf16 storage mode (no [N][N] map): with S = theta phi^T,
 * P = exp(S - lse), dP = d(attn_g) g^T, dS = P o (dP - dvec):  dtpg[b][i] = [ dS phi | dS^T theta | P^T d(attn_g) ]  (d theta | d phi | d g,
 * token-major rows of 2 D + C2 floats).  tp [B][N][2D] fp32 (theta | phi) and its bf16 copy tp_bf16; g_bf16, dag_bf16 [B][N][C2] bf16
 * token-major; lse [B][N] from the forward (gssd_self_attn_core_bf16v); dvec[b][i] = <d(attn_g)_i, attn_g_i> (gssd_rowdot_f32).  The
 * logits are recomputed on the fp32 matrix cores, everything else on bf16 operands with fp32 accumulation.  (D, C2) in {(64, 256),
 * (32, 128), (128, 512)}: gssd_self_attn_flash_bwd_supported; anything else is GSSD_EINVAL (the caller keeps the explicit-map path).
 * tp_bf16_lo (optional): bf16(tp - tp_bf16), from gssd_cast_split_f32_bf16 -- the logits are then recomputed as hi.hi + hi.lo + lo.hi on the
 * bf16 matrix cores (2^-16 of a product dropped) instead of the fp32 ones: 1/5 of the logits' matrix-pipe time. */
int gssd_self_attn_flash_bwd_bf16(const float* tp, const void* tp_bf16, const void* tp_bf16_lo, const void* g_bf16, const void* dag_bf16,
                                  const float* lse, const float* dvec, float* dtpg, int B, int N, int D, int C2, gssd_stream_t stream);
int gssd_self_attn_flash_bwd_supported(int D, int C2);

/* Row softmax in place over [rows][row_stride], first n columns; pad columns are zeroed.
 * Replaces nn.Softmax(dim=-1) on the attention logits (layers/self_attn.py:72). */
int gssd_softmax_rows_f32(float* x, int64_t rows, int n, int row_stride, gssd_stream_t stream);

/* Per-phase channel interleave [x_p | a_p] (models/...group.py:185-192). */
int gssd_slice_and_cat_f32(const float* a, const float* b, float* out, int64_t pixels, int Ca, int Cb,
                           int groups, gssd_stream_t stream);

/* Spectral norm (layers/spectral_norm.py:74-89): for each of n matrices W_i [rows_i][cols_i]
 * (row-major), optionally run one power iteration updating u_i, v_i in place, then write
 * 1 / (u^T W v) into inv_sigma[0 .. rows_i) (one copy per output channel: it is the `alpha` vector
 * of the conv that uses W_i).  The descriptor array lives in device memory. */
typedef struct gssd_sn_item {
    const float* w;
    float* u;
    float* v;
    float* inv_sigma;
    int rows, cols;
} gssd_sn_item;
int gssd_spectral_norm_f32(const gssd_sn_item* items_dev, int n, int do_power_iteration, float eps,
                           gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DCNv2 sampling (K12 + K14 gather side)
 * ------------------------------------------------------------------------------------------
 * Builds the modulated, bilinearly sampled column matrix cols[b*H*W + p][tap*C + c] for a 3x3 / stride 1 /
 * pad 1 / dil 1 deformable conv; offset/mask come from the raw conv_offset_mask output `om`
 * ([pixels][om_stride], channels [0, 2*dg*9) = offsets in (o1|o2) chunk order, [2*dg*9, 3*dg*9) = mask logits;
 * layers/dcn_v2_custom.py:80-83).  The contraction with the weight is then a 1x1 gssd_conv2d over cols.
 */
int gssd_dcn_im2col_f32(const float* x, const float* om, float* cols, int B, int H, int W, int C, int dg,
                        int om_stride, gssd_stream_t stream);
/* The same column matrix in bf16 from the bf16 map the bf16 forward sampled (fp32 interpolation, rounded once): the operand of the
 * training step's bf16 weight gradient. */
int gssd_dcn_im2col_bf16(const void* x_bf16, const float* om, void* cols_bf16, int B, int H, int W, int C, int dg, int om_stride,
                         gssd_stream_t stream);

/* Fused modulated deformable 3x3 conv (stride 1 / pad 1 / dil 1): sampling + contraction + bias in one kernel, no column
 * buffer.  Replaces the whole `_DCNv2.apply(input, offset, mask, weight, bias, ...)` call of layers/dcn_v2_custom.py:84-89
 * (dcn_v2's modulated_deformable_im2col + gemm).  x NHWC [B,H,W,C]; om as for gssd_dcn_im2col_f32; out NHWC [B,H,W,Cout].
 * w_packed: the dense OIHW weight [Cout][C][3][3] re-laid by gssd_dcn_pack_weight_f32 into K-chunk-major tiles
 * (gssd_dcn_packed_weight_elems floats, HOST query; -1 for unsupported shapes).  Channels per deformable group must be a
 * multiple of 16. */
long long gssd_dcn_packed_weight_elems(int Cout, int C);
int gssd_dcn_pack_weight_f32(const float* w_oihw, float* w_packed, int Cout, int C, int dg, gssd_stream_t stream);
int gssd_dcn_forward_f32(const float* x, const float* om, const float* w_packed, const float* bias, float* out, int B, int H,
                         int W, int C, int dg, int om_stride, int Cout, gssd_stream_t stream);
/* Work split of gssd_dcn_forward_f32: 1 = stream-K (one persistent workgroup per CU: whole tiles round by round, then equal spans of the
 * (tile, K-chunk) space of the tiles that do not fill a round; taken when the tile count neither fills whole rounds nor is below the CU count), 0 = one tile per workgroup, -1 = the
 * default (stream-K unless GSSD_DCN_STREAMK=0).  Returns the previous setting.  Both forms compute every output from the same
 * products; a tile cut into pieces adds their partial sums in a fixed order (results differ from the unsplit form by fp32 rounding only and
 * are the same from run to run).  The bf16 kernel keeps one tile per workgroup (its stream-K form measured slower: DESIGN.md, section 9). */
int gssd_dcn_streamk(int mode);
/* Hand-over of the stream-K form: a tile cut into pieces passes its partial sums through a 128 KB slab in accumulator layout, written
 * with 16-byte sc1 (write-through, agent-scope) stores and read with sc1 loads, published by a relaxed agent-scope flag store behind
 * `s_waitcnt vmcnt(0)` + a workgroup barrier -- correct for ANY placement of the pieces on XCDs / CUs (round 3 relied on workgroup
 * id & 7 == XCD and plain stores; round 4's run-time check of HW_REG_XCC_ID found launches placed differently when other streams were
 * busy).  Every wait is bounded (~1 s): a time-out is written to a host-visible word, and the NEXT gssd_dcn_forward_f32 on that device
 * returns GSSD_ELAUNCH ("an earlier stream-K launch timed out ...; its output is not trustworthy") and turns the form off for the
 * process.  Flags and slabs belong to the launch's `out` pointer (created on the first launch with that output outside a stream
 * capture: 32 MB per output on a 256-CU device), so launches that can be in flight together (different plans, captured graphs,
 * streams) never share them.  gssd_dcn_streamk_status: bit mask of the GSSD_DCN_SK_* conditions for the current device (runs the
 * one-time set-up if it has not run; do not call while capturing); *xcc_map (optional) = XCC id the placement probe saw for each residue
 * of workgroup id & 7 (4 bits each).  gssd_dcn_streamk_reset: zeroes every flag region on `stream` (plan build; after a failed launch). */
#define GSSD_DCN_SK_UNSUPPORTED 1 /* CU count not a multiple of 8, or the set-up allocations failed: one-tile form only */
#define GSSD_DCN_SK_MAPPING 2     /* information: workgroup id & 7 does not decide the XCD on an idle device (pieces of a tile then sit
                                     behind different L2s: slower hand-over, same results) */
#define GSSD_DCN_SK_TIMEOUT 4     /* a workgroup gave up waiting for a partial sum: stream-K is off */
int gssd_dcn_streamk_status(unsigned* xcc_map);
int gssd_dcn_streamk_reset(gssd_stream_t stream);
/* frees the stream-K flag / slab region (~32 MB) of the launches writing `out` (NULL: all regions of the current device); returns the number of
 * regions freed.  Synchronises (hipFree): call it when an output buffer is dropped, never under stream capture. */
int gssd_dcn_streamk_release(const void* out);

/* bf16 storage variant (configs[4]): x, w_packed, out bf16; om, bias fp32; fp32 blend and accumulation */
/* The fp32 deformable conv on the bf16 matrix cores with fp32-equivalent products (csrc/dcn_x6.hip): every operand as the exact sum of
 * three bf16 planes, six bf16 MFMAs per product term set (everything above 2^-24 of a product), fp32 accumulation; same argument meaning
 * as gssd_dcn_forward_f32 with w_packed from gssd_dcn_pack_weight_x6 (gssd_dcn_packed_weight_elems_x6 bf16 elements). */
long long gssd_dcn_packed_weight_elems_x6(int Cout, int C);
int gssd_dcn_pack_weight_x6(const float* w_oihw, void* w_packed, int Cout, int C, int dg, gssd_stream_t stream);
int gssd_dcn_forward_x6(const float* x, const float* om, const void* w_packed, const float* bias, float* out, int B, int H, int W, int C,
                        int dg, int om_stride, int Cout, gssd_stream_t stream);
/* the same with `flags`: GSSD_CONV_F16_OK = x, its bilinear samples and the weights lie inside fp16's range (see the flag) -> fp16 planes, three
 * MFMAs per product (w_packed holds both forms: 3 bf16 + 2 fp16 planes).  Round 6, ABI 8. */
/* csrc/conv_patch_x6.hip (round 6): patch-staged direct 3x3 conv for many input channels and few (<= 128) outputs, fp16 planes (GSSD_CONV_F16_OK
 * launches only).  weight_elems: 16-bit elements of the packed planes (-1: not a shape of the kernel); pack from the K-major fp32 rows. */
long long gssd_conv_patch_x6_weight_elems(int Cout, int C);
int gssd_conv_patch_x6_pack_weight(const float* w_packed, void* out, int Cout, int C, int row_stride, gssd_stream_t stream);
int gssd_conv_patch_x6_takes(const gssd_conv_desc* d);
int gssd_dcn_forward_x6_ex(const float* x, const float* om, const void* w_packed, const float* bias, float* out, int B, int H, int W, int C,
                           int dg, int om_stride, int Cout, int flags, gssd_stream_t stream);
long long gssd_dcn_packed_weight_elems_bf16(int Cout, int C);
int gssd_dcn_pack_weight_bf16(const float* w_oihw, void* w_packed, int Cout, int C, int dg, gssd_stream_t stream);
int gssd_dcn_forward_bf16(const void* x, const float* om, const void* w_packed, const float* bias, void* out, int B, int H, int W,
                          int C, int dg, int om_stride, int Cout, gssd_stream_t stream);

/* Backward of gssd_dcn_im2col_f32 (what the reference gets from the dcn_v2 extension's backward through autograd,
 * layers/dcn_v2_custom.py:84-89): given d(cols), ADDS d(x) into `dx` (fp32 atomics; the caller zero-fills or pre-loads it)
 * and ADDS d(om) ([pixels][om_stride]; offsets' and mask logits' gradients, sigmoid included) into `dom` (zero-filled by
 * the caller).  Channels per deformable group must be a multiple of 64. */
int gssd_dcn_col2im_f32(const float* x, const float* om, const float* dcols, float* dx, float* dom, int B, int H, int W,
                        int C, int dg, int om_stride, gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Input stage: data/__init__.py:33-54 (base_transform_fast / BaseTransform),
 * utils/augmentations.py:506-524 (ResizeFast, Normalize), train_lesion_multiphase_v2.py:198 ([B,4,3,H,W] -> [B,12,H,W]).
 * The resize is Pillow's `Image.resize` on uint8 (third-party; algorithm restated from src/libImaging/Resample.c).
 * ------------------------------------------------------------------------------------------ */
#define GSSD_FILTER_BILINEAR 2 /* PIL.Image.BILINEAR */
#define GSSD_FILTER_BICUBIC 3  /* PIL.Image.BICUBIC: Image.resize's default since Pillow 7.0 */

/* HOST functions: taps per output pixel, and the 22-bit fixed-point coefficient tables (`bounds` [out][2] = first input
 * index, tap count; `kk` [out][ksize]) -- Resample.c precompute_coeffs + normalize_coeffs_8bpc.  The caller uploads them. */
int gssd_resample_ksize(int in_size, int out_size, int filter);
int gssd_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk);

/* One 8-bit resampling pass over n_img interleaved images (device pointers).  horizontal: [n][H][W_in][C] ->
 * [n][H][W_out][C]; vertical: [n][H_in][W][C] -> [n][H_out][W][C].  The vertical pass also folds each study's
 * (imgs_per_study consecutive images) per-channel extrema into minmax[study][C][2] = (max of 255 - v, max of v); zero-fill it
 * first, or pass NULL.  C <= 4. */
int gssd_resize_u8_horizontal(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize,
                              int n_img, int H, int W_in, int W_out, int C, gssd_stream_t stream);
int gssd_resize_u8_vertical(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize, int n_img,
                            int H_in, int H_out, int W, int C, int imgs_per_study, int32_t* minmax, gssd_stream_t stream);

/* u8 [B][phases][S][S][C] -> fp32 NCHW [B][phases*C][S][S] (channel = phase*C + slice): x = v - mean[c]; with
 * `normalize`, (x - min) / (max - min) over the study, min / max taken from `minmax` (see above). */
int gssd_input_finish_f32(const uint8_t* img, const int32_t* minmax, float mean0, float mean1, float mean2, float* out_nchw,
                          int B, int phases, int S, int C, int normalize, gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Self_Attn backward building blocks (what the reference gets from autograd over
 * layers/self_attn.py:62-89 -- conv2d / bmm / softmax backward kernels)
 * ------------------------------------------------------------------------------------------ */
/* Batched fp32 GEMM on the matrix cores with independent transpose flags: C_b[M][N] (+)= alpha * op(A_b)[M][K] . op(B_b)[K][N];
 * op(A)[m][k] = transA ? A[k*lda + m] : A[m*lda + k]; op(B)[k][n] = transB ? B[n*ldb + k] : B[k*ldb + n]; operands `stride*`
 * floats apart.  (dA = d(ag).g, dtheta = dS.phi, dphi = dS^T.theta, dg = A^T.d(ag).) */
int gssd_bgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA, int transB,
                   long long strideA, long long strideB, long long strideC, int batch, float alpha, int accumulate,
                   gssd_stream_t stream);
/* The same GEMM with an epilogue on alpha * acc: mode 1 -> exp(. - rowvec[b][m]) (the attention probabilities from the forward's
 * log-sum-exp: torch.softmax of self_attn.py:72 without a pass over the logits), mode 2 -> aux[b][m][n] * (. - rowvec[b][m]) (softmax
 * backward where dA is produced; aux laid out like C); mode 0 = gssd_bgemm_f32 without accumulation. */
int gssd_bgemm_ex_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA, int transB,
                      long long strideA, long long strideB, long long strideC, int batch, float alpha, int mode, const float* rowvec,
                      const float* aux, gssd_stream_t stream);
/* out[row] = sum_c a[row][c] * b[row][c]  (rowsum(A o dA) = <d(ag)_i, ag_i> of the attention backward) */
int gssd_rowdot_f32(const float* a, const float* b, float* out, int64_t rows, int C, gssd_stream_t stream);
/* Softmax backward over rows, in place: dattn[r][j] <- attn[r][j] * (dattn[r][j] - sum_j attn[r][j]*dattn[r][j]); pad columns zeroed */
int gssd_softmax_bwd_rows_f32(const float* attn, float* dattn, int64_t rows, int n, int row_stride, gssd_stream_t stream);
/* Spectral-norm chain rule (layers/spectral_norm.py:83-85 with u, v constants): dW_eff' = scale[0] * dW_eff (scale may be NULL);
 * dW_orig = dW_eff' * is - <dW_eff', W_orig> * is^2 * u v^T, is = inv_sigma[0].  dw_eff rows are ld_dw floats apart. */
int gssd_sn_weight_grad_f32(const float* dw_eff, int ld_dw, const float* w_orig, const float* u, const float* v, const float* inv_sigma,
                            const float* scale, double* dot_scratch /* one zero-filled double */, float* dw_orig, int rows, int cols,
                            gssd_stream_t stream);
/* out[c][n] = w[n][c] * alpha[n] (alpha may be NULL): data-gradient weights of a spectrally normalised 1x1 conv */
int gssd_scaled_transpose_f32(const float* w, const float* alpha, float* out, int rows, int cols, gssd_stream_t stream);
/* y[c][r] = bf16(x[r][c]) for r < rows, c < cols; x rows ld_x floats apart, y rows ld_y bf16 apart (ld_y >= rows, a multiple of 8).  The
 * bf16 storage mode's weight gradients dW[n][k] = sum_m dY[m][n] A[m][k] run as NT GEMMs over the transposed operands (reduction index
 * m contiguous) on the bf16 matrix cores: gssd_conv2d_nhwc_bf16 with in = dY^T, wgt = A^T, split_k > 1, GSSD_CONV_OUT_F32. */
/* y[r][c] = bf16(x[r][c]) for c < cols and 0 for cols <= c < ld_y (rows widened to a multiple of 8 channels for the bf16 kernels). */
/* hi = bf16(x), lo = bf16(x - hi). */
int gssd_cast_split_f32_bf16(const float* x, void* hi, void* lo, int64_t n, gssd_stream_t stream);
int gssd_cast_rows_f32_bf16(const float* x, void* y, int64_t rows, int cols, int ld_x, int ld_y, gssd_stream_t stream);
int gssd_transpose_cast_f32_bf16(const float* x, void* y, int64_t rows, int cols, int64_t ld_x, int64_t ld_y, gssd_stream_t stream);
/* *out += sum a[i]*b[i] (fp64); out = a*x + b*y; y = scale[0]*x (fp64 -> fp32); d(sigma) = dot[0] + sum_c bias[c]*colsum[c] */
int gssd_dot_f32(const float* a, const float* b, int64_t n, double* out, gssd_stream_t stream);
int gssd_axpby_f32(const float* x, const float* y, float* out, int64_t n, float a, float b, gssd_stream_t stream);
int gssd_scale_cast_f64_f32(const double* x, const float* scale, float* y, int n, gssd_stream_t stream);
int gssd_sa_sigma_grad_f32(const double* dot, const double* colsum, const float* bias, int C, float* dsigma, gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * MultiBoxLoss (layers/modules/multibox_loss.py:46-120, layers/box_utils.py:70-135,160-168)
 * ------------------------------------------------------------------------------------------ */

/* Batched match(): one workgroup per image.  targets [sum n_i][5] (xmin,ymin,xmax,ymax,label) is the
 * concatenation of the per-image ground-truth lists, gt_off [B+1] their row offsets (n_i <= 64).
 * Writes loc_t [B][P][4] and conf_t [B][P] (int64, like the reference's LongTensor). */
int gssd_match_batch(const float* targets, const int* gt_off, const float* priors, int B, int P,
                     float threshold, float var0, float var1, float* loc_t, int64_t* conf_t, gssd_stream_t stream);

/* Global max of a float array (log_sum_exp's x_max, box_utils.py:167) as out_n partial maxima
 * (one per workgroup); the consumer takes the max over them. */
int gssd_reduce_max_f32(const float* x, int64_t n, float* out, int out_n, gssd_stream_t stream);

/* Hard-negative mining + the two loss sums.  sel [B][P] uint8: 0 unused, 1 positive, 2 mined negative.
 * partial [B][4] doubles (loss_l sum, loss_c sum, n_pos, spare); losses [2] floats = (loss_l/N, loss_c/N)
 * after gssd_loss_finalize.  loss_c_all [B][P] (optional, may be NULL) receives the mining scores. */
int gssd_hnm_loss(const float* loc, const float* conf, const float* loc_t, const int64_t* conf_t, const float* xmax,
                  int xmax_n, int B, int P, int C, int negpos_ratio, uint8_t* sel, double* partial, float* loss_c_all,
                  gssd_stream_t stream);
int gssd_loss_finalize(const double* partial, int B, float* losses, double* n_total, gssd_stream_t stream);
/* multibox_loss.py:117 with N taken over every rank's images (opt-in, MultiBoxLoss.global_normalizer): n_global = the all-reduced
 * (summed) n_total of gssd_loss_finalize; losses = world * local sums / n_global, *n_total = n_global / world -- the mean over ranks of these
 * losses, and of the gradients gssd_loss_backward forms with this n_total, equals the single (world x B)-image batch. */
int gssd_loss_finalize_global(const double* partial, int B, const double* n_global, int world, float* losses, double* n_total,
                              gssd_stream_t stream);
/* d(loss_l + loss_c)/d(loc, conf) scaled by grad_l, grad_c (device scalars) / N. */
int gssd_loss_backward(const float* loc, const float* conf, const float* loc_t, const int64_t* conf_t,
                       const uint8_t* sel, const double* n_total, const float* grad_l, const float* grad_c, int B,
                       int P, int C, float* dloc, float* dconf, gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Detect (layers/functions/detection_pytorch_ver_1point5.py:32-89, box_utils.py:139-157,174-238)
 * ------------------------------------------------------------------------------------------
 * One workgroup per (image, class >= 1): threshold, top_k selection, decode, greedy NMS.
 * conf_is_logits != 0 fuses the class softmax (models/...group.py:388); loc_is_boxes != 0 skips decode
 * (rows of `loc` are already x1,y1,x2,y2 -- the box_utils.nms() entry).  out [B][C][top_k][5] is fully
 * written (zero rows included).  keep_idx (optional) [B][C][top_k] int32 prior indices, -1 padded;
 * keep_cnt (optional) [B][C]. */
int gssd_detect(const float* loc, const float* conf, const float* priors, int B, int P, int C, int top_k,
                float conf_thresh, float nms_thresh, float var0, float var1, int conf_is_logits, int loc_is_boxes,
                float* out, int* keep_idx, int* keep_cnt, gssd_stream_t stream);

/* 2-class (C-class) softmax over the last axis, double exp + one rounding (models/...group.py:388). */
int gssd_softmax_lastdim_f32(const float* x, float* y, int64_t rows, int C, gssd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * AP / IoBB evaluator: test_ap_iobb.py:126-148 (make_pred's use of the Detect output), :231-328
 * (test_net: greedy TP / FP, precision / recall), :10-41 (voc_ap)
 * ------------------------------------------------------------------------------------------ */

/* Per image n: walk the class-1 rows det[n*img_stride + 5*d .. +5] = (score, x1, y1, x2, y2) of Detect's output in order;
 * a row is kept iff score > 0 and score > thresh; its box is scaled by scales[n][4] (fp32 multiply) and matched in float64
 * against gt[gt_off[n] .. gt_off[n+1]) ([.][4] pixel boxes, at most `max_gt` <= 256 per image).  Metric m < n_iou uses IoU
 * with threshold thr[m], the next n_iobb use intersection-over-detection-box.  Outputs over M = N*top_k rows:
 * conf_out[row] = score (or -inf for dropped rows); flag_out[m*M + row] = 1 (TP), 2 (FP) or 0 (neither: dropped row, or an
 * image without ground truth -- the reference leaves both counters untouched there). */
int gssd_eval_match(const float* det, long long img_stride, int N, int top_k, const float* scales, const double* gt,
                    const int* gt_off, int max_gt, double thresh, const double* thr, int n_iou, int n_iobb, float* conf_out,
                    uint8_t* flag_out, gssd_stream_t stream);

/* Global ranking + AP.  Sorts the M confidences in descending order (stable: ties keep their row order; the reference's
 * np.argsort is unstable there), accumulates TP / FP per metric and writes ap_out[n_metrics] (device, float64):
 * use_07_metric != 0: the 11-point VOC07 value (bit-exact); else the area under the precision envelope.
 * `workspace` must hold gssd_eval_workspace_bytes(M) bytes (HOST query). */
long long gssd_eval_workspace_bytes(int M);
int gssd_eval_ap(const float* conf, const uint8_t* flags, int M, int n_metrics, double npos, int use_07_metric, void* workspace,
                 long long workspace_bytes, double* ap_out, gssd_stream_t stream);

/* out[b][p][c] = sum_{k < splits[p]} ws[k][b][p][c] (ws slices B*P*C floats apart), in slice order: the deterministic reduction of
 * the multibox heads' split-K slices (GSSD_CONV_HEADS_SLICES).  splits: int8 [P], the slice count of the head that owns prior p. */
int gssd_heads_reduce_f32(const float* ws, const signed char* splits, float* out, int B, int P, int C, gssd_stream_t stream);

/* ---- PixelLink++ tail (ssd_liverdet/pixel_link) ------------------------------------------------------------
 * The trunk / Self_Attn / DCN / fuse / score-head launches are gssd_conv2d_nhwc_f32 & co; these four are the rest. */

/* out = bilinear(src -> Hd x Wd), align_corners=True (F.interpolate calls of pixel_link/model.py:342-411), NHWC [B][H][W][C];
 * with `addend` [B][Hd][Wd][C] also out2 = out + addend (the lateral add of the cascade).  addend and out2 are both NULL or both set. */
int gssd_interp_add_f32(const float* src, const float* addend, float* out, float* out2, int B, int Hs, int Ws, int Hd, int Wd, int C,
                        gssd_stream_t stream);

/* final_1 / final_2 (model.py:118-123,360,386,396,411) over nf (1 = no cascade_fuse, 4 = cascade_fuse, version "4s") feature maps
 * f[k] NHWC [B][HW][18] (channels 0-1 pixel, 2-17 link): out1 NCHW [B][2][HW], out2 NCHW [B][16][HW];
 * w1 [2][2*nf], w2 [16][16*nf] (the Conv2d weights, input channel = k * {2,16} + c as torch.cat(features, dim=1) orders them). */
int gssd_pixellink_final_f32(const float* f0, const float* f1, const float* f2, const float* f3, int nf, const float* w1,
                             const float* b1, const float* w2, const float* b2, float* out1, float* out2, int B, int HW,
                             gssd_stream_t stream);

/* PixelLinkLoss.pixel_loss + link_loss (pixel_link/criterion.py:24-104).  out1 [B][2][H][W], out2 [B][16][H][W] fp32 NCHW;
 * pixel_target [B][H][W] int64 (0/1), neg_pixel_mask [B][H][W] uint8, pixel_pos_weight [B][H][W] fp32, link_target [B][8][H][W] int64.
 * per_image [B][6] fp64 = {pixel_pos, pixel_neg, link_pos, link_neg, area, neg_area}: the reference's four losses are the batch
 * means of the first four columns.  neg_weight_out (optional) [B][H][W] = the mined-negative mask (criterion.py:47-48).
 * H*W <= 8192.  An image without a negative candidate is an IndexError in the reference; here its pixel_neg term is 0. */
int gssd_pixellink_loss_f32(const float* out1, const float* out2, const long long* pixel_target, const unsigned char* neg_pixel_mask,
                            const float* pixel_pos_weight, const long long* link_target, double* per_image, float* neg_weight_out,
                            int B, int H, int W, int neg_pos_ratio, gssd_stream_t stream);

/* Backward of the tail (training: forward -> PixelLinkLoss -> loss.backward()).  Gradient maps of the 18-channel score maps are NHWC
 * with a channel stride ld >= 18 (channels 18 .. ld-1 are never written: the caller zeroes them once).
 * gssd_interp_add_bwd_f32: d(src) += interp^T(d(out) + d(out2)) (atomics: zero d(src) first or let it hold another contribution),
 * d(addend) += d(out2); d_out or d_out2 may be NULL.
 * gssd_pixellink_final_bwd_f32: g[k] (= or += per bit k of accumulate_mask) the gradient of feature k; dw1 fp64 [2*2*nf + 2] = final_1's
 * weight rows then its bias, dw2 fp64 [16*16*nf + 16] likewise (accumulated with atomics: zero them first).
 * gssd_pixellink_loss_bwd_f32: d(out1), d(out2) of sum_k upstream4[k] * loss_k for the four means gssd_pixellink_loss_f32 defines; neg_weight
 * and per_image are that launch's outputs (the mined-negative mask, areas and link weight sums are constants under autograd). */
int gssd_interp_add_bwd_f32(const float* d_out, const float* d_out2, float* d_src, float* d_addend, int B, int Hs, int Ws, int Hd, int Wd,
                            int C, int ld, gssd_stream_t stream);
int gssd_pixellink_final_bwd_f32(const float* d_out1, const float* d_out2, const float* f0, const float* f1, const float* f2,
                                 const float* f3, int nf, const float* w1, const float* w2, float* g0, float* g1, float* g2, float* g3,
                                 int accumulate_mask, double* dw1, double* dw2, int B, int HW, int ld, gssd_stream_t stream);
int gssd_pixellink_loss_bwd_f32(const float* out1, const float* out2, const long long* pixel_target, const float* neg_weight,
                                const float* pixel_pos_weight, const long long* link_target, const double* per_image,
                                const float* upstream4, float* d_out1, float* d_out2, int B, int H, int W, gssd_stream_t stream);

/* Link decoding (pixel_link/postprocess.py:104-121 thresholds, :178-234 `func`): labels [B][H][W] int32 = 0 for background, else
 * 1 + rank of the pixel's connected component by its first pixel in raster order (the reference's root_map numbering; the reference
 * stores it as uint8 and wraps beyond 255 components, this does not).  comps [B][max_comp][6] = {pixel count, min x, min y, max x,
 * max y, sum of the positive-class probability} for the first min(max_comp, 1024) components (the rest: count 0); ncomp [B].
 * H*W <= 8192. */
int gssd_pixellink_decode_f32(const float* out1, const float* out2, int* labels, float* comps, int* ncomp, int B, int H, int W,
                              float pixel_thr, float link_thr, int max_comp, gssd_stream_t stream);

/* The fp32 Self_Attn core on the BF16 matrix cores with fp32-equivalent products (csrc/flash_attn_x6.hip; layers/self_attn.py:68-80): the
 * contract of gssd_self_attn_core_f32 (tp [B][N][2D] fp32 theta | phi, gT [B][C2][Np] fp32, out [B][N][C2] fp32, optional lse [B][N]) with both
 * products as six v_mfma_f32_16x16x32_bf16 over operands split into three bf16 planes.  `ws`: gssd_self_attn_core_x6_ws_bytes(B, N, D, C2) bytes
 * of scratch (the planes of theta | phi and of g^T, written by a first pass of the call), 16-byte aligned.  (D, C2) = (64, 256) (the 38 x 38 maps):
 * gssd_self_attn_core_x6_supported; anything else is GSSD_EINVAL (the caller keeps gssd_self_attn_core_f32). */
int gssd_self_attn_core_x6_supported(int D, int C2);
long long gssd_self_attn_core_x6_ws_bytes(int B, int N, int D, int C2);
int gssd_self_attn_core_x6_f32(const float* tp, const float* gT, float* out, int B, int N, int Np, int D, int C2, void* ws, float* lse,
                               gssd_stream_t stream);

/* ---- launch-plan runner (csrc/plan_run.hip) ----------------------------------------------------------------------------------------------
 * The host side of the reference enqueues one kernel per Python call (nn.Module.__call__ -> ATen, train_lesion_multiphase_v2.py:242-253);
 * this build's engine knows the whole static launch list of a step, so a SEGMENT of it (everything between two points where the host must
 * act: a gradient-segment hook of the data-parallel reducer, a host-side tensor op) goes down in ONE call.
 * An op is either a LAUNCH of any entry point of this header whose last parameter is the stream -- `fn` = gssd_plan_fn_index("gssd_..."),
 * `args` = its other arguments in order as 64-bit words (pointers and integers by value, sign-extended; float as its 32 bits, double as its
 * 64 bits), `stream` = index into the `streams` array of the call -- or a WAIT: streams[stream] waits for everything enqueued so far on
 * streams[fn] (event record + hipStreamWaitEvent; fn == stream is a no-op).  Ops run in array order; the first failing op's index is stored
 * in *failed_at (else -1) and its code returned.  Pointers inside args (descriptors included) must stay valid for the call only. */
#define GSSD_PLAN_LAUNCH 0
#define GSSD_PLAN_WAIT 1
#define GSSD_PLAN_MAX_ARGS 24
typedef struct gssd_plan_op {
    int kind;   /* GSSD_PLAN_LAUNCH / GSSD_PLAN_WAIT */
    int fn;     /* LAUNCH: function index; WAIT: index of the awaited stream */
    int stream; /* index into `streams` */
    int nargs;  /* LAUNCH: number of words used in args (checked against the function's parameter count) */
    uint64_t args[GSSD_PLAN_MAX_ARGS];
} gssd_plan_op;
int gssd_plan_fn_count(void);
const char* gssd_plan_fn_name(int index);
int gssd_plan_fn_index(const char* name); /* -1: not an entry point the runner can launch */
int gssd_plan_fn_nargs(int index);        /* parameters before the stream */
int gssd_plan_op_size(void);              /* sizeof(gssd_plan_op) as built */
/* Threading: callable from several host threads at once (the events behind WAIT ops come from a per-device ring whose slots are handed
   out, recorded and waited on under a mutex); the ring is chosen by the device that owns the awaited stream, so the caller's current
   device need not be the plan's.  The launches themselves need the plan's device current, as every other entry point does. */
int gssd_plan_run(const gssd_plan_op* ops, int n_ops, const gssd_stream_t* streams, int n_streams, int* failed_at);

/* 1 when gssd_conv2d_nhwc_f32 runs the descriptor on the three-plane Winograd kernel (csrc/conv_wino_x6.hip; on by default above its size gate; GSSD_WINO_X6=0 never, =2 every shape it can take) */
int gssd_conv_wino_x6_takes(const gssd_conv_desc* d);
/* 1 when gssd_conv2d_nhwc_f32 runs the descriptor on the patch-staged three-plane direct conv (csrc/conv_thin_x6.hip, round 6): the grouped 3x3
 * trunk layers with 16 -> 16, 16 -> 32 and 32 -> 32 channels per group on maps of >= 75 x 75 pixels -- conv1_2, conv2_1, conv2_2 of
 * models/ssd_multiphase_custom_group.py:434-460 -- plain or with the fused producer BatchNorm + ReLU / batch sums / GSSD_CONV_POOL2 epilogue
 * (no residual: their data gradients stay with the fp32 kernels).  GSSD_THIN_X6=0: never. */
int gssd_conv_thin_x6_takes(const gssd_conv_desc* d);

#ifdef __cplusplus
}
#endif
#endif /* GSSD_HIP_H */
