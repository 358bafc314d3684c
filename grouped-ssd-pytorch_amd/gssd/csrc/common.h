// Shared helpers for the gfx950 kernels of libgssd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "gssd_hip.h"

void gssd_set_error(const char* fmt, ...);

#define GSSD_CHECK_ARG(cond)                                                          \
    do {                                                                              \
        if (!(cond)) {                                                                \
            gssd_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
            return GSSD_EINVAL;                                                       \
        }                                                                             \
    } while (0)

#define GSSD_CHECK_LAUNCH()                                                                     \
    do {                                                                                        \
        hipError_t e__ = hipGetLastError();                                                     \
        if (e__ != hipSuccess) {                                                                \
            gssd_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return GSSD_ELAUNCH;                                                                \
        }                                                                                       \
    } while (0)

static inline hipStream_t as_stream(gssd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: every call site keeps one "done" bit per device id, so a
// process that touches several GPUs (tests on cuda:1, one-process multi-device callers) never launches a > 48 KB-LDS kernel
// without it.  Launches come from two host threads (the forward from the caller's, the backward from autograd's): the bit is
// published (gssd_attr_done, release) only AFTER hipFuncSetAttribute has returned success, so a thread that sees it set may
// launch; two threads racing on an unset bit both set the attribute (idempotent), and a failed call is retried by the next launch.
static inline bool gssd_attr_needed(unsigned* mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32) return true;
    return (__atomic_load_n(mask, __ATOMIC_ACQUIRE) & (1u << dev)) == 0;
}
static inline void gssd_attr_done(unsigned* mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 32) (void)__atomic_fetch_or(mask, 1u << dev, __ATOMIC_RELEASE);
}

// BatchNorm batch sums (gssd_conv_desc::stats) may be kept in `rep` replicas of [2 * Cout] doubles: a workgroup adds into replica
// (workgroup id mod rep), the consumers (gssd_bn_finalize_*, gssd_bn_relu_pool_*, gssd_bn_bwd_finalize_f32) add the replicas up in a
// fixed order.  Why: device-scope fp64 atomics on one cache line are served one after the other (~8 ns each, measured round 4); the
// persistent trunk kernels flush all their workgroups' sums at the END of the launch -- 131 k atomics on 16 lines = a 60 us serial
// tail on a 140 us kernel (profiles/r04_thin_knockout.txt).  16 - 32 replicas make the tail 2 - 4 us.
__device__ __forceinline__ double* gssd_stats_replica(double* stats, int rep, int cout) {
    return rep > 1 ? stats + (size_t)((blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z) % (unsigned)rep) * (2 * (size_t)cout) : stats;
}
__device__ __forceinline__ double gssd_stats_sum(const double* stats, int idx, int two_c, int rep) {
    double s = stats[idx];
    for (int r = 1; r < rep; ++r) s += stats[idx + (size_t)r * two_c];
    return s;
}

// 64-lane wavefront reductions (gfx950: wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// conv_thin.hip: returns 1 when the descriptor is not one of the thin grouped 3x3 shapes, else a GSSD_* code
int gssd_try_conv_thin(const gssd_conv_desc& d, hipStream_t stream);
// conv_wino.hip: returns 1 when the descriptor is not a Winograd shape / has no transformed weights
int gssd_try_conv_wino(const gssd_conv_desc& d, hipStream_t stream);
int gssd_try_conv_x6(const gssd_conv_desc& d, hipStream_t stream);      // csrc/conv_x6.hip: 1 = not taken
// conv_wino_x6.hip: Winograd with three-plane bf16 operands; its U planes are stored behind the fp32 U of gssd_winograd_weight_f32
long long gssd_wino_x6_plane_elems(int cout_g, int groups, int cin_g);  // bf16 elements (0: not a shape it takes)
int gssd_wino_x6_pack(const float* w_packed, void* Ux, int Cout, int groups, int cin_g, int row_stride, hipStream_t stream);
bool gssd_wino_x6_enabled();                                             // GSSD_WINO_X6=0 switches it off
bool gssd_wino_x6_wanted(const gssd_conv_desc& d);                        // the shapes it takes by default (GSSD_WINO_X6=2: all it can)
int gssd_launch_conv_wino_x6(const gssd_conv_desc& d, const void* Ux, hipStream_t stream);
// conv_patch_x6.hip: dense 3x3 convs with many input channels and <= 128 outputs, fp16 planes (GSSD_CONV_F16_OK launches with wgt_patch); else 1
int gssd_try_conv_patch_x6(const gssd_conv_desc& d, hipStream_t stream);
// conv_thin_x6.hip: conv1_2 / conv2_1 / conv2_2 shape classes on the bf16 matrix cores with three-plane operands; else returns 1
int gssd_try_conv_thin_x6(const gssd_conv_desc& d, hipStream_t stream);
// conv_thin_wino.hip: conv1_2's shape class (4 x 16 -> 16 channels, large map) with Winograd weights; else returns 1
int gssd_try_conv_thin_wino(const gssd_conv_desc& d, hipStream_t stream);
int gssd_try_conv_thin_wgrad(const gssd_conv_desc& d, const float* dy, float* dw, hipStream_t stream);
// conv_patch_wgrad.hip: conv2_1 .. conv3_3 shapes (patch-staged, one phase group per workgroup); else returns 1
int gssd_try_conv_patch_wgrad(const gssd_conv_desc& d, const float* dy, float* dw, hipStream_t stream);
// gemm_slot.hip: large plain 1x1 convs / GEMMs as a slot-scheduled 128 x 256 MFMA stream; returns 1 for every other shape
int gssd_try_gemm_slot(const gssd_conv_desc& d, hipStream_t stream);
// wgrad_slot.hip: weight gradient of large plain 1x1 convs (and the DCN contraction) as a slot-scheduled TN GEMM; else returns 1
int gssd_try_wgrad_slot(const gssd_conv_desc& d, const float* dy, float* dw, hipStream_t stream);
// conv_thin_bf16.hip: bf16 thin trunk layers (conv1_1 .. conv2_2); returns 1 when the descriptor is not one of them
int gssd_try_conv_thin_bf16(const gssd_conv_desc& d, hipStream_t stream);
// conv_flat_bf16.hip: bf16 grouped 3x3 trunk layers with 32 .. 128 channels per group (conv3_1 .. conv6); returns 1 when not one of them
int gssd_try_conv_flat_bf16(const gssd_conv_desc& d, hipStream_t stream);
