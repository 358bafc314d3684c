// bf16-storage variants of the HBM-bound passes (BASELINE.json configs[4]): bf16 NHWC in and out, all arithmetic in fp32,
// one rounding on store.  16-byte accesses = 8 channels per lane.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

namespace {

constexpr int EW_THREADS = 256;
inline int ew_blocks(long long work_items, int per_block = EW_THREADS, int cap = 8192) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }

// NCHW fp32 -> NHWC bf16 with each group's channels padded from cpg_in to a multiple of 8 (zeros): one 16-byte store per (pixel, group,
// 8-channel piece) -- one piece for 4 and 2 groups (3 and 6 channels), two for the ungrouped 12-channel input
__global__ void pack_input_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y, int B, int C, int HW, int groups, int cpg_in) {
    const int pieces = (cpg_in + 7) >> 3;
    const long long total = (long long)B * HW * groups * pieces;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pc = (int)(i % pieces);
        const long long ig = i / pieces;
        const int g = (int)(ig % groups);
        const long long bp = ig / groups;
        const int pix = (int)(bp % HW);
        const int b = (int)(bp / HW);
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cc = pc * 8 + c;
            o[c] = (__bf16)((cc < cpg_in) ? x[((long long)b * C + g * cpg_in + cc) * HW + pix] : 0.f);
        }
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
}

__global__ __launch_bounds__(256) void bn_relu_pool_bf16_kernel(
    const u16* __restrict__ raw, u16* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo, int pk, int ps, int pp,
    const double* __restrict__ stats, double count, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* running_mean, float* running_var, float momentum, float eps, int training, int relu, int srep) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_scale = sm;
    float* s_shift = sm + C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (gamma == nullptr) {
            s_scale[c] = 1.f;
            s_shift[c] = 0.f;
            continue;
        }
        double mean, var;
        if (training) {
            mean = gssd_stats_sum(stats, c, 2 * C, srep) / count;
            var = gssd_stats_sum(stats, C + c, 2 * C, srep) / count - mean * mean;
            if (var < 0.0) var = 0.0;
            if (blockIdx.x == 0) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
                running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
            }
        } else {
            mean = (double)running_mean[c];
            var = (double)running_var[c];
        }
        // gamma * (1 / sqrt(var + eps)) exactly as gssd_bn_finalize_* computes it: a layer gives the same activations whether its
        // BatchNorm runs in this pass or deferred in its consumer (pooled and unpooled plans agree bit for bit)
        const double inv = 1.0 / sqrt(var + (double)eps);
        const double sc = (double)gamma[c] * inv;
        s_scale[c] = (float)sc;
        s_shift[c] = (float)((double)beta[c] - mean * sc);
    }
    __syncthreads();
    const int C8 = C >> 3;
    const long long total = (long long)B * Ho * Wo * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long long t = i / C8;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float sc[8], sh[8], r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = s_scale[8 * c8 + e];
            sh[e] = s_shift[8 * c8 + e];
        }
        if (pk == 0) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(raw + (((long long)b * H + yo) * W + xo) * C + 8 * c8);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (float)v[e] * sc[e] + sh[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = -INFINITY;
            const int y0 = yo * ps - pp, x0 = xo * ps - pp;
            for (int dy = 0; dy < pk; ++dy) {
                const int yy = y0 + dy;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int dx = 0; dx < pk; ++dx) {
                    const int xx = x0 + dx;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(raw + (((long long)b * H + yy) * W + xx) * C + 8 * c8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = fmaxf(r[e], (float)v[e] * sc[e] + sh[e]);
                }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)(relu ? fmaxf(r[e], 0.f) : r[e]);
        *reinterpret_cast<bf16x8*>(out + i * 8) = o;
    }
}

__global__ void bn_finalize_bf16_kernel(const double* __restrict__ stats, double count, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float* running_mean, float* running_var, float momentum,
                                        float eps, int training, int C, float* __restrict__ scale, float* __restrict__ shift,
                                        u16* __restrict__ pad, int srep) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) {
        mean = gssd_stats_sum(stats, c, 2 * C, srep) / count;
        var = gssd_stats_sum(stats, C + c, 2 * C, srep) / count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (training == 1) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        }
    } else {
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    }
    double sc = (double)gamma[c] * (1.0 / sqrt(var + (double)eps));      // (the pool pass's form)
    const float sh = (float)((double)beta[c] - mean * sc);
    float scf = (float)sc;
    if (scf == 0.f) scf = 1e-30f;
    scale[c] = scf;
    shift[c] = sh;
    pad[c] = f2bf(scf > 0.f ? -3.0e38f : 3.0e38f);      // max(pad*scale + shift, 0) == 0: zero padding after BN + ReLU
}

// L2Norm: one wave per pixel, 8 channels per lane and step
__global__ __launch_bounds__(256) void l2norm_bf16_kernel(const u16* __restrict__ x, const float* __restrict__ w, u16* __restrict__ out,
                                                          long long pixels, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int C8 = C >> 3;
    for (long long p = wave0; p < pixels; p += nwaves) {
        const bf16x8* xp = reinterpret_cast<const bf16x8*>(x + p * C);
        float ss = 0.f;
        for (int c = lane; c < C8; c += 64) {
            const bf16x8 v = xp[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += (float)v[e] * (float)v[e];
        }
        ss = wave_sum(ss);
        const float inv = 1.f / (sqrtf(ss) + eps);
        bf16x8* op = reinterpret_cast<bf16x8*>(out + p * C);
        for (int c = lane; c < C8; c += 64) {
            const bf16x8 v = xp[c];
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)(w[8 * c + e] * ((float)v[e] * inv));
            op[c] = o;
        }
    }
}

}  // namespace

extern "C" int gssd_pack_input_nhwc_bf16(const float* x_nchw, void* y_nhwc, int B, int C, int H, int W, int groups,
                                         gssd_stream_t stream) {
    GSSD_CHECK_ARG(x_nchw && y_nhwc && B > 0 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0);
    hipLaunchKernelGGL(pack_input_bf16_kernel, dim3(ew_blocks((long long)B * H * W * groups * ((C / groups + 7) / 8))), dim3(EW_THREADS), 0, as_stream(stream),
                       x_nchw, reinterpret_cast<u16*>(y_nhwc), B, C, H * W, groups, C / groups);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_relu_pool_bf16(const void* raw, void* out, int B, int H, int W, int C, int Ho, int Wo, int pool_k, int pool_s,
                                      int pool_p, const double* stats, double count, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, int training, int relu,
                                      int stats_rep, gssd_stream_t stream) {
    GSSD_CHECK_ARG(stats_rep >= 0 && raw && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 4096 && Ho > 0 && Wo > 0);
    GSSD_CHECK_ARG(gamma == nullptr || (beta && running_mean && running_var));
    GSSD_CHECK_ARG(!(training && gamma) || (stats != nullptr && count > 0));
    if (pool_k == 0) GSSD_CHECK_ARG(Ho == H && Wo == W);
    else GSSD_CHECK_ARG(pool_s > 0 && pool_p >= 0 && (Ho - 1) * pool_s - pool_p < H && (Wo - 1) * pool_s - pool_p < W);
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(bn_relu_pool_bf16_kernel, dim3(ew_blocks(total, EW_THREADS * 4, 2048)), dim3(EW_THREADS), 2 * C * sizeof(float),
                       as_stream(stream), reinterpret_cast<const u16*>(raw), reinterpret_cast<u16*>(out), B, H, W, C, Ho, Wo, pool_k,
                       pool_s, pool_p, stats, count, gamma, beta, running_mean, running_var, momentum, eps, training, relu, stats_rep);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_finalize_bf16(const double* stats, double count, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float momentum, float eps, int training, int C, float* scale, float* shift,
                                     void* pad_bf16, int stats_rep, gssd_stream_t stream) {
    GSSD_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && pad_bf16 && C > 0);
    GSSD_CHECK_ARG(!training || (stats != nullptr && count > 0));
    hipLaunchKernelGGL(bn_finalize_bf16_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), stats, count, gamma, beta,
                       running_mean, running_var, momentum, eps, training, C, scale, shift, reinterpret_cast<u16*>(pad_bf16), stats_rep);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_l2norm_bf16(const void* x, const float* weight, void* out, int64_t pixels, int C, float eps, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && weight && out && pixels > 0 && C > 0 && C % 8 == 0);
    hipLaunchKernelGGL(l2norm_bf16_kernel, dim3(ew_blocks(pixels, 4, 4096)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const u16*>(x), weight, reinterpret_cast<u16*>(out), (long long)pixels, C, eps);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
