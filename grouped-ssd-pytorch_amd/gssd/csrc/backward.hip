// Backward of the HBM-bound passes of the GSSD trunk on gfx950: BatchNorm(train) + ReLU + max-pool, L2Norm, the heads'
// gradient gather, the stride-2 zero-insertion and per-channel column sums.  NHWC float4 everywhere; per-channel
// reductions are fp32 per thread -> LDS -> one fp64 atomic per channel per workgroup.
// Replaces autograd's native_batch_norm_backward / threshold_backward / max_pool2d_with_indices_backward chain behind
// loss.backward() (train_lesion_multiphase_v2.py:247-248).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef unsigned short u16;
#ifndef BN_UNR
#define BN_UNR 2
#endif
#ifndef BN_OCC
#define BN_OCC 8          // waves per SIMD the streaming BatchNorm-backward kernels are compiled for (8 = two 1024-thread workgroups per CU)
#endif

// four consecutive channels of a stored map: fp32, or bf16 as the bf16 storage mode keeps it (the mixed-precision backward reads the
// forward's own maps instead of fp32 copies of them)
template <typename RT>
__device__ __forceinline__ f32x4 ld4(const RT* p);
template <>
__device__ __forceinline__ f32x4 ld4<float>(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 ld4<u16>(const u16* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                 __uint_as_float(v.y & 0xffff0000u)};
}
__device__ __forceinline__ void st4_bf16(u16* p, const f32x4 v) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x4*>(p) = h;
}

// -----------------------------------------------------------------------------------------------------------------
// Pass 1: window-centric.  One thread per (pooled output position, channel quad).  z = raw*scale + shift, a = relu(z);
// the window's first maximum of a receives d_out; ReLU's mask (z > 0) is applied; dz is written for every window
// element (non-overlapping pools) or atomically accumulated (overlapping, dz pre-zeroed).  Per-channel sums
// s1 = sum dz, s2 = sum dz*raw feed the BatchNorm parameter gradients.
// -----------------------------------------------------------------------------------------------------------------
// POOL = false (pool_k == 0) is its own instantiation: the window code's registers (71) kept the streaming form at ONE 1024-thread
// workgroup per CU; alone it fits 64 registers = two workgroups (8 waves per SIMD)
template <typename RT, typename DT, bool POOL>
__global__ __launch_bounds__(1024, POOL ? 4 : BN_OCC) void bn_bwd_reduce_kernel(const DT* __restrict__ dout, const RT* __restrict__ raw,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ dz, double* __restrict__ sums, int B, int H,
                                                            int W, int C, int Ho, int Wo, int pk, int ps, int pp, int relu,
                                                            u16* __restrict__ dzb = nullptr) {       // dzb: non-overlapping pools may
    // hand the routed gradient to the apply pass as a bf16 map instead of dz (half the bytes of that round trip)
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [C] s1 | [C] s2
    const int C4 = C >> 2;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
    __syncthreads();
    const long long total = (long long)B * Ho * Wo * C4;
    // a thread keeps the same channel quad across its grid-stride iterations when the stride is a multiple of C4
    const long long stride = ((long long)gridDim.x * blockDim.x / C4) * C4;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
    const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int c4 = (int)(i0 % C4);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale && i0 < stride) {
        sc = *reinterpret_cast<const f32x4*>(scale + 4 * c4);
        sh = *reinterpret_cast<const f32x4*>(shift + 4 * c4);
    }
    // no pooling: a streaming pass -- BN_UNR independent iterations' loads in flight per thread (two 8 / 16-byte loads per iteration did not
    // cover the HBM latency at 32 waves per CU: ~60 % of the copy rate; 4 iterations cost half the resident waves: 2 keeps them)
    auto plain = [&](long long i, const f32x4 g, const f32x4 rv) {
        const size_t o = (size_t)i * 4;                   // Ho == H, Wo == W: dout and raw share one dense layout
        const f32x4 z = rv * sc + sh;
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = (!relu || z[e] > 0.f) ? g[e] : 0.f;
        if (dz) *reinterpret_cast<f32x4*>(dz + o) = d;      // (sums only: the apply pass re-derives dz from dout)
        a1 += d;
        a2 += d * rv;
    };
    if (!POOL && i0 < stride) {
        long long i = i0;
        for (; i + (BN_UNR - 1) * stride < total; i += BN_UNR * stride) {
            f32x4 g[BN_UNR], rv[BN_UNR];
#pragma unroll
            for (int j = 0; j < BN_UNR; ++j) {
                g[j] = ld4<DT>(dout + (i + j * stride) * 4);
                rv[j] = ld4<RT>(raw + (i + j * stride) * 4);
            }
#pragma unroll
            for (int j = 0; j < BN_UNR; ++j) plain(i + j * stride, g[j], rv[j]);
        }
        for (; i < total; i += stride) plain(i, ld4<DT>(dout + i * 4), ld4<RT>(raw + i * 4));
    }
    if (POOL && i0 < stride)
        for (long long i = i0; i < total; i += stride) {
            const f32x4 g = ld4<DT>(dout + i * 4);
            {
                // pixel index fits 32 bits (checked by the launcher): 32-bit divisions instead of three 64-bit ones
                unsigned t = (unsigned)(i / C4);
                const int xo = (int)(t % (unsigned)Wo);
                t /= (unsigned)Wo;
                const int yo = (int)(t % (unsigned)Ho);
                const int b = (int)(t / (unsigned)Ho);
                const int y0 = yo * ps - pp, x0 = xo * ps - pp;
                f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bi[4] = {-1, -1, -1, -1};
                f32x4 braw = {0.f, 0.f, 0.f, 0.f};
                for (int dy = 0; dy < pk; ++dy) {
                    const int yy = y0 + dy;
                    if ((unsigned)yy >= (unsigned)H) continue;
                    for (int dx = 0; dx < pk; ++dx) {
                        const int xx = x0 + dx;
                        if ((unsigned)xx >= (unsigned)W) continue;
                        const f32x4 rv = ld4<RT>(raw + (((size_t)b * H + yy) * W + xx) * C + 4 * c4);
                        f32x4 a = rv * sc + sh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (relu) a[e] = fmaxf(a[e], 0.f);
                            if (a[e] > best[e]) {       // first maximum wins, like max_pool2d
                                best[e] = a[e];
                                bi[e] = dy * pk + dx;
                                braw[e] = rv[e];
                            }
                        }
                    }
                }
                const bool overlap = ps < pk;
                for (int dy = 0; dy < pk; ++dy) {
                    const int yy = y0 + dy;
                    if ((unsigned)yy >= (unsigned)H) continue;
                    for (int dx = 0; dx < pk; ++dx) {
                        const int xx = x0 + dx;
                        if ((unsigned)xx >= (unsigned)W) continue;
                        f32x4 d;
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e] = (bi[e] == dy * pk + dx && (!relu || best[e] > 0.f)) ? g[e] : 0.f;
                        const size_t oi = (((size_t)b * H + yy) * W + xx) * C + 4 * c4;
                        float* o = dz + oi;
                        if (!overlap) {
                            if (dzb) st4_bf16(dzb + oi, d);
                            else *reinterpret_cast<f32x4*>(o) = d;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (d[e] != 0.f) unsafeAtomicAdd(o + e, d[e]);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = (bi[e] >= 0 && (!relu || best[e] > 0.f)) ? g[e] : 0.f;
                    a1[e] += d;
                    a2[e] += d * braw[e];
                }
            }
        }
    if (sums) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&sm[4 * c4 + e], a1[e]);
            atomicAdd(&sm[C + 4 * c4 + e], a2[e]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) unsafeAtomicAdd(sums + c, (double)sm[c]);
    }
}

// per-channel constants of draw = A*dz + Bc*raw + Cc and the BatchNorm parameter gradients
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ fstats, double count, const double* __restrict__ bsums,
                                       const float* __restrict__ gamma, float eps, int C, float* __restrict__ coefA,
                                       float* __restrict__ coefB, float* __restrict__ coefC, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int srep) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mean = gssd_stats_sum(fstats, c, 2 * C, srep) / count;
    double var = gssd_stats_sum(fstats, C + c, 2 * C, srep) / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    const double s1 = bsums[c], s2 = bsums[C + c];
    const double dg = inv * (s2 - mean * s1);        // sum dz * xhat
    const double g = (double)gamma[c];
    dgamma[c] = (float)dg;
    dbeta[c] = (float)s1;
    coefA[c] = (float)(g * inv);
    coefB[c] = (float)(-g * inv * inv * dg / count);
    coefC[c] = (float)(-g * inv * s1 / count + g * inv * inv * mean * dg / count);
}

// ``dsrc`` (optional): d(out) of a layer without pooling -- dz is then re-derived here as dsrc * [raw * scale + shift > 0] instead of
// being written by the reduce pass and read back (one HBM pass less per layer).  ``dz16`` (optional): the result as bf16, what the bf16
// data-gradient conv and weight gradient read; ``store_f32`` = 0 leaves ``dz`` as it was (only read, when dsrc is NULL).
template <typename RT, typename DT = float>
__global__ __launch_bounds__(1024, BN_OCC) void bn_bwd_apply_kernel(float* __restrict__ dz, const RT* __restrict__ raw,
                                                           const float* __restrict__ coefA, const float* __restrict__ coefB,
                                                           const float* __restrict__ coefC, long long pixels, int C,
                                                           double* __restrict__ colsum, const DT* __restrict__ dsrc,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int relu, u16* __restrict__ dz16, int store_f32) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C4 = C >> 2;
    if (colsum) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) sm[c] = 0.f;
        __syncthreads();
    }
    const long long total = pixels * C4;
    const long long stride = ((long long)gridDim.x * blockDim.x / C4) * C4;
    const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int c4 = (int)(i0 % C4);
    if (i0 < stride) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(coefA + 4 * c4);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(coefB + 4 * c4);
        const f32x4 cc = *reinterpret_cast<const f32x4*>(coefC + 4 * c4);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (dsrc && scale) {
            sc = *reinterpret_cast<const f32x4*>(scale + 4 * c4);
            sh = *reinterpret_cast<const f32x4*>(shift + 4 * c4);
        }
        auto one = [&](long long i, const f32x4 rv, const f32x4 gin) {
            f32x4 d = gin;
            if (dsrc) {
                const f32x4 z = rv * sc + sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = (!relu || z[e] > 0.f) ? gin[e] : 0.f;
            }
            const f32x4 o = a * d + bb * rv + cc;
            if (store_f32) *reinterpret_cast<f32x4*>(dz + i * 4) = o;
            if (dz16) st4_bf16(dz16 + i * 4, o);
            acc += o;
        };
        auto gload = [&](long long i) -> f32x4 { return dsrc ? ld4<DT>(dsrc + i * 4) : *reinterpret_cast<const f32x4*>(dz + i * 4); };
        long long i = i0;
        for (; i + (BN_UNR - 1) * stride < total; i += BN_UNR * stride) {      // BN_UNR iterations' loads in flight (see bn_bwd_reduce_kernel)
            f32x4 rv[BN_UNR], g[BN_UNR];
#pragma unroll
            for (int j = 0; j < BN_UNR; ++j) {
                rv[j] = ld4<RT>(raw + (i + j * stride) * 4);
                g[j] = gload(i + j * stride);
            }
#pragma unroll
            for (int j = 0; j < BN_UNR; ++j) one(i + j * stride, rv[j], g[j]);
        }
        for (; i < total; i += stride) one(i, ld4<RT>(raw + i * 4), gload(i));
    }
    if (colsum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&sm[4 * c4 + e], acc[e]);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += blockDim.x) unsafeAtomicAdd(colsum + c, (double)sm[c]);
    }
}

// column sums of a dense [rows][C] matrix (conv bias gradients)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, long long rows, int C, int stride,
                                                     double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    for (int c = threadIdx.x; c < C; c += blockDim.x) sm[c] = 0.f;
    __syncthreads();
    const long long total = rows * C;
    const long long gstride = ((long long)gridDim.x * blockDim.x / C) * C;
    const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i0 < gstride) {
        const int c = (int)(i0 % C);
        float a = 0.f;
        for (long long i = i0; i < total; i += gstride) a += x[(i / C) * stride + c];
        atomicAdd(&sm[c], a);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) unsafeAtomicAdd(out + c, (double)sm[c]);
}

__global__ void cast_f64_f32_kernel(const double* __restrict__ x, float* __restrict__ y, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = accumulate ? y[i] + (float)x[i] : (float)x[i];
}

// L2Norm backward: one wave per pixel
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ dy, float* __restrict__ dx,
                                                         const float* __restrict__ dx_add, double* __restrict__ dw,
                                                         long long pixels, int C, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [C] dw partial
    for (int c = threadIdx.x; c < C; c += blockDim.x) sm[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int C4 = C >> 2;
    for (long long p = wave0; p < pixels; p += nwaves) {
        const f32x4* xp = reinterpret_cast<const f32x4*>(x + p * C);
        const f32x4* gp = reinterpret_cast<const f32x4*>(dy + p * C);
        float ss = 0.f, dot = 0.f;
        for (int c = lane; c < C4; c += 64) {
            const f32x4 v = xp[c], g = gp[c], ww = reinterpret_cast<const f32x4*>(w)[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ss += v[e] * v[e];
                dot += ww[e] * g[e] * v[e];
            }
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float rr = sqrtf(ss), n = rr + eps;
        const float k1 = 1.f / n, k2 = (rr > 0.f) ? dot / (n * n * rr) : 0.f;
        for (int c = lane; c < C4; c += 64) {
            const f32x4 v = xp[c], g = gp[c], ww = reinterpret_cast<const f32x4*>(w)[c];
            f32x4 o = ww * g * k1 - v * k2;
            if (dx_add) o += reinterpret_cast<const f32x4*>(dx_add + p * C)[c];
            reinterpret_cast<f32x4*>(dx + p * C)[c] = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&sm[4 * c + e], g[e] * v[e] * k1);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) unsafeAtomicAdd(dw + c, (double)sm[c]);
}

// heads: dloc [B,P,4] / dconf [B,P,nc] -> dense NHWC [B, HW, A*(4+nc)] of one source
__global__ void heads_gather_kernel(const float* __restrict__ dloc, const float* __restrict__ dconf, float* __restrict__ out,
                                    int B, int HW, int A, int nc, int P, int prior_off) {
    const int Cc = A * (4 + nc);
    const long long total = (long long)B * HW * Cc;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % Cc);
        const long long bp = i / Cc;
        const int pix = (int)(bp % HW), b = (int)(bp / HW);
        float v;
        if (n < A * 4) v = dloc[((size_t)b * P + prior_off) * 4 + (size_t)pix * A * 4 + n];
        else v = dconf[((size_t)b * P + prior_off) * nc + (size_t)pix * A * nc + (n - A * 4)];
        out[i] = v;
    }
}

// stride-s zero insertion: U[b, s*i, s*j, c] = dy[b, i, j, c], zero elsewhere (dgrad of a strided conv)
__global__ void upsample_insert_kernel(const float* __restrict__ dy, float* __restrict__ u, int B, int Ho, int Wo, int H,
                                       int W, int C, int s) {
    const int C4 = C >> 2;
    const long long total = (long long)B * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y % s == 0 && x % s == 0 && y / s < Ho && x / s < Wo)
            v = *reinterpret_cast<const f32x4*>(dy + (((size_t)b * Ho + y / s) * Wo + x / s) * C + 4 * c4);
        *reinterpret_cast<f32x4*>(u + i * 4) = v;
    }
}

inline int nblocks(long long items, int cap = 2048) {
    long long b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
// Grid-stride passes that end in per-workgroup fp64 atomics on a [C] / [2C] array (BatchNorm backward sums, bias column sums): every
// workgroup finishes at about the same time and device-scope atomics on one cache line are served one after the other (~8 ns each,
// round 4: 2048 workgroups x 128 sums of a 64-channel layer = 32 k atomics per line = a 260 us tail).  1024-thread workgroups give the
// same 32 waves per CU with a quarter of the workgroups -- a quarter of the atomics.
constexpr int WIDE = 1024;
inline int nblocks_wide(long long items, int cap = 2048) {
    long long b = (items + WIDE - 1) / WIDE;
    if (b < 1) b = 1;
    if (b > cap / 4) b = cap / 4;
    return (int)b;
}

}  // namespace

extern "C" int gssd_bn_bwd_reduce_f32(const float* dout, const float* raw, const float* scale, const float* shift, float* dz,
                                      double* sums, int B, int H, int W, int C, int Ho, int Wo, int pool_k, int pool_s,
                                      int pool_p, int relu, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dout && raw && (dz || (pool_k == 0 && sums)) && B > 0 && C > 0 && C % 4 == 0 && C <= 4096);
    GSSD_CHECK_ARG((scale == nullptr) == (shift == nullptr));
    if (pool_k == 0) GSSD_CHECK_ARG(Ho == H && Wo == W);
    const long long total = (long long)B * Ho * Wo * (C / 4);
    GSSD_CHECK_ARG((long long)B * Ho * Wo < (1ll << 32));
    if (pool_k)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, float, true>), dim3(nblocks_wide(total)), dim3(WIDE), 2 * C * sizeof(float),
                           as_stream(stream), dout, raw, scale, shift, dz, sums, B, H, W, C, Ho, Wo, pool_k, pool_s, pool_p, relu);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, float, false>), dim3(nblocks_wide(total)), dim3(WIDE), 2 * C * sizeof(float),
                           as_stream(stream), dout, raw, scale, shift, dz, sums, B, H, W, C, Ho, Wo, pool_k, pool_s, pool_p, relu);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_bwd_reduce_mixed(const void* dout, int dout_bf16, const void* raw_bf16, const float* scale, const float* shift,
                                        float* dz, void* dz_bf16, double* sums, int B, int H, int W, int C, int Ho, int Wo, int pool_k,
                                        int pool_s, int pool_p, int relu, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dout && raw_bf16 && (dz || dz_bf16 || (pool_k == 0 && sums)) && B > 0 && C > 0 && C % 4 == 0 && C <= 4096);
    GSSD_CHECK_ARG(!dz_bf16 || (pool_k > 0 && pool_s >= pool_k));          // the bf16 hand-over: non-overlapping pools only (plain stores)
    GSSD_CHECK_ARG((scale == nullptr) == (shift == nullptr));
    if (pool_k == 0) GSSD_CHECK_ARG(Ho == H && Wo == W);
    const long long total = (long long)B * Ho * Wo * (C / 4);
    GSSD_CHECK_ARG((long long)B * Ho * Wo < (1ll << 32));
#define GSSD_RED(DT_, POOL_)                                                                                                            \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<u16, DT_, POOL_>), dim3(nblocks_wide(total)), dim3(WIDE), 2 * C * sizeof(float),             \
                       as_stream(stream), reinterpret_cast<const DT_*>(dout), reinterpret_cast<const u16*>(raw_bf16), scale, shift, dz, sums, B, \
                       H, W, C, Ho, Wo, pool_k, pool_s, pool_p, relu, reinterpret_cast<u16*>(dz_bf16))
    if (dout_bf16) {
        if (pool_k) GSSD_RED(u16, true); else GSSD_RED(u16, false);
    } else {
        if (pool_k) GSSD_RED(float, true); else GSSD_RED(float, false);
    }
#undef GSSD_RED
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_bwd_apply_mixed(const void* dout, int dout_bf16, float* dz, void* dz_bf16, const void* raw_bf16, const float* scale,
                                       const float* shift, int relu, const float* coef_a, const float* coef_b, const float* coef_c,
                                       int64_t pixels, int C, double* colsum, int store_f32, gssd_stream_t stream) {
    GSSD_CHECK_ARG(raw_bf16 && coef_a && coef_b && coef_c && pixels > 0 && C > 0 && C % 4 == 0 && C <= 4096);
    GSSD_CHECK_ARG((dout || dz) && (dz_bf16 || store_f32) && (!store_f32 || dz) && (scale == nullptr) == (shift == nullptr));
    if (dout_bf16)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<u16, u16>), dim3(nblocks_wide(pixels * (C / 4))), dim3(WIDE), C * sizeof(float),
                           as_stream(stream), dz, reinterpret_cast<const u16*>(raw_bf16), coef_a, coef_b, coef_c, (long long)pixels, C, colsum,
                           reinterpret_cast<const u16*>(dout), scale, shift, relu, reinterpret_cast<u16*>(dz_bf16), store_f32);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<u16, float>), dim3(nblocks_wide(pixels * (C / 4))), dim3(WIDE), C * sizeof(float),
                           as_stream(stream), dz, reinterpret_cast<const u16*>(raw_bf16), coef_a, coef_b, coef_c, (long long)pixels, C, colsum,
                           reinterpret_cast<const float*>(dout), scale, shift, relu, reinterpret_cast<u16*>(dz_bf16), store_f32);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_bwd_finalize_f32(const double* fwd_stats, double count, const double* bwd_sums, const float* gamma,
                                        float eps, int C, float* coef_a, float* coef_b, float* coef_c, float* dgamma,
                                        float* dbeta, int stats_rep, gssd_stream_t stream) {
    GSSD_CHECK_ARG(fwd_stats && bwd_sums && gamma && coef_a && coef_b && coef_c && dgamma && dbeta && C > 0 && count > 0);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), fwd_stats, count,
                       bwd_sums, gamma, eps, C, coef_a, coef_b, coef_c, dgamma, dbeta, stats_rep);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_bwd_apply_f32(float* dz, const float* raw, const float* coef_a, const float* coef_b,
                                     const float* coef_c, int64_t pixels, int C, double* colsum, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dz && raw && coef_a && coef_b && coef_c && pixels > 0 && C > 0 && C % 4 == 0 && C <= 4096);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(nblocks_wide(pixels * (C / 4))), dim3(WIDE), C * sizeof(float), as_stream(stream),
                       dz, raw, coef_a, coef_b, coef_c, (long long)pixels, C, colsum, (const float*)nullptr, nullptr, nullptr, 0, nullptr, 1);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_bwd_apply_masked_f32(const float* dout, const float* raw, const float* scale, const float* shift, int relu,
                                            const float* coef_a, const float* coef_b, const float* coef_c, float* draw,
                                            int64_t pixels, int C, double* colsum, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dout && draw && raw && coef_a && coef_b && coef_c && pixels > 0 && C > 0 && C % 4 == 0 && C <= 4096);
    GSSD_CHECK_ARG((scale == nullptr) == (shift == nullptr));
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(nblocks_wide(pixels * (C / 4))), dim3(WIDE), C * sizeof(float), as_stream(stream),
                       draw, raw, coef_a, coef_b, coef_c, (long long)pixels, C, colsum, dout, scale, shift, relu, nullptr, 1);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_colsum_f32(const float* x, int64_t rows, int C, int row_stride, double* out, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && out && rows > 0 && C > 0 && C <= 4096 && row_stride >= C);
    hipLaunchKernelGGL(colsum_kernel, dim3(nblocks_wide(rows * C, 512)), dim3(WIDE), C * sizeof(float), as_stream(stream), x,
                       (long long)rows, C, row_stride, out);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_cast_f64_f32(const double* x, float* y, int n, int accumulate, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(cast_f64_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), x, y, n, accumulate);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_l2norm_bwd_f32(const float* x, const float* weight, const float* dy, float* dx, const float* dx_add,
                                   double* dweight, int64_t pixels, int C, float eps, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && weight && dy && dx && dweight && pixels > 0 && C > 0 && C % 4 == 0 && C <= 4096);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(nblocks(pixels * 64, 1024)), dim3(256), C * sizeof(float), as_stream(stream), x,
                       weight, dy, dx, dx_add, dweight, (long long)pixels, C, eps);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_heads_gather_f32(const float* dloc, const float* dconf, float* out, int B, int HW, int A, int nc, int P,
                                     int prior_off, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dloc && dconf && out && B > 0 && HW > 0 && A > 0 && nc > 0 && prior_off >= 0 && prior_off + HW * A <= P);
    hipLaunchKernelGGL(heads_gather_kernel, dim3(nblocks((long long)B * HW * A * (4 + nc))), dim3(256), 0, as_stream(stream),
                       dloc, dconf, out, B, HW, A, nc, P, prior_off);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_upsample_insert_f32(const float* dy, float* u, int B, int Ho, int Wo, int H, int W, int C, int s,
                                        gssd_stream_t stream) {
    GSSD_CHECK_ARG(dy && u && B > 0 && Ho > 0 && Wo > 0 && H >= (Ho - 1) * s + 1 && W >= (Wo - 1) * s + 1 && C % 4 == 0 && s > 0);
    hipLaunchKernelGGL(upsample_insert_kernel, dim3(nblocks((long long)B * H * W * (C / 4))), dim3(256), 0, as_stream(stream),
                       dy, u, B, Ho, Wo, H, W, C, s);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
