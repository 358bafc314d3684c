// HBM-bound elementwise / reduction kernels of the GSSD path (gfx950): layout packing,
// BatchNorm(train|eval)+ReLU+max-pool, L2Norm, row softmax, slice_and_cat, spectral norm.
// All are written for 16-byte coalesced NHWC access and 64-lane wave reductions.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int EW_THREADS = 256;
inline int ew_blocks(long long work_items, int per_block = EW_THREADS, int cap = 8192) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ---------------------------------------------------------------------------------------------
// NCHW -> NHWC with per-group channel padding.  Reads are coalesced along x for each channel plane;
// one thread assembles one output pixel-group (cpg_out floats) -> 16-byte stores when cpg_out == 4.
// ---------------------------------------------------------------------------------------------
__global__ void pack_input_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW,
                                  int groups, int cpg_in, int cpg_out) {
    const long long total = (long long)B * HW * groups;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const long long bp = i / groups;
        const int pix = (int)(bp % HW);
        const int b = (int)(bp / HW);
        float* o = y + (bp * groups + g) * cpg_out;
        for (int c = 0; c < cpg_out; ++c)
            o[c] = (c < cpg_in) ? x[((long long)b * C + g * cpg_in + c) * HW + pix] : 0.f;
    }
}

__global__ void unpack_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW,
                                   int x_stride) {
    const long long total = (long long)B * C * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long long bc = i / HW;
        const int c = (int)(bc % C);
        const int b = (int)(bc / C);
        y[i] = x[((long long)b * HW + pix) * x_stride + c];
    }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int cin_g, int taps,
                                   int cin_g_pad, int Kpad) {
    const long long total = (long long)Cout * Kpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const int o = (int)(i / Kpad);
        const int tap = k / cin_g_pad, c = k - tap * cin_g_pad;
        float v = 0.f;
        if (tap < taps && c < cin_g) v = w[((long long)o * cin_g + c) * taps + tap];
        wp[i] = v;
    }
}

// The same packing for a whole table of weights in ONE launch (blockIdx.y = item): after an optimizer step every packed weight of the
// model is stale, and ~130 pack / copy launches of 5-8 us each sat in front of the next forward.  A plain copy of n floats is the item
// (Cout 1, cin_g n, taps 1, cin_g_pad n, Kpad n).
__global__ void pack_weight_batched_kernel(const gssd_pack_item* __restrict__ items) {
    const gssd_pack_item it = items[blockIdx.y];
    const long long total = (long long)it.Cout * it.Kpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % it.Kpad);
        const int o = (int)(i / it.Kpad);
        const int tap = k / it.cin_g_pad, c = k - tap * it.cin_g_pad;
        float v = 0.f;
        if (tap < it.taps && c < it.cin_g) v = it.w[((long long)o * it.cin_g + c) * it.taps + tap];
        it.wp[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm + ReLU + max-pool.  Each block derives (scale, shift) for all C channels into LDS from
// the fp64 batch sums (train) or the running statistics (eval), then grid-strides over output float4s.
// Block 0 also performs the running-statistics update of nn.BatchNorm2d (momentum, unbiased var).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(
    const float* __restrict__ raw, float* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo, int pk, int ps,
    int pp, const double* __restrict__ stats, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps, int training,
    int relu, int srep) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_scale = sm;
    float* s_shift = sm + C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (gamma == nullptr) {   // pool / ReLU only
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
    const int C4 = C >> 2;
    const long long total = (long long)B * Ho * Wo * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(s_scale + 4 * c4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(s_shift + 4 * c4);
        f32x4 r;
        if (pk == 0) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(raw + (((long long)b * H + yo) * W + xo) * C + 4 * c4);
            r = v * sc + sh;
        } else {
            r = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            const int y0 = yo * ps - pp, x0 = xo * ps - pp;
            for (int dy = 0; dy < pk; ++dy) {
                const int yy = y0 + dy;
                if ((unsigned)yy >= (unsigned)H) continue;
                for (int dx = 0; dx < pk; ++dx) {
                    const int xx = x0 + dx;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    const f32x4 v =
                        *reinterpret_cast<const f32x4*>(raw + (((long long)b * H + yy) * W + xx) * C + 4 * c4);
                    const f32x4 a = v * sc + sh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], a[e]);
                }
            }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = r;
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   float momentum, float eps, int training, int C, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ pad, int srep) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) {
        mean = gssd_stats_sum(stats, c, 2 * C, srep) / count;
        var = gssd_stats_sum(stats, C + c, 2 * C, srep) / count - mean * mean;
        if (var < 0.0) var = 0.0;
        if (training == 1) {      // training == 2: batch statistics again (backward) without a second running update
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
    if (scf == 0.f) scf = 1e-30f;                 // keeps the pad value mapped below zero
    scale[c] = scf;
    shift[c] = sh;
    pad[c] = scf > 0.f ? -3.0e38f : 3.0e38f;      // max(pad*scale + shift, 0) == 0: zero padding after BN+ReLU
}

// ---------------------------------------------------------------------------------------------
// L2Norm: one wave per pixel, channels strided over lanes as float4.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     float* __restrict__ out, long long pixels, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int C4 = C >> 2;
    for (long long p = wave0; p < pixels; p += nwaves) {
        const f32x4* xp = reinterpret_cast<const f32x4*>(x + p * C);
        float ss = 0.f;
        for (int c = lane; c < C4; c += 64) {
            const f32x4 v = xp[c];
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        ss = wave_sum(ss);
        const float inv = 1.f / (sqrtf(ss) + eps);
        f32x4* op = reinterpret_cast<f32x4*>(out + p * C);
        for (int c = lane; c < C4; c += 64) {
            const f32x4 v = xp[c];
            const f32x4 ww = reinterpret_cast<const f32x4*>(w)[c];
            op[c] = ww * (v * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row softmax (attention logits), one wave per row, three passes over an L2-resident row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long long rows, int n, int stride) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long rI = wave0; rI < rows; rI += nwaves) {
        float* row = x + rI * stride;
        float m = -INFINITY;
        for (int c = lane; c < n; c += 64) m = fmaxf(m, row[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < n; c += 64) {
            const float e = expf(row[c] - m);
            row[c] = e;
            s += e;
        }
        s = wave_sum(s);
        const float inv = 1.f / s;
        for (int c = lane; c < stride; c += 64) row[c] = (c < n) ? row[c] * inv : 0.f;
    }
}

__global__ void slice_and_cat_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                     long long pixels, int Ca, int Cb, int groups) {
    const int Co = Ca + Cb, Co4 = Co >> 2;
    const int ga = Ca / groups, gb = Cb / groups;
    const long long total = pixels * Co4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Co4) * 4;
        const long long p = i / Co4;
        const int g = c / (ga + gb), cc = c - g * (ga + gb);
        f32x4 v;
        if (cc < ga) v = *reinterpret_cast<const f32x4*>(a + p * Ca + g * ga + cc);
        else v = *reinterpret_cast<const f32x4*>(b + p * Cb + g * gb + (cc - ga));
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Spectral norm: one workgroup per matrix.  v <- normalize(W^T u); u <- normalize(W v); 1/sigma.
// ---------------------------------------------------------------------------------------------
__device__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(1024) void spectral_norm_kernel(const gssd_sn_item* __restrict__ items, int do_iter,
                                                            float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const gssd_sn_item it = items[blockIdx.x];
    const int R = it.rows, Cc = it.cols;
    float* su = sm;            // [R]
    float* sv = sm + R;        // [Cc]
    float* red = sv + Cc;      // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int i = tid; i < R; i += blockDim.x) su[i] = it.u[i];
    for (int i = tid; i < Cc; i += blockDim.x) sv[i] = it.v[i];
    __syncthreads();
    const bool vec = ((Cc | R) & 3) == 0 && (Cc >> 2) <= (int)blockDim.x && (((uintptr_t)it.w) & 15) == 0;
    if (do_iter) {
        float nrm = 0.f;
        if (vec) {
            // v = W^T u (round 6): thread -> (four columns, one of S row slices), eight 16-byte loads in flight per thread, the slices' partial sums
            // meet in LDS in slice order.  One workgroup owns a matrix (three dependent passes), so its time is the latency of its load chains: the
            // thread-per-column form walked all R rows per thread (512 x 1024: 64 dependent rounds of eight 4-byte loads, 414 us for the 48 matrices
            // of GSSD++ -- long enough beside the trunk's first layers to slow their persistent workgroups, profiles/r06_thin_x6_notes.txt).
            float* part = red + 16;                    // [S][Cc], S * Cc <= 4 * blockDim.x
            const int nq = Cc >> 2, S = (int)blockDim.x / nq;
            const int cq = tid % nq, rs = tid / nq;
            if (rs < S) {
                f32x4 a8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) a8[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* wc = it.w + 4 * cq;
                int rr = rs;
                for (; rr + 7 * S < R; rr += 8 * S) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) a8[q] += *reinterpret_cast<const f32x4*>(wc + (size_t)(rr + q * S) * Cc) * su[rr + q * S];
                }
                for (; rr < R; rr += S) a8[0] += *reinterpret_cast<const f32x4*>(wc + (size_t)rr * Cc) * su[rr];
                *reinterpret_cast<f32x4*>(part + rs * Cc + 4 * cq) = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
            }
            __syncthreads();
            for (int c = tid; c < Cc; c += blockDim.x) {
                float acc = part[c];
                for (int sl = 1; sl < S; ++sl) acc += part[sl * Cc + c];
                sv[c] = acc;
                nrm += acc * acc;
            }
        } else {
            // thread per column, rows serial (coalesced across threads)
            for (int c = tid; c < Cc; c += blockDim.x) {
                // eight independent partial sums keep eight loads in flight per thread (one workgroup owns the whole matrix)
                float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                int rr = 0;
                for (; rr + 8 <= R; rr += 8) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) a8[q] += it.w[(size_t)(rr + q) * Cc + c] * su[rr + q];
                }
                for (; rr < R; ++rr) a8[0] += it.w[(size_t)rr * Cc + c] * su[rr];
                const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
                sv[c] = acc;
                nrm += acc * acc;
            }
        }
        nrm = block_sum(nrm, red);
        const float inv = 1.f / fmaxf(sqrtf(nrm), eps);
        for (int c = tid; c < Cc; c += blockDim.x) sv[c] *= inv;
        __syncthreads();
    }
    // t = W v : one wave per row (vec: four rows at a time, 16-byte loads -- four independent chains per lane)
    float sig = 0.f, nrm2 = 0.f;
    auto row_done = [&](int rr, float acc) {
        if (do_iter) {
            su[rr] = acc;     // un-normalised W v
            nrm2 += acc * acc;
        } else {
            sig += su[rr] * acc;
        }
    };
    if (vec) {
        for (int r0 = wave * 4; r0 < R; r0 += nw * 4) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = lane * 4; c < Cc; c += 256) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(sv + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (r0 + q < R) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(it.w + (size_t)(r0 + q) * Cc + c);
                        acc[q] += (w4[0] * v4[0] + w4[1] * v4[1]) + (w4[2] * v4[2] + w4[3] * v4[3]);
                    }
                }
            }
            // (su[r0 + q] is read by row_done before it is overwritten: the same lane, program order)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = wave_sum(acc[q]);
                if (lane == 0 && r0 + q < R) row_done(r0 + q, t);
            }
        }
    } else {
        for (int rr = wave; rr < R; rr += nw) {
            float acc = 0.f;
            for (int c = lane; c < Cc; c += 64) acc += it.w[(size_t)rr * Cc + c] * sv[c];
            acc = wave_sum(acc);
            if (lane == 0) row_done(rr, acc);
        }
    }
    if (do_iter) {
        nrm2 = block_sum(nrm2, red);
        const float nn = fmaxf(sqrtf(nrm2), eps);
        // u = Wv / ||Wv||;  sigma = u . (W v) = ||Wv||^2 / nn
        __syncthreads();
        for (int i = tid; i < R; i += blockDim.x) it.u[i] = su[i] / nn;
        for (int i = tid; i < Cc; i += blockDim.x) it.v[i] = sv[i];
        const float is = nn / nrm2;
        for (int i = tid; i < R; i += blockDim.x) it.inv_sigma[i] = is;
    } else {
        sig = block_sum(sig, red);
        const float is = 1.f / sig;
        for (int i = tid; i < R; i += blockDim.x) it.inv_sigma[i] = is;
    }
}

__global__ void softmax_lastdim_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int C) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < rows;
         i += (long long)gridDim.x * blockDim.x) {
        const float* xr = x + i * C;
        float m = xr[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, xr[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += (float)exp((double)(xr[c] - m));
        for (int c = 0; c < C; ++c) y[i * C + c] = (float)exp((double)(xr[c] - m)) / s;
    }
}

__global__ __launch_bounds__(256) void reduce_max_kernel(const float* __restrict__ x, long long n, float* out) {
    // one partial maximum per block; consumers take the max over gridDim.x partials
    __shared__ float red[4];
    float m = -INFINITY;
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}


// Deterministic split-K for the multibox heads: slice k of a head wrote its partial sums to ws[k] (same layout as the output);
// out[b][p][c] = sum over k < splits[p] of ws[k][b][p][c], in slice order -- no atomics, run-to-run identical bits.
__global__ __launch_bounds__(256) void heads_reduce_kernel(const float* __restrict__ ws, const signed char* __restrict__ splits,
                                                          float* __restrict__ out, long long total, int P, int C) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)((i / C) % P);
        const int n = splits[p];
        float a = ws[i];
        for (int k = 1; k < n; ++k) a += ws[(size_t)k * total + i];
        out[i] = a;
    }
}

}  // namespace

extern "C" int gssd_pack_input_nhwc(const float* x, float* y, int B, int C, int H, int W, int groups, int cpg_out,
                                    gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0);
    GSSD_CHECK_ARG(cpg_out >= C / groups);
    const long long total = (long long)B * H * W * groups;
    hipLaunchKernelGGL(pack_input_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), x, y, B, C,
                       H * W, groups, C / groups, cpg_out);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_unpack_nhwc_to_nchw(const float* x, float* y, int B, int C, int H, int W, int x_stride,
                                        gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && x_stride >= C);
    hipLaunchKernelGGL(unpack_nhwc_kernel, dim3(ew_blocks((long long)B * C * H * W)), dim3(EW_THREADS), 0,
                       as_stream(stream), x, y, B, C, H * W, x_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pack_conv_weight(const float* w, float* wp, int Cout, int cin_g, int KH, int KW, int cin_g_pad,
                                     int Kpad, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w && wp && Cout > 0 && cin_g > 0 && KH > 0 && KW > 0);
    GSSD_CHECK_ARG(cin_g_pad >= cin_g && cin_g_pad % 4 == 0 && Kpad >= KH * KW * cin_g_pad && Kpad % 4 == 0);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(ew_blocks((long long)Cout * Kpad)), dim3(EW_THREADS), 0,
                       as_stream(stream), w, wp, Cout, cin_g, KH * KW, cin_g_pad, Kpad);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pack_conv_weights_batched(const gssd_pack_item* items_dev, int n_items, gssd_stream_t stream) {
    GSSD_CHECK_ARG(items_dev && n_items > 0 && n_items < 65536);
    hipLaunchKernelGGL(pack_weight_batched_kernel, dim3(48, n_items), dim3(EW_THREADS), 0, as_stream(stream), items_dev);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_relu_pool_f32(const float* raw, float* out, int B, int H, int W, int C, int Ho, int Wo,
                                     int pool_k, int pool_s, int pool_p, const double* stats, double count,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     float momentum, float eps, int training, int relu, int stats_rep, gssd_stream_t stream) {
    GSSD_CHECK_ARG(raw && out && stats_rep >= 0);
    GSSD_CHECK_ARG(gamma == nullptr || (beta && running_mean && running_var));
    GSSD_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && Ho > 0 && Wo > 0);
    GSSD_CHECK_ARG(!training || gamma == nullptr || (stats != nullptr && count > 0));
    if (pool_k == 0) GSSD_CHECK_ARG(Ho == H && Wo == W);
    else GSSD_CHECK_ARG(pool_s > 0 && pool_p >= 0 && (Ho - 1) * pool_s - pool_p < H && (Wo - 1) * pool_s - pool_p < W);
    const long long total = (long long)B * Ho * Wo * (C / 4);
    const int blocks = ew_blocks(total, EW_THREADS * 4, 2048);
    hipLaunchKernelGGL(bn_relu_pool_kernel, dim3(blocks), dim3(EW_THREADS), 2 * C * sizeof(float), as_stream(stream),
                       raw, out, B, H, W, C, Ho, Wo, pool_k, pool_s, pool_p, stats, count, gamma, beta, running_mean,
                       running_var, momentum, eps, training, relu, stats_rep);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bn_finalize_f32(const double* stats, double count, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, int training,
                                    int C, float* scale, float* shift, float* pad, int stats_rep, gssd_stream_t stream) {
    GSSD_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && pad && C > 0 && stats_rep >= 0);
    GSSD_CHECK_ARG(!training || (stats != nullptr && count > 0));
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), stats, count, gamma,
                       beta, running_mean, running_var, momentum, eps, training, C, scale, shift, pad, stats_rep);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_l2norm_f32(const float* x, const float* weight, float* out, int64_t pixels, int C, float eps,
                               gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && weight && out && pixels > 0 && C > 0 && C % 4 == 0);
    hipLaunchKernelGGL(l2norm_kernel, dim3(ew_blocks(pixels, 4, 4096)), dim3(256), 0, as_stream(stream), x, weight, out,
                       (long long)pixels, C, eps);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_softmax_rows_f32(float* x, int64_t rows, int n, int row_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && rows > 0 && n > 0 && row_stride >= n);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(ew_blocks(rows, 4, 16384)), dim3(256), 0, as_stream(stream), x,
                       (long long)rows, n, row_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_slice_and_cat_f32(const float* a, const float* b, float* out, int64_t pixels, int Ca, int Cb,
                                      int groups, gssd_stream_t stream) {
    GSSD_CHECK_ARG(a && b && out && pixels > 0 && groups > 0 && Ca % (4 * groups) == 0 && Cb % (4 * groups) == 0);
    hipLaunchKernelGGL(slice_and_cat_kernel, dim3(ew_blocks(pixels * ((Ca + Cb) / 4))), dim3(EW_THREADS), 0,
                       as_stream(stream), a, b, out, (long long)pixels, Ca, Cb, groups);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_spectral_norm_f32(const gssd_sn_item* items_dev, int n, int do_power_iteration, float eps,
                                      gssd_stream_t stream) {
    GSSD_CHECK_ARG(items_dev && n > 0);
    // rows + cols <= 1536 for every Self_Attn conv of the path (512 x 1024 is the largest); --feature_scale 2 doubles both (3072)
    const size_t smem = (4096 + 16 + 4096) * sizeof(float);          // u | v | reduction slots | the row slices' partial sums
    hipLaunchKernelGGL(spectral_norm_kernel, dim3(n), dim3(1024), smem, as_stream(stream), items_dev,
                       do_power_iteration, eps);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_softmax_lastdim_f32(const float* x, float* y, int64_t rows, int C, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && rows > 0 && C > 0);
    hipLaunchKernelGGL(softmax_lastdim_kernel, dim3(ew_blocks(rows)), dim3(EW_THREADS), 0, as_stream(stream), x, y,
                       (long long)rows, C);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_reduce_max_f32(const float* x, int64_t n, float* out, int out_n, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && out && n > 0 && out_n > 0 && out_n <= 1024 && ((uintptr_t)x % 16) == 0);
    hipLaunchKernelGGL(reduce_max_kernel, dim3(out_n), dim3(256), 0, as_stream(stream), x, (long long)n, out);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_heads_reduce_f32(const float* ws, const signed char* splits, float* out, int B, int P, int C, gssd_stream_t stream) {
    GSSD_CHECK_ARG(ws && splits && out && B > 0 && P > 0 && C > 0);
    const long long total = (long long)B * P * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(heads_reduce_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), ws, splits, out, total, P, C);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
