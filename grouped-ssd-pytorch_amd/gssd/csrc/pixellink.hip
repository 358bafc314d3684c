// PixelLink++ tail for gfx950: everything of ssd_liverdet/pixel_link that the GSSD++ kernels do not already
// cover.  The trunk, Self_Attn, the deformable conv, fuse conv + BatchNorm and the 1x1 score heads are launches of the existing
// kernels (gssd/pixellink.py); this file holds
//   - the upsample-add cascade (pixel_link/model.py:341-400): bilinear, align_corners=True, NHWC, the 2 pixel + 16 link channels of
//     a stage travel as ONE 18-channel map (every cascade op is channel-wise), fused with the lateral add;
//   - the final 1x1 convs over the concatenated cascade features, written NCHW like the reference returns them;
//   - PixelLinkLoss (pixel_link/criterion.py:24-104): OHEM pixel loss (k-th smallest background probability by a bitonic sort in LDS)
//     and the link loss, one workgroup per image, fp64 sums;
//   - the link decoding of pixel_link/postprocess.py:178-234 (`func`): connected components under directed 8-neighbour links by
//     min-label propagation in LDS, numbered in raster order of their first pixel exactly like the union-find + root_map of the
//     reference, plus per-component pixel count, bounding box and score sum.
// HBM-bound byte work on <= 75 x 75 maps: one workgroup per image (or per row block), no MFMA.
#include <math.h>
#include "common.h"

namespace {

// out = interp(src) [+ addend -> out2]; src [B][Hs][Ws][C], out [B][Hd][Wd][C]
// at::native upsample_bilinear2d, align_corners=True: scale = (in - 1) / (out - 1) (0 when out == 1), src = scale * dst,
// i0 = (int)src, i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1; value = h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11)
__global__ __launch_bounds__(256) void interp_add_kernel(const float* __restrict__ src, const float* __restrict__ addend,
                                                         float* __restrict__ out, float* __restrict__ out2, int B, int Hs, int Ws,
                                                         int Hd, int Wd, int C) {
    const long long total = (long long)B * Hd * Wd * C;
    const float sh = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
    const float sw = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int x = (int)(t % Wd);
        t /= Wd;
        const int y = (int)(t % Hd);
        const int b = (int)(t / Hd);
        const float fy = sh * (float)y, fx = sw * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float h1 = fy - (float)y0, h0 = 1.f - h1, w1 = fx - (float)x0, w0 = 1.f - w1;
        const float* sb = src + (size_t)b * Hs * Ws * C + c;
        const float p00 = sb[(size_t)(y0 * Ws + x0) * C], p01 = sb[(size_t)(y0 * Ws + x1) * C];
        const float p10 = sb[(size_t)(y1 * Ws + x0) * C], p11 = sb[(size_t)(y1 * Ws + x1) * C];
        const float v = h0 * (w0 * p00 + w1 * p01) + h1 * (w0 * p10 + w1 * p11);
        out[i] = v;
        if (addend) out2[i] = v + addend[i];
    }
}

// final_1 / final_2 (model.py:360,386 / 396,411): features f[k] [B][H][W][18] (channels 0-1 pixel, 2-17 link), k < nf;
// out1[b][o][y][x] = b1[o] + sum_{k,c<2} w1[o][k*2 + c] * f[k][..][c];  out2[b][o][y][x] = b2[o] + sum_{k,c<16} w2[o][k*16 + c] * f[k][..][2 + c]
__global__ __launch_bounds__(256) void pl_final_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                       const float* __restrict__ f2, const float* __restrict__ f3, int nf,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ out1, float* __restrict__ out2, int B, int HW) {
    __shared__ float sw1[2 * 8], sw2[16 * 64], sb[18];
    for (int i = threadIdx.x; i < 2 * 2 * nf; i += 256) sw1[i] = w1[i];
    for (int i = threadIdx.x; i < 16 * 16 * nf; i += 256) sw2[i] = w2[i];
    if (threadIdx.x < 2) sb[threadIdx.x] = b1[threadIdx.x];
    if (threadIdx.x >= 2 && threadIdx.x < 18) sb[threadIdx.x] = b2[threadIdx.x - 2];
    __syncthreads();
    const float* fs[4] = {f0, f1, f2, f3};
    const long long total = (long long)B * HW;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        float v[4][18];
        for (int k = 0; k < nf; ++k)
#pragma unroll
            for (int c = 0; c < 18; ++c) v[k][c] = fs[k][p * 18 + c];
        const int b = (int)(p / HW), pix = (int)(p - (long long)b * HW);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float a = sb[o];
            for (int k = 0; k < nf; ++k)
#pragma unroll
                for (int c = 0; c < 2; ++c) a = __builtin_fmaf(sw1[o * 2 * nf + k * 2 + c], v[k][c], a);
            out1[((size_t)b * 2 + o) * HW + pix] = a;
        }
        for (int o = 0; o < 16; ++o) {
            float a = sb[2 + o];
            for (int k = 0; k < nf; ++k)
#pragma unroll
                for (int c = 0; c < 16; ++c) a = __builtin_fmaf(sw2[o * 16 * nf + k * 16 + c], v[k][2 + c], a);
            out2[((size_t)b * 16 + o) * HW + pix] = a;
        }
    }
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// ---- PixelLinkLoss ------------------------------------------------------------------------------------------------------------------
// One workgroup (1024 threads) per image.  out[b] = {pixel_pos, pixel_neg, link_pos, link_neg, area, neg_area} per image (the
// reference's four batch means are the means of the first four columns).
constexpr int PL_SORT = 8192;      // >= H*W of the score map (75 x 75 = 5625)

__global__ __launch_bounds__(1024) void pl_loss_kernel(const float* __restrict__ out1, const float* __restrict__ out2,
                                                       const long long* __restrict__ pixel_t, const unsigned char* __restrict__ neg_mask,
                                                       const float* __restrict__ pos_w, const long long* __restrict__ link_t,
                                                       double* __restrict__ res, float* __restrict__ neg_w_out, int HW, int ratio) {
    __shared__ float key[PL_SORT];
    __shared__ double red[16];
    __shared__ int s_cnt[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* l0 = out1 + (size_t)b * 2 * HW;
    const float* l1 = l0 + HW;
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    // softmax probability of class 0 (background) as nn.Softmax2d computes it: exp(x0 - m) / (exp(x0 - m) + exp(x1 - m))
    int area = 0, ncand = 0;
    for (int i = tid; i < PL_SORT; i += 1024) {
        float k = INFINITY;
        if (i < HW) {
            const float a = l0[i], c = l1[i];
            const float m = fmaxf(a, c);
            const float e0 = expf(a - m), e1 = expf(c - m);
            const float p0 = e0 / (e0 + e1);
            if (neg_mask[(size_t)b * HW + i] == 1) {
                k = p0;
                ++ncand;
            }
            area += (int)pixel_t[(size_t)b * HW + i];
        }
        key[i] = k;
    }
    area = wave_sum(area);
    ncand = wave_sum(ncand);
    if ((tid & 63) == 0) {
        atomicAdd(&s_cnt[0], area);
        atomicAdd(&s_cnt[1], ncand);
    }
    __syncthreads();
    area = s_cnt[0];
    ncand = s_cnt[1];
    int r_pos = area * ratio;
    if (r_pos == 0) r_pos = 10000;                                   // criterion.py:41-43
    const int neg_area = min(r_pos, ncand);
    // ascending bitonic sort of the candidate probabilities (non-candidates = +inf at the end)
    for (int k = 2; k <= PL_SORT; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = tid; i < PL_SORT; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float a = key[i], c = key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) {
                        key[i] = c;
                        key[ixj] = a;
                    }
                }
            }
        }
    __syncthreads();
    // criterion.py:46-48: topk(-p0, neg_area)[-1] = the neg_area-th smallest p0; every candidate with p0 <= it is selected (ties too)
    const float thr = neg_area > 0 ? key[neg_area - 1] : -INFINITY;
    __syncthreads();
    double s_pos = 0.0, s_neg = 0.0;
    for (int i = tid; i < HW; i += 1024) {
        const float a = l0[i], c = l1[i];
        const float m = fmaxf(a, c);
        const float e0 = expf(a - m), e1 = expf(c - m);
        const float p0 = e0 / (e0 + e1);
        const long long t = pixel_t[(size_t)b * HW + i];
        const float ce = (m + logf(e0 + e1)) - (t ? c : a);           // CrossEntropyLoss(reduce=False)
        const bool sel = neg_area > 0 && p0 <= thr && neg_mask[(size_t)b * HW + i] == 1;
        if (neg_w_out) neg_w_out[(size_t)b * HW + i] = sel ? 1.f : 0.f;
        s_pos += (double)(pos_w[(size_t)b * HW + i] * ce);
        if (sel) s_neg += (double)ce;
    }
    s_pos = block_sum(s_pos, red);
    s_neg = block_sum(s_neg, red);
    // link loss (criterion.py:66-104): weights = pos_pixel_weight where the link target is 1 / 0
    double wp = 0.0, wn = 0.0, lp = 0.0, ln = 0.0;
    for (int i = tid; i < 8 * HW; i += 1024) {
        const int n = i / HW, pix = i - n * HW;
        const float a = out2[((size_t)b * 16 + 2 * n) * HW + pix], c = out2[((size_t)b * 16 + 2 * n + 1) * HW + pix];
        const float m = fmaxf(a, c);
        const long long t = link_t[((size_t)b * 8 + n) * HW + pix];
        const float ce = (m + logf(expf(a - m) + expf(c - m))) - (t ? c : a);
        const float w = pos_w[(size_t)b * HW + pix];
        if (t == 1) {
            wp += (double)w;
            lp += (double)(w * ce);
        } else if (t == 0) {
            wn += (double)w;
            ln += (double)(w * ce);
        }
    }
    wp = block_sum(wp, red);
    wn = block_sum(wn, red);
    lp = block_sum(lp, red);
    ln = block_sum(ln, red);
    if (tid == 0) {
        const double den = (double)area + (double)neg_area;
        res[b * 6 + 0] = s_pos / den;
        res[b * 6 + 1] = s_neg / den;
        res[b * 6 + 2] = wp == 0.0 ? 0.0 : lp / wp;
        res[b * 6 + 3] = wn == 0.0 ? 0.0 : ln / wn;
        res[b * 6 + 4] = (double)area;
        res[b * 6 + 5] = (double)neg_area;
    }
}

// ---- backward of the tail (training: ssd_liverdet/pixel_link train loop = forward, PixelLinkLoss, loss.backward()) ------
// Gradient maps of the 18-channel score maps carry a channel stride `ld` >= 18 (20 in the plan: the conv kernels that continue the chain
// want rows of whole 16-byte quads; channels 18 .. ld-1 stay zero).

// d(interp_add): g = d(out) [+ d(out2)] scattered to the four source corners with the forward's weights (atomics: d(src) accumulates);
// d(addend) += d(out2).  dout / dout2 may be NULL (no gradient reached that output).
__global__ __launch_bounds__(256) void interp_add_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ dout2,
                                                             float* __restrict__ dsrc, float* __restrict__ daddend, int B, int Hs, int Ws,
                                                             int Hd, int Wd, int C, int ld) {
    const long long total = (long long)B * Hd * Wd * C;
    const float sh = Hd > 1 ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
    const float sw = Wd > 1 ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const size_t po = (size_t)t * ld + c;
        const int x = (int)(t % Wd);
        t /= Wd;
        const int y = (int)(t % Hd);
        const int b = (int)(t / Hd);
        const float g2 = dout2 ? dout2[po] : 0.f;
        const float g = (dout ? dout[po] : 0.f) + g2;
        if (daddend && dout2) daddend[po] += g2;
        if (g == 0.f) continue;
        const float fy = sh * (float)y, fx = sw * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float h1 = fy - (float)y0, h0 = 1.f - h1, w1 = fx - (float)x0, w0 = 1.f - w1;
        float* sb = dsrc + (size_t)b * Hs * Ws * ld + c;
        unsafeAtomicAdd(sb + (size_t)(y0 * Ws + x0) * ld, g * h0 * w0);
        unsafeAtomicAdd(sb + (size_t)(y0 * Ws + x1) * ld, g * h0 * w1);
        unsafeAtomicAdd(sb + (size_t)(y1 * Ws + x0) * ld, g * h1 * w0);
        unsafeAtomicAdd(sb + (size_t)(y1 * Ws + x1) * ld, g * h1 * w1);
    }
}

// d(final_1 / final_2) w.r.t. the cascade features: df[k][p][c] (+)= sum_o w1[o][2k + c] d1[b][o][pix] (c < 2), sum_o w2[o][16k + c-2] d2[b][o][pix]
// (c >= 2).  acc[k] != 0: the map already holds another contribution (a feature that also feeds the next cascade step).
__global__ __launch_bounds__(256) void pl_final_bwd_kernel(const float* __restrict__ d1, const float* __restrict__ d2, int nf,
                                                           const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ g0,
                                                           float* __restrict__ g1, float* __restrict__ g2, float* __restrict__ g3, int acc_mask,
                                                           int B, int HW, int ld) {
    __shared__ float sw1[2 * 8], sw2[16 * 64];
    for (int i = threadIdx.x; i < 2 * 2 * nf; i += 256) sw1[i] = w1[i];
    for (int i = threadIdx.x; i < 16 * 16 * nf; i += 256) sw2[i] = w2[i];
    __syncthreads();
    float* gs[4] = {g0, g1, g2, g3};
    const long long total = (long long)B * HW;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(p / HW), pix = (int)(p - (long long)b * HW);
        float a1[2], a2[16];
#pragma unroll
        for (int o = 0; o < 2; ++o) a1[o] = d1[((size_t)b * 2 + o) * HW + pix];
#pragma unroll
        for (int o = 0; o < 16; ++o) a2[o] = d2[((size_t)b * 16 + o) * HW + pix];
        for (int k = 0; k < nf; ++k) {
            float* gp = gs[k] + (size_t)p * ld;
            const bool acc = (acc_mask >> k) & 1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v = 0.f;
#pragma unroll
                for (int o = 0; o < 2; ++o) v = __builtin_fmaf(sw1[o * 2 * nf + k * 2 + c], a1[o], v);
                gp[c] = acc ? gp[c] + v : v;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float v = 0.f;
#pragma unroll
                for (int o = 0; o < 16; ++o) v = __builtin_fmaf(sw2[o * 16 * nf + k * 16 + c], a2[o], v);
                gp[2 + c] = acc ? gp[2 + c] + v : v;
            }
        }
    }
}

// weight / bias gradients of final_1 / final_2: dw1[o][2k + c] = sum_p d1[o][p] f[k][p][c], db1[o] = sum_p d1[o][p] (same for final_2);
// a workgroup reduces 256 pixels through LDS (thread -> one output row o of final_2 and a 4-wide column block, the 16 + 2 leftovers
// on the first threads), then fp64 atomics.  dw layout = the conv weights' [O][nf * c] rows; db follow at dw + O * nf * c.
__global__ __launch_bounds__(256) void pl_final_wgrad_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                             const float* __restrict__ f0, const float* __restrict__ f1,
                                                             const float* __restrict__ f2, const float* __restrict__ f3, int nf,
                                                             double* __restrict__ dw1, double* __restrict__ dw2, int B, int HW) {
    constexpr int TP = 128;                 // pixels per pass (47 KB of LDS)
    __shared__ float sd[TP][18 + 1];        // d1 (2) | d2 (16) of the pass's pixels
    __shared__ float sf[TP][72 + 1];        // features: k * 18 + c
    const float* fs[4] = {f0, f1, f2, f3};
    const long long total = (long long)B * HW;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int o2 = threadIdx.x >> 4, cb = (threadIdx.x & 15) * 4;        // final_2: row o2, columns cb .. cb + 3 of [16][16 nf]
    const int lp = threadIdx.x & (TP - 1), half = threadIdx.x >> 7;      // loader: pixel lp of the pass; half 0 = d + f0, f1, half 1 = f2, f3
    for (long long base = (long long)blockIdx.x * TP; base < total; base += (long long)gridDim.x * TP) {
        const long long p = base + lp;
        __syncthreads();
        const bool ok = p < total;
        const int b = ok ? (int)(p / HW) : 0, pix = ok ? (int)(p - (long long)b * HW) : 0;
        if (half == 0) {
#pragma unroll
            for (int o = 0; o < 2; ++o) sd[lp][o] = ok ? d1[((size_t)b * 2 + o) * HW + pix] : 0.f;
#pragma unroll
            for (int o = 0; o < 16; ++o) sd[lp][2 + o] = ok ? d2[((size_t)b * 16 + o) * HW + pix] : 0.f;
        }
        for (int k = 2 * half; k < 2 * half + 2 && k < nf; ++k)
#pragma unroll
            for (int c = 0; c < 18; ++c) sf[lp][k * 18 + c] = ok ? fs[k][p * 18 + c] : 0.f;
        __syncthreads();
        for (int q = 0; q < TP; ++q) {
            const float g = sd[q][2 + o2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cb + j;                        // = k * 16 + c  (c < 16) of final_2's input
                if (col < 16 * nf) acc[j] = __builtin_fmaf(g, sf[q][(col >> 4) * 18 + 2 + (col & 15)], acc[j]);
            }
            // leftovers on threads 0 .. 4 nf + 17: final_1's 2 x 2 nf weights, then the 2 + 16 biases
            const int t = threadIdx.x;
            if (t < 4 * nf) {
                const int o = t / (2 * nf), col = t - o * 2 * nf;
                acc[4] = __builtin_fmaf(sd[q][o], sf[q][(col >> 1) * 18 + (col & 1)], acc[4]);
            } else if (t < 4 * nf + 18) {
                acc[4] += sd[q][t - 4 * nf];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (cb + j < 16 * nf) unsafeAtomicAdd(dw2 + o2 * 16 * nf + cb + j, (double)acc[j]);
    const int t = threadIdx.x;
    if (t < 4 * nf) unsafeAtomicAdd(dw1 + t, (double)acc[4]);
    else if (t < 4 * nf + 2) unsafeAtomicAdd(dw1 + 4 * nf + (t - 4 * nf), (double)acc[4]);               // db1
    else if (t < 4 * nf + 18) unsafeAtomicAdd(dw2 + 16 * 16 * nf + (t - 4 * nf - 2), (double)acc[4]);     // db2
}

// d(PixelLinkLoss) w.r.t. the two score maps (criterion.py:24-104 under autograd: the mined-negative mask, the areas and the link weight
// sums are constants of the step).  g[4] = upstream gradients of the four returned means (pixel pos, pixel neg, link pos, link neg);
// neg_w = the mined mask the forward launch wrote; per_image = its [B][6] result (area, neg_area in columns 4, 5).
__global__ __launch_bounds__(1024) void pl_loss_bwd_kernel(const float* __restrict__ out1, const float* __restrict__ out2,
                                                           const long long* __restrict__ pixel_t, const float* __restrict__ neg_w,
                                                           const float* __restrict__ pos_w, const long long* __restrict__ link_t,
                                                           const double* __restrict__ res, const float* __restrict__ g,
                                                           float* __restrict__ d1, float* __restrict__ d2, int B, int HW) {
    __shared__ double red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double den = res[b * 6 + 4] + res[b * 6 + 5];
    const float inv_den = den > 0.0 ? (float)(1.0 / (den * (double)B)) : 0.f;
    const float gp = g[0], gn = g[1], glp = g[2], gln = g[3];
    const float* l0 = out1 + (size_t)b * 2 * HW;
    const float* l1 = l0 + HW;
    for (int i = tid; i < HW; i += 1024) {
        const float a = l0[i], c = l1[i];
        const float m = fmaxf(a, c);
        const float e0 = expf(a - m), e1 = expf(c - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const long long t = pixel_t[(size_t)b * HW + i];
        const float wgt = (gp * pos_w[(size_t)b * HW + i] + gn * neg_w[(size_t)b * HW + i]) * inv_den;
        d1[((size_t)b * 2) * HW + i] = wgt * (p0 - (t ? 0.f : 1.f));
        d1[((size_t)b * 2 + 1) * HW + i] = wgt * (p1 - (t ? 1.f : 0.f));
    }
    double wp = 0.0, wn = 0.0;
    for (int i = tid; i < 8 * HW; i += 1024) {
        const int n = i / HW, pix = i - n * HW;
        const long long t = link_t[((size_t)b * 8 + n) * HW + pix];
        const float w = pos_w[(size_t)b * HW + pix];
        if (t == 1) wp += (double)w;
        else if (t == 0) wn += (double)w;
    }
    wp = block_sum(wp, red);
    wn = block_sum(wn, red);
    const float cp = wp == 0.0 ? 0.f : (float)((double)glp / (wp * (double)B));
    const float cn = wn == 0.0 ? 0.f : (float)((double)gln / (wn * (double)B));
    for (int i = tid; i < 8 * HW; i += 1024) {
        const int n = i / HW, pix = i - n * HW;
        const size_t oa = ((size_t)b * 16 + 2 * n) * HW + pix, oc = oa + HW;
        const float a = out2[oa], c = out2[oc];
        const float m = fmaxf(a, c);
        const float e0 = expf(a - m), e1 = expf(c - m);
        const float pa = e0 / (e0 + e1), pc = e1 / (e0 + e1);
        const long long t = link_t[((size_t)b * 8 + n) * HW + pix];
        const float coef = pos_w[(size_t)b * HW + pix] * (t == 1 ? cp : t == 0 ? cn : 0.f);
        d2[oa] = coef * (pa - (t == 0 ? 1.f : 0.f));
        d2[oc] = coef * (pc - (t == 1 ? 1.f : 0.f));
    }
}

// ---- link decoding ------------------------------------------------------------------------------------------------------------------
// postprocess.py:104-121 thresholds + `func` (:178-234).  neighbour order (get_neighbors :166-176): (-1,-1) (-1,0) (-1,+1) (0,+1)
// (+1,+1) (+1,0) (+1,-1) (0,-1).  Two positive pixels p, q are joined when link i of p towards q is on (either direction suffices:
// joint() is symmetric).  Label = 1 + rank of the component's first pixel in raster order.
constexpr int PL_MAXPIX = 8192;
constexpr int PL_STAT_COMPS = 1024;      // components with statistics (LDS table); the label map itself is unlimited

__global__ __launch_bounds__(1024) void pl_decode_kernel(const float* __restrict__ out1, const float* __restrict__ out2,
                                                         int* __restrict__ labels, float* __restrict__ comps, int* __restrict__ ncomp,
                                                         int H, int W, float pixel_thr, float link_thr, int max_comp) {
    __shared__ int lab[PL_MAXPIX];
    __shared__ unsigned char lk[PL_MAXPIX];
    __shared__ int nroot;
    const int b = blockIdx.x, tid = threadIdx.x, HW = H * W;
    constexpr int dy[8] = {-1, -1, -1, 0, 1, 1, 1, 0}, dx[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
    for (int i = tid; i < HW; i += 1024) {
        const float a = out1[((size_t)b * 2) * HW + i], c = out1[((size_t)b * 2 + 1) * HW + i];
        const float m = fmaxf(a, c);
        const float e0 = expf(a - m), e1 = expf(c - m);
        const bool pos = e1 / (e0 + e1) > pixel_thr;
        unsigned bits = 0;
        if (pos)
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const float la = out2[((size_t)b * 16 + 2 * n) * HW + i], lc = out2[((size_t)b * 16 + 2 * n + 1) * HW + i];
                const float lm = fmaxf(la, lc);
                const float f0 = expf(la - lm), f1 = expf(lc - lm);
                if (f1 / (f0 + f1) > link_thr) bits |= 1u << n;
            }
        lk[i] = (unsigned char)bits;
        lab[i] = pos ? i : -1;
    }
    __syncthreads();
    // union-find in LDS (one pass over the edges + one flatten pass).  Every link names a pixel with a SMALLER index as parent, so the
    // root of a tree is the component's first pixel in raster order whatever the interleaving; edges are symmetric, so each pixel only
    // looks at its four forward neighbours (0,+1) (+1,+1) (+1,0) (+1,-1).  The min-label sweeps this replaces needed one block-wide
    // sweep per step of the longest label path of the image (hundreds on a percolating random map): 1.40 ms -> 0.10 ms at B = 32.
    for (int i = tid; i < HW; i += 1024) {
        if (lab[i] < 0) continue;
        const int y = i / W, x = i - y * W;
        const unsigned mybits = lk[i];
#pragma unroll
        for (int n = 3; n < 7; ++n) {
            const int yy = y + dy[n], xx = x + dx[n];
            if (yy >= H || xx < 0 || xx >= W) continue;
            const int q = yy * W + xx;
            if (lab[q] < 0) continue;
            if (!(((mybits >> n) & 1) || ((lk[q] >> ((n + 4) & 7)) & 1))) continue;
            int ra = i, rb = q;
            while (true) {
                for (int nx = lab[ra]; nx != ra; nx = lab[ra]) ra = nx;
                for (int nx = lab[rb]; nx != rb; nx = lab[rb]) rb = nx;
                if (ra == rb) break;
                if (ra < rb) {
                    const int t = ra;
                    ra = rb;
                    rb = t;
                }
                const int old = atomicMin(&lab[ra], rb);       // ra > rb: hang the larger root under the smaller
                if (old == ra) break;                          // ra was still a root: joined
                ra = old;                                      // somebody re-parented ra meanwhile: join that parent with rb instead
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < HW; i += 1024) {
        int rt = lab[i];
        if (rt < 0) continue;
        for (int nx = lab[rt]; nx != rt; nx = lab[rt]) rt = nx;
        lab[i] = rt;                                           // only ever replaces an ancestor by the root: concurrent walks stay valid
    }
    __syncthreads();
    // rank the roots (lab[i] == i) in raster order: serial over <= HW by one thread per 1024-chunk would do; a block scan is simpler
    __shared__ int cnt[1024];
    const int per = (HW + 1023) / 1024;
    int mine = 0;
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        if (i < HW && lab[i] == i) ++mine;
    }
    cnt[tid] = mine;
    __syncthreads();
    if (tid < 64) {                                                          // wave 0: 16 entries per lane, then a 64-lane scan
        int loc[16], run = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            loc[t] = run;
            run += cnt[tid * 16 + t];
        }
        int inc = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (tid >= o) inc += v;
        }
        const int excl = inc - run;
#pragma unroll
        for (int t = 0; t < 16; ++t) cnt[tid * 16 + t] = excl + loc[t];
        if (tid == 63) nroot = inc;
    }
    __syncthreads();
    int base = cnt[tid];
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        if (i < HW && lab[i] == i) labels[(size_t)b * HW + i] = ++base;       // roots first
    }
    __syncthreads();
    __threadfence_block();
    for (int i = tid; i < HW; i += 1024) {
        const int l = lab[i];
        if (l < 0) labels[(size_t)b * HW + i] = 0;
    }
    __syncthreads();
    for (int i = tid; i < HW; i += 1024) {
        const int l = lab[i];
        if (l >= 0 && l != i) labels[(size_t)b * HW + i] = labels[(size_t)b * HW + l];
    }
    __syncthreads();
    // per-component statistics: count, min x, min y, max x, max y, score sum (score = softmax probability of class 1).  Accumulated in
    // LDS (the label array's neighbour `lk` and the scan buffer are dead): thousands of pixels of one component adding to the same six
    // GLOBAL addresses serialise at the L2 atomic unit (0.25 ms per launch).
    __shared__ float cstat[PL_STAT_COMPS * 6];
    const int ncs = max_comp < PL_STAT_COMPS ? max_comp : PL_STAT_COMPS;
    for (int i = tid; i < ncs * 6; i += 1024) {
        const int f = i % 6;
        cstat[i] = (f == 1 || f == 2) ? 1e9f : (f == 3 || f == 4) ? -1.f : 0.f;
    }
    float* cb = comps + (size_t)b * max_comp * 6;
    for (int i = ncs * 6 + tid; i < max_comp * 6; i += 1024) {               // components beyond the LDS table: reported empty
        const int f = i % 6;
        cb[i] = (f == 1 || f == 2) ? 1e9f : (f == 3 || f == 4) ? -1.f : 0.f;
    }
    __syncthreads();
    for (int i0 = 0; i0 < HW; i0 += 1024) {
        const int i = i0 + tid;
        int id = -1;
        float fx = 0.f, fy = 0.f, sc = 0.f;
        if (i < HW && lab[i] >= 0) {
            id = labels[(size_t)b * HW + i] - 1;
            if (id >= ncs) id = -1;
            const int y = i / W, x = i - y * W;
            fx = (float)x;
            fy = (float)y;
            const float a = out1[((size_t)b * 2) * HW + i], c = out1[((size_t)b * 2 + 1) * HW + i];
            const float m = fmaxf(a, c);
            const float e0 = expf(a - m), e1 = expf(c - m);
            sc = e1 / (e0 + e1);
        }
        // a wave whose active pixels all belong to ONE component (the usual case inside a large component: 64 consecutive pixels of
        // a row) reduces first and issues six atomics instead of 6 x 64 on the same addresses
        const unsigned long long act = __ballot(id >= 0);
        if (act == 0ull) continue;
        const int id0 = __builtin_amdgcn_readlane(id, __builtin_ctzll(act));
        if (__ballot(id >= 0 && id != id0) == 0ull) {
            float cnt = id >= 0 ? 1.f : 0.f, ssum = id >= 0 ? sc : 0.f;   // pixels beyond the component table (id reset to -1) add nothing
            float mnx = id >= 0 ? fx : 1e9f, mny = id >= 0 ? fy : 1e9f, mxx = id >= 0 ? fx : -1.f, mxy = id >= 0 ? fy : -1.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                cnt += __shfl_xor(cnt, o, 64);
                ssum += __shfl_xor(ssum, o, 64);
                mnx = fminf(mnx, __shfl_xor(mnx, o, 64));
                mny = fminf(mny, __shfl_xor(mny, o, 64));
                mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64));
                mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
            }
            if ((tid & 63) == 0) {
                atomicAdd(cstat + id0 * 6 + 0, cnt);
                atomicMin(reinterpret_cast<int*>(cstat + id0 * 6 + 1), __float_as_int(mnx));      // non-negative floats order like ints
                atomicMin(reinterpret_cast<int*>(cstat + id0 * 6 + 2), __float_as_int(mny));
                atomicMax(reinterpret_cast<int*>(cstat + id0 * 6 + 3), __float_as_int(mxx));
                atomicMax(reinterpret_cast<int*>(cstat + id0 * 6 + 4), __float_as_int(mxy));
                atomicAdd(cstat + id0 * 6 + 5, ssum);
            }
        } else if (id >= 0) {
            atomicAdd(cstat + id * 6 + 0, 1.f);
            atomicMin(reinterpret_cast<int*>(cstat + id * 6 + 1), __float_as_int(fx));
            atomicMin(reinterpret_cast<int*>(cstat + id * 6 + 2), __float_as_int(fy));
            atomicMax(reinterpret_cast<int*>(cstat + id * 6 + 3), __float_as_int(fx));
            atomicMax(reinterpret_cast<int*>(cstat + id * 6 + 4), __float_as_int(fy));
            atomicAdd(cstat + id * 6 + 5, sc);
        }
    }
    __syncthreads();
    for (int i = tid; i < ncs * 6; i += 1024) cb[i] = cstat[i];
    if (tid == 0) ncomp[b] = nroot;
}

}  // namespace

extern "C" int gssd_interp_add_f32(const float* src, const float* addend, float* out, float* out2, int B, int Hs, int Ws, int Hd,
                                   int Wd, int C, gssd_stream_t stream) {
    GSSD_CHECK_ARG(src && out && B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && ((addend == nullptr) == (out2 == nullptr)));
    const long long total = (long long)B * Hd * Wd * C;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(interp_add_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, addend, out, out2, B, Hs, Ws, Hd, Wd, C);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pixellink_final_f32(const float* f0, const float* f1, const float* f2, const float* f3, int nf, const float* w1,
                                        const float* b1, const float* w2, const float* b2, float* out1, float* out2, int B, int HW,
                                        gssd_stream_t stream) {
    GSSD_CHECK_ARG(f0 && w1 && b1 && w2 && b2 && out1 && out2 && B > 0 && HW > 0 && nf >= 1 && nf <= 4);
    GSSD_CHECK_ARG((nf < 2 || f1) && (nf < 3 || f2) && (nf < 4 || f3));
    const long long total = (long long)B * HW;
    hipLaunchKernelGGL(pl_final_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, as_stream(stream), f0, f1, f2, f3, nf, w1, b1, w2,
                       b2, out1, out2, B, HW);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pixellink_loss_f32(const float* out1, const float* out2, const long long* pixel_target,
                                       const unsigned char* neg_pixel_mask, const float* pixel_pos_weight, const long long* link_target,
                                       double* per_image, float* neg_weight_out, int B, int H, int W, int neg_pos_ratio,
                                       gssd_stream_t stream) {
    GSSD_CHECK_ARG(out1 && out2 && pixel_target && neg_pixel_mask && pixel_pos_weight && link_target && per_image);
    GSSD_CHECK_ARG(B > 0 && H > 0 && W > 0 && H * W <= PL_SORT && neg_pos_ratio > 0);
    hipLaunchKernelGGL(pl_loss_kernel, dim3(B), dim3(1024), 0, as_stream(stream), out1, out2, pixel_target, neg_pixel_mask,
                       pixel_pos_weight, link_target, per_image, neg_weight_out, H * W, neg_pos_ratio);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_interp_add_bwd_f32(const float* dout, const float* dout2, float* dsrc, float* daddend, int B, int Hs, int Ws, int Hd,
                                       int Wd, int C, int ld, gssd_stream_t stream) {
    GSSD_CHECK_ARG((dout || dout2) && dsrc && B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && ld >= C);
    const long long total = (long long)B * Hd * Wd * C;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(interp_add_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dout, dout2, dsrc, daddend, B, Hs, Ws, Hd, Wd,
                       C, ld);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pixellink_final_bwd_f32(const float* d_out1, const float* d_out2, const float* f0, const float* f1, const float* f2,
                                            const float* f3, int nf, const float* w1, const float* w2, float* g0, float* g1, float* g2,
                                            float* g3, int accumulate_mask, double* dw1, double* dw2, int B, int HW, int ld,
                                            gssd_stream_t stream) {
    GSSD_CHECK_ARG(d_out1 && d_out2 && f0 && w1 && w2 && g0 && dw1 && dw2 && B > 0 && HW > 0 && nf >= 1 && nf <= 4 && ld >= 18);
    GSSD_CHECK_ARG((nf < 2 || (f1 && g1)) && (nf < 3 || (f2 && g2)) && (nf < 4 || (f3 && g3)));
    const long long total = (long long)B * HW;
    hipLaunchKernelGGL(pl_final_bwd_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, as_stream(stream), d_out1, d_out2, nf, w1, w2,
                       g0, g1, g2, g3, accumulate_mask, B, HW, ld);
    GSSD_CHECK_LAUNCH();
    const int blocks = (int)((total + 127) / 128 > 1024 ? 1024 : (total + 127) / 128);
    hipLaunchKernelGGL(pl_final_wgrad_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), d_out1, d_out2, f0, f1, f2, f3, nf, dw1, dw2,
                       B, HW);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pixellink_loss_bwd_f32(const float* out1, const float* out2, const long long* pixel_target, const float* neg_weight,
                                           const float* pixel_pos_weight, const long long* link_target, const double* per_image,
                                           const float* upstream4, float* d_out1, float* d_out2, int B, int H, int W,
                                           gssd_stream_t stream) {
    GSSD_CHECK_ARG(out1 && out2 && pixel_target && neg_weight && pixel_pos_weight && link_target && per_image && upstream4 && d_out1 &&
                   d_out2 && B > 0 && H > 0 && W > 0);
    hipLaunchKernelGGL(pl_loss_bwd_kernel, dim3(B), dim3(1024), 0, as_stream(stream), out1, out2, pixel_target, neg_weight,
                       pixel_pos_weight, link_target, per_image, upstream4, d_out1, d_out2, B, H * W);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pixellink_decode_f32(const float* out1, const float* out2, int* labels, float* comps, int* ncomp, int B, int H,
                                         int W, float pixel_thr, float link_thr, int max_comp, gssd_stream_t stream) {
    GSSD_CHECK_ARG(out1 && out2 && labels && comps && ncomp && B > 0 && H > 0 && W > 0 && H * W <= PL_MAXPIX && max_comp > 0);
    hipLaunchKernelGGL(pl_decode_kernel, dim3(B), dim3(1024), 0, as_stream(stream), out1, out2, labels, comps, ncomp, H, W, pixel_thr,
                       link_thr, max_comp);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
