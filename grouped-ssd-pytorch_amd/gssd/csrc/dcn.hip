// DCNv2 sampling for gfx950: modulated, bilinearly sampled column matrix of a 3x3 / stride 1 / pad 1
// deformable convolution (the gather half of the reference's external `dcn_v2` extension called at
// layers/dcn_v2_custom.py:84-89; algorithm restated in oracle/gssd_oracle.py::dcn_v2_conv --
// parity unpinned, see there).  One wave per (pixel, tap, deformable group): the sampling position and
// the four bilinear weights are wave-uniform, each lane moves float4 channel slices, so every global
// access is a contiguous 16 B x 64 lane burst in NHWC.  The contraction with the weights is a plain
// 1x1 implicit GEMM over the columns (conv_igemm.hip).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                         float* __restrict__ cols, int B, int H, int W, int C, int dg,
                                                         int om_stride) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int HW = H * W, cpg = C / dg, cpg4 = cpg >> 2;
    const long long units = (long long)B * HW * 9 * dg;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (long long u = wave0; u < units; u += nwaves) {
        const int d = (int)(u % dg);
        long long t = u / dg;
        const int tap = (int)(t % 9);
        const long long bp = t / 9;            // b*HW + p
        const int p = (int)(bp % HW);
        const int b = (int)(bp / HW);
        const int h = p / W, w = p - h * W;
        const float* omp = om + bp * om_stride;
        const float dy = omp[d * 18 + 2 * tap];
        const float dx = omp[d * 18 + 2 * tap + 1];
        const float ml = omp[dg * 18 + d * 9 + tap];
        const float m = 1.f / (1.f + expf(-ml));            // torch.sigmoid (dcn_v2_custom.py:83)
        const float py = (float)(h - 1 + tap / 3) + dy;
        const float px = (float)(w - 1 + tap % 3) + dx;
        float* dst = cols + (bp * 9 + tap) * C + d * cpg;
        if (!(py > -1.f && px > -1.f && py < (float)H && px < (float)W)) {
            for (int c = lane; c < cpg4; c += 64) reinterpret_cast<f32x4*>(dst)[c] = zero4;
            continue;
        }
        const float y0f = floorf(py), x0f = floorf(px);
        const int y0 = (int)y0f, x0 = (int)x0f;
        const float ly = py - y0f, lx = px - x0f, hy = 1.f - ly, hx = 1.f - lx;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
        const float w00 = (y0ok && x0ok) ? hy * hx * m : 0.f;
        const float w01 = (y0ok && x1ok) ? hy * lx * m : 0.f;
        const float w10 = (y1ok && x0ok) ? ly * hx * m : 0.f;
        const float w11 = (y1ok && x1ok) ? ly * lx * m : 0.f;
        const float* xb = x + (size_t)b * HW * C + d * cpg;
        const f32x4* p00 = reinterpret_cast<const f32x4*>(xb + (size_t)((y0ok ? y0 : 0) * W + (x0ok ? x0 : 0)) * C);
        const f32x4* p01 = reinterpret_cast<const f32x4*>(xb + (size_t)((y0ok ? y0 : 0) * W + (x1ok ? x0 + 1 : 0)) * C);
        const f32x4* p10 = reinterpret_cast<const f32x4*>(xb + (size_t)((y1ok ? y0 + 1 : 0) * W + (x0ok ? x0 : 0)) * C);
        const f32x4* p11 = reinterpret_cast<const f32x4*>(xb + (size_t)((y1ok ? y0 + 1 : 0) * W + (x1ok ? x0 + 1 : 0)) * C);
        for (int c = lane; c < cpg4; c += 64) {
            const f32x4 v = p00[c] * w00 + p01[c] * w01 + p10[c] * w10 + p11[c] * w11;
            reinterpret_cast<f32x4*>(dst)[c] = v;
        }
    }
}

// Backward of the sampling.  d(cols) arrives from the dgrad of the 1x1 GEMM.  d(x) contributions are accumulated in an LDS
// window of (tile + 2R + 1)^2 pixels around the tile (trained offsets are a few pixels) and flushed to HBM with ONE fp32
// atomic per window element; corners outside the window fall back to global atomics, so any offset is handled.  The three
// scalars d(offset_y), d(offset_x), d(mask logit) are wave-reduced over the slice and added to d(om) (zero-filled by the
// caller).  The corner validity masks and floor() are constants of the differentiation, exactly as in the gather
// formulation (autograd of the oracle graph).
constexpr int DC_R = 2;                                               // window halo (pixels) around the tile
constexpr int DC_U = 4;                                               // (pixel, tap) units in flight per wave

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 64 lanes, result wave-uniform: quad xor 1, xor 2, half-row mirror, row mirror (DPP), then the four rows
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
}

// One wave per workgroup: a TH x TW tile of output pixels and one 64-channel slice (lane = channel).  The window is private
// to the wave and lanes never share an address, so the d(x) accumulation is a plain LDS read-modify-write (LDS float
// atomics cost ~160 clocks per wave instruction on gfx950 -- measured; they were 70 % of the first version).
template <int TH, int TW>
__global__ __launch_bounds__(64) void dcn_col2im_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                        const float* __restrict__ dcols, float* __restrict__ dx,
                                                        float* __restrict__ dom, int B, int H, int W, int C, int dg,
                                                        int om_stride, int tiles_y, int tiles_x) {
    constexpr int WH = TH + 2 * DC_R + 1, WW = TW + 2 * DC_R + 1;     // +1: the far bilinear corner
    __shared__ float win[WH * WW * 64];
    __shared__ float oms[TH * TW * 27];                               // this group's (dy, dx) x 9 and 9 mask logits per pixel
    const int lane = threadIdx.x;
    const int HW = H * W, cpg = C / dg;
    const int slices = C / 64;
    int bid = blockIdx.x;
    const int sl = bid % slices;  bid /= slices;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ch0 = sl * 64, d = ch0 / cpg;
    const int wy0 = ty * TH - DC_R, wx0 = tx * TW - DC_R;             // window origin (may be negative)
#pragma unroll
    for (int i = 0; i < WH * WW; ++i) win[i * 64 + lane] = 0.f;
    for (int i = lane; i < TH * TW * 27; i += 64) {
        const int pl = i / 27, j = i - pl * 27;
        const int h = ty * TH + pl / TW, w = tx * TW + pl % TW;
        float v = 0.f;
        if (h < H && w < W) {
            const float* omp = om + ((long long)b * HW + h * W + w) * om_stride;
            v = j < 18 ? omp[d * 18 + j] : omp[dg * 18 + d * 9 + (j - 18)];
        }
        oms[i] = v;
    }
    __syncthreads();
    const float* xb = x + (size_t)b * HW * C + ch0 + lane;
    float* dxb = dx + (size_t)b * HW * C + ch0 + lane;
    constexpr int NU = TH * TW * 9;
    static_assert(NU % DC_U == 0, "tile");
    for (int u0 = 0; u0 < NU; u0 += DC_U) {
        float g[DC_U], v00[DC_U], v01[DC_U], v10[DC_U], v11[DC_U], ly[DC_U], lx[DC_U], m[DC_U];
        int y0[DC_U], x0[DC_U], kk[DC_U];
        long long bp[DC_U];
        // ---- phase 1: sampling geometry (wave-uniform) and all loads of DC_U units -------------------------------------------
#pragma unroll
        for (int q = 0; q < DC_U; ++q) {
            const int u = u0 + q;
            const int tap = u % 9, pl = u / 9;
            const int h = ty * TH + pl / TW, w = tx * TW + pl % TW;
            bp[q] = (long long)b * HW + h * W + w;
            const float oy = oms[pl * 27 + 2 * tap], ox = oms[pl * 27 + 2 * tap + 1];
            m[q] = 1.f / (1.f + expf(-oms[pl * 27 + 18 + tap]));
            const float py = (float)(h - 1 + tap / 3) + oy;
            const float px = (float)(w - 1 + tap % 3) + ox;
            const bool ok = h < H && w < W && py > -1.f && px > -1.f && py < (float)H && px < (float)W;
            const float y0f = floorf(py), x0f = floorf(px);
            y0[q] = (int)y0f;
            x0[q] = (int)x0f;
            ly[q] = py - y0f;
            lx[q] = px - x0f;
            const bool y0ok = y0[q] >= 0, y1ok = y0[q] + 1 <= H - 1, x0ok = x0[q] >= 0, x1ok = x0[q] + 1 <= W - 1;
            kk[q] = ok ? ((y0ok && x0ok) ? 1 : 0) | ((y0ok && x1ok) ? 2 : 0) | ((y1ok && x0ok) ? 4 : 0) |
                             ((y1ok && x1ok) ? 8 : 0) | 16
                       : 0;
            g[q] = (kk[q] & 16) ? dcols[(bp[q] * 9 + tap) * C + ch0 + lane] : 0.f;
            v00[q] = (kk[q] & 1) ? xb[(size_t)(y0[q] * W + x0[q]) * C] : 0.f;
            v01[q] = (kk[q] & 2) ? xb[(size_t)(y0[q] * W + x0[q] + 1) * C] : 0.f;
            v10[q] = (kk[q] & 4) ? xb[(size_t)((y0[q] + 1) * W + x0[q]) * C] : 0.f;
            v11[q] = (kk[q] & 8) ? xb[(size_t)((y0[q] + 1) * W + x0[q] + 1) * C] : 0.f;
        }
        // ---- phase 2: gradients ---------------------------------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < DC_U; ++q) {
            if (!(kk[q] & 16)) continue;
            const int tap = (u0 + q) % 9;
            const float hy = 1.f - ly[q], hx = 1.f - lx[q];
            const float s_m = g[q] * (v00[q] * (hy * hx) + v01[q] * (hy * lx[q]) + v10[q] * (ly[q] * hx) + v11[q] * (ly[q] * lx[q]));
            const float s_y = g[q] * ((v10[q] - v00[q]) * hx + (v11[q] - v01[q]) * lx[q]);
            const float s_x = g[q] * ((v01[q] - v00[q]) * hy + (v11[q] - v10[q]) * ly[q]);
            const float gm = g[q] * m[q];
            const int wy = y0[q] - wy0, wx = x0[q] - wx0;            // wave-uniform
            if (wy >= 0 && wx >= 0 && wy + 1 < WH && wx + 1 < WW) {
                float* wp = win + (wy * WW + wx) * 64 + lane;
                if (kk[q] & 1) wp[0] += gm * (hy * hx);
                if (kk[q] & 2) wp[64] += gm * (hy * lx[q]);
                if (kk[q] & 4) wp[WW * 64] += gm * (ly[q] * hx);
                if (kk[q] & 8) wp[WW * 64 + 64] += gm * (ly[q] * lx[q]);
            } else {
                float* gp = dxb + (size_t)(y0[q] * W + x0[q]) * C;
                if (kk[q] & 1) unsafeAtomicAdd(gp, gm * (hy * hx));
                if (kk[q] & 2) unsafeAtomicAdd(gp + C, gm * (hy * lx[q]));
                if (kk[q] & 4) unsafeAtomicAdd(gp + (size_t)W * C, gm * (ly[q] * hx));
                if (kk[q] & 8) unsafeAtomicAdd(gp + (size_t)W * C + C, gm * (ly[q] * lx[q]));
            }
            const float t_m = wave_sum_dpp(s_m), t_y = wave_sum_dpp(s_y), t_x = wave_sum_dpp(s_x);
            if (lane == 0) {
                float* domp = dom + bp[q] * om_stride;
                unsafeAtomicAdd(domp + d * 18 + 2 * tap, t_y * m[q]);
                unsafeAtomicAdd(domp + d * 18 + 2 * tap + 1, t_x * m[q]);
                unsafeAtomicAdd(domp + dg * 18 + d * 9 + tap, t_m * m[q] * (1.f - m[q]));
            }
        }
    }
    for (int i = 0; i < WH * WW; ++i) {
        const int y = wy0 + i / WW, xx = wx0 + i % WW;
        if ((unsigned)y >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
        const float v = win[i * 64 + lane];
        if (v != 0.f) unsafeAtomicAdd(dxb + (size_t)(y * W + xx) * C, v);
    }
}

}  // namespace

extern "C" int gssd_dcn_im2col_f32(const float* x, const float* om, float* cols, int B, int H, int W, int C, int dg,
                                   int om_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && cols && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0);
    GSSD_CHECK_ARG(C % (4 * dg) == 0 && om_stride >= 27 * dg);
    const long long units = (long long)B * H * W * 9 * dg;
    long long blocks = (units + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dcn_im2col_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, om, cols, B, H, W, C, dg,
                       om_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_col2im_f32(const float* x, const float* om, const float* dcols, float* dx, float* dom, int B, int H,
                                   int W, int C, int dg, int om_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && dcols && dx && dom && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0);
    GSSD_CHECK_ARG(C % dg == 0 && (C / dg) % 64 == 0 && om_stride >= 27 * dg);
    constexpr int TH = 4, TW = 8;
    const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
    const long long blocks = (long long)B * tiles_y * tiles_x * (C / 64);
    GSSD_CHECK_ARG(blocks < (1ll << 31));
    hipLaunchKernelGGL((dcn_col2im_kernel<TH, TW>), dim3((int)blocks), dim3(64), 0, as_stream(stream), x, om, dcols, dx, dom, B, H, W, C,
                       dg, om_stride, tiles_y, tiles_x);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
