// DCNv2 sampling for gfx950: modulated, bilinearly sampled column matrix of a 3x3 / stride 1 / pad 1
// deformable convolution (the gather half of the reference's external `dcn_v2` extension called at
// layers/dcn_v2_custom.py:84-89; algorithm restated in oracle/gssd_oracle.py::dcn_v2_conv --
// parity unpinned, see there).  One wave per (pixel, tap, deformable group): the sampling position and
// the four bilinear weights are wave-uniform, each lane moves float4 channel slices, so every global
// access is a contiguous 16 B x 64 lane burst in NHWC.  The contraction with the weights is a plain
// 1x1 implicit GEMM over the columns (conv_igemm.hip).
#include "common.h"

#ifndef COL2IM_KO
#define COL2IM_KO 0      // scripts/col2im_knockout.sh
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// element type of the sampled map / of the column matrix: fp32, or bf16 for the training step of the bf16 storage mode (the forward
// sampled the SAME bf16 map, interpolated in fp32 and rounded its columns to bf16: csrc/dcn_bf16.hip)
template <typename T>
__device__ __forceinline__ f32x4 ld4c(const T* p, int c);
template <>
__device__ __forceinline__ f32x4 ld4c<float>(const float* p, int c) {
    return reinterpret_cast<const f32x4*>(p)[c];
}
template <>
__device__ __forceinline__ f32x4 ld4c<unsigned short>(const unsigned short* p, int c) {
    const uint2 v = reinterpret_cast<const uint2*>(p)[c];
    return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                 __uint_as_float(v.y & 0xffff0000u)};
}
__device__ __forceinline__ void st4c(float* p, int c, const f32x4 v) { reinterpret_cast<f32x4*>(p)[c] = v; }
__device__ __forceinline__ void st4c(unsigned short* p, int c, const f32x4 v) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<bf16x4*>(p)[c] = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

template <typename T>
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                         T* __restrict__ cols, int B, int H, int W, int C, int dg,
                                                         int om_stride) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int HW = H * W, cpg = C / dg, cpg4 = cpg >> 2;
    const long long units = (long long)B * HW * 9 * dg;
    // A wave takes 64 consecutive units: lane l computes unit l's sampling geometry ONCE (sigmoid, floor, the bounds tests -- ~60 VALU
    // instructions that every lane used to repeat for every unit: the kernel was bound by them, not by memory), then the units are
    // walked IM_UN at a time with their geometry broadcast by v_readlane and all 4 * IM_UN corner loads in flight before the first blend.
    // Branch free: a sample outside the map reads pixel 0 with zero weights.
    constexpr int IM_UN = sizeof(T) == 2 ? 2 : 4;          // measured: bf16 0.42 ms with 2 (0.46 with 4), fp32 0.86 ms with 4 (0.91 with 2)
    for (long long base = wave0 * 64; base < units; base += nwaves * 64) {
        float g_w00, g_w01, g_w10, g_w11;
        int g_o00, g_o01, g_o10, g_o11, g_dst;
        {
            const long long u = base + lane;
            const long long uu = u < units ? u : units - 1;
            const int d = (int)(uu % dg);
            long long t = uu / dg;
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
            const bool in = py > -1.f && px > -1.f && py < (float)H && px < (float)W;
            const float y0f = floorf(py), x0f = floorf(px);
            const int y0 = in ? (int)y0f : 0, x0 = in ? (int)x0f : 0;
            const float ly = py - y0f, lx = px - x0f, hy = 1.f - ly, hx = 1.f - lx;
            const bool y0ok = in && y0 >= 0, y1ok = in && y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
            g_w00 = (y0ok && x0ok) ? hy * hx * m : 0.f;
            g_w01 = (y0ok && x1ok) ? hy * lx * m : 0.f;
            g_w10 = (y1ok && x0ok) ? ly * hx * m : 0.f;
            g_w11 = (y1ok && x1ok) ? ly * lx * m : 0.f;
            const int xb = b * HW * C + d * cpg;                 // (element offsets fit 32 bits: checked by the launcher)
            g_o00 = xb + ((y0ok ? y0 : 0) * W + (x0ok ? x0 : 0)) * C;
            g_o01 = xb + ((y0ok ? y0 : 0) * W + (x1ok ? x0 + 1 : 0)) * C;
            g_o10 = xb + ((y1ok ? y0 + 1 : 0) * W + (x0ok ? x0 : 0)) * C;
            g_o11 = xb + ((y1ok ? y0 + 1 : 0) * W + (x1ok ? x0 + 1 : 0)) * C;
            g_dst = (int)((bp * 9 + tap) * C + d * cpg);
        }
        const int nj = (int)((units - base) < 64 ? (units - base) : 64);
        auto bf = [](float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };
        for (int j = 0; j < nj; j += IM_UN) {
            float wq[IM_UN][4];
            const T* pq[IM_UN][4];
            T* dq[IM_UN];
#pragma unroll
            for (int k = 0; k < IM_UN; ++k) {
                const int jk = j + k < nj ? j + k : nj - 1;        // (a ragged tail repeats the last unit: same values stored again)
                wq[k][0] = bf(g_w00, jk);
                wq[k][1] = bf(g_w01, jk);
                wq[k][2] = bf(g_w10, jk);
                wq[k][3] = bf(g_w11, jk);
                pq[k][0] = x + __builtin_amdgcn_readlane(g_o00, jk);
                pq[k][1] = x + __builtin_amdgcn_readlane(g_o01, jk);
                pq[k][2] = x + __builtin_amdgcn_readlane(g_o10, jk);
                pq[k][3] = x + __builtin_amdgcn_readlane(g_o11, jk);
                dq[k] = cols + __builtin_amdgcn_readlane(g_dst, jk);
            }
            for (int c = lane; c < cpg4; c += 64) {
                f32x4 v[IM_UN][4];
#pragma unroll
                for (int k = 0; k < IM_UN; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[k][q] = ld4c<T>(pq[k][q], c);
#pragma unroll
                for (int k = 0; k < IM_UN; ++k)
                    st4c(dq[k], c, v[k][0] * wq[k][0] + v[k][1] * wq[k][1] + v[k][2] * wq[k][2] + v[k][3] * wq[k][3]);
            }
        }
    }
}

// Backward of the sampling.  d(cols) arrives from the dgrad of the contraction.  A workgroup owns a TH x TW tile of output pixels
// and one 64-channel slice (lane = channel) and is THREE waves with different jobs over the same (pixel, tap) units:
//   waves 0, 1 (reduce, alternate units):  d(offset_y), d(offset_x), d(mask logit) = wave sums over the slice of g * (bilinear
//                     derivative of x) -- the x corners come from an LDS copy of the (tile + 2R + 1)^2 window, staged once by LDS-DMA;
//   wave 2 (scatter): d(x) += g * m * bilinear weight into an LDS accumulator window, flushed with ONE fp32 atomic per element.
// Both read d(cols) from a double-buffered LDS-DMA stage (2 pixels x 9 taps per chunk) and the per-unit sampling geometry from a
// table built once per tile, so the main loop has no global load at all: the first version (one wave doing both, 5 global loads per
// unit, 4 waves per CU) was latency bound at 8.1 ms; this one is bound by the VALU work of the two jobs running on separate SIMDs.
// Units whose corners leave the window (offsets beyond R pixels; ~10 % with |offset| ~ 0.5) are skipped here and handled by
// dcn_col2im_overflow_kernel straight from global memory, so any offset is handled.  The corner validity masks and
// floor() are constants of the differentiation, exactly as in the gather formulation (autograd of the oracle).
constexpr int DC_R = 2;                                               // window halo (pixels) around the tile
constexpr int DC_CH = 18;                                             // units per d(cols) chunk: 2 pixels x 9 taps
constexpr int DC_CHP = 20;                                            // ... padded to whole 1-KiB DMA pieces (4 units each)

__device__ __attribute__((aligned(16))) float g_zero16_dcn[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16_dcn(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

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

// geometry word: bits 0-3 corner validity (00, 01, 10, 11), bit 4 unit contributes, bit 5 all four corners inside the LDS window,
// bits 8.. window offset (pixels) of the (y0, x0) corner
struct DcGeo {
    int gi, gyx;
    float ly, lx, m;
};
template <int TH, int TW>
__device__ __forceinline__ DcGeo dc_geometry(const float* __restrict__ om, int b, int h, int w, int tap, int d, int dg, int H, int W,
                                             int om_stride, int wy0, int wx0) {
    constexpr int WH = TH + 2 * DC_R + 1, WW = TW + 2 * DC_R + 1;
    DcGeo r = {0, 0, 0.f, 0.f, 0.f};
    if (h < H && w < W) {
        const float* omp = om + ((long long)b * H * W + h * W + w) * om_stride;
        const float oy = omp[d * 18 + 2 * tap], ox = omp[d * 18 + 2 * tap + 1];
        r.m = 1.f / (1.f + expf(-omp[dg * 18 + d * 9 + tap]));
        const float py = (float)(h - 1 + tap / 3) + oy;
        const float px = (float)(w - 1 + tap % 3) + ox;
        if (py > -1.f && px > -1.f && py < (float)H && px < (float)W) {
            const float y0f = floorf(py), x0f = floorf(px);
            const int y0 = (int)y0f, x0 = (int)x0f;
            r.ly = py - y0f;
            r.lx = px - x0f;
            const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
            r.gi = ((y0ok && x0ok) ? 1 : 0) | ((y0ok && x1ok) ? 2 : 0) | ((y1ok && x0ok) ? 4 : 0) | ((y1ok && x1ok) ? 8 : 0) | 16;
            const int wy = y0 - wy0, wx = x0 - wx0;
            if (wy >= 0 && wx >= 0 && wy + 1 < WH && wx + 1 < WW) r.gi |= 32 | ((wy * WW + wx) << 8);
            r.gyx = (y0 & 0xFFFF) | (x0 << 16);
        }
    }
    return r;
}

template <int TH, int TW>
__global__ __launch_bounds__(192) void dcn_col2im_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                         const float* __restrict__ dcols, float* __restrict__ dx,
                                                         float* __restrict__ dom, int B, int H, int W, int C, int dg,
                                                         int om_stride, int tiles_y, int tiles_x) {
    constexpr int WH = TH + 2 * DC_R + 1, WW = TW + 2 * DC_R + 1;     // +1: the far bilinear corner
    constexpr int WPX = WH * WW, WPX4 = (WPX + 3) / 4 * 4;
    constexpr int NU = TH * TW * 9, NCH = NU / DC_CH;
    static_assert(NU % DC_CH == 0 && TW % 2 == 0, "tile");
    __shared__ __attribute__((aligned(16))) float xw[WPX4 * 64];      // x window
    __shared__ __attribute__((aligned(16))) float gb[2][DC_CHP * 64]; // d(cols) chunks
    __shared__ float dw[WPX * 64];                                    // d(x) accumulator window
    __shared__ int geo_i[NU];
    __shared__ float geo_ly[NU], geo_lx[NU], geo_m[NU];
    __shared__ float dacc[NU * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * W, cpg = C / dg;
    const int slices = C / 64;
    int bid = blockIdx.x;
    const int sl = bid % slices;  bid /= slices;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ch0 = sl * 64, d = ch0 / cpg;
    const int wy0 = ty * TH - DC_R, wx0 = tx * TW - DC_R;             // window origin (may be negative)
    const float* xb = x + (size_t)b * HW * C + ch0;
    float* dxb = dx + (size_t)b * HW * C + ch0 + lane;
    const float* dcb = dcols + (size_t)b * HW * 9 * C + ch0;

    auto stage_chunk = [&](int c, float* buf) {                       // wave 0 only: 5 DMA pieces of 4 units x 256 B
        const int q4 = lane >> 4, f4 = (lane & 15) * 4;
#pragma unroll
        for (int j = 0; j < DC_CHP / 4; ++j) {
            const int u = j * 4 + q4;
            const int pl = c * 2 + (u >= 9 ? 1 : 0), tap = u >= 9 ? u - 9 : u;
            const int h = ty * TH + pl / TW, w = tx * TW + pl % TW;
            const bool ok = u < DC_CH && h < H && w < W;
            const float* src = ok ? dcb + ((size_t)(h * W + w) * 9 + tap) * C + f4 : g_zero16_dcn;
            dma16_dcn(src, buf + j * 256);
        }
    };
    // ---- prologue: x window (both waves), first d(cols) chunk, accumulator clear, geometry table --------------------------------
    {
        const int q4 = lane >> 4, f4 = (lane & 15) * 4;
        for (int i = wave; i < WPX4 / 4; i += 3) {
            const int px = i * 4 + q4;
            const int y = wy0 + px / WW, xx = wx0 + px % WW;
            const bool ok = px < WPX && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
            dma16_dcn(ok ? xb + (size_t)(y * W + xx) * C + f4 : g_zero16_dcn, xw + i * 256);
        }
    }
    if (wave == 0) stage_chunk(0, gb[0]);
    for (int i = wave; i < WPX; i += 3) dw[i * 64 + lane] = 0.f;
    for (int u = tid; u < NU; u += 192) {
        const int pl = u / 9, tap = u - pl * 9;
        const DcGeo r = dc_geometry<TH, TW>(om, b, ty * TH + pl / TW, tx * TW + pl % TW, tap, d, dg, H, W, om_stride, wy0, wx0);
        geo_i[u] = r.gi;
        geo_ly[u] = r.ly;
        geo_lx[u] = r.lx;
        geo_m[u] = r.m;
    }
    __syncthreads();                                                  // (drains the DMA: window, chunk 0) + table visible

    for (int c = 0; c < ((COL2IM_KO & 4) ? 0 : NCH); ++c) {
        const float* gbuf = gb[c & 1];
        if (wave == 0 && c + 1 < NCH && !(COL2IM_KO & 32)) stage_chunk(c + 1, gb[(c + 1) & 1]);
        // the chunk's geometry: lane q holds unit q's table row; a unit's (wave-uniform) values then come out of the registers by
        // v_readlane with a constant lane instead of one dependent LDS round trip per unit
        const int gl = lane < DC_CH ? lane : 0;
        const int v_gi = geo_i[c * DC_CH + gl];
        const float v_ly = geo_ly[c * DC_CH + gl], v_lx = geo_lx[c * DC_CH + gl], v_m = geo_m[c * DC_CH + gl];
        auto geo = [&](int q, int& gi, float& ly, float& lx, float& m) {
            gi = __builtin_amdgcn_readlane(v_gi, q);
            ly = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_ly), q));
            lx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_lx), q));
            m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_m), q));
        };
        if (wave < 2) {
            if (!(COL2IM_KO & 16))
            // reduce waves: units q = wave, wave + 2, ...; three units per trip, branch free (a unit that does not contribute reads
            // window pixel 0 and multiplies by 0), so the nine wave sums and the LDS reads of a trip interleave
#pragma unroll
            for (int q0 = 0; q0 < DC_CH; q0 += 6) {
                float s_m[3], s_y[3], s_x[3], mm[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    int gi0, gi1;
                    float ly0, lx0, m0, ly1, lx1, m1;
                    geo(q0 + 2 * k, gi0, ly0, lx0, m0);
                    geo(q0 + 2 * k + 1, gi1, ly1, lx1, m1);
                    const int gi = wave ? gi1 : gi0;
                    const float ly = wave ? ly1 : ly0, lx = wave ? lx1 : lx0;
                    mm[k] = wave ? m1 : m0;
                    const int q = q0 + 2 * k + wave;
                    const bool on = (gi & 48) == 48;
                    const float g = on ? gbuf[q * 64 + lane] : 0.f;
                    const float hy = 1.f - ly, hx = 1.f - lx;
                    const float* wp = xw + (on ? gi >> 8 : 0) * 64 + lane;
                    const float r00 = wp[0], r01 = wp[64], r10 = wp[WW * 64], r11 = wp[WW * 64 + 64];
                    const float v00 = (gi & 1) ? r00 : 0.f, v01 = (gi & 2) ? r01 : 0.f;
                    const float v10 = (gi & 4) ? r10 : 0.f, v11 = (gi & 8) ? r11 : 0.f;
                    s_m[k] = g * (v00 * (hy * hx) + v01 * (hy * lx) + v10 * (ly * hx) + v11 * (ly * lx));
                    s_y[k] = g * ((v10 - v00) * hx + (v11 - v01) * lx);
                    s_x[k] = g * ((v01 - v00) * hy + (v11 - v10) * ly);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int u = c * DC_CH + q0 + 2 * k + wave;
                    const float t_m = wave_sum_dpp(s_m[k]), t_y = wave_sum_dpp(s_y[k]), t_x = wave_sum_dpp(s_x[k]);
                    if (lane == 0) {
                        dacc[u * 3] = t_y * mm[k];
                        dacc[u * 3 + 1] = t_x * mm[k];
                        dacc[u * 3 + 2] = t_m * mm[k] * (1.f - mm[k]);
                    }
                }
            }
        } else if (!(COL2IM_KO & 8)) {
            // scatter wave: read-modify-write of the accumulator window; consecutive units may hit the same pixel, so they stay in
            // order.  Branch free (a unit that does not contribute adds 0 to window pixel 0): the d(cols) reads of the whole chunk
            // hoist above the chain
#pragma unroll
            for (int q = 0; q < DC_CH; ++q) {
                int gi;
                float ly, lx, m;
                geo(q, gi, ly, lx, m);
                const bool on = (gi & 48) == 48;
                const float hy = 1.f - ly, hx = 1.f - lx;
                const float gm = on ? gbuf[q * 64 + lane] * m : 0.f;
                float* wp = dw + (on ? gi >> 8 : 0) * 64 + lane;
                const float a00 = wp[0], a01 = wp[64], a10 = wp[WW * 64], a11 = wp[WW * 64 + 64];
                wp[0] = a00 + ((gi & 1) ? gm * (hy * hx) : 0.f);
                wp[64] = a01 + ((gi & 2) ? gm * (hy * lx) : 0.f);
                wp[WW * 64] = a10 + ((gi & 4) ? gm * (ly * hx) : 0.f);
                wp[WW * 64 + 64] = a11 + ((gi & 8) ? gm * (ly * lx) : 0.f);
            }
        }
        __syncthreads();                                              // chunk c consumed by all waves, chunk c + 1 landed
    }
    // ---- flush: d(x) window (one atomic per element: neighbouring tiles' windows overlap) and the three d(om) scalars per unit ----
    // (knock-out builds for scripts/col2im_knockout.sh: 1 no d(x) flush, 2 plain stores instead of atomics, 4 no chunk loop, 8 no scatter wave,
    // 16 no reduce waves, 32 no d(cols) DMA -- results wrong on purpose)
    if (!(COL2IM_KO & 1))
    for (int i = wave; i < WPX; i += 3) {
        const int y = wy0 + i / WW, xx = wx0 + i % WW;
        if ((unsigned)y >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
        const float v = dw[i * 64 + lane];
        if (COL2IM_KO & 2) {
            if (v != 0.f) dxb[(size_t)(y * W + xx) * C] = v;
        } else if (v != 0.f) unsafeAtomicAdd(dxb + (size_t)(y * W + xx) * C, v);
    }
    for (int u = tid; u < NU; u += 192) {
        if ((geo_i[u] & 48) != 48) continue;
        const int pl = u / 9, tap = u - pl * 9;
        const int h = ty * TH + pl / TW, w = tx * TW + pl % TW;
        float* domp = dom + ((long long)b * HW + h * W + w) * om_stride;
        unsafeAtomicAdd(domp + d * 18 + 2 * tap, dacc[u * 3]);        // the group's cpg / 64 slices add into the same scalars
        unsafeAtomicAdd(domp + d * 18 + 2 * tap + 1, dacc[u * 3 + 1]);
        unsafeAtomicAdd(domp + dg * 18 + d * 9 + tap, dacc[u * 3 + 2]);
    }
}

// The units dcn_col2im_kernel leaves out (a corner outside its LDS window): one wave per (tile, 64-channel slice) lists them and does
// both jobs straight from / to global memory, DC_U units (5 loads each) in flight.  No large LDS -> many resident waves, which is what
// hides the load round trips; kept out of the main kernel so those round trips never sit between its barriers.
template <int TH, int TW>
__global__ __launch_bounds__(64) void dcn_col2im_overflow_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                                 const float* __restrict__ dcols, float* __restrict__ dx,
                                                                 float* __restrict__ dom, int B, int H, int W, int C, int dg,
                                                                 int om_stride, int tiles_y, int tiles_x) {
    constexpr int NU = TH * TW * 9, DC_U = 4;
    __shared__ int o_u[NU], o_gi[NU], o_yx[NU];
    __shared__ float o_ly[NU], o_lx[NU], o_m[NU];
    __shared__ int novf;
    const int lane = threadIdx.x;
    const int HW = H * W, cpg = C / dg;
    const int slices = C / 64;
    int bid = blockIdx.x;
    const int sl = bid % slices;  bid /= slices;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ch0 = sl * 64, d = ch0 / cpg;
    const int wy0 = ty * TH - DC_R, wx0 = tx * TW - DC_R;
    if (lane == 0) novf = 0;
    __syncthreads();
    for (int u = lane; u < NU; u += 64) {
        const int pl = u / 9, tap = u - pl * 9;
        const DcGeo r = dc_geometry<TH, TW>(om, b, ty * TH + pl / TW, tx * TW + pl % TW, tap, d, dg, H, W, om_stride, wy0, wx0);
        if ((r.gi & 48) == 16) {
            const int i = atomicAdd(&novf, 1);
            o_u[i] = u;
            o_gi[i] = r.gi;
            o_yx[i] = r.gyx;
            o_ly[i] = r.ly;
            o_lx[i] = r.lx;
            o_m[i] = r.m;
        }
    }
    __syncthreads();
    const int n = novf;
    const float* xb = x + (size_t)b * HW * C + ch0 + lane;
    float* dxb = dx + (size_t)b * HW * C + ch0 + lane;
    const float* dcb = dcols + (size_t)b * HW * 9 * C + ch0 + lane;
    for (int i0 = 0; i0 < n; i0 += DC_U) {
        float g[DC_U], v00[DC_U], v01[DC_U], v10[DC_U], v11[DC_U];
        int gis[DC_U], y0s[DC_U], x0s[DC_U];
#pragma unroll
        for (int q = 0; q < DC_U; ++q) {
            const bool on = i0 + q < n;
            const int i = on ? i0 + q : 0;
            const int u = __builtin_amdgcn_readfirstlane(o_u[i]);
            const int gi = on ? __builtin_amdgcn_readfirstlane(o_gi[i]) : 0;
            const int gyx = __builtin_amdgcn_readfirstlane(o_yx[i]);
            gis[q] = gi;
            y0s[q] = (int)(short)(gyx & 0xFFFF);
            x0s[q] = gyx >> 16;
            const int pl = u / 9, tap = u - pl * 9;
            const int h = ty * TH + pl / TW, w = tx * TW + pl % TW;
            const float* gp = xb + (long long)(y0s[q] * W + x0s[q]) * C;
            g[q] = (gi & 16) ? dcb[((size_t)(h * W + w) * 9 + tap) * C] : 0.f;
            v00[q] = (gi & 1) ? gp[0] : 0.f;
            v01[q] = (gi & 2) ? gp[C] : 0.f;
            v10[q] = (gi & 4) ? gp[(long long)W * C] : 0.f;
            v11[q] = (gi & 8) ? gp[(long long)W * C + C] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < DC_U; ++q) {
            const int gi = gis[q];
            if (!(gi & 16)) continue;
            const int i = i0 + q;
            const int u = __builtin_amdgcn_readfirstlane(o_u[i]);
            const float ly = o_ly[i], lx = o_lx[i], m = o_m[i];
            const float hy = 1.f - ly, hx = 1.f - lx;
            const float s_m = g[q] * (v00[q] * (hy * hx) + v01[q] * (hy * lx) + v10[q] * (ly * hx) + v11[q] * (ly * lx));
            const float s_y = g[q] * ((v10[q] - v00[q]) * hx + (v11[q] - v01[q]) * lx);
            const float s_x = g[q] * ((v01[q] - v00[q]) * hy + (v11[q] - v10[q]) * ly);
            const float t_m = wave_sum_dpp(s_m), t_y = wave_sum_dpp(s_y), t_x = wave_sum_dpp(s_x);
            if (lane == 0) {
                const int pl = u / 9, tap = u - pl * 9;
                const int h = ty * TH + pl / TW, w = tx * TW + pl % TW;
                float* domp = dom + ((long long)b * HW + h * W + w) * om_stride;
                unsafeAtomicAdd(domp + d * 18 + 2 * tap, t_y * m);
                unsafeAtomicAdd(domp + d * 18 + 2 * tap + 1, t_x * m);
                unsafeAtomicAdd(domp + dg * 18 + d * 9 + tap, t_m * m * (1.f - m));
            }
            const float gm = g[q] * m;
            float* gp = dxb + (long long)(y0s[q] * W + x0s[q]) * C;
            if (gi & 1) unsafeAtomicAdd(gp, gm * (hy * hx));
            if (gi & 2) unsafeAtomicAdd(gp + C, gm * (hy * lx));
            if (gi & 4) unsafeAtomicAdd(gp + (long long)W * C, gm * (ly * hx));
            if (gi & 8) unsafeAtomicAdd(gp + (long long)W * C + C, gm * (ly * lx));
        }
    }
}

}  // namespace

extern "C" int gssd_dcn_im2col_f32(const float* x, const float* om, float* cols, int B, int H, int W, int C, int dg,
                                   int om_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && cols && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0);
    GSSD_CHECK_ARG(C % (4 * dg) == 0 && om_stride >= 27 * dg && (long long)B * H * W * 9 * C < (1ll << 31));
    const long long units = (long long)B * H * W * 9 * dg;
    long long blocks = (units + 255) / 256;              // a wave per 64 units
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dcn_im2col_kernel<float>, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, om, cols, B, H, W, C, dg,
                       om_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_im2col_bf16(const void* x_bf16, const float* om, void* cols_bf16, int B, int H, int W, int C, int dg,
                                    int om_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x_bf16 && om && cols_bf16 && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0);
    GSSD_CHECK_ARG(C % (4 * dg) == 0 && om_stride >= 27 * dg && ((uintptr_t)x_bf16 % 8) == 0 && ((uintptr_t)cols_bf16 % 8) == 0);
    GSSD_CHECK_ARG((long long)B * H * W * 9 * C < (1ll << 31));
    const long long units = (long long)B * H * W * 9 * dg;
    long long blocks = (units + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dcn_im2col_kernel<unsigned short>, dim3((int)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const unsigned short*>(x_bf16), om, reinterpret_cast<unsigned short*>(cols_bf16), B, H, W, C, dg,
                       om_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_col2im_f32(const float* x, const float* om, const float* dcols, float* dx, float* dom, int B, int H,
                                   int W, int C, int dg, int om_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && dcols && dx && dom && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0);
    GSSD_CHECK_ARG(C % dg == 0 && (C / dg) % 64 == 0 && om_stride >= 27 * dg && H < 32768 && W < 32768);
    GSSD_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dcols % 16) == 0);
    constexpr int TH = 4, TW = 8;
    const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
    const long long blocks = (long long)B * tiles_y * tiles_x * (C / 64);
    GSSD_CHECK_ARG(blocks < (1ll << 31));
    hipLaunchKernelGGL((dcn_col2im_kernel<TH, TW>), dim3((int)blocks), dim3(192), 0, as_stream(stream), x, om, dcols, dx, dom, B, H, W, C,
                       dg, om_stride, tiles_y, tiles_x);
    GSSD_CHECK_LAUNCH();
    hipLaunchKernelGGL((dcn_col2im_overflow_kernel<TH, TW>), dim3((int)blocks), dim3(64), 0, as_stream(stream), x, om, dcols, dx, dom, B,
                       H, W, C, dg, om_stride, tiles_y, tiles_x);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
