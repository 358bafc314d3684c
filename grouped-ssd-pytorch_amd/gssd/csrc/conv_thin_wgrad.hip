// Weight gradient of the thin grouped 3x3 convolutions (conv1_1: 4 -> 16, conv1_2: 16 -> 16 channels per phase group at
// 300 x 300) for gfx950.  The generic split-K wgrad re-fetches the input for every tap: with 16 output channels per group
// that is ~9x the L2 traffic of the forward.  Like the forward thin kernel (conv_thin.hip) this one stages an 8 x 16
// output tile's input patch (with halo) and its dY tile ONCE per tile, wave g owns phase group g, and the nine taps are
// nine shifted LDS reads of the patch.  The 16 x (9*cin_g) gradient of a group lives in MFMA accumulators for the whole
// lifetime of the persistent workgroup and is flushed with fp32 atomics once at the end.
// MFMA: D[co][ci] += sum over 4 pixels dY[px][co] * X[px + tap][ci]  (v_mfma_f32_16x16x4_f32, k = 4 pixels).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((aligned(16))) float g_zero_page_tw[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct ThinWgradParams {
    const float* in;
    const float* dy;
    float* dw;               // packed [64][9*CIN_G], zero-filled by the caller
    const float* in_scale;
    const float* in_shift;
    const float* in_pad;
    int B, H, W, tiles_y, tiles_x;
};

template <int CIN_G, bool XF>
__global__ __launch_bounds__(256, 2) void conv_thin_wgrad_kernel(const ThinWgradParams p) {
    constexpr int COUT_G = 16, CIN = 4 * CIN_G, COUT = 64;
    constexpr int QPR = CIN / 4, PPI = 64 / QPR;            // patch: quads per pixel row, pixels per DMA piece
    constexpr int TH = 8, TW = 16, PW = TW + 2, NPATCH = (TH + 2) * PW, NPIX = TH * TW;
    constexpr int NPI = (NPATCH + PPI - 1) / PPI;           // patch DMA pieces
    constexpr int PATCH_F = NPI * PPI * CIN;                // floats
    constexpr int NACC = (CIN_G == 4) ? 3 : 9;              // accumulator tiles: (4 taps x 4 ci) x 3, or one per tap
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;
    float* dyt = smem + PATCH_F;                            // [128 px][64] floats, quads XOR-swizzled by (px & 1) << 2

    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int ntiles = p.B * tiles_per_img;
    const float* zero = g_zero_page_tw;

    // ---- fragment read offsets (floats) -----------------------------------------------------------------------------
    // k-step s covers pixels 4s..4s+3 of the tile (row s/4, columns 4*(s%4) + kq); only (s%4) and the lane matter for the
    // column-dependent terms, the row term (s/4 + dy)*PW*CIN is a compile-time immediate.
    const int a_off = kq * COUT + ((((g * COUT_G + r) >> 2) ^ ((kq & 1) << 2)) << 2) + (r & 3);   // dY[px = kq][g*16 + r]
    int b_off[4][(CIN_G == 4) ? 3 : 3];   // [s % 4][dx] (CIN_G = 16)  or  [s % 4][accumulator tile j] (CIN_G = 4)
    int b_row[(CIN_G == 4) ? 3 : 1];      // CIN_G = 4: this lane's tap row offset for tile j
    bool b_ok[(CIN_G == 4) ? 3 : 1];
#pragma unroll
    for (int sm = 0; sm < 4; ++sm)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            if constexpr (CIN_G == 4) {
                const int tap = 4 * v + (r >> 2);           // lane's tap inside accumulator tile v
                const int dx = tap % 3;
                const int col = 4 * sm + kq + dx;
                b_off[sm][v] = col * CIN + ((g ^ (col & (QPR - 1))) << 2) + (r & 3);
            } else {
                const int col = 4 * sm + kq + v;            // v = dx
                b_off[sm][v] = col * CIN + ((((g * CIN_G + r) >> 2) ^ (col & (QPR - 1))) << 2) + (r & 3);
            }
        }
    if constexpr (CIN_G == 4) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int tap = 4 * v + (r >> 2);
            b_ok[v] = tap < 9;
            b_row[v] = (b_ok[v] ? tap / 3 : 0) * PW * CIN;
        }
    }
    float sc = 1.f, sh = 0.f;
    if (XF) {
        const int ci = (CIN_G == 4) ? g * 4 + (r & 3) : g * CIN_G + r;
        sc = p.in_scale[ci];
        sh = p.in_shift[ci];
    }

    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;
        // ---- stage the input patch (quads swizzled by patch column) and the dY tile ----------------------------------------
        for (int i = g; i < NPI; i += 4) {
            const int pp = i * PPI + lane / QPR;
            const int py = pp / PW, pxx = pp - py * PW;
            const int lq = (lane % QPR) ^ (pxx & (QPR - 1));
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            const bool ok = pp < NPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* src = ok ? p.in + ((b * p.H + iy) * p.W + ix) * CIN + lq * 4 : (XF ? p.in_pad + lq * 4 : zero);
            dma16(src, patch + i * PPI * CIN);
        }
        for (int i = g; i < NPIX / 4; i += 4) {                 // 4 pixels x 16 quads per piece
            const int px = i * 4 + (lane >> 4);
            const int lq = (lane & 15) ^ ((px & 1) << 2);
            const int y = y0 + (px >> 4), x = x0 + (px & 15);
            const bool ok = y < p.H && x < p.W;
            const float* src = ok ? p.dy + ((b * p.H + y) * p.W + x) * COUT + lq * 4 : zero;
            dma16(src, dyt + i * 256);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NPIX / 4; ++s) {
            const float av = dyt[a_off + 4 * s * COUT];
            const int rowi = (s >> 2) * PW * CIN;
            if constexpr (CIN_G == 4) {
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    float bv = patch[b_off[s & 3][v] + b_row[v] + rowi];
                    if (XF) bv = fmaxf(bv * sc + sh, 0.f);
                    if (!b_ok[v]) bv = 0.f;
                    acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[v], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float bv = patch[b_off[s & 3][t % 3] + rowi + (t / 3) * PW * CIN];
                    if (XF) bv = fmaxf(bv * sc + sh, 0.f);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- flush: D rows = co (kq*4 + e), columns = r -------------------------------------------------------------------------
    constexpr int K = 9 * CIN_G;
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = g * COUT_G + kq * 4 + e;
            const int k = (CIN_G == 4) ? 16 * t + r : t * CIN_G + r;
            if (k < K) unsafeAtomicAdd(p.dw + co * K + k, acc[t][e]);
        }
}

template <int CIN_G, bool XF>
int launch_thin_wgrad(const gssd_conv_desc& d, const float* dy, float* dw, hipStream_t stream) {
    constexpr int CIN = 4 * CIN_G, PPI = 64 / (CIN / 4);
    ThinWgradParams p;
    p.in = d.in;
    p.dy = dy;
    p.dw = dw;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.in_pad = d.in_pad;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    p.tiles_y = (d.H + 7) / 8;
    p.tiles_x = (d.W + 15) / 16;
    const size_t smem = ((size_t)((180 + PPI - 1) / PPI) * PPI * CIN + 128 * 64) * sizeof(float);
    auto kern = conv_thin_wgrad_kernel<CIN_G, XF>;
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                96 * 1024) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (thin wgrad)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const long long ntiles = (long long)d.B * p.tiles_y * p.tiles_x;
    int grid = 512;
    if (ntiles < grid) grid = (int)ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

// returns 1 when the descriptor is not a thin shape
int gssd_try_conv_thin_wgrad(const gssd_conv_desc& d, const float* dy, float* dw, hipStream_t stream) {
    const int cout_g = d.Cout / d.groups;
    const bool ok = d.groups == 4 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.dil == 1 && cout_g == 16 &&
                    (d.cin_g == 4 || d.cin_g == 16) && d.in_stride == 4 * d.cin_g && d.in_ch_off == 0 && !d.m_per_image &&
                    d.H * d.W >= 75 * 75 && (long long)d.B * d.H * d.W * 64 < (1ll << 31);
    if (!ok) return 1;
    if (d.cin_g == 4) return d.in_scale ? launch_thin_wgrad<4, true>(d, dy, dw, stream) : launch_thin_wgrad<4, false>(d, dy, dw, stream);
    return d.in_scale ? launch_thin_wgrad<16, true>(d, dy, dw, stream) : launch_thin_wgrad<16, false>(d, dy, dw, stream);
}
