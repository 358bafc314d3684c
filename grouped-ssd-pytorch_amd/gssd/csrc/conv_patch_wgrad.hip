// Weight gradient of the mid-trunk grouped 3x3 convolutions (conv2_1 .. conv3_3: 16..64 channels per phase group at 150^2 / 75^2)
// for gfx950.  The split-K implicit-GEMM wgrad (conv_wgrad.hip) re-fetches the input for every tap through L2 -> LDS (an im2col
// column block per 32-pixel chunk) and runs these layers at 33..58 TFLOP/s, bound by that traffic.  Like conv_thin_wgrad.hip this
// kernel stages an 8 x 16 output tile's input patch (with halo) and its dY tile ONCE, the nine taps are nine shifted LDS reads of
// the patch, and the whole gradient block of a wave lives in MFMA accumulators for the lifetime of the persistent workgroup
// (flushed with fp32 atomics once).  Differences: a workgroup owns ONE phase group (grid.y), so the patch holds cin_g channels per
// pixel and fits LDS up to 64 channels; the four waves split the output channels (16 per wave) and, when there are fewer than four
// 16-channel blocks, the (tap, 16-input-channel) list.
// MFMA: D[co][ci] += sum over 4 pixels dY[px][co] * X[px + tap][ci]   (v_mfma_f32_16x16x4_f32, k = 4 pixels).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((aligned(16))) float g_zero_page_pw[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct PatchWgradParams {
    const float* in;
    const float* dy;
    float* dw;               // packed [Cout][9*cin_g], zero-filled by the caller
    const float* in_scale;
    const float* in_shift;
    const float* in_pad;
    int B, H, W, in_stride, in_ch_off, Cout, tiles_y, tiles_x;
};

template <int CIN_G, int COUT_G, bool XF>
__global__ __launch_bounds__(256, 2) void conv_patch_wgrad_kernel(const PatchWgradParams p) {
    constexpr int TH = 8, TW = 16, PW = TW + 2, NPATCH = (TH + 2) * PW, NPIX = TH * TW;
    constexpr int QPR = CIN_G / 4, PPI = 64 / QPR;          // patch: quads per pixel, pixels per DMA piece
    constexpr int NPI = (NPATCH + PPI - 1) / PPI;
    constexpr int PATCH_F = NPI * PPI * CIN_G;
    constexpr int QDY = COUT_G / 4, PDY = 64 / QDY;         // dY tile: quads per pixel, pixels per DMA piece
    constexpr int NDY = NPIX / PDY;
    constexpr int NCB = COUT_G / 16;                        // 16-channel output blocks
    constexpr int NSUB = 4 / NCB;                           // waves sharing an output block split the (tap, ci tile) list
    constexpr int NCI = CIN_G / 16;
    constexpr int NENT = 9 * NCI;                           // (tap, ci tile) entries
    constexpr int EPW = (NENT + NSUB - 1) / NSUB;           // entries per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;
    float* dyt = smem + PATCH_F;                            // [128 px][COUT_G], quads XOR-swizzled by (px & (QDY-1))

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y;
    const int cb = wv % NCB, sub = wv / NCB;
    const int e0 = sub * EPW;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int ntiles = p.B * tiles_per_img;
    const float* zero = g_zero_page_pw;
    const float* in_g = p.in + p.in_ch_off + g * CIN_G;
    const float* dy_g = p.dy + g * COUT_G;

    // A operand (dY): lane (r = co, kq = pixel of the k-step) reads dyt[px][cb*16 + r]
    const int a_quad = (cb * 16 + r) >> 2, a_e = r & 3;
    // B operand (input): lane (r = ci, kq = pixel) reads patch[(row, col)][ct*16 + r] for entry (tap, ct)
    float sc[NCI], sh[NCI];
#pragma unroll
    for (int ct = 0; ct < NCI; ++ct) {
        sc[ct] = 1.f;
        sh[ct] = 0.f;
        if (XF) {
            sc[ct] = p.in_scale[p.in_ch_off + g * CIN_G + ct * 16 + r];
            sh[ct] = p.in_shift[p.in_ch_off + g * CIN_G + ct * 16 + r];
        }
    }

    f32x4 acc[EPW];
#pragma unroll
    for (int t = 0; t < EPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;
        // ---- stage the group's input patch (quads swizzled by patch column) and dY tile (quads swizzled by pixel) --------------
        for (int i = wv; i < NPI; i += 4) {
            const int pp = i * PPI + lane / QPR;
            const int py = pp / PW, pxx = pp - py * PW;
            const int lq = (lane % QPR) ^ (pxx & (QPR - 1));
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            const bool ok = pp < NPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* src = ok ? in_g + ((size_t)(b * p.H + iy) * p.W + ix) * p.in_stride + lq * 4
                                  : (XF ? p.in_pad + p.in_ch_off + g * CIN_G + lq * 4 : zero);
            dma16(src, patch + i * PPI * CIN_G);
        }
        for (int i = wv; i < NDY; i += 4) {
            const int px = i * PDY + lane / QDY;
            const int lq = (lane % QDY) ^ (px & (QDY - 1));
            const int y = y0 + (px >> 4), x = x0 + (px & 15);
            const bool ok = y < p.H && x < p.W;
            const float* src = ok ? dy_g + ((size_t)(b * p.H + y) * p.W + x) * p.Cout + lq * 4 : zero;
            dma16(src, dyt + i * PDY * COUT_G);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < NPIX / 4; ++s) {
            const int px = 4 * s + kq;
            const float av = dyt[px * COUT_G + ((a_quad ^ (px & (QDY - 1))) << 2) + a_e];
            const int prow = (s >> 2), pcol = 4 * (s & 3) + kq;              // output pixel (row, col) inside the tile
#pragma unroll
            for (int t = 0; t < EPW; ++t) {
                const int ent = e0 + t;
                if (ent < NENT) {
                    const int tap = ent / NCI, ct = ent - tap * NCI;
                    const int col = pcol + tap % 3, row = prow + tap / 3;
                    float bv = patch[(row * PW + col) * CIN_G + ((((ct * 16 + r) >> 2) ^ (col & (QPR - 1))) << 2) + (r & 3)];
                    if (XF) bv = fmaxf(bv * sc[ct] + sh[ct], 0.f);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- flush: D rows = co (kq*4 + e), columns = ci r --------------------------------------------------------------------------
    constexpr int K = 9 * CIN_G;
#pragma unroll
    for (int t = 0; t < EPW; ++t) {
        const int ent = e0 + t;
        if (ent >= NENT) continue;
        const int tap = ent / NCI, ct = ent - tap * NCI;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = g * COUT_G + cb * 16 + kq * 4 + e;
            unsafeAtomicAdd(p.dw + (size_t)co * K + tap * CIN_G + ct * 16 + r, acc[t][e]);
        }
    }
}

template <int CIN_G, int COUT_G, bool XF>
int launch_patch_wgrad(const gssd_conv_desc& d, const float* dy, float* dw, hipStream_t stream) {
    constexpr int QPR = CIN_G / 4, PPI = 64 / QPR;
    PatchWgradParams p;
    p.in = d.in;
    p.dy = dy;
    p.dw = dw;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.in_pad = d.in_pad;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    p.in_stride = d.in_stride;
    p.in_ch_off = d.in_ch_off;
    p.Cout = d.Cout;
    p.tiles_y = (d.H + 7) / 8;
    p.tiles_x = (d.W + 15) / 16;
    const size_t smem = ((size_t)((180 + PPI - 1) / PPI) * PPI * CIN_G + 128 * COUT_G) * sizeof(float);
    auto kern = conv_patch_wgrad_kernel<CIN_G, COUT_G, XF>;
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) !=
            hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (patch wgrad)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const long long ntiles = (long long)d.B * p.tiles_y * p.tiles_x;
    static int per_cu = 0;                                // resident workgroups per CU: 2 for the 64-channel variant, up to 4
    if (!per_cu) {                                        // for the small ones (they hide each other's staging latency)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, smem) != hipSuccess || per_cu < 1) per_cu = 2;
        if (per_cu > 4) per_cu = 4;
    }
    int gx = 256 * per_cu / d.groups;
    if (gx < 1) gx = 1;
    if (ntiles < gx) gx = (int)ntiles;
    hipLaunchKernelGGL(kern, dim3(gx, d.groups), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

// returns 1 when the descriptor is not one of the patch-staged shapes
int gssd_try_conv_patch_wgrad(const gssd_conv_desc& d, const float* dy, float* dw, hipStream_t stream) {
    const int cout_g = d.Cout / d.groups;
    const bool ok = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.dil == 1 && !d.m_per_image && d.H * d.W >= 38 * 38 &&
                    d.in_stride % 4 == 0 && d.in_ch_off % 4 == 0 && (long long)d.B * d.H * d.W * d.in_stride < (1ll << 31) &&
                    (long long)d.B * d.H * d.W * d.Cout < (1ll << 31);
    if (!ok) return 1;
#define GSSD_PW(CI, CO)                                                                                     \
    if (d.cin_g == CI && cout_g == CO)                                                                       \
        return d.in_scale ? launch_patch_wgrad<CI, CO, true>(d, dy, dw, stream) : launch_patch_wgrad<CI, CO, false>(d, dy, dw, stream);
    GSSD_PW(16, 32)
    GSSD_PW(32, 32)
    GSSD_PW(32, 64)
    GSSD_PW(64, 64)
#undef GSSD_PW
    return 1;
}
