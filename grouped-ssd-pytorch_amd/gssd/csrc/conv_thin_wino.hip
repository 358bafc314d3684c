// Patch-staged Winograd F(2x2,3x3) for conv1_2 of the GSSD trunk (4 phase groups x 16 -> 16 channels at 300 x 300, the largest
// single layer: 53 GFLOP at B = 32) and its data gradient, for gfx950.
//
// conv_thin.hip runs this layer as nine shifted MFMA taps out of an LDS patch and is bound by the fp32 MFMA rate (82 TFLOP/s);
// the generic Winograd kernel (conv_wino.hip) fetches every input pixel ~4x through L1/L2 and is no faster here.  This kernel
// combines the two: the (8+2) x (16+2) pixel patch of an 8 x 16 output tile is staged ONCE by LDS-DMA (all 64 channels of a
// pixel as one 256-byte burst, quads XOR-swizzled by the patch column exactly like conv_thin.hip), double buffered across the
// tiles of a persistent workgroup; wave g owns phase group g and turns the patch into 32 Winograd tiles = 2 MFMA row blocks:
//     16 ds_read_b128 (one 4-channel quad per patch position)  ->  per channel: B^T d B (32 adds)  ->  16 MFMAs (one per xi)
// with the group's U = G g G^T held in 64 registers as B fragments for the lifetime of the workgroup.  2.25x fewer MFMA flops
// than the direct form; the layer becomes HBM-bound (1.47 GB in + out per launch).  The epilogue is conv_thin.hip's: output
// transform in registers, transpose through LDS, whole-row NHWC stores, fp64 batch sums per workgroup.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((aligned(16))) float g_zero_page_tw2[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct ThinWinoParams {
    const float* in;
    const float* U;          // [4 groups][16 xi][16 co][16 ci]
    const float* bias;
    float* out;
    double* stats;
    int stats_rep;
    const float* in_scale;
    const float* in_shift;
    const float* in_pad;
    const float* pool_sign;  // GSSD_CONV_POOL2: `out` is the 2x2 / stride-2 pooled raw map, max where pool_sign[c] >= 0 else min
    int B, H, W, tiles_y, tiles_x;
};

constexpr int TW_CIN = 64, TW_COUT = 64, TW_TH = 8, TW_TW = 16, TW_PW = TW_TW + 2, TW_PH = TW_TH + 2;
constexpr int TW_NPATCH = TW_PH * TW_PW;                        // 180 patch pixels
constexpr int TW_NINSTR = (TW_NPATCH + 3) / 4;                  // DMA pieces of 4 pixels (1 KB)
constexpr int TW_PATCH_F = TW_NINSTR * 4 * TW_CIN;              // floats per patch buffer
constexpr int TW_OLD = TW_COUT + 4;                             // padded out-staging row (floats)
constexpr int TW_STAGE_F = 128 * TW_OLD;

template <bool XF>
__global__ __launch_bounds__(256, 2) void conv_thin_wino_kernel(const ThinWinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                                        // [TW_PATCH_F]
    float* stage = smem + TW_PATCH_F;                           // [128][TW_OLD]
    double* lacc = reinterpret_cast<double*>(stage + TW_STAGE_F);      // [2][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave = phase group
    const int r = lane & 15, kq = lane >> 4;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int ntiles = p.B * tiles_per_img;

    f32x4 U[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) U[xi] = *reinterpret_cast<const f32x4*>(p.U + (((size_t)g * 16 + xi) * 16 + r) * 16 + kq * 4);
    f32x4 isc = f32x4{1.f, 1.f, 1.f, 1.f}, ish = f32x4{0.f, 0.f, 0.f, 0.f};
    if (XF) {
        isc = *reinterpret_cast<const f32x4*>(p.in_scale + g * 16 + kq * 4);
        ish = *reinterpret_cast<const f32x4*>(p.in_shift + g * 16 + kq * 4);
    }
    const float bias = p.bias ? p.bias[g * 16 + r] : 0.f;
    // (kept in registers for the kernel's lifetime: a VGPR-returning load inside the tile loop costs a vmcnt(0) = a wait for the previous
    // tile's stores, conv_thin_bf16.hip)
    const f32x4 sg = p.pool_sign ? *reinterpret_cast<const f32x4*>(p.pool_sign + (tid & 15) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* zero = g_zero_page_tw2;
    for (int c = tid; c < 2 * TW_COUT; c += 256) lacc[c] = 0.0;

    // A-operand read offsets: lane (r, kq) of row block i owns Winograd tile (ty = 2i + (r >> 3), tx = r & 7), whose 4 x 4
    // input patch starts at patch pixel (2 ty, 2 tx); quads are swizzled by the patch column.
    const int lqd = g * 4 + kq;
    const int tx2 = 2 * (r & 7), tyr = r >> 3;
    int coff[4];                                               // column term for b = 0..3
#pragma unroll
    for (int b = 0; b < 4; ++b) coff[b] = (tx2 + b) * TW_CIN + ((lqd ^ ((tx2 + b) & 15)) << 2);

    auto stage_patch = [&](int tile, int buf) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int y0 = tyi * TW_TH, x0 = txi * TW_TW;
        for (int i = g; i < TW_NINSTR; i += 4) {
            const int pp = i * 4 + (lane >> 4);
            const int py = pp / TW_PW, pxx = pp - py * TW_PW;
            const int lq = (lane & 15) ^ (pxx & 15);
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            const bool ok = pp < TW_NPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* src = ok ? p.in + ((size_t)(b * p.H + iy) * p.W + ix) * TW_CIN + lq * 4 : (XF ? p.in_pad + lq * 4 : zero);
            dma16(src, patch + i * 4 * TW_CIN);
        }
    };

    // batch sums: fp32 within a tile, fp64 per thread across the tiles of the workgroup (a thread always owns the same four
    // channels), one LDS / global flush at the very end
    double ds[4] = {0.0, 0.0, 0.0, 0.0}, dq[4] = {0.0, 0.0, 0.0, 0.0};
    // workgroup ids go round-robin over the 8 XCDs: remap so that each XCD works on gridDim.x / 8 CONSECUTIVE tiles of every
    // sweep (the halo columns / rows neighbouring tiles share then come from that XCD's L2)
    const int bperm = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    for (int tile = bperm; tile < ntiles; tile += gridDim.x) {
        stage_patch(tile, 0);                                  // the other resident workgroup computes meanwhile
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        const float* pb = patch;

#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 raw[16];
            const int rbase = (2 * (2 * i + tyr)) * TW_PW * TW_CIN;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    raw[a * 4 + b] = *reinterpret_cast<const f32x4*>(pb + rbase + a * (TW_PW * TW_CIN) + coff[b]);
            f32x4 acc[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) acc[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d[16], t[16], V[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) d[q] = XF ? fmaxf(raw[q][e] * isc[e] + ish[e], 0.f) : raw[q][e];
#pragma unroll
                for (int j = 0; j < 4; ++j) {                  // B^T d
                    t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
                    t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
                    t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
                    t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {                  // (B^T d) B
                    V[a * 4 + 0] = t[a * 4 + 0] - t[a * 4 + 2];
                    V[a * 4 + 1] = t[a * 4 + 1] + t[a * 4 + 2];
                    V[a * 4 + 2] = t[a * 4 + 2] - t[a * 4 + 1];
                    V[a * 4 + 3] = t[a * 4 + 1] - t[a * 4 + 3];
                }
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[xi], U[xi][e], acc[xi], 0, 0, 0);
            }
            // output transform: lane holds M[tile 4 kq + e][co r] for all 16 xi
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[0][j] = acc[0 * 4 + j][e] + acc[1 * 4 + j][e] + acc[2 * 4 + j][e];
                    s[1][j] = acc[1 * 4 + j][e] - acc[2 * 4 + j][e] - acc[3 * 4 + j][e];
                }
                // tile t = 4 kq + e of row block i sits at (2 i + (t >> 3), t & 7): transpose through LDS
                const int tt = kq * 4 + e;
                const int py = 2 * (2 * i + (tt >> 3)), px = 2 * (tt & 7);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float* sp = stage + ((py + a) * TW_TW + px) * TW_OLD + g * 16 + r;
                    sp[0] = s[a][0] + s[a][1] + s[a][2] + bias;
                    sp[TW_OLD] = s[a][1] - s[a][2] - s[a][3] + bias;
                }
            }
        }
        __syncthreads();
        // ---- whole-row NHWC stores + batch sums (a thread always owns the same 4 channels: 256 % 16 == 0) ----------------------
        {
            const int b = tile / tiles_per_img;
            const int trem = tile - b * tiles_per_img;
            const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
            const int y0 = tyi * TW_TH, x0 = txi * TW_TW;
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
            const int c4 = tid & 15;
            if (p.pool_sign) {
                // GSSD_CONV_POOL2: the staged 8 x 16 tile is 4 x 8 pooling windows (tile origins are even): a thread reduces the
                // four pixels of a window for its 4 channels -- batch sums over ALL pixels as before -- and stores the maximum
                // where the channel's BatchNorm weight is >= 0, the minimum where it is negative.  The full map is never written.
                const int Hp = (p.H + 1) >> 1, Wp = (p.W + 1) >> 1;
                // batch sums: the same pixels in the same order as the unpooled epilogue (identical statistics), no store
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int px = (tid + 256 * it) >> 4;
                    const int y = y0 + (px >> 4), x = x0 + (px & 15);
                    if (y < p.H && x < p.W) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * TW_OLD + c4 * 4);
                        s4 += v;
                        q4 += v * v;
                    }
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int w = (tid + 256 * it) >> 4;              // window 0 .. 31 of the tile: (w >> 3, w & 7)
                    const int wy = w >> 3, wx = w & 7;
                    const int y = y0 + 2 * wy, x = x0 + 2 * wx;
                    if (y >= p.H || x >= p.W) continue;
                    f32x4 mx, mn;
                    bool first = true;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                            if (y + a >= p.H || x + bb >= p.W) continue;
                            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + ((2 * wy + a) * TW_TW + 2 * wx + bb) * TW_OLD + c4 * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                mx[e] = first ? v[e] : fmaxf(mx[e], v[e]);
                                mn[e] = first ? v[e] : fminf(mn[e], v[e]);
                            }
                            first = false;
                        }
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = sg[e] >= 0.f ? mx[e] : mn[e];
                    *reinterpret_cast<f32x4*>(p.out + ((size_t)(b * Hp + (y >> 1)) * Wp + (x >> 1)) * TW_COUT + c4 * 4) = o;
                }
            } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int px = (tid + 256 * it) >> 4;
                const int y = y0 + (px >> 4), x = x0 + (px & 15);
                if (y < p.H && x < p.W) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * TW_OLD + c4 * 4);
                    *reinterpret_cast<f32x4*>(p.out + ((size_t)(b * p.H + y) * p.W + x) * TW_COUT + c4 * 4) = v;
                    s4 += v;
                    q4 += v * v;
                }
            }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ds[e] += (double)s4[e];
                dq[e] += (double)q4[e];
            }
        }
        __syncthreads();                                       // stage and patch are free again
    }
    if (p.stats) {
        const int c4 = tid & 15;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsafeAtomicAdd(lacc + c4 * 4 + e, ds[e]);
            unsafeAtomicAdd(lacc + TW_COUT + c4 * 4 + e, dq[e]);
        }
        __syncthreads();
        if (tid < TW_COUT) {
            double* st = gssd_stats_replica(p.stats, p.stats_rep, TW_COUT);
            unsafeAtomicAdd(st + tid, lacc[tid]);
            unsafeAtomicAdd(st + TW_COUT + tid, lacc[TW_COUT + tid]);
        }
    }
}

template <bool XF>
int launch_thin_wino(const gssd_conv_desc& d, hipStream_t stream) {
    ThinWinoParams p;
    p.in = d.in;
    p.U = d.wgt_wino;
    p.bias = d.bias;
    p.out = d.out;
    p.stats = d.stats;
    p.stats_rep = d.stats_rep;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.in_pad = d.in_pad;
    p.pool_sign = (d.flags & GSSD_CONV_POOL2) ? d.pool_sign : nullptr;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    p.tiles_y = (d.H + TW_TH - 1) / TW_TH;
    p.tiles_x = (d.W + TW_TW - 1) / TW_TW;
    constexpr size_t smem = ((size_t)TW_PATCH_F + TW_STAGE_F) * sizeof(float) + 2 * TW_COUT * sizeof(double);
    auto kern = conv_thin_wino_kernel<XF>;
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
            hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (thin winograd)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const long long ntiles = (long long)d.B * p.tiles_y * p.tiles_x;
    int grid = 512;                                       // two resident workgroups per CU (80 KB of LDS, <= 256 VGPRs each)
    if (ntiles < grid) grid = (int)ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

// returns 1 when the descriptor is not the conv1_2 shape class (4 groups x 16 -> 16 channels, large map) with Winograd weights
int gssd_try_conv_thin_wino(const gssd_conv_desc& d, hipStream_t stream) {
    if (!d.wgt_wino) return 1;
    const bool ok = d.groups == 4 && d.Cout == 64 && d.cin_g == 16 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 &&
                    d.dil == 1 && d.in_stride == 64 && d.in_ch_off == 0 && d.out_mode == GSSD_OUT_NHWC && d.out_stride == 64 &&
                    d.out_ch_off == 0 && !d.m_per_image && !d.relu && !d.gate && !d.resid && !d.alpha && d.split_k <= 1 &&
                    d.H * d.W >= 75 * 75 && ((uintptr_t)d.out % 16) == 0 && ((uintptr_t)d.in % 16) == 0 &&
                    (long long)d.B * d.H * d.W * 64 < (1ll << 31);
    if (!ok) return 1;
    return d.in_scale ? launch_thin_wino<true>(d, stream) : launch_thin_wino<false>(d, stream);
}
