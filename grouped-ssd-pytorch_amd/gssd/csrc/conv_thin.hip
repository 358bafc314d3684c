// Thin grouped 3x3 convolutions of the GSSD trunk (conv1_1, conv1_2, conv2_1: 4 phase groups with only
// 4..16 input and 16..32 output channels per group at 300^2 / 150^2) for gfx950.
//
// With so few channels per group an implicit GEMM that re-fetches the input for every tap needs 2/N bytes
// of L2 traffic per FLOP (N = 16): ~20 TB/s at the fp32 MFMA peak.  This kernel instead stages each spatial
// tile's input ONCE:
//   * a 256-thread workgroup owns a TH x TW (<= 128 pixel) output tile of one image and ALL channels;
//     wave g computes conv group g (= CT phase g), so a pixel's full channel vector is fetched as one
//     contiguous burst by LDS-DMA (global_load_lds_dwordx4) together with its 1-pixel halo;
//   * the nine taps are nine shifted ds_reads of that patch -- the input crosses L2 -> LDS ~1.5x, not 9x;
//   * the group's whole weight matrix (<= 32 x 144) lives in registers as MFMA B fragments for the
//     lifetime of the (persistent) workgroup;
//   * the 16-byte quads of a pixel row are XOR-swizzled with the patch row index on the DMA source side, so
//     the 16 lanes of a fragment read hit 16 different LDS slots;
//   * the epilogue transposes the accumulators through LDS and writes whole NHWC pixel rows (256..512 B
//     contiguous per pixel, 4..8 KB per tile row); BatchNorm batch sums are kept in registers across
//     tiles and flushed with one fp64 atomic per channel per workgroup.
// fp32 MFMA (v_mfma_f32_16x16x4_f32) = exact fp32 FMA, so results match the generic kernel / fp32 reference
// to accumulation-order rounding.  Reference call sites: models/ssd_multiphase_custom_group.py:444.
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((aligned(16))) float g_zero_page_thin[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct ThinParams {
    const float* in;
    const float* wgt;    // packed [Cout][9*CIN_G]
    const float* bias;
    float* out;
    double* stats;
    int stats_rep;
    const float* in_scale;   // fused producer BN+ReLU (NULL = none)
    const float* in_shift;
    const float* in_pad;
    int B, H, W;
    int TH, TW, tiles_y, tiles_x;
    unsigned inv_pw, inv_tw;   // ceil(65536 / (TW+2)), ceil(65536 / TW)
    int acc_off;               // float offset of the fp64 batch-sum accumulators inside dynamic LDS
};

// FIXED: the 8 x 16 tile is a compile-time constant (all pixel <-> tile arithmetic folds; fragment offsets need 3 registers)
template <int CIN_G, int COUT_G, bool FIXED, bool XF>
__global__ __launch_bounds__(256, (COUT_G > 16 ? 2 : 3)) void conv_thin_kernel(const ThinParams p) {
    constexpr int CIN = 4 * CIN_G, COUT = 4 * COUT_G;
    constexpr int QPR = CIN / 4;                  // 16-byte quads per pixel row
    constexpr int PPI = 64 / QPR;                 // pixels per DMA wave instruction
    constexpr int NT = COUT_G / 16;
    constexpr int MTILES = 8;                     // 128 pixel slots
    constexpr int KS = (CIN_G == 4) ? 1 : CIN_G / 16;   // 16-k steps per tap (CIN_G = 4: one 4-k step)
    constexpr int OLD = COUT + 4;                 // padded out-staging row (floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = conv group
    const int r = lane & 15, kq = lane >> 4;
    const int TH = FIXED ? 8 : p.TH, TW = FIXED ? 16 : p.TW, PW = TW + 2, PH = TH + 2;
    const int npix = TH * TW, npatch = PH * PW;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int ntiles = p.B * tiles_per_img;

    // ---- weights -> registers (B fragments): lane (n = r, kq) holds W[nt*16 + n][tap][kq*4 .. +3] -------------
    f32x4 wf[9][KS][NT];
    float wf1[9][NT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float* wr = p.wgt + (size_t)(g * COUT_G + j * 16 + r) * (9 * CIN_G) + t * CIN_G;
            if constexpr (CIN_G == 4) {
                wf1[t][j] = wr[kq];
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) wf[t][ks][j] = *reinterpret_cast<const f32x4*>(wr + ks * 16 + kq * 4);
            }
        }
    // fused producer BatchNorm + ReLU: this lane always reads the same input channels (quad g*CIN_G/4 + ks*4 + kq)
    constexpr bool xf = XF;                    // fused producer BatchNorm + ReLU (compile-time: no selects in the loop)
    f32x4 isc[KS], ish[KS];
    float isc1 = 1.f, ish1 = 0.f;
    if (xf) {
        if constexpr (CIN_G == 4) {
            isc1 = p.in_scale[g * 4 + kq];
            ish1 = p.in_shift[g * 4 + kq];
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                isc[ks] = *reinterpret_cast<const f32x4*>(p.in_scale + g * CIN_G + ks * 16 + kq * 4);
                ish[ks] = *reinterpret_cast<const f32x4*>(p.in_shift + g * CIN_G + ks * 16 + kq * 4);
            }
        }
    }
    const float* zero = g_zero_page_thin;
    float bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = p.bias ? p.bias[g * COUT_G + j * 16 + r] : 0.f;

    // Fragment-read offsets (floats): the 16-byte quads of a patch pixel are XOR-swizzled with its patch COLUMN, so a
    // tap only adds a row term (dy*PW*CIN) and selects one of three precomputed column terms (dx = 0, 1, 2).
    const int lqd0 = (CIN_G == 4) ? g : g * (CIN_G / 4) + kq;          // logical quad this lane reads (ks = 0)
    constexpr int FT = FIXED ? 1 : MTILES;        // FIXED: m-tile i is tile row i, so foff[i] = foff[0] + i*PW*CIN
    int foff[FT][3];
#pragma unroll
    for (int i = 0; i < FT; ++i) {
        int px = i * 16 + r;
        if (px >= npix) px = 0;                     // dummy lanes read a valid pixel; results are discarded
        const int ty = px / TW, tx = px - ty * TW;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int col = tx + dx;
            foff[i][dx] = (ty * PW + col) * CIN + ((lqd0 ^ (col & (QPR - 1))) << 2) + ((CIN_G == 4) ? kq : 0);
        }
    }
    constexpr int C4 = COUT / 4;
    constexpr int NST = 128 * C4 / 256;
    const int ninstr = (npatch + PPI - 1) / PPI;
    // x / PW and x / TW for x < 4096 by multiply-shift (host-computed reciprocals, exact in that range)
    const unsigned inv_pw = FIXED ? (65536u + 17u) / 18u : p.inv_pw, inv_tw = FIXED ? 4096u : p.inv_tw;

    // batch sums: fp32 within a tile, then fp64 accumulators in LDS (beyond the patch / staging area) for the whole
    // lifetime of the workgroup; flushed with one fp64 global atomic per channel at the end
    double* lacc = reinterpret_cast<double*>(smem + p.acc_off);     // [2][COUT]
    for (int c = tid; c < 2 * COUT; c += 256) lacc[c] = 0.0;

    // workgroup ids go round-robin over the 8 XCDs: remap so that each XCD works on gridDim.x / 8 CONSECUTIVE tiles of every
    // sweep (the halo columns / rows neighbouring tiles share then come from that XCD's L2)
    const int bperm = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    for (int tile = bperm; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;

        // ---- stage the (TH+2) x (TW+2) patch: pixel rows of CIN floats, quads XOR-swizzled by the patch row -----
        for (int i = g; i < ninstr; i += 4) {
            const int pp = i * PPI + lane / QPR;
            const int py = (int)(((unsigned)pp * inv_pw) >> 16), pxx = pp - py * PW;
            const int lq = (lane % QPR) ^ (pxx & (QPR - 1));
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            const bool ok = pp < npatch && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* src = ok ? p.in + ((b * p.H + iy) * p.W + ix) * CIN + lq * 4 : (xf ? p.in_pad + lq * 4 : zero);
            dma16(src, smem + i * PPI * CIN);
        }
        __syncthreads();

        // ---- 9 taps x KS k-steps of MFMAs straight from the patch -------------------------------------------------
        f32x4 acc[MTILES][NT];
#pragma unroll
        for (int i = 0; i < MTILES; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // software-pipelined by one (m-tile pair, tap) step; sched_barrier keeps the compiler from hoisting all 72
        // fragment reads (256 VGPRs) to the top
        auto FO = [&](int i, int dx) -> int { return FIXED ? foff[0][dx] + i * (18 * CIN) : foff[FIXED ? 0 : i][dx]; };
        auto ld4 = [&](int off, int ks) -> f32x4 {
            static_assert(KS == 1, "ks > 0 needs its own column terms");
            return *reinterpret_cast<const f32x4*>(smem + off);
        };
        auto tf4 = [&](f32x4 v, int ks) -> f32x4 {      // producer BN + ReLU on a fragment
            if (xf) {
                v = v * isc[ks] + ish[ks];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            return v;
        };
        auto tf1 = [&](float v) -> float { return xf ? fmaxf(v * isc1 + ish1, 0.f) : v; };
        auto ld1 = [&](int off) -> float {
            return smem[off];
        };
        f32x4 a0[KS], a1[KS], n0[KS], n1[KS];
        float s0 = 0.f, s1 = 0.f, m0 = 0.f, m1 = 0.f;
        if constexpr (CIN_G == 4) {
            s0 = tf1(ld1(FO(0, 0)));
            s1 = tf1(ld1(FO(1, 0)));
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                a0[ks] = tf4(ld4(FO(0, 0), ks), ks);
                a1[ks] = tf4(ld4(FO(1, 0), ks), ks);
            }
        }
#pragma unroll
        for (int ip = 0; ip < MTILES; ip += 2) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int tn = (t + 1) % 9, ipn = (t == 8) ? ip + 2 : ip;
                if (ipn < MTILES) {
                    const int roff = (tn / 3) * PW * CIN;
                    if constexpr (CIN_G == 4) {
                        m0 = ld1(FO(ipn, tn % 3) + roff);
                        m1 = ld1(FO(ipn + 1, tn % 3) + roff);
                    } else {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            n0[ks] = ld4(FO(ipn, tn % 3) + roff, ks);
                            n1[ks] = ld4(FO(ipn + 1, tn % 3) + roff, ks);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);     // next step's fragment reads stay ahead of this step's MFMAs
                if constexpr (CIN_G == 4) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[ip][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(s0, wf1[t][j], acc[ip][j], 0, 0, 0);
                        acc[ip + 1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(s1, wf1[t][j], acc[ip + 1][j], 0, 0, 0);
                    }
                    s0 = tf1(m0);        // the load was issued before this step's MFMAs: its latency is covered
                    s1 = tf1(m1);
                } else {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                acc[ip][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks][s], wf[t][ks][j][s], acc[ip][j], 0, 0, 0);
                                acc[ip + 1][j] =
                                    __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks][s], wf[t][ks][j][s], acc[ip + 1][j], 0, 0, 0);
                            }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        a0[ks] = tf4(n0[ks], ks);    // issued before this step's MFMAs: latency covered
                        a1[ks] = tf4(n1[ks], ks);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();          // every wave is done reading the patch: reuse LDS for the output tile

        // ---- epilogue: + bias, transpose through LDS, whole-row NHWC stores + batch sums -------------------------------
#pragma unroll
        for (int i = 0; i < MTILES; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    smem[(i * 16 + kq * 4 + e) * OLD + g * COUT_G + j * 16 + r] = acc[i][j][e] + bias[j];
        __syncthreads();
        // 256 % C4 == 0: a thread always owns the same 4 channels
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
        const int c4 = tid % C4;
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int px = (tid + 256 * it) / C4;
            const int ty = (int)(((unsigned)px * inv_tw) >> 16), tx = px - ty * TW;
            const int y = y0 + ty, x = x0 + tx;
            if (px < npix && y < p.H && x < p.W) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(smem + px * OLD + c4 * 4);
                *reinterpret_cast<f32x4*>(p.out + ((size_t)(b * p.H + y) * p.W + x) * COUT + c4 * 4) = v;
                s4 += v;
                q4 += v * v;
            }
        }
        if (p.stats) {
            // lanes l, l+C4, l+2*C4, ... of a wave own the same channel quad
#pragma unroll
            for (int o = 32; o >= C4; o >>= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s4[e] += __shfl_xor(s4[e], o, 64);
                    q4[e] += __shfl_xor(q4[e], o, 64);
                }
            if (lane < C4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsafeAtomicAdd(lacc + lane * 4 + e, (double)s4[e]);
                    unsafeAtomicAdd(lacc + COUT + lane * 4 + e, (double)q4[e]);
                }
            }
        }
        __syncthreads();
    }

    if (p.stats && tid < COUT) {
        double* st = gssd_stats_replica(p.stats, p.stats_rep, COUT);
        unsafeAtomicAdd(st + tid, lacc[tid]);
        unsafeAtomicAdd(st + COUT + tid, lacc[COUT + tid]);
    }
}

template <int CIN_G, int COUT_G, bool FIXED, bool XF>
int launch_thin_impl(const gssd_conv_desc& d, hipStream_t stream) {
    constexpr int CIN = 4 * CIN_G, COUT = 4 * COUT_G;
    ThinParams p;
    p.in = d.in;
    p.wgt = d.wgt;
    p.bias = d.bias;
    p.out = d.out;
    p.stats = d.stats;
    p.stats_rep = d.stats_rep;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.in_pad = d.in_pad;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    // tile shape: 5 x 25 divides 150 and 75 exactly; 8 x 16 otherwise
    if (!FIXED) {
        p.TH = 5;
        p.TW = 25;
    } else {
        p.TH = 8;
        p.TW = 16;
    }
    p.inv_pw = (65536u + (unsigned)(p.TW + 2) - 1) / (unsigned)(p.TW + 2);
    p.inv_tw = (65536u + (unsigned)p.TW - 1) / (unsigned)p.TW;
    p.tiles_y = (d.H + p.TH - 1) / p.TH;
    p.tiles_x = (d.W + p.TW - 1) / p.TW;
    const int npatch = (p.TH + 2) * (p.TW + 2);
    constexpr int PPI = 64 / (CIN / 4);
    const size_t patch_bytes = (size_t)((npatch + PPI - 1) / PPI) * PPI * CIN * sizeof(float);
    const size_t out_bytes = (size_t)128 * (COUT + 4) * sizeof(float);
    const size_t work = patch_bytes > out_bytes ? patch_bytes : out_bytes;
    p.acc_off = (int)(work / sizeof(float));
    const size_t smem = work + 2 * COUT * sizeof(double);
    auto kern = conv_thin_kernel<CIN_G, COUT_G, FIXED, XF>;
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                96 * 1024) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (thin conv)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const long long ntiles = (long long)d.B * p.tiles_y * p.tiles_x;
    const int per_cu = smem > 60 * 1024 ? 2 : 3;
    int grid = 256 * per_cu;
    if (ntiles < grid) grid = (int)ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

template <int CIN_G, int COUT_G>
int launch_thin(const gssd_conv_desc& d, hipStream_t stream) {
    // compile-time 8 x 16 tile (97.4 % of 300 x 300); a runtime 5 x 25 tile that divides 150 and 75 exactly spilled registers
    // and lost (round 1), so only the FIXED instantiations are built
    return d.in_scale ? launch_thin_impl<CIN_G, COUT_G, true, true>(d, stream)
                      : launch_thin_impl<CIN_G, COUT_G, true, false>(d, stream);
}

}  // namespace

// Eligibility + dispatch; called from gssd_conv2d_nhwc_f32 (conv_igemm.hip).  Returns 1 if not eligible.
int gssd_try_conv_thin(const gssd_conv_desc& d, hipStream_t stream) {
    const int cout_g = d.Cout / d.groups;
    const bool shape_ok = d.groups == 4 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.dil == 1 &&
                          d.in_stride == 4 * d.cin_g && d.in_ch_off == 0 && d.out_mode == GSSD_OUT_NHWC &&
                          d.out_stride == d.Cout && d.out_ch_off == 0 && !d.m_per_image && !d.relu && !d.gate &&
                          !d.resid && !d.alpha && d.split_k == 1 && d.wgt_row_stride == 9 * d.cin_g &&
                          d.H * d.W >= 75 * 75 && ((uintptr_t)d.out % 16) == 0;
    if (!shape_ok || (d.flags & GSSD_CONV_POOL2)) return 1;      // (no pooled epilogue here: conv1_2 takes the Winograd thin kernel)
    if (d.cin_g == 4 && cout_g == 16) return launch_thin<4, 16>(d, stream);
    if (d.cin_g == 16 && cout_g == 16) return launch_thin<16, 16>(d, stream);
    // conv2_1 with Winograd weights goes to conv_wino<32> (round 4: this direct kernel is pipe-bound -- fp32 MFMA + VALU 93 % busy -- at
    // 2.25 x the MACs of the Winograd form: 300 -> 262 us at B = 32); GSSD_CONV21_WINO=0 keeps it here
    static const bool c21_wino = []() { const char* e = getenv("GSSD_CONV21_WINO"); return !(e && e[0] == '0'); }();
    if (d.cin_g == 16 && cout_g == 32) return (c21_wino && d.wgt_wino) ? 1 : launch_thin<16, 32>(d, stream);
    return 1;
}
