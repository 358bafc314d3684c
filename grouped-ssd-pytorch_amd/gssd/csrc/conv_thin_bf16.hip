// Thin grouped 3x3 convolutions of the trunk in bf16 storage mode (conv1_1 .. conv2_2: 4 phase groups with 8..32 input and
// 16..32 output channels per group on the 300^2 / 150^2 maps -- 60 % of the trunk's bytes).  In bf16 these layers are pure HBM
// streams (arithmetic intensity 10..40 FLOP/B against a ridge of ~310), so the kernel is organised around moving every byte once:
//   * a persistent 256-thread workgroup walks 8 x 16-pixel tiles; wave g computes phase group g, so a pixel's whole channel
//     vector (64..256 B) is fetched as one burst by 16-byte LDS-DMA together with the 1-pixel halo (patch = 10 x 18 pixels);
//   * fused producer BatchNorm + ReLU (deferred BN of conv1_1 / conv2_1) is applied ONCE per patch element in LDS (fp32 math, one
//     bf16 rounding -- the same rounding the separate BN pass applies), not once per tap; out-of-image pixels are staged as zeros
//     and skipped by the transform, which is exactly zero padding after BN + ReLU;
//   * the nine taps are shifted ds_read_b128 of the patch: a lane's 8 consecutive k of v_mfma_f32_16x16x32_bf16 lie inside one tap;
//     the 16-byte units of a pixel are XOR-swizzled with the patch column on the DMA source side (conflict-free fragment reads);
//   * the group's weights (<= 32 x 288 bf16) live in registers as MFMA operands for the workgroup's lifetime;
//   * operand roles are swapped (A = weights, B = pixels) so a lane ends up with 4 (cout_g 16) or 8 (cout_g 32, staged in the
//     channel order of conv_bf16.hip) consecutive output channels of one pixel: 8 / 16-byte NHWC stores straight from registers;
//   * BatchNorm batch sums (of the fp32 accumulators) stay in registers across tiles: 16 shuffles + one fp64 atomic per channel
//     per workgroup at the end.
// Where the time goes (round 2, B = 32; -DTHIN_TIMING builds the per-phase shader-clock breakdown that scripts/thin_timing.py prints):
//   * the epilogue's stores are the largest single cost: with the stores compiled out conv1_1 ran 2.5 x faster and conv1_2 1.5 x; the
//     L2 receives the output as 32-byte write requests (TCP_TCC_WRITE_REQ = bytes / 32: a wave owns one group's 32-byte slice of each
//     pixel's 128 / 256-byte vector) and every EA write is a full 64-byte one (TCC_EA0_WRREQ_64B = all), so the slices do merge in L2;
//   * pairing rows through v_permlane16_swap (16-byte instead of 8-byte stores, half the store instructions) gave conv1_1 177 -> 161 us
//     and conv1_2 nothing: part issue-bound, mostly bound by the number of write requests;  nontemporal stores are much worse
//     (conv1_2 +48 %: the slices then reach HBM unmerged);
//   * 16-row patches (half the barriers and load round trips per pixel, 1.27 x instead of 1.41 x halo): -4 .. -7 %;
//   * the producer BatchNorm constants in registers instead of an LDS table (the table reads were 4 x the patch bytes): conv1_2 -5 %.
//   conv1_1 194 -> 161 us, conv1_2 231 -> 211 us, conv2_1 142 -> 135 us, conv2_2 176 -> 172 us.
//   * measured and rejected after that: a wave owning a PAIR of groups and half the rows, its two 8-byte pieces per pixel rearranged by
//     v_permlane32_swap + v_permlane16_swap into one 16-byte store of a 64-byte contiguous slice (half the write requests, same
//     instruction count): conv1_1 214 us, conv1_2 263 us -- wider slices are not what the store path wants either (conv2_1, whose
//     slices are 64 bytes by construction, sits at the same 2.1 TB/s); as was the LDS output tile (whole 128-byte lines) earlier.
// Measured and rejected earlier (conv1_1 190 us / conv1_2 232 us / conv2_1 142 us at the time): a second patch stage with the next
// tile's DMA in flight under the current tile (179 / 303 / 138 us), a persistent grid of 5 instead of 3 workgroups per CU
// (199 / 237 / 176 us), whole-pixel-vector stores through an LDS output tile (196 / 256 / 152 us).
// Also measured and rejected: this kernel for the 64-channel groups of conv3_x (weights of a group = 144 / 288 registers per lane, one
// workgroup per CU, rows in blocks of four): conv3_1 114 us vs 99 us, conv3_2 213 us vs 189 us on the generic conv_bf16 -- one wave
// per SIMD cannot hide the 92 KB patch load, the transform pass and the fragment reads behind its own MFMAs.
// Round 3, measured and rejected: a two-tile software pipeline of this kernel (8-row tiles, two patch buffers, the packed bf16
// outputs of tile t kept in registers and stored at the top of tile t + 1, so that neither the next patch's DMA nor the stores sit
// directly in front of the vmcnt(0) wait that covers them -- stores and loads share that counter on gfx950): conv1_1 171 us,
// conv1_2 232 us, conv2_1 149 us against 159 / 210 / 137 us.  The wait was not the limiter: scripts/ubench/store_patterns.hip writes
// this very store pattern (four waves, one 32-byte slice each of every 128-byte vector) at 5.3 TB/s in isolation, and the per-phase
// stamps put a conv2_1 tile at ~27 k cycles per workgroup for ~105 KB moved = 7.8 B/cycle per CU with two workgroups resident --
// the combined load + store rate one CU's memory pipe sustains (a device-wide copy runs 5.2 .. 5.5 TB/s = 9 B/cycle/CU).  What these
// kernels lose against a copy is phase overlap inside the CU (load -> wait -> MFMA -> store per workgroup), not store efficiency.
// Also rejected: a per-lane table of the staging instructions' patch offsets (one add + select per DMA instead of the division by the
// patch width, the swizzle and the bounds arithmetic; a wave-uniform interior-tile fast path): conv1_2 234 us, conv2_2 226 us against
// 217 / 192 us -- its 2 x 11 registers cost more (three workgroups per CU leave 170 VGPRs) than the ~300 VALU instructions per tile save.
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#ifdef THIN_TIMING
__device__ unsigned long long g_thin_timing[8];
extern "C" int gssd_thin_timing_read(unsigned long long* out8) {
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_thin_timing), 64);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_thin_timing), z, 64);
    return 0;
}
#define TSTAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; }
#else
#define TSTAMP(k)
#endif
// knock-out builds (scripts/thin_knockout.py; `make thin_ko`): which part of the tile loop costs what.  Results are WRONG on purpose.
//   1 no producer transform   2 no fragment reads   4 no MFMAs   8 trivial epilogue (no bias / batch sums / pooling)   16 no stores   32 no patch DMA
//   64 linear tile order   128 no unit swizzle on the DMA source   256 no batch-sum atomics   512 no weight prologue
#ifndef THIN_KO
#define THIN_KO 0
#endif

namespace {

__device__ __attribute__((aligned(16))) u16 g_zero_thin_h[8] = {0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct ThinBfParams {
    const u16* in;
    const u16* wgt;      // packed bf16 [Cout][9*CIN_G]
    const float* bias;
    u16* out;
    double* stats;
    int stats_rep;
    const float* in_scale;
    const float* in_shift;
    const float* pool_sign;   // GSSD_CONV_POOL2: `out` is the 2x2 / stride-2 pooled raw map, max where pool_sign[c] >= 0 else min
    int B, H, W, tiles_y, tiles_x;
};

template <int UPR>
__device__ __forceinline__ int swz(int col) {
    return UPR >= 8 ? (col & (UPR - 1)) & 15 : ((col ^ (col >> 2)) & (UPR - 1));
}

constexpr int TW = 16, PW = TW + 2;       // tile width, patch width

// TH = output rows per staged patch (a multiple of RH): one DMA round trip, one transform pass and three barriers per TH x 16 pixels.
// 16 rows where the (TH + 2) x 18 patch still lets 2-3 workgroups share a CU (<= 64 input channels): half the barriers and exposed
// load latency per pixel of the 8-row tile and 10 % less halo (1.27 x instead of 1.41 x).
// POOL: the GSSD_CONV_POOL2 epilogue (its own instance: both epilogues in one kernel cost registers -- spills at the occupancy caps)
template <int CIN_G, int COUT_G, bool XF, int TH, bool POOL>
__global__ __launch_bounds__(256, (COUT_G > 16 ? 2 : 3)) void conv_thin_bf16_kernel(const ThinBfParams p) {
    constexpr int RH = 2;                          // rows per MFMA / epilogue batch: a row PAIR (pooling window rows, paired 16-byte stores)
    constexpr int PH = TH + 2, NPATCH = PH * PW;
    constexpr int CIN = 4 * CIN_G, COUT = 4 * COUT_G;
    constexpr int UPR = CIN / 8;                   // 16-byte units per pixel
    constexpr int PPI = 64 / UPR;                  // pixels per DMA wave instruction
    constexpr int NT = COUT_G / 16;
    constexpr int KR = (3 * CIN_G + 31) / 32;      // 32-k MFMA steps per INPUT ROW (its 3 taps x CIN_G channels, zero padded): 1 / 2 / 3
    constexpr int NINSTR = (NPATCH + PPI - 1) / PPI;
    constexpr int CPL = NT == 1 ? 4 : 8;           // consecutive output channels per lane
    extern __shared__ __attribute__((aligned(16))) u16 patch[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = conv group
    const int r = lane & 15, kq = lane >> 4;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int ntiles = p.B * tiles_per_img;

    // ---- weights -> registers (A operand), K ordered by INPUT ROW: step (dy, kk) covers k' = 32 kk + 8 kq .. + 8 of the row's
    // (dx, channel) run, dx = k' / CIN_G (zero weights where dx > 2).  Round 4: with the contraction ordered this way the B fragments
    // of one input row (its KR shifted reads) serve the THREE output rows that touch it, so a row costs KR LDS reads instead of
    // ceil(9 CIN_G / 32) (2 instead of 5 at 16 channels per group, 3 instead of 9 at 32) and they are issued a row pair ahead of
    // their MFMAs -- the knock-out builds put 54 of conv1_2's 199 us into fragment reads the MFMAs waited for one by one.
    bf16x8 wf[3][KR][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ch = NT == 1 ? r : 8 * (r >> 2) + 4 * j + (r & 3);
        const u16* wr = p.wgt + (size_t)(g * COUT_G + ch) * (9 * CIN_G);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int kk = 0; kk < KR; ++kk) {
                const int k = 32 * kk + 8 * kq, dx = k / CIN_G, c = k - dx * CIN_G;
                wf[dy][kk][j] = (dx < 3 && !(THIN_KO & 512)) ? *reinterpret_cast<const bf16x8*>(wr + (dy * 3 + dx) * CIN_G + c)
                                                             : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
    }
    // this lane's output channels (inside the group): cb .. cb + CPL
    const int cb = NT == 1 ? 4 * kq : 8 * kq;
    // bias and the pooled epilogue's BatchNorm-weight signs live in registers for the kernel's lifetime: an ordinary (VGPR-returning) load
    // inside the tile loop makes hipcc wait vmcnt(0) where its destination registers are next written -- loads and stores share that
    // counter on gfx950, so every row batch then waited for the previous batch's STORES to be acknowledged (round 4: the ISA of the
    // pooled epilogue had one such wait per row pair, the unpooled <16,32> kernel one per 8-row batch)
    float bias[CPL];
    unsigned smask[CPL];       // pooled epilogue: sign-flip mask of the channel (min = -max(-x): one max per window instead of max AND min)
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        bias[c] = p.bias ? p.bias[g * COUT_G + cb + c] : 0.f;
        smask[c] = (POOL && p.pool_sign[g * COUT_G + cb + c] < 0.f) ? 0x80000000u : 0u;
    }

    // ---- fragment offsets (elements) inside a patch row per k-step (lanes whose dx > 2 carry zero weights: they read dx = 2) -----
    int foff[KR];
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) {
        const int k = 32 * kk + 8 * kq;
        int dx = k / CIN_G;
        const int coff = k - dx * CIN_G;
        if (dx > 2) dx = 2;
        const int col = r + dx;
        const int unit = (g * CIN_G + coff) >> 3;
        foff[kk] = col * CIN + ((unit ^ swz<UPR>(col)) << 3);
    }

    float ssum[CPL], ssq[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) ssum[c] = ssq[c] = 0.f;
    float xs[8], xh[8];                                    // producer BatchNorm scale / shift of this thread's eight channels
    if constexpr (XF) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xs[e] = p.in_scale[(tid % UPR) * 8 + e];
            xh[e] = p.in_shift[(tid % UPR) * 8 + e];
        }
    }

#ifdef THIN_TIMING
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    const int bperm = (gridDim.x & 7) == 0 && !(THIN_KO & 64) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    for (int tile = bperm; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;

        // ---- stage the patch: pixel vectors of CIN bf16, units XOR-swizzled by the patch column --------------------------------
        for (int i = g; i < NINSTR; i += 4) {
            const int pp = i * PPI + lane / UPR;
            const int py = pp / PW, pxx = pp - py * PW;
            const int lu = (THIN_KO & 128) ? (lane % UPR) : (lane % UPR) ^ swz<UPR>(pxx);
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            const bool ok = pp < NPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const u16* src = ok ? p.in + ((size_t)(b * p.H + iy) * p.W + ix) * CIN + lu * 8 : g_zero_thin_h;
            if (!(THIN_KO & 32)) dma16(src, patch + i * PPI * CIN);
        }
        TSTAMP(0)
        __syncthreads();
        TSTAMP(1)
        if constexpr (XF && !(THIN_KO & 1)) {
            // producer BatchNorm + ReLU once per patch element (in-image pixels only: the rest stays 0 = padding after the transform).
            // A thread always transforms the SAME eight channels (unit tid % UPR of every (256 / UPR)-th pixel; the swizzle only
            // moves where that unit sits inside the pixel), so its scale / shift live in 16 registers for the kernel's lifetime.
            // Round 4 (knock-out builds: this pass cost 30 us of conv1_2's 199): the patch position advances by additions instead of a
            // division per element, tiles whose patch lies inside the image skip the bounds tests (wave-uniform), the arithmetic runs on
            // channel PAIRS (v_pk_fma_f32) and the ReLU on the rounded pairs (v_pk_max_i16 against 0: a negative bf16 is a negative int16;
            // -0 and negatives that round to -0 become +0 exactly like fmaxf(x, 0) followed by the rounding).
            constexpr int STEP = 256 / UPR, SY = STEP / PW, SX = STEP % PW;
            const bool inside = y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= p.H && x0 + TW + 1 <= p.W;
            int pp = tid / UPR, py = pp / PW, pxx = pp - py * PW;
            for (; pp < NPATCH; pp += STEP) {
                const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
                if (inside || ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)) {
                    u16* at = patch + (pp * UPR + ((tid % UPR) ^ swz<UPR>(pxx))) * 8;
                    typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef short s16x2 __attribute__((ext_vector_type(2)));
                    u32x4t v = *reinterpret_cast<const u32x4t*>(at);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f32x2 x = {__builtin_bit_cast(float, v[e] << 16), __builtin_bit_cast(float, v[e] & 0xffff0000u)};
                        x = __builtin_elementwise_fma(x, f32x2{xs[2 * e], xs[2 * e + 1]}, f32x2{xh[2 * e], xh[2 * e + 1]});
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const bf16x2 h = {(__bf16)x[0], (__bf16)x[1]};
                        const s16x2 z = __builtin_elementwise_max(__builtin_bit_cast(s16x2, h), s16x2{0, 0});
                        v[e] = __builtin_bit_cast(unsigned, z);
                    }
                    *reinterpret_cast<u32x4t*>(at) = v;
                }
                py += SY;
                pxx += SX;
                if (pxx >= PW) {
                    pxx -= PW;
                    ++py;
                }
            }
            TSTAMP(2)
            __syncthreads();
            TSTAMP(3)
        }

        // ---- MFMAs straight from the patch, a row PAIR per step.  F[rho] = the KR fragments of input (patch) row 2 pr + rho; output rows
        // 2 pr and 2 pr + 1 contract F[0..2] and F[1..3]; the fragments of the next pair's two new rows are read BEFORE this pair's MFMAs
        const int x = x0 + r;
        auto load_row = [&](int row, bf16x8 (&dst)[KR]) {
#pragma unroll
            for (int kk = 0; kk < KR; ++kk) {
                if (THIN_KO & 2) dst[kk] = wf[0][kk][0];
                else dst[kk] = *reinterpret_cast<const bf16x8*>(patch + foff[kk] + row * PW * CIN);
            }
        };
        bf16x8 F[4][KR], G[2][KR];
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) load_row(rho, F[rho]);
#pragma unroll
        for (int rb = 0; rb < TH; rb += RH) {
            if (y0 + rb >= p.H) break;
            if (rb + RH < TH) {
                load_row(rb + 4, G[0]);
                load_row(rb + 5, G[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[RH][NT];
#pragma unroll
            for (int i = 0; i < RH; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int kk = 0; kk < KR; ++kk)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int i = 0; i < RH; ++i) {
                            if (THIN_KO & 4) acc[i][j][kk & 3] += (float)F[i + dy][kk][j];
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy][kk][j], F[i + dy][kk], acc[i][j], 0, 0, 0);
                        }
#pragma unroll
            for (int kk = 0; kk < KR; ++kk) {
                F[0][kk] = F[2][kk];
                F[1][kk] = F[3][kk];
                F[2][kk] = G[0][kk];
                F[3][kk] = G[1][kk];
            }
#ifdef THIN_TIMING
            asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[RH - 1][NT - 1][3]));      // the MFMAs have retired
#endif
            TSTAMP(6)
            if (THIN_KO & 8) {
#pragma unroll
                for (int i = 0; i < RH; ++i) {
                    const int y = y0 + rb + i;
                    if (y >= p.H || x >= p.W) continue;
                    u16* dst = p.out + ((size_t)(b * p.H + y) * p.W + x) * COUT + g * COUT_G + cb;
                    if (POOL) {
                        if ((i & 1) || (r & 1)) continue;
                        dst = p.out + ((size_t)(b * ((p.H + 1) >> 1) + (y >> 1)) * ((p.W + 1) >> 1) + (x >> 1)) * COUT + g * COUT_G + cb;
                    }
                    if constexpr (CPL == 4) {
                        if (!(THIN_KO & 16)) *reinterpret_cast<bf16x4*>(dst) = bf16x4{(__bf16)acc[i][0][0], (__bf16)acc[i][0][1], (__bf16)acc[i][0][2], (__bf16)acc[i][0][3]};
                    } else {
                        bf16x8 h;
#pragma unroll
                        for (int c = 0; c < 8; ++c) h[c] = (__bf16)acc[i][c >> 2][c & 3];
                        if (!(THIN_KO & 16)) *reinterpret_cast<bf16x8*>(dst) = h;
                    }
                    ssum[0] += acc[i][0][0];
                }
                continue;
            }
            if constexpr (POOL) {
                // GSSD_CONV_POOL2 (include/gssd_hip.h): rows (i, i + 1) of a lane and columns (r, r ^ 1) of neighbouring lanes form one
                // pooling window (tile origins are even); batch sums over every pixel as usual, then ONE value per window and channel --
                // the maximum where the channel's BatchNorm weight is >= 0, the minimum where it is negative -- rounded to bf16 (the
                // rounding commutes with max / min) and stored by the even lane at the pooled position.  The full map is never written.
                const int Hp = (p.H + 1) >> 1, Wp = (p.W + 1) >> 1;
                // x ^ smask: the channel's values with the sign flipped where the BatchNorm weight is negative -- the window's maximum of
                // those, flipped back, is the maximum (gamma >= 0) or the minimum (gamma < 0) of the window, exactly
                auto flip = [](float v, unsigned m) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ m); };
                if (y0 + rb + RH <= p.H && x0 + TW <= p.W) {
                    // the 8 x 16 block lies inside the image (wave-uniform; 90 % of the tiles): no per-pixel guards.  Batch sums in the
                    // unpooled epilogue's order and arithmetic form (CPL 4: multiply, then add; CPL 8: fused) -- identical statistics.
#pragma unroll
                    for (int i = 0; i < RH; i += 2) {
                        float m[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; ++c) {
                            const float v0 = (CPL == 4 ? acc[i][0][c] : acc[i][c >> 2][c & 3]) + bias[c];
                            const float v1 = (CPL == 4 ? acc[i + 1][0][c] : acc[i + 1][c >> 2][c & 3]) + bias[c];
                            if constexpr (CPL == 4) {
                                float q0 = v0 * v0, q1 = v1 * v1;
                                asm volatile("" : "+v"(q0), "+v"(q1));          // (no contraction into the additions below)
                                ssum[c] += v0;
                                ssq[c] += q0;
                                ssum[c] += v1;
                                ssq[c] += q1;
                            } else {
                                ssum[c] += v0;
                                ssq[c] = __builtin_fmaf(v0, v0, ssq[c]);
                                ssum[c] += v1;
                                ssq[c] = __builtin_fmaf(v1, v1, ssq[c]);
                            }
                            m[c] = fmaxf(flip(v0, smask[c]), flip(v1, smask[c]));
                        }
#pragma unroll
                        for (int c = 0; c < CPL; ++c)        // neighbouring column = neighbouring lane: DPP quad_perm [1, 0, 3, 2]
                            m[c] = fmaxf(m[c], __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, m[c]), 0xB1, 0xF, 0xF, true)));
                        if (!(r & 1)) {
                            u16* dst = p.out + ((size_t)(b * Hp + ((y0 + rb + i) >> 1)) * Wp + (x >> 1)) * COUT + g * COUT_G + cb;
                            if constexpr (CPL == 4) {
                                bf16x4 h;
#pragma unroll
                                for (int c = 0; c < 4; ++c) h[c] = (__bf16)flip(m[c], smask[c]);
                                if (!(THIN_KO & 16)) *reinterpret_cast<bf16x4*>(dst) = h;
                            } else {
                                bf16x8 h;
#pragma unroll
                                for (int c = 0; c < 8; ++c) h[c] = (__bf16)flip(m[c], smask[c]);
                                if (!(THIN_KO & 16)) *reinterpret_cast<bf16x8*>(dst) = h;
                            }
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int i = 0; i < RH; i += 2) {
                    const int y = y0 + rb + i;
                    const bool ok0 = y < p.H && x < p.W, ok1 = y + 1 < p.H && x < p.W;
                    float mx[CPL];
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const float v0 = (CPL == 4 ? acc[i][0][c] : acc[i][c >> 2][c & 3]) + bias[c];
                        const float v1 = (CPL == 4 ? acc[i + 1][0][c] : acc[i + 1][c >> 2][c & 3]) + bias[c];
                        // row by row and in the arithmetic form of the unpooled epilogue (CPL 4: select, then add; CPL 8: the guarded
                        // `ssq += v * v` hipcc contracts into an fma): identical batch sums
                        if constexpr (CPL == 4) {
                            ssum[c] += ok0 ? v0 : 0.f;
                            ssq[c] += ok0 ? v0 * v0 : 0.f;
                            ssum[c] += ok1 ? v1 : 0.f;
                            ssq[c] += ok1 ? v1 * v1 : 0.f;
                        } else {
                            if (ok0) {
                                ssum[c] += v0;
                                ssq[c] += v0 * v0;
                            }
                            if (ok1) {
                                ssum[c] += v1;
                                ssq[c] += v1 * v1;
                            }
                        }
                        mx[c] = ok1 ? fmaxf(flip(v0, smask[c]), flip(v1, smask[c])) : flip(v0, smask[c]);
                    }
                    // neighbouring column = neighbouring lane: DPP quad_perm [1, 0, 3, 2] (one VALU move, no LDS crossbar)
                    const bool pok = __builtin_amdgcn_mov_dpp((int)ok0, 0xB1, 0xF, 0xF, true) != 0;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const float pmx = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, mx[c]), 0xB1, 0xF, 0xF, true));
                        if (pok) mx[c] = fmaxf(mx[c], pmx);
                    }
                    if (ok0 && !(r & 1)) {
                        u16* dst = p.out + ((size_t)(b * Hp + (y >> 1)) * Wp + (x >> 1)) * COUT + g * COUT_G + cb;
                        if constexpr (CPL == 4) {
                            bf16x4 h;
#pragma unroll
                            for (int c = 0; c < 4; ++c) h[c] = (__bf16)flip(mx[c], smask[c]);
                            if (!(THIN_KO & 16)) *reinterpret_cast<bf16x4*>(dst) = h;
                        } else {
                            bf16x8 h;
#pragma unroll
                            for (int c = 0; c < 8; ++c) h[c] = (__bf16)flip(mx[c], smask[c]);
                            if (!(THIN_KO & 16)) *reinterpret_cast<bf16x8*>(dst) = h;
                        }
                    }
                }
                continue;
            }
            if constexpr (!POOL) {
            // ---- epilogue: + bias, batch sums, 16-byte NHWC stores (lane: pixel (y0 + rb + i, x0 + r), CPL consecutive channels) -----
            if constexpr (CPL == 4) {
                // 4 channels = 8 bytes per lane and row: rows are taken in pairs and the kq-even / kq-odd lane rows swap halves
                // (v_permlane16_swap), so an even lane stores 16 contiguous bytes of row i (its own 4 channels | its neighbour's) and
                // an odd lane 16 bytes of row i + 1 -- half the store instructions for the same bytes (the epilogue was store-ISSUE
                // bound: without its stores the conv1_1 kernel ran 2.5 x faster)
#pragma unroll
                for (int i = 0; i < RH; i += 2) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const bool ok = y0 + rb + i + h < p.H && x < p.W;
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            v[c] = acc[i + h][0][c] + bias[c];
                            ssum[c] += ok ? v[c] : 0.f;
                            ssq[c] += ok ? v[c] * v[c] : 0.f;
                        }
                        const bf16x4 hv = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        pk[h][0] = reinterpret_cast<const unsigned*>(&hv)[0];
                        pk[h][1] = reinterpret_cast<const unsigned*>(&hv)[1];
                    }
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(pk[0][d], pk[1][d], false, false);
                        pk[0][d] = sw[0];
                        pk[1][d] = sw[1];
                    }
                    const int y = y0 + rb + i + (kq & 1);
                    if (y < p.H && x < p.W) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        u16* dst = p.out + ((size_t)(b * p.H + y) * p.W + x) * COUT + g * COUT_G + 4 * (kq & ~1);
                        if (!(THIN_KO & 16)) *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < RH; ++i) {
                    const int y = y0 + rb + i;
                    if (y >= p.H || x >= p.W) continue;
                    float v[CPL];
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        v[c] = acc[i][c >> 2][c & 3] + bias[c];
                        ssum[c] += v[c];
                        ssq[c] += v[c] * v[c];
                    }
                    u16* dst = p.out + ((size_t)(b * p.H + y) * p.W + x) * COUT + g * COUT_G + cb;
                    bf16x8 h;
#pragma unroll
                    for (int c = 0; c < 8; ++c) h[c] = (__bf16)v[c];
                    if (!(THIN_KO & 16)) *reinterpret_cast<bf16x8*>(dst) = h;
                }
            }
            }
        }
        TSTAMP(4)
        __syncthreads();          // every wave is done reading the patch
        TSTAMP(5)
    }

#ifdef THIN_TIMING
    if (lane == 0 && g == 1)
        for (int k = 0; k < 7; ++k) atomicAdd(&g_thin_timing[k], tacc[k]);
    if (tid == 0) atomicAdd(&g_thin_timing[7], 1ull);
#endif
    if (p.stats && !(THIN_KO & 256)) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            double s = (double)ssum[c], q = (double)ssq[c];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s += __shfl_xor(s, o, 64);
                q += __shfl_xor(q, o, 64);
            }
            if (r == 0) {
                double* st = gssd_stats_replica(p.stats, p.stats_rep, COUT);
                unsafeAtomicAdd(st + g * COUT_G + cb + c, s);
                unsafeAtomicAdd(st + COUT + g * COUT_G + cb + c, q);
            }
        }
    }
}

template <int CIN_G, int COUT_G, bool XF, bool POOL>
int launch_thin_bf16(const gssd_conv_desc& d, hipStream_t stream) {
    constexpr int CIN = 4 * CIN_G;
    constexpr int TH = CIN <= 64 ? 16 : 8;
    constexpr int NPATCH = (TH + 2) * PW;
    constexpr int UPR = CIN / 8, PPI = 64 / UPR;
    ThinBfParams p;
    p.in = reinterpret_cast<const u16*>(d.in);
    p.wgt = reinterpret_cast<const u16*>(d.wgt);
    p.bias = d.bias;
    p.out = reinterpret_cast<u16*>(d.out);
    p.stats = d.stats;
    p.stats_rep = d.stats_rep;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.pool_sign = (d.flags & GSSD_CONV_POOL2) ? d.pool_sign : nullptr;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    p.tiles_y = (d.H + TH - 1) / TH;
    p.tiles_x = (d.W + TW - 1) / TW;
    const size_t smem = (size_t)((NPATCH + PPI - 1) / PPI) * PPI * CIN * sizeof(u16);
    const long long ntiles = (long long)d.B * p.tiles_y * p.tiles_x;
    int grid = 256 * (COUT_G > 16 ? 2 : 3);      // workgroups per CU = the kernel's launch bound (three for conv2_1, whose 166 registers
    //                                              would allow it: 90 us against 80 with two -- round 4)
    if (ntiles < grid) grid = (int)ntiles;
    hipLaunchKernelGGL((conv_thin_bf16_kernel<CIN_G, COUT_G, XF, TH, POOL>), dim3(grid), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

// Eligibility + dispatch; called from gssd_conv2d_nhwc_bf16 (conv_bf16.hip).  Returns 1 if not eligible.
int gssd_try_conv_thin_bf16(const gssd_conv_desc& d, hipStream_t stream) {
    const int cout_g = d.Cout / d.groups;
    const bool shape_ok = d.groups == 4 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.dil == 1 &&
                          d.in_stride == 4 * d.cin_g && d.in_ch_off == 0 && d.out_mode == GSSD_OUT_NHWC && d.out_stride == d.Cout &&
                          d.out_ch_off == 0 && !d.m_per_image && !d.relu && !d.gate && !d.resid && !d.alpha && d.split_k == 1 &&
                          d.wgt_row_stride == 9 * d.cin_g && d.H * d.W >= 75 * 75 && (d.flags == 0 || (d.flags == GSSD_CONV_POOL2 && d.pool_sign));
    if (!shape_ok) return 1;
    const bool pool = (d.flags & GSSD_CONV_POOL2) != 0;
#define THIN_CASE(CI, CO)                                                                                                          \
    if (d.cin_g == CI && cout_g == CO)                                                                                             \
        return d.in_scale ? (pool ? launch_thin_bf16<CI, CO, true, true>(d, stream) : launch_thin_bf16<CI, CO, true, false>(d, stream)) \
                          : (pool ? launch_thin_bf16<CI, CO, false, true>(d, stream) : launch_thin_bf16<CI, CO, false, false>(d, stream));
    THIN_CASE(8, 16)
    THIN_CASE(16, 16)
    THIN_CASE(16, 32)
    THIN_CASE(32, 32)
#undef THIN_CASE
    return 1;
}
