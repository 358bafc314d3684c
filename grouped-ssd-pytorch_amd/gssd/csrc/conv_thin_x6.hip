// Thin grouped 3x3 trunk convolutions of the fp32 mode on the BF16 matrix cores with fp32-equivalent products (round 6): conv1_2
// (4 groups x 16 -> 16 channels, 300 x 300), conv2_1 (16 -> 32, 150 x 150) and conv2_2 (32 -> 32, 150 x 150) of
// models/ssd_multiphase_custom_group.py:434-460 (vgg()).
//
// Until round 5 these three layers ran fp32-MFMA kernels -- conv_thin_wino.hip (471 us at B = 32 inside the step) and conv_wino.hip<32>
// (269 / 429 us) -- 1.17 ms of the step's critical path; this kernel: 370 / 175 / 256 us = 0.80 ms (profiles/r06_thin_x6_*.txt).  The three-plane Winograd kernel (conv_wino_x6.hip) loses at these
// channel counts: its producers transform and split every input tile once per 32-channel output block.  This kernel is the DIRECT
// convolution, patch-staged like conv_thin.hip, with the split done ONCE per input element:
//   * a persistent 256-thread workgroup (two per CU) owns an 8 x 16-pixel tile of one image and a channel SLAB: all 64 input channels (four phase groups) of conv1_2, 32 channels = two groups
//     of conv2_1, one group of conv2_2 (the slab is fixed per workgroup, so its weights never change);
//   * staging: every thread loads 8 fp32 channels of 3 .. 6 of the 10 x 18 patch pixels (32 bytes each), one tile AHEAD into registers; applies the producer's deferred BatchNorm + ReLU (gssd_conv_desc::in_scale / in_shift; out-of-image pixels
//     become zeros = zero padding after the transform), splits every value into the exact sum of three bf16 (x = h + m + l, conv_x6.hip) and
//     writes the three planes to LDS as [plane][pixel][64 channels], 16-byte units XOR-swizzled by the patch column (conv_thin_bf16.hip);
//   * wave roles: wave = (group, 16-channel half of the group's outputs, half of the tile's rows) -- 4 x 1 x 1 for conv1_2, 2 x 2 x 1 for
//     conv2_1, 1 x 2 x 2 for conv2_2.  The wave's weights (16 output channels x 9 taps x cin_g) are split in
//     the prologue and live in registers as MFMA A operands for the workgroup's lifetime, K ordered by INPUT ROW (3 taps x cin_g, padded to
//     32-k steps) so that one input row's fragments serve the three output rows that touch it;
//   * input-row-major K loop: the fragments of ONE patch row are live at a time (3 planes x KR reads of 16 bytes); each of the <= 3 output rows
//     the row contributes to gets six v_mfma_f32_16x16x32_bf16 per 32-k step (every product term above 2^-24, smallest first), accumulated in
//     the matrix pipe's fp32 accumulators (K <= 288: the in-place chain is short);
//   * operand roles swapped (A = weights, B = pixels): a lane ends up with 4 consecutive output channels of one pixel -- 16-byte NHWC fp32
//     stores straight from the accumulators.  Epilogue: + bias, BatchNorm batch sums (fp32 per lane across the workgroup's tiles, fp64 across
//     lanes / workgroups / replicas), and either the full map or -- GSSD_CONV_POOL2 -- the 2 x 2 / stride 2 max- / min-pooled raw map by the
//     sign of the consumer BatchNorm's weight (conv1_2, conv2_2: the full-resolution map is never written).
// Not Winograd: the products are those of the direct convolution, so the result is as close to float64 as an exact fp32 FMA chain
// (tests/test_gpu_thin_x6.py).  GSSD_THIN_X6=0 keeps the round-5 kernels (ablation / A-B).
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#ifndef TX6_FENCE
#define TX6_FENCE 1
#endif
#ifndef TX6_KO
#define TX6_KO 0              // knock-outs (timing only, results wrong): 1 no transform / split, 2 no MFMAs, 4 no fragment reads, 8 no stores, 16 no loads
#endif

// per-phase stamps (make thin_x6_timing; scripts/bench_thin_x6.py prints them): s_memtime deltas of wave 1 of every workgroup, summed
#ifdef TX6_TIMING
__device__ unsigned long long g_tx6_timing[8];
extern "C" int gssd_thin_x6_timing_read(unsigned long long* out8) {
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tx6_timing), 64);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_tx6_timing), z, 64);
    return 0;
}
#define TSTAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; }
#else
#define TSTAMP(k)
#endif

namespace {

__device__ __attribute__((aligned(16))) float g_zero_thin_x6[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

constexpr int TH = 8, TW = 16, PW = TW + 2, PH = TH + 2, NPATCH = PH * PW;      // 180 patch pixels
constexpr int NT_ = 256;                  // threads per workgroup (four waves)
constexpr int NPAD = 192;                 // patch pixels incl. the padding of the staging passes
// input channels staged per workgroup: the whole pixel vector of conv1_2, two groups of conv2_1, one group of conv2_2
constexpr int slab_channels(int cin_g, int cout_g) { return (cin_g == 16 && cout_g == 16) ? 64 : 32; }
// 16-byte units of a staged pixel are XOR-swizzled by the patch column so that the lane groups a ds_read_b128 is serviced in
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- MI355X_MICROARCH.md, LDS table) hit 16 different 16-byte slots of the 256-byte bank row.
// Found by exhaustive search over the linear maps of the column bits against this kernel's fragment addresses (three k-step / group patterns):
// 128-byte pixels: col & 7 (conv_thin_bf16.hip's); 64-byte pixels: bit 2 of the column into bit 1 of the unit.  [(col >> 1) & 7 and
// (col >> 2) & 3, which are conflict-free for 16 CONSECUTIVE lanes, are 2-way conflicts for the real groups.]
template <int UPS>
__device__ __forceinline__ int swz(int col) { return UPS == 8 ? (col & 7) : ((col >> 1) & 2); }

struct ThinX6Params {
    const float* in;
    const float* wgt;        // packed fp32 K-major [Cout][9 * cin_g] (gssd_pack_conv_weight_f32)
    const float* bias;
    float* out;
    double* stats;
    int stats_rep;
    const float* in_scale;
    const float* in_shift;
    const float* in_pad;     // per input channel: a value the producer transform maps to 0 (out-of-image patch pixels LOAD it: address select)
    const float* pool_sign;  // GSSD_CONV_POOL2: `out` is the pooled raw map, max where pool_sign[c] >= 0 else min
    int B, H, W, tiles_y, tiles_x;
    int* ctr;                // [slabs][8 XCDs] tile counters, zero before the launch (NULL: static tile shares)
};

// the split of TWO values at once, planes as packed bf16 pairs (a in the low half): conv_x6.hip::split3_pair
__device__ __forceinline__ void split3_pair(const float a, const float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
    const float sa = ra - __builtin_bit_cast(float, pm << 16), sb = rb - __builtin_bit_cast(float, pm & 0xffff0000u);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sa, sb}, bf16x2));
}

// F16 form (round 6; launches with the fused producer BatchNorm + ReLU, i.e. operands that are normalised activations): TWO fp16 planes per operand,
// x = h + l' / 2048 with h = fp16(x), l' = fp16((x - h) * 2048) -- round to nearest twice, |x - h - l' / 2048| <= 2^-24 |x|: what fp32 itself keeps --
// and THREE v_mfma_f32_16x16x32_f16 per product: h h' into one accumulator, h l' + l' h' into a second one that enters with the factor 1 / 2048 (the
// l' l'' term is below 2^-24).  Half the matrix instructions and LDS fragment reads, a split of 3 instead of 5.5 vector instructions per value.
// The residual is scaled because fp16 has no exponent range to spare (unscaled it would be a subnormal for |x| < 0.25); h overflows above 65 504,
// which a BatchNorm + ReLU output does not reach; values below 6e-5 lose relative, not absolute, accuracy.  Accuracy against float64: that of an
// fp32 FMA chain (4e-7 of the output scale; the three-plane bf16 form: 1.5e-7).  GSSD_X6_F16=0: the bf16 planes everywhere.
__device__ __forceinline__ void split2_pair(const float a, const float b, unsigned& ph, unsigned& pl) {
    const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
    const f32x2 r = (f32x2{a, b} - __builtin_convertvector(h, f32x2)) * 2048.f;
    ph = __builtin_bit_cast(unsigned, h);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

template <int CIN_G, int COUT_G, bool XF, bool POOL, bool F16>
__global__ __launch_bounds__(NT_, 2) void conv_thin_x6_kernel(const ThinX6Params p) {
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int CIN = 4 * CIN_G, COUT = 4 * COUT_G;
    constexpr int CW = slab_channels(CIN_G, COUT_G);
    constexpr int UPS = CW / 8;                    // 16-byte units (8 bf16) per staged pixel: 8 / 4
    constexpr int PPP = NT_ / UPS;                 // pixels per staging pass: 32 / 64
    constexpr int NIT = (NPATCH + PPP - 1) / PPP;  // staging items per thread: 6 / 3
    constexpr int PLANE = NPAD * CW;               // u16 elements per plane
    static_assert(NIT * PPP <= NPAD, "patch padding");
    constexpr int NG = CW / CIN_G;                 // groups per slab: 4 / 2 / 1
    constexpr int NGI = 4 / NG;                    // slabs per tile: 1 / 2 / 4
    constexpr int CH = COUT_G / 16;                // 16-channel output tiles per group
    constexpr int UNITS = NG * CH;                 // (group, output half) pairs per slab: 4 / 4 / 2
    constexpr int RS = 4 / UNITS;                  // row splits: 1 / 1 / 2
    constexpr int RW = TH / RS;                    // output rows per wave: 8 / 8 / 4
    constexpr int KR = (3 * CIN_G + 31) / 32;      // 32-k steps per input row: 2 / 3
    extern __shared__ __attribute__((aligned(16))) u16 planes[];          // [3][NPAD][CW], then the scale / shift table [2][CW] floats, then two ints
    float* const xtab = reinterpret_cast<float*>(planes + NPL * PLANE);
    int* const s_next = reinterpret_cast<int*>(xtab + 2 * CW);            // [2]: tiles claimed by thread 0 for the workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int unit = wave % UNITS, rs = wave / UNITS;
    const int gi = unit % NG, ch = unit / NG;
    // XCD-aware persistent order (conv_thin.hip): each XCD works on gridDim / 8 consecutive slots of every sweep
    const int bperm = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int gpi = NGI == 1 ? 0 : bperm % NGI;           // channel slab of this workgroup (fixed: its weights never change)
    const int wg0 = bperm / NGI, nwg = (int)gridDim.x / NGI;
    const int g = gpi * NG + gi;                          // conv group of this wave
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int ntiles = p.B * tiles_per_img;

    // ---- weights -> registers: the three planes of W[16 output channels][dy][32 kk + 8 kq .. + 8] (k' runs over (dx, channel) of one input row)
    bf16x8 wf[NPL][3][KR];
    {
        const float* wr = p.wgt + (size_t)(g * COUT_G + 16 * ch + r) * (9 * CIN_G);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int kk = 0; kk < KR; ++kk) {
                const int k = 32 * kk + 8 * kq, dx = k / CIN_G, c = k - dx * CIN_G;
                f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
                if (dx < 3) {
                    w0 = *reinterpret_cast<const f32x4*>(wr + (dy * 3 + dx) * CIN_G + c);
                    w1 = *reinterpret_cast<const f32x4*>(wr + (dy * 3 + dx) * CIN_G + c + 4);
                }
                unsigned q[3][4];
                if constexpr (F16) {
                    split2_pair(w0[0], w0[1], q[0][0], q[1][0]);
                    split2_pair(w0[2], w0[3], q[0][1], q[1][1]);
                    split2_pair(w1[0], w1[1], q[0][2], q[1][2]);
                    split2_pair(w1[2], w1[3], q[0][3], q[1][3]);
                } else {
                    split3_pair(w0[0], w0[1], q[0][0], q[1][0], q[2][0]);
                    split3_pair(w0[2], w0[3], q[0][1], q[1][1], q[2][1]);
                    split3_pair(w1[0], w1[1], q[0][2], q[1][2], q[2][2]);
                    split3_pair(w1[2], w1[3], q[0][3], q[1][3], q[2][3]);
                }
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) wf[pl][dy][kk] = __builtin_bit_cast(bf16x8, u32x4{q[pl][0], q[pl][1], q[pl][2], q[pl][3]});
            }
    }
    // this lane's output channels (inside the layer): co0 .. co0 + 3
    const int co0 = g * COUT_G + 16 * ch + 4 * kq;
    float bias[4];
    unsigned smask[4];           // pooled epilogue: sign-flip mask (min = -max(-x))
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bias[c] = p.bias ? p.bias[co0 + c] : 0.f;
        smask[c] = (POOL && p.pool_sign[co0 + c] < 0.f) ? 0x80000000u : 0u;
    }
    if (XF && tid < CW) {
        xtab[tid] = p.in_scale[gpi * CW + tid];
        xtab[CW + tid] = p.in_shift[gpi * CW + tid];
    }

    // ---- fragment offsets (u16 elements inside a plane) per k-step: lanes whose dx > 2 carry zero weights and read dx = 2
    int foff[KR];
#pragma unroll
    for (int kk = 0; kk < KR; ++kk) {
        const int k = 32 * kk + 8 * kq;
        int dx = k / CIN_G;
        const int coff = k - dx * CIN_G;
        if (dx > 2) dx = 2;
        const int col = r + dx;
        const int un = (gi * CIN_G + coff) >> 3;
        foff[kk] = col * CW + ((un ^ swz<UPS>(col)) << 3);
    }

    // ---- staging roles: thread -> (16-byte unit su of the slab, patch pixel sp + PPP it)
    const int su = tid % UPS, sp = tid / UPS;
    int s_py[NIT], s_px[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pp = sp + PPP * it;
        s_py[it] = pp / PW;
        s_px[it] = pp - s_py[it] * PW;
    }
    // batch sums: fp32 inside a tile (a fixed set of values in a fixed order), fp64 across tiles -- which tiles a workgroup gets is decided at run time,
    // and a sum of fp32 tile sums in fp64 does not depend on their order at fp32 resolution (the same argument as the fp64 atomics at the end)
    double sd[4] = {0.0, 0.0, 0.0, 0.0}, qd[4] = {0.0, 0.0, 0.0, 0.0};

    f32x4 pre[NIT][2];
    // out-of-image patch pixels load the padding vector instead (zeros, or -- fused producer transform -- the value the transform maps to 0:
    // zero padding AFTER BatchNorm + ReLU): one address select per item instead of a select per element; tiles whose patch lies inside the
    // image (workgroup-uniform, ~90 % of them) skip the bounds arithmetic altogether
    const float* const padv = XF ? p.in_pad + gpi * CW + su * 8 : g_zero_thin_x6;
    auto issue_loads = [&](int tile) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;
        if (TX6_KO & 16) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) pre[it][0] = pre[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        const float* const base = p.in + ((size_t)(b * p.H + y0 - 1) * p.W + (x0 - 1)) * CIN + gpi * CW + su * 8;     // (only dereferenced inside the image)
        const bool inside = y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= p.H && x0 + TW + 1 <= p.W;
        if (inside) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const float* src = base + (s_py[it] * p.W + s_px[it]) * CIN;
                if (sp + PPP * it >= NPATCH) src = padv;                 // the padding pixels of the last staging pass
                pre[it][0] = *reinterpret_cast<const f32x4*>(src);
                pre[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int iy = y0 - 1 + s_py[it], ix = x0 - 1 + s_px[it];
                const bool ok = sp + PPP * it < NPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const float* src = ok ? base + (s_py[it] * p.W + s_px[it]) * CIN : padv;
                pre[it][0] = *reinterpret_cast<const f32x4*>(src);
                pre[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
        }
    };

    // Tiles are CLAIMED, not dealt out: `ctr[slab]` (zeroed by the launcher in front of the launch) hands the slab's tiles to whichever workgroup
    // asks next.  A persistent grid with a static share per workgroup assumes that all of it is resident at once -- two workgroups of this kernel
    // fill a CU's register file, so on a CU that is busy with another stream's kernel they start only when that kernel ends, and then still run
    // their whole share: inside the step's hipGraph conv1_2 ran 950 us against 370 us alone, behind the 48 workgroups of the spectral-norm launch
    // on the side stream (profiles/r06_thin_x6_notes.txt; the same scheme in conv_thin_bf16.hip, whose tiles take ~8 us, LOST: 137 -> 233 us for
    // conv1_1 -- 768 workgroups' returning atomics on eight addresses, whatever the tiles per claim; it keeps its static shares).  Thread 0 claims two tiles ahead; everyone reads the claim behind the barrier that
    // ends a tile.  p.ctr == NULL (no counter pool yet under a stream capture): the static share.
    // One counter per (slab, XCD): XCD x (= workgroup id & 7, the dispatch order) claims inside its own eighth of the tile list, so neighbouring
    // tiles -- which share their halo rows / columns -- still meet in one L2 (a single global counter: conv2_2 244 -> 316 us); the other stream's
    // workgroups spread over the XCDs evenly, so the eighths end together.
    const bool xsplit = (gridDim.x & 7) == 0;
    const int xcd = xsplit ? (int)(blockIdx.x & 7) : 0;
    const int per = xsplit ? (ntiles + 7) >> 3 : ntiles;
    const int t_lo = xcd * per, t_hi = min(ntiles, t_lo + per);
    int* const ctr = p.ctr ? p.ctr + gpi * 8 + xcd : nullptr;
    auto claim = [&](int stat) -> int {
        if (!ctr) return stat;
        const int t = t_lo + atomicAdd(ctr, 1);
        return t < t_hi ? t : ntiles;
    };
    if (tid == 0) {                                      // the first two tiles (a static first tile -- no atomic in front of the first loads -- measured the same)
        s_next[0] = claim(wg0);
        s_next[1] = claim(wg0 + nwg);
    }
    __syncthreads();                                     // the scale / shift table, the first tiles
    int tile = __builtin_amdgcn_readfirstlane(s_next[0]), next = __builtin_amdgcn_readfirstlane(s_next[1]);
    if (tile < ntiles) issue_loads(tile);
#ifdef TX6_TIMING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    while (tile < ntiles) {
        // the tile after next is claimed a whole tile period ahead: the atomic's round trip (every workgroup of an XCD asks one address) is
        // never waited for -- claimed at the top of S and stored for everyone at the bottom of M
        int claimed = ntiles;
        if (tid == 0 && next < ntiles) claimed = claim(next + nwg);
        {
        // ---- transform + split + plane writes of the prefetched patch ------------------------------------------------------------------
#ifdef TX6_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP(0)                                         // waited for the prefetched loads (and this wave's stores)
#endif
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (sp + PPP * it >= NPATCH) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = pre[it][e >> 2][e & 3];
            if constexpr (XF) {
                if (!(TX6_KO & 1)) {
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(xtab + su * 8), s1 = *reinterpret_cast<const f32x4*>(xtab + su * 8 + 4);
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(xtab + CW + su * 8), h1 = *reinterpret_cast<const f32x4*>(xtab + CW + su * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e)          // (out-of-image pixels hold in_pad: the transform maps it to 0 = zero padding AFTER BatchNorm + ReLU)
                        v[e] = fmaxf(__builtin_fmaf(v[e], e < 4 ? s0[e & 3] : s1[e & 3], e < 4 ? h0[e & 3] : h1[e & 3]), 0.f);
                }
            }
            unsigned q[3][4];
            if (TX6_KO & 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) q[0][e] = q[1][e] = q[2][e] = __builtin_bit_cast(unsigned, v[2 * e]);
            } else if constexpr (F16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) split2_pair(v[2 * e], v[2 * e + 1], q[0][e], q[1][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) split3_pair(v[2 * e], v[2 * e + 1], q[0][e], q[1][e], q[2][e]);
            }
            const int at = ((sp + PPP * it) * UPS + (su ^ swz<UPS>(s_px[it]))) * 8;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4*>(planes + pl * PLANE + at) = u32x4{q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
        }
        TSTAMP(1)                                         // transform + split + plane writes
        }
        __syncthreads();                                  // the planes are complete
        {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;
        if (next < ntiles) issue_loads(next);             // the next tile's patch: in flight under this tile's MFMAs

        // ---- input-row-major K loop: patch row rho feeds output rows rho, rho - 1, rho - 2 (dy = 0, 1, 2) -------------------------------
        const int x = x0 + r;
        const int row0 = rs * RW;                       // first output row of this wave inside the tile
        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc[4], accx[4];
#pragma unroll
        for (int rho = 0; rho < RW + 2; ++rho) {
            bf16x8 F[NPL][KR];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int kk = 0; kk < KR; ++kk) {
                    if (TX6_KO & 4) F[pl][kk] = wf[pl][0][kk];
                    else F[pl][kk] = *reinterpret_cast<const bf16x8*>(planes + pl * PLANE + foff[kk] + (row0 + rho) * (PW * CW));
                }
            if (rho < RW) {
                acc[rho & 3] = f32x4{0.f, 0.f, 0.f, 0.f};   // (a bias riding in the accumulator costs accuracy: the MFMA's adder truncates, and
                accx[rho & 3] = f32x4{0.f, 0.f, 0.f, 0.f};  // the products are small against a bias of the output's size: 7.2e-7 vs 3.1e-7 on conv2_1's shape)
            }
#if TX6_FENCE
            // keep the row's MFMAs one uninterrupted block: a vector instruction that hipcc moves between two MFMAs of one accumulator chain
            // costs the chain its forwarding path (+43 cycles per intrusion, MI355X_MICROARCH.md) -- the epilogue's instructions stay outside
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int o = rho - dy;
                if (o < 0 || o >= RW) continue;
                f32x4 c = acc[o & 3], cx = accx[o & 3];
#pragma unroll
                for (int kk = 0; kk < KR; ++kk) {
                    if (TX6_KO & 2) {
                        c[kk & 3] += (float)F[0][kk][0];
                        continue;
                    }
                    if constexpr (F16) {
                        // h h' -> main; h l' + l' h' -> the scaled accumulator (first operand: weight planes)
                        cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[1][dy][kk]), __builtin_bit_cast(f16x8, F[0][kk]), cx, 0, 0, 0);
                        cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[0][dy][kk]), __builtin_bit_cast(f16x8, F[1][kk]), cx, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[0][dy][kk]), __builtin_bit_cast(f16x8, F[0][kk]), c, 0, 0, 0);
                    } else {
                        // six products, smallest first (first operand: weight planes)
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][dy][kk], F[1][kk], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][dy][kk], F[0][kk], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][dy][kk], F[2][kk], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][dy][kk], F[0][kk], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][dy][kk], F[1][kk], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][dy][kk], F[0][kk], c, 0, 0, 0);
                    }
                }
                acc[o & 3] = c;
                accx[o & 3] = cx;
            }
#if TX6_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
            // output rows complete in order: row rho - 2 after patch row rho; the epilogue takes them in PAIRS (pooling windows)
            if (rho < 3 || ((rho - 2) & 1) == 0) continue;
            const int oa = rho - 3;                     // rows oa, oa + 1 are complete
            const int ya = y0 + row0 + oa;
            float v0[4], v1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (F16) {
                    v0[c] = __builtin_fmaf(accx[oa & 3][c], 1.f / 2048.f, acc[oa & 3][c]) + bias[c];
                    v1[c] = __builtin_fmaf(accx[(oa + 1) & 3][c], 1.f / 2048.f, acc[(oa + 1) & 3][c]) + bias[c];
                } else {
                    v0[c] = acc[oa & 3][c] + bias[c];
                    v1[c] = acc[(oa + 1) & 3][c] + bias[c];
                }
            }
            auto flip = [](float t, unsigned m) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t) ^ m); };
            const int Hp = (p.H + 1) >> 1, Wp = (p.W + 1) >> 1;
            if (ya + 2 <= p.H && x0 + TW <= p.W) {
                // the row pair lies inside the image (wave-uniform, ~95 % of them): no per-pixel guards.  Batch sums of the full map in the
                // guarded path's order and arithmetic (row oa, then row oa + 1; add, fused multiply-add): identical statistics
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ssum[c] += v0[c];
                    ssq[c] = __builtin_fmaf(v0[c], v0[c], ssq[c]);
                    ssum[c] += v1[c];
                    ssq[c] = __builtin_fmaf(v1[c], v1[c], ssq[c]);
                }
                if constexpr (POOL) {
                    // rows (oa, oa + 1) of a lane and columns (r, r ^ 1) of neighbouring lanes form one window (tile origins are even): one value
                    // per window and channel, max where the consumer BatchNorm's weight is >= 0, min where it is negative (x ^ smask flips the sign)
                    float m[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) m[c] = fmaxf(flip(v0[c], smask[c]), flip(v1[c], smask[c]));
#pragma unroll
                    for (int c = 0; c < 4; ++c)          // neighbouring column = neighbouring lane: DPP quad_perm [1, 0, 3, 2]
                        m[c] = fmaxf(m[c], __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, m[c]), 0xB1, 0xF, 0xF, true)));
                    if (!(r & 1) && !(TX6_KO & 8)) {
                        float* dst = p.out + ((size_t)(b * Hp + (ya >> 1)) * Wp + (x >> 1)) * COUT + co0;
                        *reinterpret_cast<f32x4*>(dst) = f32x4{flip(m[0], smask[0]), flip(m[1], smask[1]), flip(m[2], smask[2]), flip(m[3], smask[3])};
                    }
                } else if (!(TX6_KO & 8)) {
                    float* dst = p.out + ((size_t)(b * p.H + ya) * p.W + x) * COUT + co0;
                    *reinterpret_cast<f32x4*>(dst) = f32x4{v0[0], v0[1], v0[2], v0[3]};
                    *reinterpret_cast<f32x4*>(dst + (size_t)p.W * COUT) = f32x4{v1[0], v1[1], v1[2], v1[3]};
                }
                continue;
            }
            const bool ok0 = ya < p.H && x < p.W, ok1 = ya + 1 < p.H && x < p.W;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a0 = ok0 ? v0[c] : 0.f, a1 = ok1 ? v1[c] : 0.f;
                ssum[c] += a0;
                ssq[c] = __builtin_fmaf(a0, a0, ssq[c]);
                ssum[c] += a1;
                ssq[c] = __builtin_fmaf(a1, a1, ssq[c]);
            }
            if constexpr (POOL) {
                float m[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) m[c] = ok1 ? fmaxf(flip(v0[c], smask[c]), flip(v1[c], smask[c])) : flip(v0[c], smask[c]);
                const bool pok = __builtin_amdgcn_mov_dpp((int)ok0, 0xB1, 0xF, 0xF, true) != 0;       // the neighbouring column exists
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float pm = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, m[c]), 0xB1, 0xF, 0xF, true));
                    if (pok) m[c] = fmaxf(m[c], pm);
                }
                if (ok0 && !(r & 1) && !(TX6_KO & 8)) {
                    float* dst = p.out + ((size_t)(b * Hp + (ya >> 1)) * Wp + (x >> 1)) * COUT + co0;
                    *reinterpret_cast<f32x4*>(dst) = f32x4{flip(m[0], smask[0]), flip(m[1], smask[1]), flip(m[2], smask[2]), flip(m[3], smask[3])};
                }
            } else {
                if (!(TX6_KO & 8)) {
                    float* dst = p.out + ((size_t)(b * p.H + ya) * p.W + x) * COUT + co0;
                    if (ok0) *reinterpret_cast<f32x4*>(dst) = f32x4{v0[0], v0[1], v0[2], v0[3]};
                    if (ok1) *reinterpret_cast<f32x4*>(dst + (size_t)p.W * COUT) = f32x4{v1[0], v1[1], v1[2], v1[3]};
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sd[c] += (double)ssum[c];
            qd[c] += (double)ssq[c];
        }
        TSTAMP(3)                                         // fragment reads + MFMAs + epilogues
        }
        if (tid == 0) s_next[0] = claimed;
        __syncthreads();          // every wave is done reading the planes; the tile after next is known
        TSTAMP(4)
        tile = next;
        next = __builtin_amdgcn_readfirstlane(s_next[0]);
    }
#ifdef TX6_TIMING
    if (lane == 0 && wave == 1)
        for (int k = 0; k < 5; ++k) atomicAdd(&g_tx6_timing[k], tacc[k]);
    if (tid == 0) atomicAdd(&g_tx6_timing[7], 1ull);
#endif

    if (p.ctr && tid == 0) {
        // the last workgroup to finish leaves the counters at zero for the slot's next launch (no memset node in front of every launch: a
        // 6 us node on the step's critical path, three times per step)
        __threadfence();
        if (atomicAdd(p.ctr + 32, 1) == (int)gridDim.x - 1) {
            for (int i = 0; i <= 32; ++i) p.ctr[i] = 0;
            __threadfence();
        }
    }
    if (p.stats) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double s = sd[c], q = qd[c];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s += __shfl_xor(s, o, 64);
                q += __shfl_xor(q, o, 64);
            }
            if (r == 0) {
                double* st = gssd_stats_replica(p.stats, p.stats_rep, COUT);
                unsafeAtomicAdd(st + co0 + c, s);
                unsafeAtomicAdd(st + COUT + co0 + c, q);
            }
        }
    }
}

// Tile counters: per device a pool of 1024 slots of 64 ints (32 tile counters + the count of finished workgroups), created and zeroed by the first
// launch outside a stream capture; a launch takes the next slot, and its last workgroup leaves the slot zeroed for the next user (a replayed graph
// always finds its own slot at zero).  GSSD_TX6_DYNAMIC=0, or no pool yet while capturing: NULL = static tile shares.
int* tx6_counters(hipStream_t stream) {
    static const bool off = [] { const char* e = getenv("GSSD_TX6_DYNAMIC"); return e && e[0] == '0'; }();
    if (off) return nullptr;
    constexpr int SLOTS = 1024;
    static int* pool[32] = {};
    static std::atomic<unsigned> next[32];
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32) return nullptr;
    if (!pool[dev]) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        if (cap != hipStreamCaptureStatusNone) return nullptr;
        std::lock_guard<std::mutex> lock(mu);
        if (!pool[dev]) {
            int* q = nullptr;
            if (hipMalloc(&q, SLOTS * 64 * sizeof(int)) != hipSuccess || hipMemset(q, 0, SLOTS * 64 * sizeof(int)) != hipSuccess ||
                hipDeviceSynchronize() != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            pool[dev] = q;
        }
    }
    return pool[dev] + 64 * (next[dev].fetch_add(1) % SLOTS);
}

template <int CIN_G, int COUT_G, bool XF, bool POOL, bool F16>
int launch_thin_x6_impl(const gssd_conv_desc& d, hipStream_t stream) {
    constexpr int NPL = F16 ? 2 : 3;
    ThinX6Params p;
    p.in = d.in;
    p.wgt = d.wgt;
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
    p.tiles_y = (d.H + TH - 1) / TH;
    p.tiles_x = (d.W + TW - 1) / TW;
    constexpr int CW = slab_channels(CIN_G, COUT_G);
    constexpr size_t smem = (size_t)NPL * NPAD * CW * sizeof(u16) + 2 * CW * sizeof(float) + 16;
    auto kern = conv_thin_x6_kernel<CIN_G, COUT_G, XF, POOL, F16>;
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (thin x6 conv)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    constexpr int NGI = 4 / (CW / CIN_G);
    const long long ntiles = (long long)d.B * p.tiles_y * p.tiles_x;
    long long grid = 512;                                 // two workgroups per CU
    if (ntiles * NGI < grid) grid = ntiles * NGI;
    if (NGI > 1 && grid % NGI) grid += NGI - grid % NGI;  // every slab needs its workgroups
    p.ctr = tx6_counters(stream);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT_), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// the two-plane fp16 form for launches the caller flags GSSD_CONV_F16_OK (operands inside fp16's range; never inferred from the descriptor); GSSD_X6_F16=0: bf16 planes everywhere
template <int CIN_G, int COUT_G, bool XF, bool POOL>
int launch_thin_x6(const gssd_conv_desc& d, hipStream_t stream) {
    static const bool f16_off = [] { const char* e = getenv("GSSD_X6_F16"); return e && e[0] == '0'; }();
    if (!f16_off && (d.flags & GSSD_CONV_F16_OK)) return launch_thin_x6_impl<CIN_G, COUT_G, XF, POOL, true>(d, stream);
    return launch_thin_x6_impl<CIN_G, COUT_G, XF, POOL, false>(d, stream);
}

}  // namespace

// GSSD_THIN_X6=0: the round-5 kernels (conv_thin_wino.hip / conv_wino.hip / conv_thin.hip) keep these layers (ablation / A-B)
static bool thin_x6_enabled() {
    static const bool on = [] {
        const char* e = getenv("GSSD_THIN_X6");
        return !(e && e[0] == '0');
    }();
    return on;
}

// conv3_1 (32 -> 64 channels per group, 75 x 75; round 6, late): four waves = the group's four 16-channel output tiles; GSSD_THIN_X6_CONV31=0: conv_wino_x6
// (fp16 planes only -- flagged launches: the three-plane bf16 instance of this shape spills)
static bool thin_x6_conv31() {
    static const bool on = [] {
        const char* e = getenv("GSSD_THIN_X6_CONV31");
        const char* f = getenv("GSSD_X6_F16");
        return !(e && e[0] == '0') && !(f && f[0] == '0');
    }();
    return on;
}

static bool thin_x6_shape(const gssd_conv_desc& d) {
    const int cout_g = d.groups > 0 ? d.Cout / d.groups : 0;
    return d.groups == 4 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.dil == 1 && d.in_stride == 4 * d.cin_g && d.in_ch_off == 0 &&
           d.out_mode == GSSD_OUT_NHWC && d.out_stride == d.Cout && d.out_ch_off == 0 && !d.m_per_image && !d.relu && !d.gate && !d.resid && !d.alpha &&
           !d.out2 && d.split_k <= 1 && d.wgt_row_stride == 9 * d.cin_g && d.H * d.W >= 75 * 75 && ((uintptr_t)d.out % 16) == 0 &&
           ((uintptr_t)d.in % 16) == 0 && ((uintptr_t)d.wgt % 16) == 0 && ((d.flags & ~GSSD_CONV_F16_OK) == 0 || ((d.flags & ~GSSD_CONV_F16_OK) == GSSD_CONV_POOL2 && d.pool_sign)) &&
           (long long)d.B * d.H * d.W * d.in_stride < (1ll << 31) &&
           ((d.cin_g == 16 && (cout_g == 16 || cout_g == 32)) || (d.cin_g == 32 && (cout_g == 32 || (cout_g == 64 && (d.flags & GSSD_CONV_F16_OK) && thin_x6_conv31()))));
}

extern "C" int gssd_conv_thin_x6_takes(const gssd_conv_desc* d) { return d && thin_x6_enabled() && thin_x6_shape(*d) ? 1 : 0; }

// Eligibility + dispatch; called from gssd_conv2d_nhwc_f32 (conv_igemm.hip).  Returns 1 if not eligible.
int gssd_try_conv_thin_x6(const gssd_conv_desc& d, hipStream_t stream) {
    if (!thin_x6_enabled() || !thin_x6_shape(d)) return 1;
    const int cout_g = d.Cout / d.groups;
    const bool pool = (d.flags & GSSD_CONV_POOL2) != 0;
#define TX6_CASE(CI, CO)                                                                                                              \
    if (d.cin_g == CI && cout_g == CO)                                                                                                \
        return d.in_scale ? (pool ? launch_thin_x6<CI, CO, true, true>(d, stream) : launch_thin_x6<CI, CO, true, false>(d, stream))   \
                          : (pool ? launch_thin_x6<CI, CO, false, true>(d, stream) : launch_thin_x6<CI, CO, false, false>(d, stream));
    TX6_CASE(16, 16)
    TX6_CASE(16, 32)
    TX6_CASE(32, 32)
    TX6_CASE(32, 64)
#undef TX6_CASE
    return 1;
}
