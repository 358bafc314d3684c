// fp32 fused modulated deformable 3x3 convolution on the BF16 matrix cores with fp32-equivalent products (the fp32 mode's deformable conv;
// GSSD_DCN_X6=0 runs dcn_fused.hip): the same algorithm and entry contract as dcn_fused.hip (fp32 x, fp32 offsets, fp32 blend, fp32 weights,
// fp32 output; replaces the reference's DCNv2 im2col + GEMM, layers/dcn_v2.py -> models/ssd_multiphase_custom_group.py:319-345).
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 matrix rate on gfx950.  An fp32 number is the exact sum of three bf16 numbers
// (x = h + m + l, 8 + 8 + 8 mantissa bits, each rounded to nearest); a product x y is then h h' + (h m' + m h') + (h l' + l h' + m m') + terms
// below 2^-24 |x y| -- six bf16 MFMAs with fp32 accumulation reproduce the fp32 product to the last bit or two, and still cost 3/8 of the
// fp32 instruction's matrix-pipe time.  The sampled column is blended in fp32 exactly as in dcn_fused.hip and split into its three planes when
// it is written to LDS; the weights are split once, when they are packed.
//
// Round 5's kernel (round 4's: one workgroup of four waves, weight planes in ONE LDS buffer, every chunk = [27 fragment reads, barrier]
// [12 DMA pieces + 16 corner requests in a row] [192 MFMAs, the next chunk's blend behind the last two fragment rows]: 2.12 ms.  The steps
// from there, with same-box knock-outs and per-phase timings, are in profiles/r05_dcn_x6_v2_knockout.txt and DESIGN.md section 9):
//   * 128 x 256 tile, K chunk = 32 channels of one tap, tap fastest (a pixel's 128-byte line is revisited by the next tap while it is in L2);
//   * LDS: ONE activation stage (3 planes x 128 rows x 64 B = 24 KB) + the weight planes in four half buffers (X: the wave's column tiles
//     0, 1; Y: 2, 3; two of each: 96 KB) + the sampling table of a deformable group (9 taps x 128 rows x 20 B = 23 KB) = 143 KB;
//   * the loop is rotated by half a chunk: iteration `it` runs the column tiles 2, 3 of chunk it - 1 (part A), then 0, 1 of chunk it (part B).
//     What an iteration reads (Y of chunk it - 1, X of chunk it, the stage) was written during the previous iteration; what is written during
//     it (Y of chunk it, X of chunk it + 1, the planes of chunk it + 1) goes where the previous iteration read;
//   * eight MATRIX waves (64 x 64 tiles) do nothing but fragment reads and MFMAs: column tile outer, fragment row inner; the four rows'
//     activation fragments stay in registers for a chunk and are reloaded IN PLACE behind the last column tile's MFMAs, the weight fragments
//     of the next tile are prefetched behind the current tile's MFMAs -- no fragment phase in front of a barrier;
//   * four LOADER waves own the memory side: the sampling table, the corner requests (inline assembly, two chunks ahead, counted waits),
//     blend + split, the plane writes, the weight DMA.  A wave issues in order, so a corner request or DMA piece that waits for a slot of the
//     vector memory path (one wave-wide 16-byte request per 16 cycles per CU: ~1.3 us per chunk beside 1.33 us of MFMAs) would hold back
//     the MFMAs behind it -- as it did when the eight waves did both jobs (1.85 ms; timing build: the older wave of each SIMD ran ahead
//     and waited ~550 ns per chunk at the barriers, the younger one was the critical path);
//   * two barriers per iteration: M behind the matrix waves' first group of part B (they hold chunk it's fragments -> the stage may be
//     overwritten; the loaders' Y pieces have landed -> the Y half may be read), E at the end.
// 1.76 ms per launch at the GSSD++ shape (2.12 in round 4); what bounds it now: memory side alone 1.33 ms, MFMAs alone 1.08 ms, together
// 1.76 (the two sides share LDS and the issue ports; scripts/dcn_x6_knockout.sh, scripts/dcn_x6_timing.sh).
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#ifndef X6_KO
#define X6_KO 0        // knock-outs (scripts/dcn_x6_knockout.sh): 1 no blend / split VALU, 2 no MFMAs, 4 no weight DMA, 8 no x loads, 16 no fragment reads,
#endif                 // 32 no barrier, 128 every tile's corners from the same 128 pixels (cache hits)

namespace {

constexpr int BM = 128, BN = 256, BKC = 32;           // tile; channels per K chunk: 64-byte bf16 rows
constexpr int WTM = 64, MT = WTM / 16;
constexpr int NP = 3;                                 // planes of the split
constexpr int A_STAGE = BM * BKC, B_STAGE = BN * BKC;            // u16 elements per plane

#ifndef X6_DMA_AUX
#define X6_DMA_AUX 0       // cache policy bits of the weight DMA (experiments: 2 = nt)
#endif
__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, X6_DMA_AUX);
}

__device__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }       // 64-byte rows (a four-way swizzle (row >> 2) & 3 measured 1.5 % slower)

__device__ __forceinline__ int chan_of_row(int row) {        // LDS row of the weight tile -> output channel inside the BN tile
    const int j = row >> 4, rho = row & 15;
    return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
}

// x = h + m + l, each bf16 (round to nearest even); exact to 2^-25 |x|
__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// the same split of TWO values at once, planes as packed bf16 pairs (a in the low half): one v_cvt_pk_bf16_f32 per plane and PAIR, and the
// packed result is the operand dword (the element-wise form converts every element alone and then once more to pack: 7 converts per pair)
// Round 6, the three-MFMA form (conv_thin_x6.hip) for a kernel whose accumulator count leaves no room for a second set: THREE fp16 planes per
// operand -- h = fp16(x), h6 = h / 64 (exact), l6 = fp16((x - h) * 64): x = h + l6 / 64 to 2^-24 |x| -- and the three products
// h h' + l6 h6' + h6 l6' into ONE accumulator (the scale 2^6 sits half on either operand of the two cross terms, so that neither a residual nor a
// down-scaled leading plane leaves fp16's normal range for operands between 4e-3 and 1e3; below that the ABSOLUTE error stays under 5e-10).
// Same planes, same LDS images, same DMA pieces as the bf16 form; half the matrix instructions.  GSSD_X6_F16=0: the bf16 planes.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3h_pair(const float a, const float b, unsigned& ph, unsigned& p6, unsigned& pl) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f16x2_t h = __builtin_convertvector(f32x2_t{a, b}, f16x2_t);
    const f32x2_t r = (f32x2_t{a, b} - __builtin_convertvector(h, f32x2_t)) * 64.f;
    ph = __builtin_bit_cast(unsigned, h);
    p6 = __builtin_bit_cast(unsigned, h * f16x2_t{(_Float16)0.015625f, (_Float16)0.015625f});
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}

__device__ __forceinline__ void split3_pair(const float a, const float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
    const float sa = ra - __builtin_bit_cast(float, pm << 16), sb = rb - __builtin_bit_cast(float, pm & 0xffff0000u);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sa, sb}, bf16x2));
}

// wp: [3 planes][n_tiles][chunks][BN rows in staging order][32] bf16 (slot-swizzled), chunk = (d * cpg / 32 + c32) * 9 + tap
#ifndef X6_MW
#define X6_MW 8          // matrix waves: 8 = 2 (rows) x 4 (columns) of 64 x 64; 4 = 2 x 2 of 64 x 128 (experiment)
#endif
constexpr int MW = X6_MW, LW = 4, THREADS = 64 * (MW + LW), LTHREADS = 64 * LW;
constexpr int WCOLS = MW / 2, WTN = BN / WCOLS, NT = WTN / 16;
constexpr int HB_ROWS = BN / 2, HB_PLANE = HB_ROWS * BKC, HB_ELEMS = NP * HB_PLANE;
constexpr int NTH = NT / 2, NG = NTH * MT;
constexpr int TAB_N = 9 * BM;
constexpr int LDS_BYTES = (NP * A_STAGE + 4 * HB_ELEMS) * 2 + TAB_N * 16 + TAB_N * 4;
constexpr int TPT = (TAB_N + LTHREADS - 1) / LTHREADS;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
static_assert(MT == 4 && (NT == 4 || NT == 8) && BN == 256, "written for 128 x 256 tiles, eight or four matrix waves");
#ifdef X6_TIMING
__device__ unsigned long long g_x6_timing[8];
#ifndef X6_TWAVE
#define X6_TWAVE 0
#endif
#define X6_T(k)                                                            \
    if (wave == X6_TWAVE) {                                                \
        const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); \
        t_acc[k] += t_now - t_last;                                        \
        t_last = t_now;                                                    \
    }
#else
#define X6_T(k)
#endif
#define X6_BARRIER(VM, LGKM)                                                                                   \
    if (X6_KO & 32) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" ::"n"(VM), "n"(LGKM) : "memory");           \
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"(VM), "n"(LGKM) : "memory")

template <bool F16, int KSPLIT>
__global__ __launch_bounds__(THREADS, 1) void dcn_x6_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                          const u16* __restrict__ wp, const float* __restrict__ bias,
                                                          float* __restrict__ out, int M, int H, int W, int C, int dg, int om_stride,
                                                          int Cout, int ntn, int mtiles, long long plane_elems) {
    constexpr int ksplit = KSPLIT;          // (a template parameter: as a run-time value it cost the one-part launch 7 %, 1.36 -> 1.46 ms)
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    u16* const As = smem_h;                                   // [3][BM][32]
    u16* const Bh = smem_h + NP * A_STAGE;                    // [X | Y][2][3][BN / 2][32]
    f32x4* const tabw = reinterpret_cast<f32x4*>(smem_h + NP * A_STAGE + 4 * HB_ELEMS);      // [9][BM] corner weights (x mask)
    int* const tabp = reinterpret_cast<int*>(tabw + TAB_N);                                   // [9][BM] corner position + step flags
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Split K (round 6): `ksplit` workgroups share a tile, each with dg / ksplit deformable groups of the K loop; their sums meet in the zero-filled
    // output by fp32 atomic adds.  With two parts (the bias rides on part 0) the result does not depend on the order: 0 + a is exact, a + b = b + a.
    // Why: small batches (46 row tiles at batch 4, 256 CUs) fill twice as many CUs; the launcher splits only while one round holds all parts.
    int mt, nt, kz;
    {
        const int xcd = blockIdx.x & 7;
        const int slot = (int)(blockIdx.x >> 3) / ksplit;
        kz = (int)(blockIdx.x >> 3) % ksplit;
#ifndef X6_MAP
#define X6_MAP 1         // 1: the column tiles of a row tile on the SAME XCD (consecutive slots: its x lines come out of one L2; same-box A/B
                         // 1.856 vs 1.882 - 1.894 ms); 0: on neighbouring XCDs (each XCD streams one column tile's weight planes)
#endif
        if (X6_MAP) {                                        // (a contiguous range of row tiles per XCD -- neighbours share halo rows -- measured the same)
            nt = slot % ntn;
            mt = (slot / ntn) * 8 + xcd;
        } else if (8 % ntn == 0) {
            nt = xcd % ntn;
            mt = slot * (8 / ntn) + xcd / ntn;
        } else {
            const int id = slot * 8 + xcd;
            nt = id % ntn;
            mt = id / ntn;
        }
    }
    if (mt >= mtiles) return;
    const int m0 = mt * BM;
    const int HW = H * W, cpg = C / dg, cpc = cpg / BKC;
    const int d0 = kz * (dg / ksplit);                           // this workgroup's deformable groups: d0 .. d0 + dg / ksplit - 1
    const int nchunks = (dg / ksplit) * cpc * 9;
    const u16* wslab = wp + (F16 ? 3 * plane_elems : 0) + ((size_t)nt * dg + d0) * cpc * 9 * B_STAGE;      // plane 0; plane p at + p * plane_elems (the fp16 planes lie behind the bf16 planes)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#ifdef X6_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memrealtime();
#endif

    if (wave >= MW) {
        // ================================================ the memory side: four waves ====================================================
        const int lt = tid - 64 * MW, lwave = wave - MW;
#ifndef X6_LPRIO
#define X6_LPRIO 3
#endif
        __builtin_amdgcn_s_setprio(X6_LPRIO);                // the youngest waves of the SIMD would otherwise issue last
        // gather roles: thread -> (pixel rows gp and gp + 64, 8-channel slot gq)
        const int gq = lt & 3, gp = lt >> 2;
        const int a_wr0 = gp * BKC + ((gq ^ swz(gp)) << 3);
        // sampling table of one deformable group (9 taps x BM rows)
        float t_dy[TPT], t_dx[TPT], t_ml[TPT];
        auto tab_load = [&](int d) {
#pragma unroll
            for (int u = 0; u < TPT; ++u) {
                const int e = lt + LTHREADS * u;
                const int tap = e / BM, m = m0 + (e - tap * BM);
                t_dy[u] = t_dx[u] = t_ml[u] = 0.f;
                if (e < TAB_N && m < M) {
                    const float* omp = om + (size_t)m * om_stride;
                    t_dy[u] = omp[d * 18 + 2 * tap];
                    t_dx[u] = omp[d * 18 + 2 * tap + 1];
                    t_ml[u] = omp[dg * 18 + d * 9 + tap];
                }
            }
        };
        auto tab_finish = [&]() {                            // the arithmetic of dcn_fused.hip
#pragma unroll
            for (int u = 0; u < TPT; ++u) {
                const int e = lt + LTHREADS * u;
                if (e >= TAB_N) continue;
                const int tap = e / BM, m = m0 + (e - tap * BM);
                f32x4 wv = zero4;
                int pos = 0;
                if (m < M) {
                    const int b = m / HW, pix = m - b * HW;
                    const int h = pix / W, w = pix - h * W;
                    const float msk = 1.f / (1.f + expf(-t_ml[u]));
                    const float py = (float)(h - 1 + tap / 3) + t_dy[u];
                    const float px = (float)(w - 1 + tap % 3) + t_dx[u];
                    if (py > -1.f && px > -1.f && py < (float)H && px < (float)W) {
                        const float y0f = floorf(py), x0f = floorf(px);
                        const int y0 = (int)y0f, x0 = (int)x0f;
                        const float ly = py - y0f, lx = px - x0f, hy = 1.f - ly, hx = 1.f - lx;
                        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
                        wv[0] = (y0ok && x0ok) ? hy * hx * msk : 0.f;
                        wv[1] = (y0ok && x1ok) ? hy * lx * msk : 0.f;
                        wv[2] = (y1ok && x0ok) ? ly * hx * msk : 0.f;
                        wv[3] = (y1ok && x1ok) ? ly * lx * msk : 0.f;
                        const int ya = y0ok ? y0 : 0, xa = x0ok ? x0 : 0;
                        const int yb = y1ok ? y0 + 1 : H - 1, xb = x1ok ? x0 + 1 : W - 1;
                        pos = (int)((unsigned)(b * HW + ya * W + xa) | ((unsigned)(xb - xa) << 30) | ((unsigned)(yb - ya) << 31));
                    }
                }
                tabw[e] = wv;
                tabp[e] = pos;
            }
        };
        // corner requests of one chunk: 2 cells x 4 corners x 2 halves of 4 fp32 channels per thread, by inline assembly (counted waits).
        // Contract of the inline-assembly loads (as csrc/conv_x6.hip): gv is written ONLY by corner_reqs() and read only behind
        // X6_CORNERS_WAIT, which names all sixteen registers as "+v" operands -- nothing may copy or reuse them in between (the compiler's
        // wait-count pass does not know the loads; the loop ends with s_waitcnt vmcnt(0) before the registers can die)
        f32x4 gw[2];
        f32x4 gv[2][4][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gw[j] = zero4;
#pragma unroll
            for (int k = 0; k < 4; ++k) gv[j][k][0] = gv[j][k][1] = zero4;
        }
        f32x4 gw_next[2] = {zero4, zero4};                   // the weights travel with the requests: read from the table when they are issued
        int ld_tap = 0, ld_cc = 0, ld_d = d0;                // the chunk the next corner requests are for
        int pos_next[2] = {0, 0};
        auto table_read = [&]() {                            // the table entries of the chunk requested next (LDS reads: issued early)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = ld_tap * BM + gp + 64 * j;
                gw_next[j] = tabw[e];
                pos_next[j] = tabp[e];
            }
        };
        const float* pc[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) pc[j][k] = x;
        auto corner_addr = [&]() {                           // the eight corner addresses of the chunk requested next
            const int cb = ld_d * cpg + ld_cc * BKC + gq * 8;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int pos = pos_next[j];
                if (X6_KO & 128) pos = (pos & 0xC0000000) | (gp + 64 * j);       // experiment: every tile reads the same 128 pixels (cache hits)
                const unsigned i00 = (unsigned)(pos & 0x3FFFFFFF);
                const unsigned dxb = ((unsigned)pos >> 30) & 1u, dyb = (unsigned)pos >> 31;
                const unsigned i10 = i00 + dyb * (unsigned)W;
                pc[j][0] = x + (size_t)i00 * (unsigned)C + cb;
                pc[j][1] = x + (size_t)(i00 + dxb) * (unsigned)C + cb;
                pc[j][2] = x + (size_t)i10 * (unsigned)C + cb;
                pc[j][3] = x + (size_t)(i10 + dxb) * (unsigned)C + cb;
            }
        };
        auto corner_reqs = [&](int q0, int q1) {             // requests q0 .. q1 - 1 of the 16: (cell, corner, half)
            if (X6_KO & 8) return;
#pragma unroll
            for (int q = q0; q < q1; ++q) {
                const int j = q >> 3, k = (q >> 1) & 3;
                if (q & 1) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(gv[j][k][1]) : "v"(pc[j][k]) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv[j][k][0]) : "v"(pc[j][k]) : "memory");
            }
        };
        auto corners = [&]() {
            corner_addr();
            corner_reqs(0, 16);
        };
        // all 16 requests of this wave have landed; N younger DMA pieces may still be in flight
#define X6_CORNERS_WAIT(N)                                                                                                              \
    if (!(X6_KO & 8))                                                                                                                   \
    asm volatile("s_waitcnt vmcnt(%16)"                                                                                                 \
                 : "+v"(gv[0][0][0]), "+v"(gv[0][0][1]), "+v"(gv[0][1][0]), "+v"(gv[0][1][1]), "+v"(gv[0][2][0]), "+v"(gv[0][2][1]),    \
                   "+v"(gv[0][3][0]), "+v"(gv[0][3][1]), "+v"(gv[1][0][0]), "+v"(gv[1][0][1]), "+v"(gv[1][1][0]), "+v"(gv[1][1][1]),    \
                   "+v"(gv[1][2][0]), "+v"(gv[1][2][1]), "+v"(gv[1][3][0]), "+v"(gv[1][3][1])                                           \
                 : "n"(N))
        auto advance_ld = [&]() {
            if (++ld_tap == 9) {
                ld_tap = 0;
                if (++ld_cc == cpc) {
                    ld_cc = 0;
                    ++ld_d;
                }
            }
        };
        // blend + split of the thread's 16 column values into packed plane dwords (registers); written behind the barrier
        // (F16: two planes travel through LDS -- h and l6; the matrix waves make h6 = h / 64 from h in registers: a third fewer plane writes, fragment
        // reads and weight DMA pieces on a launch bound by its memory side)
        constexpr int NPL = F16 ? 2 : NP;                    // planes in LDS / in the packed weights
        u32x2 pln[2][2][NP];                                 // [cell][channel half][plane]
        auto blend_all = [&]() {
            if (X6_KO & 1) return;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float ve[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        // the blend of dcn_fused.hip: the same four products, the same order
                        ve[e] = gv[j][0][hh][e] * gw[j][0] + gv[j][1][hh][e] * gw[j][1] + gv[j][2][hh][e] * gw[j][2] + gv[j][3][hh][e] * gw[j][3];
                    unsigned h0, m0_, l0, h1, m1, l1;
                    if constexpr (F16) {
                        split3h_pair(ve[0], ve[1], h0, m0_, l0);
                        split3h_pair(ve[2], ve[3], h1, m1, l1);
                    } else {
                        split3_pair(ve[0], ve[1], h0, m0_, l0);
                        split3_pair(ve[2], ve[3], h1, m1, l1);
                    }
                    pln[j][hh][0] = u32x2{h0, h1};
                    pln[j][hh][1] = F16 ? u32x2{l0, l1} : u32x2{m0_, m1};
                    pln[j][hh][2] = u32x2{l0, l1};
                }
        };
        auto write_planes = [&]() {
            if (X6_KO & 1) return;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        *reinterpret_cast<u32x2*>(As + pl * A_STAGE + a_wr0 + j * 64 * BKC + 4 * hh) = pln[j][hh][pl];
        };
        // weight planes of (chunk, half) -> half buffer (half, parity): 24 (F16: 16) 1-KiB pieces (plane, wave column, j & 1), six (four) per loader wave
        constexpr int DPH = 8 * NPL / LW;
        auto dma_half = [&](int chunk, int half, int parity) {
            if (X6_KO & 4) return;
            u16* dst = Bh + (half * 2 + parity) * HB_ELEMS;
            const u16* src = wslab + (size_t)chunk * B_STAGE + lane * 8;
#pragma unroll
            for (int q = 0; q < DPH; ++q) {
                const int p = q * LW + lwave;                // piece 0..23
                const int pl = p >> 3, g8 = p & 7;
                const int G = (g8 / NTH) * NT + half * NTH + (g8 % NTH);   // 16-row group of the plane's [BN][32] tile
                dma16(src + (size_t)pl * plane_elems + G * 512, dst + pl * HB_PLANE + g8 * 512);
            }
        };
        constexpr int ND = (X6_KO & 4) ? 0 : DPH, NL = (X6_KO & 8) ? 0 : 16;

        // prologue: table of group 0; corners of chunk 0 -> planes -> stage; X half of chunk 0; corners of chunk 1 requested
        tab_load(d0);
        tab_finish();
        X6_BARRIER(0, 0);                                    // P1: the table (only the loaders read it)
        table_read();
        corners();
        dma_half(0, 0, 0);
        gw[0] = gw_next[0];
        gw[1] = gw_next[1];
        X6_CORNERS_WAIT(ND);
        blend_all();
        write_planes();
        advance_ld();                                        // nchunks >= 9
        table_read();
        corners();                                           // chunk 1
        advance_ld();
        X6_BARRIER(NL, 0);                                   // P2
        int tb_next = cpc * 9, tb_d = d0 + 1;                     // first chunk of the next group, and the group
        // A loader's iteration: [DMA pieces: Y of chunk it, X of chunk it + 1] [chunk it + 1's corners have landed: blend + split into
        // registers] [first cell's corner requests of chunk it + 2] M(it) [plane writes] [second cell's requests] E(it).  The 28 requests of a
        // wave are ~1.3 us of the vector memory path per chunk (scripts/ubench/vmem_rates.hip: 52 / 39 B per clock for pieces / corner
        // segments with four waves) beside 1.33 us of MFMAs on the other waves: they are dealt out over both halves of the iteration.
        // vmcnt, in issue order: corners of chunk it + 1 -> vmcnt(12); Y pieces -> vmcnt(6 + GPRE) at M; X pieces -> vmcnt(16) at E.
#ifndef X6_GPRE
#define X6_GPRE 4        // corner requests of chunk it + 2 issued in front of M(it), the rest behind it (same-box sweep 0 / 4 / 8 / 12 / 16: 1.92 / 1.90 / 1.98 / 1.98 / 2.06 ms)
#endif
        constexpr int GPRE = X6_GPRE, NPRE = (X6_KO & 8) ? 0 : GPRE;
        for (int it = 0; it < nchunks; ++it) {
            const int par = it & 1;
            const bool make_tab = it + 2 == tb_next && it + 2 < nchunks;      // chunk it + 2 opens a group: its table is made in front of M(it)
            if (make_tab) tab_load(tb_d);
            dma_half(min(it, nchunks - 1), 1, par);          // Y of chunk it
            dma_half(min(it + 1, nchunks - 1), 0, par ^ 1);  // X of chunk it + 1
            gw[0] = gw_next[0];
            gw[1] = gw_next[1];
            if (!make_tab) table_read();                     // chunk it + 2's entries: the LDS round trip runs beside the blend
            X6_CORNERS_WAIT(2 * ND);                         // chunk it + 1's corners
            blend_all();
            if (make_tab) {
                tab_finish();                                // every loader has read the old table (its last use: in front of E(it - 1))
                tb_next += cpc * 9;
                ++tb_d;
                X6_T(0)
                X6_BARRIER(ND, 0);                           // M(it): this wave's Y pieces have landed
                X6_T(1)
                write_planes();
                table_read();                                // (the new group's table is complete behind M)
                corners();                                   // chunk it + 2
            } else {
                corner_addr();                               // chunk it + 2 (past the end: the last chunk again, never used)
                corner_reqs(0, GPRE);
                X6_T(0)
                X6_BARRIER(ND + NPRE, 0);                    // M(it)
                X6_T(1)
                write_planes();
                corner_reqs(GPRE, 16);
            }
            if (it + 3 < nchunks) advance_ld();
            X6_T(2)
            X6_BARRIER(NL, 0);                               // E(it): planes written, X pieces landed; the corner requests stay in flight
            X6_T(3)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef X6_CORNERS_WAIT
    } else {
        // ================================================ the matrix side: eight waves ===================================================
        const int wm = wave / WCOLS, wn = wave % WCOLS;
        const int r = lane & 15, kq = lane >> 4;
        const int fo = r * BKC + ((kq ^ swz(r)) << 3);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = zero4;
        bf16x8 areg[MT][NP], breg[2][NP];
        // F16: LDS planes (h, l6) -> registers [0] and [2]; [1] = h / 64 (exact: a power of two; below fp16's normal range the product it enters
        // is below 2^-24 of the leading one anyway)
        auto derive = [&](bf16x8& h6, const bf16x8& h) {
            const f16x8_t s = {(_Float16)0.015625f, (_Float16)0.015625f, (_Float16)0.015625f, (_Float16)0.015625f,
                               (_Float16)0.015625f, (_Float16)0.015625f, (_Float16)0.015625f, (_Float16)0.015625f};
            h6 = __builtin_bit_cast(bf16x8, __builtin_bit_cast(f16x8_t, h) * s);
        };
        auto a_load_row = [&](int i) {
            const u16* Ab = As + (wm * WTM + i * 16) * BKC + fo;
            if constexpr (F16) {
                if (X6_KO & 16) {
                    asm volatile("" : "=v"(areg[i][0]));
                    asm volatile("" : "=v"(areg[i][2]));
                } else {
                    areg[i][0] = *reinterpret_cast<const bf16x8*>(Ab);
                    areg[i][2] = *reinterpret_cast<const bf16x8*>(Ab + A_STAGE);
                }
                derive(areg[i][1], areg[i][0]);
                return;
            }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                if (X6_KO & 16) asm volatile("" : "=v"(areg[i][pl]));
                else areg[i][pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_STAGE);
            }
        };
        auto b_load = [&](int which, int half, int parity, int jj) {
            const u16* Bb = Bh + (half * 2 + parity) * HB_ELEMS + (wn * NTH * 16 + jj * 16) * BKC + fo;
            if constexpr (F16) {
                if (X6_KO & 16) {
                    asm volatile("" : "=v"(breg[which][0]));
                    asm volatile("" : "=v"(breg[which][2]));
                } else {
                    breg[which][0] = *reinterpret_cast<const bf16x8*>(Bb);
                    breg[which][2] = *reinterpret_cast<const bf16x8*>(Bb + HB_PLANE);
                }
                derive(breg[which][1], breg[which][0]);
                return;
            }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                if (X6_KO & 16) asm volatile("" : "=v"(breg[which][pl]));
                else breg[which][pl] = *reinterpret_cast<const bf16x8*>(Bb + pl * HB_PLANE);
            }
        };
        // the six products of one fragment pair, smallest first (first operand: weight planes, second: column planes)
        auto mma_row = [&](int i, int j, int which) {
            if (X6_KO & 2) return;
            f32x4 c = acc[i][j];
            if constexpr (F16) {
                // planes 0: h, 1: h / 64, 2: (x - h) * 64 -- l6 h6', h6 l6', h h'
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, breg[which][2]), __builtin_bit_cast(f16x8_t, areg[i][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, breg[which][1]), __builtin_bit_cast(f16x8_t, areg[i][2]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, breg[which][0]), __builtin_bit_cast(f16x8_t, areg[i][0]), c, 0, 0, 0);
            } else {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][1], areg[i][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][2], areg[i][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][1], areg[i][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][0], c, 0, 0, 0);
            }
            acc[i][j] = c;
        };
        X6_BARRIER(0, 0);                                    // P1
        X6_BARRIER(0, 0);                                    // P2: chunk 0's planes and X half are in LDS
        auto iteration = [&](int it, auto first_c, auto last_c) {
            constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
            const int par = it & 1;
            __builtin_amdgcn_sched_barrier(0);
            // ---- part A: column tiles 2, 3 of chunk it - 1 (Y half, prefetched), then chunk it's fragments in place ----
            if (!FIRST) {
#pragma unroll
                for (int jj = 0; jj < NTH; ++jj) {
                    if (jj + 1 < NTH) b_load((jj + 1) & 1, 1, par ^ 1, jj + 1);
                    else if (!LAST) b_load((jj + 1) & 1, 0, par, 0);       // X of chunk it
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        mma_row(i, NTH + jj, jj & 1);
                        if (jj + 1 == NTH && !LAST) a_load_row(i);         // chunk it's planes, in place behind the row's last use
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i) a_load_row(i);
                b_load(0, 0, par, 0);
            }
            if (!LAST) {
                // ---- part B: column tiles 0, 1 of chunk it ----
#pragma unroll
                for (int jj = 0; jj < NTH; ++jj) {
                    if (jj + 1 < NTH) b_load((jj + 1) & 1, 0, par, jj + 1);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int g = jj * MT + i;
                        mma_row(i, jj, jj & 1);
                        if (g == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            X6_T(0)
                            X6_BARRIER(0, 0);                // M(it): every wave holds chunk it's activation fragments
                            X6_T(1)
                        }
                        if (g == NG - 1) b_load(0, 1, par, 0);             // Y of chunk it for the next iteration's first column tile
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                X6_T(2)
                if (X6_KO & 32) asm volatile("" ::: "memory");
                else asm volatile("s_barrier" ::: "memory"); // E(it)
                X6_T(3)
            }
        };
        iteration(0, std::true_type{}, std::false_type{});
        for (int it = 1; it < nchunks; ++it) iteration(it, std::false_type{}, std::false_type{});
        iteration(nchunks, std::false_type{}, std::true_type{});

        // ---- epilogue: + bias, 16-byte NHWC fp32 stores (lane: pixel = lane & 15, 8 consecutive channels per tile pair) -----------
#pragma unroll
        for (int u = 0; u < NT / 2; ++u) {
            const int n0 = nt * BN + wn * WTN + 32 * u + 8 * kq;
            float bv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) bv[c] = (bias && kz == 0 && n0 + c < Cout) ? bias[n0 + c] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + wm * WTM + i * 16 + r;
                if (m >= M) continue;
                float* dst = out + (size_t)m * Cout + n0;
                if (ksplit > 1) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (n0 + c < Cout) atomicAdd(dst + c, acc[i][2 * u + (c >> 2)][c & 3] + bv[c]);
                } else if (n0 + 8 <= Cout) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{acc[i][2 * u][0] + bv[0], acc[i][2 * u][1] + bv[1], acc[i][2 * u][2] + bv[2], acc[i][2 * u][3] + bv[3]};
                    *reinterpret_cast<f32x4*>(dst + 4) =
                        f32x4{acc[i][2 * u + 1][0] + bv[4], acc[i][2 * u + 1][1] + bv[5], acc[i][2 * u + 1][2] + bv[6], acc[i][2 * u + 1][3] + bv[7]};
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (n0 + c < Cout) dst[c] = acc[i][2 * u + (c >> 2)][c & 3] + bv[c];
                }
            }
        }
    }
#ifdef X6_TIMING
    if (lane == 0 && wave == X6_TWAVE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&g_x6_timing[k], t_acc[k]);
    }
#endif
}
#undef X6_T
#undef X6_BARRIER

// OIHW fp32 [Cout][C][3][3] -> three bf16 planes, each [n_tiles][chunks][BN staging rows][32] with the slot swizzle; rows beyond Cout zero
__global__ void dcn_pack_weight_x6_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int C, int dg, long long total, int f16) {
    const int cpg = C / dg, cpc = cpg / BKC, nchunks = dg * cpc * 9;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int slot = (int)((i >> 3) & 3);
        const int row = (int)((i >> 5) % BN);
        const long long t = (i >> 5) / BN;
        const int chunk = (int)(t % nchunks);
        const int nt = (int)(t / nchunks);
        const int q = slot ^ swz(row);
        const int tap = chunk % 9, cc = chunk / 9;                                // chunk = ((group * cpc + channel block) * 9 + tap)
        const int c = cc * BKC + q * 8 + e;
        const int n = nt * BN + chan_of_row(row);
        const float v = n < Cout ? w[((size_t)n * C + c) * 9 + tap] : 0.f;
        if (f16) {                 // behind the three bf16 planes: the two fp16 planes of the three-MFMA form, h and (v - h) * 64 (h / 64 is made in registers)
            const _Float16 fh = (_Float16)v;
            wp[i + 3 * total] = __builtin_bit_cast(u16, fh);
            wp[i + 4 * total] = __builtin_bit_cast(u16, (_Float16)((v - (float)fh) * 64.f));
        }
        __bf16 h, m, l;
        split3(v, h, m, l);
        wp[i] = __builtin_bit_cast(u16, h);
        wp[i + total] = __builtin_bit_cast(u16, m);
        wp[i + 2 * total] = __builtin_bit_cast(u16, l);
    }
}

}  // namespace

#ifdef X6_TIMING
extern "C" int gssd_dcn_x6_timing_read(unsigned long long* out8) {       // debug build only: read and clear
    unsigned long long z[8] = {};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_x6_timing), sizeof(z)) != hipSuccess) return GSSD_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_x6_timing), z, sizeof(z)) != hipSuccess) return GSSD_ELAUNCH;
    return GSSD_OK;
}
#endif

// GSSD_X6_F16=0: bf16 planes and six MFMAs per product; read once -- the packed weights and the kernel instance have to agree
static bool dcn_x6_f16() {
    static const bool on = [] { const char* e = getenv("GSSD_X6_F16"); return !(e && e[0] == '0'); }();
    return on;
}

extern "C" long long gssd_dcn_packed_weight_elems_x6(int Cout, int C) {          // 16-bit elements: three bf16 planes, then two fp16 planes
    if (Cout <= 0 || C <= 0 || C % BKC != 0) return -1;
    return 5ll * ((Cout + BN - 1) / BN) * BN * 9 * C;
}

extern "C" int gssd_dcn_pack_weight_x6(const float* w_oihw, void* w_packed, int Cout, int C, int dg, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_oihw && w_packed && Cout > 0 && C > 0 && dg > 0 && C % dg == 0 && (C / dg) % BKC == 0);
    const long long total = gssd_dcn_packed_weight_elems_x6(Cout, C) / 5;
    hipLaunchKernelGGL(dcn_pack_weight_x6_kernel, dim3((int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_oihw, reinterpret_cast<u16*>(w_packed), Cout, C, dg, total, 1);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// flags & GSSD_CONV_F16_OK: x, the blended samples and the weights lie inside fp16's range (|v| < 65 504; full precision from 4e-3 up) -- the
// caller's promise, e.g. activations behind a train-mode BatchNorm: three fp16 MFMAs per product instead of six bf16 ones.  Never inferred.
extern "C" int gssd_dcn_forward_x6_ex(const float* x, const float* om, const void* w_packed, const float* bias, float* out, int B, int H,
                                      int W, int C, int dg, int om_stride, int Cout, int flags, gssd_stream_t stream) {
    const bool f16 = dcn_x6_f16() && (flags & GSSD_CONV_F16_OK);
    GSSD_CHECK_ARG(x && om && w_packed && out && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0 && Cout > 0 && Cout % 8 == 0);
    GSSD_CHECK_ARG(C % dg == 0 && (C / dg) % BKC == 0 && om_stride >= 27 * dg);
    GSSD_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)out % 16) == 0);
    const long long Mll = (long long)B * H * W;
    GSSD_CHECK_ARG(Mll < (1ll << 30) && Mll * C < (1ll << 32));          // 30-bit pixel index + 2 flag bits; 32-bit element offsets
    const int M = (int)Mll;
    const int ntn = (Cout + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    static unsigned attr_mask[4] = {0, 0, 0, 0};
    constexpr int NTHREADS = THREADS;
    // two parts per tile while ONE round of the CUs holds them all (GSSD_DCN_X6_SPLITK=0: one part).  Measured (scripts/bench_dcn_x6.py): batch 4,
    // 46 tiles: 0.79 -> 0.38 ms with four parts (not used: run-to-run bits); batch 32, 361 tiles = 1.41 rounds: two parts (2.82 rounds) are 5 % SLOWER, 1.82 against 1.72 ms --
    // the launch is bound by the request rate of the shared vector-memory / L2 path, not by the CUs of the half-empty second round.
    static const bool no_split = [] { const char* e = getenv("GSSD_DCN_X6_SPLITK"); return e && e[0] == '0'; }();
    int ksplit = 1;
    if (!no_split && X6_MAP) {
        int dev = 0, ncu = 256;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        if (dg % 2 == 0 && (long long)mtiles * ntn * 2 <= ncu) ksplit = 2;      // (two parts only: four would finish in any order, (a + b) + c != (a + c) + b)
        if (ksplit > 1 && hipMemsetAsync(out, 0, (size_t)M * Cout * sizeof(float), as_stream(stream)) != hipSuccess) {
            gssd_set_error("gssd_dcn_forward_x6: hipMemsetAsync of the output failed");
            return GSSD_ELAUNCH;
        }
    }
    const auto kernel = ksplit == 2 ? (f16 ? dcn_x6_kernel<true, 2> : dcn_x6_kernel<false, 2>) : (f16 ? dcn_x6_kernel<true, 1> : dcn_x6_kernel<false, 1>);
    const int ai = (f16 ? 1 : 0) + 2 * (ksplit - 1);
    if (gssd_attr_needed(&attr_mask[ai])) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
            hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", LDS_BYTES);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask[ai]);
    }
    int blocks;
    if (X6_MAP) {
        blocks = (mtiles + 7) / 8 * 8 * ntn * ksplit;
    } else if (8 % ntn == 0) {
        const int per = 8 / ntn;
        blocks = ((mtiles + per - 1) / per) * 8;
    } else {
        blocks = ((mtiles * ntn + 7) / 8) * 8;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(NTHREADS), LDS_BYTES, as_stream(stream), x, om, reinterpret_cast<const u16*>(w_packed), bias,
                       out, M, H, W, C, dg, om_stride, Cout, ntn, mtiles, gssd_dcn_packed_weight_elems_x6(Cout, C) / 5);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_forward_x6(const float* x, const float* om, const void* w_packed, const float* bias, float* out, int B, int H,
                                   int W, int C, int dg, int om_stride, int Cout, gssd_stream_t stream) {
    return gssd_dcn_forward_x6_ex(x, om, w_packed, bias, out, B, H, W, C, dg, om_stride, Cout, 0, stream);
}
