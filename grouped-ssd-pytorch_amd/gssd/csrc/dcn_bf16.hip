// bf16 fused modulated deformable 3x3 convolution (BASELINE.json configs[4]): the same algorithm and entry contract as
// dcn_fused.hip (sampling + contraction + bias in one kernel, no column buffer) with bf16 x / weights / output, fp32 offsets,
// fp32 bilinear blend and fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// With the matrix pipe 16x faster than in fp32 the kernel is bound by the gather and by streaming the weight slab, so the tile
// is the simple two-workgroups-per-CU shape: BM = 128 pixels x BN = 256 channels, 4 waves (2 x 2, 64 x 128 each), K chunks of 32
// channels of one tap = one MFMA k-step = 64-byte rows; per chunk a thread gathers 2 cells (4 corners x 16 B = 8 channels each).
// Operand roles are swapped and the weight rows of a 32-channel block are staged in the order of conv_bf16.hip, so a lane ends up
// with 8 consecutive output channels of one pixel: 16-byte NHWC stores.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#ifndef X6_KO
#define X6_KO 0        // knock-outs of the round-5 kernel (as csrc/dcn_x6.hip): 1 no blend, 2 no MFMAs, 4 no weight DMA, 8 no x loads, 16 no fragment reads, 32 no barrier
#endif

namespace {

constexpr int BM = 128, BN = 256, BKC = 32;          // tile; channels (bf16) per K chunk: 64 B per row
constexpr int WTM = 64, WTN = 128, MT = WTM / 16, NT = WTN / 16;
constexpr int A_STAGE = BM * BKC, B_STAGE = BN * BKC;            // u16 elements
constexpr int LDS_BYTES = 2 * (A_STAGE + B_STAGE) * 2 + 9 * BM * 16 + 9 * BM * 4;

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }       // 64-byte rows: conflict-free ds_read_b128

__device__ __forceinline__ int chan_of_row(int row) {        // LDS row of the weight tile -> output channel inside the BN tile
    const int j = row >> 4, rho = row & 15;
    return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
}

// wp: [n_tiles][chunks][BN rows in staging order][32] bf16 (slot-swizzled), chunk = (d * cpg/32 + c32) * 9 + tap
__global__ __launch_bounds__(256, 2) void dcn_bf16_kernel(const u16* __restrict__ x, const float* __restrict__ om,
                                                         const u16* __restrict__ wp, const float* __restrict__ bias,
                                                         u16* __restrict__ out, int M, int H, int W, int C, int dg, int om_stride,
                                                         int Cout, int ntn, int mtiles) {
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    u16* const As = smem_h;                                   // [2][BM][32]
    u16* const Bs = smem_h + 2 * A_STAGE;                     // [2][BN][32]
    f32x4* const setw = reinterpret_cast<f32x4*>(smem_h + 2 * (A_STAGE + B_STAGE));            // [9][BM]
    int* const setp = reinterpret_cast<int*>(smem_h + 2 * (A_STAGE + B_STAGE) + 9 * BM * 8);   // [9][BM]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
    int mt, nt;
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        if (8 % ntn == 0) {
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
    const int nchunks = dg * cpc * 9;
    const u16* wslab = wp + (size_t)nt * nchunks * B_STAGE;

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    // gather roles: thread -> (pixel row pl = (tid >> 2) + 64*j, 8-channel slot q = tid & 3)
    const int gq = tid & 3, gp = tid >> 2;
    const int a_wr0 = gp * BKC + ((gq ^ swz(gp)) << 3);
    const int fo = r * BKC + ((kq ^ swz(r)) << 3);

    auto setups = [&](int d) {
        for (int e = tid; e < 9 * BM; e += 256) {
            const int tap = e / BM, pl = e - tap * BM;
            const int m = m0 + pl;
            f32x4 wv = zero4;
            int pos = 0;
            if (m < M) {
                const int b = m / HW, pix = m - b * HW;
                const int h = pix / W, w = pix - h * W;
                const float* omp = om + (size_t)m * om_stride;
                const float dy = omp[d * 18 + 2 * tap];
                const float dx = omp[d * 18 + 2 * tap + 1];
                const float ml = omp[dg * 18 + d * 9 + tap];
                const float msk = 1.f / (1.f + expf(-ml));
                const float py = (float)(h - 1 + tap / 3) + dy;
                const float px = (float)(w - 1 + tap % 3) + dx;
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
            setw[e] = wv;
            setp[e] = pos;
        }
    };

    int ch_tap = 0, ch_c = 0, ch_d = 0;
    f32x4 gw[2];
    bf16x8 gv[2][4];

    auto gather_issue = [&]() {
        const int cb = ch_d * cpg + ch_c * BKC + gq * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = ch_tap * BM + gp + 64 * j;
            gw[j] = setw[e];
            const int pos = setp[e];
            const unsigned i00 = (unsigned)(pos & 0x3FFFFFFF);
            const unsigned dxb = ((unsigned)pos >> 30) & 1u, dyb = (unsigned)pos >> 31;
            const unsigned i10 = i00 + dyb * (unsigned)W;
            gv[j][0] = *reinterpret_cast<const bf16x8*>(x + (size_t)(i00 * (unsigned)C + cb));
            gv[j][1] = *reinterpret_cast<const bf16x8*>(x + (size_t)((i00 + dxb) * (unsigned)C + cb));
            gv[j][2] = *reinterpret_cast<const bf16x8*>(x + (size_t)(i10 * (unsigned)C + cb));
            gv[j][3] = *reinterpret_cast<const bf16x8*>(x + (size_t)((i10 + dxb) * (unsigned)C + cb));
        }
    };
    auto gather_finish = [&](int buf) {
        u16* Ad = As + buf * A_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = (__bf16)((float)gv[j][0][e] * gw[j][0] + (float)gv[j][1][e] * gw[j][1] + (float)gv[j][2][e] * gw[j][2] +
                                (float)gv[j][3][e] * gw[j][3]);
            *reinterpret_cast<bf16x8*>(Ad + a_wr0 + j * 64 * BKC) = o;
        }
    };
    auto b_issue = [&](int chunk, int buf) {
        const u16* src = wslab + (size_t)chunk * B_STAGE + lane * 8;
        u16* dst = Bs + buf * B_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = j * 4 + wave;                        // 16 pieces of 1 KiB
            dma16(src + piece * 512, dst + piece * 512);
        }
    };
    auto advance = [&]() {
        if (++ch_tap == 9) {
            ch_tap = 0;
            if (++ch_c == cpc) {
                ch_c = 0;
                ++ch_d;
            }
        }
    };

    setups(0);
    __syncthreads();
    gather_issue();
    b_issue(0, 0);
    gather_finish(0);
    advance();
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunks;
        if (more) {
            if (ch_tap == 0 && ch_c == 0) {
                setups(ch_d);
                __syncthreads();
            }
            gather_issue();
            b_issue(ch + 1, buf ^ 1);
        }
        const u16* Ab = As + buf * A_STAGE + wm * WTM * BKC + fo;
        const u16* Bb = Bs + buf * B_STAGE + wn * WTN * BKC + fo;
        bf16x8 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * BKC);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * BKC);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (more) {
            gather_finish(buf ^ 1);
            advance();
        }
        __syncthreads();
    }

    // ---- epilogue: + bias, 16-byte NHWC bf16 stores (lane: pixel = lane & 15, 8 consecutive channels per tile pair) ---------
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
        const int n0 = nt * BN + wn * WTN + 32 * u + 8 * kq;
        float bv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bv[c] = (bias && n0 + c < Cout) ? bias[n0 + c] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * WTM + i * 16 + r;
            if (m >= M) continue;
            if (n0 + 8 <= Cout) {
                bf16x8 o;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = (__bf16)(acc[i][2 * u + (c >> 2)][c & 3] + bv[c]);
                *reinterpret_cast<bf16x8*>(out + (size_t)m * Cout + n0) = o;
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (n0 + c < Cout) out[(size_t)m * Cout + n0 + c] = __builtin_bit_cast(u16, (__bf16)(acc[i][2 * u + (c >> 2)][c & 3] + bv[c]));
            }
        }
    }
}

// ---- round 5: eight matrix waves + four loader waves -----------------------------------------------------------------------------------------
namespace v3 {
// The loop and the wave roles of csrc/dcn_x6.hip's kernel (see there) with ONE bf16 plane: x is bf16 (one 16-byte request per corner and 8
// channels), the blend is fp32 and rounds to bf16 once, one MFMA per fragment pair.  With a sixth of the MFMAs the kernel is bound by the
// vector memory path alone: 32 KB of corner segments + 16 KB of weights per chunk.
#ifndef DCNB_LW
#define DCNB_LW 8         // loader waves: 8 = one cell (8 channels of one pixel row) per thread, 4 = two
#endif
constexpr int MW = 8, LW = DCNB_LW, THREADS = 64 * (MW + LW), LTHREADS = 64 * LW;
constexpr int NC = 512 / LTHREADS;                    // cells per loader thread
constexpr int WTN = 64, NT = WTN / 16;                // matrix waves: 2 (rows) x 4 (columns) of 64 x 64
constexpr int NP = 1;
constexpr int HB_ROWS = BN / 2, HB_PLANE = HB_ROWS * BKC, HB_ELEMS = NP * HB_PLANE;
constexpr int NTH = NT / 2, NG = NTH * MT;
constexpr int TAB_N = 9 * BM;
constexpr int LDS_BYTES = (NP * A_STAGE + 4 * HB_ELEMS) * 2 + TAB_N * 16 + TAB_N * 4;
constexpr int TPT = (TAB_N + LTHREADS - 1) / LTHREADS;
constexpr int DPH = 8 / LW;                           // DMA pieces per loader wave and half (a half = 8 pieces of 1 KiB)
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
static_assert(MT == 4 && NT == 4 && BN == 256 && NG == 8, "written for 128 x 256 tiles, eight MFMA waves");
#ifdef X6_TIMING
__device__ unsigned long long g_dcnb_timing[8];
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

__global__ __launch_bounds__(THREADS, 1) void dcn_bf16_v3_kernel(const u16* __restrict__ x, const float* __restrict__ om,
                                                          const u16* __restrict__ wp, const float* __restrict__ bias,
                                                          u16* __restrict__ out, int M, int H, int W, int C, int dg, int om_stride,
                                                          int Cout, int ntn, int mtiles, long long /*unused*/) {
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    u16* const As = smem_h;                                   // [3][BM][32]
    u16* const Bh = smem_h + NP * A_STAGE;                    // [X | Y][2][3][BN / 2][32]
    f32x4* const tabw = reinterpret_cast<f32x4*>(smem_h + NP * A_STAGE + 4 * HB_ELEMS);      // [9][BM] corner weights (x mask)
    int* const tabp = reinterpret_cast<int*>(tabw + TAB_N);                                   // [9][BM] corner position + step flags
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mt, nt;
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        if (8 % ntn == 0) {
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
    const int nchunks = dg * cpc * 9;
    const u16* wslab = wp + (size_t)nt * nchunks * B_STAGE;      // plane 0; plane p at + p * plane_elems
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
        // gather roles: thread -> (pixel rows gp (+ 64 with four loader waves), 8-channel slot gq)
        const int gq = lt & 3, gp = lt >> 2;
        constexpr int CSTEP = BM / NC;                       // row distance of a thread's cells
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
        // corner requests of one chunk: 2 cells x 4 corners x 2 halves of 4 fp32 channels per thread, by inline assembly (counted waits)
        f32x4 gw[2];
        bf16x8 gv[2][4];                                     // [cell][corner]: 8 bf16 channels
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            gw[j] = zero4;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[j][k][e] = (__bf16)0.f;
        }
        f32x4 gw_next[2] = {zero4, zero4};                   // the weights travel with the requests: read from the table when they are issued
        int ld_tap = 0, ld_cc = 0, ld_d = 0;                // the chunk the next corner requests are for
        int pos_next[2] = {0, 0};
        auto table_read = [&]() {                            // the table entries of the chunk requested next (LDS reads: issued early)
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int e = ld_tap * BM + gp + CSTEP * j;
                gw_next[j] = tabw[e];
                pos_next[j] = tabp[e];
            }
        };
        const u16* pc[2][4];
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) pc[j][k] = x;
        auto corner_addr = [&]() {                           // the eight corner addresses of the chunk requested next
            const int cb = ld_d * cpg + ld_cc * BKC + gq * 8;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                int pos = pos_next[j];
                if (X6_KO & 128) pos = (pos & 0xC0000000) | (gp + CSTEP * j);       // experiment: every tile reads the same 128 pixels (cache hits)
                const unsigned i00 = (unsigned)(pos & 0x3FFFFFFF);
                const unsigned dxb = ((unsigned)pos >> 30) & 1u, dyb = (unsigned)pos >> 31;
                const unsigned i10 = i00 + dyb * (unsigned)W;
                pc[j][0] = x + (size_t)i00 * (unsigned)C + cb;
                pc[j][1] = x + (size_t)(i00 + dxb) * (unsigned)C + cb;
                pc[j][2] = x + (size_t)i10 * (unsigned)C + cb;
                pc[j][3] = x + (size_t)(i10 + dxb) * (unsigned)C + cb;
            }
        };
        auto corner_reqs = [&](int q0, int q1) {             // requests q0 .. q1 - 1 of the 4 NC: (cell, corner), 16 bytes = 8 channels per lane
            if (X6_KO & 8) return;
#pragma unroll
            for (int q = q0; q < q1; ++q) {
                const int j = q >> 2, k = q & 3;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv[j][k]) : "v"(pc[j][k]) : "memory");
            }
        };
        auto corners = [&]() {
            corner_addr();
            corner_reqs(0, 4 * NC);
        };
        // all 16 requests of this wave have landed; N younger DMA pieces may still be in flight
#define X6_CORNERS_WAIT(N)                                                                                                            \
    if (!(X6_KO & 8)) {                                                                                                               \
        if (NC == 2)                                                                                                                  \
            asm volatile("s_waitcnt vmcnt(%8)"                                                                                        \
                         : "+v"(gv[0][0]), "+v"(gv[0][1]), "+v"(gv[0][2]), "+v"(gv[0][3]), "+v"(gv[1][0]), "+v"(gv[1][1]),             \
                           "+v"(gv[1][2]), "+v"(gv[1][3])                                                                              \
                         : "n"(N));                                                                                                    \
        else                                                                                                                          \
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(gv[0][0]), "+v"(gv[0][1]), "+v"(gv[0][2]), "+v"(gv[0][3]) : "n"(N));            \
    }
        auto advance_ld = [&]() {
            if (++ld_tap == 9) {
                ld_tap = 0;
                if (++ld_cc == cpc) {
                    ld_cc = 0;
                    ++ld_d;
                }
            }
        };
        // blend of the thread's 16 column values (fp32, the products and their order as in round 4's kernel), rounded to bf16 once
        bf16x8 pln[2];
        auto blend_all = [&]() {
            if (X6_KO & 1) return;
#pragma unroll
            for (int j = 0; j < NC; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    pln[j][e] = (__bf16)((float)gv[j][0][e] * gw[j][0] + (float)gv[j][1][e] * gw[j][1] + (float)gv[j][2][e] * gw[j][2] +
                                         (float)gv[j][3][e] * gw[j][3]);
        };
        auto write_planes = [&]() {
            if (X6_KO & 1) return;
#pragma unroll
            for (int j = 0; j < NC; ++j) *reinterpret_cast<bf16x8*>(As + a_wr0 + j * CSTEP * BKC) = pln[j];
        };
        // weights of (chunk, half) -> half buffer (half, parity): 8 1-KiB pieces (wave column, j & 1), two per loader wave
        auto dma_half = [&](int chunk, int half, int parity) {
            if (X6_KO & 4) return;
            u16* dst = Bh + (half * 2 + parity) * HB_ELEMS;
            const u16* src = wslab + (size_t)chunk * B_STAGE + lane * 8;
#pragma unroll
            for (int q = 0; q < DPH; ++q) {
                const int g8 = q * LW + lwave;               // piece 0..7: (wave column, j & 1)
                const int G = (g8 >> 1) * NT + half * NTH + (g8 & 1);      // 16-row group of the [BN][32] tile
                dma16(src + G * 512, dst + g8 * 512);
            }
        };
        constexpr int ND = (X6_KO & 4) ? 0 : DPH, NL = (X6_KO & 8) ? 0 : 4 * NC;

        // prologue: table of group 0; corners of chunk 0 -> planes -> stage; X half of chunk 0; corners of chunk 1 requested
        tab_load(0);
        tab_finish();
        X6_BARRIER(0, 0);                                    // P1: the table (only the loaders read it)
        table_read();
        corners();
        dma_half(0, 0, 0);
#pragma unroll
        for (int j = 0; j < NC; ++j) gw[j] = gw_next[j];
        X6_CORNERS_WAIT(ND);
        blend_all();
        write_planes();
        advance_ld();                                        // nchunks >= 9
        table_read();
        corners();                                           // chunk 1
        advance_ld();
        X6_BARRIER(NL, 0);                                   // P2
        int tb_next = cpc * 9, tb_d = 1;                     // first chunk of the next group, and the group
        // A loader's iteration: [DMA pieces: Y of chunk it, X of chunk it + 1] [chunk it + 1's corners have landed: blend + split into
        // registers] [first cell's corner requests of chunk it + 2] M(it) [plane writes] [second cell's requests] E(it).  The 28 requests of a
        // wave are ~1.3 us of the vector memory path per chunk (scripts/ubench/vmem_rates.hip: 52 / 39 B per clock for pieces / corner
        // segments with four waves) beside 1.33 us of MFMAs on the other waves: they are dealt out over both halves of the iteration.
        // vmcnt, in issue order: corners of chunk it + 1 -> vmcnt(12); Y pieces -> vmcnt(6 + GPRE) at M; X pieces -> vmcnt(16) at E.
#ifndef X6_GPRE
#define X6_GPRE (2 * NC)  // corner requests of chunk it + 2 issued in front of M(it), the rest behind it (same-box sweep 0 / 4 / 8 / 12 / 16: 1.92 / 1.90 / 1.98 / 1.98 / 2.06 ms)
#endif
        constexpr int GPRE = X6_GPRE, NPRE = (X6_KO & 8) ? 0 : GPRE;
        for (int it = 0; it < nchunks; ++it) {
            const int par = it & 1;
            const bool make_tab = it + 2 == tb_next && it + 2 < nchunks;      // chunk it + 2 opens a group: its table is made in front of M(it)
            if (make_tab) tab_load(tb_d);
            dma_half(min(it, nchunks - 1), 1, par);          // Y of chunk it
            dma_half(min(it + 1, nchunks - 1), 0, par ^ 1);  // X of chunk it + 1
#pragma unroll
            for (int j = 0; j < NC; ++j) gw[j] = gw_next[j];
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
                corner_reqs(GPRE, 4 * NC);
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
        const int wm = wave >> 2, wn = wave & 3;
        const int r = lane & 15, kq = lane >> 4;
        const int fo = r * BKC + ((kq ^ swz(r)) << 3);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = zero4;
        bf16x8 areg[MT][NP], breg[2][NP];
        auto a_load_row = [&](int i) {
            const u16* Ab = As + (wm * WTM + i * 16) * BKC + fo;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                if (X6_KO & 16) asm volatile("" : "=v"(areg[i][pl]));
                else areg[i][pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_STAGE);
            }
        };
        auto b_load = [&](int which, int half, int parity, int jj) {
            const u16* Bb = Bh + (half * 2 + parity) * HB_ELEMS + (wn * NTH * 16 + jj * 16) * BKC + fo;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                if (X6_KO & 16) asm volatile("" : "=v"(breg[which][pl]));
                else breg[which][pl] = *reinterpret_cast<const bf16x8*>(Bb + pl * HB_PLANE);
            }
        };
        auto mma_row = [&](int i, int j, int which) {
            if (X6_KO & 2) return;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][0], acc[i][j], 0, 0, 0);
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

        // ---- epilogue: + bias, 16-byte NHWC bf16 stores (lane: pixel = lane & 15, 8 consecutive channels per tile pair) ---------
#pragma unroll
        for (int u = 0; u < NT / 2; ++u) {
            const int n0 = nt * BN + wn * WTN + 32 * u + 8 * kq;
            float bv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) bv[c] = (bias && n0 + c < Cout) ? bias[n0 + c] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + wm * WTM + i * 16 + r;
                if (m >= M) continue;
                if (n0 + 8 <= Cout) {
                    bf16x8 o;
#pragma unroll
                    for (int c = 0; c < 8; ++c) o[c] = (__bf16)(acc[i][2 * u + (c >> 2)][c & 3] + bv[c]);
                    *reinterpret_cast<bf16x8*>(out + (size_t)m * Cout + n0) = o;
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (n0 + c < Cout) out[(size_t)m * Cout + n0 + c] = __builtin_bit_cast(u16, (__bf16)(acc[i][2 * u + (c >> 2)][c & 3] + bv[c]));
                }
            }
        }
    }
#ifdef X6_TIMING
    if (lane == 0 && wave == X6_TWAVE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&g_dcnb_timing[k], t_acc[k]);
    }
#endif
}
#undef X6_T
#undef X6_BARRIER
}  // namespace v3

// OIHW fp32 [Cout][C][3][3] -> bf16 [n_tiles][chunks][BN staging rows][32] with the slot swizzle; rows beyond Cout are zero
__global__ void dcn_pack_weight_bf16_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int C, int dg, long long total) {
    const int cpg = C / dg, cpc = cpg / BKC, nchunks = dg * cpc * 9;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int slot = (int)((i >> 3) & 3);
        const int row = (int)((i >> 5) % BN);
        const long long t = (i >> 5) / BN;
        const int chunk = (int)(t % nchunks);
        const int nt = (int)(t / nchunks);
        const int q = slot ^ swz(row);
        const int tap = chunk % 9, cc = chunk / 9;
        const int c = cc * BKC + q * 8 + e;
        const int n = nt * BN + chan_of_row(row);
        wp[i] = __builtin_bit_cast(u16, (__bf16)(n < Cout ? w[((size_t)n * C + c) * 9 + tap] : 0.f));
    }
}

}  // namespace

#ifdef X6_TIMING
extern "C" int gssd_dcn_bf16_timing_read(unsigned long long* out8) {       // debug build only (scripts/dcn_x6_timing.sh bf16): read and clear
    unsigned long long z[8] = {};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(v3::g_dcnb_timing), sizeof(z)) != hipSuccess) return GSSD_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(v3::g_dcnb_timing), z, sizeof(z)) != hipSuccess) return GSSD_ELAUNCH;
    return GSSD_OK;
}
#endif

extern "C" long long gssd_dcn_packed_weight_elems_bf16(int Cout, int C) {
    if (Cout <= 0 || C <= 0 || C % BKC != 0) return -1;
    return (long long)((Cout + BN - 1) / BN) * BN * 9 * C;
}

extern "C" int gssd_dcn_pack_weight_bf16(const float* w_oihw, void* w_packed, int Cout, int C, int dg, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_oihw && w_packed && Cout > 0 && C > 0 && dg > 0 && C % dg == 0 && (C / dg) % BKC == 0);
    const long long total = gssd_dcn_packed_weight_elems_bf16(Cout, C);
    hipLaunchKernelGGL(dcn_pack_weight_bf16_kernel, dim3((int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_oihw, reinterpret_cast<u16*>(w_packed), Cout, C, dg, total);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_forward_bf16(const void* x, const float* om, const void* w_packed, const float* bias, void* out, int B, int H,
                                     int W, int C, int dg, int om_stride, int Cout, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && w_packed && out && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0 && Cout > 0 && Cout % 8 == 0);
    GSSD_CHECK_ARG(C % dg == 0 && (C / dg) % BKC == 0 && om_stride >= 27 * dg);
    GSSD_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)out % 16) == 0);
    const long long Mll = (long long)B * H * W;
    GSSD_CHECK_ARG(Mll < (1ll << 30) && Mll * C < (1ll << 32));          // 30-bit pixel index + 2 flag bits; 32-bit element offsets
    const int M = (int)Mll;
    const int ntn = (Cout + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    static unsigned attr_mask = 0;
    // GSSD_DCN_BF16_V3=1: the loader / matrix-wave kernel (opt-in experiment, round 5: bit-identical output; 749 vs 779 us at B = 32,
    // 277 vs 346 at B = 11, 528 vs 487 at B = 22 -- with a sixth of the MFMAs per chunk the loaders' serial chain per chunk [table read ->
    // corner wait -> blend -> addresses -> requests] is the bound, 510-560 of 850 ns; eight loader waves with one cell per thread instead of
    // four with two: the same 746 us -- the memory side alone takes 0.68 ms; scripts/dcn_bf16_ab.sh, scripts/dcn_bf16_timing.sh)
    static const bool use_v3 = getenv("GSSD_DCN_BF16_V3") && atoi(getenv("GSSD_DCN_BF16_V3")) == 1;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(v3::dcn_bf16_v3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                v3::LDS_BYTES) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", LDS_BYTES);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask);
    }
    int blocks;
    if (8 % ntn == 0) {
        const int per = 8 / ntn;
        blocks = ((mtiles + per - 1) / per) * 8;
    } else {
        blocks = ((mtiles * ntn + 7) / 8) * 8;
    }
    if (use_v3)
        hipLaunchKernelGGL(v3::dcn_bf16_v3_kernel, dim3(blocks), dim3(v3::THREADS), v3::LDS_BYTES, as_stream(stream),
                           reinterpret_cast<const u16*>(x), om, reinterpret_cast<const u16*>(w_packed), bias, reinterpret_cast<u16*>(out), M, H, W,
                           C, dg, om_stride, Cout, ntn, mtiles, 0ll);
    else
        hipLaunchKernelGGL(dcn_bf16_kernel, dim3(blocks), dim3(256), LDS_BYTES, as_stream(stream), reinterpret_cast<const u16*>(x), om,
                           reinterpret_cast<const u16*>(w_packed), bias, reinterpret_cast<u16*>(out), M, H, W, C, dg, om_stride, Cout, ntn,
                           mtiles);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
