// Grouped 3x3 trunk convolutions with 32 .. 128 input and 64 .. 256 output channels per phase group in bf16 storage mode
// (conv3_1 .. conv5_3 and the dilated conv6 of models/ssd_multiphase_custom_group.py:434-460): "flat window" implicit GEMM.
//
// Why not the generic conv_bf16 kernel: it re-stages every input element once per tap with per-lane address arithmetic and bounds
// tests, and it applies a deferred producer BatchNorm + ReLU to every FRAGMENT (9 taps x 2 column waves per element): ~450 VALU
// instructions per 32 MFMAs -- the matrix pipe idles behind the vector pipe (0.10 .. 0.16 of the bf16 MFMA peak, round 2).
//
// Here a 256-thread workgroup owns 128 CONSECUTIVE output pixels of one image in raster order (tiles never cross images) and one
// phase group.  With stride 1 the input pixels those outputs touch are the CONTIGUOUS raster range
//       [m0 - (dil*W + dil),  m0 + 128 + (dil*W + dil))          (128 + 2 W + 2 pixels for dil = 1)
// so the whole window is staged ONCE by 16-byte LDS-DMA with one linear source address per lane (no divisions, no per-tap tests),
// and tap (dy, dx) of output pixel m is window pixel m + (dy-1)*dil*W + (dx-1)*dil: a fragment of 16 consecutive output pixels is 16
// consecutive window pixels for EVERY tap -- one add per fragment read, conflict-free under an XOR swizzle of the pixel's 16-byte
// units (swizzle applied on the DMA source side; checked against gfx950's ds_read_b128 lane groups).  What the flat window gets
// wrong are the taps that leave the image sideways (they land on the neighbouring row's pixel): those lanes are zeroed after the
// read (x < dil for dx = 0, x >= W - dil for dx = 2: four v_cndmask per fragment, six taps of nine); pixels above / below the image
// are staged as zeros.
//   * a deferred producer BatchNorm + ReLU is applied ONCE per staged element, in LDS, in fp32 with one bf16 rounding (exactly the
//     value the separate BN pass would have stored); zero-staged pixels are skipped = zero padding AFTER the transform;
//   * 128-channel groups are staged in two 64-channel halves (window <= 36 KB: two workgroups per CU);
//   * weights stream through an LDS ring of [cout tile][one tap of the staged channels] slices -- 64 k = whole 128-byte lines of the
//     K-major weight rows for 64 staged channels (8-row DMA pieces, 3 stages), 32 k for the 32-channel groups (16-row pieces, 4 stages);
//     XOR-swizzled units, rows in conv_bf16's channel order so that a lane ends up with 8 consecutive output channels of its pixel
//     = one 16-byte NHWC store.  Slice s + NSTG - 1 is in flight while slice s feeds the MFMAs -- counted `s_waitcnt vmcnt(N)` + raw
//     `s_barrier` (a __syncthreads() would drain the ring: its fence waits vmcnt(0)); one barrier per tap;
//   * 2 x 2 waves, a wave owns 64 pixels x (cout tile / 2) channels: 4 pixel fragments + 2..4 weight fragments per 8..16 MFMAs;
//   * workgroup id -> (XCD, group): both XCDs of a pair stream ONE group's weights through their L2.
// BatchNorm batch sums come from the fp32 accumulators (+ bias) before rounding, as in conv_bf16.hip.
#include <type_traits>
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#ifdef FLAT_TIMING
// debug build (scripts/flat_timing.py): wave 0 of every workgroup accumulates the shader clocks between its phase boundaries
__device__ unsigned long long g_flat_timing[8];
extern "C" int gssd_flat_timing_read(unsigned long long* out8) {
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_flat_timing), 64);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_flat_timing), z, 64);
    return 0;
}
#define FSTAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; }
#else
#define FSTAMP(k)
#endif

namespace {

__device__ __attribute__((aligned(16))) u16 g_zero_flat_h[8] = {0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// wait until at most N of this wave's vector-memory operations (LDS-DMA pieces) are outstanding, then the workgroup barrier --
// NO fence: the DMA pieces of later ring stages stay in flight across it
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    // lgkmcnt(0): this wave's fragment reads of the previous slice have RETURNED (hipcc sinks their MFMAs below the barrier), so the
    // DMA another wave issues right behind the barrier may overwrite that ring stage
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FlatParams {
    const u16* in;
    const u16* wgt;        // packed bf16 rows [Cout][wrow], k = tap * cin_g + c
    const float* bias;
    u16* out;
    double* stats;
    int stats_rep;
    const float* in_scale;
    const float* in_shift;
    int B, H, W, HW, C, Cout, dil, mtiles, ntn, npp, wrow, total;
};

// XOR applied to the 16-byte unit index of window pixel pp (conflict-free ds_read_b128 for 16 consecutive pixels at any base)
template <int UPR>
__device__ __forceinline__ int swz(int pp) {
    return UPR == 8 ? (pp & 7) : ((pp ^ (pp >> 1)) & 3);
}
// weight ring rows are 64 bytes (4 units): unit kq of row rho sits at kq ^ wswz(rho >> 2)
__device__ __forceinline__ int wswz(int t) { return (0x78 >> (2 * t)) & 3; }

// WMW = waves along the pixel dimension: 2 -> 128 pixels / 256 threads (two workgroups per CU), 4 -> 256 pixels / 512 threads (one per
// CU): the weight slices are then streamed once per 256 pixels.
// The workgroups are PERSISTENT (grid = resident workgroups, each walks its XCD pair's tiles): measured with the per-phase stamps of
// scripts/flat_timing.py a one-tile workgroup spent a third of its life waiting for its own output stores to drain before its slot
// could be reused (every workgroup of a round stores at the same moment); here the stores of tile t drain under tile t+1's window
// load, and the BatchNorm batch sums go out once per workgroup instead of once per tile.
template <int CIN_G, int COUT_T, bool XF, int NSTG, int WMW, int WNW>
__global__ __launch_bounds__(WMW * WNW * 64, WMW * WNW == 4 ? 2 : 1) void conv_flat_bf16_kernel(const FlatParams p) {
    constexpr int NW = WNW * WMW, NTHR = 64 * NW, BM = 64 * WMW;
    constexpr int CP = CIN_G < 64 ? CIN_G : 64;      // channels of the group staged at a time
    constexpr int NHALF = CIN_G / CP;
    constexpr int UPR = CP / 8, PPI = 64 / UPR;      // 16-byte units per staged pixel, pixels per DMA instruction
    constexpr int KST = CP / 32;                     // 32-k MFMA steps per tap = per ring slice
    constexpr int MT = 4, NT = COUT_T / 16 / WNW;    // 16-pixel / 16-channel tiles per wave (a wave: 64 pixels x COUT_T / WNW channels)
    constexpr int WCH = COUT_T / WNW;                // channels per wave
    constexpr int RPP = 64 / UPR;                    // weight rows per DMA piece (a slice row is CP bf16 = UPR units)
    constexpr int NWD = COUT_T / RPP / NW;           // weight DMA pieces per wave per ring stage
    static_assert(NWD >= 1, "every wave stages at least one weight piece per slice");
    constexpr int STG = COUT_T * CP;                 // elements per ring stage
    constexpr int NG = NT / 2;
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    u16* const ring = lds;
    u16* const patch = lds + NSTG * STG;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int r = lane & 15, kq = lane >> 4;

    // workgroup id -> XCD pair x owns group x: both XCDs of the pair stream ONE group's weights through their L2
    const int xcd = blockIdx.x & 7;
    const int g = xcd >> 1;
    const int W = p.W, HW = p.HW, D = p.dil;
    const int cout_g = p.Cout >> 2;
    const int halo = D * W + D;
    const int npp = p.npp;                           // BM + 2 * halo

    int wfo[KST];                                    // weight fragment offset inside a 16-row tile of a stage, per 32-k step
#pragma unroll
    for (int cs = 0; cs < KST; ++cs) wfo[cs] = r * CP + (((cs * 4 + kq) ^ (UPR == 8 ? (r & 7) : wswz(r >> 2))) << 3);
    const int pb0 = halo + wm * 64 + r;              // window index of this lane's output pixel in pixel tile 0

    // producer BatchNorm constants of this thread's eight channels (logical unit tid % UPR) of both halves: ordinary global loads, forced
    // to complete HERE -- a plain load still pending while LDS-DMA pieces are in flight makes hipcc wait vmcnt(0) at its first use
    float xs[NHALF][8], xh[NHALF][8];
    if constexpr (XF) {
#pragma unroll
        for (int h = 0; h < NHALF; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xs[h][e] = p.in_scale[g * CIN_G + h * CP + (tid % UPR) * 8 + e];
                xh[h][e] = p.in_shift[g * CIN_G + h * CP + (tid % UPR) * 8 + e];
            }
#pragma unroll
        for (int h = 0; h < NHALF; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(xs[h][e]), "v"(xh[h][e]));
    }
    float ssum[NG][8], ssq[NG][8];                   // BatchNorm batch sums of this lane's channels over all its tiles
#pragma unroll
    for (int u = 0; u < NG; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) ssum[u][c] = ssq[u][c] = 0.f;
    int nt_cur = -1;                                 // channel tile the sums / bias registers belong to (changes only when ntn > 1)
    float bias[NG][8];                               // this lane's bias values: loaded when the channel tile changes, never inside the
                                                     // epilogue (an ordinary load there waits vmcnt(0) = for the previous stores to drain)

#ifdef FLAT_TIMING
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    auto flush_stats = [&](int nt) {
        // per-channel sums of this workgroup's tiles -> fp64 atomics; the ring (idle here) carries the [WMW][COUT_T][2] partial sums
        wait_lds_barrier();
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float s = ssum[u][c], q = ssq[u][c];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s += __shfl_xor(s, o, 64);
                    q += __shfl_xor(q, o, 64);
                }
                if (r == 0) {
                    const int cl = wn * WCH + 32 * u + 8 * kq + c;
                    red[(wm * COUT_T + cl) * 2 + 0] = s;
                    red[(wm * COUT_T + cl) * 2 + 1] = q;
                }
                ssum[u][c] = ssq[u][c] = 0.f;
            }
        wait_lds_barrier();
        if (tid < COUT_T) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < WMW; ++w) {
                s += (double)red[(w * COUT_T + tid) * 2];
                q += (double)red[(w * COUT_T + tid) * 2 + 1];
            }
            const int n = g * cout_g + nt * COUT_T + tid;
            double* st = gssd_stats_replica(p.stats, p.stats_rep, p.Cout);
            unsafeAtomicAdd(st + n, s);
            unsafeAtomicAdd(st + p.Cout + n, q);
        }
        wait_lds_barrier();
    };

    // each XCD of the pair walks its own contiguous half of the group's tiles, consecutive workgroups side by side: neighbouring tiles
    // (whose windows overlap by 2 W + 2 pixels) run at the same time in the SAME L2 (alternating the two XCDs fetched every halo
    // twice: PMC 241 MB per conv3_2 launch for 185 MB algorithmic)
    const int half = (p.total + 1) >> 1;
    const int item_end = (xcd & 1) ? p.total : half;
    const int step = (int)(gridDim.x >> 3);
    for (int item = (xcd & 1) * half + (int)(blockIdx.x >> 3); item < item_end; item += step) {
        // item -> (image, pixel tile, channel tile)
        const int nt = item % p.ntn;
        const int rest = item / p.ntn;
        const int mt = rest % p.mtiles;
        const int b = rest / p.mtiles;
        const int m0 = mt * BM;
        const int qstart = m0 - halo;
        const u16* const in_g = p.in + (size_t)b * HW * p.C + g * CIN_G;
        if (nt != nt_cur) {
            if (p.stats && nt_cur >= 0) flush_stats(nt_cur);
            nt_cur = nt;
#pragma unroll
            for (int u = 0; u < NG; ++u)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    bias[u][c] = p.bias ? p.bias[g * cout_g + nt * COUT_T + wn * WCH + 32 * u + 8 * kq + c] : 0.f;
#pragma unroll
            for (int u = 0; u < NG; ++u)
#pragma unroll
                for (int c = 0; c < 8; ++c) asm volatile("" ::"v"(bias[u][c]));
        }

        // ---- weight ring: per-lane source rows.  Piece pc covers ring rows pc*RPP ..; ring row (tile j, rho) holds channel
        //      32 (j >> 1) + 8 (rho >> 2) + 4 (j & 1) + (rho & 3); unit u of a row sits at u ^ (row & 7) (128-byte rows) or
        //      u ^ wswz(rho >> 2) (64-byte rows) --------------------------------------------------------------------------------------
        const u16* wsrc[NWD];
#pragma unroll
        for (int q = 0; q < NWD; ++q) {
            const int row = (q * NW + wave) * RPP + lane / UPR;
            const int j = row >> 4, rho = row & 15;
            const int ch = 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
            const int lq = (lane % UPR) ^ (UPR == 8 ? (row & 7) : wswz(rho >> 2));
            wsrc[q] = p.wgt + (size_t)(g * cout_g + nt * COUT_T + ch) * p.wrow + lq * 8;
        }
        auto issue_w = [&](int kb, int stage) {
#pragma unroll
            for (int q = 0; q < NWD; ++q) dma16(wsrc[q] + kb, ring + stage * STG + (q * NW + wave) * 512);
        };
        // sideways taps of the flat window land on the neighbouring row: lanes at the left / right image border are zeroed per tap
        bool left[MT], right[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int x = (m0 + wm * 64 + 16 * i + r) % W;
            left[i] = x < D;
            right[i] = x >= W - D;
        }
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int h = 0; h < NHALF; ++h) {
            wait_lds_barrier();                      // every wave is done with the previous window and ring (previous half / tile)
            // ---- stage the window (one linear source range, zeros outside the image).  Instruction n covers window pixels
            //      n * PPI + lane / UPR: PPI is a multiple of the swizzle period, so a lane's unit permutation is the same for every
            //      n and its source address advances by PPI pixel vectors per instruction -- one add, one range test, one select ----
            {
                const int ninstr = (npp + PPI - 1) / PPI;
                const int pp0 = lane / UPR;
                const int lu = (lane % UPR) ^ swz<UPR>(pp0);
                const u16* src0 = in_g + ((long long)(qstart + pp0) * p.C + h * CP + lu * 8);
                const long long adv = (long long)PPI * p.C;
                for (int n = wave; n < ninstr; n += NW) {
                    const int pp = n * PPI + pp0;
                    const bool ok = pp < npp && (unsigned)(qstart + pp) < (unsigned)HW;
                    dma16(ok ? src0 + n * adv : g_zero_flat_h, patch + n * PPI * CP);
                }
            }
            // ---- ring prologue: slices (= taps) 0 .. NSTG - 2 ---------------------------------------------------------------------
#pragma unroll
            for (int s = 0; s < NSTG - 1; ++s) issue_w(s * CIN_G + h * CP, s);
            FSTAMP(0)
            wait_vm_barrier<(NSTG - 1) * NWD>();     // the window has landed (and the previous tile's stores have drained)
            FSTAMP(1)
            if constexpr (XF) {
                // producer BatchNorm + ReLU once per staged element; a thread always owns the same eight channels
                // (two window pixels per iteration: the second read is in flight under the first one's arithmetic)
                for (int pp = tid / UPR; pp < npp; pp += 2 * (NTHR / UPR)) {
                    const int pp2 = pp + NTHR / UPR;
                    const bool ok1 = (unsigned)(qstart + pp) < (unsigned)HW;
                    const bool ok2 = pp2 < npp && (unsigned)(qstart + pp2) < (unsigned)HW;
                    u16* at1 = patch + pp * CP + (((tid % UPR) ^ swz<UPR>(pp)) << 3);
                    u16* at2 = patch + (ok2 ? pp2 : pp) * CP + (((tid % UPR) ^ swz<UPR>(ok2 ? pp2 : pp)) << 3);
                    bf16x8 v1 = *reinterpret_cast<const bf16x8*>(at1), v2 = *reinterpret_cast<const bf16x8*>(at2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v1[e] = (__bf16)fmaxf((float)v1[e] * xs[h][e] + xh[h][e], 0.f);
                        v2[e] = (__bf16)fmaxf((float)v2[e] * xs[h][e] + xh[h][e], 0.f);
                    }
                    if (ok1) *reinterpret_cast<bf16x8*>(at1) = v1;
                    if (ok2) *reinterpret_cast<bf16x8*>(at2) = v2;
                }
                wait_lds_barrier();
                FSTAMP(2)
            }
            // ---- 9 taps: slice s from ring stage s % NSTG, slice s + NSTG - 1 issued behind the barrier ----------------------------
            static_for<0, 9>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int dy = s / 3, dx = s % 3;
                constexpr int infl = (8 - s) < (NSTG - 2) ? (8 - s) : (NSTG - 2);
                wait_vm_barrier<infl * NWD>();
                if constexpr (s + NSTG - 1 < 9) issue_w((s + NSTG - 1) * CIN_G + h * CP, (s + NSTG - 1) % NSTG);
                const int toff = (dy - 1) * D * W + (dx - 1) * D;
                const u16* const stage = ring + (s % NSTG) * STG + wn * NT * 16 * CP;
#pragma unroll
                for (int cs = 0; cs < KST; ++cs) {
                    bf16x8 pf[MT], wf[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(stage + j * 16 * CP + wfo[cs]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int pl = pb0 + 16 * i + toff;
                        u32x4 v = *reinterpret_cast<const u32x4*>(patch + pl * CP + (((cs * 4 + kq) ^ swz<UPR>(pl)) << 3));
                        if constexpr (dx == 0) v = left[i] ? u32x4{0u, 0u, 0u, 0u} : v;
                        if constexpr (dx == 2) v = right[i] ? u32x4{0u, 0u, 0u, 0u} : v;
                        pf[i] = __builtin_bit_cast(bf16x8, v);
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], pf[i], acc[i][j], 0, 0, 0);
                }
            });
#ifdef FLAT_TIMING
            asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[MT - 1][NT - 1][3]));      // the MFMAs have retired
#endif
            FSTAMP(3)
        }

        // ---- epilogue: + bias, batch sums, 16-byte NHWC stores (lane: pixel m0 + wm*64 + 16 i + r, 8 consecutive channels per u) ---
        u16* const out_img = p.out + (size_t)b * HW * p.Cout;
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int cl = wn * WCH + 32 * u + 8 * kq;                      // first of this lane's 8 channels inside the cout tile
            const int n0 = g * cout_g + nt * COUT_T + cl;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + wm * 64 + 16 * i + r;
                if (m >= HW) continue;
                bf16x8 hv;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float v = acc[i][2 * u + (c >> 2)][c & 3] + bias[u][c];
                    ssum[u][c] += v;
                    ssq[u][c] += v * v;
                    hv[c] = (__bf16)v;
                }
                *reinterpret_cast<bf16x8*>(out_img + (size_t)m * p.Cout + n0) = hv;
            }
        }
        FSTAMP(4)
    }
    if (p.stats && nt_cur >= 0) flush_stats(nt_cur);
#ifdef FLAT_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FSTAMP(5)
    if (tid == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&g_flat_timing[k], tacc[k]);
        atomicAdd(&g_flat_timing[7], 1ull);
    }
#endif
}

int g_force_bm = getenv("GSSD_FLAT_BM") ? atoi(getenv("GSSD_FLAT_BM")) : 0;     // 0 = choose per shape; 128 / 256 = force (ablation, tests)
thread_local bool g_dry = false;   // gssd_conv_flat_bf16_takes: run the dispatch logic without launching
thread_local int g_dry_bm = 0;

template <int CIN_G, int COUT_T, bool XF, int NSTG, int WMW, int WNW>
int launch_flat_n(const gssd_conv_desc& d, const FlatParams& p, size_t smem, hipStream_t stream) {
    if (g_dry) {
        g_dry_bm = 64 * WMW;
        return GSSD_OK;
    }
    auto kern = conv_flat_bf16_kernel<CIN_G, COUT_T, XF, NSTG, WMW, WNW>;
    static unsigned attr_mask = 0;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (flat bf16 conv)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    // persistent: one workgroup per resident slot (two 256-thread or one 512-thread workgroup per CU), never more than there are tiles;
    // GSSD_FLAT_PERSIST=0 launches one workgroup per tile instead (ablation: 15 .. 25 % slower, the slot waits for its own stores)
    static const bool persist = !(getenv("GSSD_FLAT_PERSIST") && atoi(getenv("GSSD_FLAT_PERSIST")) == 0);
    int grid = (p.total + 1) / 2 * 8;
    const int slots = WMW * WNW == 4 ? 512 : 256;
    if (persist && grid > slots) grid = slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WMW * WNW * 64), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

template <int CIN_G, int COUT_T, bool XF>
int launch_flat(const gssd_conv_desc& d, hipStream_t stream) {
    constexpr int CP = CIN_G < 64 ? CIN_G : 64, PPI = 64 / (CP / 8), RPP = 64 / (CP / 8);
    constexpr int NSTG_DEF = CP == 64 ? 3 : 4;       // ring stages: 2 slices in flight at 64 k, 3 at 32 k
    FlatParams p;
    p.in = reinterpret_cast<const u16*>(d.in);
    p.wgt = reinterpret_cast<const u16*>(d.wgt);
    p.bias = d.bias;
    p.out = reinterpret_cast<u16*>(d.out);
    p.stats = d.stats;
    p.stats_rep = d.stats_rep;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    p.HW = d.H * d.W;
    p.C = d.in_stride;
    p.Cout = d.Cout;
    p.dil = d.dil;
    p.ntn = (d.Cout / 4) / COUT_T;
    p.wrow = d.wgt_row_stride;
    const int halo = d.dil * d.W + d.dil;
    const size_t stage = (size_t)COUT_T * CP * sizeof(u16);
    auto window = [&](int bm) { return (size_t)((bm + 2 * halo + PPI - 1) / PPI * PPI) * CP * sizeof(u16); };
    auto geometry = [&](int bm) {
        p.mtiles = (p.HW + bm - 1) / bm;
        p.npp = bm + 2 * halo;
        p.total = d.B * p.mtiles * p.ntn;
    };
    // Pixels per workgroup.  128-channel tiles: 2 x 2 waves on 128 pixels (a 512-thread, 256-pixel form measured the same or slightly
    // slower: 44 / 75 / 72 us against 42 / 73 / 71 on conv4_x; kept behind GSSD_FLAT_BM=256).  64-channel tiles: 4 x 1 waves on 256
    // pixels -- every wave still owns 64 pixels x 64 channels (16 MFMAs per 8 fragment reads) -- where the map fills such tiles.
    const int force_bm = g_force_bm;
    const int t256 = (p.HW + 255) / 256;
    const bool fill256 = (double)p.HW / (256.0 * t256) >= 0.9 && (long long)d.B * t256 * p.ntn * 4 >= 1024;
    if constexpr (COUT_T == 64) {
        if ((force_bm == 256 || (force_bm == 0 && fill256)) && NSTG_DEF * stage + window(256) <= 80 * 1024) {
            geometry(256);
            return launch_flat_n<CIN_G, COUT_T, XF, NSTG_DEF, 4, 1>(d, p, NSTG_DEF * stage + window(256), stream);
        }
    } else if constexpr (COUT_T / RPP / 8 >= 1) {
        if (force_bm == 256 && NSTG_DEF * stage + window(256) <= 160 * 1024) {
            geometry(256);
            return launch_flat_n<CIN_G, COUT_T, XF, NSTG_DEF, 4, 2>(d, p, NSTG_DEF * stage + window(256), stream);
        }
    }
    geometry(128);
    // two workgroups per CU or not at all (the generic kernel takes what does not fit): wide windows (dilated conv6) run a 2-stage ring
    if (NSTG_DEF * stage + window(128) <= 80 * 1024) return launch_flat_n<CIN_G, COUT_T, XF, NSTG_DEF, 2, 2>(d, p, NSTG_DEF * stage + window(128), stream);
    if (2 * stage + window(128) <= 80 * 1024) return launch_flat_n<CIN_G, COUT_T, XF, 2, 2, 2>(d, p, 2 * stage + window(128), stream);
    return 1;
}

}  // namespace

extern "C" int gssd_conv_flat_bf16_tile(int bm) {
    const int prev = g_force_bm;
    if (bm == 0 || bm == 128 || bm == 256) g_force_bm = bm;
    return prev;
}

extern "C" int gssd_conv_flat_bf16_takes(const gssd_conv_desc* d) {
    if (!d) return 0;
    g_dry = true;
    g_dry_bm = 0;
    const int rc = gssd_try_conv_flat_bf16(*d, nullptr);
    g_dry = false;
    return rc == GSSD_OK ? g_dry_bm : 0;
}

// Eligibility + dispatch; called from gssd_conv2d_nhwc_bf16 (conv_bf16.hip).  Returns 1 if not eligible.
int gssd_try_conv_flat_bf16(const gssd_conv_desc& d, hipStream_t stream) {
    static const bool off = getenv("GSSD_NO_CONV_FLAT") != nullptr && atoi(getenv("GSSD_NO_CONV_FLAT")) != 0;   // ablation
    if (off) return 1;
    const int cout_g = d.Cout / d.groups;
    const bool shape_ok = d.groups == 4 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == d.dil && d.in_stride == 4 * d.cin_g &&
                          d.in_ch_off == 0 && d.out_mode == GSSD_OUT_NHWC && d.out_stride == d.Cout && d.out_ch_off == 0 &&
                          !d.m_per_image && !d.relu && !d.gate && !d.resid && !d.alpha && d.split_k == 1 && d.flags == 0 &&
                          d.wgt_row_stride >= 9 * d.cin_g && d.H * d.W >= 128 && d.dil < d.W && cout_g % 64 == 0 &&
                          (long long)d.H * d.W * d.in_stride < (1ll << 30);
    if (!shape_ok) return 1;
#define FLAT_CASE(CI)                                                                                                     \
    if (d.cin_g == CI) {                                                                                                  \
        if (cout_g % 128 == 0)                                                                                            \
            return d.in_scale ? launch_flat<CI, 128, true>(d, stream) : launch_flat<CI, 128, false>(d, stream);           \
        return d.in_scale ? launch_flat<CI, 64, true>(d, stream) : launch_flat<CI, 64, false>(d, stream);                 \
    }
    FLAT_CASE(32)
    FLAT_CASE(64)
    FLAT_CASE(128)
#undef FLAT_CASE
    return 1;
}
