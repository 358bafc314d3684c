// Fused modulated deformable 3x3 convolution (DCNv2, stride 1 / pad 1 / dil 1) for gfx950: sampling + contraction in ONE
// kernel -- no column buffer, no library GEMM.  Replaces what the reference gets from the external `dcn_v2` extension at
// layers/dcn_v2_custom.py:84-89 (modulated_deformable_im2col + gemm); algorithm restated in oracle/gssd_oracle.py::dcn_v2_conv.
//
//   out[m][n] = bias[n] + sum_{d, c, tap} W[n][c][tap] * mask(m,d,tap) * bilinear(x[b, :, :, c], (h-1+i+dy, w-1+j+dx))
//
// A 256-thread workgroup (4 waves, 2 x 2, ONE per CU, one wave per SIMD) owns BM = 128 output pixels x BN = 256 output channels; a
// wave owns 64 x 128 (4 x 8 MFMA tiles of v_mfma_f32_16x16x4_f32, 128 accumulator registers).  K runs in chunks of 32 input channels
// of ONE tap (128 B of every sampled pixel vector = one cache line per corner), ordered (deformable group, channel chunk, tap): the
// nine taps of a chunk re-read the same window of x, so the gather hits L1 / L2 and x leaves HBM about once.
//
//   per (pixel, tap) of the current deformable group, once per group:  4 bilinear weights (mask folded in, invalid corners
//       zeroed) + 4 clamped corner byte offsets -> LDS "setup" table (36 KB)
//   per chunk:  B tile (256 x 32 weights, pre-packed chunk-major and pre-swizzled: memory image == LDS image); A tile: every
//       thread loads the 4 corners of 4 (pixel, 4-channel) cells as float4 straight from x, blends, writes one ds_write_b128 per cell
//   fragments: ds_read_b128 per 16 x 16 tile per 16 k (lane (r, kq) holds k = 4*kq + s for MFMA step s; A and B use the same
//       permutation); 128-byte rows, slot' = slot ^ (r & 7) makes the b128 reads conflict free
//
// The main loop is ONE instruction stream of 256 "slots" per chunk: an MFMA holds the matrix pipe for 32 cycles but an issue slot for
// 4, so every staging instruction of chunk ch+1 (table reads, 16 gather loads, 8 weight loads, 12 LDS writes, the blend, the
// fragment reads, the chunk's single barrier) is placed behind a specific MFMA of chunk ch and issues in its shadow; a scheduling fence
// after every slot keeps hipcc from regrouping them (left alone it chains dependent MFMAs and clusters the memory ops).
// Measured (B = 32, 38 x 38, 1024 -> 512, dg 4; scripts/bench_dcn.py): 3.64 ms = 120 TFLOP/s (0.76 of the fp32 peak; the first
// version -- end-of-chunk barrier, staging before / after the MFMA block, two workgroups per CU -- ran 4.26 ms).  The bare MFMA stream
// of this tiling runs 3.13 ms (722 tiles on 256 CUs = 3 rounds for 2.82 rounds of work); of the 0.5 ms on top ~0.35 are the 16 gather
// loads and ~0.1 the weight tile: every VGPR-returning 64-lane x 16-byte load costs ~50 cycles of MFMA issue however it is spread over
// the slots (ablations in round 2: one per slot / one per four slots / wave-staggered slots all measure the same; an LDS read costs
// ~4, an LDS-DMA piece less than a load -- hence the weight tile by LDS-DMA).  Next step: stage the x window of a channel chunk in
// LDS once per 9 taps (10x fewer vector-memory instructions) and gather from LDS.
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <type_traits>
#include "common.h"

#ifndef DCN_B_DMA
#define DCN_B_DMA 1
#endif
#ifndef LD_EVERY
#define LD_EVERY 4
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 256, BKC = 32;          // tile, channels per K chunk (one tap, 128 B per pixel)
constexpr int WTM = 64, WTN = 128, MT = WTM / 16, NT = WTN / 16;
constexpr int A_STAGE = BM * BKC, B_STAGE = BN * BKC;            // floats
constexpr int SET_FLOATS = 9 * BM * 8;                           // per (tap, pixel): 4 weights + 4 corner byte offsets
constexpr int LDS_FLOATS = 2 * (A_STAGE + B_STAGE) + SET_FLOATS;

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// wp: [n_tiles][chunks][BN][32] (slot-swizzled rows), chunk = (d * cpg/32 + c32) * 9 + tap
//
// SK ("stream-K"): the grid is one persistent workgroup per CU.  722 tiles on 256 CUs cost 3 rounds for 2.82 rounds of work (542 tiles at
// batch 24: 3 rounds for 2.1); here every workgroup computes floor(tiles / workgroups) whole tiles of its XCD round by round, then an
// equal contiguous span of the (tile, chunk) space of the XCD's remaining tiles.  A remaining tile is cut into pieces owned by
// consecutive ranks of the SAME XCD (same L2); the piece with the tile's LAST chunks is met first (start of its owner's span), the piece
// with the first chunks last.  Partial sums travel through the tile's place in `out`: every piece but the last waits until the tile's
// flag says "the sums of chunks c_end .. are in place", adds its own, every piece but the first publishes the longer suffix, the first
// adds the bias and resets the flag.  Spans are handed out in DEcreasing workgroup id, so a waiter always has a higher id than the
// workgroup it waits for: with in-order dispatch the provider is resident or done whenever a waiter spins, whatever share of the CUs
// other streams occupy.  The order of the additions is fixed by the chain: results are deterministic.
template <bool SK>
__global__ __launch_bounds__(256, 1) void dcn_fused_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                          const float* __restrict__ wp, const float* __restrict__ bias,
                                                          float* __restrict__ out, int M, int H, int W, int C, int dg,
                                                          int om_stride, int Cout, int ntn, int mtiles, int* __restrict__ flags, float* __restrict__ sk_ws,
                                                          unsigned sk_ws_bytes, unsigned* __restrict__ sk_err) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                                   // [2][BM][32]
    float* const Bs = smem + 2 * A_STAGE;                     // [2][BN][32]
    f32x4* const setw = reinterpret_cast<f32x4*>(smem + 2 * (A_STAGE + B_STAGE));                 // [9][BM] weights
    u32x4* const seto = reinterpret_cast<u32x4*>(smem + 2 * (A_STAGE + B_STAGE) + 9 * BM * 4);    // [9][BM] corner byte offsets

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
    // XCD-aware tile order: workgroup id L runs on XCD L & 7; an XCD streams ONE weight slab (N tile) when the N-tile count
    // divides 8, and walks consecutive M tiles (shared halo rows of x stay in that L2).  Tile (slot, xcd) <-> id L = slot * 8 + xcd.
    const int xcd = blockIdx.x & 7;
    auto tile_of = [&](int slot, int& mt_, int& nt_) {
        if (8 % ntn == 0) {
            nt_ = xcd % ntn;
            mt_ = slot * (8 / ntn) + xcd / ntn;
        } else {
            const int id = slot * 8 + xcd;
            nt_ = id % ntn;
            mt_ = id / ntn;
        }
    };
    const int HW = H * W, cpg = C / dg, cpc = cpg / BKC;       // chunks of channels per deformable group
    const int nchunks = dg * cpc * 9;
    // Stream-K, phase-aligned: every workgroup first computes `sk_full` whole tiles (round i: slot i * R + rank, like the rounds of the
    // one-tile grid -- all workgroups of an XCD then stream the same weight chunk at the same time, which is what keeps the 9.4 MB weight
    // slab an L2 hit), then its share of the XCD's T remaining tiles (T < R).  Round 4: the remaining phase is aligned too.  Ranks 0 ..
    // T-1 OWN a tile each and compute its chunks [0, n_o), side by side through the same weight chunks; ranks T .. R-1 HELP: they share
    // the tails [n_o, N) of the T tiles as equal contiguous spans of that (tile, tail chunk) space, n_o = N T / R, so every workgroup
    // has the same number of chunks.  (Round 3 cut the (tile, chunk) space into R equal spans: the 32 workgroups of an XCD then sat at
    // 32 different chunk positions and each streamed the 9.4 MB slab from the Infinity Cache on its own -- PMC 2.64 GB per launch.)
    long long sk_pos = 0, sk_hi = 0;
    int sk_full = 0, sk_round = 0, sk_R = 1, sk_rank = 0, sk_T = 0, sk_no = 0;
    if (SK) {
        sk_R = gridDim.x >> 3;
        sk_rank = sk_R - 1 - (int)(blockIdx.x >> 3);
        const int nx = 8 % ntn == 0 ? (mtiles - xcd / ntn + (8 / ntn) - 1) / (8 / ntn) : (mtiles * ntn - xcd + 7) / 8;
        sk_full = (nx > 0 ? nx : 0) / sk_R;
        sk_T = (nx > 0 ? nx : 0) - sk_full * sk_R;
        if (sk_T > 0) {
            sk_no = (int)(((long long)nchunks * sk_T + sk_R / 2) / sk_R);
            sk_no = sk_no < 1 ? 1 : (sk_no > nchunks - 1 ? nchunks - 1 : sk_no);
            if (sk_rank < sk_T) {                        // owner: positions are the chunks of its own tile
                sk_pos = 0;
                sk_hi = sk_no;
            } else {                                     // helper: positions in the (tile, tail chunk) space
                const long long tot = (long long)sk_T * (nchunks - sk_no);
                const int h = sk_rank - sk_T, nh = sk_R - sk_T;
                sk_pos = tot * h / nh;
                sk_hi = tot * (h + 1) / nh;
            }
        }
        if (sk_full == 0 && sk_pos >= sk_hi) return;
    }
    int mt = 0, nt = 0, m0 = 0, c_begin = 0, c_end = nchunks, tile_id = 0;
    if (!SK) {
        tile_of((int)(blockIdx.x >> 3), mt, nt);
        if (mt >= mtiles) return;
        m0 = mt * BM;
    }
    const float* wslab = wp;
    const char* xbytes = reinterpret_cast<const char*>(x);

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // gather roles: thread -> (pixel row pl = (tid >> 3) + 32*j, 4-channel quad q = tid & 7), j = 0..3
    const int gq = tid & 7, gp = tid >> 3;
    const int a_wr0 = gp * BKC + ((gq ^ (gp & 7)) << 2);                   // (gp + 32 j) & 7 == gp & 7
    // fragment read offsets (floats) inside a 16-row block: row r, physical slot (4*ks + kq) ^ (r & 7)
    const int fo0 = r * BKC + ((kq ^ (r & 7)) << 2);
    const int fo1 = r * BKC + (((4 + kq) ^ (r & 7)) << 2);

    auto setups = [&](int d) {
        // 9 taps x 128 pixels of deformable group d (1152 entries over 256 threads)
        for (int e = tid; e < 9 * BM; e += 256) {
            const int tap = e / BM, pl = e - tap * BM;
            const int m = m0 + pl;
            f32x4 wv = zero4;
            u32x4 ov = {0u, 0u, 0u, 0u};
            if (m < M) {
                const int b = m / HW, pix = m - b * HW;
                const int h = pix / W, w = pix - h * W;
                const float* omp = om + (size_t)m * om_stride;
                const float dy = omp[d * 18 + 2 * tap];
                const float dx = omp[d * 18 + 2 * tap + 1];
                const float ml = omp[dg * 18 + d * 9 + tap];
                const float msk = 1.f / (1.f + expf(-ml));                 // torch.sigmoid (dcn_v2_custom.py:83)
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
                    const int ya = y0ok ? y0 : 0, xa = x0ok ? x0 : 0;                    // corners clamped into the image:
                    const int yb = y1ok ? y0 + 1 : H - 1, xb = x1ok ? x0 + 1 : W - 1;    // zero-weight corners read a valid pixel
                    const unsigned rowb = (unsigned)C * 4u;
                    const unsigned base = (unsigned)(b * HW) * rowb;
                    ov[0] = base + (unsigned)(ya * W + xa) * rowb;
                    ov[1] = base + (unsigned)(ya * W + xb) * rowb;
                    ov[2] = base + (unsigned)(yb * W + xa) * rowb;
                    ov[3] = base + (unsigned)(yb * W + xb) * rowb;
                }
            }
            setw[e] = wv;
            seto[e] = ov;
        }
    };

    int ch_tap = 0, ch_c = 0, ch_d = 0;          // (tap, channel chunk, deformable group) of the NEXT chunk to stage
    f32x4 gw[4];
    u32x4 go[4];
    f32x4 gv[4][4];
    f32x4 blend[4];
    unsigned cb = 0;

    auto advance = [&]() {                                         // branch-free
        const int t = ch_tap + 1;
        const bool wt = t == 9;
        ch_tap = wt ? 0 : t;
        const int c = ch_c + (wt ? 1 : 0);
        const bool wc = c == cpc;
        ch_c = wc ? 0 : c;
        ch_d += wc ? 1 : 0;
    };

    // (the segment loop follows the definition of the slot stream below)

    // One chunk = 256 MFMAs per wave (8192 matrix-pipe cycles).  Slot k = MFMA k followed by at most a few staging instructions of
    // chunk ch+1 that issue in that MFMA's 32-cycle shadow; a scheduling fence after every slot keeps hipcc from regrouping them:
    //   0..11   the second half (k 16..31) of this chunk's fragment reads         12..15 sampling-table reads of the 4 cells
    //   16..31  16 gather loads (4 corners x 4 cells, one per slot)               32..39 8 x 16-byte loads of the next weight tile
    //   120..127 its 8 LDS writes      160..191 blend of the 4 cells (>= 4000 cycles after their loads) + the 4 LDS writes of the next
    //   activation tile      192 the chunk's barrier      194..205 the first half (k 0..15) of the NEXT chunk's fragment reads
    f32x4 af[2][MT], bf[2][NT], bstage[8];
    const float* stage_src = nullptr;
    float* stage_dst = nullptr;
    float* stage_dst_w = nullptr;
    float* a_dst = nullptr;
    auto slot = [&](auto kc, auto stage_c, const float* Ab, const float* Bb, const float* Abn, const float* Bbn) {
        constexpr int K = decltype(kc)::value;
        constexpr bool stage = decltype(stage_c)::value;
        constexpr int ks = K >> 7, s = (K >> 5) & 3, i = (K >> 3) & 3, j = K & 7;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][i][s], bf[ks][j][s], acc[i][j], 0, 0, 0);
        if constexpr (K < 4) af[1][K] = *reinterpret_cast<const f32x4*>(Ab + K * 16 * BKC + fo1);
        if constexpr (K >= 4 && K < 12) bf[1][K - 4] = *reinterpret_cast<const f32x4*>(Bb + (K - 4) * 16 * BKC + fo1);
        if constexpr (stage) {
            if constexpr (K >= 12 && K < 16) {
                const int e = ch_tap * BM + gp + 32 * (K - 12);
                gw[K - 12] = setw[e];
                go[K - 12] = seto[e];
            }
            if constexpr (K >= 16 && K < 16 + 16 * LD_EVERY && (K - 16) % LD_EVERY == 0) {
                constexpr int n = (K - 16) / LD_EVERY, c = n >> 2, k = n & 3;
                gv[c][k] = *reinterpret_cast<const f32x4*>(xbytes + (go[c][k] + cb));
            }
#if DCN_B_DMA
            if constexpr (K >= 112 && K < 144 && (K & 3) == 0) dma16(stage_src + ((K - 112) >> 2) * 1024, stage_dst_w + ((K - 112) >> 2) * 1024);
#else
            if constexpr (K >= 32 && K < 40) bstage[K - 32] = *reinterpret_cast<const f32x4*>(stage_src + (K - 32) * 1024);
            if constexpr (K >= 120 && K < 128) *reinterpret_cast<f32x4*>(stage_dst + (K - 120) * 1024) = bstage[K - 120];
#endif
            if constexpr (K >= 160 && K < 192) {
                constexpr int c = (K - 160) >> 3, ph = (K - 160) & 7;
                // scalar FMAs on purpose (compiled with -fno-slp-vectorize): a packed-f32 VALU op beside an MFMA costs ~25 cycles
                // more than the two scalar ops it replaces (MI355X_MICROARCH.md, filler prices)
                if constexpr (ph == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) blend[c][e] = gv[c][0][e] * gw[c][0];
                }
                if constexpr (ph >= 1 && ph <= 3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) blend[c][e] = __builtin_fmaf(gv[c][ph][e], gw[c][ph], blend[c][e]);
                }
                if constexpr (ph == 4) *reinterpret_cast<f32x4*>(a_dst + c * 32 * BKC) = blend[c];
            }
            // the ONE barrier of the chunk sits inside the MFMA stream: both tiles of chunk ch+1 are in LDS, the k 0..15 fragment
            // registers are dead since slot 127 -> they are refilled for chunk ch+1 while the k 16..31 MFMAs of this chunk still run
            if constexpr (K == 192) __syncthreads();
            if constexpr (K >= 194 && K < 198) af[0][K - 194] = *reinterpret_cast<const f32x4*>(Abn + (K - 194) * 16 * BKC + fo0);
            if constexpr (K >= 198 && K < 206) bf[0][K - 198] = *reinterpret_cast<const f32x4*>(Bbn + (K - 198) * 16 * BKC + fo0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- segments: one tile (all chunks) per workgroup, or the pieces of this workgroup's stream-K span ---------------------------
    for (;;) {
        if (SK) {
            int slot_i;
            if (sk_round < sk_full) {
                slot_i = sk_round * sk_R + sk_rank;
                ++sk_round;
                c_begin = 0;
                c_end = nchunks;
            } else {
                if (sk_pos >= sk_hi) break;
                if (sk_rank < sk_T) {
                    slot_i = sk_full * sk_R + sk_rank;
                    c_begin = 0;
                    c_end = sk_no;
                    sk_pos = sk_hi;
                } else {
                    const int L = nchunks - sk_no;
                    const int rs = (int)(sk_pos / L);
                    slot_i = sk_full * sk_R + rs;
                    c_begin = sk_no + (int)(sk_pos - (long long)rs * L);
                    const long long left = sk_hi - sk_pos;
                    c_end = (nchunks - c_begin) < left ? nchunks : c_begin + (int)left;
                    sk_pos += c_end - c_begin;
                }
            }
            tile_of(slot_i, mt, nt);
            m0 = mt * BM;
            tile_id = slot_i * 8 + xcd;
        }
        wslab = wp + (size_t)nt * nchunks * B_STAGE;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = zero4;
        // ---- prologue: chunk c_begin, unscheduled ------------------------------------------------------------------------------
        ch_tap = c_begin % 9;
        ch_c = (c_begin / 9) % cpc;
        ch_d = c_begin / (9 * cpc);
        setups(ch_d);
        __syncthreads();
        {
            cb = (unsigned)(ch_d * cpg + ch_c * BKC + gq * 4) * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = ch_tap * BM + gp + 32 * j;
                const f32x4 w4 = setw[e];
                const u32x4 o = seto[e];
                f32x4 v = zero4;
#pragma unroll
                for (int k = 0; k < 4; ++k) v += *reinterpret_cast<const f32x4*>(xbytes + (o[k] + cb)) * w4[k];
                *reinterpret_cast<f32x4*>(As + a_wr0 + j * 32 * BKC) = v;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                dma16(wslab + (size_t)c_begin * B_STAGE + (j * 4 + wave) * 256 + lane * 4, Bs + (j * 4 + wave) * 256);
        }
        advance();
        __syncthreads();
        // fragments k 0..15 of the first chunk
        {
            const float* Ab = As + wm * WTM * BKC;
            const float* Bb = Bs + wn * WTN * BKC;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(Ab + i * 16 * BKC + fo0);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(Bb + j * 16 * BKC + fo0);
        }
        for (int ch = c_begin; ch < c_end - 1; ++ch) {
            const int buf = (ch - c_begin) & 1;
            if (ch_tap == 0 && ch_c == 0) {          // the next chunk opens a new deformable group: new sampling table
                setups(ch_d);
                __syncthreads();
            }
            cb = (unsigned)(ch_d * cpg + ch_c * BKC + gq * 4) * 4u;
            stage_src = wslab + (size_t)(ch + 1) * B_STAGE + wave * 256 + lane * 4;
            stage_dst = Bs + (buf ^ 1) * B_STAGE + wave * 256 + lane * 4;
            stage_dst_w = Bs + (buf ^ 1) * B_STAGE + wave * 256;
            a_dst = As + (buf ^ 1) * A_STAGE + a_wr0;
            const float* Ab = As + buf * A_STAGE + wm * WTM * BKC;
            const float* Bb = Bs + buf * B_STAGE + wn * WTN * BKC;
            const float* Abn = As + (buf ^ 1) * A_STAGE + wm * WTM * BKC;
            const float* Bbn = Bs + (buf ^ 1) * B_STAGE + wn * WTN * BKC;
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 256>([&](auto kc) { slot(kc, std::true_type{}, Ab, Bb, Abn, Bbn); });
            advance();
        }
        {
            const int buf = (c_end - 1 - c_begin) & 1;
            const float* Ab = As + buf * A_STAGE + wm * WTM * BKC;
            const float* Bb = Bs + buf * B_STAGE + wn * WTN * BKC;
            static_for<0, 256>([&](auto kc) { slot(kc, std::false_type{}, Ab, Bb, Ab, Bb); });
        }

        // ---- epilogue: + bias, NHWC store; a stream-K piece that owns only part of the tile's chunks hands over / picks up partial sums
        // A tile cut into pieces (consecutive ranks of one XCD, the piece with the LAST chunks is met first): every piece but the last
        // waits until the tile's flag says "the sums of chunks c_end .. nchunks-1 are in `out`" (flag = their count), adds its own, and
        // every piece but the first publishes the longer suffix the same way; the first piece adds the bias and resets the flag.
        const bool head = c_begin == 0, tail = c_end == nchunks;
        // The hand-over is PLACEMENT INDEPENDENT (cdna_hip_programming.md, Guideline 16: "a wrong placement guess is slower, not wrong"):
        // partial sums travel through a 128 KB slab per cut tile in ACCUMULATOR layout ([MFMA tile][thread] float4: every instruction
        // moves 1 KB), written with 16-byte sc1 (write-through, agent-scope) stores and read with sc1 loads; the provider drains its
        // stores (vmcnt(0)) and joins its waves before ONE lane publishes the flag with a relaxed agent-scope store, the consumer polls
        // the flag relaxed.  Round 3 passed the sums through `out` with plain stores and relied on workgroup id & 7 == XCD (one L2 for
        // both sides); round 4's run-time check of HW_REG_XCC_ID showed launches whose workgroups sit elsewhere (other streams busy),
        // so that form could read stale sums.  Keeping a tile's pieces on one XCD is still the fast case (same-L2 reads).
        const unsigned slab = ((unsigned)(tile_id - sk_full * sk_R * 8)) * (unsigned)(MT * NT * 256 * 16);   // remaining tile rs of XCD x: rs * 8 + x
        const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(sk_ws, 0, SK ? (int)sk_ws_bytes : 0, 0x00020000);
        if (SK && !tail) {
            if (tid == 0) {
                // bounded: a provider that never arrives (aborted launch) costs ~1 s and an error, not a hang
                int spins = 0;
                while (__hip_atomic_load(flags + tile_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nchunks - c_end) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1 << 21)) {
                        __hip_atomic_fetch_or(sk_err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wsr, tid * 16 + (i * NT + j) * 4096, slab, 16 /* sc1 */);
                    acc[i][j] += __builtin_bit_cast(f32x4, v);
                }
        }
        if (SK && !head) {                           // publish the sums of chunks c_begin .. nchunks-1
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), wsr, tid * 16 + (i * NT + j) * 4096, slab, 16 /* sc1 */);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + tile_id, nchunks - c_begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = nt * BN + wn * WTN + j * 16 + r;
                if (n >= Cout) continue;
                const float bv = bias ? bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int mb = m0 + wm * WTM + i * 16 + kq * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (mb + e < M) out[(size_t)(mb + e) * Cout + n] = acc[i][j][e] + bv;
                }
            }
            if (SK && !tail && tid == 0) __hip_atomic_store(flags + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!SK) break;
        __syncthreads();                             // LDS (tiles, sampling table) is free for the next piece
    }
}

// OIHW [Cout][C][3][3] -> [n_tiles][chunks][BN][32] with the slot swizzle; rows beyond Cout are zero
__global__ void dcn_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int C, int dg, long long total) {
    const int cpg = C / dg, cpc = cpg / BKC, nchunks = dg * cpc * 9;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const int slot = (int)((i >> 2) & 7);
        const int nl = (int)((i >> 5) % BN);
        const long long t = (i >> 5) / BN;
        const int chunk = (int)(t % nchunks);
        const int nt = (int)(t / nchunks);
        const int q = slot ^ (nl & 7);                // logical quad stored in this physical slot
        const int tap = chunk % 9, cc = chunk / 9;    // cc = d * cpc + c32
        const int c = cc * BKC + q * 4 + e;
        const int n = nt * BN + nl;
        wp[i] = n < Cout ? w[((size_t)n * C + c) * 9 + tap] : 0.f;
    }
}

}  // namespace

extern "C" long long gssd_dcn_packed_weight_elems(int Cout, int C) {
    if (Cout <= 0 || C <= 0 || C % BKC != 0) return -1;
    return (long long)((Cout + BN - 1) / BN) * BN * 9 * C;
}

extern "C" int gssd_dcn_pack_weight_f32(const float* w_oihw, float* w_packed, int Cout, int C, int dg, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_oihw && w_packed && Cout > 0 && C > 0 && dg > 0 && C % dg == 0 && (C / dg) % BKC == 0);
    const long long total = gssd_dcn_packed_weight_elems(Cout, C);
    hipLaunchKernelGGL(dcn_pack_weight_kernel, dim3((int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_oihw, w_packed, Cout, C, dg, total);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

static int g_dcn_sk_force = -1;     // -1: GSSD_DCN_STREAMK decides (default on), 0 / 1: forced (tests, ablation)
extern "C" int gssd_dcn_streamk(int mode) {
    const int prev = g_dcn_sk_force;
    g_dcn_sk_force = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}

namespace {

// ---- host state of the stream-K form, per device ---------------------------------------------------------------------------------
// Per OUTPUT buffer (two launches that may be in flight together -- two plans, two captured graphs, several streams -- write different
// outputs, so they never share state): one flag int per tile and a 128 KB partial-sum slab per cut tile (at most one per
// workgroup: CUs x 128 KB = 32 MB).  Regions are created on a launch outside stream capture (they allocate); a launch that meets an
// unknown output while capturing takes the one-tile form.  The hand-over itself is placement independent (kernel epilogue); the XCC
// probe below is information for gssd_dcn_streamk_status (is the fast same-L2 case the common one on this device?).
constexpr int SK_MAX_REGIONS = 64;
constexpr size_t SK_SLAB_BYTES = (size_t)MT * NT * 256 * 16;
struct SkRegion { const void* out; int* flags; float* ws; int ints; };
struct SkDev {
    int cus = 0;                 // 0: not looked at, -1: unusable
    int status = 0;              // GSSD_DCN_SK_* bits
    unsigned xcc_map = 0;        // probed XCC id per residue of workgroup id & 7, 4 bits each
    unsigned* err_host = nullptr;   // pinned + mapped; bit 1: a wait timed out
    unsigned* err_dev = nullptr;
    SkRegion regions[SK_MAX_REGIONS];
    int nregions = 0;
    SkRegion retired[SK_MAX_REGIONS];   // replaced inside a launch call; freed by gssd_dcn_streamk_release
    int nretired = 0;
};
SkDev g_sk[16];
std::mutex g_sk_mu;              // (first launches may come from two host threads: forward on the caller's, backward on autograd's)

__global__ void sk_probe_kernel(unsigned* __restrict__ xcc_of_block) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = xcc;
    // stay resident long enough for the whole grid to be dispatched side by side (placement of a grid that fills the chip)
    __builtin_amdgcn_s_sleep(127);
}

// one-time set-up outside stream capture: CU count, placement probe, flag pool (zeroed and SYNCHRONISED before first use), error word
void sk_init(SkDev& s, int dev) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    s.cus = -1;
    if (cus < 8 || cus % 8 != 0 || cus > 1024) { s.status |= GSSD_DCN_SK_UNSUPPORTED; return; }
    unsigned* probe = nullptr;
    unsigned host[1024];
    bool ok = hipMalloc(&probe, cus * sizeof(unsigned)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(sk_probe_kernel, dim3(cus), dim3(256), 0, nullptr, probe);
        ok = hipMemcpy(host, probe, cus * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(probe);
    }
    if (!ok) { s.status |= GSSD_DCN_SK_UNSUPPORTED; (void)hipGetLastError(); return; }
    unsigned map = 0, seen = 0;
    bool good = true;
    for (int r = 0; r < 8 && good; ++r) {
        const unsigned id = host[r];
        for (int b = r; b < cus; b += 8) good = good && host[b] == id;
        good = good && id < 16 && !(seen & (1u << id));
        seen |= 1u << id;
        map |= (id & 15u) << (4 * r);
    }
    s.xcc_map = map;
    if (!good) s.status |= GSSD_DCN_SK_MAPPING;          // information only: pieces of a tile may sit behind different L2s (slower, not wrong)
    unsigned* eh = nullptr;
    void* ed = nullptr;
    if (hipHostMalloc((void**)&eh, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        s.status |= GSSD_DCN_SK_UNSUPPORTED;
        return;
    }
    *eh = 0;
    if (hipHostGetDevicePointer(&ed, eh, 0) != hipSuccess) {
        (void)hipGetLastError();
        s.status |= GSSD_DCN_SK_UNSUPPORTED;
        return;
    }
    s.err_host = eh;
    s.err_dev = reinterpret_cast<unsigned*>(ed);
    s.cus = cus;
}

// flags + slabs of the launch writing `out`; created (zeroed, SYNCHRONISED before any stream's first use) when `may_alloc`
SkRegion* sk_region(SkDev& s, const void* out, int ints, bool may_alloc) {
    for (int i = 0; i < s.nregions; ++i)
        if (s.regions[i].out == out && s.regions[i].ints >= ints) return &s.regions[i];
    if (!may_alloc) return nullptr;
    // a smaller region of the same output buffer (the address served a smaller launch before) is replaced, not kept beside the new one
    // -- RETIRED, not freed: this runs inside a launch call (no device synchronisation, no hipFree there: another thread may hold a capture
    // open); gssd_dcn_streamk_release frees the retired regions
    for (int i = 0; i < s.nregions; ++i)
        if (s.regions[i].out == out) {
            if (s.nretired == SK_MAX_REGIONS) return nullptr;
            s.retired[s.nretired++] = s.regions[i];
            s.regions[i] = s.regions[--s.nregions];
            break;
        }
    if (s.nregions == SK_MAX_REGIONS) return nullptr;
    int* fl = nullptr;
    float* ws = nullptr;
    const size_t fbytes = (size_t)((ints + 31) & ~31) * sizeof(int);
    if (hipMalloc(&fl, fbytes) != hipSuccess || hipMalloc(&ws, (size_t)s.cus * SK_SLAB_BYTES) != hipSuccess ||
        hipMemset(fl, 0, fbytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        if (fl) (void)hipFree(fl);
        if (ws) (void)hipFree(ws);
        return nullptr;
    }
    s.regions[s.nregions] = SkRegion{out, fl, ws, (ints + 31) & ~31};
    return &s.regions[s.nregions++];
}

}  // namespace

extern "C" int gssd_dcn_streamk_status(unsigned* xcc_map) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16) return GSSD_DCN_SK_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_sk_mu);
    SkDev& s = g_sk[dev];
    if (s.cus == 0) sk_init(s, dev);                     // (not to be called while a stream of this thread is capturing)
    if (s.err_host) {
        const unsigned e = __atomic_load_n(s.err_host, __ATOMIC_ACQUIRE);
        if (e & 2u) s.status |= GSSD_DCN_SK_TIMEOUT;
    }
    if (xcc_map) *xcc_map = s.xcc_map;
    return s.status;
}

extern "C" int gssd_dcn_streamk_reset(gssd_stream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16) return GSSD_OK;
    std::lock_guard<std::mutex> lock(g_sk_mu);
    SkDev& s = g_sk[dev];
    for (int i = 0; i < s.nregions; ++i)
        if (hipMemsetAsync(s.regions[i].flags, 0, (size_t)s.regions[i].ints * sizeof(int), as_stream(stream)) != hipSuccess) {
            gssd_set_error("gssd_dcn_streamk_reset: hipMemsetAsync failed");
            return GSSD_ELAUNCH;
        }
    return GSSD_OK;
}

// frees the flag / slab region of the launches that write `out` (NULL: every region of the current device).  A region is ~32 MB on a 256-CU
// device and is created by the first stream-K launch for an output: a caller that drops or reallocates its outputs releases theirs (ADVICE r4;
// gssd/plan_ops.py does when a forward plan is destroyed).  hipFree synchronises: not to be called while a stream is capturing.
extern "C" int gssd_dcn_streamk_release(const void* out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16) return GSSD_OK;
    std::lock_guard<std::mutex> lock(g_sk_mu);
    SkDev& s = g_sk[dev];
    int kept = 0;
    for (int i = 0; i < s.nregions; ++i) {
        if (out == nullptr || s.regions[i].out == out) {
            (void)hipFree(s.regions[i].flags);
            (void)hipFree(s.regions[i].ws);
        } else {
            s.regions[kept++] = s.regions[i];
        }
    }
    const int freed = s.nregions - kept;
    s.nregions = kept;
    for (int i = 0; i < s.nretired; ++i) {               // regions replaced inside a launch call (sk_region)
        (void)hipFree(s.retired[i].flags);
        (void)hipFree(s.retired[i].ws);
    }
    s.nretired = 0;
    return freed;
}

extern "C" int gssd_dcn_forward_f32(const float* x, const float* om, const float* w_packed, const float* bias, float* out, int B,
                                    int H, int W, int C, int dg, int om_stride, int Cout, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && w_packed && out && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0 && Cout > 0);
    GSSD_CHECK_ARG(C % dg == 0 && (C / dg) % BKC == 0 && om_stride >= 27 * dg);
    GSSD_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0);
    const long long Mll = (long long)B * H * W;
    GSSD_CHECK_ARG(Mll < (1ll << 31) && Mll * C < (1ll << 30));          // 32-bit byte offsets into x
    const int M = (int)Mll;
    const int ntn = (Cout + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    static unsigned attr_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    constexpr int smem = LDS_FLOATS * (int)sizeof(float);
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
                hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", smem);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask);
    }
    // grid: ids round-robin over the 8 XCDs; slots cover ceil(mtiles / (8 / ntn)) groups when ntn divides 8
    int blocks;
    if (8 % ntn == 0) {
        const int per = 8 / ntn;
        blocks = ((mtiles + per - 1) / per) * 8;
    } else {
        blocks = ((mtiles * ntn + 7) / 8) * 8;
    }
    // Stream-K form (one persistent workgroup per CU, equal spans of the (tile, chunk) space): taken when every XCD has at least as
    // many tiles as workgroups (a tile then straddles at most one span boundary) and the tile count does not already fill whole rounds.
    // GSSD_DCN_STREAMK=0 keeps one tile per workgroup.  Per-device state (placement probe, flag pool, error word): SkDev above.
    static const bool sk_off = []() { const char* e = getenv("GSSD_DCN_STREAMK"); return e && e[0] == '0'; }();
    bool sk = (g_dcn_sk_force < 0 ? !sk_off : g_dcn_sk_force == 1) && dev >= 0 && dev < 16;
    if (sk) {
        std::lock_guard<std::mutex> sk_lock(g_sk_mu);
        SkDev& s = g_sk[dev];
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(as_stream(stream), &cap);
        const bool may_alloc = cap == hipStreamCaptureStatusNone;
        if (s.cus == 0 && may_alloc) sk_init(s, dev);                        // (a first launch inside a capture takes the one-tile form)
        if (s.err_host && !(s.status & GSSD_DCN_SK_TIMEOUT)) {
            const unsigned e = __atomic_load_n(s.err_host, __ATOMIC_ACQUIRE);
            if (e & 2u) {
                s.status |= GSSD_DCN_SK_TIMEOUT;
                gssd_set_error("gssd_dcn_forward_f32: an earlier stream-K launch timed out waiting for a partial sum (aborted launch?); its "
                               "output is not trustworthy -- stream-K is now off on device %d (GSSD_DCN_STREAMK=0 avoids it from the start)", dev);
                return GSSD_ELAUNCH;
            }
        }
        const int nslots = blocks / 8;                       // tiles of the fullest XCD; the emptiest has nslots - 1 or nslots
        sk = s.cus > 0 && !(s.status & (GSSD_DCN_SK_UNSUPPORTED | GSSD_DCN_SK_TIMEOUT)) && nslots >= 1 && blocks > s.cus && blocks % s.cus != 0;
        SkRegion* rg = sk ? sk_region(s, out, blocks, may_alloc) : nullptr;
        sk = sk && rg;
        if (sk) {
            hipLaunchKernelGGL(dcn_fused_kernel<true>, dim3(s.cus), dim3(256), smem, as_stream(stream), x, om, w_packed, bias, out, M, H, W,
                               C, dg, om_stride, Cout, ntn, mtiles, rg->flags, rg->ws, (unsigned)((size_t)s.cus * SK_SLAB_BYTES), s.err_dev);
        }
    }
    if (!sk) {
        hipLaunchKernelGGL(dcn_fused_kernel<false>, dim3(blocks), dim3(256), smem, as_stream(stream), x, om, w_packed, bias, out, M, H,
                           W, C, dg, om_stride, Cout, ntn, mtiles, (int*)nullptr, (float*)nullptr, 0u, (unsigned*)nullptr);
    }
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
