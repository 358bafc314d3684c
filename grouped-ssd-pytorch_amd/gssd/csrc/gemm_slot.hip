// Large 1x1 convolutions / plain GEMMs on the fp32 matrix cores as ONE slot-scheduled instruction stream (the structure that took
// the fused deformable conv from 0.55 to 0.76 of the fp32 peak, csrc/dcn_fused.hip, without its gather):
//
//   out[m][n] = epilogue( sum_k A[m][k] * Wp[n][k] ),   A = NHWC activations (row stride in_stride), Wp = K-major weight rows
//
// A 256-thread workgroup (2 x 2 waves) owns BM = 128 rows x BN = 128 columns, TWO workgroups per CU (64 KB of LDS each); a wave owns
// 64 x 64 = 4 x 4 tiles of v_mfma_f32_16x16x4_f32 (64 accumulator registers).  (BN = 256, one workgroup per CU, 128 accumulators: the
// tiling of dcn_fused; kept as a template instance for ablation, see slot_shape.)  K runs in chunks of 32 floats: both tiles of chunk ch+1 land
// in the other LDS stage by 16-byte LDS-DMA (8 rows x 128 B per wave instruction, source-side XOR swizzle: slot' = slot ^ (row & 7),
// so every ds_read_b128 fragment read is conflict free) while the 128 MFMAs of chunk ch run.  Every staging instruction sits behind
// a specific MFMA ("slot") and a scheduling fence after each slot keeps hipcc from regrouping them:
//   (128 slots per chunk)   0..7  second-half (k 16..31) fragment reads of this chunk     8, 10, .. 22  the 8 DMA pieces of the next chunk
//   96   the chunk's single barrier                                    98..105  first-half fragment reads of the NEXT chunk
// Compared with conv_igemm<128x128> (same tile and residency, barrier at the chunk end, fragments read after it: 0.57-0.65 of peak)
// the MFMA pipe never waits for a barrier or a fragment read: 0.61-0.84 of peak on the shapes of scripts/bench_gemm.py
// (M = 46 208: K = 256 -> 97 TFLOP/s, K = 1024 -> 119-127, K = 4096 -> 132; the DCN d(cols) GEMM 133; generic kernel 77 / 100 / 113 / 107).
// Taken for: KH = KW = 1, stride 1, groups 1, K % 32 == 0, no fused input transform, no split-K, wide enough (see slot_shape);
// everything else stays on conv_igemm.  Epilogue = conv_igemm's (alpha, bias, gate / residual / second output, ReLU, transposed and
// split-transposed stores, BatchNorm statistics).
#include <type_traits>
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BK = 32;
constexpr int WTM = 64, MT = WTM / 16;
constexpr int A_STAGE = BM * BK;
#ifndef DMA_EVERY
#define DMA_EVERY 2
#endif

__device__ __attribute__((aligned(16))) float g_zero_gs[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
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

// BN = 256: one workgroup per CU (96 KB of LDS, 128 accumulator registers per lane);  BN = 128: two per CU (64 KB, 64 registers) --
// the second one's MFMAs cover the first one's prologue / epilogue, which is what short reductions (K <= 512) and grids of about one
// round need.
// SWAP (NHWC outputs): the MFMA takes the weight fragment as its first operand, so the accumulator tile is [channel 4 kq + e][pixel r]
// -- a lane holds FOUR CONSECUTIVE CHANNELS of one pixel and the epilogue moves 16 bytes per instruction (bias, 1/sigma, gated
// residual, second output, statistics): the pixel-major form spends 3 x 64 four-byte memory instructions per wave on the Self_Attn
// output conv (K = 256: its 8 chunks of MFMAs were a third of the tile's time).  The transposed outputs keep the pixel-major form
// (there four consecutive PIXELS of one channel are the 16-byte unit).
template <int BN, bool SWAP>
__global__ __launch_bounds__(256, BN == 256 ? 1 : 2) void gemm_slot_kernel(const gssd_conv_desc p, const int M, const int ntn, const int mtiles) {
    constexpr int WTN = BN / 2, NT = WTN / 16, B_STAGE = BN * BK, NBP = BN / 32;     // NBP: B DMA pieces per wave per chunk
    constexpr int SL = MT * NT * 8;                                                  // MFMAs (slots) per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                     // [2][BM][32]
    float* const Bs = smem + 2 * A_STAGE;       // [2][BN][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
    // XCD-aware tile order: workgroup id L runs on XCD L & 7; when the N-tile count divides 8 an XCD keeps ONE weight slab and walks
    // M tiles, otherwise consecutive ids share the M tile (A rows from that L2)
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
    const int img = p.m_per_image ? blockIdx.z : 0;
    const int m0 = mt * BM, n0 = nt * BN;
    const int K = p.K;
    const int nchunks = K / BK;
    const float* __restrict__ in = p.in + (size_t)img * p.in_batch_stride + p.in_ch_off;
    const float* __restrict__ wgt = p.wgt + (size_t)img * p.wgt_batch_stride;

    // DMA lane roles: lane L lands at (row_in = L >> 3, slot = L & 7) of its 8-row piece and fetches logical k-quad slot ^ row_in
    const int row_in = lane >> 3;
    const int lq = (lane & 7) ^ row_in;
    const float* a_src[4];
    int a_step[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + (j * 4 + wave) * 8 + row_in;
        const bool ok = m < M;
        a_src[j] = ok ? in + (size_t)m * p.in_stride + 4 * lq : g_zero_gs;
        a_step[j] = ok ? BK : 0;
    }
    const float* b_src[NBP];
    int b_step[NBP];
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        const int n = n0 + (j * 4 + wave) * 8 + row_in;
        const bool ok = n < p.Cout;
        b_src[j] = ok ? wgt + (size_t)n * p.wgt_row_stride + 4 * lq : g_zero_gs;
        b_step[j] = ok ? BK : 0;
    }

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    const int fo0 = r * BK + ((kq ^ (r & 7)) << 2);
    const int fo1 = r * BK + (((4 + kq) ^ (r & 7)) << 2);

    // ---- prologue: chunk 0 -------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dma16(a_src[j], As + (j * 4 + wave) * 256);
        a_src[j] += a_step[j];
    }
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        dma16(b_src[j], Bs + (j * 4 + wave) * 256);
        b_src[j] += b_step[j];
    }
    __syncthreads();

    f32x4 af[2][MT], bf[2][NT];
    float* a_dst = nullptr;
    float* b_dst = nullptr;
    auto slot = [&](auto kc, auto stage_c, const float* Ab, const float* Bb, const float* Abn, const float* Bbn) {
        constexpr int KK = decltype(kc)::value;
        constexpr bool stage = decltype(stage_c)::value;
        constexpr int ks = KK / (SL / 2), s = (KK / (MT * NT)) & 3, i = (KK / NT) & 3, j = KK % NT;
        if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[ks][j][s], af[ks][i][s], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][i][s], bf[ks][j][s], acc[i][j], 0, 0, 0);
        if constexpr (KK < MT) af[1][KK] = *reinterpret_cast<const f32x4*>(Ab + KK * 16 * BK + fo1);
        if constexpr (KK >= MT && KK < MT + NT) bf[1][KK - MT] = *reinterpret_cast<const f32x4*>(Bb + (KK - MT) * 16 * BK + fo1);
        if constexpr (stage) {
            constexpr int D0 = MT + NT;                     // the 4 + NBP DMA pieces of the next chunk, every DMA_EVERY slots
            if constexpr (KK >= D0 && KK < D0 + (4 + NBP) * DMA_EVERY && (KK - D0) % DMA_EVERY == 0) {
                constexpr int n = (KK - D0) / DMA_EVERY;
                if constexpr (n < 4) {
                    dma16(a_src[n], a_dst + n * 1024);
                    a_src[n] += a_step[n];
                } else {
                    dma16(b_src[n - 4], b_dst + (n - 4) * 1024);
                    b_src[n - 4] += b_step[n - 4];
                }
            }
            // the ONE barrier of the chunk sits inside the MFMA stream: both tiles of chunk ch+1 are in LDS, the first-half fragment
            // registers are dead since slot SL/2 - 1 -> they are refilled for chunk ch+1 while the second-half MFMAs of this chunk run
            constexpr int BAR = SL * 3 / 4;
            if constexpr (KK == BAR) __syncthreads();
            if constexpr (KK >= BAR + 2 && KK < BAR + 2 + MT) af[0][KK - BAR - 2] = *reinterpret_cast<const f32x4*>(Abn + (KK - BAR - 2) * 16 * BK + fo0);
            if constexpr (KK >= BAR + 2 + MT && KK < BAR + 2 + MT + NT)
                bf[0][KK - BAR - 2 - MT] = *reinterpret_cast<const f32x4*>(Bbn + (KK - BAR - 2 - MT) * 16 * BK + fo0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    {
        const float* Ab = As + wm * WTM * BK;
        const float* Bb = Bs + wn * WTN * BK;
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(Ab + i * 16 * BK + fo0);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(Bb + j * 16 * BK + fo0);
    }
    for (int ch = 0; ch < nchunks - 1; ++ch) {
        const int buf = ch & 1;
        a_dst = As + (buf ^ 1) * A_STAGE + wave * 256;
        b_dst = Bs + (buf ^ 1) * B_STAGE + wave * 256;
        const float* Ab = As + buf * A_STAGE + wm * WTM * BK;
        const float* Bb = Bs + buf * B_STAGE + wn * WTN * BK;
        const float* Abn = As + (buf ^ 1) * A_STAGE + wm * WTM * BK;
        const float* Bbn = Bs + (buf ^ 1) * B_STAGE + wn * WTN * BK;
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, SL>([&](auto kc) { slot(kc, std::true_type{}, Ab, Bb, Abn, Bbn); });
    }
    {
        const int buf = (nchunks - 1) & 1;
        const float* Ab = As + buf * A_STAGE + wm * WTM * BK;
        const float* Bb = Bs + buf * B_STAGE + wn * WTN * BK;
        static_for<0, SL>([&](auto kc) { slot(kc, std::false_type{}, Ab, Bb, Ab, Bb); });
    }

    // ---- epilogue (conv_igemm's, groups == 1, no split-K, no head layout).  One workgroup per CU: nothing overlaps this phase, so the
    //      common store path is kept to one FMA + one store per element (row pointers hoisted, statistics only when asked for) ------
    if constexpr (SWAP) {
        // acc[i][j][e]: channel n0 + wn*WTN + 16 j + 4 kq + e, pixel m0 + wm*WTM + 16 i + r   (NHWC outputs only)
        const float gate = p.gate ? *p.gate : 0.f;
        const bool want_stats = p.stats != nullptr;
        float ssum[NT][4], ssq[NT][4];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) ssum[j][e] = ssq[j][e] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * WTN + j * 16 + 4 * kq;
            if (n >= p.Cout) continue;                             // (Cout is a multiple of 4 on this path: slot_shape)
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, alpha4 = {1.f, 1.f, 1.f, 1.f};
            if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.alpha) alpha4 = *reinterpret_cast<const f32x4*>(p.alpha + n);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = m0 + wm * WTM + i * 16 + r;
                if (m >= M) continue;
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(acc[i][j][e], alpha4[e], bias4[e]);
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ssum[j][e] += t[e];
                        ssq[j][e] = __builtin_fmaf(t[e], t[e], ssq[j][e]);
                    }
                }
                const size_t o = (size_t)img * p.out_batch_stride + (size_t)m * p.out_stride + p.out_ch_off + n;
                if (p.gate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] *= gate;
                    if (p.out2) *reinterpret_cast<f32x4*>(p.out2 + o) = t;
                }
                if (p.resid) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.resid + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] += rv[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], 0.f);
                }
                *reinterpret_cast<f32x4*>(p.out + o) = t;
            }
        }
        if (p.stats) {
            // per-channel sums over this tile's rows: the 16 lanes sharing kq hold 16 pixels of the same channels
            __syncthreads();
            float* red = smem;  // [2][BN][2]
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sv = ssum[j][e], qv = ssq[j][e];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        sv += __shfl_xor(sv, o, 64);
                        qv += __shfl_xor(qv, o, 64);
                    }
                    if (r == 0) {
                        red[(wm * BN + wn * WTN + j * 16 + 4 * kq + e) * 2 + 0] = sv;
                        red[(wm * BN + wn * WTN + j * 16 + 4 * kq + e) * 2 + 1] = qv;
                    }
                }
            __syncthreads();
            if (tid < BN && n0 + tid < p.Cout) {
                const double sv = (double)red[tid * 2 + 0] + (double)red[(BN + tid) * 2 + 0];
                const double qv = (double)red[tid * 2 + 1] + (double)red[(BN + tid) * 2 + 1];
                double* st = gssd_stats_replica(p.stats, p.stats_rep, p.Cout);
                unsafeAtomicAdd(st + n0 + tid, sv);
                unsafeAtomicAdd(st + p.Cout + n0 + tid, qv);
            }
        }
        return;
    }
    const float gate = p.gate ? *p.gate : 0.f;
    const bool want_stats = p.stats != nullptr;
    const bool plain = p.out_mode == GSSD_OUT_NHWC && !p.gate && !p.resid;
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) ssum[j] = ssq[j] = 0.f;
    float biasv[NT], alphav[NT];
    bool nok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WTN + j * 16 + r;
        nok[j] = n < p.Cout;
        biasv[j] = (p.bias && nok[j]) ? p.bias[n] : 0.f;
        alphav[j] = (p.alpha && nok[j]) ? p.alpha[n] : 1.f;
    }
    if (plain) {
        float* const obase = p.out + (size_t)img * p.out_batch_stride + p.out_ch_off + n0 + wn * WTN + r;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + wm * WTM + i * 16 + kq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool mok = mb + e < M;
                float* const orow = obase + (size_t)(mb + e) * p.out_stride;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float t = __builtin_fmaf(acc[i][j][e], alphav[j], biasv[j]);
                    if (want_stats && mok && nok[j]) {
                        ssum[j] += t;
                        ssq[j] = __builtin_fmaf(t, t, ssq[j]);
                    }
                    if (p.relu) t = fmaxf(t, 0.f);
                    if (mok && nok[j]) orow[j * 16] = t;
                }
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WTN + j * 16 + r;
        const bool n_ok = nok[j];
        const float bias = biasv[j], alpha = alphav[j];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + wm * WTM + i * 16 + kq * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[i][j][e] * alpha + bias;
                if (want_stats && mb + e < M && n_ok) {
                    ssum[j] += v[e];
                    ssq[j] += v[e] * v[e];
                }
            }
            if (!n_ok) continue;
            if (p.out_mode == GSSD_OUT_SPLIT_T && n >= p.split_n) {
                const int hw = p.Ho * p.Wo;                   // all images in one M range: the image index comes from the row
                const int bi = p.m_per_image ? img : mb / hw;
                const int ml = p.m_per_image ? mb : mb - bi * hw;
                if (ml < p.out_b_stride && (p.m_per_image || mb < M)) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (mb + e < M) ? v[e] : 0.f;
                    *reinterpret_cast<f32x4*>(p.out_b + (size_t)bi * p.outb_batch_stride + (size_t)(n - p.split_n) * p.out_b_stride + ml) = o;
                }
            } else if (p.out_mode == GSSD_OUT_TRANSPOSED) {
                if (mb < p.out_stride) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[e];
                        if (p.relu) t = fmaxf(t, 0.f);
                        o[e] = (mb + e < M) ? t : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(p.out + (size_t)img * p.out_batch_stride + (size_t)n * p.out_stride + mb) = o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = mb + e;
                    if (m >= M) continue;
                    float t = v[e];
                    const size_t o = (size_t)img * p.out_batch_stride + (size_t)m * p.out_stride + p.out_ch_off + n;
                    if (p.gate) {
                        t *= gate;
                        if (p.out2) p.out2[o] = t;
                        if (p.resid) t += p.resid[o];
                    } else if (p.resid) {
                        t += p.resid[o];
                    }
                    if (p.relu) t = fmaxf(t, 0.f);
                    p.out[o] = t;
                }
            }
        }
    }
    }
    if (p.stats) {
        // per-channel sum / sum^2 over this tile's rows: lanes sharing (lane & 15) -> LDS over wm -> fp64 atomics
        __syncthreads();
        float* red = smem;  // [2][BN][2]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = ssum[j], q = ssq[j];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (kq == 0) {
                red[(wm * BN + wn * WTN + j * 16 + r) * 2 + 0] = s;
                red[(wm * BN + wn * WTN + j * 16 + r) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.Cout) {
            const double s = (double)red[tid * 2 + 0] + (double)red[(BN + tid) * 2 + 0];
            const double q = (double)red[tid * 2 + 1] + (double)red[(BN + tid) * 2 + 1];
            double* st = gssd_stats_replica(p.stats, p.stats_rep, p.Cout);
            unsafeAtomicAdd(st + n0 + tid, s);
            unsafeAtomicAdd(st + p.Cout + n0 + tid, q);
        }
    }
}

}  // namespace

// shape classes: 256 = the 128 x 256 stream (one workgroup per CU), 128 = the 128 x 128 stream (two per CU), 0 = not taken
static int slot_shape(const gssd_conv_desc& d) {
    if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad != 0 || d.groups != 1 || d.split_k != 1) return 0;
    if (d.in_scale || d.out_mode == GSSD_OUT_HEADS || d.K % BK != 0 || d.K < 2 * BK) return 0;
    if (d.out_mode == GSSD_OUT_SPLIT_T && (d.split_n % 16 != 0)) return 0;
    const long long M = (long long)(d.m_per_image ? 1 : d.B) * d.Ho * d.Wo;
    const int images = d.m_per_image ? d.B : 1;
    const int mtiles = (int)((M + BM - 1) / BM);
    // (A 128 x 256 / one-workgroup-per-CU form -- the tiling of dcn_fused, where the gather forces it -- was measured on every shape of
    // scripts/bench_gemm.py and never won: K = 512, N = 9216: 3.38 vs 3.28 ms; K = 256: 127 vs 125 us -- the second resident workgroup's
    // MFMAs cover the ~20 us of prologue + epilogue a lone workgroup leaves exposed per tile.  Its switch is gone; the BN = 256 template
    // instance stays compilable for scripts/bench_gemm.py-style experiments.)
    // 128 x 128: any wide enough plain GEMM with at least ~3/8 of a round of workgroups (below that the generic kernel's 128 x 64
    // tiles spread the work over more CUs: M = 3200, N = 512 measures 55 us there, 64 us here)
    const int ntn = (d.Cout + 127) / 128;
    const double nfill = (double)d.Cout / (double)(ntn * 128);
    if (nfill < 0.75 || (long long)mtiles * ntn * images < 192) return 0;
    // Per-image batched projections only: last-round fill against the generic kernel's 128 x 64 tiling (three resident workgroups per
    // CU).  B = 32 x 19 x 19 tokens x 768 columns per image is 576 workgroups = 1.1 rounds of 512 here and 1152 = 1.5 rounds of 768
    // there: measured 252 vs 200 us.  Plain (non-batched) GEMMs are faster here at every fill measured (scripts/bench_gemm.py:
    // M = 46 208, N = 256 -> 1.4 rounds: 192 vs 240 us; N = 384 -> 2.1 rounds: 166 vs 189 us): a CU left with one workgroup in the
    // last round runs it at nearly twice the shared rate.
    if (!d.m_per_image) return 128;
    auto fill = [](long long wgs, long long slots) { return (double)wgs / (double)(((wgs + slots - 1) / slots) * slots); };
    const long long w128 = (long long)mtiles * ntn * images, w64 = (long long)mtiles * ((d.Cout + 63) / 64) * images;
    if (fill(w128, 512) * 1.2 >= fill(w64, 768)) return 128;
    return 0;
}

extern "C" int gssd_gemm_slot_takes(const gssd_conv_desc* d) { return d && slot_shape(*d) ? 1 : 0; }

template <int BN, bool SWAP>
static int launch_slot(const gssd_conv_desc& d, hipStream_t stream) {
    const long long M = (long long)(d.m_per_image ? 1 : d.B) * d.Ho * d.Wo;
    const int images = d.m_per_image ? d.B : 1;
    const int ntn = (d.Cout + BN - 1) / BN, mtiles = (int)((M + BM - 1) / BM);
    static unsigned attr_mask = 0;
    constexpr int smem = 2 * (A_STAGE + BN * BK) * (int)sizeof(float);
    if (gssd_attr_needed(&attr_mask) &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_slot_kernel<BN, SWAP>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", smem);
        return GSSD_ELAUNCH;
    }
    gssd_attr_done(&attr_mask);
    int blocks;
    if (8 % ntn == 0) {
        const int per = 8 / ntn;
        blocks = ((mtiles + per - 1) / per) * 8;
    } else {
        blocks = ((mtiles * ntn + 7) / 8) * 8;
    }
    hipLaunchKernelGGL((gemm_slot_kernel<BN, SWAP>), dim3(blocks, 1, images), dim3(256), smem, stream, d, (int)M, ntn, mtiles);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// returns 1 when the descriptor is not a plain 1x1 / GEMM shape worth a slot stream (the caller falls through to conv_igemm)
int gssd_try_gemm_slot(const gssd_conv_desc& d, hipStream_t stream) {
    const int cls = slot_shape(d);
    // channel-major accumulators (16-byte epilogue accesses) for NHWC outputs with 16-byte aligned channel vectors; GSSD_GEMM_SLOT_SWAP=0
    // keeps the pixel-major form everywhere (ablation)
    static const bool no_swap = getenv("GSSD_GEMM_SLOT_SWAP") != nullptr && atoi(getenv("GSSD_GEMM_SLOT_SWAP")) == 0;
    // (measured: gated residual epilogues -- the Self_Attn output convs at 38 x 38 -- 223 -> 170 us and 195 -> 144 us; plain NHWC epilogues
    // with statistics -- the fuse convs -- 208 -> 231 us and 205 -> 214 us: those keep the pixel-major form)
    const bool swap = !no_swap && d.out_mode == GSSD_OUT_NHWC && (d.gate || d.resid) && d.Cout % 4 == 0 && d.out_stride % 4 == 0 && d.out_ch_off % 4 == 0 &&
                      ((uintptr_t)d.out % 16) == 0 && (!d.resid || ((uintptr_t)d.resid % 16) == 0) && (!d.out2 || ((uintptr_t)d.out2 % 16) == 0) &&
                      (!d.bias || ((uintptr_t)d.bias % 16) == 0) && (!d.alpha || ((uintptr_t)d.alpha % 16) == 0) && d.out_batch_stride % 4 == 0;
    if (cls == 256) return swap ? launch_slot<256, true>(d, stream) : launch_slot<256, false>(d, stream);
    if (cls == 128) return swap ? launch_slot<128, true>(d, stream) : launch_slot<128, false>(d, stream);
    return 1;
}
