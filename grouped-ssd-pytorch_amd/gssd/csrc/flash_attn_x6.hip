// Flash-style Self_Attn core of the fp32 mode on the BF16 matrix cores with fp32-equivalent products (round 5): the same contract as
// gssd_self_attn_core_f32 (csrc/flash_attn.hip; layers/self_attn.py:68-80: bmm -> Softmax(dim=-1) -> bmm, fp32 in, fp32 out) with both
// products -- the logits theta . phi and the value product P . g -- computed as six v_mfma_f32_16x16x32_bf16 over operands that are the
// exact sum of three bf16 planes (x = h + m + l, every product term above 2^-24; conv_x6.hip / dcn_x6.hip).  flash_attn.hip runs them
// on v_mfma_f32_16x16x4_f32, 1/16 of the bf16 matrix rate: the N = 1444 block is 405 us of an fp32 GSSD++ step.
//
//   pass 1 (split_planes_kernel)  theta | phi [B][N][2D] fp32 -> three bf16 planes, same layout;  g^T [B][C2][Np] fp32 -> three bf16 planes
//                                 [B][C2][Np32] with the keys of every 32-block in the order the MFMA's k index wants
//                                 (k = 8 kq + e  <->  key 32 t + 16 (e >> 2) + 4 kq + (e & 3), as the bf16-value core of flash_attn.hip)
//   pass 2 (flash_attn_x6_kernel) a 256-thread workgroup owns 64 queries of one image, wave w 16 of them and ALL C2 value channels; K / V
//                                 tiles of BKV keys (three planes each) are staged by LDS-DMA; everything stays in the "column = query"
//                                 orientation of the MFMA's C layout: S^T = K . Q^T lands as lane (q, kq) <- keys 16 kt + 4 kq + reg, exactly
//                                 the k slots of the next product's B operand; the probabilities are split into their three planes in
//                                 registers; the six products of a tile are summed from zero and added to the running output by the vector
//                                 ALU (the bf16 MFMA's adder truncates); softmax, running maximum and row sums are fp32 as before.
#include <math.h>
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

namespace {

constexpr int NP = 3;

__device__ __attribute__((aligned(16))) u16 g_zero16_x6[8] = {0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// rows x cols fp32 (row stride ld_in) -> three bf16 planes [rows][cols_out] (plane stride `plane`); perm32: column c of the output holds input
// column 32 t + perm(c % 32) (the key order of the value product); columns >= cols are zero.  One thread = 8 output columns = 16 bytes per plane.
// F16 (round 6): TWO fp16 planes instead -- h = fp16(x), l' = fp16((x - h) * 2048), x = h + l' / 2048 to 2^-24 |x| -- for the three-MFMA form of
// the core (conv_thin_x6.hip): theta, phi and g are projections of normalised activations, far inside fp16's range.
template <bool F16>
__global__ void split_planes_kernel(const float* __restrict__ in, u16* __restrict__ out, long long rows, int cols, int ld_in, int cols_out,
                                    long long plane, int perm32) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int units = cols_out >> 3;
    if (i >= rows * units) return;
    const long long row = i / units;
    const int u = (int)(i - row * units);
    const float* src = in + row * ld_in;
    bf16x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int c = 8 * u + e;
        if (perm32) {
            // slot s = 8 kq + e of a 32-block holds key 16 (e >> 2) + 4 kq + (e & 3)
            const int s = c & 31, kq = s >> 3, ee = s & 7;
            c = (c & ~31) + 16 * (ee >> 2) + 4 * kq + (ee & 3);
        }
        const float v = c < cols ? src[c] : 0.f;
        if (F16) {
            const _Float16 fh = (_Float16)v;
            const _Float16 fl = (_Float16)((v - (float)fh) * 2048.f);
            h[e] = __builtin_bit_cast(__bf16, fh);
            m[e] = __builtin_bit_cast(__bf16, fl);
        } else {
            __bf16 a, b, d;
            split3(v, a, b, d);
            h[e] = a;
            m[e] = b;
            l[e] = d;
        }
    }
    u16* dst = out + row * cols_out + 8 * u;
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + plane) = m;
    if (!F16) *reinterpret_cast<bf16x8*>(dst + 2 * plane) = l;
}

// tpp: planes of theta | phi [3][B][N][2D]; gp: planes of g^T [3][B][C2][Np32] (keys permuted inside 32-blocks); out [B][N][C2] fp32
// NW waves per workgroup = 16 NW queries: every workgroup streams ALL keys / values of its image through LDS, so the L2 -> LDS traffic per query
// falls with NW (four waves: 2.0 GB per N = 1444 launch = the bound, 346 us; twelve waves = 192 queries: exactly one round of 8 x 32 workgroups)
template <int D, int C2, int BKV, int NW, bool F16>
__global__ __launch_bounds__(64 * NW, 1) void flash_attn_x6_kernel(const u16* __restrict__ tpp, const u16* __restrict__ gp,
                                                                               float* __restrict__ out, int N, int Np32, int qtiles,
                                                                               long long tp_plane, long long g_plane,
                                                                               float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int NPX = F16 ? 2 : 3;                      // operand planes of this instance
    constexpr int UK = D / 8, UV = BKV / 8;               // 16-byte units per K row / V row
    constexpr int SWK = (UK < 8 ? UK : 8) - 1;            // K rows are >= 64 bytes: unit' = unit ^ (row & SWK)
    constexpr int KT = BKV / 16, KB = BKV / 32, CT = C2 / 16, DI = D / 32;
    constexpr int K_PLANE = BKV * D, V_PLANE = C2 * BKV;
    static_assert(BKV == 32 || BKV == 64, "value rows of 64 or 128 bytes");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x / qtiles, qt = blockIdx.x - b * qtiles;
    const int q = qt * (16 * NW) + wave * 16 + r;
    const u16* tpb = tpp + (size_t)b * N * (2 * D);
    const u16* gb = gp + (size_t)b * C2 * Np32;

    // query fragments (B operand): lane (q, kq) holds theta[q][32 i + 8 kq .. + 7] of every plane
    bf16x8 qf[DI][NPX];
#pragma unroll
    for (int i = 0; i < DI; ++i)
#pragma unroll
        for (int pl = 0; pl < NPX; ++pl) {
            if (q < N) qf[i][pl] = *reinterpret_cast<const bf16x8*>(tpb + pl * tp_plane + (size_t)q * (2 * D) + 32 * i + 8 * kq);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[i][pl][e] = (__bf16)0.f;
        }
    f32x4 o[CT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CT; ++c) o[c] = zero4;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (N + BKV - 1) / BKV;
    constexpr int STAGE = NPX * (K_PLANE + V_PLANE);           // u16 elements per stage; two stages: the next tile lands under this tile's MFMAs
    // the DMA pieces of a tile, one at a time: piece slot j of this wave = K pieces first (KW per wave), then V pieces (VW per wave).  In the
    // loop they are dealt out between the MFMA groups of the running tile (round 5: a wave issues in order, and the vector memory path takes
    // one 1-KiB piece per 16 cycles per CU -- a burst of 60 pieces at the top of a tile held back every wave's MFMAs for up to ~1 000 cycles)
    constexpr int K_RPP = 64 / UK, K_PIECES = NPX * BKV / K_RPP, V_RPP = 64 / UV, V_PIECES = NPX * C2 / V_RPP;
    constexpr int KW = (K_PIECES + NW - 1) / NW, VW = (V_PIECES + NW - 1) / NW;
    auto stage_piece = [&](const int t, const int buf, const int j) {
        const int key0 = t * BKV;
        u16* const Kd = smem + buf * STAGE;
        u16* const Vd = Kd + NPX * K_PLANE;
        if (j < KW) {   // K tile: BKV rows of D bf16 per plane; piece = 1 KiB = 64 / UK rows
            const int row_in = lane / UK, slot = lane % UK;
            const int piece = j * NW + wave;
            if (K_PIECES % NW != 0 && piece >= K_PIECES) return;
            const int pl = piece / (BKV / K_RPP), row = (piece - pl * (BKV / K_RPP)) * K_RPP + row_in;
            const int unit = slot ^ (row & SWK);
            const u16* src = key0 + row < N ? tpb + pl * tp_plane + (size_t)(key0 + row) * (2 * D) + D + 8 * unit : g_zero16_x6;
            dma16(src, Kd + piece * 512);
        } else {        // V tile: C2 rows of BKV bf16 per plane
            const int row_in = lane / UV, slot = lane % UV;
            const int piece = (j - KW) * NW + wave;
            if (V_PIECES % NW != 0 && piece >= V_PIECES) return;
            const int pl = piece / (C2 / V_RPP), row = (piece - pl * (C2 / V_RPP)) * V_RPP + row_in;
            const int unit = slot ^ (BKV == 32 ? ((row & 8) ? 3 : 0) : (row & 7));
            const u16* src = key0 + 8 * unit < Np32 ? gb + pl * g_plane + (size_t)row * Np32 + key0 + 8 * unit : g_zero16_x6;
            dma16(src, Vd + piece * 512);
        }
    };
    auto stage = [&](const int t, const int buf) {
#pragma unroll
        for (int j = 0; j < KW + VW; ++j) stage_piece(t, buf, j);
    };
#ifndef FX6_SPREAD
#define FX6_SPREAD 1         // 1: the next tile's pieces between the MFMA groups of this tile; 0: all of them at the tile's top
#endif
    stage(0, 0);
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * BKV, buf = t & 1;
        __syncthreads();                                          // (vmcnt(0) + barrier) tile t has landed; everyone is done with tile t - 1
        const int tn = t + 1 < ntiles ? t + 1 : t;             // past the end: the last tile again, into the stage nobody reads any more
        if (!FX6_SPREAD && t + 1 < ntiles) stage(t + 1, buf ^ 1);
        const u16* const Ks = smem + buf * STAGE;
        const u16* const Vs = Ks + NPX * K_PLANE;

        // ---- S^T = K . Q^T: six products per 32 channels, smallest first; the tile's terms summed in one accumulator ---------------
        f32x4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int row = kt * 16 + r;
            f32x4 acc = zero4, accx = zero4;
#pragma unroll
            for (int i = 0; i < DI; ++i) {
                bf16x8 kf[NPX];
#pragma unroll
                for (int pl = 0; pl < NPX; ++pl)
                    kf[pl] = *reinterpret_cast<const bf16x8*>(Ks + pl * K_PLANE + row * D + (((4 * i + kq) ^ (row & SWK)) << 3));
                if constexpr (F16) {
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[1]), __builtin_bit_cast(f16x8, qf[i][0]), accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[0]), __builtin_bit_cast(f16x8, qf[i][1]), accx, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[0]), __builtin_bit_cast(f16x8, qf[i][0]), acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1], qf[i][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[NPX - 1], qf[i][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[i][NPX - 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1], qf[i][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[i][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[i][0], acc, 0, 0, 0);
                }
            }
            if constexpr (F16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(accx[e], 1.f / 2048.f, acc[e]);
            }
            s[kt] = acc;
            // K pieces of the next tile behind the logits of this one
            if (FX6_SPREAD)
#pragma unroll
                for (int j = 0; j < KW; ++j)
                    if (j * KT / KW == kt) {
                        __builtin_amdgcn_sched_barrier(0);       // (keeps the piece's address arithmetic here: hoisted, it costs registers the 12-wave form does not have)
                        stage_piece(tn, buf ^ 1, j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
        // ---- online softmax over the keys of this tile (fp32, as flash_attn.hip) ----------------------------------------------------
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (key0 + kt * 16 + 4 * kq + e >= N) s[kt][e] = -INFINITY;
                mx = fmaxf(mx, s[kt][e]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
        bf16x8 pb[KB][NPX];
        if constexpr (F16) {
            // the probabilities go through the matrix cores scaled by 1024 (undone with the row sum at the end): fp16 keeps 11 bits only down to
            // 6e-5, and a softmax over 1 444 keys has many smaller terms.  Pairs at a time, planes as packed dwords (conv_thin_x6.hip::split2_pair)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                u32x4 ph, pl;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const float p0 = __expf(s[2 * kb + (e2 >> 1)][(2 * e2) & 3] - m_new), p1 = __expf(s[2 * kb + (e2 >> 1)][(2 * e2 + 1) & 3] - m_new);
                    psum += p0;
                    psum += p1;
                    const f32x2 ps = f32x2{p0, p1} * 1024.f;
                    const f16x2 h = __builtin_convertvector(ps, f16x2);
                    const f32x2 rr = (ps - __builtin_convertvector(h, f32x2)) * 2048.f;
                    ph[e2] = __builtin_bit_cast(unsigned, h);
                    pl[e2] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, f16x2));
                }
                pb[kb][0] = __builtin_bit_cast(bf16x8, ph);
                pb[kb][1] = __builtin_bit_cast(bf16x8, pl);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p = __expf(s[2 * kb + (e >> 2)][e & 3] - m_new);
                    psum += p;
                    __bf16 h, m, l;
                    split3(p, h, m, l);
                    pb[kb][0][e] = h;
                    pb[kb][1][e] = m;
                    pb[kb][NPX - 1][e] = l;
                }
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__any(alpha != 1.f)) {                                // the running maximum usually stops moving after a few tiles
#pragma unroll
            for (int c = 0; c < CT; ++c) o[c] *= alpha;
        }
        // ---- O^T += V . P^T ---------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int row = c * 16 + r;
            f32x4 acc = zero4, accx = zero4;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                bf16x8 vf[NPX];
                const int unit = (4 * kb + kq) ^ (BKV == 32 ? ((row & 8) ? 3 : 0) : (row & 7));
#pragma unroll
                for (int pl = 0; pl < NPX; ++pl) vf[pl] = *reinterpret_cast<const bf16x8*>(Vs + pl * V_PLANE + row * BKV + (unit << 3));
                if constexpr (F16) {
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf[1]), __builtin_bit_cast(f16x8, pb[kb][0]), accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf[0]), __builtin_bit_cast(f16x8, pb[kb][1]), accx, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf[0]), __builtin_bit_cast(f16x8, pb[kb][0]), acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[1], pb[kb][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[NPX - 1], pb[kb][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], pb[kb][NPX - 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[1], pb[kb][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], pb[kb][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], pb[kb][0], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[c][e] += F16 ? __builtin_fmaf(accx[e], 1.f / 2048.f, acc[e]) : acc[e];
            if (F16 && (c & 1)) __builtin_amdgcn_sched_barrier(0);      // (keeps hipcc from hoisting the value fragments of many tiles: the twelve-wave form has 168 registers)
            // V pieces of the next tile: VW of them dealt out over the first three quarters of the value tiles
            if (FX6_SPREAD)
#pragma unroll
                for (int j = 0; j < VW; ++j)
                    if (j * (3 * CT / 4) / VW == c) {
                        __builtin_amdgcn_sched_barrier(0);
                        stage_piece(tn, buf ^ 1, KW + j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = (F16 ? 1.f / 1024.f : 1.f) / l_run;          // (F16: the probabilities went through the matrix cores scaled by 1024)
    if (lse != nullptr && q < N && kq == 0) lse[(size_t)b * N + q] = m_run + logf(l_run);
    if (q < N) {
        float* dst = out + ((size_t)b * N + q) * C2 + 4 * kq;
#pragma unroll
        for (int c = 0; c < CT; ++c) *reinterpret_cast<f32x4*>(dst + 16 * c) = o[c] * inv;
    }
}

template <int D, int C2, int BKV, int NW, bool F16>
int launch_x6(const u16* tpp, const u16* gp, float* out, int B, int N, int Np32, float* lse, hipStream_t stream) {
    constexpr int smem = 2 * (F16 ? 2 : 3) * (BKV * D + C2 * BKV) * (int)sizeof(u16);      // two stages
    static unsigned attr_mask = 0;
    auto kern = flash_attn_x6_kernel<D, C2, BKV, NW, F16>;
    if (gssd_attr_needed(&attr_mask) &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", smem);
        return GSSD_ELAUNCH;
    }
    gssd_attr_done(&attr_mask);
    const int qtiles = (N + 16 * NW - 1) / (16 * NW);
    hipLaunchKernelGGL(kern, dim3(B * qtiles), dim3(64 * NW), smem, stream, tpp, gp, out, N, Np32, qtiles, (long long)B * N * 2 * D,
                       (long long)B * C2 * Np32, lse);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

inline long long round32(long long n) { return (n + 31) / 32 * 32; }

}  // namespace

// (128, 512) -- the 19 x 19 maps, N = 361 -- was built and measured slower than the fp32-MFMA core (135 vs 101 us: 192 workgroups of one wave per
// SIMD do not fill the chip) and is not instantiated
extern "C" int gssd_self_attn_core_x6_supported(int D, int C2) { return D == 64 && C2 == 256; }

extern "C" long long gssd_self_attn_core_x6_ws_bytes(int B, int N, int D, int C2) {
    if (B <= 0 || N <= 0 || !gssd_self_attn_core_x6_supported(D, C2)) return -1;
    return 2ll * NP * ((long long)B * N * 2 * D + (long long)B * C2 * round32(N));
}

extern "C" int gssd_self_attn_core_x6_f32(const float* tp, const float* gT, float* out, int B, int N, int Np, int D, int C2, void* ws,
                                          float* lse, gssd_stream_t stream) {
    GSSD_CHECK_ARG(tp && gT && out && ws && B > 0 && N > 0 && Np >= N && Np % 4 == 0);
    GSSD_CHECK_ARG(((uintptr_t)tp % 16) == 0 && ((uintptr_t)gT % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)ws % 16) == 0);
    GSSD_CHECK_ARG((long long)B * ((N + 63) / 64) < (1ll << 31));
    if (!gssd_self_attn_core_x6_supported(D, C2)) {
        gssd_set_error("self-attention core (three-plane): unsupported (theta/phi channels %d, g channels %d); built: (64,256)", D, C2);
        return GSSD_EINVAL;
    }
    hipStream_t s = as_stream(stream);
    const int Np32 = (int)round32(N);
    u16* tpp = reinterpret_cast<u16*>(ws);
    const long long tp_plane = (long long)B * N * 2 * D, g_plane = (long long)B * C2 * Np32;
    u16* gp = tpp + NP * tp_plane;
    // two fp16 planes, three MFMAs per product: built and measured in round 6 -- 313.7 vs 318.1 us (the core is bound by streaming every key / value
    // through LDS per workgroup and by the exponentials, not by the matrix pipe; the twelve-wave instance spills 34 registers) -- opt-in only
    static const bool f16 = [] { const char* e = getenv("GSSD_FLASH_X6_F16"); return e && e[0] == '1'; }();
    {
        const long long n = (long long)B * N * (2 * D / 8);
        if (f16) hipLaunchKernelGGL(split_planes_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tp, tpp, (long long)B * N, 2 * D, 2 * D, 2 * D, tp_plane, 0);
        else hipLaunchKernelGGL(split_planes_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tp, tpp, (long long)B * N, 2 * D, 2 * D, 2 * D, tp_plane, 0);
        GSSD_CHECK_LAUNCH();
    }
    {
        const long long n = (long long)B * C2 * (Np32 / 8);
        if (f16) hipLaunchKernelGGL(split_planes_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gT, gp, (long long)B * C2, N, Np, Np32, g_plane, 1);
        else hipLaunchKernelGGL(split_planes_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gT, gp, (long long)B * C2, N, Np, Np32, g_plane, 1);
        GSSD_CHECK_LAUNCH();
    }
    if (f16) return N > 1024 ? launch_x6<64, 256, 32, 12, true>(tpp, gp, out, B, N, Np32, lse, s) : launch_x6<64, 256, 32, 4, true>(tpp, gp, out, B, N, Np32, lse, s);
    return N > 1024 ? launch_x6<64, 256, 32, 12, false>(tpp, gp, out, B, N, Np32, lse, s) : launch_x6<64, 256, 32, 4, false>(tpp, gp, out, B, N, Np32, lse, s);
}
