// Flash-style Self_Attn core for gfx950 (fp32 MFMA): attn_g[i][c] = sum_j softmax_j(theta_i . phi_j) * g[c][j], without ever
// materialising the [N, N] attention map (layers/self_attn.py:68-80: bmm -> Softmax(dim=-1) -> bmm; no 1/sqrt(d) scaling).
//
//   tp  [B][N][2*D]   token-major projections, theta = channels [0, D), phi = channels [D, 2D)       (queries / keys)
//   gT  [B][C2][Np]   value projection, channel-major, tokens contiguous (Np = N rounded up to 4, pad = 0)
//   out [B][N][C2]    token-major attn_g (the input of the o 1x1 conv)
//
// A 256-thread workgroup owns 64 queries of one image; wave w owns 16 of them and ALL C2 value channels.  Everything is kept in the
// "column = query" orientation of v_mfma_f32_16x16x4_f32's C layout, so no lane transposition is ever needed:
//   S^T tile = K . Q^T   (A = keys [key][d] from LDS, B = queries [query][d] in registers)  -> lane (q = lane & 15, kq = lane >> 4)
//                         holds S[q][key = 16*kt + 4*kq + reg]: exactly the B-operand layout (k = 4*kq + s) of the next product
//   online softmax        row max / rescale per query: 4 registers x BKV/16 tiles, then two lane shuffles (xor 16, 32) across kq
//   O^T += V . P^T        (A = values [channel][key] from LDS, B = P in registers) -> lane holds O[q][c = 16*ct + 4*kq + reg]:
//                         four consecutive channels of one token = one 16-byte NHWC store
// (Measured and rejected, round 2, N = 1444: six waves = 96 queries per workgroup -- exactly one round of 512 workgroups, three waves
// per SIMD -- 581 us against 418 us (the 168-register cap costs more than the fuller round gains); two LDS stages of 32 keys with
// the next tile in flight under the current one, one barrier per tile: 420 us against 407 us -- the staging wait is already covered
// by the CU's second workgroup.)
// K and V tiles are staged by 16-byte LDS-DMA with a source-side XOR swizzle (quad' = quad ^ (row & 15)), which makes every
// ds_read_b128 fragment read conflict free; two workgroups share a CU (80 KB of LDS each) and hide each other's staging.
#include <math.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage a [ROWS][QR quads] tile: lane L of a 1-KiB piece lands at (row_in = L / QR, slot = L % QR) and fetches logical quad
// slot ^ (row & SW).  row_ptr(row) -> global pointer of that row's first float or nullptr (zero row); quad_ok(quad) masks columns.
template <int ROWS, int QR, typename RowPtr, typename QuadOk>
__device__ __forceinline__ void stage_tile(float* lds, int wave, int lane, RowPtr row_ptr, QuadOk quad_ok) {
    constexpr int RPP = 64 / QR;                       // rows per 1-KiB piece
    constexpr int PIECES = ROWS / RPP;
    constexpr int SW = (QR < 16 ? QR : 16) - 1;
    const int row_in = lane / QR, slot = lane % QR;
#pragma unroll
    for (int p0 = 0; p0 < PIECES; p0 += 4) {
        const int piece = p0 + wave;
        if (PIECES % 4 != 0 && piece >= PIECES) break;
        const int row = piece * RPP + row_in;
        const int quad = slot ^ (row & SW);
        const float* rp = row_ptr(row);
        const float* src = (rp != nullptr && quad_ok(quad)) ? rp + 4 * quad : g_zero16;
        dma16(src, lds + piece * 256);
    }
}

template <int D, int C2, int BKV>
__global__ __launch_bounds__(256, (C2 > 256 ? 1 : 2)) void flash_attn_kernel(const float* __restrict__ tp, const float* __restrict__ kp,
                                                           const float* __restrict__ gT, float* __restrict__ out, int N, int Nk,
                                                           int Np, int qtiles, int d_real, int kstride, int out_bf16,
                                                           float* __restrict__ lse, int out_stride, int g_batch_rows) {
    // out_stride / g_batch_rows: the launch may cover a C2-wide SLICE of wider rows (g channels 1024 = two launches of 512: the
    // accumulators of all 1024 would need 256 registers)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ks = smem;                    // [BKV][D]
    float* const Vs = smem + BKV * D;          // [C2][BKV]
    constexpr int QRK = D / 4, QRV = BKV / 4;
    constexpr int SWK = (QRK < 16 ? QRK : 16) - 1, SWV = (QRV < 16 ? QRV : 16) - 1;
    constexpr int KT = BKV / 16, CT = C2 / 16, DI = D / 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x / qtiles, qt = blockIdx.x - b * qtiles;
    const int q = qt * 64 + wave * 16 + r;                       // this lane's query (column of every C-layout tile)
    const int tps = 2 * d_real;                                   // floats per token of tp (d_real <= D; D - d_real zero filled)
    const float* tpb = tp + (size_t)b * N * tps;
    const float* kpb = kp + (size_t)b * Nk * kstride;            // keys: phi of the same tokens (kp = tp + d_real, Nk = N) or
    const float* gTb = gT + (size_t)b * g_batch_rows * Np;        // the pooled phi / g of max_pool_factor > 1 (Nk < N)

    // query fragments: B operand, lane (q, kq) holds theta[q][16 i + 4 kq + s]
    f32x4 qf[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) {
        qf[i] = (q < N && 16 * i + 4 * kq < d_real) ? *reinterpret_cast<const f32x4*>(tpb + (size_t)q * tps + 16 * i + 4 * kq)
                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;                        // l_run: this lane's share (its kq keys) of the row sum

    const int ntiles = (Nk + BKV - 1) / BKV;
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * BKV;
        __syncthreads();                                          // every wave is done with the previous tiles
        stage_tile<BKV, QRK>(Ks, wave, lane,
                             [&](int row) { return key0 + row < Nk ? kpb + (size_t)(key0 + row) * kstride : (const float*)nullptr; },
                             [&](int quad) { return 4 * quad < d_real; });
        stage_tile<C2, QRV>(Vs, wave, lane, [&](int row) { return gTb + (size_t)row * Np + key0; },
                            [&](int quad) { return key0 + 4 * quad < Np; });
        __syncthreads();                                          // (waits for the DMA: vmcnt(0) + barrier)

        // ---- S^T = K . Q^T ----------------------------------------------------------------------------------------------------
        // (independent accumulators back to back: a dependent v_mfma_f32_16x16x4_f32 waits 40 cycles, an independent one 32)
        f32x4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            f32x4 kf[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int row = kt * 16 + r;
                kf[kt] = *reinterpret_cast<const f32x4*>(Ks + row * D + (((4 * i + kq) ^ (row & SWK)) << 2));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][e], qf[i][e], s[kt], 0, 0, 0);
        }
        // ---- online softmax over the keys of this tile ----------------------------------------------------------------------------
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (key0 + kt * 16 + 4 * kq + e >= Nk) s[kt][e] = -INFINITY;
                mx = fmaxf(mx, s[kt][e]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                     // finite: every tile holds at least one valid key
        const float alpha = __expf(m_run - m_new);                // 0 on the first tile (m_run = -inf)
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __expf(s[kt][e] - m_new);
                s[kt][e] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__any(alpha != 1.f)) {                                // the running maximum usually stops moving after a few tiles
#pragma unroll
            for (int c = 0; c < CT; ++c) o[c] *= alpha;
        }
        // ---- O^T += V . P^T -------------------------------------------------------------------------------------------------------
        constexpr int CG = CT < 4 ? CT : 4;
#pragma unroll
        for (int cg = 0; cg < CT; cg += CG) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x4 vf[CG];
#pragma unroll
                for (int cc = 0; cc < CG; ++cc) {
                    const int row = (cg + cc) * 16 + r;
                    vf[cc] = *reinterpret_cast<const f32x4*>(Vs + row * BKV + (((4 * kt + kq) ^ (row & SWV)) << 2));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int cc = 0; cc < CG; ++cc)
                        o[cg + cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[cc][e], s[kt][e], o[cg + cc], 0, 0, 0);
            }
        }
    }
    // row sums: the four kq lanes of a query hold disjoint key subsets
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_run;
    // log-sum-exp of the row's logits: the backward rebuilds the probabilities as exp(s - lse) in a GEMM epilogue (no softmax pass)
    if (lse != nullptr && q < N && kq == 0) lse[(size_t)b * N + q] = m_run + logf(l_run);
    if (q < N) {
        if (out_bf16) {             // bf16 storage mode (configs[4]): the o conv reads bf16
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            unsigned short* dst = reinterpret_cast<unsigned short*>(out) + ((size_t)b * N + q) * out_stride + 4 * kq;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const f32x4 v = o[c] * inv;
                *reinterpret_cast<bf16x4*>(dst + 16 * c) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            }
        } else {
            float* dst = out + ((size_t)b * N + q) * out_stride + 4 * kq;
#pragma unroll
            for (int c = 0; c < CT; ++c) *reinterpret_cast<f32x4*>(dst + 16 * c) = o[c] * inv;
        }
    }
}

template <int D, int C2, int BKV>
int launch(const float* tp, const float* kp, const float* gT, float* out, int B, int N, int Nk, int Np, int d_real, int kstride,
           int out_bf16, float* lse, hipStream_t stream, int out_stride = C2, int g_batch_rows = C2) {
    constexpr int smem = (BKV * D + C2 * BKV) * (int)sizeof(float);
    static unsigned attr_mask = 0;
    auto kern = flash_attn_kernel<D, C2, BKV>;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", smem);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask);
    }
    const int qtiles = (N + 63) / 64;
    hipLaunchKernelGGL(kern, dim3(B * qtiles), dim3(256), smem, stream, tp, kp, gT, out, N, Nk, Np, qtiles, d_real, kstride, out_bf16, lse,
                       out_stride, g_batch_rows);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}


// ---- bf16 storage mode (configs[4]): fp32 logits, bf16 values -------------------------------------------------------------
// theta / phi stay fp32 (a logit of magnitude ~50 rounded to bf16 would move its probability by tens of percent) and S^T = K . Q^T
// stays on the fp32 matrix cores; the value product -- 80 % of the block's FLOPs -- runs on v_mfma_f32_16x16x32_bf16 with P rounded
// to bf16 and g stored bf16.  The lane (q, kq) holds P for keys 16 kt + 4 kq + {0..3} of two adjacent 16-key tiles = the 8 k-values
// of one bf16 MFMA if that MFMA's k index is DEFINED as k = 8 kq + e  <->  key 32 t + 16 (e >> 2) + 4 kq + (e & 3); the producer
// conv writes g^T with the keys of every 32-block in that order (GSSD_CONV_OUTB_BF16_PERM32), so a lane's 8 values of V are one
// contiguous 16-byte ds_read_b128.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

template <int D, int C2, int BKV>
__global__ __launch_bounds__(256, 2) void flash_attn_mixed_kernel(const float* __restrict__ tp, const u16* __restrict__ gT,
                                                                  u16* __restrict__ out, int N, int Np32, int qtiles,
                                                                  float* __restrict__ lse, int out_stride, int g_batch_rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ks = smem;                                            // [BKV][D] fp32
    u16* const Vs = reinterpret_cast<u16*>(smem + BKV * D);            // [C2][BKV] bf16, keys permuted inside 32-blocks
    constexpr int QRK = D / 4, UV = BKV / 8;                           // 16-byte units per K row / per V row
    constexpr int SWK = (QRK < 16 ? QRK : 16) - 1, SWV = UV - 1;
    constexpr int KT = BKV / 16, KB = BKV / 32, CT = C2 / 16, DI = D / 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x / qtiles, qt = blockIdx.x - b * qtiles;
    const int q = qt * 64 + wave * 16 + r;
    const float* tpb = tp + (size_t)b * N * (2 * D);
    const u16* gTb = gT + (size_t)b * g_batch_rows * Np32;

    f32x4 qf[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i)
        qf[i] = (q < N) ? *reinterpret_cast<const f32x4*>(tpb + (size_t)q * (2 * D) + 16 * i + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (N + BKV - 1) / BKV;
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * BKV;
        __syncthreads();
        stage_tile<BKV, QRK>(Ks, wave, lane,
                             [&](int row) { return key0 + row < N ? tpb + (size_t)(key0 + row) * (2 * D) + D : (const float*)nullptr; },
                             [&](int) { return true; });
        {   // V tile: C2 rows of BKV bf16 (UV 16-byte units), unit' = unit ^ (row & SWV)
            constexpr int RPP = 64 / UV, PIECES = C2 / RPP;
            const int row_in = lane / UV, slot = lane % UV;
#pragma unroll
            for (int p0 = 0; p0 < PIECES; p0 += 4) {
                const int piece = p0 + wave;
                const int row = piece * RPP + row_in;
                const int unit = slot ^ (row & SWV);
                const bool ok = key0 + 8 * unit < Np32;
                const float* src = ok ? reinterpret_cast<const float*>(gTb + (size_t)row * Np32 + key0 + 8 * unit) : g_zero16;
                dma16(src, reinterpret_cast<float*>(Vs) + piece * 256);
            }
        }
        __syncthreads();

        f32x4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            f32x4 kf[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int row = kt * 16 + r;
                kf[kt] = *reinterpret_cast<const f32x4*>(Ks + row * D + (((4 * i + kq) ^ (row & SWK)) << 2));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][e], qf[i][e], s[kt], 0, 0, 0);
        }
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
        bf16x8 pb[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 p = (__bf16)__expf(s[2 * kb + (e >> 2)][e & 3] - m_new);
                pb[kb][e] = p;
                psum += (float)p;                    // the denominator sums the ROUNDED probabilities the numerator uses
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int c = 0; c < CT; ++c) o[c] *= alpha;
        }
        constexpr int CG = CT < 4 ? CT : 4;
#pragma unroll
        for (int cg = 0; cg < CT; cg += CG) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                bf16x8 vf[CG];
#pragma unroll
                for (int cc = 0; cc < CG; ++cc) {
                    const int row = (cg + cc) * 16 + r;
                    vf[cc] = *reinterpret_cast<const bf16x8*>(Vs + row * BKV + (((4 * kb + kq) ^ (row & SWV)) << 3));
                }
#pragma unroll
                for (int cc = 0; cc < CG; ++cc) o[cg + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[cc], pb[kb], o[cg + cc], 0, 0, 0);
            }
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_run;
    if (lse != nullptr && q < N && kq == 0) lse[(size_t)b * N + q] = m_run + logf(l_run);       // for the training step's backward
    if (q < N) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        u16* dst = out + ((size_t)b * N + q) * out_stride + 4 * kq;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const f32x4 v = o[c] * inv;
            *reinterpret_cast<bf16x4*>(dst + 16 * c) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
    }
}

template <int D, int C2, int BKV>
int launch_mixed(const float* tp, const u16* gT, u16* out, int B, int N, int Np32, float* lse, hipStream_t stream, int out_stride = C2,
                 int g_batch_rows = C2) {
    constexpr int smem = BKV * D * (int)sizeof(float) + C2 * BKV * (int)sizeof(u16);
    static unsigned attr_mask = 0;
    auto kern = flash_attn_mixed_kernel<D, C2, BKV>;
    if (gssd_attr_needed(&attr_mask) &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", smem);
        return GSSD_ELAUNCH;
    }
    gssd_attr_done(&attr_mask);
    const int qtiles = (N + 63) / 64;
    hipLaunchKernelGGL(kern, dim3(B * qtiles), dim3(256), smem, stream, tp, gT, out, N, Np32, qtiles, lse, out_stride, g_batch_rows);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

extern "C" int gssd_self_attn_core_kv_f32(const float* tp, const float* kp, const float* gT, void* out_v, int B, int N, int Nk, int Nkp,
                                          int D, int C2, int kstride, int out_bf16, float* lse, gssd_stream_t stream) {
    float* out = reinterpret_cast<float*>(out_v);
    GSSD_CHECK_ARG(tp && kp && gT && out && B > 0 && N > 0 && Nk > 0 && Nkp >= Nk && Nkp % 4 == 0);
    GSSD_CHECK_ARG(((uintptr_t)tp % 16) == 0 && ((uintptr_t)kp % 16) == 0 && ((uintptr_t)gT % 16) == 0 && ((uintptr_t)out % 16) == 0);
    GSSD_CHECK_ARG((long long)B * ((N + 63) / 64) < (1ll << 31));
    hipStream_t s = as_stream(stream);
    GSSD_CHECK_ARG(D > 0 && D % 4 == 0 && C2 > 0 && kstride >= D && kstride % 4 == 0);
    if (D == 64 && C2 == 256) return launch<64, 256, 64>(tp, kp, gT, out, B, N, Nk, Nkp, D, kstride, out_bf16, lse, s);
    if (D == 128 && C2 == 512) return launch<128, 512, 32>(tp, kp, gT, out, B, N, Nk, Nkp, D, kstride, out_bf16, lse, s);
    if (D == 256 && C2 == 1024) {           // --feature_scale 2 on the 2048-channel map: two launches of 512 g channels each
        const size_t esz = out_bf16 ? 2 : 4;
        const int rc = launch<256, 512, 32>(tp, kp, gT, out, B, N, Nk, Nkp, D, kstride, out_bf16, lse, s, 1024, 1024);
        if (rc != GSSD_OK) return rc;
        return launch<256, 512, 32>(tp, kp, gT + (size_t)512 * Nkp, reinterpret_cast<float*>(reinterpret_cast<char*>(out) + 512 * esz), B, N, Nk,
                                    Nkp, D, kstride, out_bf16, nullptr, s, 1024, 1024);
    }
    if (D == 32 && C2 == 128) return launch<32, 128, 64>(tp, kp, gT, out, B, N, Nk, Nkp, D, kstride, out_bf16, lse, s);
    if (D <= 16 && C2 == 32) return launch<16, 32, 64>(tp, kp, gT, out, B, N, Nk, Nkp, D, kstride, out_bf16, lse, s);     // small maps (Self_Attn(64): op-level tests)
    if (D <= 16 && C2 == 64) return launch<16, 64, 64>(tp, kp, gT, out, B, N, Nk, Nkp, D, kstride, out_bf16, lse, s);
    gssd_set_error("self-attention core: unsupported (theta/phi channels %d, g channels %d); built: (64,256) (128,512) (32,128) (<=16,32|64)", D, C2);
    return GSSD_EINVAL;
}

extern "C" int gssd_self_attn_core_f32(const float* tp, const float* gT, void* out_v, int B, int N, int Np, int D, int C2,
                                       int out_bf16, gssd_stream_t stream) {
    GSSD_CHECK_ARG(tp && D > 0);
    return gssd_self_attn_core_kv_f32(tp, tp + D, gT, out_v, B, N, N, Np, D, C2, 2 * D, out_bf16, nullptr, stream);   // keys = phi of the same tokens
}

extern "C" int gssd_self_attn_core_bf16v(const float* tp, const void* gT_bf16, void* out_bf16, int B, int N, int Np32, int D, int C2,
                                         float* lse, gssd_stream_t stream) {
    GSSD_CHECK_ARG(tp && gT_bf16 && out_bf16 && B > 0 && N > 0 && Np32 >= N && Np32 % 32 == 0);
    GSSD_CHECK_ARG(((uintptr_t)tp % 16) == 0 && ((uintptr_t)gT_bf16 % 16) == 0 && ((uintptr_t)out_bf16 % 8) == 0);
    GSSD_CHECK_ARG((long long)B * ((N + 63) / 64) < (1ll << 31));
    hipStream_t s = as_stream(stream);
    const u16* g = reinterpret_cast<const u16*>(gT_bf16);
    u16* o = reinterpret_cast<u16*>(out_bf16);
    if (D == 64 && C2 == 256) return launch_mixed<64, 256, 64>(tp, g, o, B, N, Np32, lse, s);
    if (D == 128 && C2 == 512) return launch_mixed<128, 512, 32>(tp, g, o, B, N, Np32, lse, s);
    if (D == 256 && C2 == 1024) {           // two launches of 512 g channels each (see gssd_self_attn_core_kv_f32)
        const int rc = launch_mixed<256, 512, 32>(tp, g, o, B, N, Np32, lse, s, 1024, 1024);
        if (rc != GSSD_OK) return rc;
        return launch_mixed<256, 512, 32>(tp, g + (size_t)512 * Np32, o + 512, B, N, Np32, nullptr, s, 1024, 1024);
    }
    if (D == 32 && C2 == 128) return launch_mixed<32, 128, 64>(tp, g, o, B, N, Np32, lse, s);
    gssd_set_error("self-attention core (bf16 values): unsupported (theta/phi channels %d, g channels %d)", D, C2);
    return GSSD_EINVAL;
}
