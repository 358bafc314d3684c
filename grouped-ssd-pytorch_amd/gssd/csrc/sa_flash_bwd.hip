// Flash-style backward of Self_Attn's attention core (layers/self_attn.py:68-80: attn = softmax(theta phi^T), attn_g = attn g) for the
// training step of the bf16 storage mode on gfx950: no [N][N] map is materialised (the explicit path of gssd/backward.py::_sa writes
// and re-reads two 267 MB maps per 38 x 38 block at batch 32 and spends 1.5 ms in five batched fp32 GEMMs).
//
// With S = theta phi^T, P = exp(S - lse) (lse: the rows' log-sum-exp kept by the forward), dP = d(attn_g) g^T, D_i = <d(attn_g)_i, attn_g_i>,
// dS = P o (dP - D):   d g = P^T d(attn_g),   d phi = dS^T theta,   d theta = dS phi.
// Two launches of ONE kernel template, no atomics: the workgroup OWNS 64 tokens (their accumulators live in registers for the whole
// kernel) and STREAMS the other side in blocks of 32 tokens through LDS:
//   OWN_KEYS = true : owns keys j   -> d g_j, d phi_j;  streams the queries  (theta, d(attn_g), lse, D)
//   OWN_KEYS = false: owns queries i -> d theta_i;      streams the keys     (phi, g); S and dP are recomputed (45 % more MFMA work than a
//                     one-pass scheme with atomics on d theta -- and deterministic).
// Every tile is computed "streamed rows x own columns": S_tile = (streamed fp32 [16][D]) x (own fp32 [16][D])^T on v_mfma_f32_16x16x4_f32
// (a bf16 logit would move its probability by tens of per cent: the one fp32 product, and 2/3 of the kernel's MFMA time), dP_tile on
// v_mfma_f32_16x16x32_bf16 from the natural [token][channel] bf16 images.  The tile's lane layout (column = own token, 4 streamed rows per
// lane) IS the B-operand layout of the accumulating products acc^T[channel][own token] += X^T[channel][streamed] . tile[streamed][own]:
// two tiles are packed into one bf16 k = 32 operand, and X^T comes out of the natural LDS image through ds_read_b64_tr_b16 (the same
// transpose-read scheme as csrc/conv_wgrad_bf16.hip).  Accumulators leave as 16-byte rows of the token-major [theta | phi | g] gradient.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

namespace {

__device__ __attribute__((aligned(16))) unsigned g_zero_page_fb[4] = {0, 0, 0, 0};

__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ s16x4 tr_read(const u16* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_ptr);
}
__device__ __forceinline__ bf16x8 tr_pair(const u16* p0, const u16* p1) {
    const s16x4 lo = tr_read(p0), hi = tr_read(p1);
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct FbParams {
    const float* tp;       // [B][N][2D] fp32: theta | phi
    const u16* tp16;       // the same, bf16: hi = bf16(x)
    const u16* tp16lo;     // lo = bf16(x - hi): theta phi^T is recomputed as hi.hi + hi.lo + lo.hi on the bf16 matrix cores (NULL: fp32 MFMA)
    const u16* g16;        // [B][N][C2] bf16, token-major
    const u16* dag16;      // [B][N][C2] bf16: d(attn_g)
    const float* lse;      // [B][N]
    const float* dvec;     // [B][N]: <d(attn_g)_i, attn_g_i>
    float* dtpg;           // [B][N][2D + C2] fp32: d theta | d phi | d g
    int N, own_blocks;
};

constexpr int OWN = 64, STR = 32;

// X3: the logits from the two-term bf16 split of theta / phi -- three v_mfma_f32_16x16x32_bf16 per 32 channels instead of eight
// v_mfma_f32_16x16x4_f32 (1/5 of the matrix-pipe time; the dropped lo.lo term is 2^-16 of a product: a logit of magnitude 50 moves by
// < 1e-3, its probability by < 0.1 % -- a quarter of the bf16 rounding P gets anyway)
template <int D, int C2, bool OWN_KEYS, bool X3>
__global__ __launch_bounds__(256, (D >= 128 ? 1 : 2)) void sa_flash_bwd_kernel(const FbParams p) {      // (128, 512): 256 accumulator +
    // operand registers per lane -- one workgroup per CU with the unified 512-register file
    constexpr int DI = D / 16, CS = C2 / 32, CT = C2 / 16, CTOT = 2 * D + C2;
    constexpr int UF = D / 4;                    // 16-byte units per fp32 row
    constexpr int UD = D / 8, NCD = D / 16;      // 16-byte units / 32-byte chunks per bf16 [D] row
    constexpr int UC = C2 / 8, NCC = C2 / 16;    // the same per bf16 [C2] row
    static_assert(UF == 32 || UF == 16 || UF == 8, "D = 128, 64 or 32");
    constexpr int XF = (UF < 16 ? UF : 16) - 1;      // fp32 rows: 16-byte units XOR-swizzled in their low four bits
    static_assert(NCC >= 8 && UC % 16 == 0, "C2 >= 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fb[];
    float* Sf = reinterpret_cast<float*>(smem_fb);                       // [STR][D] fp32, units XOR (t & XF)
    u16* Sd = reinterpret_cast<u16*>(Sf + STR * D);                     // [STR][D] bf16 for transpose reads, chunks XOR swz
    u16* Sl = reinterpret_cast<u16*>(Sf);                               // X3: the lo image [STR][D] bf16 in place of the fp32 one (same layout as Sd)
    u16* An = Sd + STR * D;                                             // [STR][C2] bf16, natural 16-byte reads: units XOR (t & 15)
    u16* At = An + STR * C2;                                            // [STR][C2] bf16, transpose reads: chunks XOR (t & 7)   (OWN_KEYS)
    float* ls = reinterpret_cast<float*>(OWN_KEYS ? At + STR * C2 : At);    // [STR] lse, [STR] D of the streamed queries          (OWN_KEYS)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int b = blockIdx.x / p.own_blocks, ob = blockIdx.x - b * p.own_blocks;
    const int N = p.N;
    const int own_tok = ob * OWN + wave * 16 + r;                        // this lane's own token (a column of every tile)
    const bool own_ok = own_tok < N;
    const size_t bN = (size_t)b * N;
    const float* own_f = p.tp + bN * (2 * D) + (OWN_KEYS ? D : 0);      // phi (keys) or theta (queries)
    const float* str_f = p.tp + bN * (2 * D) + (OWN_KEYS ? 0 : D);
    const u16* str_d = p.tp16 + bN * (2 * D) + (OWN_KEYS ? 0 : D);
    const u16* str_l = X3 ? p.tp16lo + bN * (2 * D) + (OWN_KEYS ? 0 : D) : nullptr;
    const u16* own_c = (OWN_KEYS ? p.g16 : p.dag16) + bN * C2;
    const u16* str_c = (OWN_KEYS ? p.dag16 : p.g16) + bN * C2;
    const void* zero = g_zero_page_fb;

    // ---- the own side: operand fragments in registers for the whole kernel -----------------------------------------------------------
    constexpr int DS = D / 32;
    f32x4 kown[X3 ? 1 : DI];
    bf16x8 kh[X3 ? DS : 1], kl[X3 ? DS : 1];       // X3: lane (own token r, kq) holds channels 32 s + 8 kq .. + 7, split hi / lo
    bf16x8 vown[CS];
    if constexpr (!X3) {
#pragma unroll
        for (int i = 0; i < DI; ++i)
            kown[i] = own_ok ? *reinterpret_cast<const f32x4*>(own_f + (size_t)own_tok * (2 * D) + 16 * i + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, bq = {0.f, 0.f, 0.f, 0.f};
            if (own_ok) {
                a = *reinterpret_cast<const f32x4*>(own_f + (size_t)own_tok * (2 * D) + 32 * s + 8 * kq);
                bq = *reinterpret_cast<const f32x4*>(own_f + (size_t)own_tok * (2 * D) + 32 * s + 8 * kq + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 h0 = (__bf16)a[e], h1 = (__bf16)bq[e];
                kh[s][e] = h0;
                kh[s][4 + e] = h1;
                kl[s][e] = (__bf16)(a[e] - (float)h0);
                kl[s][4 + e] = (__bf16)(bq[e] - (float)h1);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < CS; ++s) {
        if (own_ok) vown[s] = *reinterpret_cast<const bf16x8*>(own_c + (size_t)own_tok * C2 + 32 * s + 8 * kq);
        else vown[s] = __builtin_bit_cast(bf16x8, (s16x8){0, 0, 0, 0, 0, 0, 0, 0});
    }
    float lse_own = 0.f, d_own = 0.f;
    if (!OWN_KEYS && own_ok) {
        lse_own = p.lse[bN + own_tok];
        d_own = p.dvec[bN + own_tok];
    }

    f32x4 accD[DI];
    f32x4 accV[OWN_KEYS ? CT : 1];
#pragma unroll
    for (int i = 0; i < DI; ++i) accD[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (OWN_KEYS ? CT : 1); ++i) accV[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transpose-read suppliers: row j = (lane & 15) >> 2 of the group's 4 rows, 8-byte piece q = lane & 3
    const int sj = (lane & 15) >> 2, sq = lane & 3;
    int trd[2], trc[2];                          // element offsets (chunk 0) of the two reads: streamed tokens 4 kq + sj and 16 + 4 kq + sj
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tt = 16 * h + 4 * kq + sj;
        trd[h] = tt * D + ((NCD >= 8 ? (tt & 7) : NCD == 4 ? ((tt >> 1) & 3) : ((tt >> 2) & 1)) << 4) + sq * 4;
        trc[h] = tt * C2 + ((tt & 7) << 4) + sq * 4;
    }

    const int nblk = (N + STR - 1) / STR;
    for (int blk = 0; blk < nblk; ++blk) {
        const int s0 = blk * STR;
        // ---- stage the streamed block -------------------------------------------------------------------------------------------------
        {
            constexpr int RF = 64 / UF;                                  // rows per DMA piece
            if constexpr (!X3)
            for (int i = wave; i < STR / RF; i += 4) {
                const int t = i * RF + lane / UF, u = lane % UF;
                const int tok = s0 + t;
                const void* src = tok < N ? (const void*)(str_f + (size_t)tok * (2 * D) + 4 * (u ^ (t & XF))) : zero;
                dma16(src, Sf + i * RF * D);
            }
            constexpr int RD = 64 / UD;
            for (int i = wave; i < STR / RD; i += 4) {
                const int t = i * RD + lane / UD, u = lane % UD;
                const int tok = s0 + t;
                const int sw = NCD >= 8 ? (t & 7) : NCD == 4 ? ((t >> 1) & 3) : ((t >> 2) & 1);
                const size_t so = (size_t)tok * (2 * D) + (((u >> 1) ^ sw) << 4) + ((u & 1) << 3);
                dma16(tok < N ? (const void*)(str_d + so) : zero, Sd + i * RD * D);
                if constexpr (X3) dma16(tok < N ? (const void*)(str_l + so) : zero, Sl + i * RD * D);
            }
            constexpr int RC = 64 / UC;                                  // rows per piece
            static_assert(UC <= 64, "C2 <= 512");
            for (int i = wave; i < STR / RC; i += 4) {
                const int t = i * RC + lane / UC, u = lane % UC;
                const int tok = s0 + t;
                const void* src = tok < N ? (const void*)(str_c + (size_t)tok * C2 + ((u ^ (t & 15)) << 3)) : zero;
                dma16(src, An + i * RC * C2);
                if (OWN_KEYS) {
                    const void* src2 = tok < N ? (const void*)(str_c + (size_t)tok * C2 + (((u >> 1) ^ (t & 7)) << 4) + ((u & 1) << 3)) : zero;
                    dma16(src2, At + i * RC * C2);
                }
            }
            if (OWN_KEYS && tid < 2 * STR) {
                const int t = tid & (STR - 1);
                const int tok = s0 + t;
                float v = 0.f;
                if (tok < N) v = tid < STR ? p.lse[bN + tok] : p.dvec[bN + tok];
                ls[tid] = v;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();

        // ---- two tiles: streamed rows 16 h + 4 kq + e, own column r ------------------------------------------------------------------
        f32x4 Pt[2], dSt[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = 16 * h + r;                                    // the row this lane FEEDS (A operand); its results are rows 4 kq + e
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!X3) {
#pragma unroll
                for (int i = 0; i < DI; ++i) {
                    const f32x4 qf = *reinterpret_cast<const f32x4*>(Sf + t * D + (((4 * i + kq) ^ (t & XF)) << 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[e], kown[i][e], s, 0, 0, 0);
                }
            } else {
                const int swt = NCD >= 8 ? (t & 7) : NCD == 4 ? ((t >> 1) & 3) : ((t >> 2) & 1);
#pragma unroll
                for (int q = 0; q < DS; ++q) {
                    const int U = 4 * q + kq;                         // 16-byte unit of the row: channels 32 q + 8 kq ..
                    const int off = t * D + (((U >> 1) ^ swt) << 4) + ((U & 1) << 3);
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Sd + off);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(Sl + off);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, kh[q], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, kl[q], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, kh[q], s, 0, 0, 0);
                }
            }
            f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(An + t * C2 + (((4 * c + kq) ^ (t & 15)) << 3));
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, vown[c], dp, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tr = 16 * h + 4 * kq + e;
                const float l = OWN_KEYS ? ls[tr] : lse_own;
                const float dd = OWN_KEYS ? ls[STR + tr] : d_own;
                const float pe = (s0 + tr < N) ? __expf(s[e] - l) : 0.f;
                Pt[h][e] = pe;
                dSt[h][e] = pe * (dp[e] - dd);
            }
        }
        bf16x8 Pb, dSb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Pb[e] = (__bf16)Pt[0][e];
            Pb[4 + e] = (__bf16)Pt[1][e];
            dSb[e] = (__bf16)dSt[0][e];
            dSb[4 + e] = (__bf16)dSt[1][e];
        }
        // ---- accumulate: acc^T[channel][own] += X^T[channel][streamed 32] . tile[streamed 32][own] --------------------------------------
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            const bf16x8 a = tr_pair(Sd + (trd[0] ^ (i << 4)), Sd + (trd[1] ^ (i << 4)));
            accD[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dSb, accD[i], 0, 0, 0);
        }
        if constexpr (OWN_KEYS) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const bf16x8 a = tr_pair(At + (trc[0] ^ (c << 4)), At + (trc[1] ^ (c << 4)));
                accV[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Pb, accV[c], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- store: rows 4 kq + e of tile i = channels 16 i + 4 kq + e of the own token -------------------------------------------------------
    if (own_ok) {
        float* dst = p.dtpg + (bN + own_tok) * CTOT;
#pragma unroll
        for (int i = 0; i < DI; ++i) *reinterpret_cast<f32x4*>(dst + (OWN_KEYS ? D : 0) + 16 * i + 4 * kq) = accD[i];
        if constexpr (OWN_KEYS) {
#pragma unroll
            for (int c = 0; c < CT; ++c) *reinterpret_cast<f32x4*>(dst + 2 * D + 16 * c + 4 * kq) = accV[c];
        }
    }
}

template <int D, int C2>
int launch_fb(const FbParams& p0, int B, hipStream_t stream) {
    FbParams p = p0;
    p.own_blocks = (p.N + OWN - 1) / OWN;
    const size_t base = (size_t)STR * D * 4 + (size_t)STR * D * 2 + (size_t)STR * C2 * 2;
    const size_t smem_k = base + (size_t)STR * C2 * 2 + 2 * STR * sizeof(float), smem_q = base;
    const bool x3 = p.tp16lo != nullptr;
    auto kk = x3 ? sa_flash_bwd_kernel<D, C2, true, true> : sa_flash_bwd_kernel<D, C2, true, false>;
    auto kq = x3 ? sa_flash_bwd_kernel<D, C2, false, true> : sa_flash_bwd_kernel<D, C2, false, false>;
    static unsigned attr_mask[2] = {0, 0};
    unsigned* am = &attr_mask[x3 ? 1 : 0];
    if (gssd_attr_needed(am)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_k) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_q) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (attention backward)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(am);
    hipLaunchKernelGGL(kk, dim3(B * p.own_blocks), dim3(256), smem_k, stream, p);
    GSSD_CHECK_LAUNCH();
    hipLaunchKernelGGL(kq, dim3(B * p.own_blocks), dim3(256), smem_q, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

extern "C" int gssd_self_attn_flash_bwd_supported(int D, int C2) {
    return (D == 64 && C2 == 256) || (D == 32 && C2 == 128) || (D == 128 && C2 == 512) ? 1 : 0;
}

extern "C" int gssd_self_attn_flash_bwd_bf16(const float* tp, const void* tp_bf16, const void* tp_bf16_lo, const void* g_bf16,
                                             const void* dag_bf16, const float* lse, const float* dvec, float* dtpg, int B, int N, int D,
                                             int C2, gssd_stream_t stream) {
    GSSD_CHECK_ARG(tp && tp_bf16 && g_bf16 && dag_bf16 && lse && dvec && dtpg && B > 0 && N > 0);
    GSSD_CHECK_ARG(((uintptr_t)tp % 16) == 0 && ((uintptr_t)tp_bf16 % 16) == 0 && ((uintptr_t)g_bf16 % 16) == 0 &&
                   ((uintptr_t)dag_bf16 % 16) == 0 && ((uintptr_t)dtpg % 16) == 0);
    GSSD_CHECK_ARG((long long)B * N * (2 * D + C2) < (1ll << 31) && (long long)B * ((N + OWN - 1) / OWN) < (1ll << 31));
    FbParams p;
    p.tp = tp;
    p.tp16 = reinterpret_cast<const u16*>(tp_bf16);
    p.tp16lo = reinterpret_cast<const u16*>(tp_bf16_lo);
    p.g16 = reinterpret_cast<const u16*>(g_bf16);
    p.dag16 = reinterpret_cast<const u16*>(dag_bf16);
    p.lse = lse;
    p.dvec = dvec;
    p.dtpg = dtpg;
    p.N = N;
    p.own_blocks = 0;
    hipStream_t s = as_stream(stream);
    if (D == 64 && C2 == 256) return launch_fb<64, 256>(p, B, s);
    if (D == 32 && C2 == 128) return launch_fb<32, 128>(p, B, s);
    if (D == 128 && C2 == 512) return launch_fb<128, 512>(p, B, s);
    gssd_set_error("attention backward (flash form): unsupported (theta/phi channels %d, g channels %d)", D, C2);
    return GSSD_EINVAL;
}
