// bf16 implicit-GEMM convolution for gfx950 (BASELINE.json configs[4]: bf16 weights / activations, bf16 MFMA, fp32 accumulate):
// the same launch contract as conv_igemm.hip (struct gssd_conv_desc) with bf16 NHWC activations and bf16 K-major weights.
//
//   v_mfma_f32_16x16x32_bf16: a lane's operand is 8 consecutive bf16 = 16 bytes = ONE ds_read_b128 per 16x16 tile per 32 k.
//   Tiles are staged by 16-byte LDS-DMA exactly like the fp32 kernel (64 bf16 = 128 B per row, source-side XOR swizzle).
//   Operand roles are SWAPPED (A = weight rows, B = pixels): the C layout then puts a PIXEL in lane & 15 and four weight rows in the
//   lane's registers, and the weight rows of a 32-channel block are staged in the order
//       LDS row (tile j, rho) -> channel 32*(j>>1) + 8*(rho>>2) + 4*(j&1) + (rho&3)
//   so a lane holds EIGHT CONSECUTIVE channels of its pixel across a tile pair: one 16-byte NHWC store (2-byte scattered stores
//   would waste the write path of what are HBM-bound layers in bf16).  BatchNorm batch sums come from the fp32 accumulators
//   (before rounding); everything after the accumulator (1/sigma, bias, gate, residual) is fp32, rounded once on store.
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

namespace {

constexpr int BK = 64;          // bf16 per tile row = 128 B = 8 DMA lanes

__device__ __attribute__((aligned(16))) u16 g_zero_page_h[8] = {0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BN>            // (instantiated with 16 where a wave owns ONE 16-channel tile: no pairing, identity)
__device__ __forceinline__ int chan_of_row(int row) {        // LDS row of the weight tile -> output channel inside the BN tile
    if (BN < 32) return row;
    const int j = row >> 4, rho = row & 15;
    return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
}

__device__ __forceinline__ float bf2f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }

// (csrc/conv_igemm.hip: the counted wait of the NSTG >= 3 K loop -- no fence, the ring's younger pieces stay in flight)
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// NSTG: LDS stages of the K loop; 3 = the small-map form (32- / 64-row tiles, two chunks in flight), see csrc/conv_igemm.hip
template <int BM, int BN, int WM, int WN, int NSTG>
__global__ __launch_bounds__(WM * WN * 64) void conv_bf16_kernel(const gssd_conv_desc p, const int M, const int tiles_per_group) {
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int NW = WM * WN, NTHR = NW * 64;
    constexpr int AR = BM / (8 * NW);
    constexpr int BPIECES = BN / 8;
    constexpr int BR = (BPIECES + NW - 1) / NW;
    constexpr int STAGE = (BM + BN) * BK;             // u16 elements
    static_assert(NT == 1 || NT % 2 == 0, "channel pairing needs an even tile count per wave");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, kq = lane >> 4;
    const int ny = p.groups * tiles_per_group;
    int mt, by;
    if (p.m_per_image) {
        mt = blockIdx.x;
        by = blockIdx.y;
    } else {
        const int slot = blockIdx.x >> 3;
        mt = (slot / ny) * 8 + (blockIdx.x & 7);
        by = slot % ny;
    }
    if (mt * BM >= M) return;
    const int g = by / tiles_per_group;
    const int n0g = (by % tiles_per_group) * BN;
    const int cout_g = p.Cout / p.groups;
    const int m0 = mt * BM;
    const int img = p.m_per_image ? blockIdx.z : 0;
    const int kz = p.m_per_image ? 0 : blockIdx.z;
    const int HoWo = p.Ho * p.Wo;
    const int K = p.K;
    const int taps = p.KH * p.KW;

    const u16* __restrict__ in = reinterpret_cast<const u16*>(p.in) + (size_t)img * p.in_batch_stride + p.in_ch_off + g * p.cin_g;
    const u16* __restrict__ wgt =
        reinterpret_cast<const u16*>(p.wgt) + (size_t)img * p.wgt_batch_stride + (size_t)(g * cout_g) * p.wgt_row_stride;
    const u16* zero = g_zero_page_h;
    const bool xf = p.in_scale != nullptr;
    float* xtab = reinterpret_cast<float*>(smem_h + NSTG * STAGE);       // [2][cin_g]: scale | shift
    if (xf) {
        for (int c = tid; c < p.cin_g; c += NTHR) {
            xtab[c] = p.in_scale[p.in_ch_off + g * p.cin_g + c];
            xtab[p.cin_g + c] = p.in_shift[p.in_ch_off + g * p.cin_g + c];
        }
    }
    // out-of-image taps of a fused-BatchNorm input read a per-channel pad value the transform maps to 0 (bf16 copy of in_pad)
    const u16* padp = xf ? reinterpret_cast<const u16*>(p.in_pad) + p.in_ch_off + g * p.cin_g : nullptr;

    const int row_in = lane >> 3;
    const int lq = (lane & 7) ^ row_in;              // logical oct (k = 8*lq within the chunk)
    int a_iy0[AR], a_ix0[AR], a_off[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int m = m0 + (j * NW + wave) * 8 + row_in;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        int b = 0, pix = mm;
        if (!p.m_per_image) {
            b = mm / HoWo;
            pix = mm - b * HoWo;
        }
        const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
        a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 20);
        a_ix0[j] = ox * p.stride - p.pad;
        a_off[j] = ((b * p.H + oy * p.stride - p.pad) * p.W + a_ix0[j]) * p.in_stride;
    }
    int b_off[BR];
    bool b_ok[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int row = (j * NW + wave) * 8 + row_in;
        const int ch = chan_of_row<(NT == 1 ? 16 : BN)>(row);
        b_ok[j] = (j * NW + wave) < BPIECES && (n0g + ch) < cout_g;
        b_off[j] = (n0g + ch) * p.wgt_row_stride + 8 * lq;
    }
    const int nchunks_all = (K + BK - 1) / BK;
    const int cps = (nchunks_all + p.split_k - 1) / p.split_k;
    const int ch_begin = kz * cps;
    const int ch_end = min(nchunks_all, ch_begin + cps);
    int a_tap = (ch_begin * BK + 8 * lq) / p.cin_g;
    int a_c = (ch_begin * BK + 8 * lq) - a_tap * p.cin_g;

    auto issue = [&](int chunk, int buf) {
        u16* As = smem_h + buf * STAGE;
        u16* Bs = As + BM * BK;
        const int ty = a_tap / p.KW, tx = a_tap - ty * p.KW;
        const int dy = ty * p.dil, dx = tx * p.dil;
        const bool tap_ok = a_tap < taps;
        const int toff = (dy * p.W + dx) * p.in_stride + a_c;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const bool ok = tap_ok && (unsigned)(a_iy0[j] + dy) < (unsigned)p.H && (unsigned)(a_ix0[j] + dx) < (unsigned)p.W;
            const u16* src = ok ? in + (a_off[j] + toff) : ((xf && tap_ok) ? padp + a_c : zero);
            dma16(src, As + (j * NW + wave) * 8 * BK);
        }
        const int k0 = chunk * BK;
        const bool kok = k0 + 8 * lq < K;
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            if ((j * NW + wave) < BPIECES) {
                const u16* src = (b_ok[j] && kok) ? wgt + (b_off[j] + k0) : zero;
                dma16(src, Bs + (j * NW + wave) * 8 * BK);
            }
        }
        a_c += BK;
        while (a_c >= p.cin_g) {
            a_c -= p.cin_g;
            ++a_tap;
        }
    };

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    // fragment read offsets (u16 elements) inside a stage: row (base + r), physical 16-byte slot (4*ks + kq) ^ (r & 7)
    const int fo0 = r * BK + ((kq ^ (r & 7)) << 3);
    const int fo1 = r * BK + (((4 + kq) ^ (r & 7)) << 3);

    int f_c0 = (ch_begin * BK + 8 * kq) % p.cin_g, f_c1 = (ch_begin * BK + 32 + 8 * kq) % p.cin_g;
    static_assert(NSTG == 2 || BPIECES % NW == 0, "counted waits need the same number of DMA pieces per chunk in every wave");
    constexpr int PER = AR + BR;                     // DMA pieces per chunk and wave
    if constexpr (NSTG == 2) {
        if (ch_begin < ch_end) issue(ch_begin, 0);
        __syncthreads();
    } else {
#pragma unroll
        for (int st = 0; st < NSTG - 1; ++st)
            if (ch_begin + st < ch_end) issue(ch_begin + st, st);
        if (xf) __syncthreads();
    }
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        int buf;
        if constexpr (NSTG == 2) {
            buf = (ch - ch_begin) & 1;
            if (ch + 1 < ch_end) issue(ch + 1, buf ^ 1);
        } else {
            buf = (ch - ch_begin) % NSTG;
            if (NSTG >= 4 && ch + 2 < ch_end) wait_vm_barrier<2 * PER>();
            else if (ch + 1 < ch_end) wait_vm_barrier<PER>();
            else wait_vm_barrier<0>();
            if (ch + NSTG - 1 < ch_end) issue(ch + NSTG - 1, (ch - ch_begin + NSTG - 1) % NSTG);
        }
        const u16* As = smem_h + buf * STAGE + wm * WTM * BK;
        const u16* Bs = smem_h + buf * STAGE + BM * BK + wn * WTN * BK;
        bf16x8 af[2][MT], bf[2][NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks ? fo1 : fo0;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(As + i * 16 * BK + fo);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[ks][j] = *reinterpret_cast<const bf16x8*>(Bs + j * 16 * BK + fo);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (xf) {
                // fused producer BatchNorm + ReLU: max(x*scale[c] + shift[c], 0) in fp32, rounded to bf16 (the same rounding the
                // separate BN pass applies when it stores the activation)
                const int fc = ks ? f_c1 : f_c0;
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(xtab + fc), sc1 = *reinterpret_cast<const f32x4*>(xtab + fc + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(xtab + p.cin_g + fc),
                            sh1 = *reinterpret_cast<const f32x4*>(xtab + p.cin_g + fc + 4);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    bf16x8 v = af[ks][i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (__bf16)fmaxf((float)v[e] * sc0[e] + sh0[e], 0.f);
                        v[e + 4] = (__bf16)fmaxf((float)v[e + 4] * sc1[e] + sh1[e], 0.f);
                    }
                    af[ks][i] = v;
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
        }
        if (xf) {
            f_c0 += BK;
            while (f_c0 >= p.cin_g) f_c0 -= p.cin_g;
            f_c1 += BK;
            while (f_c1 >= p.cin_g) f_c1 -= p.cin_g;
        }
        if constexpr (NSTG == 2) __syncthreads();
    }
    if constexpr (NSTG != 2) __syncthreads();       // the epilogue reuses the stages

    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    // acc[i][j][e]: pixel m = m0 + wm*WTM + 16 i + r, channel (inside the BN tile) = chan_of_row(wn*WTN + 16 j + 4 kq + e)
    constexpr int CPL = NT == 1 ? 4 : 8;             // consecutive channels per lane per group
    constexpr int NG = NT == 1 ? 1 : NT / 2;         // channel groups per lane
    const float gate = p.gate ? *p.gate : 0.f;
    const bool out_f32 = (p.flags & GSSD_CONV_OUT_F32) != 0;
    float ssum[NG][CPL], ssq[NG][CPL];
#pragma unroll
    for (int u = 0; u < NG; ++u)
#pragma unroll
        for (int c = 0; c < CPL; ++c) ssum[u][c] = ssq[u][c] = 0.f;

#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int cb = NT == 1 ? 4 * kq : 32 * u + 8 * kq;               // first of this lane's consecutive channels
        const int ng0 = n0g + wn * WTN + cb;
        const int n0 = g * cout_g + ng0;
        float bias[CPL], alpha[CPL];
        bool nok[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            nok[c] = ng0 + c < cout_g;
            bias[c] = (p.bias && nok[c] && kz == 0) ? p.bias[n0 + c] : 0.f;
            alpha[c] = (p.alpha && nok[c]) ? p.alpha[n0 + c] : 1.f;
        }
        const bool all_ok = ng0 + CPL <= cout_g;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * WTM + i * 16 + r;
            const bool m_ok = m < M;
            float v[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const float a = NT == 1 ? acc[i][0][c] : acc[i][2 * u + (c >> 2)][c & 3];
                v[c] = a * alpha[c] + bias[c];
                if (m_ok && nok[c]) {
                    ssum[u][c] += v[c];
                    ssq[u][c] += v[c] * v[c];
                }
            }
            if (p.out_mode == GSSD_OUT_TRANSPOSED || (p.out_mode == GSSD_OUT_SPLIT_T && n0g >= p.split_n)) {
                // per image [channel][m] fp32 rows (the value projection of Self_Attn), zero padded up to the row stride
                const bool second = p.out_mode == GSSD_OUT_SPLIT_T;
                const int rs = second ? p.out_b_stride : p.out_stride;
                float* base = second ? p.out_b + (size_t)img * p.outb_batch_stride : p.out + (size_t)img * p.out_batch_stride;
                const int nsub = second ? p.split_n : 0;
                if (m < rs) {
                    if (second && (p.flags & GSSD_CONV_OUTB_BF16_PERM32)) {
                        // bf16 rows whose 32-token blocks are stored in the key order of the bf16-value attention core:
                        // token 32 t + 16 a + 4 b + c  ->  position 32 t + 8 b + 4 a + c
                        u16* hb = reinterpret_cast<u16*>(p.out_b) + (size_t)img * p.outb_batch_stride;
                        const int mp = (m & ~31) | (((m >> 2) & 3) << 3) | (((m >> 4) & 1) << 2) | (m & 3);
#pragma unroll
                        for (int c = 0; c < CPL; ++c)
                            if (nok[c]) hb[(size_t)(n0 + c - nsub) * rs + mp] = f2bf(m_ok ? v[c] : 0.f);
                    } else {
#pragma unroll
                        for (int c = 0; c < CPL; ++c)
                            if (nok[c]) base[(size_t)(n0 + c - nsub) * rs + m] = m_ok ? (p.relu ? fmaxf(v[c], 0.f) : v[c]) : 0.f;
                    }
                }
                continue;
            }
            if (!m_ok) continue;
            if (p.out_mode == GSSD_OUT_HEADS) {
                const int b = m / HoWo, pix = m - b * HoWo;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    if (!nok[c]) continue;
                    const int n = n0 + c;
                    float* dst = (n < p.split_n) ? p.out + ((size_t)b * p.out_batch_stride + p.out_off + (size_t)pix * p.split_n + n)
                                                 : p.out_b + ((size_t)b * p.outb_batch_stride + p.outb_off +
                                                              (size_t)pix * (p.Cout - p.split_n) + (n - p.split_n));
                    if (p.flags & GSSD_CONV_HEADS_SLICES)             // deterministic split-K: one output copy per slice (see conv_igemm.hip)
                        dst[(size_t)kz * ((n < p.split_n) ? (size_t)p.B * p.out_batch_stride : (size_t)p.B * p.outb_batch_stride)] = v[c];
                    else if (p.split_k > 1) unsafeAtomicAdd(dst, v[c]);
                    else *dst = v[c];
                }
                continue;
            }
            const size_t o = (size_t)img * p.out_batch_stride + (size_t)m * p.out_stride + p.out_ch_off + n0;
            if (p.gate || p.resid) {
                const u16* rp = reinterpret_cast<const u16*>(p.resid);
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    float t = v[c];
                    if (p.gate) {
                        t *= gate;
                        if (p.out2 && nok[c]) reinterpret_cast<u16*>(p.out2)[o + c] = f2bf(t);
                    }
                    if (rp && nok[c]) t += (p.flags & GSSD_CONV_RESID_F32) ? reinterpret_cast<const float*>(p.resid)[o + c] : bf2f(rp[o + c]);
                    v[c] = t;
                }
            }
            if (p.relu) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            if (out_f32) {
                if (p.split_k > 1) {                   // K slices accumulate into a zero-filled map (weight gradients as NT GEMMs)
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
                        if (nok[c]) unsafeAtomicAdd(p.out + o + c, v[c]);
                } else if (all_ok) {
#pragma unroll
                    for (int c = 0; c < CPL; c += 4) *reinterpret_cast<f32x4*>(p.out + o + c) = f32x4{v[c], v[c + 1], v[c + 2], v[c + 3]};
                } else {
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
                        if (nok[c]) p.out[o + c] = v[c];
                }
            } else {
                u16* ob = reinterpret_cast<u16*>(p.out);
                if (all_ok) {
                    if constexpr (CPL == 8) {
                        bf16x8 h;
#pragma unroll
                        for (int c = 0; c < 8; ++c) h[c] = (__bf16)v[c];
                        *reinterpret_cast<bf16x8*>(ob + o) = h;
                    } else {
                        bf16x4 h;
#pragma unroll
                        for (int c = 0; c < 4; ++c) h[c] = (__bf16)v[c];
                        *reinterpret_cast<bf16x4*>(ob + o) = h;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
                        if (nok[c]) ob[o + c] = f2bf(v[c]);
                }
            }
        }
    }

    if (p.stats) {
        // per-channel sum / sum^2 over this tile's pixels: the 16 lanes sharing kq hold 16 pixels of the same channels
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_h);  // [WM][BN][2]
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                float s = ssum[u][c], q = ssq[u][c];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s += __shfl_xor(s, o, 64);
                    q += __shfl_xor(q, o, 64);
                }
                if (r == 0) {
                    const int cl = wn * WTN + (NT == 1 ? 4 * kq : 32 * u + 8 * kq) + c;
                    red[(wm * BN + cl) * 2 + 0] = s;
                    red[(wm * BN + cl) * 2 + 1] = q;
                }
            }
        __syncthreads();
        if (tid < BN && n0g + tid < cout_g) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s += (double)red[(w * BN + tid) * 2 + 0];
                q += (double)red[(w * BN + tid) * 2 + 1];
            }
            const int n = g * cout_g + n0g + tid;
            double* st = gssd_stats_replica(p.stats, p.stats_rep, p.Cout);
            unsafeAtomicAdd(st + n, s);
            unsafeAtomicAdd(st + p.Cout + n, q);
        }
    }
}

template <int BM, int BN, int WM, int WN, int NSTG = 2>
int launch_cfg(const gssd_conv_desc& d, int M, int images, hipStream_t stream) {
    static unsigned attr_mask = 0;
    constexpr size_t smem_base = NSTG * (size_t)(BM + BN) * BK * sizeof(u16);
    const size_t smem = smem_base + (d.in_scale ? 2 * (size_t)d.cin_g * sizeof(float) : 0);
    auto kern = conv_bf16_kernel<BM, BN, WM, WN, NSTG>;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(smem_base + 8192)) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", smem_base + 8192);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask);
    }
    const int cout_g = d.Cout / d.groups;
    const int tiles = (cout_g + BN - 1) / BN;
    const int mtiles = (M + BM - 1) / BM;
    dim3 grid = d.m_per_image ? dim3(mtiles, d.groups * tiles, images) : dim3((mtiles + 7) / 8 * 8 * d.groups * tiles, 1, d.split_k);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, d, M, tiles);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// OIHW fp32 -> packed bf16 rows [Cout][Kpad], k = (kh*KW + kw)*cin_g_pad + c
__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int cin_g, int KH, int KW,
                                        int cin_g_pad, int Kpad) {
    const long long total = (long long)Cout * Kpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad), n = (int)(i / Kpad);
        const int tap = k / cin_g_pad, c = k - tap * cin_g_pad;
        float v = 0.f;
        if (tap < KH * KW && c < cin_g) v = w[((size_t)n * cin_g + c) * KH * KW + tap];
        wp[i] = f2bf(v);
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}

// bf16 -> fp32 (exact), 8 elements per thread
__global__ void cast_bf16_f32_kernel(const u16* __restrict__ x, float* __restrict__ y, long long n) {
    const long long n8 = n >> 3;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + 8 * i);
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = (float)v[e];
            b[e] = (float)v[e + 4];
        }
        *reinterpret_cast<f32x4*>(y + 8 * i) = a;
        *reinterpret_cast<f32x4*>(y + 8 * i + 4) = b;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) y[8 * n8 + threadIdx.x] = bf2f(x[8 * n8 + threadIdx.x]);
}

// hi = bf16(x), lo = bf16(x - hi): the two-term split whose three cross products reproduce an fp32 product to 2^-16 on the bf16 matrix cores
__global__ void cast_split_bf16_kernel(const float* __restrict__ x, u16* __restrict__ hi, u16* __restrict__ lo, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const u16 h = f2bf(v);
        hi[i] = h;
        lo[i] = f2bf(v - bf2f(h));
    }
}

// y[r][c] = bf16(x[r][c]) for c < cols, 0 for cols <= c < ld_y: rows widened to a multiple of 8 channels for the 16-byte lanes of the
// bf16 kernels (the merged head gradients: 36 -> 40 channels)
__global__ void cast_rows_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y, long long rows, int cols, int ld_x, int ld_y) {
    const long long total = rows * ld_y;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ld_y;
        const int c = (int)(i - r * ld_y);
        y[i] = c < cols ? f2bf(x[r * ld_x + c]) : (u16)0;
    }
}

}  // namespace

extern "C" int gssd_cast_split_f32_bf16(const float* x, void* hi, void* lo, int64_t n, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && hi && lo && n > 0);
    hipLaunchKernelGGL(cast_split_bf16_kernel, dim3((int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0, as_stream(stream), x,
                       reinterpret_cast<u16*>(hi), reinterpret_cast<u16*>(lo), (long long)n);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_cast_rows_f32_bf16(const float* x, void* y, int64_t rows, int cols, int ld_x, int ld_y, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && rows > 0 && cols > 0 && ld_x >= cols && ld_y >= cols);
    const long long total = rows * ld_y;
    hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3((int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), x, reinterpret_cast<u16*>(y), (long long)rows, cols, ld_x, ld_y);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_cast_bf16_f32(const void* x, float* y, int64_t n, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && n > 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0);
    const long long thr = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((int)(thr > 16384 ? 16384 : (thr < 1 ? 1 : thr))), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const u16*>(x), y, (long long)n);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pack_conv_weight_bf16(const float* w_oihw, void* w_packed, int Cout, int cin_g, int KH, int KW, int cin_g_pad,
                                          int Kpad, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_oihw && w_packed && Cout > 0 && cin_g > 0 && KH > 0 && KW > 0 && cin_g_pad >= cin_g && cin_g_pad % 8 == 0);
    GSSD_CHECK_ARG(Kpad >= KH * KW * cin_g_pad && Kpad % 8 == 0);
    const long long total = (long long)Cout * Kpad;
    hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_oihw, reinterpret_cast<u16*>(w_packed), Cout, cin_g, KH, KW, cin_g_pad, Kpad);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_cast_f32_bf16(const float* x, void* y, int64_t n, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0,
                       as_stream(stream), x, reinterpret_cast<u16*>(y), (long long)n);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_conv2d_nhwc_bf16(const gssd_conv_desc* dp, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dp != nullptr);
    const gssd_conv_desc& d = *dp;
    GSSD_CHECK_ARG(d.in && d.wgt && d.out);
    GSSD_CHECK_ARG(d.B > 0 && d.H > 0 && d.W > 0 && d.Ho > 0 && d.Wo > 0);
    GSSD_CHECK_ARG(d.groups > 0 && d.Cout > 0 && d.Cout % d.groups == 0);
    GSSD_CHECK_ARG(d.cin_g > 0 && d.cin_g % 8 == 0 && d.in_stride % 8 == 0 && d.in_ch_off % 8 == 0);
    GSSD_CHECK_ARG(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.dil > 0 && d.pad >= 0);
    GSSD_CHECK_ARG(d.K == d.KH * d.KW * d.cin_g && d.wgt_row_stride >= d.K && d.wgt_row_stride % 8 == 0);
    GSSD_CHECK_ARG(((uintptr_t)d.in % 16) == 0 && ((uintptr_t)d.wgt % 16) == 0 && ((uintptr_t)d.out % 16) == 0);
    GSSD_CHECK_ARG(d.out_mode >= 0 && d.out_mode <= 3);
    const bool out_f32 = (d.flags & GSSD_CONV_OUT_F32) != 0;
    if (d.out_mode == GSSD_OUT_TRANSPOSED)
        GSSD_CHECK_ARG(d.m_per_image && out_f32 && d.out_stride % 4 == 0 && d.out_batch_stride % 4 == 0 && !d.gate && !d.resid);
    if (d.out_mode == GSSD_OUT_HEADS) GSSD_CHECK_ARG(d.out_b && d.split_n > 0 && d.split_n < d.Cout && !d.m_per_image && out_f32);
    if (d.out_mode == GSSD_OUT_SPLIT_T) {
        GSSD_CHECK_ARG(d.m_per_image && d.groups == 1 && d.out_b && d.split_n > 0 && d.split_n < d.Cout && d.split_n % 64 == 0 && out_f32);
        GSSD_CHECK_ARG(d.out_b_stride % 4 == 0 && d.out_b_stride >= d.Ho * d.Wo && d.outb_batch_stride % 4 == 0);
        if (d.flags & GSSD_CONV_OUTB_BF16_PERM32) GSSD_CHECK_ARG(d.out_b_stride % 32 == 0 && d.outb_batch_stride % 8 == 0);
        GSSD_CHECK_ARG(!d.gate && !d.resid && !d.relu && !d.stats && d.split_k == 1);
    }
    if (d.out_mode == GSSD_OUT_NHWC) GSSD_CHECK_ARG((d.out_stride % 8 == 0 && d.out_ch_off % 8 == 0) || out_f32);
    if (d.m_per_image) GSSD_CHECK_ARG(d.in_batch_stride % 8 == 0 && d.wgt_batch_stride % 8 == 0);
    GSSD_CHECK_ARG(d.split_k >= 1 && d.split_k <= 64);
    GSSD_CHECK_ARG((d.in_scale == nullptr) == (d.in_shift == nullptr) && (d.in_scale == nullptr) == (d.in_pad == nullptr));
    // split-K accumulates with fp32 atomics into a zero-filled fp32 output: the heads and plain NHWC fp32 outputs
    if (d.split_k > 1)
        GSSD_CHECK_ARG(out_f32 && !d.m_per_image && !d.stats && !d.relu && !d.gate && !d.resid &&
                       (d.out_mode == GSSD_OUT_HEADS || d.out_mode == GSSD_OUT_NHWC));
    GSSD_CHECK_ARG((d.H + 2 * d.pad - d.dil * (d.KH - 1) - 1) / d.stride + 1 == d.Ho);
    GSSD_CHECK_ARG((d.W + 2 * d.pad - d.dil * (d.KW - 1) - 1) / d.stride + 1 == d.Wo);
    const int images = d.m_per_image ? d.B : 1;
    const long long Mll = (long long)(d.m_per_image ? 1 : d.B) * d.Ho * d.Wo;
    GSSD_CHECK_ARG(Mll < (1ll << 31));
    GSSD_CHECK_ARG((long long)(d.m_per_image ? 1 : d.B) * d.H * d.W * d.in_stride < (1ll << 31));
    GSSD_CHECK_ARG((long long)(d.Cout / d.groups + 256) * d.wgt_row_stride < (1ll << 31));
    const int M = (int)Mll;
    const int cout_g = d.Cout / d.groups;
    hipStream_t s = as_stream(stream);
    {
        const int rc = gssd_try_conv_thin_bf16(d, s);      // conv1_1 .. conv2_2: patch-staged HBM-stream kernel
        if (rc != 1) return rc;
    }
    if (d.flags & GSSD_CONV_POOL2) {
        gssd_set_error("GSSD_CONV_POOL2: no bf16 kernel with a pooled epilogue takes this descriptor (thin trunk shapes only)");
        return GSSD_EINVAL;
    }
    {
        const int rc = gssd_try_conv_flat_bf16(d, s);      // conv3_1 .. conv6: flat-window kernel (csrc/conv_flat_bf16.hip)
        if (rc != 1) return rc;
    }
    if (d.in_scale) GSSD_CHECK_ARG(d.cin_g <= 1024 && !d.m_per_image);
    // small maps (<= 10 x 10 at batch 32; per-image GEMMs of <= 100 tokens): 32- / 64-row tiles, three-stage K loop (csrc/conv_igemm.hip)
    static const bool no_small = getenv("GSSD_NO_SMALL_TILES") != nullptr;
    if (!no_small && cout_g > 32 && d.split_k == 1 && !(d.out_mode == GSSD_OUT_SPLIT_T && d.split_n % 64 != 0)) {
        // (per-image GEMMs count all their images: the 19 x 19 projections -- 361 tokens x 32 images -- keep the 128-row tiles)
        const long long mtot = (long long)M * images;
        if (mtot <= 512 || (d.m_per_image && mtot <= 4096 && M <= 128)) return launch_cfg<32, 64, 1, 4, 3>(d, M, images, s);
        if (mtot <= 4096) return launch_cfg<64, 64, 2, 2, 3>(d, M, images, s);
    }
    // Measured and rejected (round 5), kept behind GSSD_BF16_BIG_TILES=1: 256 x 128 tiles on eight waves for the large GEMM-shaped launches (the
    // 38 x 38 / 19 x 19 Self_Attn projections and output convs, the fuse convs).  The idea: 128 x 64 tiles re-read the activation tile once per
    // output tile (566 MB of L2 -> LDS traffic per fuse_11 launch for 94 MB of operands).  Result: 7 launches 74.5 us at 197 TFLOP/s against
    // 58.9 us at 287 -- with K = 256 .. 1024 a tile runs 4 .. 16 K steps, so the launch is bound by the tiles' prologue / epilogue, not by the
    // traffic of the main loop, and the larger tile only costs occupancy.
    static const int big = [] { const char* e = getenv("GSSD_BF16_BIG_TILES"); return e ? atoi(e) : 0; }();
    if (big && !d.m_per_image && d.split_k == 1 && M >= 8192 && cout_g % 128 == 0 && d.K % BK == 0 && d.K >= 256 &&
        (d.out_mode == GSSD_OUT_NHWC || (d.out_mode == GSSD_OUT_SPLIT_T && d.split_n % 128 == 0))) {
        if (big == 2) return launch_cfg<256, 128, 4, 2, 3>(d, M, images, s);     // round 6 sweep: the same tile behind a three-stage ring (144 KB)
        if (big == 3) return launch_cfg<192, 128, 4, 2, 3>(d, M, images, s);     // 192 rows: 46 208 pixels = 241 row tiles -> 723 / 964 tiles = 2.8 / 3.8 rounds
        if (big == 4) return launch_cfg<128, 128, 2, 2, 4>(d, M, images, s);     // four stages of 32 KB, four 64 x 64 waves
        if (big == 5) return launch_cfg<128, 128, 2, 2, 3>(d, M, images, s);
        if (big == 6) return launch_cfg<128, 128, 4, 2, 4>(d, M, images, s);     // eight 32 x 64 waves
        return launch_cfg<256, 128, 4, 2>(d, M, images, s);      // (256 x 256 on eight waves: 256 registers and 672 bytes of scratch per lane)
    }
    if (cout_g > 64) {
        const long long mt = (M + 127) / 128, z = d.m_per_image ? images : d.split_k;
        const long long b128 = mt * d.groups * ((cout_g + 127) / 128) * z, b64 = mt * d.groups * ((cout_g + 63) / 64) * z;
        const double e128 = (double)b128 / (double)(((b128 + 511) / 512) * 512);
        const double e64 = 0.94 * (double)b64 / (double)(((b64 + 767) / 768) * 768);
        // (the three-stage K loop of the small tiles on these 128-row tiles: bf16 fwd + loss 3.87 -> 4.11 ms, measured round 4)
        if (e64 > e128 || d.K <= 256 || (d.out_mode == GSSD_OUT_SPLIT_T && d.split_n % 128 != 0)) return launch_cfg<128, 64, 2, 2>(d, M, images, s);
        return launch_cfg<128, 128, 2, 2>(d, M, images, s);
    }
    if (cout_g > 32) return launch_cfg<128, 64, 2, 2>(d, M, images, s);
    if (cout_g > 16) return launch_cfg<128, 32, 4, 1>(d, M, images, s);
    return launch_cfg<128, 16, 4, 1>(d, M, images, s);
}
