// Implicit-GEMM convolution for gfx950 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   out[m][n] = epilogue( sum_k A[m][k] * Wp[n][k] )
//   m : output pixel (linear over b, oy, ox, or over one image when m_per_image)
//   n : output channel inside conv group g
//   k : (tap, input channel of the group) flattened, k = tap*cin_g + c
//
// Both operands are K-contiguous in memory (NHWC activations; K-major packed weights).  Tiles are staged
// straight from global memory into LDS with 16-byte LDS-DMA loads (global_load_lds_dwordx4: no VGPR
// round trip, no ds_write): a wave instruction lands 64 lanes x 16 B = 8 tile rows x 128 B (BK = 32
// floats) contiguously.  Because the DMA image is lane-linear, the bank swizzle is applied on the SOURCE
// side: LDS slot q' of row r holds logical k-quad q' ^ (r & 7), and fragment reads XOR the same way
// (one ds_read_b128 per 16x16 fragment per 16 k; lane (r = lane&15, kq = lane>>4) holds
// k = 16*ks + 4*kq + s for MFMA step s -- A and B use the same k permutation, so the sum is unchanged).
// Out-of-image taps and tile tails read a 16-byte zero page instead of branching.
//
// A 256-thread workgroup (4 waves) owns a BM x BN output tile; the DMA of K-chunk i+1 is issued before the
// MFMAs of chunk i into the other LDS stage (one barrier per chunk).  fp32 MFMA is exact fp32 FMA at the
// vector rate, so results match an fp32 reference to accumulation-order rounding.
//
// Replaces the implicit cuDNN/ATen kernels behind nn.Conv2d / torch.bmm on the reference path;
// see include/gssd_hip.h for the call-site map.
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;          // floats per tile row = 128 B = 8 DMA lanes

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// wait until at most N of this wave's LDS-DMA pieces are outstanding and this wave's fragment reads have returned, then the workgroup
// barrier -- NO fence (a __syncthreads() waits vmcnt(0) and would drain the ring of the NSTG >= 3 form)
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// NSTG = LDS stages of the K loop.  2: the chunk's DMA is issued one chunk ahead and every chunk ends in a __syncthreads() -- right for
// the large layers, whose MFMAs per chunk cover the load latency.  3 (round 4, the small-map launches: <= 10 x 10 maps, M <= 3200): two
// chunks in flight under counted waits, and 32- / 64-row tiles -- with a 128-row tile a launch on the 1 x 1 map (M = 32) issued 4 x the
// MFMAs it needed and paid one L2 round trip per 32-k chunk (17 - 25 us per launch for microseconds of work; the tail of a step is ~50
// such launches in a dependent chain, profiles/r04_a_critical_path_f32.txt).
template <int BM, int BN, int WM, int WN, int NSTG>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_kernel(const gssd_conv_desc p, const int M,
                                                         const int tiles_per_group) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int NW = WM * WN, NTHR = NW * 64;     // waves / threads per workgroup
    constexpr int AR = BM / (8 * NW);               // A DMA instructions per wave per chunk (8 rows each)
    constexpr int BPIECES = BN / 8;                 // B 1-KiB pieces per chunk
    constexpr int BR = (BPIECES + NW - 1) / NW;     // B DMA instructions per wave per chunk
    constexpr int STAGE = (BM + BN) * BK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, kq = lane >> 4;
    // Flat XCD-aware grid: workgroup ids go round-robin over the 8 XCDs (one L2 each).  Id L -> XCD L & 7, slot L >> 3; the
    // slots of an XCD run through all (group, N-tile) pairs of one M tile before the next M tile, so the A rows a wide
    // layer reads once per N tile come from that L2 after the first fetch (fuse_11: 4 N tiles -> HBM fetch 384 -> ~100 MB).
    // (per-image batched GEMMs keep the plain 2-D grid: their M extent is a handful of tiles and padding it to 8 only adds
    // empty workgroups)
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
    const int kz = p.m_per_image ? 0 : blockIdx.z;   // split-K slice (split_k > 1 only without m_per_image)
    const int HoWo = p.Ho * p.Wo;
    const int K = p.K;
    const int taps = p.KH * p.KW;

    const float* __restrict__ in = p.in + (size_t)img * p.in_batch_stride + p.in_ch_off + g * p.cin_g;
    const float* __restrict__ wgt =
        p.wgt + (size_t)img * p.wgt_batch_stride + (size_t)(g * cout_g) * p.wgt_row_stride;
    const float* zero = g_zero_page;
    // fused producer BatchNorm + ReLU: fragments are read as max(x*scale[c] + shift[c], 0); out-of-image taps DMA the
    // per-channel pad value (mapped to 0 by the transform) instead of the zero page
    const bool xf = p.in_scale != nullptr;
    float* xtab = smem + NSTG * STAGE;               // [2][cin_g]: scale | shift of this group's input channels
    if (xf) {
        for (int c = tid; c < p.cin_g; c += NTHR) {
            xtab[c] = p.in_scale[p.in_ch_off + g * p.cin_g + c];
            xtab[p.cin_g + c] = p.in_shift[p.in_ch_off + g * p.cin_g + c];
        }
    }
    const float* padp = xf ? p.in_pad + p.in_ch_off + g * p.cin_g : nullptr;

    // ---- DMA lane roles: lane L lands at (row_in = L>>3, slot = L&7) of its 8-row piece and must fetch the
    //      logical k-quad slot ^ row_in (source-side swizzle) ----------------------------------------------------
    const int row_in = lane >> 3;
    const int lq = (lane & 7) ^ row_in;              // logical quad (k = 4*lq within the chunk)
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
        a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 20);      // invalid rows fail every bounds test
        a_ix0[j] = ox * p.stride - p.pad;
        a_off[j] = ((b * p.H + oy * p.stride - p.pad) * p.W + a_ix0[j]) * p.in_stride;
    }
    int b_off[BR];
    bool b_ok[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int row = (j * NW + wave) * 8 + row_in;
        b_ok[j] = (j * NW + wave) < BPIECES && (n0g + row) < cout_g;
        b_off[j] = (n0g + row) * p.wgt_row_stride + 4 * lq;
    }
    const int nchunks_all = (K + BK - 1) / BK;
    const int cps = (nchunks_all + p.split_k - 1) / p.split_k;      // chunks per split-K slice
    const int ch_begin = kz * cps;
    const int ch_end = min(nchunks_all, ch_begin + cps);
    int a_tap = (ch_begin * BK + 4 * lq) / p.cin_g;
    int a_c = (ch_begin * BK + 4 * lq) - a_tap * p.cin_g;

    auto issue = [&](int chunk, int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + BM * BK;
        const int ty = a_tap / p.KW, tx = a_tap - ty * p.KW;
        const int dy = ty * p.dil, dx = tx * p.dil;
        const bool tap_ok = a_tap < taps;
        const int toff = (dy * p.W + dx) * p.in_stride + a_c;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const bool ok = tap_ok && (unsigned)(a_iy0[j] + dy) < (unsigned)p.H && (unsigned)(a_ix0[j] + dx) < (unsigned)p.W;
            const float* src = ok ? in + (a_off[j] + toff) : (xf ? padp + a_c : zero);
            dma16(src, As + (j * NW + wave) * 8 * BK);
        }
        const int k0 = chunk * BK;
        const bool kok = k0 + 4 * lq < K;
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            if ((j * NW + wave) < BPIECES) {          // wave-uniform
                const float* src = (b_ok[j] && kok) ? wgt + (b_off[j] + k0) : zero;
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

    // fragment read offsets (floats) inside a stage: row (base + r), physical slot (4*ks + kq) ^ (r & 7)
    const int fo0 = r * BK + ((kq ^ (r & 7)) << 2);
    const int fo1 = r * BK + (((4 + kq) ^ (r & 7)) << 2);

    int f_c0 = (ch_begin * BK + 4 * kq) % p.cin_g, f_c1 = (ch_begin * BK + 16 + 4 * kq) % p.cin_g;   // fragment channels
    static_assert(NSTG == 2 || BPIECES % NW == 0, "counted waits need the same number of DMA pieces per chunk in every wave");
    constexpr int PER = AR + BR;                     // DMA pieces per chunk and wave
    if constexpr (NSTG == 2) {
        if (ch_begin < ch_end) issue(ch_begin, 0);
        __syncthreads();
    } else {
#pragma unroll
        for (int st = 0; st < NSTG - 1; ++st)
            if (ch_begin + st < ch_end) issue(ch_begin + st, st);
        if (xf) __syncthreads();                     // (the scale / shift table; drains the prologue pieces: harmless)
    }
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        int buf;
        if constexpr (NSTG == 2) {
            buf = (ch - ch_begin) & 1;
            if (ch + 1 < ch_end) issue(ch + 1, buf ^ 1);
        } else {
            buf = (ch - ch_begin) % NSTG;
            // chunk ch has landed (at most the younger chunks' pieces are outstanding); every wave is done with chunk ch - 1's stage
            if (NSTG >= 4 && ch + 2 < ch_end) wait_vm_barrier<2 * PER>();
            else if (ch + 1 < ch_end) wait_vm_barrier<PER>();
            else wait_vm_barrier<0>();
            if (ch + NSTG - 1 < ch_end) issue(ch + NSTG - 1, (ch - ch_begin + NSTG - 1) % NSTG);
        }
        const float* As = smem + buf * STAGE + wm * WTM * BK;
        const float* Bs = smem + buf * STAGE + BM * BK + wn * WTN * BK;
        // both 16-k fragment sets are requested up front: the second set's LDS latency hides behind the first set's MFMAs
        f32x4 af[2][MT], bf[2][NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks ? fo1 : fo0;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[ks][i] = *reinterpret_cast<const f32x4*>(As + i * 16 * BK + fo);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[ks][j] = *reinterpret_cast<const f32x4*>(Bs + j * 16 * BK + fo);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (xf) {
                const int fc = ks ? f_c1 : f_c0;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(xtab + fc);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(xtab + p.cin_g + fc);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x4 v = af[ks][i] * sc + sh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    af[ks][i] = v;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][i][s], bf[ks][j][s], acc[i][j], 0, 0, 0);
        }
        if (xf) {
            f_c0 += BK;
            while (f_c0 >= p.cin_g) f_c0 -= p.cin_g;
            f_c1 += BK;
            while (f_c1 >= p.cin_g) f_c1 -= p.cin_g;
        }
        if constexpr (NSTG == 2) __syncthreads();
    }
    if constexpr (NSTG != 2) __syncthreads();       // the epilogue's reduction reuses the stages

    // ---- epilogue ---------------------------------------------------------------------------
    const float gate = p.gate ? *p.gate : 0.f;
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) ssum[j] = ssq[j] = 0.f;

#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ng = n0g + wn * WTN + j * 16 + r;
        const bool n_ok = ng < cout_g;
        const int n = g * cout_g + ng;
        const float bias = (p.bias && n_ok && kz == 0) ? p.bias[n] : 0.f;
        const float alpha = (p.alpha && n_ok) ? p.alpha[n] : 1.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + wm * WTM + i * 16 + kq * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[i][j][e] * alpha + bias;
                if (mb + e < M && n_ok) {
                    ssum[j] += v[e];
                    ssq[j] += v[e] * v[e];
                }
            }
            if (!n_ok) continue;
            if (p.out_mode == GSSD_OUT_SPLIT_T && n0g >= p.split_n) {
                // second column range of a merged projection: per image [n - split_n][m], zero padded up to the row stride.  With all
                // images in one M range (m_per_image == 0: Ho*Wo % 4 == 0, so the four rows of a lane stay inside one image) the image
                // index comes from the row
                const int bi = p.m_per_image ? img : mb / HoWo;
                const int ml = p.m_per_image ? mb : mb - bi * HoWo;
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
                    if (p.out_mode == GSSD_OUT_HEADS) {
                        const int b = m / HoWo, pix = m - b * HoWo;
                        float* dst = (n < p.split_n)
                                         ? p.out + ((size_t)b * p.out_batch_stride + p.out_off + (size_t)pix * p.split_n + n)
                                         : p.out_b + ((size_t)b * p.outb_batch_stride + p.outb_off +
                                                      (size_t)pix * (p.Cout - p.split_n) + (n - p.split_n));
                        if (p.flags & GSSD_CONV_HEADS_SLICES) {        // deterministic split-K: slice kz has its own copy of the
                            dst[(size_t)kz * ((n < p.split_n) ? (size_t)p.B * p.out_batch_stride      // outputs, summed in order by
                                                                   : (size_t)p.B * p.outb_batch_stride)] = t;   // gssd_heads_reduce_f32
                        } else if (p.split_k > 1) {
                            unsafeAtomicAdd(dst, t);                    // caller zero-fills the buffers
                        } else {
                            *dst = t;
                        }
                    } else {
                        const size_t o = (size_t)img * p.out_batch_stride + (size_t)m * p.out_stride + p.out_ch_off + n;
                        if (p.gate) {
                            t *= gate;
                            if (p.out2) p.out2[o] = t;
                            if (p.resid) t += p.resid[o];
                        } else if (p.resid) {
                            t += p.resid[o];
                        }
                        if (p.relu) t = fmaxf(t, 0.f);
                        if (p.split_k > 1) unsafeAtomicAdd(p.out + o, t);
                        else p.out[o] = t;
                    }
                }
            }
        }
    }

    if (p.stats) {
        // per-channel sum / sum^2 over this tile's rows: lanes sharing (lane&15) -> LDS over wm -> fp64 atomics
        __syncthreads();
        float* red = smem;  // [WM][BN][2]
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
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    constexpr size_t smem_base = NSTG * (size_t)(BM + BN) * BK * sizeof(float);
    const size_t smem = smem_base + (d.in_scale ? 2 * (size_t)d.cin_g * sizeof(float) : 0);
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, NSTG>;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(smem_base + 4096)) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", smem_base + 4096);
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const int cout_g = d.Cout / d.groups;
    const int tiles = (cout_g + BN - 1) / BN;
    const int mtiles = (M + BM - 1) / BM;
    dim3 grid = d.m_per_image ? dim3(mtiles, d.groups * tiles, images) : dim3((mtiles + 7) / 8 * 8 * d.groups * tiles, 1, d.split_k);
    static_assert(BM % (8 * WM * WN) == 0, "A rows must split evenly over the waves");
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, d, M, tiles);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

extern "C" int gssd_conv2d_nhwc_f32(const gssd_conv_desc* dp, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dp != nullptr);
    const gssd_conv_desc& d = *dp;
    GSSD_CHECK_ARG(d.in && d.wgt && d.out);
    GSSD_CHECK_ARG(d.B > 0 && d.H > 0 && d.W > 0 && d.Ho > 0 && d.Wo > 0);
    GSSD_CHECK_ARG(d.groups > 0 && d.Cout > 0 && d.Cout % d.groups == 0);
    GSSD_CHECK_ARG(d.cin_g > 0 && d.cin_g % 4 == 0 && d.in_stride % 4 == 0 && d.in_ch_off % 4 == 0);
    GSSD_CHECK_ARG(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.dil > 0 && d.pad >= 0);
    GSSD_CHECK_ARG(d.K == d.KH * d.KW * d.cin_g && d.wgt_row_stride >= d.K && d.wgt_row_stride % 4 == 0);
    GSSD_CHECK_ARG(((uintptr_t)d.in % 16) == 0 && ((uintptr_t)d.wgt % 16) == 0);
    GSSD_CHECK_ARG(d.out_mode >= 0 && d.out_mode <= 3);
    if (d.out_mode == GSSD_OUT_SPLIT_T) {
        GSSD_CHECK_ARG((d.m_per_image || (d.Ho * d.Wo) % 4 == 0) && d.groups == 1 && d.out_b && d.split_n > 0 && d.split_n < d.Cout &&
                       d.split_n % 64 == 0);
        GSSD_CHECK_ARG(d.out_b_stride % 4 == 0 && d.out_b_stride >= d.Ho * d.Wo && d.outb_batch_stride % 4 == 0 && ((uintptr_t)d.out_b % 16) == 0);
        GSSD_CHECK_ARG(!d.gate && !d.resid && !d.relu && !d.stats && d.split_k == 1);
    }
    if (d.out_mode == GSSD_OUT_TRANSPOSED) {
        GSSD_CHECK_ARG(d.m_per_image && d.out_stride % 4 == 0 && ((uintptr_t)d.out % 16) == 0);
        GSSD_CHECK_ARG(d.out_batch_stride % 4 == 0 && !d.gate && !d.resid);
    }
    if (d.out_mode == GSSD_OUT_HEADS) GSSD_CHECK_ARG(d.out_b && d.split_n > 0 && d.split_n < d.Cout && !d.m_per_image);
    if (d.m_per_image) GSSD_CHECK_ARG(d.in_batch_stride % 4 == 0 && d.wgt_batch_stride % 4 == 0);
    GSSD_CHECK_ARG(d.split_k >= 1 && d.split_k <= 64);
    GSSD_CHECK_ARG((d.in_scale == nullptr) == (d.in_shift == nullptr) && (d.in_scale == nullptr) == (d.in_pad == nullptr));
    // split-K accumulates with fp32 atomics into a zero-filled output: linear epilogues only
    if (d.split_k > 1) GSSD_CHECK_ARG(!d.m_per_image && !d.stats && !d.relu && !d.gate && !d.resid && d.out_mode != GSSD_OUT_TRANSPOSED);
    // the conv arithmetic must reproduce Ho/Wo
    GSSD_CHECK_ARG((d.H + 2 * d.pad - d.dil * (d.KH - 1) - 1) / d.stride + 1 == d.Ho);
    GSSD_CHECK_ARG((d.W + 2 * d.pad - d.dil * (d.KW - 1) - 1) / d.stride + 1 == d.Wo);

    const int images = d.m_per_image ? d.B : 1;
    const long long Mll = (long long)(d.m_per_image ? 1 : d.B) * d.Ho * d.Wo;
    GSSD_CHECK_ARG(Mll < (1ll << 31));
    // 32-bit element offsets inside the kernel (one image set / one weight matrix per launch)
    GSSD_CHECK_ARG((long long)(d.m_per_image ? 1 : d.B) * d.H * d.W * d.in_stride < (1ll << 31));
    GSSD_CHECK_ARG((long long)(d.Cout / d.groups + 256) * d.wgt_row_stride < (1ll << 31));
    const int M = (int)Mll;
    const int cout_g = d.Cout / d.groups;
    hipStream_t s = as_stream(stream);
    if (d.wgt_patch) {
        const int rc = gssd_try_conv_patch_x6(d, s);  // many input channels, <= 128 outputs, GSSD_CONV_F16_OK: patch-staged direct conv on fp16 planes (round 6)
        if (rc != 1) return rc;
    }
    if (d.wgt_x6) {
        const int rc = gssd_try_conv_x6(d, s);        // fp32-equivalent products on the bf16 matrix cores (caller packed wgt_x6)
        if (rc != 1) return rc;
    }
    {
        const int rc = gssd_try_conv_thin_x6(d, s);   // conv1_2 / conv2_1 / conv2_2: patch-staged direct conv, three-plane bf16 operands (round 6)
        if (rc != 1) return rc;
    }
    {
        const int rc = gssd_try_conv_thin_wino(d, s); // conv1_2 (and its dgrad): patch-staged Winograd
        if (rc != 1) return rc;
    }
    {
        const int rc = gssd_try_conv_thin(d, s);      // conv1_1 / conv2_1 (conv1_2 without Winograd weights): patch-staged kernel
        if (rc != 1) return rc;
    }
    {
        const int rc = gssd_try_conv_wino(d, s);      // compute-bound 3x3 trunk layers: Winograd F(2x2,3x3)
        if (rc != 1) return rc;
    }
    if (d.flags & GSSD_CONV_POOL2) {
        gssd_set_error("GSSD_CONV_POOL2: no fp32 kernel with a pooled epilogue takes this descriptor (Winograd trunk shapes only)");
        return GSSD_EINVAL;
    }
    if (d.in_scale) GSSD_CHECK_ARG(d.cin_g <= 512 && !d.m_per_image);
    {
        static const bool no_slot = getenv("GSSD_NO_GEMM_SLOT") != nullptr;      // ablation switch (scripts/layer_times.py)
        const int rc = no_slot ? 1 : gssd_try_gemm_slot(d, s);                   // large plain 1x1 convs / GEMMs: 128 x 256 slot stream
        if (rc != 1) return rc;
    }
    // small maps (<= 10 x 10 at batch 32; per-image GEMMs of <= 100 tokens): 32- / 64-row tiles, three-stage K loop
    static const bool no_small = getenv("GSSD_NO_SMALL_TILES") != nullptr;       // ablation switch
    if (!no_small && cout_g > 32 && d.split_k == 1 && !(d.out_mode == GSSD_OUT_SPLIT_T && d.split_n % 64 != 0)) {
        // (per-image GEMMs count all their images: the 19 x 19 projections -- 361 tokens x 32 images -- keep the 128-row tiles)
        const long long mtot = (long long)M * images;
        if (mtot <= 512 || (d.m_per_image && mtot <= 4096 && M <= 128)) return launch_cfg<32, 64, 1, 4, 3>(d, M, images, s);
        if (mtot <= 4096) return launch_cfg<64, 64, 2, 2, 3>(d, M, images, s);
    }
    if (cout_g > 64) {
        // 128x128 tiles run 2 workgroups per CU (LDS), 128x64 tiles 3: pick the one whose last round of workgroups is
        // fuller (wave quantisation decides small 19x19 / 38x38 layers); the wide tile wins ties (less B re-read).
        const long long mt = (M + 127) / 128, z = d.m_per_image ? images : d.split_k;
        const long long b128 = mt * d.groups * ((cout_g + 127) / 128) * z, b64 = mt * d.groups * ((cout_g + 63) / 64) * z;
        const double e128 = (double)b128 / (double)(((b128 + 511) / 512) * 512);
        const double e64 = 0.94 * (double)b64 / (double)(((b64 + 767) / 768) * 768);
        // short reductions (K <= 256: the attention output conv) are prologue / epilogue bound: three resident 128x64
        // workgroups per CU overlap those phases better than two 128x128 ones
        if (e64 > e128 || d.K <= 256 || (d.out_mode == GSSD_OUT_SPLIT_T && d.split_n % 128 != 0)) return launch_cfg<128, 64, 2, 2>(d, M, images, s);
        return launch_cfg<128, 128, 2, 2>(d, M, images, s);
    }
    if (cout_g > 32) return launch_cfg<128, 64, 2, 2>(d, M, images, s);
    if (cout_g > 16) return launch_cfg<128, 32, 4, 1>(d, M, images, s);
    return launch_cfg<128, 16, 4, 1>(d, M, images, s);
}
