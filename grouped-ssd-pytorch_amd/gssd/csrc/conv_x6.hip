// fp32 implicit-GEMM convolution on the BF16 matrix cores with fp32-equivalent products (round 4): the plain-conv form of dcn_x6.hip.
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 matrix rate on gfx950.  An fp32 number is the exact sum of three bf16 numbers
// (x = h + m + l); a product x y is then h h' + (h m' + m h') + (h l' + l h' + m m') + terms below 2^-24 |x y| -- six
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation reproduce the fp32 product to the last bit or two at 3/8 of the fp32 instruction's
// matrix-pipe time.  No Winograd transform: the products are those of the direct convolution (the error of F(2x2,3x3) goes away too).
//
//   out[m][n] = sum_k A[m][k] Wp[n][k],  m: output pixel (linear over b, oy, ox), n: output channel of group g,
//   k: (32-channel block, tap) -- blocks outermost, so the fused producer BatchNorm's scale / shift change every `taps` chunks.
//
// Activations: a thread loads 8 fp32 channels of one tap of one output pixel (32 bytes; out-of-image taps are zero), applies the deferred
// BatchNorm + ReLU of the producer (gssd_conv_desc::in_scale / in_shift), splits into the three planes and writes them to LDS.  Weights are
// split once, when they are packed (gssd_conv_x6_pack_weight), and staged by LDS-DMA.  Takes the NHWC epilogues (per-channel scale, bias,
// gate / second output / residual, ReLU, batch sums) and the split-transposed store of the merged Self_Attn projection; transposed and
// head outputs, split-K and non-contiguous per-image batches stay with the fp32-MFMA kernels.
//
// Replaces the cuDNN kernels behind nn.Conv2d of the reference's trunk (layers/..., models/ssd_multiphase_custom_group.py: vgg()).
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

namespace {

#ifndef X6_SCHED
#define X6_SCHED 0           // N > 0: N (+1, +2 for the narrower tiles) vector instructions pinned behind every MFMA of the K loop by
                             // sched_group_barrier; 0: the compiler's own order -- 5-6 % faster with two workgroups per CU (same-box sweep 0 / 1 / 2 / 3:
                             // conv4_2 367 / 379 / 389 / 387 us, conv6 173 / 186 / 185 / 184), unlike dcn_x6 (one wave per SIMD) where the pinned order is the gain
#endif
#ifndef X6_KO
#define X6_KO 0             // knock-outs (scripts/conv_x6_knockout.sh): 1 no transform / split, 2 no MFMAs, 4 no weight DMA, 8 no activation loads,
#endif                      // 16 no fragment reads, 32 no LDS writes of the planes
#ifndef X6_LOCAL_SUM
#define X6_LOCAL_SUM 1      // the six products of a chunk are summed from zero and added to the running sum by the vector ALU (see the K loop)
#endif
constexpr int BM = 128, BKC = 32;
constexpr int NP = 3;
constexpr int A_STAGE = BM * BKC;                     // u16 elements per plane

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __host__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }       // 64-byte rows: conflict-free ds_read_b128

__device__ __host__ __forceinline__ int chan_of_row(int row) {        // LDS row of the weight tile -> output channel inside the BN tile
    const int j = row >> 4, rho = row & 15;
    return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
}

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

template <int BN>
struct Cfg {
    static constexpr int NBS = BN >= 128 ? 1 : 2;     // weight stages (one buffer: fragments -> barrier -> next DMA behind the MFMAs)
    static constexpr int B_STAGE = BN * BKC;
    static constexpr int LDS_BYTES = (2 * NP * A_STAGE + NBS * NP * B_STAGE) * 2;      // + the scale / shift table [2][cin_g] floats
    static constexpr int OCC = BN <= 128 ? 2 : 1;
};

// wp: [3 planes][groups][n_tiles][chunks][BN rows in staging order][32] bf16 (slot-swizzled), chunk = c32 * taps + tap
template <int BN, bool XF>
__global__ __launch_bounds__(256, Cfg<BN>::OCC) void conv_x6_kernel(const gssd_conv_desc p, const int M, const int ntn, const int mtiles,
                                                                   const long long plane_elems) {
    constexpr int NBS = Cfg<BN>::NBS, B_STAGE = Cfg<BN>::B_STAGE;
    constexpr int WTM = 64, WTN = BN / 2, MT = WTM / 16, NT = WTN / 16;
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    u16* const As = smem_h;                                   // [2][3][BM][32]
    u16* const Bs = smem_h + 2 * NP * A_STAGE;                // [NBS][3][BN][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
    // flat XCD-aware grid: the slots of an XCD run through all (group, N tile) pairs of one M tile before the next M tile
    const int ny = p.groups * ntn;
    const int slot = blockIdx.x >> 3;
    const int mt = (slot / ny) * 8 + (blockIdx.x & 7);
    const int by = slot % ny;
    if (mt >= mtiles) return;
    const int g = by / ntn, nt = by - g * ntn;
    const int cout_g = p.Cout / p.groups;
    const int n0g = nt * BN;
    const int m0 = mt * BM;
    const int HoWo = p.Ho * p.Wo;
    const int taps = p.KH * p.KW;
    const int cpc = p.cin_g / BKC;
    const int nchunks = cpc * taps;
    const float* __restrict__ in = p.in + p.in_ch_off + g * p.cin_g;
    const u16* wslab = reinterpret_cast<const u16*>(p.wgt_x6) + (size_t)by * nchunks * B_STAGE;      // plane 0; plane q at + q * plane_elems
    constexpr bool xf = XF;
    const float* xsc = xf ? p.in_scale + p.in_ch_off + g * p.cin_g : nullptr;
    const float* xsh = xf ? p.in_shift + p.in_ch_off + g * p.cin_g : nullptr;

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    // gather roles: thread -> (pixel row gp + 64 j, 8-channel slot gq)
    const int gq = tid & 3, gp = tid >> 2;
    const int a_wr0 = gp * BKC + ((gq ^ swz(gp)) << 3);
    const int fo = r * BKC + ((kq ^ swz(r)) << 3);
    int g_iy0[2], g_ix0[2], g_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + gp + 64 * j;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int b = mm / HoWo, pix = mm - b * HoWo;
        const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
        g_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 20);
        g_ix0[j] = ox * p.stride - p.pad;
        g_off[j] = ((b * p.H + oy * p.stride - p.pad) * p.W + g_ix0[j]) * p.in_stride + gq * 8;
    }


    // ---- the K loop, one instruction stream per chunk ------------------------------------------------------------------------------
    // chunk ch: fragments of chunk ch -> registers; global loads of chunk ch + 2's activations (registers) and the LDS-DMA of chunk ch + 1's
    // weights are issued; the MFMAs of chunk ch run INTERLEAVED with the vector work on chunk ch + 1's activations (loaded during chunk
    // ch - 1: deferred BatchNorm + ReLU, the three-plane split, LDS writes into the other stage).  Everything is unconditional (past the end
    // the last chunk is loaded again and the writes go to a stage nobody reads), so the chunk is one basic block the scheduler can interleave.
    int ld_ty = 0, ld_tx = 0, ld_c = 0, ld_tap = 0;          // chunk whose activations are loaded next
    int fin_c = 0, fin_tap = 0;                             // chunk whose activations are split next (its 32-channel block)
    f32x4 gvA[2][2], gvB[2][2];
    bool gokA[2], gokB[2];
    float* const xtab = reinterpret_cast<float*>(smem_h + 2 * NP * A_STAGE + NBS * NP * B_STAGE);      // [2][cin_g] scale | shift
    if (xf) {
        for (int c = tid; c < p.cin_g; c += 256) {
            xtab[c] = xsc[c];
            xtab[p.cin_g + c] = xsh[c];
        }
    }

    // ASM: the loads are issued from inline assembly, invisible to the compiler's wait-count pass (which otherwise waits for ALL vector
    // memory operations -- the weight DMA just issued included -- before the first use of the previous chunk's values); the chunk body
    // waits for them by count itself
    auto gather_issue = [&](f32x4 (&gv)[2][2], bool (&gok)[2], const bool ASM) {
        const int dy = ld_ty * p.dil, dx = ld_tx * p.dil;
        const int toff = (dy * p.W + dx) * p.in_stride + ld_c * BKC;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gok[j] = (unsigned)(g_iy0[j] + dy) < (unsigned)p.H && (unsigned)(g_ix0[j] + dx) < (unsigned)p.W;
            const float* src = in + (gok[j] ? g_off[j] + toff : gq * 8);
            if (X6_KO & 8) continue;
            if (ASM) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv[j][0]) : "v"(src) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(gv[j][1]) : "v"(src) : "memory");
                continue;
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) gv[j][hh] = *reinterpret_cast<const f32x4*>(src + 4 * hh);
        }
    };
    auto advance_ld = [&]() {
        ++ld_tap;
        if (++ld_tx == p.KW) {
            ld_tx = 0;
            ++ld_ty;
        }
        if (ld_tap == taps) {
            ld_tap = ld_ty = ld_tx = 0;
            ++ld_c;
        }
    };
    auto advance_fin = [&]() {
        if (++fin_tap == taps) {
            fin_tap = 0;
            ++fin_c;
        }
    };
    // quarter `part` (pixel j = part >> 1, channel half hh = part & 1) of this thread's 16 values: transform, split into the planes
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    auto finish_part = [&](int part, const f32x4 (&gv)[2][2], const bool (&gok)[2], int buf) {
        if (X6_KO & 1) return;
        const int j = part >> 1, hh = part & 1;
        bf16x4 oh, om_, ol;
        f32x4 v = gv[j][hh];
        if (xf) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(xtab + fin_c * BKC + gq * 8 + 4 * hh);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(xtab + p.cin_g + fin_c * BKC + gq * 8 + 4 * hh);
            // (element by element: packed fp32 instructions do not run beside the MFMAs, scripts/ubench/mfma16_valu_overlap.hip)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        }
        if (!gok[j]) v = zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h, m, l;
            split3(v[e], h, m, l);
            oh[e] = h;
            om_[e] = m;
            ol[e] = l;
        }
        if (X6_KO & 32) return;
        u16* Ad = As + buf * NP * A_STAGE + a_wr0 + j * 64 * BKC + 4 * hh;
        *reinterpret_cast<bf16x4*>(Ad) = oh;
        *reinterpret_cast<bf16x4*>(Ad + A_STAGE) = om_;
        *reinterpret_cast<bf16x4*>(Ad + 2 * A_STAGE) = ol;
    };
    auto b_issue = [&](int chunk, int buf) {
        if (X6_KO & 4) return;
        u16* dst = Bs + (NBS == 2 ? buf : 0) * NP * B_STAGE;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const u16* src = wslab + (size_t)pl * plane_elems + (size_t)chunk * B_STAGE + lane * 8;
#pragma unroll
            for (int j = 0; j < B_STAGE / 512 / 4; ++j) {
                const int piece = j * 4 + wave;                        // 1-KiB pieces of the plane's tile
                dma16(src + piece * 512, dst + pl * B_STAGE + piece * 512);
            }
        }
    };

    gather_issue(gvA, gokA, false);                // chunk 0
    b_issue(0, 0);
    if (nchunks > 1) advance_ld();
    if (xf) __syncthreads();                       // the scale / shift table
#pragma unroll
    for (int part = 0; part < 4; ++part) finish_part(part, gvA, gokA, 0);
    if (nchunks > 1) advance_fin();
    gather_issue(gvA, gokA, false);                // chunk 1 (or chunk 0 again)
    if (nchunks > 2) advance_ld();
    __syncthreads();

    auto chunk = [&](const int ch, f32x4 (&gv_use)[2][2], const bool (&gok_use)[2], f32x4 (&gv_ld)[2][2], bool (&gok_ld)[2]) {
        const int buf = ch & 1;
        const u16* Ab = As + buf * NP * A_STAGE + wm * WTM * BKC + fo;
        const u16* Bb = Bs + (NBS == 2 ? buf : 0) * NP * B_STAGE + wn * WTN * BKC + fo;
        bf16x8 afr[2][NP], bf[NP][NT];          // activation fragments row by row (two rows in registers), weight fragments all at once
        auto a_row = [&](int i) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                if (X6_KO & 16) asm volatile("" : "=v"(afr[i & 1][pl]));
                else afr[i & 1][pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_STAGE + i * 16 * BKC);
            }
        };
        a_row(0);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (X6_KO & 16) {
                    asm volatile("" : "=v"(bf[pl][j]));
                    continue;
                }
                bf[pl][j] = *reinterpret_cast<const bf16x8*>(Bb + pl * B_STAGE + j * 16 * BKC);
            }
        }
        if (NBS == 1) {
            // one weight buffer: every wave holds its fragments in registers before the next chunk's planes may land
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        b_issue(min(ch + 1, nchunks - 1), buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        gather_issue(gv_ld, gok_ld, true);         // chunk ch + 2: the four youngest vector-memory operations of the chunk
        if (!(X6_KO & 8)) {
            // chunk ch + 1's values (loaded during the previous chunk): everything older than this chunk's DMA pieces and loads
            constexpr int YOUNGER = ((X6_KO & 4) ? 0 : NP * (B_STAGE / 512 / 4)) + 4;
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(gv_use[0][0]), "+v"(gv_use[0][1]), "+v"(gv_use[1][0]), "+v"(gv_use[1][1]) : "n"(YOUNGER));
        }
        __builtin_amdgcn_sched_barrier(0);
        // six products per fragment pair, smallest first (a: activation planes, b: weight planes).  The bf16 MFMA's adder truncates: summing
        // the six products of a chunk from zero and adding the chunk's sum to the running sum with the vector ALU (round to nearest) keeps the
        // result closer to float64 than the fp32-MFMA kernels (3-4 x closer than accumulating in place: scripts/bench_conv_x6.py)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (i + 1 < MT) a_row(i + 1);
            const bf16x8 (&af)[NP] = afr[i & 1];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (X6_KO & 2) continue;
#if X6_LOCAL_SUM
                f32x4 c = zero4;
#else
                f32x4 c = acc[i][j];
#endif
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][j], af[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2][j], af[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][j], af[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[0], c, 0, 0, 0);
#if X6_LOCAL_SUM
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] += c[e];
#else
                acc[i][j] = c;
#endif
            }
            finish_part(i, gv_use, gok_use, buf ^ 1);
        }
#if X6_SCHED
        // one MFMA, then the vector instructions that fit into its 16 cycles
#define X6_ROW(DS_READS)                                                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x100, DS_READS, 0); /* next row's fragments, scale / shift */                    \
    _Pragma("unroll") for (int k = 0; k < NT * 6; ++k) {                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x002, BN == 64 ? X6_SCHED + 2 : BN == 128 ? X6_SCHED + 1 : X6_SCHED, 0);     \
    }                                                                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x200, NP, 0); /* the quarter's planes */
        static_assert(MT == 4, "the schedule below is written for four fragment rows");
        X6_ROW(NP + (XF ? 2 : 0))
        X6_ROW(NP + (XF ? 2 : 0))
        X6_ROW(NP + (XF ? 2 : 0))
        X6_ROW((XF ? 2 : 0))
#undef X6_ROW
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 2 < nchunks) advance_fin();
        if (ch + 3 < nchunks) advance_ld();
        // the next chunk's weights have landed and its planes are written; the activation loads of chunk ch + 2 stay in flight
        if (X6_KO & 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // two chunks per trip: the register sets of the activation loads alternate (no copies, no wait for a load before its chunk)
    for (int ch = 0; ch < nchunks; ch += 2) {
        chunk(ch, gvA, gokA, gvB, gokB);
        if (ch + 1 < nchunks) chunk(ch + 1, gvB, gokB, gvA, gokA);
    }
    // The last chunk ends with `s_waitcnt vmcnt(4)`: its four look-ahead loads (the last chunk again) are still in flight, and they were
    // issued from inline assembly, so the compiler's wait-count pass does not know about them.  gvA / gvB are dead from here on and their
    // registers may be handed to epilogue temporaries: drain the loads before anything else can live there.  gvA / gvB are
    // never copied between gather_issue() and the counted wait that names them as "+v" operands.
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(gvA[0][0]), "+v"(gvA[0][1]), "+v"(gvA[1][0]), "+v"(gvA[1][1]),
                                        "+v"(gvB[0][0]), "+v"(gvB[0][1]), "+v"(gvB[1][0]), "+v"(gvB[1][1]) : : "memory");
    __syncthreads();

    // ---- epilogue: + bias, batch sums of the pre-activation output, ReLU, 16-byte NHWC stores (lane: pixel r, 8 consecutive channels
    //      per tile pair) ------------------------------------------------------------------------------------------------------------
    float* red = reinterpret_cast<float*>(smem_h);          // [2 wm][BN][2]
    const float gate = p.gate ? *p.gate : 0.f;
    const bool split_t = p.out_mode == GSSD_OUT_SPLIT_T && n0g >= p.split_n;      // workgroup-uniform: split_n is a multiple of the tile
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
        const int nl = wn * WTN + 32 * u + 8 * kq;            // channel inside the tile
        const int ng = n0g + nl;                              // ... inside the group
        const bool n_ok = ng + 8 <= cout_g;
        const int n = g * cout_g + ng;
        float bv[8], av[8], ssum[8], ssq[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            bv[c] = (p.bias && n_ok) ? p.bias[n + c] : 0.f;
            av[c] = (p.alpha && n_ok) ? p.alpha[n + c] : 1.f;
            ssum[c] = ssq[c] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * WTM + i * 16 + r;
            if (m >= M || !n_ok) continue;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v[c] = acc[i][2 * u + (c >> 2)][c & 3] * av[c] + bv[c];
                ssum[c] += v[c];
                ssq[c] += v[c] * v[c];
            }
            if (split_t) {
                // second column range of a merged projection: per image [n - split_n][pixel] (the row tails up to out_b_stride are never
                // written: the caller zero-fills the buffer once)
                const int bi = m / HoWo, ml = m - bi * HoWo;
                float* dst = p.out_b + (size_t)bi * p.outb_batch_stride + (size_t)(n - p.split_n) * p.out_b_stride + ml;
#pragma unroll
                for (int c = 0; c < 8; ++c) dst[(size_t)c * p.out_b_stride] = v[c];
                continue;
            }
            // conv_igemm's epilogue order: gate, second output, residual, ReLU
            const size_t o = (size_t)m * p.out_stride + p.out_ch_off + n;
            if (p.gate) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] *= gate;
                if (p.out2) {
                    *reinterpret_cast<f32x4*>(p.out2 + o) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(p.out2 + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
            if (p.resid) {
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(p.resid + o), r1 = *reinterpret_cast<const f32x4*>(p.resid + o + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] += r0[c];
                    v[4 + c] += r1[c];
                }
            }
            if (p.relu) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            *reinterpret_cast<f32x4*>(p.out + o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(p.out + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (p.stats) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float s = ssum[c], q = ssq[c];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s += __shfl_xor(s, o, 64);
                    q += __shfl_xor(q, o, 64);
                }
                if (r == 0) {
                    red[(wm * BN + nl + c) * 2 + 0] = s;
                    red[(wm * BN + nl + c) * 2 + 1] = q;
                }
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        if (tid < BN && n0g + tid < cout_g) {
            const double s = (double)red[tid * 2 + 0] + (double)red[(BN + tid) * 2 + 0];
            const double q = (double)red[tid * 2 + 1] + (double)red[(BN + tid) * 2 + 1];
            const int n = g * cout_g + n0g + tid;
            double* st = gssd_stats_replica(p.stats, p.stats_rep, p.Cout);
            unsafeAtomicAdd(st + n, s);
            unsafeAtomicAdd(st + p.Cout + n, q);
        }
    }
}

// packed fp32 rows [Cout][row_stride] (k = tap * cin_g + c) -> three bf16 planes in the kernel's DMA order; rows beyond cout_g zero
__global__ void conv_x6_pack_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int groups, int cin_g, int taps, int row_stride,
                                    int BN, long long total) {
    const int cout_g = Cout / groups, ntn = (cout_g + BN - 1) / BN, nchunks = (cin_g / BKC) * taps;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int slot = (int)((i >> 3) & 3);
        const int row = (int)((i >> 5) % BN);
        long long t = (i >> 5) / BN;
        const int chunk = (int)(t % nchunks);
        t /= nchunks;
        const int nt = (int)(t % ntn), g = (int)(t / ntn);
        const int q = slot ^ swz(row);
        const int tap = chunk % taps, c32 = chunk / taps;
        const int c = c32 * BKC + q * 8 + e;
        const int ng = nt * BN + chan_of_row(row);
        __bf16 h, m, l;
        split3(ng < cout_g ? w[(size_t)(g * cout_g + ng) * row_stride + tap * cin_g + c] : 0.f, h, m, l);
        wp[i] = __builtin_bit_cast(u16, h);
        wp[i + total] = __builtin_bit_cast(u16, m);
        wp[i + 2 * total] = __builtin_bit_cast(u16, l);
    }
}

template <int BN, bool XF>
int launch(const gssd_conv_desc& d, int M, hipStream_t stream) {
    static unsigned attr_mask = 0;
    auto kern = conv_x6_kernel<BN, XF>;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<BN>::LDS_BYTES + 4096) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", Cfg<BN>::LDS_BYTES + 4096);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask);
    }
    const int cout_g = d.Cout / d.groups;
    const int ntn = (cout_g + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    const long long plane = (long long)d.groups * ntn * BN * d.KH * d.KW * d.cin_g;
    hipLaunchKernelGGL(kern, dim3((mtiles + 7) / 8 * 8 * d.groups * ntn), dim3(256), Cfg<BN>::LDS_BYTES + (d.in_scale ? 8 * d.cin_g : 0), stream, d, M, ntn, mtiles, plane);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

bool shape_ok(int cin_g, int cout_g, int groups) {
    return cin_g > 0 && cin_g % BKC == 0 && cout_g >= 32 && cout_g % 8 == 0 && groups > 0;
}

}  // namespace

// N tile of a (cout_g, groups, M) launch.  GSSD_X6_BN overrides (experiments); the packed weights depend on it, so it is read once.
extern "C" int gssd_conv_x6_tile(int cout_g, int groups, long long M) {
    static const int forced = getenv("GSSD_X6_BN") ? atoi(getenv("GSSD_X6_BN")) : 0;
    if (forced == 64 || forced == 128 || forced == 256) return forced;
    (void)groups;
    (void)M;
    // measured on the trunk shapes at B = 32 (scripts/bench_conv_x6.py, GSSD_X6_BN sweep): two resident 128-column workgroups beat one
    // 256-column workgroup everywhere (conv6 174 vs 248 us, fuse_11 166 vs 193); the 256 instance stays for the sweep
    return cout_g > 64 ? 128 : 64;
}

extern "C" long long gssd_conv_x6_weight_elems(int Cout, int groups, int cin_g, int taps, int BN) {         // bf16 elements (three planes)
    if (Cout <= 0 || groups <= 0 || Cout % groups != 0 || taps <= 0 || !(BN == 64 || BN == 128 || BN == 256)) return -1;
    if (!shape_ok(cin_g, Cout / groups, groups)) return -1;
    const int cout_g = Cout / groups;
    return 3ll * groups * ((cout_g + BN - 1) / BN) * BN * taps * cin_g;
}

extern "C" int gssd_conv_x6_pack_weight(const float* w_packed, void* w_x6, int Cout, int groups, int cin_g, int taps, int row_stride, int BN,
                                        gssd_stream_t stream) {
    const long long n = gssd_conv_x6_weight_elems(Cout, groups, cin_g, taps, BN);
    GSSD_CHECK_ARG(w_packed && w_x6 && n > 0 && row_stride >= taps * cin_g);
    const long long total = n / 3;
    hipLaunchKernelGGL(conv_x6_pack_kernel, dim3((int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_packed, reinterpret_cast<u16*>(w_x6), Cout, groups, cin_g, taps, row_stride, BN, total);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// 1 when gssd_conv2d_nhwc_f32 runs this descriptor here (it needs d->wgt_x6 packed for gssd_conv_x6_tile(...)'s tile)
extern "C" int gssd_conv_x6_takes(const gssd_conv_desc* dp) {
    if (!dp) return 0;
    const gssd_conv_desc& d = *dp;
    if (!d.wgt_x6 || d.groups <= 0 || d.Cout % d.groups != 0) return 0;
    if (!shape_ok(d.cin_g, d.Cout / d.groups, d.groups)) return 0;
    if (d.split_k > 1 || (d.flags & ~GSSD_CONV_OUT_F32) || (d.out2 && !d.gate)) return 0;
    if (d.out_mode == GSSD_OUT_SPLIT_T) {
        // merged Self_Attn projection: columns [0, split_n) NHWC, the rest transposed per image; whole tiles on either side
        const int bn = gssd_conv_x6_tile(d.Cout / d.groups, d.groups, (long long)d.B * d.Ho * d.Wo);
        if (d.groups != 1 || !d.out_b || d.split_n <= 0 || d.split_n % bn != 0 || d.out_b_stride < d.Ho * d.Wo) return 0;
    } else if (d.out_mode != GSSD_OUT_NHWC) {
        return 0;
    }
    if (d.m_per_image) {
        // per-image descriptors of contiguous batches are the same flat M range
        if (d.wgt_batch_stride != 0 || d.in_batch_stride != (long long)d.H * d.W * d.in_stride ||
            d.out_batch_stride != (long long)d.Ho * d.Wo * d.out_stride)
            return 0;
    }
    if ((d.resid && ((uintptr_t)d.resid % 16)) || (d.out2 && ((uintptr_t)d.out2 % 16))) return 0;
    if (d.in_stride % 4 || d.in_ch_off % 4 || d.out_stride % 4 || d.out_ch_off % 4) return 0;
    if (((uintptr_t)d.in % 16) || ((uintptr_t)d.out % 16) || ((uintptr_t)d.wgt_x6 % 16)) return 0;
    if ((long long)d.B * d.H * d.W * d.in_stride >= (1ll << 31)) return 0;
    if (d.in_scale && d.cin_g > 512) return 0;               // the LDS table of the fused input transform
    return 1;
}

int gssd_try_conv_x6(const gssd_conv_desc& d, hipStream_t stream) {
    if (!gssd_conv_x6_takes(&d)) return 1;
    const long long Mll = (long long)d.B * d.Ho * d.Wo;
    const int M = (int)Mll;
    switch (gssd_conv_x6_tile(d.Cout / d.groups, d.groups, Mll)) {
        case 64: return d.in_scale ? launch<64, true>(d, M, stream) : launch<64, false>(d, M, stream);
        case 128: return d.in_scale ? launch<128, true>(d, M, stream) : launch<128, false>(d, M, stream);
        default: return d.in_scale ? launch<256, true>(d, M, stream) : launch<256, false>(d, M, stream);
    }
}
