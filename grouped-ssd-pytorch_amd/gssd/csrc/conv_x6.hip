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
#include <type_traits>
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// the same split of TWO values at once, planes as packed bf16 pairs (a in the low half): one v_cvt_pk_bf16_f32 per plane and pair
__device__ __forceinline__ void split3_pair(const float a, const float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
    const float sa = ra - __builtin_bit_cast(float, pm << 16), sb = rb - __builtin_bit_cast(float, pm & 0xffff0000u);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sa, sb}, bf16x2));
}

// Round 6, the three-MFMA form for FORWARD launches on bounded activations (GSSD_CONV_F16_OK set by the
// caller): three fp16 planes per operand -- h = fp16(x), h6 = h / 64, l6 = fp16((x - h) * 64), x = h + l6 / 64 to 2^-24 |x| -- and the products
// h h' + l6 h6' + h6 l6' (dcn_x6.hip).  Same planes, LDS images and DMA pieces as the bf16 form, half the matrix instructions.  The packed weights
// hold both forms (the bf16 planes, then the fp16 planes); data gradients keep the bf16 planes.  GSSD_X6_F16=0: bf16 everywhere.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3h_pair(const float a, const float b, unsigned& ph, unsigned& p6, unsigned& pl) {
    const f16x2_t h = __builtin_convertvector(f32x2{a, b}, f16x2_t);
    const f32x2 r = (f32x2{a, b} - __builtin_convertvector(h, f32x2)) * 64.f;
    ph = __builtin_bit_cast(unsigned, h);
    p6 = __builtin_bit_cast(unsigned, h * f16x2_t{(_Float16)0.015625f, (_Float16)0.015625f});
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}

#ifndef X6_V2
#define X6_V2 1             // round 5's K loop for the 64- / 128-column tiles (conv_x6_v2_kernel); 0: round 4's (conv_x6_kernel, also the 256-column sweep instance)
#endif

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

#if X6_V2
// ---- round 5: the K loop of dcn_x6.hip's v2 kernel for the plain convolution ---------------------------------------------------------------
// Round 4's loop read all weight fragments of a chunk (+ the first activation row) in front of a barrier, then issued the weight DMA and the
// activation loads in a burst; "feeding" and MFMAs added up (profiles/r04_f_conv_x6_knockout.txt).  Here, as in dcn_x6.hip (see there):
//   * iteration `it` = [part A: the column tiles of the second half of chunk it - 1] [part B: the first half of chunk it];
//   * ONE activation stage (the four rows' fragments are reloaded in place behind the last column tile's MFMAs), the weight planes in four
//     half buffers (X: first half of the wave's column tiles, Y: second half; two of each): what an iteration reads was DMA'd during the
//     previous one -- 24 + 48 KB for 128 columns as before, 24 + 24 KB for 64;
//   * column tile outer, fragment row inner; the weight fragments of the next tile are prefetched behind the current tile's MFMAs;
//   * every memory request sits behind its own group of six MFMAs; waits are counted (vmcnt is in issue order);
//   * two barriers per iteration: B1 in front of part A's last column tile (the previous iteration's planes and X pieces are in LDS),
//     B2 behind part B's first group (every wave holds its fragments, its loads and Y pieces have landed).
template <int BN>
struct Cfg2 {
    static constexpr int WTN = BN / 2, NT = WTN / 16, NTH = NT / 2, MT = 4;
    static constexpr int NG = NTH * MT;                   // groups of six MFMAs per part: 8 / 4
    static constexpr int HB_ROWS = BN / 2, HB_PLANE = HB_ROWS * BKC, HB_ELEMS = NP * HB_PLANE;
    static constexpr int PH = NP * HB_ROWS / 16;          // 1-KiB pieces per half: 12 / 6
    static constexpr int DPW = 2 * PH / 4;                // pieces per wave and iteration: 6 / 3
    static constexpr int NY = (PH + 3) / 4;               // the wave's first NY pieces cover its share of the Y half: 3 / 2
    static constexpr int LDS_BYTES = (NP * A_STAGE + 4 * HB_ELEMS) * 2;      // + the scale / shift table
    static_assert(BN == 64 || BN == 128, "tiles");
};

template <int BN, bool XF, bool F16>
__global__ __launch_bounds__(256, 2) void conv_x6_v2_kernel(const gssd_conv_desc p, const int M, const int ntn, const int mtiles,
                                                           const long long plane_elems) {
    using K = Cfg2<BN>;
    // F16: TWO planes travel through LDS and the weight DMA -- h and l6 = (x - h) * 64; h6 = h / 64 is made from h in registers behind the fragment
    // reads (four packed multiplies per fragment).  Why: at three planes a chunk moves 144 KB through the CU's LDS port (fragment reads 96, plane
    // writes 24, DMA 24) = 1 125 cycles at 128 B per cycle against 768 cycles of MFMAs; two planes: 96 KB = 750.  The LDS images keep their
    // three-plane strides; the piece counts of the schedule below follow NPL.  Measured (scripts/bench_conv_x6.py, same box): 64-column tiles
    // -20 % (conv3_1 241 -> 193 us, conv3_2 396 -> 324); 128-column tiles +2 .. +5 % on 3x3 launches (conv4_2 229 -> 241, the packed multiplies
    // sit between twice as many MFMAs per fragment read) -- those keep the third plane in LDS.  fp16 plane order in memory: h, l6, h6.
    constexpr int NPL = (F16 && BN == 64) ? 2 : NP;
    constexpr int WTM = 64, WTN = K::WTN, MT = K::MT, NT = K::NT, NTH = K::NTH, NG = K::NG;
    constexpr int PH = NPL * K::HB_ROWS / 16;             // 1-KiB pieces per half: 12 / 6 (F16: 8 / 4)
    constexpr int DPW = 2 * PH / 4;                       // pieces per wave and iteration: 6 / 3 (4 / 2)
    constexpr int NY = (PH + 3) / 4;                      // the wave's first NY pieces cover its share of the Y half: 3 / 2 (2 / 1)
    constexpr int HB_PLANE = K::HB_PLANE, HB_ELEMS = K::HB_ELEMS, B_STAGE = BN * BKC;
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    u16* const As = smem_h;                                   // [3][BM][32]
    u16* const Bh = smem_h + NP * A_STAGE;                    // [X | Y][2][3][BN / 2][32]
    float* const xtab = reinterpret_cast<float*>(smem_h + NP * A_STAGE + 4 * HB_ELEMS);      // [2][cin_g] scale | shift
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
    const u16* wslab = reinterpret_cast<const u16*>(p.wgt_x6) + (F16 ? 3 * plane_elems : 0) + (size_t)by * nchunks * B_STAGE;      // plane 0; plane q at + q * plane_elems (the fp16 planes lie behind the bf16 planes)
    constexpr bool xf = XF;

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    // gather roles: thread -> (pixel rows gp and gp + 64, 8-channel slot gq)
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
    if (xf) {
        const float* xsc = p.in_scale + p.in_ch_off + g * p.cin_g;
        const float* xsh = p.in_shift + p.in_ch_off + g * p.cin_g;
        for (int c = tid; c < p.cin_g; c += 256) {
            xtab[c] = xsc[c];
            xtab[p.cin_g + c] = xsh[c];
        }
    }

    // ---- activation loads of one chunk: 2 pixels x 2 halves of 4 fp32 channels per thread, by inline assembly (counted waits) ----------------
    int ld_ty = 0, ld_tx = 0, ld_c = 0, ld_tap = 0;          // the chunk the next loads are for = the chunk split next
    f32x4 gv[2][2];
    bool gok[2] = {false, false};
    const float* src[2] = {in, in};
    gv[0][0] = gv[0][1] = gv[1][0] = gv[1][1] = zero4;
    auto load_addr = [&]() {
        const int dy = ld_ty * p.dil, dx = ld_tx * p.dil;
        const int toff = (dy * p.W + dx) * p.in_stride + ld_c * BKC;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gok[j] = (unsigned)(g_iy0[j] + dy) < (unsigned)p.H && (unsigned)(g_ix0[j] + dx) < (unsigned)p.W;
            src[j] = in + (gok[j] ? g_off[j] + toff : gq * 8);
        }
    };
    auto load_req = [&](int q) {                             // request q = (pixel, half): 16 bytes per lane
        if (X6_KO & 8) return;
        const int j = q >> 1;
        if (q & 1) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(gv[j][1]) : "v"(src[j]) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv[j][0]) : "v"(src[j]) : "memory");
    };
    int fin_c = 0;                                           // 32-channel block of the chunk in gv (for the scale / shift table)
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
    // quarter `part` (pixel part >> 1, channel half part & 1) of this thread's 16 values: deferred BatchNorm + ReLU, split, 8 bytes per plane
    auto finish_part = [&](int part) {
        if (X6_KO & 1) return;
        const int j = part >> 1, hh = part & 1;
        f32x4 v = gv[j][hh];
        if (xf) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(xtab + fin_c * BKC + gq * 8 + 4 * hh);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(xtab + p.cin_g + fin_c * BKC + gq * 8 + 4 * hh);
            // (element by element: packed fp32 instructions do not run beside the MFMAs, scripts/ubench/mfma16_valu_overlap.hip)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        }
        if (!gok[j]) v = zero4;
        unsigned h0, m0_, l0, h1, m1, l1;
        if constexpr (F16) {
            split3h_pair(v[0], v[1], h0, m0_, l0);
            split3h_pair(v[2], v[3], h1, m1, l1);
        } else {
            split3_pair(v[0], v[1], h0, m0_, l0);
            split3_pair(v[2], v[3], h1, m1, l1);
        }
        if (X6_KO & 32) return;
        u16* Ad = As + a_wr0 + j * 64 * BKC + 4 * hh;
        *reinterpret_cast<u32x2*>(Ad) = u32x2{h0, h1};
        if constexpr (F16) {
            *reinterpret_cast<u32x2*>(Ad + A_STAGE) = u32x2{l0, l1};
            if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(Ad + 2 * A_STAGE) = u32x2{m0_, m1};
        } else {
            *reinterpret_cast<u32x2*>(Ad + A_STAGE) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(Ad + 2 * A_STAGE) = u32x2{l0, l1};
        }
    };
    // piece q (0 .. DPW - 1) of this wave in an iteration: the first PH pieces of the iteration's list are the Y half of chunk cy, the rest the
    // X half of chunk cx; piece = (plane, wave column, column tile inside the half)
    auto dma_piece = [&](int q, int cy, int cx, int par) {
        if (X6_KO & 4) return;
        const int idx = q * 4 + wave;
        const bool isx = idx >= PH;
        const int pc = isx ? idx - PH : idx;
        const int pl = pc / (2 * NTH), g8 = pc - pl * (2 * NTH);
        const int half = isx ? 0 : 1, parity = isx ? par ^ 1 : par;
        const int G = (g8 / NTH) * NT + half * NTH + (g8 % NTH);       // 16-row group of the plane's [BN][32] tile
        const u16* s = wslab + (size_t)(isx ? cx : cy) * B_STAGE + (size_t)pl * plane_elems + G * 512 + lane * 8;
        dma16(s, Bh + (half * 2 + parity) * HB_ELEMS + pl * HB_PLANE + g8 * 512);
    };

    bf16x8 areg[MT][NP], breg[2][NP];
    // F16: LDS planes (h, l6) -> registers [0] and [2]; [1] = h / 64 (exact; where it leaves fp16's normal range the product it enters is below
    // 2^-24 of the leading one)
    auto derive = [&](bf16x8& h6, const bf16x8& h) {
        const f16x8_t sc = {(_Float16)0.015625f, (_Float16)0.015625f, (_Float16)0.015625f, (_Float16)0.015625f,
                            (_Float16)0.015625f, (_Float16)0.015625f, (_Float16)0.015625f, (_Float16)0.015625f};
        h6 = __builtin_bit_cast(bf16x8, __builtin_bit_cast(f16x8_t, h) * sc);
    };
    auto a_load_row = [&](int i) {
        const u16* Ab = As + (wm * WTM + i * 16) * BKC + fo;
        if constexpr (F16) {
            if (X6_KO & 16) {
                asm volatile("" : "=v"(areg[i][0]));
                asm volatile("" : "=v"(areg[i][2]));
            } else {
                areg[i][0] = *reinterpret_cast<const bf16x8*>(Ab);
                areg[i][2] = *reinterpret_cast<const bf16x8*>(Ab + A_STAGE);
                if constexpr (NPL == 3) areg[i][1] = *reinterpret_cast<const bf16x8*>(Ab + 2 * A_STAGE);
            }
            if constexpr (NPL == 2) derive(areg[i][1], areg[i][0]);
            else if (X6_KO & 16) asm volatile("" : "=v"(areg[i][1]));
            return;
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if (X6_KO & 16) asm volatile("" : "=v"(areg[i][pl]));
            else areg[i][pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_STAGE);
        }
    };
    auto b_load = [&](int which, int half, int parity, int jj) {
        const u16* Bb = Bh + (half * 2 + parity) * HB_ELEMS + (wn * NTH * 16 + jj * 16) * BKC + fo;
        if constexpr (F16) {
            if (X6_KO & 16) {
                asm volatile("" : "=v"(breg[which][0]));
                asm volatile("" : "=v"(breg[which][2]));
            } else {
                breg[which][0] = *reinterpret_cast<const bf16x8*>(Bb);
                breg[which][2] = *reinterpret_cast<const bf16x8*>(Bb + HB_PLANE);
                if constexpr (NPL == 3) breg[which][1] = *reinterpret_cast<const bf16x8*>(Bb + 2 * HB_PLANE);
            }
            if constexpr (NPL == 2) derive(breg[which][1], breg[which][0]);
            else if (X6_KO & 16) asm volatile("" : "=v"(breg[which][1]));
            return;
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if (X6_KO & 16) asm volatile("" : "=v"(breg[which][pl]));
            else breg[which][pl] = *reinterpret_cast<const bf16x8*>(Bb + pl * HB_PLANE);
        }
    };
    // six products per fragment pair, smallest first (first operand: weight planes).  The bf16 MFMA's adder truncates: summing the six products
    // of a chunk from zero and adding the chunk's sum to the running sum with the vector ALU (round to nearest) keeps the result closer to
    // float64 than the fp32-MFMA kernels (3-4 x closer than accumulating in place: scripts/bench_conv_x6.py)
    auto mma_row = [&](int i, int j, int which) {
        if (X6_KO & 2) return;
#if X6_LOCAL_SUM
        f32x4 c = zero4;
#else
        f32x4 c = acc[i][j];
#endif
        if constexpr (F16) {
            // planes 0: h, 1: h / 64, 2: (x - h) * 64 -- l6 h6', h6 l6', h h'
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, breg[which][2]), __builtin_bit_cast(f16x8_t, areg[i][1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, breg[which][1]), __builtin_bit_cast(f16x8_t, areg[i][2]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, breg[which][0]), __builtin_bit_cast(f16x8_t, areg[i][0]), c, 0, 0, 0);
        } else {
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][1], areg[i][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][2], areg[i][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][1], areg[i][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][0], c, 0, 0, 0);
        }
#if X6_LOCAL_SUM
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] += c[e];
#else
        acc[i][j] = c;
#endif
    };
    constexpr int NGQ = (X6_KO & 8) ? 0 : 4, NDQ = (X6_KO & 4) ? 0 : DPW, NYQ = (X6_KO & 4) ? 0 : NY;
    // all four loads have landed; N younger DMA pieces of this wave may still be in flight
#define X6_LOADS_WAIT(N) \
    if (!(X6_KO & 8)) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(gv[0][0]), "+v"(gv[0][1]), "+v"(gv[1][0]), "+v"(gv[1][1]) : "n"(N))

    // ---- prologue: planes of chunk 0, X half of chunk 0's weights --------------------------------------------------------------------------------
    load_addr();
#pragma unroll
    for (int q = 0; q < 4; ++q) load_req(q);
    if (xf) __syncthreads();                       // the scale / shift table
    if (!(X6_KO & 4)) {
        // X half of chunk 0: PH pieces over four waves (the counts differ by wave when PH = 6: everything is awaited)
        for (int pc = wave; pc < PH; pc += 4) {
            const int pl = pc / (2 * NTH), g8 = pc - pl * (2 * NTH);
            const int G = (g8 / NTH) * NT + (g8 % NTH);
            dma16(wslab + (size_t)pl * plane_elems + G * 512 + lane * 8, Bh + pl * HB_PLANE + g8 * 512);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(gv[0][0]), "+v"(gv[0][1]), "+v"(gv[1][0]), "+v"(gv[1][1]));
#pragma unroll
    for (int part = 0; part < 4; ++part) finish_part(part);
    if (nchunks > 1) {
        advance_ld();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // Issue order of a wave's memory operations in an iteration (128 columns: 8 groups per part, 6 pieces; 64 columns: 4 groups, 3 pieces):
    //   128: g0: D0 L0, g1: D1 L1, g2: D2 L2, g3: L3 | B1 | g4: D3, g5: D4, g6: D5      B1: vmcnt(NY + 4), B2: vmcnt(DPW - NY)
    //    64: | B1 | g0: D0 L0 L1, g1: D1 L2 L3, g2: D2                                   B1: vmcnt(0),      B2: vmcnt(DPW - NY)
    // (D0 .. D(NY-1) cover the wave's share of the Y half; L = activation load)
    auto iteration = [&](int it, auto first_c, auto last_c) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        const int par = it & 1;
        const int cy = min(it, nchunks - 1), cx = min(it + 1, nchunks - 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- part A ----
        if (!FIRST) {
#pragma unroll
            for (int jj = 0; jj < NTH; ++jj) {
                if (jj + 1 == NTH && !LAST) {
                    // B1: the previous iteration's planes and X pieces are in LDS (every wave waited for its own), its table reads are done
                    constexpr int K_B1 = NTH == 1 ? 0 : NYQ + NGQ;
                    if (X6_KO & 64) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K_B1) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(K_B1) : "memory");
                }
                if (jj + 1 < NTH) b_load((jj + 1) & 1, 1, par ^ 1, jj + 1);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int gidx = jj * MT + i;            // group 0 .. NG - 1: six MFMAs
                    if (!LAST && gidx == 0) load_addr();
                    if (jj + 1 == NTH && !LAST && i == 0) b_load((jj + 1) & 1, 0, par, 0);          // X of chunk it (behind B1)
                    mma_row(i, NTH + jj, jj & 1);
                    if (!LAST) {
                        if (NTH == 2) {
                            if (gidx < NY) dma_piece(gidx, cy, cx, par);
                            if (gidx < 4) load_req(gidx);
                            if (gidx >= 4 && gidx < 4 + DPW - NY) dma_piece(gidx - 4 + NY, cy, cx, par);
                        } else {
                            // (the pieces behind the wave's Y share follow the loads of their group: B2's count leaves exactly them in flight)
                            if (gidx < NY) dma_piece(gidx, cy, cx, par);
                            if (gidx < 2) {
                                load_req(2 * gidx);
                                load_req(2 * gidx + 1);
                            }
                            if (gidx >= NY && gidx < DPW) dma_piece(gidx, cy, cx, par);
                        }
                        if (jj + 1 == NTH) a_load_row(i);    // chunk it's planes, in place behind the row's last use
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            load_addr();
#pragma unroll
            for (int q = 0; q < NY; ++q) dma_piece(q, cy, cx, par);
#pragma unroll
            for (int q = 0; q < 4; ++q) load_req(q);
#pragma unroll
            for (int q = NY; q < DPW; ++q) dma_piece(q, cy, cx, par);
#pragma unroll
            for (int i = 0; i < MT; ++i) a_load_row(i);
            b_load(NTH & 1, 0, par, 0);
        }
        if (!LAST) {
            // ---- part B (weight fragments of tile jj in breg[(NTH + jj) & 1]: part A's last tile left X's first tile in breg[NTH & 1]) ----
#pragma unroll
            for (int jj = 0; jj < NTH; ++jj) {
                if (jj + 1 < NTH) b_load((NTH + jj + 1) & 1, 0, par, jj + 1);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int gidx = jj * MT + i;            // group 0 .. NG - 1: six MFMAs; the four quarters of the thread's values ride behind
                    mma_row(i, jj, (NTH + jj) & 1);
                    if (gidx == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        // B2: every wave holds chunk it's activation fragments; this wave's loads and Y pieces have landed
                        constexpr int K_B2 = NDQ - NYQ;
                        if (X6_KO & 64) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K_B2) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(K_B2) : "memory");
                        X6_LOADS_WAIT(DPW - NY);             // (already true: ties the load registers to the wait for the compiler)
                    }
#ifndef X6_FIN
#define X6_FIN 1        // 1: a quarter rides on TWO groups of MFMAs (scheduling region = groups 2k - 1, 2k); 0: on one
#endif
                    if (NTH == 2) {
                        if (X6_FIN) {
                            if (gidx == 1) finish_part(0);
                            if (gidx == 3) finish_part(1);
                            if (gidx == 5) finish_part(2);
                            if (gidx == 7) finish_part(3);
                        } else {
                            if (gidx == 1) finish_part(0);
                            if (gidx == 2) finish_part(1);
                            if (gidx == 4) finish_part(2);
                            if (gidx == 5) finish_part(3);
                        }
                    } else {
                        if (gidx == 1) finish_part(0), finish_part(1);
                        if (gidx == 2) finish_part(2);
                        if (gidx == 3) finish_part(3);
                    }
                    if (gidx == NG - 1) b_load(0, 1, par, 0);   // Y of chunk it for the next iteration's first column tile
                    if (!(X6_FIN && NTH == 2) || (gidx & 1) == 0 || gidx == NG - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (it + 2 < nchunks) advance_ld();
            fin_c = ld_c;                                    // block of the chunk the next iteration requests (and splits in its part B)
        }
    };
    // fin_c: block of the chunk whose values sit in gv while part B runs = the chunk requested in this iteration's part A (ld_c at that time)
    fin_c = ld_c;
    iteration(0, std::true_type{}, std::false_type{});
    for (int it = 1; it < nchunks; ++it) iteration(it, std::false_type{}, std::false_type{});
    iteration(nchunks, std::false_type{}, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(gv[0][0]), "+v"(gv[0][1]), "+v"(gv[1][0]), "+v"(gv[1][1]) : : "memory");
    __syncthreads();
#undef X6_LOADS_WAIT

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

#endif

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
        const float v = ng < cout_g ? w[(size_t)(g * cout_g + ng) * row_stride + tap * cin_g + c] : 0.f;
        __bf16 h, m, l;
        split3(v, h, m, l);
        wp[i] = __builtin_bit_cast(u16, h);
        wp[i + total] = __builtin_bit_cast(u16, m);
        wp[i + 2 * total] = __builtin_bit_cast(u16, l);
        // behind the bf16 planes: the fp16 planes of the three-MFMA form -- h, (v - h) * 64, h / 64 (64-column tiles make the last one in registers)
        const _Float16 fh = (_Float16)v;
        wp[i + 3 * total] = __builtin_bit_cast(u16, fh);
        wp[i + 4 * total] = __builtin_bit_cast(u16, (_Float16)((v - (float)fh) * 64.f));
        wp[i + 5 * total] = __builtin_bit_cast(u16, (_Float16)(fh * (_Float16)0.015625f));
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

#if X6_V2
template <int BN, bool XF, bool F16>
int launch2_impl(const gssd_conv_desc& d, int M, hipStream_t stream) {
    static unsigned attr_mask = 0;
    auto kern = conv_x6_v2_kernel<BN, XF, F16>;
    constexpr int lds = Cfg2<BN>::LDS_BYTES;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds + 4096) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", lds + 4096);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask);
    }
    const int cout_g = d.Cout / d.groups;
    const int ntn = (cout_g + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    const long long plane = (long long)d.groups * ntn * BN * d.KH * d.KW * d.cin_g;
    hipLaunchKernelGGL(kern, dim3((mtiles + 7) / 8 * 8 * d.groups * ntn), dim3(256), lds + (d.in_scale ? 8 * d.cin_g : 0), stream, d, M, ntn, mtiles, plane);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
// the three-MFMA fp16 form: only launches the CALLER flags GSSD_CONV_F16_OK (operands inside fp16's range: include/gssd_hip.h) -- never inferred
template <int BN, bool XF>
int launch2(const gssd_conv_desc& d, int M, hipStream_t stream) {
    static const bool f16_off = [] { const char* e = getenv("GSSD_X6_F16"); return e && e[0] == '0'; }();
    if (!f16_off && (d.flags & GSSD_CONV_F16_OK)) return launch2_impl<BN, XF, true>(d, M, stream);
    return launch2_impl<BN, XF, false>(d, M, stream);
}
#endif

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
    return 6ll * groups * ((cout_g + BN - 1) / BN) * BN * taps * cin_g;      // three bf16 planes, then three fp16 planes
}

extern "C" int gssd_conv_x6_pack_weight(const float* w_packed, void* w_x6, int Cout, int groups, int cin_g, int taps, int row_stride, int BN,
                                        gssd_stream_t stream) {
    const long long n = gssd_conv_x6_weight_elems(Cout, groups, cin_g, taps, BN);
    GSSD_CHECK_ARG(w_packed && w_x6 && n > 0 && row_stride >= taps * cin_g);
    const long long total = n / 6;
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
    if (d.split_k > 1 || (d.flags & ~(GSSD_CONV_OUT_F32 | GSSD_CONV_F16_OK)) || (d.out2 && !d.gate)) return 0;
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
#if X6_V2
        case 64: return d.in_scale ? launch2<64, true>(d, M, stream) : launch2<64, false>(d, M, stream);
        case 128: return d.in_scale ? launch2<128, true>(d, M, stream) : launch2<128, false>(d, M, stream);
#else
        case 64: return d.in_scale ? launch<64, true>(d, M, stream) : launch<64, false>(d, M, stream);
        case 128: return d.in_scale ? launch<128, true>(d, M, stream) : launch<128, false>(d, M, stream);
#endif
        default: return d.in_scale ? launch<256, true>(d, M, stream) : launch<256, false>(d, M, stream);
    }
}
