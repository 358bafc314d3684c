// fp32 fused modulated deformable 3x3 convolution on the BF16 matrix cores with fp32-equivalent products (round 4; the fp32 mode's
// deformable conv, GSSD_DCN_X6=0 runs dcn_fused.hip): the same algorithm and entry contract as dcn_fused.hip (fp32 x, fp32 offsets, fp32 blend, fp32 weights, fp32 output).
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 matrix rate on gfx950.  An fp32 number is the exact sum of three bf16 numbers
// (x = h + m + l, 8 + 8 + 8 mantissa bits, each rounded to nearest); a product x y is then h h' + (h m' + m h') + (h l' + l h' + m m') + terms
// below 2^-24 |x y| -- six bf16 MFMAs with fp32 accumulation reproduce the fp32 product to the last bit or two, and still cost 3/8 of the
// fp32 instruction's matrix-pipe time.  The sampled column is blended in fp32 exactly as before and split into its three planes when it is
// written to LDS; the weights are split once, when they are packed.
// K loop (end of round 4, X6_PIPE): the corner loads of chunk ch + 1 are requested first thing in chunk ch, and its blend + split + LDS writes
// are dealt out behind the MFMAs of the chunk's last two fragment rows (one wave per SIMD: the vector work runs in the MFMAs' shadow instead of
// after them): 2.60 -> 2.22 ms.
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#ifndef X6_KO
#define X6_KO 0        // knock-outs (scripts/dcn_x6_knockout.sh): 1 no blend / split VALU, 2 no MFMAs, 4 no weight DMA, 8 no x loads, 16 no fragment reads, 32 no barrier
#endif

namespace {

#ifndef X6_BN
#define X6_BN 256
#endif
#ifndef X6_SPLIT_ROWS
#define X6_SPLIT_ROWS 2   // fragment rows of a chunk that carry the next chunk's blend + split: the last 2 (two quarters each), 3 (1, 1, 2) or all 4
#endif
#ifndef X6_V2
#define X6_V2 1        // round 5's K loop (v2::dcn_x6_v2_kernel); 0: round 4's kernel (chunk order tap-fastest)
#endif
#ifndef X6_PIPE
#define X6_PIPE 1         // the pipelined K loop (value = vector instructions scheduled behind every MFMA); 0: the plain loop
#endif
// BN = 256: the sampled columns are computed for two output tiles instead of four (the kernel is bound by the fp32 corner loads); the three
// weight planes of a chunk are then 48 KB, so they are staged in ONE buffer: fragments to registers, barrier, next chunk's DMA behind the MFMAs
constexpr int BM = 128, BN = X6_BN, BKC = 32;          // tile; channels per K chunk: 64-byte bf16 rows
constexpr int WTM = 64, WTN = BN / 2, MT = WTM / 16, NT = WTN / 16;
constexpr int NBS = BN > 128 ? 1 : 2;                 // weight stages
constexpr int NP = 3;                                 // planes of the split
constexpr int A_STAGE = BM * BKC, B_STAGE = BN * BKC;            // u16 elements per plane
constexpr int LDS_BYTES = (2 * NP * A_STAGE + NBS * NP * B_STAGE) * 2 + 9 * BM * 16 + 9 * BM * 4;

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }       // 64-byte rows: conflict-free ds_read_b128

__device__ __forceinline__ int chan_of_row(int row) {        // LDS row of the weight tile -> output channel inside the BN tile
    const int j = row >> 4, rho = row & 15;
    return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
}

// x = h + m + l, each bf16 (round to nearest even); exact to 2^-25 |x|
__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// the same split of TWO values at once, planes as packed bf16 pairs (a in the low half): one v_cvt_pk_bf16_f32 per plane and PAIR, and the
// packed result is the operand dword (the element-wise form converts every element alone and then once more to pack: 7 converts per pair)
__device__ __forceinline__ void split3_pair(const float a, const float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
    const float sa = ra - __builtin_bit_cast(float, pm << 16), sb = rb - __builtin_bit_cast(float, pm & 0xffff0000u);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sa, sb}, bf16x2));
}

// wp: [3 planes][n_tiles][chunks][BN rows in staging order][32] bf16 (slot-swizzled), chunk = (d * cpg/32 + c32) * 9 + tap
__global__ __launch_bounds__(256, 1) void dcn_x6_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                       const u16* __restrict__ wp, const float* __restrict__ bias,
                                                       float* __restrict__ out, int M, int H, int W, int C, int dg, int om_stride,
                                                       int Cout, int ntn, int mtiles, long long plane_elems) {
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    u16* const As = smem_h;                                   // [2][3][BM][32]
    u16* const Bs = smem_h + 2 * NP * A_STAGE;                // [2][3][BN][32]
    f32x4* const setw = reinterpret_cast<f32x4*>(smem_h + 2 * NP * A_STAGE + NBS * NP * B_STAGE);            // [9][BM]
    int* const setp = reinterpret_cast<int*>(smem_h + 2 * NP * A_STAGE + NBS * NP * B_STAGE + 9 * BM * 8);   // [9][BM]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
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
    const int m0 = mt * BM;
    const int HW = H * W, cpg = C / dg, cpc = cpg / BKC;
    const int nchunks = dg * cpc * 9;
    const u16* wslab = wp + (size_t)nt * nchunks * B_STAGE;      // plane 0; plane p at + p * plane_elems

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    // gather roles: thread -> (pixel row pl = (tid >> 2) + 64*j, 8-channel slot q = tid & 3)
    const int gq = tid & 3, gp = tid >> 2;
    const int a_wr0 = gp * BKC + ((gq ^ swz(gp)) << 3);
    const int fo = r * BKC + ((kq ^ swz(r)) << 3);

    auto setups = [&](int d) {
        for (int e = tid; e < 9 * BM; e += 256) {
            const int tap = e / BM, pl = e - tap * BM;
            const int m = m0 + pl;
            f32x4 wv = zero4;
            int pos = 0;
            if (m < M) {
                const int b = m / HW, pix = m - b * HW;
                const int h = pix / W, w = pix - h * W;
                const float* omp = om + (size_t)m * om_stride;
                const float dy = omp[d * 18 + 2 * tap];
                const float dx = omp[d * 18 + 2 * tap + 1];
                const float ml = omp[dg * 18 + d * 9 + tap];
                const float msk = 1.f / (1.f + expf(-ml));
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
                    const int ya = y0ok ? y0 : 0, xa = x0ok ? x0 : 0;
                    const int yb = y1ok ? y0 + 1 : H - 1, xb = x1ok ? x0 + 1 : W - 1;
                    pos = (int)((unsigned)(b * HW + ya * W + xa) | ((unsigned)(xb - xa) << 30) | ((unsigned)(yb - ya) << 31));
                }
            }
            setw[e] = wv;
            setp[e] = pos;
        }
    };

    int ch_tap = 0, ch_c = 0, ch_d = 0;
    f32x4 gw[2];
    f32x4 gv[2][4][2];                                      // [cell][corner][half]: 8 fp32 channels per corner

    auto gather_issue = [&]() {
        if (X6_KO & 8) return;
        const int cb = ch_d * cpg + ch_c * BKC + gq * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = ch_tap * BM + gp + 64 * j;
            gw[j] = setw[e];
            const int pos = setp[e];
            const unsigned i00 = (unsigned)(pos & 0x3FFFFFFF);
            const unsigned dxb = ((unsigned)pos >> 30) & 1u, dyb = (unsigned)pos >> 31;
            const unsigned i10 = i00 + dyb * (unsigned)W;
            const float* p0 = x + (size_t)i00 * (unsigned)C + cb;
            const float* p1 = x + (size_t)(i00 + dxb) * (unsigned)C + cb;
            const float* p2 = x + (size_t)i10 * (unsigned)C + cb;
            const float* p3 = x + (size_t)(i10 + dxb) * (unsigned)C + cb;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                gv[j][0][hh] = *reinterpret_cast<const f32x4*>(p0 + 4 * hh);
                gv[j][1][hh] = *reinterpret_cast<const f32x4*>(p1 + 4 * hh);
                gv[j][2][hh] = *reinterpret_cast<const f32x4*>(p2 + 4 * hh);
                gv[j][3][hh] = *reinterpret_cast<const f32x4*>(p3 + 4 * hh);
            }
        }
    };
    auto gather_finish = [&](int buf) {
        if (X6_KO & 1) return;
        u16* Ad = As + buf * NP * A_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x8 oh, om_, ol;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // the blend of dcn_fused.hip: the same four products, the same order -- element by element: the packed forms
                    // (v_pk_mul_f32 / v_pk_fma_f32) do not run beside the MFMAs, a v_fma_f32 does (scripts/ubench/mfma16_valu_overlap.hip;
                    // the file is built with -fno-slp-vectorize so that the compiler does not pair them up again)
                    const float ve = gv[j][0][hh][e] * gw[j][0] + gv[j][1][hh][e] * gw[j][1] + gv[j][2][hh][e] * gw[j][2] + gv[j][3][hh][e] * gw[j][3];
                    __bf16 h, m, l;
                    split3(ve, h, m, l);
                    oh[4 * hh + e] = h;
                    om_[4 * hh + e] = m;
                    ol[4 * hh + e] = l;
                }
            }
            *reinterpret_cast<bf16x8*>(Ad + a_wr0 + j * 64 * BKC) = oh;
            *reinterpret_cast<bf16x8*>(Ad + A_STAGE + a_wr0 + j * 64 * BKC) = om_;
            *reinterpret_cast<bf16x8*>(Ad + 2 * A_STAGE + a_wr0 + j * 64 * BKC) = ol;
        }
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
    auto advance = [&]() {
        if (++ch_tap == 9) {
            ch_tap = 0;
            if (++ch_c == cpc) {
                ch_c = 0;
                ++ch_d;
            }
        }
    };

#if X6_PIPE
    // ---- the K loop as one instruction stream per chunk (one wave per SIMD: what overlaps must overlap inside the wave) -------------------
    // chunk ch: weight fragments + the first activation row -> registers, barrier, DMA of chunk ch + 1's weight planes, corner loads of chunk
    // ch + 2 (inline assembly: the compiler's wait-count pass would drain the DMA with them); then the MFMAs of chunk ch row by row, the next
    // row's fragments read behind them and a quarter of chunk ch + 1's blend + split + LDS writes dealt out behind every row (its corners
    // were loaded during chunk ch - 1: two register sets).  v_mfma_f32_16x16x32_bf16 hides one 8-cycle or two 4-cycle vector instructions
    // (scripts/ubench/mfma16_valu_overlap.hip); the body is unconditional (past the end the last chunk is loaded again, the writes go to a
    // stage nobody reads) so that it is one basic block.
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    f32x4 gw1[2] = {};
    f32x4 gv1[2][4][2] = {};
    auto issue = [&]() {
        if (X6_KO & 8) return;
        const int cb = ch_d * cpg + ch_c * BKC + gq * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = ch_tap * BM + gp + 64 * j;
            gw1[j] = setw[e];
            const int pos = setp[e];
            const unsigned i00 = (unsigned)(pos & 0x3FFFFFFF);
            const unsigned dxb = ((unsigned)pos >> 30) & 1u, dyb = (unsigned)pos >> 31;
            const unsigned i10 = i00 + dyb * (unsigned)W;
            const float* pc[4] = {x + (size_t)i00 * (unsigned)C + cb, x + (size_t)(i00 + dxb) * (unsigned)C + cb, x + (size_t)i10 * (unsigned)C + cb,
                                  x + (size_t)(i10 + dxb) * (unsigned)C + cb};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gv1[j][k][0] = *reinterpret_cast<const f32x4*>(pc[k]);
                gv1[j][k][1] = *reinterpret_cast<const f32x4*>(pc[k] + 4);
            }
        }
    };
    // quarter `part` (cell j = part >> 1, channel half hh = part & 1) of this thread's 16 column values: blend, split, three 8-byte LDS writes
    auto finish_part = [&](int part, int buf) {
        if (X6_KO & 1) return;
        const int j = part >> 1, hh = part & 1;
        float ve[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            // the blend of dcn_fused.hip: the same four products, the same order (element by element, no packed fp32 instructions)
            ve[e] = gv1[j][0][hh][e] * gw1[j][0] + gv1[j][1][hh][e] * gw1[j][1] + gv1[j][2][hh][e] * gw1[j][2] + gv1[j][3][hh][e] * gw1[j][3];
        unsigned h0, m0, l0, h1, m1, l1;
        split3_pair(ve[0], ve[1], h0, m0, l0);
        split3_pair(ve[2], ve[3], h1, m1, l1);
        const u32x2 oh = {h0, h1}, om_ = {m0, m1}, ol = {l0, l1};
        u16* Ad = As + buf * NP * A_STAGE + a_wr0 + j * 64 * BKC + 4 * hh;
        *reinterpret_cast<u32x2*>(Ad) = oh;
        *reinterpret_cast<u32x2*>(Ad + A_STAGE) = om_;
        *reinterpret_cast<u32x2*>(Ad + 2 * A_STAGE) = ol;
    };
    static_assert(NBS == 1 && MT == 4, "the pipelined loop is written for one weight buffer and four fragment rows");

    setups(0);
    __syncthreads();
    issue();                                       // chunk 0
    b_issue(0, 0);
#pragma unroll
    for (int part = 0; part < 4; ++part) finish_part(part, 0);
    if (nchunks > 1) advance();
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks && ch_tap == 0 && ch_c == 0) {      // chunk ch + 1 opens a deformable group: its sampling table
            setups(ch_d);
            __syncthreads();
        }
        issue();             // corners of chunk ch + 1, first thing in the chunk (past the end: the last chunk again, written to a stage nobody reads)
        __builtin_amdgcn_sched_barrier(0);
        const u16* Ab = As + buf * NP * A_STAGE + wm * WTM * BKC + fo;
        const u16* Bb = Bs + wn * WTN * BKC + fo;
        bf16x8 afr[2][NP], bf[NP][NT];
        auto a_row = [&](int i) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) afr[i & 1][pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_STAGE + i * 16 * BKC);
        };
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(Bb + pl * B_STAGE + j * 16 * BKC);
        a_row(0);
        // one weight buffer: every wave holds its weight fragments in registers before the next chunk's planes may land
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        b_issue(min(ch + 1, nchunks - 1), 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (i + 1 < MT) a_row(i + 1);
            const bf16x8 (&af)[NP] = afr[i & 1];
            // six products per fragment pair, smallest first (a: column planes, b: weight planes)
            if (!(X6_KO & 2))
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][j], af[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2][j], af[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][j], af[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[0], c, 0, 0, 0);
                acc[i][j] = c;
            }
            // the corners were requested in front of row 0: its 48 MFMAs (768 cycles) cover an L2 hit; rows 1 .. 3 carry the four quarters.
            // Every row is its own scheduling region: [the next row's fragment reads] [MFMA, vector instructions]* [the quarter's planes]
#if X6_SPLIT_ROWS == 2
            constexpr int PARTS[4] = {0, 0, 2, 2};
#elif X6_SPLIT_ROWS == 3
            constexpr int PARTS[4] = {0, 1, 1, 2};
#else
            constexpr int PARTS[4] = {1, 1, 1, 1};
#endif
            constexpr int PFIRST[4] = {0, PARTS[0], PARTS[0] + PARTS[1], PARTS[0] + PARTS[1] + PARTS[2]};
#pragma unroll
            for (int q = 0; q < PARTS[i]; ++q) finish_part(PFIRST[i] + q, buf ^ 1);
            if (i + 1 < MT) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
#pragma unroll
            for (int k = 0; k < NT * 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (PARTS[i] == 1) __builtin_amdgcn_sched_group_barrier(0x002, X6_PIPE, 0);
                if (PARTS[i] == 2) __builtin_amdgcn_sched_group_barrier(0x002, 2 * X6_PIPE, 0);
            }
            if (PARTS[i] == 1) __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);
            if (PARTS[i] == 2) __builtin_amdgcn_sched_group_barrier(0x200, 2 * NP, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 2 < nchunks) advance();
        __syncthreads();
    }
#else
    setups(0);
    __syncthreads();
    gather_issue();
    b_issue(0, 0);
    gather_finish(0);
    advance();
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunks;
        if (NBS == 2 && more) {
            if (ch_tap == 0 && ch_c == 0) {
                setups(ch_d);
                __syncthreads();
            }
            gather_issue();
            b_issue(ch + 1, buf ^ 1);
        }
        const u16* Ab = As + buf * NP * A_STAGE + wm * WTM * BKC + fo;
        const u16* Bb = Bs + (NBS == 2 ? buf : 0) * NP * B_STAGE + wn * WTN * BKC + fo;
        bf16x8 af[NP][MT], bf[NP][NT];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < MT; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_STAGE + i * 16 * BKC);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(Bb + pl * B_STAGE + j * 16 * BKC);
        }
        if (NBS == 1 && more) {
            // one weight buffer: every wave holds its fragments in registers before the next chunk's planes may land
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ch_tap == 0 && ch_c == 0) {
                setups(ch_d);
                __syncthreads();
            }
            gather_issue();
            b_issue(ch + 1, 0);
        }
        // six products per fragment pair, smallest first (a: column planes, b: weight planes)
        if (!(X6_KO & 2))
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][j], af[1][i], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2][j], af[0][i], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[2][i], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][j], af[0][i], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[1][i], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[0][i], c, 0, 0, 0);
                acc[i][j] = c;
            }
        if (more) {
            gather_finish(buf ^ 1);
            advance();
        }
        __syncthreads();
    }

#endif

    // ---- epilogue: + bias, 16-byte NHWC fp32 stores (lane: pixel = lane & 15, 8 consecutive channels per tile pair) -----------
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
        const int n0 = nt * BN + wn * WTN + 32 * u + 8 * kq;
        float bv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bv[c] = (bias && n0 + c < Cout) ? bias[n0 + c] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * WTM + i * 16 + r;
            if (m >= M) continue;
            float* dst = out + (size_t)m * Cout + n0;
            if (n0 + 8 <= Cout) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{acc[i][2 * u][0] + bv[0], acc[i][2 * u][1] + bv[1], acc[i][2 * u][2] + bv[2], acc[i][2 * u][3] + bv[3]};
                *reinterpret_cast<f32x4*>(dst + 4) =
                    f32x4{acc[i][2 * u + 1][0] + bv[4], acc[i][2 * u + 1][1] + bv[5], acc[i][2 * u + 1][2] + bv[6], acc[i][2 * u + 1][3] + bv[7]};
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (n0 + c < Cout) dst[c] = acc[i][2 * u + (c >> 2)][c & 3] + bv[c];
            }
        }
    }
}


#if X6_V2
// ---- round 5: the K loop as a software pipeline without an exposed fragment phase, memory instructions dealt out between the MFMAs ------------
// Round 4's loop (below, X6_V2 = 0) held all 24 weight fragments of a chunk in registers because the three weight planes (48 KB) had a single
// LDS buffer: every chunk began with 27 ds_read_b128 per wave + a barrier before its first MFMA, then 12 DMA + 16 corner requests in a row --
// knock-outs: MFMAs alone 1.12 ms, + fragment reads 1.28, + DMA 1.60, everything 2.12.  What the measurements of this round say
// (scripts/ubench/vmem_rates.hip, mfma_rates.hip, scripts/dcn_x6_knockout.sh):
//   * a CU gets 52 B / clk of contiguous 1-KiB pieces and ~40 B / clk of 128-byte corner segments out of L2, as plain loads or LDS DMA alike:
//     the 112 KB a chunk needs are ~2 600 cycles of the vector memory path beside 3 072 cycles of MFMAs -- requests issued in a burst stall the
//     wave (and its matrix pipe) while the queue drains, so every request sits behind its own group of six MFMAs here;
//   * the DMA's cost is issue, not latency (the same eight chunks over and over: no change);
//   * the channel-block-fastest chunk order (one tap's table instead of nine: room for everything) costs the corner loads their L2 hits --
//     the next tap revisits a pixel's line 8 chunks = 4 MB of traffic later -- so the order stays tap-fastest, and LDS is found elsewhere:
// 24 KB ONE activation stage (the four rows' fragments are reloaded IN PLACE, behind the last column tile's MFMAs, from the stage the previous
// iteration wrote; a second barrier in mid-iteration lets the next chunk's planes overwrite it) + 96 KB weight planes in four half buffers +
// 23 KB sampling table of a group = 143 KB.  The loop is rotated by half a chunk: iteration `it` runs the column tiles j = 4..7 of chunk it - 1
// (part A), then j = 0..3 of chunk it (part B).  What an iteration reads (Y half of chunk it - 1, X half of chunk it) was DMA'd during the
// previous iteration; what it DMAs (Y of chunk it, X of chunk it + 1) goes to the halves the previous iteration read.  Column tile outer,
// fragment row inner: the weight fragments of one tile are prefetched behind the previous tile's MFMAs (2 x 12 VGPRs instead of 96).
// Part A carries the 16 corner requests of chunk it + 1 (inline assembly, counted waits) and the 12 DMA pieces, one per group of six MFMAs;
// part B carries the blend + split + plane writes of chunk it + 1, one pair of values per group.
namespace v2 {
#ifdef X6_TIMING
// debug build (scripts/dcn_x6_timing.sh): wave 0 of every workgroup accumulates the 100-MHz real-time ticks between its phase boundaries
__device__ unsigned long long g_x6_timing[8];
#define X6_T(k)                                                   \
    if (X6_TIMING_ON) {                                           \
        const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); \
        t_acc[k] += t_now - t_last;                               \
        t_last = t_now;                                           \
    }
#else
#define X6_T(k)
#endif
// eight waves (two per SIMD: with one, every scalar / vector / LDS instruction of the wave takes one of the 768 issue slots the 192 MFMAs of a
// chunk leave it -- the four-wave form of this loop ran 1.96 ms, every knock-out paid) on 64 x 64 wave tiles: 2 (rows) x 4 (columns)
constexpr int THREADS = 512, NWAVES = THREADS / 64;
constexpr int WTN = 64, NT = WTN / 16;                // shadows the four-wave constants of the file
constexpr int HB_ROWS = BN / 2;                       // rows of a half buffer: (wave column, j & 1, r)
constexpr int HB_PLANE = HB_ROWS * BKC;               // u16 elements per plane of a half
constexpr int HB_ELEMS = NP * HB_PLANE;               // 24 KB
constexpr int NTH = NT / 2;                           // column tiles per half
constexpr int NG = NTH * MT;                          // groups of six MFMAs per part
constexpr int TAB_N = 9 * BM;
constexpr int LDS_BYTES = (NP * A_STAGE + 4 * HB_ELEMS) * 2 + TAB_N * 16 + TAB_N * 4;
constexpr int TPT = (TAB_N + THREADS - 1) / THREADS;  // table entries per thread
constexpr int DPW = 24 / NWAVES;                      // DMA pieces per wave and half
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
static_assert(MT == 4 && NT == 4 && BN == 256 && NG == 8, "written for 128 x 256 tiles on eight waves");
// vmcnt is counted in issue order: the thread's 8 corner requests have landed when at most X6_W younger operations of the wave are
// outstanding.  Issue order of an iteration's part A (groups 0..7): two requests behind each of the groups 0..3, one DMA piece behind each of
// the groups 2..7 -> younger than the last request: D1..D5.
#define X6_W 5
#define X6_STR2(x) #x
#define X6_STR(x) X6_STR2(x)

__global__ __launch_bounds__(THREADS, 1) void dcn_x6_v2_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                          const u16* __restrict__ wp, const float* __restrict__ bias,
                                                          float* __restrict__ out, int M, int H, int W, int C, int dg, int om_stride,
                                                          int Cout, int ntn, int mtiles, long long plane_elems) {
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    u16* const As = smem_h;                                   // [3][BM][32]
    u16* const Bh = smem_h + NP * A_STAGE;                    // [X | Y][2][3][BN / 2][32]
    f32x4* const tabw = reinterpret_cast<f32x4*>(smem_h + NP * A_STAGE + 4 * HB_ELEMS);      // [9][BM] corner weights (x mask)
    int* const tabp = reinterpret_cast<int*>(tabw + TAB_N);                                   // [9][BM] corner position + step flags
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, kq = lane >> 4;
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
    const int m0 = mt * BM;
    const int HW = H * W, cpg = C / dg, cpc = cpg / BKC;
    const int nchunks = dg * cpc * 9;
    const u16* wslab = wp + (size_t)nt * nchunks * B_STAGE;      // plane 0; plane p at + p * plane_elems

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

#ifdef X6_TIMING
#ifndef X6_TWAVE
#define X6_TWAVE 0
#endif
    const bool X6_TIMING_ON = wave == X6_TWAVE;
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memrealtime();
#endif
    // gather roles: thread -> (pixel row gp, 8-channel slot gq)
    const int gq = tid & 3, gp = tid >> 2;
    const int a_wr0 = gp * BKC + ((gq ^ swz(gp)) << 3);
    const int fo = r * BKC + ((kq ^ swz(r)) << 3);

    // ---- sampling table of one deformable group (9 taps x BM rows): the om values are requested at the top of an iteration, the entries are
    // made and written behind its second barrier (nobody reads the old table any more), the next iteration's corner requests read them ----------
    float t_dy[TPT], t_dx[TPT], t_ml[TPT];
    auto tab_load = [&](int d) {
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int e = tid + THREADS * u;
            const int tap = e / BM, m = m0 + (e - tap * BM);
            t_dy[u] = t_dx[u] = t_ml[u] = 0.f;
            if (e < TAB_N && m < M) {
                const float* omp = om + (size_t)m * om_stride;
                t_dy[u] = omp[d * 18 + 2 * tap];
                t_dx[u] = omp[d * 18 + 2 * tap + 1];
                t_ml[u] = omp[dg * 18 + d * 9 + tap];
            }
        }
    };
    auto tab_finish = [&]() {                            // the arithmetic of dcn_fused.hip
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int e = tid + THREADS * u;
            if (e >= TAB_N) continue;
            const int tap = e / BM, m = m0 + (e - tap * BM);
            f32x4 wv = zero4;
            int pos = 0;
            if (m < M) {
                const int b = m / HW, pix = m - b * HW;
                const int h = pix / W, w = pix - h * W;
                const float msk = 1.f / (1.f + expf(-t_ml[u]));
                const float py = (float)(h - 1 + tap / 3) + t_dy[u];
                const float px = (float)(w - 1 + tap % 3) + t_dx[u];
                if (py > -1.f && px > -1.f && py < (float)H && px < (float)W) {
                    const float y0f = floorf(py), x0f = floorf(px);
                    const int y0 = (int)y0f, x0 = (int)x0f;
                    const float ly = py - y0f, lx = px - x0f, hy = 1.f - ly, hx = 1.f - lx;
                    const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
                    wv[0] = (y0ok && x0ok) ? hy * hx * msk : 0.f;
                    wv[1] = (y0ok && x1ok) ? hy * lx * msk : 0.f;
                    wv[2] = (y1ok && x0ok) ? ly * hx * msk : 0.f;
                    wv[3] = (y1ok && x1ok) ? ly * lx * msk : 0.f;
                    const int ya = y0ok ? y0 : 0, xa = x0ok ? x0 : 0;
                    const int yb = y1ok ? y0 + 1 : H - 1, xb = x1ok ? x0 + 1 : W - 1;
                    pos = (int)((unsigned)(b * HW + ya * W + xa) | ((unsigned)(xb - xa) << 30) | ((unsigned)(yb - ya) << 31));
                }
            }
            tabw[e] = wv;
            tabp[e] = pos;
        }
    };

    // ---- corner requests of one chunk: 4 corners x 2 halves of 4 fp32 channels per thread, by inline assembly ------------------------------------
    f32x4 gw = zero4;
    f32x4 gv[4][2];
    const float* pc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) gv[k][0] = gv[k][1] = zero4, pc[k] = x;
    int ld_tap = 0, ld_cc = 0, ld_d = 0;                    // the chunk the next corner requests are for
    auto corner_addr = [&]() {                               // weights + the four corner addresses from the table
        const int cb = ld_d * cpg + ld_cc * BKC + gq * 8;
        const int e = ld_tap * BM + gp;
        gw = tabw[e];
        int pos = tabp[e];
        if (X6_KO & 128) pos = (pos & 0xC0000000) | gp;       // experiment: every tile reads the same 128 pixels (cache hits)
        if (X6_KO & 256) pos = (pos & 0xC0000000) | min(m0 + gp, M - 2);      // experiment: no offsets (own pixel, dense lines)
        const unsigned i00 = (unsigned)(pos & 0x3FFFFFFF);
        const unsigned dxb = ((unsigned)pos >> 30) & 1u, dyb = (unsigned)pos >> 31;
        const unsigned i10 = i00 + dyb * (unsigned)W;
        pc[0] = x + (size_t)i00 * (unsigned)C + cb;
        pc[1] = x + (size_t)(i00 + dxb) * (unsigned)C + cb;
        pc[2] = x + (size_t)i10 * (unsigned)C + cb;
        pc[3] = x + (size_t)(i10 + dxb) * (unsigned)C + cb;
    };
    auto corner_req = [&](int q) {                           // request q = (corner, half): 16 bytes per lane
        if (X6_KO & 8) return;
        const int k = q >> 1;
        if (q & 1) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(gv[k][1]) : "v"(pc[k]) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv[k][0]) : "v"(pc[k]) : "memory");
    };
    // the four requests of channel half HH (or all eight: X6_CELL_WAIT) have landed; N younger requests of this wave may still be in flight
#define X6_HALF_WAIT(HH, N) \
    if (!(X6_KO & 8)) asm volatile("s_waitcnt vmcnt(" X6_STR(N) ")" : "+v"(gv[0][HH]), "+v"(gv[1][HH]), "+v"(gv[2][HH]), "+v"(gv[3][HH]))
#define X6_CELL_WAIT(N)    \
    X6_HALF_WAIT(0, N);    \
    X6_HALF_WAIT(1, N)
    auto advance_ld = [&]() {
        if (++ld_tap == 9) {
            ld_tap = 0;
            if (++ld_cc == cpc) {
                ld_cc = 0;
                ++ld_d;
            }
        }
    };
    // half `hh` (channels 4 hh .. 4 hh + 3) of this thread's 8 column values: blend, split, one 8-byte write per plane
    auto blend_half = [&](int hh) {
        if (X6_KO & 1) return;
        float ve[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            // the blend of dcn_fused.hip: the same four products, the same order (element by element, no packed fp32 instructions)
            ve[e] = gv[0][hh][e] * gw[0] + gv[1][hh][e] * gw[1] + gv[2][hh][e] * gw[2] + gv[3][hh][e] * gw[3];
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pair(ve[0], ve[1], h0, m0_, l0);
        split3_pair(ve[2], ve[3], h1, m1, l1);
        u16* Ad = As + a_wr0 + 4 * hh;
        *reinterpret_cast<u32x2*>(Ad) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(Ad + A_STAGE) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(Ad + 2 * A_STAGE) = u32x2{l0, l1};
    };
    // weight planes of (chunk, half) -> half buffer (half, parity): 24 1-KiB pieces (plane, wave column, j & 1), three per wave
    auto dma_piece = [&](int chunk, int half, int parity, int q) {
        if (X6_KO & 4) return;
        if (X6_KO & 64) chunk &= 7;                           // experiment: always the same eight chunks
        u16* dst = Bh + (half * 2 + parity) * HB_ELEMS;
        const u16* src = wslab + (size_t)chunk * B_STAGE + lane * 8;
        const int p = q * NWAVES + wave;                     // piece 0..23
        const int pl = p >> 3, g8 = p & 7;
        const int G = (g8 >> 1) * NT + half * NTH + (g8 & 1);      // 16-row group of the plane's [BN][32] tile
        dma16(src + (size_t)pl * plane_elems + G * 512, dst + pl * HB_PLANE + g8 * 512);
    };

    bf16x8 areg[MT][NP], breg[2][NP];
    auto a_load_row = [&](int i) {
        const u16* Ab = As + (wm * WTM + i * 16) * BKC + fo;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if (X6_KO & 16) asm volatile("" : "=v"(areg[i][pl]));
            else areg[i][pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_STAGE);
        }
    };
    auto b_load = [&](int which, int half, int parity, int jj) {
        const u16* Bb = Bh + (half * 2 + parity) * HB_ELEMS + (wn * NTH * 16 + jj * 16) * BKC + fo;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if (X6_KO & 16) asm volatile("" : "=v"(breg[which][pl]));
            else breg[which][pl] = *reinterpret_cast<const bf16x8*>(Bb + pl * HB_PLANE);
        }
    };
    // the six products of one fragment pair, smallest first (first operand: weight planes, second: column planes)
    auto mma_row = [&](int i, int j, int which) {
        if (X6_KO & 2) return;
        f32x4 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][1], areg[i][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][2], areg[i][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][1], areg[i][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[which][0], areg[i][0], c, 0, 0, 0);
        acc[i][j] = c;
    };

    // ---- prologue: table of group 0, corners + planes of chunk 0, X half of chunk 0's weights ---------------------------------------------------
    tab_load(0);
    tab_finish();
    __syncthreads();
    corner_addr();
#pragma unroll
    for (int q = 0; q < 8; ++q) corner_req(q);
#pragma unroll
    for (int q = 0; q < DPW; ++q) dma_piece(0, 0, 0, q);
    X6_CELL_WAIT(3);
    blend_half(0);
    blend_half(1);
    advance_ld();                                            // nchunks >= 9
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    int tb_next = cpc * 9, tb_d = 1;                         // first chunk of the next group, and the group

    // ---- iteration `it`: [part A: j = 2, 3 of chunk it - 1] [part B: j = 0, 1 of chunk it]; FIRST has no chunk behind it, LAST none in front ----
    // Two barriers, both with slack.  B1 (behind part A's first column tile): the previous iteration's planes and X pieces are in LDS, its
    // table is read -> the stage / the X half may be read, the X half the previous part B used may be overwritten, the table rewritten.
    // B2 (behind part B's first group): every wave holds chunk it's activation fragments and its Y pieces have landed -> the stage may take
    // chunk it + 1's planes, the Y half may be read (its first fragments are prefetched behind part B's last group).
    // ONE memory request per group of six MFMAs: the vector memory path takes one wave-wide 16-byte request per 16 cycles, so eight waves can
    // issue one each per 128 cycles; a denser stretch stalls the waves IN ORDER in front of their next MFMAs (the timing build showed the
    // first column tile of part A, carrying 11 of the 14 requests, at 1.6 x its MFMA time and a 550-ns wait at B1 behind it).
    // Slots (A0..A7, B0..B7): D0 D1 D2 (Y pieces of chunk it) | L0 | B1 | L1 L2 L3 L4 | L5 | B2 | L6 L7 | D3 D4 D5 (X pieces of chunk it + 1);
    // L0..L3 = the first channel half of the four corners, L4..L7 the second.  vmcnt counts in issue order:
    //   B1: vmcnt(4) = the previous iteration's D5 has landed;   B2: vmcnt(6) = D0..D2 have landed;
    //   first half of the column values (B3, in front of D3): vmcnt(4) = L0..L3;   second half (B6): vmcnt(3) = L4..L7.
    constexpr int NDH = (X6_KO & 4) ? 0 : DPW, NLQ = (X6_KO & 8) ? 0 : 1;
    constexpr int K_B1 = NDH + NLQ, K_B2 = 6 * NLQ;
    auto vm_slot = [&](int sl, int cy, int cx, int par) {    // the memory request of slot sl = 0..15 (A0..A7, B0..B7)
        auto dma_q = [&](int q) { dma_piece(q < DPW ? cy : cx, q < DPW ? 1 : 0, q < DPW ? par : par ^ 1, q % DPW); };
        if (sl < 3) dma_q(sl);
        else if (sl < 11) {
            const int l = sl - 3;                            // L0..L7: corner l & 3, channel half l >> 2
            corner_req(2 * (l & 3) + (l >> 2));
        } else if (sl < 14) dma_q(sl - 8);
    };
    auto iteration = [&](int it, auto first_c, auto last_c) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        const int par = it & 1;
        const int cy = min(it, nchunks - 1), cx = min(it + 1, nchunks - 1);
        bool make_tab = false;
        if (!LAST) {
            make_tab = it + 2 == tb_next && it + 2 < nchunks;
            if (make_tab) tab_load(tb_d);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- part A ----
        if (!FIRST) {
#pragma unroll
            for (int jj = 0; jj < NTH; ++jj) {
                if (jj + 1 < NTH) b_load((jj + 1) & 1, 1, par ^ 1, jj + 1);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int g = jj * MT + i;               // group 0..7: six MFMAs
                    if (!LAST && g == 3) corner_addr();
                    if (jj + 1 == NTH && !LAST && i == 0) b_load((jj + 1) & 1, 0, par, 0);          // X of chunk it (behind B1)
                    mma_row(i, NTH + jj, jj & 1);
                    if (!LAST) {
                        vm_slot(g, cy, cx, par);
                        if (jj + 1 == NTH) a_load_row(i);    // chunk it's planes, in place behind the row's last use
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (jj == 0 && !LAST) {
                    X6_T(0)
                    if (X6_KO & 32) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K_B1) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(K_B1) : "memory");          // B1
                    X6_T(1)
                    if (make_tab) {
                        tab_finish();
                        tb_next += cpc * 9;
                        ++tb_d;
                    }
                }
            }
        } else {
            corner_addr();
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) vm_slot(sl, cy, cx, par);
#pragma unroll
            for (int i = 0; i < MT; ++i) a_load_row(i);
            b_load(0, 0, par, 0);
        }
        if (!LAST) {
            // ---- part B ----
#pragma unroll
            for (int jj = 0; jj < NTH; ++jj) {
                if (jj + 1 < NTH) b_load((jj + 1) & 1, 0, par, jj + 1);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int g = jj * MT + i;               // group 0..7: six MFMAs; the two halves of the thread's column values ride on 3..4 and 6..7
                    if (g == 3) X6_HALF_WAIT(0, 4);
                    if (g == 6) X6_HALF_WAIT(1, 3);
                    mma_row(i, jj, jj & 1);
                    vm_slot(8 + g, cy, cx, par);
                    if (g == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        X6_T(2)
                        if (X6_KO & 32) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K_B2) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(K_B2) : "memory");          // B2
                        X6_T(4)
                    }
                    if (g == 3) blend_half(0);
                    if (g == 6) blend_half(1);
                    if (g == NG - 1) b_load(0, 1, par, 0);   // Y of chunk it for the next iteration's first column tile
                    if (g != 3 && g != 6) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (it + 2 < nchunks) advance_ld();
            X6_T(5)
        }
    };
    X6_T(6)
    iteration(0, std::true_type{}, std::false_type{});
    for (int it = 1; it < nchunks; ++it) iteration(it, std::false_type{}, std::false_type{});
    iteration(nchunks, std::false_type{}, std::true_type{});

    X6_T(0)
    // ---- epilogue: + bias, 16-byte NHWC fp32 stores (lane: pixel = lane & 15, 8 consecutive channels per tile pair) -----------
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
        const int n0 = nt * BN + wn * WTN + 32 * u + 8 * kq;
        float bv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bv[c] = (bias && n0 + c < Cout) ? bias[n0 + c] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * WTM + i * 16 + r;
            if (m >= M) continue;
            float* dst = out + (size_t)m * Cout + n0;
            if (n0 + 8 <= Cout) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{acc[i][2 * u][0] + bv[0], acc[i][2 * u][1] + bv[1], acc[i][2 * u][2] + bv[2], acc[i][2 * u][3] + bv[3]};
                *reinterpret_cast<f32x4*>(dst + 4) =
                    f32x4{acc[i][2 * u + 1][0] + bv[4], acc[i][2 * u + 1][1] + bv[5], acc[i][2 * u + 1][2] + bv[6], acc[i][2 * u + 1][3] + bv[7]};
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (n0 + c < Cout) dst[c] = acc[i][2 * u + (c >> 2)][c & 3] + bv[c];
            }
        }
    }
#ifdef X6_TIMING
    X6_T(7)
    if (tid == X6_TWAVE * 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&g_x6_timing[k], t_acc[k]);
    }
#endif
}
}  // namespace v2
#endif

// OIHW fp32 [Cout][C][3][3] -> three bf16 planes, each [n_tiles][chunks][BN staging rows][32] with the slot swizzle; rows beyond Cout zero
__global__ void dcn_pack_weight_x6_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int C, int dg, long long total) {
    const int cpg = C / dg, cpc = cpg / BKC, nchunks = dg * cpc * 9;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int slot = (int)((i >> 3) & 3);
        const int row = (int)((i >> 5) % BN);
        const long long t = (i >> 5) / BN;
        const int chunk = (int)(t % nchunks);
        const int nt = (int)(t / nchunks);
        const int q = slot ^ swz(row);
        const int tap = chunk % 9, cc = chunk / 9;                                // chunk = ((group * cpc + channel block) * 9 + tap)
        const int c = cc * BKC + q * 8 + e;
        const int n = nt * BN + chan_of_row(row);
        __bf16 h, m, l;
        split3(n < Cout ? w[((size_t)n * C + c) * 9 + tap] : 0.f, h, m, l);
        wp[i] = __builtin_bit_cast(u16, h);
        wp[i + total] = __builtin_bit_cast(u16, m);
        wp[i + 2 * total] = __builtin_bit_cast(u16, l);
    }
}

}  // namespace

#ifdef X6_TIMING
extern "C" int gssd_dcn_x6_timing_read(unsigned long long* out8) {       // debug build only: read and clear
    unsigned long long z[8] = {};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(v2::g_x6_timing), sizeof(z)) != hipSuccess) return GSSD_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(v2::g_x6_timing), z, sizeof(z)) != hipSuccess) return GSSD_ELAUNCH;
    return GSSD_OK;
}
#endif

extern "C" long long gssd_dcn_packed_weight_elems_x6(int Cout, int C) {          // bf16 elements (three planes)
    if (Cout <= 0 || C <= 0 || C % BKC != 0) return -1;
    return 3ll * ((Cout + BN - 1) / BN) * BN * 9 * C;
}

extern "C" int gssd_dcn_pack_weight_x6(const float* w_oihw, void* w_packed, int Cout, int C, int dg, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_oihw && w_packed && Cout > 0 && C > 0 && dg > 0 && C % dg == 0 && (C / dg) % BKC == 0);
    const long long total = gssd_dcn_packed_weight_elems_x6(Cout, C) / 3;
    hipLaunchKernelGGL(dcn_pack_weight_x6_kernel, dim3((int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_oihw, reinterpret_cast<u16*>(w_packed), Cout, C, dg, total);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_forward_x6(const float* x, const float* om, const void* w_packed, const float* bias, float* out, int B, int H,
                                   int W, int C, int dg, int om_stride, int Cout, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && w_packed && out && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0 && Cout > 0 && Cout % 8 == 0);
    GSSD_CHECK_ARG(C % dg == 0 && (C / dg) % BKC == 0 && om_stride >= 27 * dg);
    GSSD_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)out % 16) == 0);
    const long long Mll = (long long)B * H * W;
    GSSD_CHECK_ARG(Mll < (1ll << 30) && Mll * C < (1ll << 32));          // 30-bit pixel index + 2 flag bits; 32-bit element offsets
    const int M = (int)Mll;
    const int ntn = (Cout + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    static unsigned attr_mask = 0;
#if X6_V2
    const auto kernel = v2::dcn_x6_v2_kernel;
    constexpr int LDS_BYTES = v2::LDS_BYTES;
#else
    const auto kernel = dcn_x6_kernel;
#endif
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
            hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", LDS_BYTES);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask);
    }
    int blocks;
    if (8 % ntn == 0) {
        const int per = 8 / ntn;
        blocks = ((mtiles + per - 1) / per) * 8;
    } else {
        blocks = ((mtiles * ntn + 7) / 8) * 8;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(X6_V2 ? 512 : 256), LDS_BYTES, as_stream(stream), x, om, reinterpret_cast<const u16*>(w_packed), bias,
                       out, M, H, W, C, dg, om_stride, Cout, ntn, mtiles, gssd_dcn_packed_weight_elems_x6(Cout, C) / 3);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
