// Winograd F(2x2, 3x3) on the BF16 matrix cores with fp32-equivalent products (round 5): the grouped 3x3 / stride 1 / pad 1 trunk
// convolutions conv2_1 .. conv5_3 (and their data gradients) of models/ssd_multiphase_custom_group.py:434-460 (vgg()).
//
// conv_wino.hip runs the 16 Winograd-domain GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[co][ci]  on v_mfma_f32_16x16x4_f32 -- 1/16 of
// the bf16 matrix rate of gfx950.  Here both operands are the exact sum of three bf16 planes (x = h + m + l, conv_x6.hip) and a product is six
// v_mfma_f32_16x16x32_bf16 (every term above 2^-24), fp32 accumulation: 6 / 16 of the fp32 instruction's matrix-pipe time per product, on top
// of Winograd's 2.25 x fewer products.  The transforms stay fp32 and are the ones of conv_wino.hip (V = B^T d B straight from global memory,
// Y = A^T M A in registers), so the result differs from that kernel only by the products' last bit.
//
// What shapes the kernel: with one wave per SIMD a wave issues a vector instruction every ~8 cycles, and the transform + three-plane split cost
// ~2000 of them per 32-channel chunk of 16 tiles against 384 MFMAs (6.1 k cycles) -- a single instruction stream that does both runs at 27 k
// cycles per chunk whatever the order (measured: scripts/wino_x6_timing.py, profiles/r05_e_*).  Hence TWO ROLES, one wave of each per SIMD:
//   * a workgroup = 8 waves owns 64 consecutive tiles of the linearised (image, tile_y, tile_x) list and NB = 64 (or 32) output channels of one
//     group, persistent over a contiguous range of such items.  Waves 0-3 are PRODUCERS (tile group w: 16 tiles): patch loads, producer BatchNorm
//     + ReLU, the Winograd input transform, the split, operand planes -> LDS in MFMA-operand order (lane-linear: 16 bytes per lane, xi and plane).
//     Waves 4-7 are CONSUMERS (tile group w - 4): U planes by LDS-DMA, fragment reads, the MFMAs, the output transform and the epilogue.
//     Waves w and w + 4 share a SIMD: its matrix pipe and its vector issue slots are used by different instruction streams;
//   * K runs in chunks of 32 input channels: lane (tile r, quad kq) owns channels 4 kq .. 4 kq + 3 of both 16-channel halves of the chunk (one
//     16-byte load per patch position and half, conv_wino.hip's layout); the halves fill slots 0-3 / 4-7 of the lane's bf16x8 MFMA operand and U is
//     packed in the same slot order (the MFMA only needs both operands to agree on which channel sits in which k slot);
//   * a chunk runs as four BLOCKS, one per row i of the 4 x 4 Winograd domain, in the order 0, 3, 1, 2: block i needs t_i. = (B^T d)_i. (two patch
//     rows, loaded ONCE per chunk and kept in the producer's registers), V_i. = t_i. B (four xi) and their planes; the consumer folds the block's
//     products into the 2 x 2 outputs right away, Y[a][b] += A[i][a] (M_i. A)[b] (Y = A^T M A is linear in M), so it keeps 16 NBT output
//     registers instead of 64 NBT accumulators;
//   * per block two barriers: A -- the block's planes are in LDS and its U planes have landed; B -- the consumers hold the planes in registers,
//     the producers may write the next block's.  U stages (4 xi x 3 planes x NB x 32 bf16) are double buffered.
#include "common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

#ifndef WX6_KO
#define WX6_KO 0              // knock-outs (scripts/wino_x6_knockout.sh; results wrong by construction): 1 no vector work (transform / split), 2 no MFMAs,
#endif                        // 4 no U DMA, 8 no patch loads, 16 no fragment reads
#ifndef WX6_LOCAL_SUM
#define WX6_LOCAL_SUM 1       // the six products of a chunk are summed from zero and added to the running sum by the vector ALU (the bf16 MFMA's
#endif                        // adder truncates: conv_x6.hip)

namespace {

constexpr int NP = 3, XG = 4;

__device__ __host__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }       // 64-byte rows: conflict-free ds_read_b128 (conv_x6.hip)

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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the split of TWO values at once, planes as packed bf16 pairs (a in the low half): one v_cvt_pk_bf16_f32 per plane and pair (the element-wise
// form costs one per element: the compiler does not pair the converts of two chains), and the packed result IS the operand dword
__device__ __forceinline__ void split3_pair(const float a, const float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
    const float sa = ra - __builtin_bit_cast(float, pm << 16), sb = rb - __builtin_bit_cast(float, pm & 0xffff0000u);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sa, sb}, bf16x2));
}

// The two-plane fp16 form (round 6, conv_thin_x6.hip): x = h + l' / 2048, h = fp16(x), l' = fp16((x - h) * 2048) -- |x - h - l' / 2048| <= 2^-24 |x| --
// and three v_mfma_f32_16x16x32_f16 per product (h h'; h l' + l' h' into a sum that enters with 1 / 2048).  For launches whose input is a BatchNorm +
// ReLU output (the fused producer transform: the trunk's forward convs) or that the caller marks GSSD_CONV_F16_OK (bounded activations: the DCN
// offset conv); data gradients keep the bf16 planes (fp16 has no exponent range for them).  GSSD_X6_F16=0: bf16 planes everywhere.
__device__ __forceinline__ void split2_pair(const float a, const float b, unsigned& ph, unsigned& pl) {
    const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
    const f32x2 r = (f32x2{a, b} - __builtin_convertvector(h, f32x2)) * 2048.f;
    ph = __builtin_bit_cast(unsigned, h);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

struct WinoX6Params {
    const float* in;
    const u16* Ux;           // [groups][cout blocks][chunks][16 xi][3 planes][NB co][32 slots], slot groups swizzled (swz)
    const float* bias;
    float* out;
    const float* resid;
    const float* in_scale;
    const float* in_shift;
    const float* in_pad;
    const float* pool_sign;
    double* stats;
    int stats_rep;
    int B, H, W, in_stride, in_ch_off, Cout, cin_g, cout_g, out_stride, out_ch_off;
    int tiles_y, tiles_x, ntiles;
    int ncb, nchunks, npairs, gx;
    int vec_ok;
    unsigned pad_off;
    // GSSD_OUT_HEADS (round 6: the 38 x 38 multibox head): channels [0, split_n) -> out, the rest -> out_b, each at its source's offset inside an image's priors
    float* out_b;
    int heads, split_n;
    long long out_bs, outb_bs, out_off, outb_off;
};

struct Step {                // a step = one 32-channel chunk of one item + the item's patch origin / validity mask of this lane's tile
    int item, c;
    unsigned pix, valid;
};

// PSEL: out-of-image patch positions are replaced by the padding value with a select (else: the loads already fetched it, WinoX6Params::pad_off)
template <int NBT, bool XF, int EPI, bool PSEL, bool F16>      // NB = 16 NBT output channels per workgroup; EPI: 0 plain, 1 + residual, 2 pooled raw map (GSSD_CONV_POOL2)
__global__ __launch_bounds__(512, 1) void conv_wino_x6_kernel(const WinoX6Params p) {
    constexpr int NPX = F16 ? 2 : 3;                                 // operand planes of this instance
    constexpr int NB = 16 * NBT, TILE = NB * 32, STAGE = XG * NPX * TILE, CHUNK = 16 * NPX * TILE;
    constexpr int PWAVE = XG * NPX * 512;                             // u16 elements of one tile group's planes of a block: [xi][plane][lane][8]
    extern __shared__ __attribute__((aligned(16))) u16 smem[];      // U [2][STAGE] | P [4 tile groups][PWAVE]
    u16* const Pl = smem + 2 * STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;
    const int wv = wave & 3;                                         // tile group
    const int r = lane & 15, kq = lane >> 4;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int nitems = (p.ntiles + 63) >> 6;
    // flat grid: workgroup id = pair + npairs * x.  Ids go round-robin over the 8 XCDs (each with its own L2): the workgroups that stream the
    // U planes of one (group, output block) pair share an XCD (npairs a multiple of 8) or two
    const int pair = blockIdx.x % p.npairs, bx = blockIdx.x / p.npairs;
    const int g = pair / p.ncb, cb = pair - g * p.ncb;
    const int n0 = cb * NB;
    const int item_begin = (int)(((long long)bx * nitems) / p.gx);
    const int item_end = (int)(((long long)(bx + 1) * nitems) / p.gx);
    if (item_begin >= item_end) return;

    const int cb_ld = p.in_ch_off + g * p.cin_g + kq * 4;            // + chunk * 32 + half * 16
    const u16* Ug = p.Ux + (size_t)pair * p.nchunks * CHUNK + lane * 8;
    const int nchunks = p.nchunks;
    const bool tailB = (p.cin_g & 31) != 0;                           // the last chunk has one 16-channel half only
    const int nsteps = (item_end - item_begin) * nchunks;            // a step = one 32-channel chunk of one item = four blocks
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto decode = [&](int it, unsigned& pix_off, unsigned& valid) {
        const int t = (it * 4 + wv) * 16 + r;
        valid = 0;
        int pix0 = 0;
        if (t < p.ntiles) {
            const int b = t / tiles_per_img, rem = t - b * tiles_per_img;
            const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
            const int y = 2 * ty - 1, x = 2 * tx - 1;
            pix0 = (b * p.H + y) * p.W + x;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((unsigned)(y + i) < (unsigned)p.H && (unsigned)(x + j) < (unsigned)p.W) valid |= 1u << (i * 4 + j);
        }
        pix_off = (unsigned)((pix0 * p.in_stride + cb_ld) * 4);        // may wrap for border tiles: only used where valid
    };
    auto advance = [&](Step& s) {                // the step after s; past the end: s again
        Step n = s;
        if (++n.c == nchunks) {
            n.c = 0;
            ++n.item;
            if (n.item >= item_end) return;      // (s stays the last step)
            decode(n.item, n.pix, n.valid);
        }
        s = n;
    };

    if (producer) {
        // =============================== producer: patch rows -> activated -> Winograd row -> three planes -> LDS ===============================
    // one patch row (4 positions, both 16-channel halves) of a step: row[half][position].  Loads, activation and the plane construction are dealt
    // out in QUARTERS (quarter q: half q >> 1; loads: positions 2 (q & 1), + 1; vector work: channels e = 2 (q & 1), + 1 of the lane's quad) so
    // that a block can issue its memory instructions a few at a time between its MFMA groups: a burst of 12 DMA pieces + 16 loads in front of a
    // block cost the wave ~80 cycles of issue time each (6.5 k cycles per step by scripts/wino_x6_timing.py)
    typedef f32x4 Row[2][4];
    auto load_row_q = [&](Row& row, const Step& s, const int i, const int q) {
        const bool hasB = !(tailB && s.c == nchunks - 1);
        const int hf = q >> 1;
        const int ch16 = 2 * s.c + (hf && hasB ? 1 : 0);          // (a missing half re-reads the first one: its result is masked to zero)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * (q & 1) + jj;
            const int pos = i * 4 + j;
            const unsigned off = (s.valid >> pos) & 1 ? s.pix + (unsigned)(((i * p.W + j) * p.in_stride + ch16 * 16) * 4)
                                                      : (PSEL ? 0u : p.pad_off + (unsigned)((cb_ld + ch16 * 16) * 4));
            if (WX6_KO & 8) continue;
            row[hf][j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in) + off);
        }
    };
    auto load_row = [&](Row& row, const Step& s, const int i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) load_row_q(row, s, i, q);
    };
    // producer BatchNorm scale / shift / padding value of a step's two 16-channel halves, in registers (loaded a step ahead: a load in front of its
    // first use costs the single wave of a SIMD the whole L2 latency)
    struct XfTab {
        f32x4 sc[2], sh[2], pd[2];
    };
    auto load_xf = [&](XfTab& t, const Step& s) {
        const bool hasB = !(tailB && s.c == nchunks - 1);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int ch16 = 2 * s.c + (hf && hasB ? 1 : 0);
            if (XF) {
                t.sc[hf] = *reinterpret_cast<const f32x4*>(p.in_scale + cb_ld + ch16 * 16);
                t.sh[hf] = *reinterpret_cast<const f32x4*>(p.in_shift + cb_ld + ch16 * 16);
                t.pd[hf] = *reinterpret_cast<const f32x4*>(p.in_pad + cb_ld + ch16 * 16);
            } else {
                t.sc[hf] = f32x4{1.f, 1.f, 1.f, 1.f};
                t.sh[hf] = t.pd[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    // padding + producer BatchNorm + ReLU of a row, in place, once (a row serves two passes): quarter q
    auto activate_row_q = [&](Row& row, const Step& s, const XfTab& xt, const int i, const int q) {
        if (WX6_KO & 1) return;
        const bool hasB = !(tailB && s.c == nchunks - 1);
        const int hf = q >> 1;
        const float keep = (hf && !hasB) ? 0.f : 1.f;
#pragma unroll
        for (int ee = 0; ee < 2; ++ee) {
            const int e = 2 * (q & 1) + ee;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float d = row[hf][j][e];
                if (PSEL) d = (s.valid >> (i * 4 + j)) & 1 ? d : xt.pd[hf][e];      // (fetched from offset 0: replace by the padding value)
                if (XF) d = fmaxf(d * xt.sc[hf][e] + xt.sh[hf][e], 0.f);
                row[hf][j][e] = d * keep;
            }
        }
    };
    auto activate_row = [&](Row& row, const Step& s, const XfTab& xt, const int i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) activate_row_q(row, s, xt, i, q);
    };
    // Winograd row: t = ra + sg * rb, V = t B, three-plane split -> the operand planes of its four xi: quarter q
    // planes of a block as operand dwords: P[xi][plane] = 4 dwords = the lane's 8 k slots (dword 2 * half + e / 2 holds channels e, e + 1)
    auto make_planes_q = [&](const Row& ra, const Row& rb, const float sg, u32x4 (&P)[4][NPX], const int q) {
        if (WX6_KO & 1) return;
        const int hf = q >> 1, e = 2 * (q & 1);
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t0[j] = ra[hf][j][e] + sg * rb[hf][j][e];
            t1[j] = ra[hf][j][e + 1] + sg * rb[hf][j][e + 1];
        }
        const float v0[4] = {t0[0] - t0[2], t0[1] + t0[2], t0[2] - t0[1], t0[1] - t0[3]};
        const float v1[4] = {t1[0] - t1[2], t1[1] + t1[2], t1[2] - t1[1], t1[1] - t1[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (F16) {
                unsigned ph, pl;
                split2_pair(v0[j], v1[j], ph, pl);
                P[j][0][2 * hf + (q & 1)] = ph;
                P[j][1][2 * hf + (q & 1)] = pl;
            } else {
                unsigned ph, pm, pl;
                split3_pair(v0[j], v1[j], ph, pm, pl);
                P[j][0][2 * hf + (q & 1)] = ph;
                P[j][1][2 * hf + (q & 1)] = pm;
                P[j][NPX - 1][2 * hf + (q & 1)] = pl;
            }
        }
    };
    auto make_planes = [&](const Row& ra, const Row& rb, const float sg, u32x4 (&P)[4][NPX]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) make_planes_q(ra, rb, sg, P, q);
    };
        u16* const Pw = Pl + wv * PWAVE + lane * 8;
        auto put_planes = [&](const u32x4 (&P)[4][NPX]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < NPX; ++q) *reinterpret_cast<u32x4*>(Pw + (j * NPX + q) * 512) = P[j][q];
        };
        // barrier B of the block that is running (its planes are in the consumers' registers), the next block's planes -> LDS, barrier A of the next
        auto hand_over = [&](const u32x4 (&P)[4][NPX]) {
            __builtin_amdgcn_s_barrier();                 // B
            put_planes(P);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // A
        };
        Step cur;
        cur.item = item_begin;
        cur.c = 0;
        decode(cur.item, cur.pix, cur.valid);
        Step nxt = cur;
        advance(nxt);
        Row d0, d1, d2, d3, e2;                       // e2: the next step's row 2 (row 2 is the last to die)
        u32x4 P[4][NPX];
        XfTab xc, xn;
        load_xf(xc, cur);
        load_xf(xn, nxt);
        load_row(d0, cur, 0);
        load_row(d2, cur, 2);
        load_row(d1, cur, 1);
        load_row(d3, cur, 3);
        activate_row(d0, cur, xc, 0);
        activate_row(d2, cur, xc, 2);
        make_planes(d0, d2, -1.f, P);                 // row 0: t0 = d0 - d2
        put_planes(P);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // A of block 0
        for (int s = 0; s < nsteps; ++s) {
            // while the consumers run row 0: row 3 (t3 = d1 - d3); rows d0 / (next) d2 reload
            load_row(d0, nxt, 0);
            load_row(e2, nxt, 2);
            activate_row(d1, cur, xc, 1);
            activate_row(d3, cur, xc, 3);
            make_planes(d1, d3, -1.f, P);
            hand_over(P);
            // while they run row 3: row 1 (t1 = d1 + d2)
            load_row(d3, nxt, 3);
            make_planes(d1, d2, 1.f, P);
            hand_over(P);
            // while they run row 1: row 2 (t2 = d2 - d1)
            make_planes(d2, d1, -1.f, P);
            hand_over(P);
            // while they run row 2: the next step's row 0 (past the end: this step's again, unused)
            load_row(d1, nxt, 1);
            activate_row(d0, nxt, xn, 0);
            activate_row(e2, nxt, xn, 2);
            make_planes(d0, e2, -1.f, P);
            hand_over(P);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 4; ++j) d2[hf][j] = e2[hf][j];
            cur = nxt;
            xc = xn;
            advance(nxt);
            load_xf(xn, nxt);
        }
    } else {
        // =============================== consumer: U planes by LDS-DMA, MFMAs, output transform, epilogue =========================================
        f32x4 Y[2][2][NBT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int nb = 0; nb < NBT; ++nb) Y[a][b][nb] = zero4;
        f32x4 ssum[NBT], ssq[NBT];
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb) ssum[nb] = ssq[nb] = zero4;
        const bool vec = ((p.out_stride | p.out_ch_off | p.cout_g) & 3) == 0 && p.vec_ok;
        const int fo = r * 32 + ((kq ^ swz(r)) << 3);          // fragment offset inside a 16-row block of a tile
        // U planes of (chunk c, Winograd row i) -> LDS slot: XG * NPX * TILE / 512 pieces of 1 KB, consumer wave w moves pieces w, w + 4, ..
        auto stage_U_part = [&](const int c, const int i, const int slot, const int part) {      // part of NBT (-1: all)
            if (WX6_KO & 4) return;
            const u16* src = Ug + (size_t)c * CHUNK + i * STAGE;
            u16* dst = smem + slot * STAGE;
            constexpr int PW = STAGE / 512 / 4;       // pieces per consumer wave
#pragma unroll
            for (int k = 0; k < PW; ++k) {
                if (part >= 0 && k * NBT / PW != part) continue;
                const int piece = 4 * k + wv;
                dma16(src + piece * 512, dst + piece * 512);
            }
        };
        auto stage_U = [&](const int c, const int i, const int slot) { stage_U_part(c, i, slot, -1); };
    auto epilogue = [&](int item) {
            // lane (r, kq) holds Y[a][b] of channels n0 + nb*16 + 4*kq + j of tile r (conv_wino.hip's epilogue)
            const int t = (item * 4 + wv) * 16 + r;
            if (t < p.ntiles) {
                const int b = t / tiles_per_img, rem = t - b * tiles_per_img;
                const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
                const int y = 2 * ty, x = 2 * tx;
                const bool y1 = y + 1 < p.H, x1 = x + 1 < p.W;
                const int ch0 = g * p.cout_g + n0 + kq * 4;
                const size_t o00 = EPI == 2 ? ((size_t)(b * p.tiles_y + ty) * p.tiles_x + tx) * p.out_stride + p.out_ch_off + ch0
                                            : ((size_t)(b * p.H + y) * p.W + x) * p.out_stride + p.out_ch_off + ch0;
#pragma unroll
                for (int nb = 0; nb < NBT; ++nb) {
                    const int nrem = p.cout_g - (n0 + nb * 16 + kq * 4);      // channels of this quad that exist (zero rows of U beyond)
                    if (nrem <= 0) continue;
                    f32x4 bia = zero4;
                    if (p.bias) {
                        if (vec && nrem >= 4) bia = *reinterpret_cast<const f32x4*>(p.bias + ch0 + nb * 16);
                        else
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < nrem) bia[j] = p.bias[ch0 + nb * 16 + j];
                    }
                    f32x4 v[2][2];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) v[a][c2] = Y[a][c2][nb] + bia;
                    // (GSSD_OUT_HEADS: this quad of channels lies on one side of split_n -- both sides are multiples of four)
                    const int nq = n0 + nb * 16 + kq * 4;
                    float* const obase = (p.heads && nq >= p.split_n) ? p.out_b : p.out;
                    auto put = [&](size_t o, const f32x4& val) {
                        if (vec && nrem >= 4) *reinterpret_cast<f32x4*>(obase + o) = val;
                        else
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < nrem) obase[o + j] = val[j];
                    };
                    if (EPI == 2) {
                        // GSSD_CONV_POOL2: a Winograd tile IS a pooling window; batch sums in the order of the unpooled epilogue
                        f32x4 mx = v[0][0], mn = v[0][0];
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2) {
                                if ((a == 1 && !y1) || (c2 == 1 && !x1)) continue;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    mx[j] = fmaxf(mx[j], v[a][c2][j]);
                                    mn[j] = fminf(mn[j], v[a][c2][j]);
                                    ssum[nb][j] += v[a][c2][j];
                                    ssq[nb][j] = __builtin_fmaf(v[a][c2][j], v[a][c2][j], ssq[nb][j]);
                                }
                            }
                        f32x4 sgn = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (vec && nrem >= 4) sgn = *reinterpret_cast<const f32x4*>(p.pool_sign + ch0 + nb * 16);
                        else
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < nrem) sgn[j] = p.pool_sign[ch0 + nb * 16 + j];
                        f32x4 res;
#pragma unroll
                        for (int j = 0; j < 4; ++j) res[j] = sgn[j] >= 0.f ? mx[j] : mn[j];
                        put(o00 + nb * 16, res);
                    } else {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2) {
                                if ((a == 1 && !y1) || (c2 == 1 && !x1)) continue;
                                size_t o = o00 + nb * 16 + ((size_t)a * p.W + c2) * p.out_stride;
                                if (p.heads) {
                                    const size_t pix = (size_t)(y + a) * p.W + x + c2;
                                    o = nq < p.split_n ? (size_t)b * p.out_bs + p.out_off + pix * p.split_n + nq
                                                       : (size_t)b * p.outb_bs + p.outb_off + pix * (p.Cout - p.split_n) + (nq - p.split_n);
                                }
                                f32x4 val = v[a][c2];
                                if (EPI == 1) {
                                    if (vec && nrem >= 4) val += *reinterpret_cast<const f32x4*>(p.resid + o);
                                    else
#pragma unroll
                                        for (int j = 0; j < 4; ++j)
                                            if (j < nrem) val[j] += p.resid[o + j];
                                }
                                put(o, val);
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    ssum[nb][j] += val[j];
                                    ssq[nb][j] = __builtin_fmaf(val[j], val[j], ssq[nb][j]);
                                }
                            }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int nb = 0; nb < NBT; ++nb) Y[a][b2][nb] = zero4;
        };


        // the four xi of Winograd row I: 24 NBT MFMAs from the row's planes and LDS slot `slot`, folded into the 2 x 2 outputs right away
        // (Y = A^T M A is linear in M: Y[a][b] += A[I][a] (M_I. A)[b]; A^T = [1 1 1 0; 0 1 -1 -1])

        // Winograd row I: the planes of its four xi (LDS -> registers, then barrier B), 24 NBT MFMAs against the U planes of LDS slot `slot`, folded
        // into the 2 x 2 outputs right away (Y[a][b] += A[I][a] (M_I. A)[b]; A^T = [1 1 1 0; 0 1 -1 -1])
        const u16* const Pr = Pl + wv * PWAVE + lane * 8;
        auto run_row = [&](const int I, const int slot, const int c_next, const int i_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the block's U planes have landed
            __builtin_amdgcn_s_barrier();                 // A: everyone's have, the producers' planes are in LDS
            bf16x8 Pc[4][NPX];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < NPX; ++q) Pc[j][q] = *reinterpret_cast<const bf16x8*>(Pr + (j * NPX + q) * 512);
            const u16* ub = smem + slot * STAGE + fo;
            // the 12 weight fragments of output tile nb + 1 are requested before the 24 MFMAs of tile nb (an LDS read issued right in front of
            // its MFMA costs the wave the LDS latency: 64 such waits per step were ~10 k of the consumer's 23 k cycles); the next block's DMA pieces
            // go out a few per tile
            bf16x8 u[2][XG][NPX];
            auto frags = [&](const int nb, bf16x8 (&dst)[XG][NPX]) {
#pragma unroll
                for (int xl = 0; xl < XG; ++xl)
#pragma unroll
                    for (int q = 0; q < NPX; ++q) {
                        if (WX6_KO & 16) asm volatile("" : "=v"(dst[xl][q]));
                        else dst[xl][q] = *reinterpret_cast<const bf16x8*>(ub + (xl * NPX + q) * TILE + nb * 16 * 32);
                    }
            };
            // the first tile's weight fragments are requested behind the planes' reads (the block's U planes are in LDS since barrier A): LDS
            // returns in order, so the planes are in registers when at most these XG * NPX reads are outstanding -- barrier B does not wait
            // for them, and their round trip runs beside it instead of behind it
            frags(0, u[0]);
            if (WX6_KO & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(XG * NPX) : "memory");
            __builtin_amdgcn_s_barrier();                 // B: the plane buffer is free for the next block
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb) {
                if (nb + 1 < NBT) frags(nb + 1, u[(nb + 1) & 1]);
                stage_U_part(c_next, i_next, slot ^ 1, nb);
                f32x4 m[XG];
#pragma unroll
                for (int xl = 0; xl < XG; ++xl) {
                    const bf16x8 (&uc)[NPX] = u[nb & 1][xl];
                    // six products, smallest first (uc: weight planes, Pc: activation planes), summed from zero
                    f32x4 s6 = zero4;
                    if constexpr (F16) {
                        // h h' from zero; h l' + l' h' from zero, entering with 1 / 2048
                        f32x4 sx = zero4;
                        if (!(WX6_KO & 2)) {
                            sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, uc[1]), __builtin_bit_cast(f16x8, Pc[xl][0]), sx, 0, 0, 0);
                            sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, uc[0]), __builtin_bit_cast(f16x8, Pc[xl][1]), sx, 0, 0, 0);
                            s6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, uc[0]), __builtin_bit_cast(f16x8, Pc[xl][0]), s6, 0, 0, 0);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) s6[e] = __builtin_fmaf(sx[e], 1.f / 2048.f, s6[e]);
                    } else if (!(WX6_KO & 2)) {
                        s6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uc[1], Pc[xl][1], s6, 0, 0, 0);
                        s6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uc[NPX - 1], Pc[xl][0], s6, 0, 0, 0);
                        s6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uc[0], Pc[xl][NPX - 1], s6, 0, 0, 0);
                        s6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uc[1], Pc[xl][0], s6, 0, 0, 0);
                        s6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uc[0], Pc[xl][1], s6, 0, 0, 0);
                        s6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uc[0], Pc[xl][0], s6, 0, 0, 0);
                    }
                    m[xl] = s6;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s0 = m[0][e] + m[1][e] + m[2][e], s1 = m[1][e] - m[2][e] - m[3][e];
                    if (I < 3) {
                        Y[0][0][nb][e] += s0;
                        Y[0][1][nb][e] += s1;
                    }
                    if (I == 1) {
                        Y[1][0][nb][e] += s0;
                        Y[1][1][nb][e] += s1;
                    }
                    if (I >= 2) {
                        Y[1][0][nb][e] -= s0;
                        Y[1][1][nb][e] -= s1;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int item = item_begin, c = 0;
        stage_U(0, 0, 0);
        for (int s = 0; s < nsteps; ++s) {
            const int cn = c + 1 == nchunks ? 0 : c + 1;          // the next step's chunk (past the end: harmless)
            run_row(0, 0, c, 3);
            run_row(3, 1, c, 1);
            run_row(1, 0, c, 2);
            run_row(2, 1, cn, 0);
            if (c == nchunks - 1) {
                epilogue(item);
                ++item;
            }
            c = cn;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the DMA past the end)
        __builtin_amdgcn_s_barrier();                         // A of the block past the end (the producers' last hand-over); they exit behind it
        if (p.stats) {                                        // one flush per workgroup: 16 tile lanes of a kq -> LDS over waves -> fp64 atomics
            // (only the four consumer waves are left: the barriers below count the waves that have not ended)
            float* red = reinterpret_cast<float*>(smem);      // [4 waves][NB][2]
            float fs[4 * NBT], fq[4 * NBT];                   // value index i: channel n0 + 16 * (i >> 2) + 4 * kq + (i & 3)
#pragma unroll
            for (int i = 0; i < 4 * NBT; ++i) {
                fs[i] = ssum[i >> 2][i & 3];
                fq[i] = ssq[i >> 2][i & 3];
            }
            // halving exchange over the 16 tile lanes: lane r ends with value index r & (4 NBT - 1)
#pragma unroll
            for (int w = 8; w >= 1; w >>= 1) {
                if (4 * NBT <= w) continue;
                const bool up = (r & w) != 0;
#pragma unroll
                for (int i = 0; i < w; ++i) {
                    const float ks = up ? fs[i + w] : fs[i], gs = up ? fs[i] : fs[i + w];
                    const float kq2 = up ? fq[i + w] : fq[i], gq = up ? fq[i] : fq[i + w];
                    fs[i] = ks + __shfl_xor(gs, w, 64);
                    fq[i] = kq2 + __shfl_xor(gq, w, 64);
                }
            }
            if (4 * NBT == 8) {                               // the two tile-lane halves both hold value index r & 7
                fs[0] += __shfl_xor(fs[0], 8, 64);
                fq[0] += __shfl_xor(fq[0], 8, 64);
            }
            {
                const int vi = r & (4 * NBT - 1);
                if (r < 4 * NBT) {
                    red[(wv * NB + (vi >> 2) * 16 + kq * 4 + (vi & 3)) * 2 + 0] = fs[0];
                    red[(wv * NB + (vi >> 2) * 16 + kq * 4 + (vi & 3)) * 2 + 1] = fq[0];
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            const int ct = tid - 256;
            if (ct < NB && n0 + ct < p.cout_g) {
                double s = 0.0, q = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s += (double)red[(w * NB + ct) * 2 + 0];
                    q += (double)red[(w * NB + ct) * 2 + 1];
                }
                const int n = g * p.cout_g + n0 + ct;
                double* st = gssd_stats_replica(p.stats, p.stats_rep, p.Cout);
                unsafeAtomicAdd(st + n, s);
                unsafeAtomicAdd(st + p.Cout + n, q);
            }
        }
    }
}

// output-channel block of a layer
inline int wx6_nb(int cout_g) { return cout_g > 32 ? 64 : 32; }

// != 0: the padding vector lies this many bytes behind `in` (the layout the engine builds: directly behind the dense map, 32-bit reachable):
// out-of-image patch positions then LOAD their padding value (address select) instead of a per-element select afterwards
inline unsigned wx6_pad_off(const gssd_conv_desc& d) {
    if (!(d.in_scale && d.in_pad && (uintptr_t)d.in_pad > (uintptr_t)d.in)) return 0;
    const unsigned long long diff = (unsigned long long)((uintptr_t)d.in_pad - (uintptr_t)d.in);
    if (diff == (unsigned long long)d.B * d.H * d.W * d.in_stride * sizeof(float) && diff + (unsigned long long)d.in_stride * 4 < (1ull << 32) &&
        (diff & 15) == 0)
        return (unsigned)diff;
    return 0;
}

// packed K-major weights [Cout][tap * cin_g + ci] (row stride `ws`) -> the three bf16 planes of U = G g G^T in the kernel's staging order
__global__ void wino_x6_weight_kernel(const float* __restrict__ w, u16* __restrict__ Ux, int groups, int cout_g, int ncb, int NB, int cin_g, int nchunks,
                                      int ws) {
    const int cin_pad = nchunks * 32, cout_pad = ncb * NB;
    const int TILE = NB * 32, CHUNK = 16 * NP * TILE;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * cout_pad * cin_pad) return;
    const int ci = i % cin_pad, cop = i / cin_pad;
    const int g = cop / cout_pad, cg = cop - g * cout_pad;
    const bool live = cg < cout_g && ci < cin_g;
    float k[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) k[a][b] = live ? w[(size_t)(g * cout_g + cg) * ws + (a * 3 + b) * cin_g + ci] : 0.f;
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {                         // G g (the same arithmetic as wino_weight_kernel: U is the fp32 kernel's, split)
        t[0][b] = k[0][b];
        t[1][b] = 0.5f * (k[0][b] + k[1][b] + k[2][b]);
        t[2][b] = 0.5f * (k[0][b] - k[1][b] + k[2][b]);
        t[3][b] = k[2][b];
    }
    const int cb = cg / NB, row = cg % NB;
    const int c = ci / 32, wi = ci % 32;
    const int sub = wi >> 4, kq = (wi & 15) >> 2, e = wi & 3;
    const size_t base = ((size_t)(g * ncb + cb) * nchunks + c) * CHUNK + row * 32 + ((kq ^ swz(row & 15)) << 3) + 4 * sub + e;
    const size_t f16_base = (size_t)groups * ncb * nchunks * CHUNK;          // behind the bf16 planes
    const size_t base16 = ((size_t)(g * ncb + cb) * nchunks + c) * (16 * 2 * TILE) + row * 32 + ((kq ^ swz(row & 15)) << 3) + 4 * sub + e;
#pragma unroll
    for (int a = 0; a < 4; ++a) {                         // (G g) G^T
        const float u[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            __bf16 h, m, l;
            split3(u[b], h, m, l);
            u16* dst = Ux + base + (size_t)(a * 4 + b) * NP * TILE;
            dst[0 * TILE] = __builtin_bit_cast(u16, h);
            dst[1 * TILE] = __builtin_bit_cast(u16, m);
            dst[2 * TILE] = __builtin_bit_cast(u16, l);
            // the two fp16 planes of the same U (h, (u - h) * 2048), behind all bf16 planes, chunks of 16 x 2 x TILE
            const _Float16 fh = (_Float16)u[b];
            const _Float16 fl = (_Float16)((u[b] - (float)fh) * 2048.f);
            u16* d16 = Ux + f16_base + base16 + (size_t)(a * 4 + b) * 2 * TILE;
            d16[0 * TILE] = __builtin_bit_cast(u16, fh);
            d16[1 * TILE] = __builtin_bit_cast(u16, fl);
        }
    }
}

template <int NBT, bool XF, int EPI, bool PSEL, bool F16>
int launch_wino_x6_impl(const gssd_conv_desc& d, const u16* Ux, hipStream_t stream) {
    constexpr int NB = 16 * NBT, NPX = F16 ? 2 : 3;
    WinoX6Params p;
    p.in = d.in;
    p.Ux = Ux;
    p.bias = d.bias;
    p.out = d.out;
    p.resid = d.resid;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.in_pad = d.in_pad;
    p.pool_sign = (d.flags & GSSD_CONV_POOL2) ? d.pool_sign : nullptr;
    p.stats = d.stats;
    p.stats_rep = d.stats_rep;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    p.in_stride = d.in_stride;
    p.in_ch_off = d.in_ch_off;
    p.Cout = d.Cout;
    p.cin_g = d.cin_g;
    p.cout_g = d.Cout / d.groups;
    p.out_stride = d.out_stride;
    p.out_ch_off = d.out_ch_off;
    p.tiles_y = (d.H + 1) / 2;
    p.tiles_x = (d.W + 1) / 2;
    p.ntiles = d.B * p.tiles_y * p.tiles_x;
    p.ncb = (p.cout_g + NB - 1) / NB;
    p.nchunks = (d.cin_g + 31) / 32;
    p.npairs = p.ncb * d.groups;
    p.heads = d.out_mode == GSSD_OUT_HEADS;
    p.out_b = d.out_b;
    p.split_n = d.split_n;
    p.out_bs = d.out_batch_stride;
    p.outb_bs = d.outb_batch_stride;
    p.out_off = d.out_off;
    p.outb_off = d.outb_off;
    p.vec_ok = (((uintptr_t)d.out | (uintptr_t)d.bias | (uintptr_t)d.resid | (uintptr_t)p.pool_sign) & 15) == 0;
    if (p.heads) p.vec_ok = p.vec_ok && (((uintptr_t)d.out_b & 15) == 0) && ((d.out_batch_stride | d.outb_batch_stride | d.out_off | d.outb_off | d.split_n | (d.Cout - d.split_n)) & 3) == 0;
    p.pad_off = PSEL ? 0u : wx6_pad_off(d);
    constexpr size_t smem = (2 * (size_t)XG * NPX * NB * 32 + 4 * (size_t)XG * NPX * 512) * sizeof(u16);      // two U stages + the planes of a block
    auto kern = conv_wino_x6_kernel<NBT, XF, EPI, PSEL, F16>;
    static unsigned attr_mask = 0;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (winograd x6)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const int nitems = (p.ntiles + 63) / 64;
    int gx = 256 / p.npairs;                              // one workgroup per CU
    if (gx < 1) gx = 1;
    if (gx > nitems) gx = nitems;
    p.gx = gx;
    hipLaunchKernelGGL(kern, dim3(gx * p.npairs), dim3(512), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// the two-plane fp16 form: launches marked GSSD_CONV_F16_OK by the caller (operands inside fp16's range; never inferred); never a data gradient (EPI 1)
template <int NBT, bool XF, int EPI, bool PSEL>
int launch_wino_x6_sel(const gssd_conv_desc& d, const u16* Ux, hipStream_t stream) {
    static const bool f16_off = [] { const char* e = getenv("GSSD_X6_F16"); return e && e[0] == '0'; }();
    if constexpr (EPI != 1) {
        if (!f16_off && (d.flags & GSSD_CONV_F16_OK)) {
            const int cout_g = d.Cout / d.groups;
            const long long nb = cout_g > 32 ? 64 : 32, ncb = (cout_g + nb - 1) / nb, nchunks = (d.cin_g + 31) / 32;
            return launch_wino_x6_impl<NBT, XF, EPI, PSEL, true>(d, Ux + (long long)d.groups * ncb * nchunks * 16 * NP * nb * 32, stream);
        }
    }
    return launch_wino_x6_impl<NBT, XF, EPI, PSEL, false>(d, Ux, stream);
}

template <int NBT, bool XF, int EPI>
int launch_wino_x6(const gssd_conv_desc& d, const u16* Ux, hipStream_t stream) {
    if (XF && wx6_pad_off(d)) return launch_wino_x6_sel<NBT, XF, EPI, false>(d, Ux, stream);
    return launch_wino_x6_sel<NBT, XF, EPI, true>(d, Ux, stream);
}

}  // namespace

// bf16 elements of the three-plane U of a layer (0: not a shape this kernel takes)
long long gssd_wino_x6_plane_elems(int cout_g, int groups, int cin_g) {
    if (cin_g % 16 != 0 || cout_g < 24) return 0;
    const long long NB = wx6_nb(cout_g), ncb = (cout_g + NB - 1) / NB, nchunks = (cin_g + 31) / 32;
    return (long long)groups * ncb * nchunks * 16 * (NP + 2) * NB * 32;      // three bf16 planes, then two fp16 planes
}

int gssd_wino_x6_pack(const float* w_packed, void* Ux, int Cout, int groups, int cin_g, int row_stride, hipStream_t stream) {
    const int cout_g = Cout / groups;
    const int NB = wx6_nb(cout_g), ncb = (cout_g + NB - 1) / NB, nchunks = (cin_g + 31) / 32;
    const int n = groups * ncb * NB * nchunks * 32;
    hipLaunchKernelGGL(wino_x6_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, w_packed, reinterpret_cast<u16*>(Ux), groups, cout_g,
                       ncb, NB, cin_g, nchunks, row_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// GSSD_WINO_X6=0: the fp32-MFMA Winograd kernel (conv_wino.hip) keeps every launch (ablation / A-B); =2: every shape the kernel takes
int gssd_wino_x6_mode() {
    static const int mode = [] {
        const char* e = getenv("GSSD_WINO_X6");
        return e ? atoi(e) : 1;
    }();
    return mode;
}
bool gssd_wino_x6_enabled() { return gssd_wino_x6_mode() != 0; }

// the shapes it is used for by default: where it beat conv_wino.hip in a same-box A/B at batch 32 (scripts/wino_x6_ab.sh, round 5: conv3_1 176 vs
// 238 us, conv3_2 303 vs 353, conv4_1 145 vs 166, conv4_2 260 vs 273; 19 x 19 maps: 90 vs 91; 32-channel blocks (conv2_2) and 16-channel groups
// (conv2_1) lose: the vector work per output channel doubles / half of every MFMA's k slots are padding)
bool gssd_wino_x6_wanted(const gssd_conv_desc& d) {
    const int mode = gssd_wino_x6_mode();
    if (mode == 0) return false;
    if (mode == 2) return true;
    const int cout_g = d.Cout / d.groups;
    // the multibox head of a large map (round 6: 38 x 38, 24 = 16 loc | 8 conf channels of 512): flagged GSSD_CONV_F16_OK, 95 us against 173 us on the
    // fp32-MFMA implicit GEMM with its split-K slices; the 19 x 19 head stays there (217 vs 178 us)
    if (d.out_mode == GSSD_OUT_HEADS)
        return (d.flags & GSSD_CONV_F16_OK) && d.groups == 1 && cout_g >= 24 && d.cin_g >= 32 && (long long)d.B * ((d.H + 1) / 2) * ((d.W + 1) / 2) >= 8192;
    // whole 64-channel blocks, or at least 80 % of the padded ones: the dense DCN offset conv (1024 -> 108 channels = two blocks with 20 padding
    // rows) lost with the first producers (651 vs 601 us) and wins since the pair-wise split: 527 vs 575 us (512 -> 108: 277 vs 316)
    const int padded = (cout_g + 63) / 64 * 64;
    const bool blocks_ok = cout_g % 64 == 0 || (cout_g > 64 && 5 * cout_g >= 4 * padded);
    return blocks_ok && d.cin_g >= 32 && (long long)d.B * ((d.H + 1) / 2) * ((d.W + 1) / 2) >= 8192;
}

// called by gssd_try_conv_wino() once the descriptor is known to be a Winograd shape: `Ux` = the three-plane U behind the fp32 U
int gssd_launch_conv_wino_x6(const gssd_conv_desc& d, const void* Ux, hipStream_t stream) {
    const u16* ux = reinterpret_cast<const u16*>(Ux);
    const int epi = (d.flags & GSSD_CONV_POOL2) ? 2 : d.resid ? 1 : 0;
    const bool wide = wx6_nb(d.Cout / d.groups) == 64;
#define WX6_GO(NBT_)                                                                                                                        \
    (d.in_scale ? (epi == 2 ? launch_wino_x6<NBT_, true, 2>(d, ux, stream) : epi == 1 ? launch_wino_x6<NBT_, true, 1>(d, ux, stream)       \
                                                                                       : launch_wino_x6<NBT_, true, 0>(d, ux, stream))      \
                : (epi == 2 ? launch_wino_x6<NBT_, false, 2>(d, ux, stream) : epi == 1 ? launch_wino_x6<NBT_, false, 1>(d, ux, stream)     \
                                                                                        : launch_wino_x6<NBT_, false, 0>(d, ux, stream)))
    return wide ? WX6_GO(4) : WX6_GO(2);
#undef WX6_GO
}
