// Winograd F(2x2, 3x3) on the BF16 matrix cores with fp32-equivalent products (round 5): the grouped 3x3 / stride 1 / pad 1 trunk
// convolutions conv2_1 .. conv5_3 (and their data gradients) of models/ssd_multiphase_custom_group.py:434-460 (vgg()).
//
// conv_wino.hip runs the 16 Winograd-domain GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[co][ci]  on v_mfma_f32_16x16x4_f32 -- 1/16 of
// the bf16 matrix rate of gfx950.  Here both operands are the exact sum of three bf16 planes (x = h + m + l, conv_x6.hip) and a product is six
// v_mfma_f32_16x16x32_bf16 (every term above 2^-24), fp32 accumulation: 6 / 16 of the fp32 instruction's matrix-pipe time per product, on top
// of Winograd's 2.25 x fewer products.  The transforms stay fp32 and are the ones of conv_wino.hip (V = B^T d B in registers straight from
// global memory, Y = A^T M A in registers), so the result differs from that kernel only by the products' last bit.
//
// Work decomposition: a workgroup (4 waves, one per SIMD) is persistent and walks a contiguous range of items of one (group, 32-channel output
// block); an item = 64 consecutive tiles of the linearised (image, tile_y, tile_x) list, 16 per wave.  K runs in chunks of 32 input channels: lane
// (tile r, quad kq) owns channels 4 kq .. 4 kq + 3 of both 16-channel halves of the chunk -- the loads, the fused producer BatchNorm + ReLU and
// the transform keep conv_wino.hip's layout (one 16-byte load per patch position and half) and the two halves fill slots 0-3 / 4-7 of the
// lane's bf16x8 MFMA operand; U is packed in the same slot order (the MFMA only needs both operands to agree on which channel sits in which
// k slot).  The vector ALU reaches only the 256 architectural registers and the accumulators take 128 more (16 xi x 2 x f32x4), so the planes of a
// chunk (16 xi x 3 x 4 registers) never exist at once: the column transform t = B^T d is done in place in the patch registers, and the chunk then
// runs ROW by row of the Winograd domain -- V_i. = t_i. B, its three-plane split (48 registers) and the 48 MFMAs of its four xi -- with row i + 1's
// vector work in the same basic block as row i's MFMAs (the bf16 MFMA does not use the vector ALU's lanes: the two overlap).  A row is also the
// LDS stage of the U planes (LDS-DMA, 4 xi = 24 KB, double buffered).  Rows of t die as they are consumed; the next step's patch loads are issued
// into the freed registers (first half during row 1, second half during row 3).
#include "common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#ifndef WX6_LOCAL_SUM
#define WX6_LOCAL_SUM 1       // the six products of a chunk are summed from zero and added to the running sum by the vector ALU (the bf16 MFMA's
#endif                        // adder truncates: conv_x6.hip)

namespace {

constexpr int NB = 32, NBT = 2, NP = 3, XG = 4;
constexpr int TILE_ELEMS = 32 * 32;                 // bf16 per (xi, plane): 32 output channels x 32 k slots
constexpr int STAGE_ELEMS = XG * NP * TILE_ELEMS;   // 24 KB
constexpr int CHUNK_ELEMS = 16 * NP * TILE_ELEMS;   // 96 KB per (group, output block, 32-channel chunk)

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

struct WinoX6Params {
    const float* in;
    const u16* Ux;           // [groups][cout blocks][chunks][16 xi][3 planes][32 co][32 slots], slot groups swizzled (swz)
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
};

// one 16-channel half of a chunk, in place: producer BatchNorm + ReLU, padding, then the column transform raw[i * 4 + j] := (B^T d)[i][j]
template <bool XF>
__device__ __forceinline__ void xform_cols(f32x4 (&raw)[16], const f32x4& sc, const f32x4& sh, const f32x4& padq, const unsigned valid,
                                           const bool select_pad) {
    if (select_pad) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[q][e] = (valid >> q) & 1 ? raw[q][e] : padq[e];
    }
    if (XF) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[q][e] = fmaxf(raw[q][e] * sc[e] + sh[e], 0.f);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 d0 = raw[0 * 4 + j], d1 = raw[1 * 4 + j], d2 = raw[2 * 4 + j], d3 = raw[3 * 4 + j];
        raw[0 * 4 + j] = d0 - d2;
        raw[1 * 4 + j] = d1 + d2;
        raw[2 * 4 + j] = d2 - d1;
        raw[3 * 4 + j] = d1 - d3;
    }
}

// row I of the Winograd domain: V[I][j] = (t[I][.] B)[j] for the lane's 8 channels (tA: slots 0-3, tB: slots 4-7), x = h + m + l
template <int I>
__device__ __forceinline__ void make_row(const f32x4 (&tA)[16], const f32x4 (&tB)[16], bf16x8 (&P)[4][NP]) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t0 = sub ? tB[I * 4 + 0][e] : tA[I * 4 + 0][e], t1 = sub ? tB[I * 4 + 1][e] : tA[I * 4 + 1][e];
            const float t2 = sub ? tB[I * 4 + 2][e] : tA[I * 4 + 2][e], t3 = sub ? tB[I * 4 + 3][e] : tA[I * 4 + 3][e];
            const float v[4] = {t0 - t2, t1 + t2, t2 - t1, t1 - t3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 h, m, l;
                split3(v[j], h, m, l);
                P[j][0][4 * sub + e] = h;
                P[j][1][4 * sub + e] = m;
                P[j][2][4 * sub + e] = l;
            }
        }
    }
}

template <bool XF, int EPI>      // EPI: 0 plain, 1 + residual, 2 pooled raw map (GSSD_CONV_POOL2)
__global__ __launch_bounds__(256, 1) void conv_wino_x6_kernel(const WinoX6Params p) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];      // [2][STAGE_ELEMS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int nitems = (p.ntiles + 63) >> 6;
    // flat grid: workgroup id = pair + npairs * x.  Ids go round-robin over the 8 XCDs (each with its own L2): the workgroups that stream the
    // U planes of one (group, output block) pair share an XCD (npairs a multiple of 8) or two
    const int pair = blockIdx.x % p.npairs, bx = blockIdx.x / p.npairs;
    const int g = pair / p.ncb, cb = pair - g * p.ncb;
    const int n0 = cb * NB;
    int item = (int)(((long long)bx * nitems) / p.gx);
    const int item_end = (int)(((long long)(bx + 1) * nitems) / p.gx);
    if (item >= item_end) return;

    const int cb_ld = p.in_ch_off + g * p.cin_g + kq * 4;            // + chunk * 32 + half * 16
    const u16* Ug = p.Ux + (size_t)pair * p.nchunks * CHUNK_ELEMS + lane * 8;

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
    // patch position q of the lane's tile, channels cb_ld + ch16 * 16 .. + 3 (ch16 = 2 * chunk + half)
    auto load_raw1 = [&](f32x4 (&raw)[16], unsigned pix_off, unsigned valid, int ch16, int q) {
        const int i = q >> 2, j = q & 3;
        const unsigned off = (valid >> q) & 1 ? pix_off + (unsigned)(((i * p.W + j) * p.in_stride + ch16 * 16) * 4)
                                              : (p.pad_off ? p.pad_off + (unsigned)((cb_ld + ch16 * 16) * 4) : 0u);
        raw[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in) + off);
    };
    // stage (chunk c, xi group xg) -> LDS slot: 24 pieces of 1 KB, wave w moves pieces w, w + 4, ..
    auto stage_U = [&](int c, int xg, int slot) {
        const u16* src = Ug + (size_t)c * CHUNK_ELEMS + xg * STAGE_ELEMS;
        u16* dst = smem + slot * STAGE_ELEMS;
#pragma unroll
        for (int k = 0; k < STAGE_ELEMS / 512 / 4; ++k) {
            const int piece = 4 * k + wv;
            dma16(src + piece * 512, dst + piece * 512);
        }
    };

    f32x4 acc[16][NBT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb) acc[xi][nb] = zero4;
    f32x4 ssum[NBT], ssq[NBT];
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) ssum[nb] = ssq[nb] = zero4;
    const bool vec = ((p.out_stride | p.out_ch_off | p.cout_g) & 3) == 0 && p.vec_ok;

    unsigned in_cur, valid_cur;
    decode(item, in_cur, valid_cur);
    stage_U(0, 0, 0);
    f32x4 tA[16], tB[16];                                 // patch (raw), then t = B^T d in place; slots 0-3 / 4-7 of the operand
#pragma unroll
    for (int q = 0; q < 16; ++q) load_raw1(tA, in_cur, valid_cur, 0, q);
    if (p.cin_g > 16) {
#pragma unroll
        for (int q = 0; q < 16; ++q) load_raw1(tB, in_cur, valid_cur, 1, q);
    }
    int slot = 0;
    const int fo = r * 32 + ((kq ^ swz(r)) << 3);          // fragment offset inside a 16-row half of a tile

    // the four xi of row `xg` (U planes from LDS slot `sl`) against the row's operand planes
    auto mfma_row = [&](const int xg, const int sl, const bf16x8 (&P)[4][NP]) {
        const u16* ub = smem + sl * STAGE_ELEMS + fo;
#pragma unroll
        for (int xl = 0; xl < XG; ++xl) {
            const int xi = xg * XG + xl;
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb) {
                bf16x8 u[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) u[q] = *reinterpret_cast<const bf16x8*>(ub + (xl * NP + q) * TILE_ELEMS + nb * 16 * 32);
                // six products, smallest first (u: weight planes, P: activation planes)
#if WX6_LOCAL_SUM
                f32x4 s = zero4;
#else
                f32x4 s = acc[xi][nb];
#endif
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[1], P[xl][1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[2], P[xl][0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[0], P[xl][2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[1], P[xl][0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[0], P[xl][1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[0], P[xl][0], s, 0, 0, 0);
#if WX6_LOCAL_SUM
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[xi][nb][e] += s[e];
#else
                acc[xi][nb] = s;
#endif
            }
        }
    };
    // top of a stage: this wave's DMA pieces of the stage have landed (vmcnt(0): with them every patch load issued a stage ago), then
    // everyone's have and everyone is done reading the other slot
    auto stage_sync = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    };

    for (;;) {
        const int item_next = item + 1;
        const bool have_next = item_next < item_end;
        unsigned in_next = in_cur, valid_next = 0;
        if (have_next) decode(item_next, in_next, valid_next);
        for (int c = 0; c < p.nchunks; ++c) {
            const bool hasB = c * 32 + 16 < p.cin_g;
            const bool last = c + 1 == p.nchunks;
            const bool more = !last || have_next;
            const unsigned ld_in = last ? in_next : in_cur;
            const unsigned ld_valid = last ? valid_next : valid_cur;
            const int ld_c = last ? 0 : c + 1;
            const bool ld_hasB = ld_c * 32 + 16 < p.cin_g;
            // ---- column transforms of both halves, in place ------------------------------------------------------------------------------
            {
                f32x4 sc = zero4, sh = zero4, padq = zero4;
                if (XF) {
                    sc = *reinterpret_cast<const f32x4*>(p.in_scale + cb_ld + c * 32);
                    sh = *reinterpret_cast<const f32x4*>(p.in_shift + cb_ld + c * 32);
                    padq = *reinterpret_cast<const f32x4*>(p.in_pad + cb_ld + c * 32);
                }
                xform_cols<XF>(tA, sc, sh, padq, valid_cur, !p.pad_off);
            }
            if (hasB) {
                f32x4 sc = zero4, sh = zero4, padq = zero4;
                if (XF) {
                    sc = *reinterpret_cast<const f32x4*>(p.in_scale + cb_ld + c * 32 + 16);
                    sh = *reinterpret_cast<const f32x4*>(p.in_shift + cb_ld + c * 32 + 16);
                    padq = *reinterpret_cast<const f32x4*>(p.in_pad + cb_ld + c * 32 + 16);
                }
                xform_cols<XF>(tB, sc, sh, padq, valid_cur, !p.pad_off);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) tB[q] = zero4;          // a 16-channel tail: slots 4-7 are zero on both sides (U is zero-padded)
            }
            // ---- rows: vector work of row i + 1 beside the MFMAs of row i ------------------------------------------------------------------
            bf16x8 P0[4][NP], P1[4][NP];
            f32x4 nA[16], nB[16];                             // the next step's patch
            make_row<0>(tA, tB, P0);
            stage_sync();
            stage_U(c, 1, slot ^ 1);
            make_row<1>(tA, tB, P1);
            mfma_row(0, slot, P0);
            stage_sync();
            stage_U(c, 2, slot);
            if (more) {
#pragma unroll
                for (int q = 0; q < 16; ++q) load_raw1(nA, ld_in, ld_valid, 2 * ld_c, q);
            }
            make_row<2>(tA, tB, P0);
            mfma_row(1, slot ^ 1, P1);
            stage_sync();
            stage_U(c, 3, slot ^ 1);
            make_row<3>(tA, tB, P1);
            mfma_row(2, slot, P0);
            stage_sync();
            if (more) {
                stage_U(ld_c, 0, slot);
                if (ld_hasB) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) load_raw1(nB, ld_in, ld_valid, 2 * ld_c + 1, q);
                }
            }
            mfma_row(3, slot ^ 1, P1);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                tA[q] = nA[q];
                tB[q] = nB[q];
            }
        }

        // ---- output transform + epilogue (conv_wino.hip's): lane (r, kq) holds M[co n0 + nb*16 + 4*kq + j][tile r] for all 16 xi ------------
        {
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
                    f32x4 s0[4], s1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {             // A^T M, four channels at a time
                        s0[j] = acc[0 * 4 + j][nb] + acc[1 * 4 + j][nb] + acc[2 * 4 + j][nb];
                        s1[j] = acc[1 * 4 + j][nb] - acc[2 * 4 + j][nb] - acc[3 * 4 + j][nb];
                    }
                    f32x4 bia = zero4;
                    if (p.bias) {
                        if (vec && nrem >= 4) bia = *reinterpret_cast<const f32x4*>(p.bias + ch0 + nb * 16);
                        else
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < nrem) bia[j] = p.bias[ch0 + nb * 16 + j];
                    }
                    f32x4 v[2][2];
                    v[0][0] = s0[0] + s0[1] + s0[2] + bia;
                    v[0][1] = s0[1] - s0[2] - s0[3] + bia;
                    v[1][0] = s1[0] + s1[1] + s1[2] + bia;
                    v[1][1] = s1[1] - s1[2] - s1[3] + bia;
                    auto put = [&](size_t o, const f32x4& val) {
                        if (vec && nrem >= 4) *reinterpret_cast<f32x4*>(p.out + o) = val;
                        else
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < nrem) p.out[o + j] = val[j];
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
                        f32x4 sg = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (vec && nrem >= 4) sg = *reinterpret_cast<const f32x4*>(p.pool_sign + ch0 + nb * 16);
                        else
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < nrem) sg[j] = p.pool_sign[ch0 + nb * 16 + j];
                        f32x4 res;
#pragma unroll
                        for (int j = 0; j < 4; ++j) res[j] = sg[j] >= 0.f ? mx[j] : mn[j];
                        put(o00 + nb * 16, res);
                    } else {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2) {
                                if ((a == 1 && !y1) || (c2 == 1 && !x1)) continue;
                                const size_t o = o00 + nb * 16 + ((size_t)a * p.W + c2) * p.out_stride;
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
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!have_next) break;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb) acc[xi][nb] = zero4;
        item = item_next;
        in_cur = in_next;
        valid_cur = valid_next;
    }

    if (p.stats) {                                        // one flush per workgroup: 16 tile lanes of a kq -> LDS over waves -> fp64 atomics
        __syncthreads();                                  // all waves are done with the U stages (no DMA in flight: the last step issued none)
        float* red = reinterpret_cast<float*>(smem);      // [4 waves][NB][2]
        float fs[8], fq[8];                               // value index i: channel n0 + 16 * (i >> 2) + 4 * kq + (i & 3)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fs[i] = ssum[i >> 2][i & 3];
            fq[i] = ssq[i >> 2][i & 3];
        }
        // halving exchange over the 16 tile lanes: lane r ends with value index r & 7 (lanes r and r ^ 8 hold the same value, folded below)
#pragma unroll
        for (int w = 4; w >= 1; w >>= 1) {
            const bool up = (r & w) != 0;
#pragma unroll
            for (int i = 0; i < w; ++i) {
                const float ks = up ? fs[i + w] : fs[i], gs = up ? fs[i] : fs[i + w];
                const float kq2 = up ? fq[i + w] : fq[i], gq = up ? fq[i] : fq[i + w];
                fs[i] = ks + __shfl_xor(gs, w, 64);
                fq[i] = kq2 + __shfl_xor(gq, w, 64);
            }
        }
        fs[0] += __shfl_xor(fs[0], 8, 64);
        fq[0] += __shfl_xor(fq[0], 8, 64);
        if (r < 8) {
            red[(wv * NB + (r >> 2) * 16 + kq * 4 + (r & 3)) * 2 + 0] = fs[0];
            red[(wv * NB + (r >> 2) * 16 + kq * 4 + (r & 3)) * 2 + 1] = fq[0];
        }
        __syncthreads();
        if (tid < NB && n0 + tid < p.cout_g) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s += (double)red[(w * NB + tid) * 2 + 0];
                q += (double)red[(w * NB + tid) * 2 + 1];
            }
            const int n = g * p.cout_g + n0 + tid;
            double* st = gssd_stats_replica(p.stats, p.stats_rep, p.Cout);
            unsafeAtomicAdd(st + n, s);
            unsafeAtomicAdd(st + p.Cout + n, q);
        }
    }
}

// packed K-major weights [Cout][tap * cin_g + ci] (row stride `ws`) -> the three bf16 planes of U = G g G^T in the kernel's staging order
__global__ void wino_x6_weight_kernel(const float* __restrict__ w, u16* __restrict__ Ux, int groups, int cout_g, int ncb, int cin_g, int nchunks,
                                      int ws) {
    const int cin_pad = nchunks * 32, cout_pad = ncb * NB;
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
    const size_t base = ((size_t)(g * ncb + cb) * nchunks + c) * CHUNK_ELEMS + row * 32 + ((kq ^ swz(row & 15)) << 3) + 4 * sub + e;
#pragma unroll
    for (int a = 0; a < 4; ++a) {                         // (G g) G^T
        const float u[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            __bf16 h, m, l;
            split3(u[b], h, m, l);
            u16* dst = Ux + base + (size_t)(a * 4 + b) * NP * TILE_ELEMS;
            dst[0 * TILE_ELEMS] = __builtin_bit_cast(u16, h);
            dst[1 * TILE_ELEMS] = __builtin_bit_cast(u16, m);
            dst[2 * TILE_ELEMS] = __builtin_bit_cast(u16, l);
        }
    }
}

template <bool XF, int EPI>
int launch_wino_x6(const gssd_conv_desc& d, const u16* Ux, hipStream_t stream) {
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
    p.vec_ok = (((uintptr_t)d.out | (uintptr_t)d.bias | (uintptr_t)d.resid | (uintptr_t)p.pool_sign) & 15) == 0;
    p.pad_off = 0;
    if (XF && d.in_pad && (uintptr_t)d.in_pad > (uintptr_t)d.in) {
        const unsigned long long diff = (unsigned long long)((uintptr_t)d.in_pad - (uintptr_t)d.in);
        // only the layout the engine builds: the vector directly behind the dense map (so the 32-bit offsets of the kernel reach it)
        if (diff == (unsigned long long)d.B * d.H * d.W * d.in_stride * sizeof(float) && diff + (unsigned long long)d.in_stride * 4 < (1ull << 32) &&
            (diff & 15) == 0)
            p.pad_off = (unsigned)diff;
    }
    constexpr size_t smem = 2 * (size_t)STAGE_ELEMS * sizeof(u16);
    auto kern = conv_wino_x6_kernel<XF, EPI>;
    static unsigned attr_mask = 0;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (winograd x6)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const int nitems = (p.ntiles + 63) / 64;
    int gx = 256 / p.npairs;                              // one workgroup per CU (448 registers per lane)
    if (gx < 1) gx = 1;
    if (gx > nitems) gx = nitems;
    p.gx = gx;
    hipLaunchKernelGGL(kern, dim3(gx * p.npairs), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

// bf16 elements of the three-plane U of a layer (0: not a shape this kernel takes)
long long gssd_wino_x6_plane_elems(int cout_g, int groups, int cin_g) {
    if (cin_g % 16 != 0 || cout_g < 24) return 0;
    const long long ncb = (cout_g + NB - 1) / NB, nchunks = (cin_g + 31) / 32;
    return (long long)groups * ncb * nchunks * CHUNK_ELEMS;
}

int gssd_wino_x6_pack(const float* w_packed, void* Ux, int Cout, int groups, int cin_g, int row_stride, hipStream_t stream) {
    const int cout_g = Cout / groups;
    const int ncb = (cout_g + NB - 1) / NB, nchunks = (cin_g + 31) / 32;
    const int n = groups * ncb * NB * nchunks * 32;
    hipLaunchKernelGGL(wino_x6_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, w_packed, reinterpret_cast<u16*>(Ux), groups, cout_g,
                       ncb, cin_g, nchunks, row_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// GSSD_WINO_X6=1 switches it on (default off: see DESIGN.md -- with one wave per SIMD the vector work of the three-plane split does not hide)
bool gssd_wino_x6_enabled() {
    static const bool on = [] {
        const char* e = getenv("GSSD_WINO_X6");
        return e && e[0] == '1';
    }();
    return on;
}

// called by gssd_try_conv_wino() once the descriptor is known to be a Winograd shape: `Ux` = the three-plane U behind the fp32 U
int gssd_launch_conv_wino_x6(const gssd_conv_desc& d, const void* Ux, hipStream_t stream) {
    const u16* ux = reinterpret_cast<const u16*>(Ux);
    const int epi = (d.flags & GSSD_CONV_POOL2) ? 2 : d.resid ? 1 : 0;
    if (d.in_scale) return epi == 2 ? launch_wino_x6<true, 2>(d, ux, stream) : epi == 1 ? launch_wino_x6<true, 1>(d, ux, stream) : launch_wino_x6<true, 0>(d, ux, stream);
    return epi == 2 ? launch_wino_x6<false, 2>(d, ux, stream) : epi == 1 ? launch_wino_x6<false, 1>(d, ux, stream) : launch_wino_x6<false, 0>(d, ux, stream);
}
