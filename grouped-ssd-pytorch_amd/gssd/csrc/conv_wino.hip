// Winograd F(2x2, 3x3) for the grouped 3x3 / stride 1 / pad 1 trunk convolutions (conv2_x .. conv5_x and their data
// gradients) on gfx950.  These layers sit far above the fp32 ridge, so the direct implicit GEMM is bound by the
// fp32 MFMA rate; Winograd spends 16 multiplies per 2x2 output tile and (cin, cout) pair instead of 36, i.e. 2.25x fewer
// MFMA flops for the same convolution:
//     V = B^T d B   (4x4 input tile d, per channel)            -- in registers, straight from global memory
//     M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[co][ci]    -- 16 independent GEMMs (xi = 0..15) on v_mfma_f32_16x16x4_f32
//     Y = A^T M A   (2x2 outputs)                              -- in registers (all 16 xi of a (tile, co) live in one lane)
// with U = G g G^T precomputed per launch of the weight packer (gssd_winograd_weight_f32).
//
// Work decomposition: a wave owns 16 consecutive tiles of the linearised (image, tile_y, tile_x) list of one group (so any
// map size fills the chip without spatial padding waste) and NB output channels; the 4 waves of a workgroup share the
// U slices, which are staged through LDS with the LDS-DMA path, double buffered over 16-channel chunks.  The A operand
// never touches LDS: lane (tile r, k-quad kq) fetches the 16 B of its four channels for each of the 16 patch positions
// directly (the next chunk's 16 loads are in flight during the current chunk's MFMAs), applies the fused producer
// BatchNorm + ReLU if requested, and transforms in registers.  One 16-byte load therefore feeds four k-steps, like the
// b128 trick of the wgrad kernel.  Accumulators: 16 xi x NB/16 tiles of 16x16 = 256 VGPRs for NB = 64 (one wave per SIMD;
// the unified 512-entry register file of gfx950 is what makes this tiling possible).
#include "common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef WINO_TIMING
// debug build (scripts/wino_timing.py): wave 0 of every workgroup accumulates the shader clocks between its phase boundaries
__device__ unsigned long long g_wino_timing[8];
extern "C" int gssd_wino_timing_read(unsigned long long* out8) {
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wino_timing), 64);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_wino_timing), z, 64);
    return 0;
}
#define WSTAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; }
#else
#define WSTAMP(k)
#endif

namespace {

__device__ __attribute__((aligned(16))) float g_zero_page_wino[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct WinoParams {
    const float* in;
    const float* U;          // [groups][16][cout_pad][cin_g]  (rows >= cout_g are zero)
    const float* bias;
    float* out;
    const float* resid;
    const float* in_scale;
    const float* in_shift;
    const float* in_pad;
    const float* pool_sign;  // GSSD_CONV_POOL2: `out` is the 2x2 / stride-2 (ceil) pooled raw map, max where pool_sign[c] >= 0 else min
    double* stats;
    int stats_rep;
    int B, H, W, in_stride, in_ch_off, Cout, cin_g, cout_g, cout_pad, out_stride, out_ch_off;
    int tiles_y, tiles_x, ntiles;      // per group: B * tiles_y * tiles_x
    int vec_ok;                        // bias / resid / out / pool_sign pointers are 16-byte aligned
    unsigned pad_off;                  // != 0: the padding vector lies pad_off bytes behind `in` (32-bit reachable): out-of-image patch
                                       // positions LOAD their padding value (address select) instead of a per-element select afterwards
};

template <int NB, bool XF, bool PERSIST, int EPI>      // EPI: 0 plain, 1 + residual, 2 pooled raw map (GSSD_CONV_POOL2)
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const WinoParams p) {
    constexpr int NBT = NB / 16;                  // 16-wide output-channel tiles per wave
    constexpr int STAGE = 16 * NB * 16;           // floats per U stage: [xi][n][16 ci]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int nitems = (p.ntiles + 63) >> 6;      // work item = 64 tiles (16 per wave) x NB output channels
    int g, n0, item;
    int item_end = nitems;
    if (PERSIST) {
        // persistent workgroup b of gridDim.x walks the contiguous item range [b n / G, (b + 1) n / G): consecutive items
        // overlap in two patch rows, which the same CU then finds in L1 / L2
        g = blockIdx.z;
        n0 = blockIdx.y * NB;
        item = (int)(((long long)blockIdx.x * nitems) / gridDim.x);
        item_end = (int)(((long long)(blockIdx.x + 1) * nitems) / gridDim.x);
    } else {
        // One item per workgroup, flat grid.  Workgroup ids go round-robin over the 8 XCDs, each with its own L2: the
        // output-channel blocks of one tile group get consecutive slots ON THE SAME XCD (ids 8 apart), so the second block
        // finds the input patch in that L2 instead of fetching it from HBM again.
        const int ncb = p.cout_pad / NB;
        const int per_group = ((nitems + 7) >> 3) * 8 * ncb;
        g = blockIdx.x / per_group;
        const int id = blockIdx.x - g * per_group;
        const int s = id >> 3;
        item = (id & 7) * ((nitems + 7) >> 3) + s / ncb;       // XCD k walks a contiguous eighth of the tile list: neighbouring
        n0 = (s % ncb) * NB;                                    // items (overlapping patch rows) also share its L2
        if (s / ncb >= ((nitems + 7) >> 3)) item = nitems;
    }
    if (item >= item_end) return;

    // Lane (r, kq) fetches quad kq of tile r -- the MFMA's own layout.  (Rounds 1-2 loaded quad (l & 3) of tile (l >> 2), four
    // consecutive lanes per 64-byte segment, and permuted with 64 ds_bpermute per chunk: beside fp32 MFMAs those cost ~800 cycles per
    // chunk in full (profiles/r03_mfma_valu_overlap.txt), the four-times-higher line count per load instruction costs nothing that shows:
    // same-box A/B conv2_2 450 -> 401 us, conv4_2 305 -> 290, conv5_x 103 -> 98, conv3_1 224 -> 231.)
    const int ld_quad = kq;
    const int cb_ld = p.in_ch_off + g * p.cin_g + ld_quad * 4;       // + chunk * 16   (loading lanes)
    const int cb_mf = p.in_ch_off + g * p.cin_g + kq * 4;            //                (MFMA lanes: BN scale / shift)
    const float* Ug = p.U + ((size_t)g * 16 * p.cout_pad + n0) * p.cin_g;     // + (xi * cout_pad + n) * cin_g + ci

    // Per item a loading lane keeps just the byte offset of its tile's patch origin and a 16-bit validity mask; out-of-image
    // positions are fetched from offset 0 (always mapped) and replaced by the padding value before the transform.
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

    f32x4 acc[16][NBT];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb) acc[xi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ssum[NBT], ssq[NBT];              // per-lane batch sums of channels n0 + nb*16 + 4*kq + j over the lane's tiles
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) ssum[nb] = ssq[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 16-byte bias / residual / output accesses (every channel quad starts on a 16-byte boundary)
    const bool vec = ((p.out_stride | p.out_ch_off | p.cout_g) & 3) == 0 && p.vec_ok;

    const int nchunks = p.cin_g >> 4;
    f32x4 raw[16];
    auto load_raw1 = [&](unsigned pix_off, unsigned valid, int c, int q) {
        const int i = q >> 2, j = q & 3;
        const unsigned off = (valid >> q) & 1 ? pix_off + (unsigned)(((i * p.W + j) * p.in_stride + c * 16) * 4)
                                              : (p.pad_off ? p.pad_off + (unsigned)((cb_ld + c * 16) * 4) : 0u);
        raw[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.in) + off);
    };
    // U stage: 16 * NBT pieces of 1 KB; piece (xi, nb) = 16 rows (n) x 64 B; lane -> row lane >> 2, quad lane & 3.  Wave w
    // moves pieces w, w + 4, ...: slot s (0 .. 4*NBT-1) -> piece 4*s + w.
    const unsigned u_lane = (unsigned)((((lane >> 2) * p.cin_g) + (((lane & 3) ^ ((lane >> 4) & 1)) << 2)) * 4);
    auto stage_U1 = [&](int c, int buf, int s4) {
        const int pc = 4 * s4 + wv;
        const int xi = pc / NBT, nb = pc - xi * NBT;
        // quads of row n sit at position quad ^ ((n >> 2) & 1): the eight lanes a ds_read_b128 serves per cycle (rows r..r+7,
        // 64 B apart) then cover all 64 banks instead of hitting 32 of them twice
        // (uniform base + one 32-bit lane offset: no 64-bit VGPR address arithmetic per piece)
        const unsigned off = u_lane + (unsigned)((((xi * p.cout_pad + nb * 16) * p.cin_g) + c * 16) * 4);
        dma16(reinterpret_cast<const float*>(reinterpret_cast<const char*>(Ug) + off), smem + buf * STAGE + pc * 256);
    };

#ifdef WINO_TIMING
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    unsigned in_cur, valid_cur;
    decode(item, in_cur, valid_cur);
#pragma unroll
    for (int s4 = 0; s4 < 4 * NBT; ++s4) stage_U1(0, 0, s4);
#pragma unroll
    for (int q = 0; q < 16; ++q) load_raw1(in_cur, valid_cur, 0, q);
    int buf = 0;

    // The workgroup is persistent: it walks the items of its range of its (group, output-channel block).  The
    // stream of (item, chunk) steps is software pipelined as one sequence -- the loads of the NEXT step (next chunk, or the
    // next item's first chunk) are issued piecewise between the MFMA groups of the current one -- so the load latency at an
    // item boundary hides behind the last chunk's MFMAs and the epilogue's stores overlap the next item's first chunk.
    for (;;) {
        const int item_next = item + 1;
        const bool have_next = PERSIST && item_next < item_end;
        unsigned in_next = in_cur, valid_next = 0;
        if (have_next) decode(item_next, in_next, valid_next);
        for (int c = 0; c < nchunks; ++c) {
            // ---- padding select + input transform of chunk c (registers) -----------------------------------------------------
#ifdef WINO_TIMING
            WSTAMP(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WSTAMP(1)
#endif
            f32x4 V[16];
            {
                f32x4 sc, sh, padq = f32x4{0.f, 0.f, 0.f, 0.f};
                if (XF) {
                    sc = *reinterpret_cast<const f32x4*>(p.in_scale + cb_mf + c * 16);
                    sh = *reinterpret_cast<const f32x4*>(p.in_shift + cb_mf + c * 16);
                    padq = *reinterpret_cast<const f32x4*>(p.in_pad + cb_ld + c * 16);
                }
                // padding select in place, then the arithmetic one channel at a time (32 transient registers).  Round 3 measured the
                // packed form (v_pk_add_f32 on channel
                // pairs): 29.7k instead of 24.9k cycles per 8 chunks -- packed fp32 is no gain beside fp32 MFMAs, nor is dealing this
                // work out between the MFMAs (the fp32 MFMA shares the vector ALU's fp32 lanes: a just-in-time transform inside the MFMA
                // loop ran 13.1k cycles per chunk against 9.3k + 3.1k here)
                if (!p.pad_off) {                              // (64 selects of 8.4 cycles each beside fp32 MFMAs: skipped when the loads did it)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) raw[q][e] = (valid_cur >> q) & 1 ? raw[q][e] : padq[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d[16], t[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) d[q] = XF ? fmaxf(raw[q][e] * sc[e] + sh[e], 0.f) : raw[q][e];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {         // B^T d
                        t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
                        t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
                        t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
                        t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {         // (B^T d) B
                        V[i * 4 + 0][e] = t[i * 4 + 0] - t[i * 4 + 2];
                        V[i * 4 + 1][e] = t[i * 4 + 1] + t[i * 4 + 2];
                        V[i * 4 + 2][e] = t[i * 4 + 2] - t[i * 4 + 1];
                        V[i * 4 + 3][e] = t[i * 4 + 1] - t[i * 4 + 3];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);            // raw[] is dead from here on: the next step's loads reuse its registers
#ifdef WINO_TIMING
            asm volatile("s_nop 0" ::"v"(V[0][0]), "v"(V[15][3]));
            WSTAMP(2)
#endif
            __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0): this wave's DMA pieces of the current stage have landed
            __syncthreads();                              // ... everyone's have; the other buffer is free again
            WSTAMP(3)
            const bool last = c + 1 == nchunks;
            const bool more = !last || have_next;
            const unsigned ld_in = last ? in_next : in_cur;
            const unsigned ld_valid = last ? valid_next : valid_cur;
            const int ld_c = last ? 0 : c + 1;
            // ---- 16 GEMM slices: acc[xi][nb] += V_xi (16 tiles x 4 k) * U_xi (4 k x 16 co), four k-steps per 16-byte fragment --
            const float* ub = smem + buf * STAGE + r * 16 + ((kq ^ ((r >> 2) & 1)) << 2);
            f32x4 bf[2][NBT];
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb) bf[0][nb] = *reinterpret_cast<const f32x4*>(ub + nb * 256);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                if (xi + 1 < 16) {
#pragma unroll
                    for (int nb = 0; nb < NBT; ++nb)
                        bf[(xi + 1) & 1][nb] = *reinterpret_cast<const f32x4*>(ub + ((xi + 1) * NBT + nb) * 256);
                }
                // the next step's traffic is issued a piece at a time between the MFMA groups: with one wave per SIMD a burst
                // of 32 memory instructions in front of the MFMAs would stall the wave on the memory pipe's issue queue
                if (more) {
                    load_raw1(ld_in, ld_valid, ld_c, xi);
#pragma unroll
                    for (int s4 = xi * NBT / 4; s4 < (xi + 1) * NBT / 4; ++s4) stage_U1(ld_c, buf ^ 1, s4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nb = 0; nb < NBT; ++nb)
                        acc[xi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[xi & 1][nb][j], V[xi][j], acc[xi][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
#ifdef WINO_TIMING
            asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[15][NBT - 1][3]));      // the MFMAs have retired
#endif
            WSTAMP(4)
        }

        // ---- output transform + epilogue: lane (r, kq) holds M[co n0 + nb*16 + 4*kq + j][tile r] for all 16 xi (the MFMAs run with
        // U as the A operand: four CONSECUTIVE output channels of ONE tile per accumulator quad -> one tile decode per lane, 16-byte
        // bias / residual loads and stores; round 3: 64 four-byte stores and four tile decodes per lane before) ----------------------
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
                    const int nrem = p.cout_g - (n0 + nb * 16 + kq * 4);      // channels of this quad that exist (padded rows of U beyond)
                    if (nrem <= 0) continue;
                    f32x4 s0[4], s1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {             // A^T M, four channels at a time
                        s0[j] = acc[0 * 4 + j][nb] + acc[1 * 4 + j][nb] + acc[2 * 4 + j][nb];
                        s1[j] = acc[1 * 4 + j][nb] - acc[2 * 4 + j][nb] - acc[3 * 4 + j][nb];
                    }
                    f32x4 bia = f32x4{0.f, 0.f, 0.f, 0.f};
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
                        // GSSD_CONV_POOL2: a Winograd tile IS a pooling window (2 x 2 outputs at even coordinates): the lane reduces its
                        // four outputs (those that exist, ceil mode) and stores one value at the pooled position (ty, tx).  The batch
                        // sums take the outputs in the order the unpooled epilogue adds them: identical statistics.
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
                    // keep the accumulator read-out in small groups: the next item's prefetched patch is live here, so hoisting
                    // all reads out of the AGPR file at once would not fit the 256 architectural VGPRs
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        WSTAMP(5)
        if (!have_next) break;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb) acc[xi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        item = item_next;
        in_cur = in_next;
        valid_cur = valid_next;
    }

    if (p.stats) {                                        // one flush per workgroup: the 16 tile lanes of a kq -> LDS over waves -> fp64 atomics
        __syncthreads();                                  // all waves are done with the U stages
        float* red = smem;                                // [4 waves][NB][2]
        // the 4 * NBT sums of a lane are folded over its 16 tile lanes by a halving exchange (lane keeps the half its r bit selects and
        // adds the partner's): 4 * NBT - 1 lane exchanges per quantity instead of 4 per value; lane r ends with value index r
        // (NBT = 4: channel n0 + 16 * (r >> 2) + 4 * kq + (r & 3); NBT = 2: lanes r and r ^ 8 hold the same value)
        float fs[16], fq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            fs[i] = ssum[(i >> 2) % NBT][i & 3];
            fq[i] = ssq[(i >> 2) % NBT][i & 3];
        }
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1) {
            if (4 * NBT <= w) continue;                   // (NBT = 2: only 8 values, the top exchange is skipped)
            const bool up = (r & w) != 0;
#pragma unroll
            for (int i = 0; i < w; ++i) {
                const float ks = up ? fs[i + w] : fs[i], gs = up ? fs[i] : fs[i + w];
                const float kq2 = up ? fq[i + w] : fq[i], gq = up ? fq[i] : fq[i + w];
                fs[i] = ks + __shfl_xor(gs, w, 64);
                fq[i] = kq2 + __shfl_xor(gq, w, 64);
            }
        }
        if (4 * NBT == 8) {                               // fold the two tile-lane halves that both hold value index r & 7
            fs[0] += __shfl_xor(fs[0], 8, 64);
            fq[0] += __shfl_xor(fq[0], 8, 64);
        }
        {
            const int vi = r & (4 * NBT - 1);             // value index held by this lane
            if (r < 4 * NBT) {
                red[(wv * NB + (vi >> 2) * 16 + kq * 4 + (vi & 3)) * 2 + 0] = fs[0];
                red[(wv * NB + (vi >> 2) * 16 + kq * 4 + (vi & 3)) * 2 + 1] = fq[0];
            }
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
#ifdef WINO_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WSTAMP(6)
    if (tid == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&g_wino_timing[k], tacc[k]);
        atomicAdd(&g_wino_timing[7], 1ull);
    }
#endif
}

// packed K-major weights [Cout][tap * cin_g + ci] (row stride `ws`) -> U[g][xi][co][ci] = (G g G^T)_xi; rows cout_g..cout_pad-1
// of every group are zero
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int groups, int cout_g, int cout_pad,
                                   int cin_g, int ws) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * cout_pad * cin_g) return;
    const int ci = i % cin_g, cop = i / cin_g;
    const int g = cop / cout_pad, cg = cop - g * cout_pad;
    float k[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) k[a][b] = cg < cout_g ? w[(size_t)(g * cout_g + cg) * ws + (a * 3 + b) * cin_g + ci] : 0.f;
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {                         // G g
        t[0][b] = k[0][b];
        t[1][b] = 0.5f * (k[0][b] + k[1][b] + k[2][b]);
        t[2][b] = 0.5f * (k[0][b] - k[1][b] + k[2][b]);
        t[3][b] = k[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {                         // (G g) G^T
        const float u[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) U[(((size_t)g * 16 + a * 4 + b) * cout_pad + cg) * cin_g + ci] = u[b];
    }
}

// output-channel block of the kernel for a layer (0 = not a Winograd shape) and the padded per-group row count of U
inline int wino_nb(int cout_g, int groups) {
    if (cout_g % 64 == 0) return 64;
    if (cout_g % 32 == 0) return 32;
    if (groups == 1 && cout_g >= 24) return cout_g > 32 ? 64 : 32;     // e.g. the DCN offset / mask conv (108 channels)
    return 0;
}
inline int wino_cout_pad(int cout_g, int nb) { return (cout_g + nb - 1) / nb * nb; }
// rows per group of U for a layer: the kernel block's padding, or (16-channel groups: conv_thin_wino.hip) none; 0 = no U
inline int wino_u_rows(int cout_g, int groups) {
    const int nb = wino_nb(cout_g, groups);
    if (nb) return wino_cout_pad(cout_g, nb);
    return cout_g % 16 == 0 ? cout_g : 0;
}

template <int NB, bool XF, bool PERSIST, int EPI>
int launch_wino(const gssd_conv_desc& d, hipStream_t stream) {
    WinoParams p;
    p.in = d.in;
    p.U = d.wgt_wino;
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
    p.cout_pad = wino_cout_pad(p.cout_g, NB);
    p.out_stride = d.out_stride;
    p.out_ch_off = d.out_ch_off;
    p.tiles_y = (d.H + 1) / 2;
    p.tiles_x = (d.W + 1) / 2;
    p.ntiles = d.B * p.tiles_y * p.tiles_x;
    constexpr size_t smem = 2 * (size_t)16 * NB * 16 * sizeof(float);
    p.vec_ok = (((uintptr_t)d.out | (uintptr_t)d.bias | (uintptr_t)d.resid | (uintptr_t)p.pool_sign) & 15) == 0;
    p.pad_off = 0;
    if (XF && d.in_pad && (uintptr_t)d.in_pad > (uintptr_t)d.in) {
        const unsigned long long diff = (unsigned long long)((uintptr_t)d.in_pad - (uintptr_t)d.in);
        // only the layout the engine builds: the vector directly behind the dense map (so the 32-bit offsets of the kernel reach it)
        if (diff == (unsigned long long)d.B * d.H * d.W * d.in_stride * sizeof(float) && diff + (unsigned long long)d.in_stride * 4 < (1ull << 32) &&
            (diff & 15) == 0)
            p.pad_off = (unsigned)diff;
    }
    auto kern = conv_wino_kernel<NB, XF, PERSIST, EPI>;
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
            hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (winograd)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const int nitems = (p.ntiles + 63) / 64;
    static int per_cu = 0;                                // resident workgroups per CU (registers / LDS decide: 1 or 2)
    if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, smem) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 2) per_cu = 2;
    }
    int gx = 256 * per_cu / ((p.cout_pad / NB) * d.groups);
    if (gx < 1) gx = 1;
    if (gx > nitems || !PERSIST) gx = nitems;
    const dim3 grid = PERSIST ? dim3(gx, p.cout_pad / NB, d.groups)
                              : dim3(((nitems + 7) / 8) * 8 * (p.cout_pad / NB) * d.groups, 1, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

// returns 1 when the descriptor is not a Winograd shape (or carries no transformed weights)
int gssd_try_conv_wino(const gssd_conv_desc& d, hipStream_t stream) {
    if (!d.wgt_wino) return 1;
    const int cout_g = d.Cout / d.groups;
    // (GSSD_OUT_HEADS: only the form conv_wino_x6.hip takes -- one reduction slice, no batch sums / residual / pooling, both sides of split_n whole quads)
    const bool heads = d.out_mode == GSSD_OUT_HEADS && d.out_b && d.split_k <= 1 && !d.stats && !d.resid && !(d.flags & GSSD_CONV_POOL2) &&
                       d.split_n % 4 == 0 && (d.Cout - d.split_n) % 4 == 0 && gssd_wino_x6_plane_elems(cout_g, d.groups, d.cin_g) > 0 && gssd_wino_x6_wanted(d);
    const bool ok = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.dil == 1 && d.cin_g % 16 == 0 && wino_nb(cout_g, d.groups) != 0 &&
                    (d.out_mode == GSSD_OUT_NHWC || heads) && !d.alpha && !d.gate && !d.out2 && !d.relu && d.split_k <= 1 && !d.m_per_image &&
                    d.in_stride % 4 == 0 && d.in_ch_off % 4 == 0 && ((uintptr_t)d.wgt_wino % 16) == 0 &&
                    (long long)d.B * d.H * d.W * d.in_stride < (1ll << 30);     // 32-bit BYTE offsets into the input
    if (!ok) return 1;
    if ((d.flags & GSSD_CONV_POOL2) && (d.resid || !d.pool_sign)) return 1;
    // the same transforms with the 16 GEMMs on the bf16 matrix cores (three-plane operands, conv_wino_x6.hip): its U planes lie behind the fp32 U
    if (gssd_wino_x6_plane_elems(cout_g, d.groups, d.cin_g) > 0 && gssd_wino_x6_wanted(d))
        return gssd_launch_conv_wino_x6(d, d.wgt_wino + 16ll * d.groups * wino_u_rows(cout_g, d.groups) * d.cin_g, stream);
    // NB = 64 holds 256 accumulators per lane and has no registers left for a prefetched patch across the epilogue: one item
    // per workgroup there; the NB = 32 variant (conv2_2: two chunks per item) runs persistent.  (Round 2: the persistent NB = 32
    // variant forced onto the 64 / 128-channel layers measures 15-30 % slower -- conv3_2 489 vs 415 us, conv4_2 436 vs 336 us: it
    // transforms every input tile once per 32-channel block.)
    const int epi = (d.flags & GSSD_CONV_POOL2) ? 2 : d.resid ? 1 : 0;
#define WINO_GO(NB_, P_)                                                                                              \
    (d.in_scale ? (epi == 2 ? launch_wino<NB_, true, P_, 2>(d, stream)                                               \
                            : epi == 1 ? launch_wino<NB_, true, P_, 1>(d, stream) : launch_wino<NB_, true, P_, 0>(d, stream))   \
                : (epi == 2 ? launch_wino<NB_, false, P_, 2>(d, stream)                                              \
                            : epi == 1 ? launch_wino<NB_, false, P_, 1>(d, stream) : launch_wino<NB_, false, P_, 0>(d, stream)))
    if (wino_nb(cout_g, d.groups) == 64) return WINO_GO(64, false);
    return WINO_GO(32, true);
#undef WINO_GO
}

extern "C" long long gssd_winograd_weight_elems(int Cout, int groups, int cin_g) {
    if (Cout <= 0 || groups <= 0 || Cout % groups || cin_g <= 0) return -1;
    const int rows = wino_u_rows(Cout / groups, groups);
    if (!rows) return -1;
    // fp32 U, then (shapes conv_wino_x6.hip takes) its three bf16 planes in that kernel's staging order
    return 16ll * groups * rows * cin_g + (wino_nb(Cout / groups, groups) ? gssd_wino_x6_plane_elems(Cout / groups, groups, cin_g) / 2 : 0);
}

extern "C" int gssd_conv_wino_x6_takes(const gssd_conv_desc* d) {
    if (!d || !d->wgt_wino || !gssd_wino_x6_enabled()) return 0;
    const int cout_g = d->Cout / d->groups;
    const bool heads = d->out_mode == GSSD_OUT_HEADS && d->split_k <= 1      // (out / out_b may still be unset when a plan names its kernels)
                       && !d->stats && !d->resid && !(d->flags & GSSD_CONV_POOL2) &&
                       d->split_n % 4 == 0 && (d->Cout - d->split_n) % 4 == 0;
    const bool ok = d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 && d->cin_g % 16 == 0 && wino_nb(cout_g, d->groups) != 0 &&
                    (d->out_mode == GSSD_OUT_NHWC || heads) && !d->alpha && !d->gate && !d->out2 && !d->relu && d->split_k <= 1 && !d->m_per_image &&
                    d->in_stride % 4 == 0 && d->in_ch_off % 4 == 0 && (long long)d->B * d->H * d->W * d->in_stride < (1ll << 30);
    if (!ok || ((d->flags & GSSD_CONV_POOL2) && (d->resid || !d->pool_sign))) return 0;
    return gssd_wino_x6_plane_elems(cout_g, d->groups, d->cin_g) > 0 && gssd_wino_x6_wanted(*d);
}

extern "C" int gssd_winograd_weight_f32(const float* w_packed, float* U, int Cout, int groups, int cin_g, int row_stride,
                                        gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_packed && U && Cout > 0 && groups > 0 && Cout % groups == 0 && cin_g > 0 && row_stride >= 9 * cin_g);
    const int cout_g = Cout / groups, cout_pad = wino_u_rows(cout_g, groups);
    GSSD_CHECK_ARG(cout_pad != 0);
    const int n = groups * cout_pad * cin_g;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), w_packed, U, groups, cout_g,
                       cout_pad, cin_g, row_stride);
    GSSD_CHECK_LAUNCH();
    if (wino_nb(cout_g, groups) && gssd_wino_x6_plane_elems(cout_g, groups, cin_g) > 0)
        return gssd_wino_x6_pack(w_packed, U + 16ll * groups * cout_pad * cin_g, Cout, groups, cin_g, row_stride, as_stream(stream));
    return GSSD_OK;
}
