// Weight gradient of the grouped 3x3 trunk convolutions on the bf16 matrix cores (gfx950), for the training step of the bf16 storage
// mode (BASELINE.json configs[4] as a training config; gssd/backward.py).  dW[co][tap][ci] = sum over pixels dY[px][co] * X[px + tap][ci]
// is a TN product: the reduction index (the pixel) is the SLOW index of both operands as they sit in memory (NHWC), while
// v_mfma_f32_16x16x32_bf16 wants 8 consecutive reduction elements per lane.  gfx950's LDS transpose read closes that gap without a
// transposed copy: both operands are staged in their natural [pixel][channel] layout (LDS-DMA, 16 bytes per lane) and every fragment is
// two ds_read_b64_tr_b16 -- within a 16-lane group the lanes address a [4 pixel][16 channel] block (4 lanes x 8 bytes per pixel row) and
// each lane receives one channel's 4 pixels.
//
// Structure (the bf16 twin of conv_patch_wgrad.hip): a persistent workgroup owns NG phase groups x NCOW output channels, walks 8 x 16
// output tiles, stages the tile's input patch (10 x 18 pixels with the halo) and its dY tile once, and the nine taps are nine shifted
// reads of the patch; the whole gradient block lives in MFMA accumulators for the lifetime of the workgroup and is flushed with fp32
// atomics once.  The k index of an MFMA (32 pixels = two tile rows) maps to pixels so that the 32 lanes the LDS serves per cycle touch 8
// CONSECUTIVE pixels: k = 8*kq + 4*t + j  <->  tile row 2*s + (kq >> 1), column 8*t + 4*(kq & 1) + j   (kq = lane >> 4, t = which of the
// two reads, j = element); the 32-byte channel chunks of a pixel are XOR-swizzled by the pixel's column so that those 8 pixels land on
// all 64 banks whatever the channel count.  A deferred BatchNorm + ReLU on the input (gssd_conv_desc::in_scale) is applied ONCE per
// staged element in LDS (fp32 math, rounded to bf16 like the forward's staging does), not per fragment: beside bf16 MFMAs 24 VALU
// instructions per fragment would cost more than the MFMAs they feed.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

namespace {

constexpr int SCHED_EVERY = 4;         // entries between two scheduling barriers of the k-step (bounds the fragment reads in flight)

__device__ __attribute__((aligned(16))) u16 g_zero_page_wb[8] = {0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ s16x4 tr_read(const u16* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_ptr);
}

struct WgBfParams {
    const u16* in;
    const u16* dy;
    float* dw;               // packed [Cout][9*cin_g] fp32, zero-filled by the caller
    const float* in_scale;
    const float* in_shift;
    int B, H, W, in_stride, in_ch_off, Cout, cout_g, co_splits, tiles_y, tiles_x;
    // A group's cin_g input channels are walked as bpg blocks of CIN_G channels (bpg = 1: the trunk's grouped convs; groups = 1: a dense
    // conv as cin / CIN_G blocks feeding the same outputs).  Block blk = group * bpg + bi reads input channels [blk * CIN_G, + CIN_G),
    // the group's cout_g output channels, and owns columns tap * cin_g + bi * CIN_G + .. of the packed rows (k_row = taps * cin_g).
    int bpg, cin_g, k_row;
};

// swizzle of the 32-byte chunk index by the pixel column: NC chunks per pixel
template <int NC>
__device__ __forceinline__ int swz(int col) {
    if constexpr (NC >= 8) return col & 7;
    else if constexpr (NC == 4) return (col >> 1) & 3;
    else if constexpr (NC == 2) return (col >> 2) & 1;
    else return 0;
}

// CIN_G: input channels per group; NG: groups per workgroup (one wave each when NG == 4); a wave owns COB 16-channel output blocks and
// EPW of the 9 * CIN_G / 16 (tap, 16-input-channel) entries; WCO x WEN waves per group (NG * WCO * WEN == 4).
template <int CIN_G, int NG, int COB, int WCO, int WEN, bool XF, int NTAP>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const WgBfParams p) {
    static_assert(NTAP == 9 || NTAP == 1, "3x3 (stride 1, pad 1) or 1x1");
    static_assert(NG * WCO * WEN == 4, "four waves");
    constexpr int HALO = NTAP == 9 ? 1 : 0, NDX = NTAP == 9 ? 3 : 1;
    constexpr int TH = 8, TW = 16, PW = TW + 2 * HALO, NPATCH = (TH + 2 * HALO) * PW, NPIX = TH * TW;
    constexpr int NCI = CIN_G / 16, CPW = NCI / WEN, EPW = NTAP * CPW;  // a wave's entries: the taps x CPW of the NCI 16-channel input chunks
    static_assert(NCI % WEN == 0 && (CPW & (CPW - 1)) == 0 && (COB & (COB - 1)) == 0, "chunk split");
    constexpr int NCOW = 16 * COB * WCO;                    // output channels per group and workgroup
    constexpr int NC = NG * NCI;                            // 32-byte chunks per patch pixel
    constexpr int NCD = NG * COB * WCO;                     // 32-byte chunks per dY pixel
    constexpr int UP = 2 * NC, PPI = 64 / UP, NPI = (NPATCH + PPI - 1) / PPI;     // 16-byte units per pixel, pixels per DMA piece
    constexpr int UD = 2 * NCD, PDY = 64 / UD, NDY = NPIX / PDY;
    constexpr int PATCH_E = NPI * PPI * NC * 16;            // elements
    static_assert(UP <= 64 && UD <= 64 && NC <= 8 && NCD <= 8, "chunk counts");
    static_assert((NPI * PPI * NC) % NCD == 0, "the dY image starts on a multiple of its pixel stride");
    extern __shared__ __attribute__((aligned(16))) u16 smem_w[];
    u16* patch = smem_w;
    u16* dyt = smem_w + PATCH_E;
    float* xf_s = reinterpret_cast<float*>(dyt + NPIX * NCD * 16);      // [2][NG * CIN_G] scale, shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int gl = wv / (WCO * WEN);                        // group inside the workgroup
    const int wc = (wv / WEN) % WCO, we = wv % WEN;
    const int gq = blockIdx.y / p.co_splits, cs = blockIdx.y - gq * p.co_splits;
    const int g0 = gq * NG;                                 // first group of the workgroup
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int ntiles = p.B * tiles_per_img;
    const u16* zero = g_zero_page_wb;
    const u16* in_g = p.in + p.in_ch_off + g0 * CIN_G;
    // dY channels of the workgroup: groups g0 .. g0 + NG - 1, channels [cs * NCOW, + NCOW) of each (NG > 1 only with co_splits == 1 and
    // NCOW == cout_g: one contiguous range)
    const int grp0 = g0 / p.bpg;
    const u16* dy_g = p.dy + grp0 * p.cout_g + cs * NCOW;

    if constexpr (XF) {
        for (int i = tid; i < NG * CIN_G; i += 256) {
            xf_s[i] = p.in_scale[p.in_ch_off + g0 * CIN_G + i];
            xf_s[NG * CIN_G + i] = p.in_shift[p.in_ch_off + g0 * CIN_G + i];
        }
    }

    // per-lane read geometry: supplier row j = (lane & 15) >> 2, 8-byte piece q = lane & 3 of a 32-byte chunk
    const int sj = (lane & 15) >> 2, sq = lane & 3;
    // A (dY) : pixel column of read t = 8 t + 4 (kq & 1) + sj, tile row 2 s + (kq >> 1)
    // (element offsets with the swizzle of chunk 0 folded in: chunk c of the same pixel is at offset ^ (c << 4), the pixel bases being
    // multiples of the pixel stride)
    int a_addr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = 8 * t + 4 * (kq & 1) + sj;
        a_addr[t] = ((((kq >> 1) * TW + col) * NCD) * 16 + (swz<NCD>(col) << 4) + sq * 4) ^ (((gl * WCO + wc) * COB) << 4);
    }
    // B (patch): column 8 t + 4 (kq & 1) + sj + dx, row 2 s + (kq >> 1) + dy
    int b_addr[NDX][2];
#pragma unroll
    for (int dx = 0; dx < NDX; ++dx)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = 8 * t + 4 * (kq & 1) + sj + dx;
            b_addr[dx][t] = ((((kq >> 1) * PW + col) * NC) * 16 + (swz<NC>(col) << 4) + sq * 4) ^ ((gl * NCI + we * CPW) << 4);
        }

    f32x4 acc[COB][EPW];
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int t = 0; t < EPW; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
        const int y0 = tyi * TH, x0 = txi * TW;
        // ---- stage the input patch and the dY tile (natural [pixel][channel] images, 32-byte chunks swizzled by the pixel column) ----
        for (int i = wv; i < NPI; i += 4) {
            const int pp = i * PPI + lane / UP;
            const int py = pp / PW, pxx = pp - py * PW;
            const int u = lane % UP;
            const int ch = (((u >> 1) ^ swz<NC>(pxx)) << 4) + ((u & 1) << 3);
            const int iy = y0 - HALO + py, ix = x0 - HALO + pxx;
            const bool ok = pp < NPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const u16* src = ok ? in_g + ((size_t)(b * p.H + iy) * p.W + ix) * p.in_stride + ch : zero;
            dma16(src, patch + i * PPI * NC * 16);
        }
        for (int i = wv; i < NDY; i += 4) {
            const int px = i * PDY + lane / UD;
            const int u = lane % UD;
            const int chl = (((u >> 1) ^ swz<NCD>(px & 15)) << 4) + ((u & 1) << 3);      // channel inside the workgroup's NG * NCOW
            const int y = y0 + (px >> 4), x = x0 + (px & 15);
            const bool ok = y < p.H && x < p.W && (cs * NCOW + (NG > 1 ? 0 : chl)) < p.cout_g;
            const u16* src = ok ? dy_g + ((size_t)(b * p.H + y) * p.W + x) * p.Cout + chl : zero;
            dma16(src, dyt + i * PDY * NCD * 16);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if constexpr (XF) {
            // deferred BatchNorm + ReLU of the producer, once per staged element; out-of-image pixels stay zero
            for (int i = tid; i < NPATCH * UP; i += 256) {
                const int pp = i / UP, u = i - pp * UP;
                const int py = pp / PW, pxx = pp - py * PW;
                const int iy = y0 - HALO + py, ix = x0 - HALO + pxx;
                if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
                const int ch = (((u >> 1) ^ swz<NC>(pxx)) << 4) + ((u & 1) << 3);
                uint4* q = reinterpret_cast<uint4*>(patch + pp * NC * 16 + u * 8);
                uint4 v = *q;
                unsigned w[4] = {v.x, v.y, v.z, v.w};
                const float* sc = xf_s + ch;
                const float* sh = xf_s + NG * CIN_G + ch;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = fmaxf(__uint_as_float(w[e] << 16) * sc[2 * e] + sh[2 * e], 0.f);
                    const float hi = fmaxf(__uint_as_float(w[e] & 0xffff0000u) * sc[2 * e + 1] + sh[2 * e + 1], 0.f);
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    bf16x2 h;
                    h[0] = (__bf16)lo;
                    h[1] = (__bf16)hi;
                    w[e] = __builtin_bit_cast(unsigned, h);
                }
                *q = uint4{w[0], w[1], w[2], w[3]};
            }
            __syncthreads();
        }
        // ---- 4 k-steps of 32 pixels (two tile rows each).  Every wave runs the same instruction stream: what differs between waves
        // (group, output blocks, input chunks) sits in the address registers ---------------------------------------------------------
#pragma unroll
        for (int s = 0; s < NPIX / 32; ++s) {
            // the per-entry addresses are one v_xor each: keep the compiler from hoisting all of them out of the tile loop (LICM turned
            // 72 of them into live registers next to 144 accumulators: spills)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                asm volatile("" : "+v"(a_addr[t]));
#pragma unroll
                for (int dx = 0; dx < NDX; ++dx) asm volatile("" : "+v"(b_addr[dx][t]));
            }
            bf16x8 A[COB];
#pragma unroll
            for (int c = 0; c < COB; ++c) {
                const s16x4 lo = tr_read(dyt + (a_addr[0] ^ (c << 4)) + 2 * s * TW * NCD * 16);
                const s16x4 hi = tr_read(dyt + (a_addr[1] ^ (c << 4)) + 2 * s * TW * NCD * 16);
                A[c] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int t = 0; t < EPW; ++t) {
                const int tap = t / CPW, cti = t - tap * CPW;
                const int dyy = tap / 3, dxx = tap - 3 * dyy;
                const int rowoff = (2 * s + dyy) * PW * NC * 16;
                const s16x4 lo = tr_read(patch + (b_addr[dxx][0] ^ (cti << 4)) + rowoff);
                const s16x4 hi = tr_read(patch + (b_addr[dxx][1] ^ (cti << 4)) + rowoff);
                const bf16x8 Bv = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int c = 0; c < COB; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[c], Bv, acc[c][t], 0, 0, 0);
                // bound the scheduler's look-ahead: it hoists every fragment read of a k-step otherwise (72+ live registers)
                if (t % SCHED_EVERY == SCHED_EVERY - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- flush: D rows = co (kq*4 + e), columns = ci r ------------------------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < COB; ++c) {
        const int col0 = cs * NCOW + (wc * COB + c) * 16 + kq * 4;          // channel inside the group
#pragma unroll
        for (int t = 0; t < EPW; ++t) {
            const int tap = t / CPW, ct = we * CPW + (t - tap * CPW);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (col0 + e >= p.cout_g) continue;
                const int blk = g0 + gl, grp = blk / p.bpg, bi = blk - grp * p.bpg;
                const int co = grp * p.cout_g + col0 + e;
                unsafeAtomicAdd(p.dw + (size_t)co * p.k_row + tap * p.cin_g + bi * CIN_G + ct * 16 + r, acc[c][t][e]);
            }
        }
    }
}

template <int CIN_G, int NG, int COB, int WCO, int WEN, bool XF, int NTAP>
int launch_wgrad_bf16(const gssd_conv_desc& d, const void* dy, float* dw, hipStream_t stream) {
    constexpr int NCI = CIN_G / 16, NC = NG * NCI, NCD = NG * COB * WCO, NCOW = 16 * COB * WCO;
    constexpr int NPATCH = NTAP == 9 ? 180 : 128;
    constexpr int UP = 2 * NC, PPI = 64 / UP, NPI = (NPATCH + PPI - 1) / PPI;
    WgBfParams p;
    p.in = reinterpret_cast<const u16*>(d.in);
    p.dy = reinterpret_cast<const u16*>(dy);
    p.dw = dw;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    p.in_stride = d.in_stride;
    p.in_ch_off = d.in_ch_off;
    p.Cout = d.Cout;
    p.bpg = d.cin_g / CIN_G;
    const int nblk = d.groups * p.bpg;
    p.cout_g = d.Cout / d.groups;
    p.cin_g = d.cin_g;
    p.k_row = NTAP * d.cin_g;
    p.co_splits = (p.cout_g + NCOW - 1) / NCOW;
    p.tiles_y = (d.H + 7) / 8;
    p.tiles_x = (d.W + 15) / 16;
    const size_t smem = ((size_t)NPI * PPI * NC * 16 + 128 * NCD * 16) * sizeof(u16) + 2 * NG * CIN_G * sizeof(float);
    auto kern = conv_wgrad_bf16_kernel<CIN_G, NG, COB, WCO, WEN, XF, NTAP>;
    static unsigned attr_mask = 0;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) !=
            hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (bf16 wgrad)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    const long long ntiles = (long long)d.B * p.tiles_y * p.tiles_x;
    const int gy = nblk / NG * p.co_splits;
    int gx = 512 / gy;                                    // two resident workgroups per CU
    if (gx < 1) gx = 1;
    if (ntiles < gx) gx = (int)ntiles;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

bool shape_ok(const gssd_conv_desc& d) {
    const bool k3 = d.KH == 3 && d.KW == 3 && d.pad == 1, k1 = d.KH == 1 && d.KW == 1 && d.pad == 0;
    if (!((k3 || k1) && d.stride == 1 && d.dil == 1 && !d.m_per_image && d.groups > 0)) return false;
    if (d.Cout % d.groups || d.in_stride % 8 || d.in_ch_off % 8 || d.Cout % 8 || (d.Cout / d.groups) % 8) return false;
    if ((long long)d.B * d.H * d.W * d.in_stride >= (1ll << 31) || (long long)d.B * d.H * d.W * d.Cout >= (1ll << 31)) return false;
    const int cg = d.cin_g, ng = d.Cout / d.groups;
    if (cg > 128) return cg % 128 == 0;                       // 128-channel input blocks of a group (dense convs: one group)
    if (k1) return cg == 128 || cg == 64;
    if (d.groups % 4 == 0 && ((cg == 16 && (ng == 16 || ng == 32)) || (cg == 32 && ng == 32))) return true;
    if (cg == 32 || cg == 64) return ng % 64 == 0;
    return cg == 128;
}

}  // namespace

extern "C" int gssd_conv2d_wgrad_bf16_supported(const gssd_conv_desc* d) { return d && shape_ok(*d) ? 1 : 0; }

extern "C" int gssd_conv2d_wgrad_bf16(const gssd_conv_desc* dp, const void* dy, float* dw_packed, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dp && dy && dw_packed);
    const gssd_conv_desc& d = *dp;
    GSSD_CHECK_ARG(d.in && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)d.in % 16) == 0 && ((uintptr_t)dw_packed % 16) == 0);
    GSSD_CHECK_ARG((d.in_scale == nullptr) == (d.in_shift == nullptr));
    if (!shape_ok(d)) {
        gssd_set_error("gssd_conv2d_wgrad_bf16: not a shape of the bf16 patch-staged weight gradient (3x3, stride 1, pad 1; %d -> %d "
                       "channels per group, %d groups)", d.cin_g, d.Cout / d.groups, d.groups);
        return GSSD_EINVAL;
    }
    const int cg = d.cin_g, ng = d.Cout / d.groups;
    hipStream_t s = as_stream(stream);
#define GSSD_WB1(CI, NG_, COB_, WCO_, WEN_, NT)                                                                  \
    return d.in_scale ? launch_wgrad_bf16<CI, NG_, COB_, WCO_, WEN_, true, NT>(d, dy, dw_packed, s)              \
                      : launch_wgrad_bf16<CI, NG_, COB_, WCO_, WEN_, false, NT>(d, dy, dw_packed, s);
#define GSSD_WB(CI, NG_, COB_, WCO_, WEN_) GSSD_WB1(CI, NG_, COB_, WCO_, WEN_, 9)
    if (d.KH == 1) {                                             // 1x1: the tile is the patch; 64 output channels per workgroup
        if (cg == 64) { GSSD_WB1(64, 1, 4, 1, 4, 1) }
        // long rows (the deformable conv's 9 * Cin columns): 128 x 128 gradient blocks halve the L2 -> LDS traffic per MFMA
        if (cg >= 2048 && ng % 128 == 0) { GSSD_WB1(128, 1, 4, 2, 2, 1) }
        GSSD_WB1(128, 1, 4, 1, 4, 1)                             // (cg a multiple of 128: blocks)
    }
    if (cg == 16 && ng == 16) { GSSD_WB(16, 4, 1, 1, 1) }        // conv1_2: a wave per phase group, 9 tiles
    if (cg == 16 && ng == 32) { GSSD_WB(16, 4, 2, 1, 1) }        // conv2_1: 18 tiles per wave
    if (cg == 32 && ng == 32 && d.groups % 4 == 0) { GSSD_WB(32, 4, 2, 1, 1) }        // conv2_2: 36 tiles per wave
    if (cg == 32) { GSSD_WB(32, 1, 2, 2, 2) }                    // conv3_1: 64 output channels per workgroup, 18 tiles per wave
    if (cg == 64) { GSSD_WB(64, 1, 2, 2, 2) }                    // conv3_2 / conv3_3 / conv4_1: 36 tiles per wave
    GSSD_WB(128, 1, 2, 1, 4)                                     // conv4_2 .. conv5_3 and the dense convs (DCN offset / mask conv, heads): 32
                                                                 // output channels per workgroup, 36 tiles per wave
#undef GSSD_WB
#undef GSSD_WB1
}
