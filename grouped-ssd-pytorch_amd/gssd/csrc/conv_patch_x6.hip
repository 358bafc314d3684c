// Dense 3x3 convolutions with MANY input channels and FEW outputs (<= 128) of the fp32 mode, on the fp16 matrix cores with fp32-equivalent
// products (round 6): the deformable conv's offset conv (models/ssd_multiphase_custom_group.py:330-338 -> layers/dcn_v2_custom.py:75-83: 1024 -> 108
// channels on the 38 x 38 map) -- 0.5 ms of the step's critical path on conv_wino_x6 / conv_x6, both bound by what they do PER INPUT VALUE: Winograd
// transforms every 4 x 4 input tile for each block of 64 outputs (with 108 outputs the transform outweighs the products), the implicit GEMM loads and
// splits every input value nine times, once per tap.  Here an input value is loaded and split ONCE per 32-channel chunk:
//   * a workgroup (four waves) owns an 8 x 16 output tile and all (<= 128) output channels, accumulators in registers for the whole K loop;
//   * per chunk of 32 input channels the 10 x 18 patch is staged through registers (loads a chunk ahead), split into two fp16 planes (h, (x - h) * 2048:
//     conv_thin_x6.hip) and written to LDS as [plane][pixel][32 channels], 16-byte units swizzled by the patch column;
//   * the nine taps of the chunk are nine K steps of 32: the A fragment of (tap, output row) is a shifted 16-pixel window of the patch, the weight planes of
//     (chunk, tap) -- packed once in LDS-image order -- arrive by LDS-DMA through a three-stage ring, two taps ahead; one barrier per tap;
//   * wave w owns output channels 32 w .. 32 w + 31: per tap 16 activation + 4 weight fragment reads for 48 MFMAs (three per product: h h' into the main
//     accumulator, h l' + l h' into a second one that enters with 2^-11).
// Only launches the caller flags GSSD_CONV_F16_OK (include/gssd_hip.h) and points at the packed planes (gssd_conv_desc::wgt_patch): everything else stays
// where it was.  GSSD_PATCH_X6=0 switches it off.
#include "common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

namespace {

__device__ __attribute__((aligned(16))) float g_zero_px6[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int TH = 8, TW = 16, PW = TW + 2, PHT = TH + 2, NPATCH = PHT * PW;      // 180 patch pixels
constexpr int NPAD = 192, CK = 32, BN = 128, NTHR = 256, NSTG = 3;
constexpr int PLANE = NPAD * CK;                  // u16 elements of one activation plane (12 KB)
constexpr int WPLANE = BN * CK;                   // u16 elements of one weight plane of a (chunk, tap) stage (8 KB)
constexpr int WSTAGE = 2 * WPLANE;                // both planes: 16 KB
constexpr int LDS_BYTES = (2 * PLANE + NSTG * WSTAGE) * 2;      // 24 + 48 KB: two workgroups per CU
constexpr int NLD = NPAD * (CK / 4) / NTHR;       // 16-byte loads per thread and chunk: 6

// 64-byte pixels: bit 2 of the patch column into bit 1 of the 16-byte unit (conv_thin_x6.hip: conflict-free for the lane groups of a ds_read_b128)
__device__ __forceinline__ int swz_a(int col) { return (col >> 1) & 2; }
// 64-byte weight rows (dcn_x6.hip)
__device__ __host__ __forceinline__ int swz_b(int row) { return (row & 8) ? 3 : 0; }

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void split2_pair(const float a, const float b, unsigned& ph, unsigned& pl) {
    const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
    const f32x2 r = (f32x2{a, b} - __builtin_convertvector(h, f32x2)) * 2048.f;
    ph = __builtin_bit_cast(unsigned, h);
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

struct PatchX6Params {
    const float* in;
    const u16* wp;           // [chunk][tap][plane][128 rows][32] fp16, rows' 16-byte units swizzled (gssd_conv_patch_x6_pack_weight)
    const float* bias;
    float* out;
    int B, H, W, C, in_stride, Cout, out_stride, out_ch_off, tiles_y, tiles_x, relu;
};

__global__ __launch_bounds__(NTHR, 2) void conv_patch_x6_kernel(const PatchX6Params p) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* const planes = smem;                       // [2][NPAD][32]
    u16* const wring = smem + 2 * PLANE;            // [NSTG][2][128][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    // XCD-aware tile order: workgroup ids go round-robin over the XCDs; an XCD takes a contiguous eighth of the tile list (neighbours share halos in L2)
    const int ntiles = p.B * p.tiles_y * p.tiles_x;
    const int per = (ntiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || tile >= ntiles) return;
    const int tpi = p.tiles_y * p.tiles_x;
    const int b = tile / tpi, trem = tile - b * tpi;
    const int y0 = (trem / p.tiles_x) * TH, x0 = (trem % p.tiles_x) * TW;
    const int nchunks = p.C / CK, nsteps = nchunks * 9;

    // ---- staging roles: thread -> 16-byte unit u (4 fp32 channels) of patch pixels pp0 + 32 it ----
    const int su = tid & 7, pp0 = tid >> 3;
    const float* src[NLD];
    int wr[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int pp = pp0 + 32 * it;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = pp < NPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        src[it] = ok ? p.in + ((size_t)(b * p.H + iy) * p.W + ix) * p.in_stride + 4 * su : nullptr;
        wr[it] = pp * CK + (((su >> 1) ^ swz_a(px)) << 3) + 4 * (su & 1);
    }
    f32x4 pre[NLD];
    auto load_chunk = [&](int cc) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) pre[it] = src[it] ? *reinterpret_cast<const f32x4*>(src[it] + cc * CK) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto write_planes = [&]() {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            unsigned h0, l0, h1, l1;
            split2_pair(pre[it][0], pre[it][1], h0, l0);
            split2_pair(pre[it][2], pre[it][3], h1, l1);
            *reinterpret_cast<u32x2*>(planes + wr[it]) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(planes + PLANE + wr[it]) = u32x2{l0, l1};
        }
    };
    // weight planes of K step s (= chunk * 9 + tap) -> ring slot s % NSTG: 16 pieces of 1 KB, four per wave
    auto dma_stage = [&](int s) {
        const u16* g = p.wp + (size_t)s * WSTAGE + lane * 8;
        u16* d = wring + (s % NSTG) * WSTAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = 4 * q + wave;
            dma16(g + piece * 512, d + piece * 512);
        }
    };

    f32x4 acc[TH][2], acx[TH][2];
#pragma unroll
    for (int i = 0; i < TH; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: planes of chunk 0, weight stages 0 and 1 ----
    load_chunk(0);
    dma_stage(0);
    dma_stage(1);
    write_planes();
    if (nchunks > 1) load_chunk(1);

    const int boff = (32 * wave + r) * CK + ((kq ^ swz_b(r)) << 3);       // this lane's weight fragment of column tile 0 (tile 1: + 16 rows)
    for (int s = 0; s < nsteps; ++s) {
        const int cc = s / 9, tap = s - cc * 9;
        const int dy = tap / 3, dx = tap - dy * 3;
        // this wave's pieces of stage s have landed (issued two steps ago; younger: stage s + 1's four pieces and, behind a chunk start, the six loads
        // of the next chunk's patch -- all counted in issue order)
        // (inline assembly, not __syncthreads(): its release fence waits for vmcnt(0) -- the DMA pieces and patch loads in flight)
        if (s == nsteps - 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (nothing younger was issued)
        else if (tap == 1 && cc + 1 < nchunks) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // every wave's pieces of stage s (and the planes written last step) are in LDS; every wave is done with step s - 1 (ring slot (s + 2) % 3 is free)
        if (s + 2 < nsteps) dma_stage(s + 2);
        if (tap == 0 && s > 0 && cc + 1 < nchunks) load_chunk(cc + 1);       // (chunk 1's loads were issued in the prologue)
        const u16* wst = wring + (s % NSTG) * WSTAGE;
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bh[j] = *reinterpret_cast<const f16x8*>(wst + boff + j * 16 * CK);
            bl[j] = *reinterpret_cast<const f16x8*>(wst + WPLANE + boff + j * 16 * CK);
        }
        const int col = r + dx;
        const int aoff = col * CK + ((kq ^ swz_a(col)) << 3);
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            const u16* ap = planes + (i + dy) * (PW * CK) + aoff;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap);
            const f16x8 al = *reinterpret_cast<const f16x8*>(ap + PLANE);
            // (no two consecutive MFMAs on one accumulator)
            acx[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[0], ah, acx[i][0], 0, 0, 0);
            acx[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[1], ah, acx[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[0], ah, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[1], ah, acc[i][1], 0, 0, 0);
            acx[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[0], al, acx[i][0], 0, 0, 0);
            acx[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[1], al, acx[i][1], 0, 0, 0);
        }
        if (tap == 8 && cc + 1 < nchunks) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading chunk cc's planes
            write_planes();       // chunk cc + 1 (loaded during this chunk); visible behind the next step's barrier
        }
    }

    // ---- epilogue: main + cross / 2048 + bias (+ ReLU); lane: pixel r of output row i, channels 32 wave + 16 j + 4 kq .. + 3 ----
    const int x = x0 + r;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n0 = 32 * wave + 16 * j + 4 * kq;
        if (n0 >= p.Cout) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n0);
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            const int y = y0 + i;
            if (y >= p.H || x >= p.W) continue;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = __builtin_fmaf(acx[i][j][e], 1.f / 2048.f, acc[i][j][e]) + bv[e];
                if (p.relu) v[e] = fmaxf(v[e], 0.f);
            }
            *reinterpret_cast<f32x4*>(p.out + ((size_t)(b * p.H + y) * p.W + x) * p.out_stride + p.out_ch_off + n0) = v;
        }
    }
}

// packed fp32 K-major rows [Cout][9 * C] (k = tap * C + c) -> the two fp16 planes in LDS-image order
__global__ void conv_patch_x6_pack_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int C, int row_stride, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), slot = (int)((i >> 3) & 3), row = (int)((i >> 5) & (BN - 1));
        const long long s = i >> 12;                      // K step = chunk * 9 + tap (WPLANE = 4096 elements per plane and step)
        const int cc = (int)(s / 9), tap = (int)(s - 9 * (long long)cc);
        const int c = cc * CK + ((slot ^ swz_b(row)) << 3) + e;
        const float v = row < Cout ? w[(size_t)row * row_stride + (size_t)tap * C + c] : 0.f;
        const _Float16 h = (_Float16)v;
        wp[s * WSTAGE + (i & (WPLANE - 1))] = __builtin_bit_cast(u16, h);
        wp[s * WSTAGE + WPLANE + (i & (WPLANE - 1))] = __builtin_bit_cast(u16, (_Float16)((v - (float)h) * 2048.f));
    }
}

bool px6_enabled() {
    static const bool on = [] {
        const char* a = getenv("GSSD_PATCH_X6");
        const char* b = getenv("GSSD_X6_F16");
        return !(a && a[0] == '0') && !(b && b[0] == '0');
    }();
    return on;
}

}  // namespace

extern "C" long long gssd_conv_patch_x6_weight_elems(int Cout, int C) {          // 16-bit elements (two fp16 planes); -1: not a shape of the kernel
    if (Cout <= 0 || Cout > BN || Cout % 4 != 0 || C <= 0 || C % CK != 0) return -1;
    return 2ll * BN * 9 * C;
}

extern "C" int gssd_conv_patch_x6_pack_weight(const float* w_packed, void* out, int Cout, int C, int row_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_packed && out && row_stride >= 9 * C && gssd_conv_patch_x6_weight_elems(Cout, C) > 0);
    const long long total = (long long)BN * 9 * C;            // elements per plane
    hipLaunchKernelGGL(conv_patch_x6_pack_kernel, dim3((int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       w_packed, reinterpret_cast<u16*>(out), Cout, C, row_stride, total);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// 1 when gssd_conv2d_nhwc_f32 runs this descriptor here
extern "C" int gssd_conv_patch_x6_takes(const gssd_conv_desc* dp) {
    if (!dp || !px6_enabled()) return 0;
    const gssd_conv_desc& d = *dp;
    if (!d.wgt_patch || !(d.flags & GSSD_CONV_F16_OK) || (d.flags & ~(GSSD_CONV_F16_OK | GSSD_CONV_OUT_F32))) return 0;
    if (d.groups != 1 || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1 || d.Ho != d.H || d.Wo != d.W) return 0;
    if (gssd_conv_patch_x6_weight_elems(d.Cout, d.cin_g) <= 0) return 0;
    if (d.out_mode != GSSD_OUT_NHWC || d.m_per_image || d.split_k > 1 || d.in_scale || d.stats || d.alpha || d.gate || d.resid || d.out2) return 0;
    if (d.in_stride % 4 || d.in_ch_off % 4 || d.out_stride % 4 || d.out_ch_off % 4) return 0;
    if (((uintptr_t)d.in % 16) || ((uintptr_t)d.out % 16) || ((uintptr_t)d.wgt_patch % 16) || (d.bias && ((uintptr_t)d.bias % 16))) return 0;
    if ((long long)d.B * d.H * d.W * d.in_stride >= (1ll << 31)) return 0;
    return 1;
}

int gssd_try_conv_patch_x6(const gssd_conv_desc& d, hipStream_t stream) {
    if (!gssd_conv_patch_x6_takes(&d)) return 1;
    PatchX6Params p;
    p.in = d.in + d.in_ch_off;
    p.wp = reinterpret_cast<const u16*>(d.wgt_patch);
    p.bias = d.bias;
    p.out = d.out;
    p.B = d.B;
    p.H = d.H;
    p.W = d.W;
    p.C = d.cin_g;
    p.in_stride = d.in_stride;
    p.Cout = d.Cout;
    p.out_stride = d.out_stride;
    p.out_ch_off = d.out_ch_off;
    p.tiles_y = (d.H + TH - 1) / TH;
    p.tiles_x = (d.W + TW - 1) / TW;
    p.relu = d.relu;
    static unsigned attr_mask = 0;
    auto kern = conv_patch_x6_kernel;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", LDS_BYTES);
            return GSSD_ELAUNCH;
        }
        gssd_attr_done(&attr_mask);
    }
    const int ntiles = p.B * p.tiles_y * p.tiles_x;
    const int per = (ntiles + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(per * 8), dim3(NTHR), LDS_BYTES, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
