// bf16 fused modulated deformable 3x3 convolution (BASELINE.json configs[4]): the same algorithm and entry contract as
// dcn_fused.hip (sampling + contraction + bias in one kernel, no column buffer) with bf16 x / weights / output, fp32 offsets,
// fp32 bilinear blend and fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// With the matrix pipe 16x faster than in fp32 the kernel is bound by the gather and by streaming the weight slab, so the tile
// is the simple two-workgroups-per-CU shape: BM = 128 pixels x BN = 256 channels, 4 waves (2 x 2, 64 x 128 each), K chunks of 32
// channels of one tap = one MFMA k-step = 64-byte rows; per chunk a thread gathers 2 cells (4 corners x 16 B = 8 channels each).
// Operand roles are swapped and the weight rows of a 32-channel block are staged in the order of conv_bf16.hip, so a lane ends up
// with 8 consecutive output channels of one pixel: 16-byte NHWC stores.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

namespace {

constexpr int BM = 128, BN = 256, BKC = 32;          // tile; channels (bf16) per K chunk: 64 B per row
constexpr int WTM = 64, WTN = 128, MT = WTM / 16, NT = WTN / 16;
constexpr int A_STAGE = BM * BKC, B_STAGE = BN * BKC;            // u16 elements
constexpr int LDS_BYTES = 2 * (A_STAGE + B_STAGE) * 2 + 9 * BM * 16 + 9 * BM * 4;

__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }       // 64-byte rows: conflict-free ds_read_b128

__device__ __forceinline__ int chan_of_row(int row) {        // LDS row of the weight tile -> output channel inside the BN tile
    const int j = row >> 4, rho = row & 15;
    return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
}

// wp: [n_tiles][chunks][BN rows in staging order][32] bf16 (slot-swizzled), chunk = (d * cpg/32 + c32) * 9 + tap
__global__ __launch_bounds__(256, 2) void dcn_bf16_kernel(const u16* __restrict__ x, const float* __restrict__ om,
                                                         const u16* __restrict__ wp, const float* __restrict__ bias,
                                                         u16* __restrict__ out, int M, int H, int W, int C, int dg, int om_stride,
                                                         int Cout, int ntn, int mtiles) {
    extern __shared__ __attribute__((aligned(16))) u16 smem_h[];
    u16* const As = smem_h;                                   // [2][BM][32]
    u16* const Bs = smem_h + 2 * A_STAGE;                     // [2][BN][32]
    f32x4* const setw = reinterpret_cast<f32x4*>(smem_h + 2 * (A_STAGE + B_STAGE));            // [9][BM]
    int* const setp = reinterpret_cast<int*>(smem_h + 2 * (A_STAGE + B_STAGE) + 9 * BM * 8);   // [9][BM]

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
    const u16* wslab = wp + (size_t)nt * nchunks * B_STAGE;

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
    bf16x8 gv[2][4];

    auto gather_issue = [&]() {
        const int cb = ch_d * cpg + ch_c * BKC + gq * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = ch_tap * BM + gp + 64 * j;
            gw[j] = setw[e];
            const int pos = setp[e];
            const unsigned i00 = (unsigned)(pos & 0x3FFFFFFF);
            const unsigned dxb = ((unsigned)pos >> 30) & 1u, dyb = (unsigned)pos >> 31;
            const unsigned i10 = i00 + dyb * (unsigned)W;
            gv[j][0] = *reinterpret_cast<const bf16x8*>(x + (size_t)(i00 * (unsigned)C + cb));
            gv[j][1] = *reinterpret_cast<const bf16x8*>(x + (size_t)((i00 + dxb) * (unsigned)C + cb));
            gv[j][2] = *reinterpret_cast<const bf16x8*>(x + (size_t)(i10 * (unsigned)C + cb));
            gv[j][3] = *reinterpret_cast<const bf16x8*>(x + (size_t)((i10 + dxb) * (unsigned)C + cb));
        }
    };
    auto gather_finish = [&](int buf) {
        u16* Ad = As + buf * A_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = (__bf16)((float)gv[j][0][e] * gw[j][0] + (float)gv[j][1][e] * gw[j][1] + (float)gv[j][2][e] * gw[j][2] +
                                (float)gv[j][3][e] * gw[j][3]);
            *reinterpret_cast<bf16x8*>(Ad + a_wr0 + j * 64 * BKC) = o;
        }
    };
    auto b_issue = [&](int chunk, int buf) {
        const u16* src = wslab + (size_t)chunk * B_STAGE + lane * 8;
        u16* dst = Bs + buf * B_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = j * 4 + wave;                        // 16 pieces of 1 KiB
            dma16(src + piece * 512, dst + piece * 512);
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
        if (more) {
            if (ch_tap == 0 && ch_c == 0) {
                setups(ch_d);
                __syncthreads();
            }
            gather_issue();
            b_issue(ch + 1, buf ^ 1);
        }
        const u16* Ab = As + buf * A_STAGE + wm * WTM * BKC + fo;
        const u16* Bb = Bs + buf * B_STAGE + wn * WTN * BKC + fo;
        bf16x8 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * BKC);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * BKC);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (more) {
            gather_finish(buf ^ 1);
            advance();
        }
        __syncthreads();
    }

    // ---- epilogue: + bias, 16-byte NHWC bf16 stores (lane: pixel = lane & 15, 8 consecutive channels per tile pair) ---------
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
            if (n0 + 8 <= Cout) {
                bf16x8 o;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = (__bf16)(acc[i][2 * u + (c >> 2)][c & 3] + bv[c]);
                *reinterpret_cast<bf16x8*>(out + (size_t)m * Cout + n0) = o;
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (n0 + c < Cout) out[(size_t)m * Cout + n0 + c] = __builtin_bit_cast(u16, (__bf16)(acc[i][2 * u + (c >> 2)][c & 3] + bv[c]));
            }
        }
    }
}

// OIHW fp32 [Cout][C][3][3] -> bf16 [n_tiles][chunks][BN staging rows][32] with the slot swizzle; rows beyond Cout are zero
__global__ void dcn_pack_weight_bf16_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int C, int dg, long long total) {
    const int cpg = C / dg, cpc = cpg / BKC, nchunks = dg * cpc * 9;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int slot = (int)((i >> 3) & 3);
        const int row = (int)((i >> 5) % BN);
        const long long t = (i >> 5) / BN;
        const int chunk = (int)(t % nchunks);
        const int nt = (int)(t / nchunks);
        const int q = slot ^ swz(row);
        const int tap = chunk % 9, cc = chunk / 9;
        const int c = cc * BKC + q * 8 + e;
        const int n = nt * BN + chan_of_row(row);
        wp[i] = __builtin_bit_cast(u16, (__bf16)(n < Cout ? w[((size_t)n * C + c) * 9 + tap] : 0.f));
    }
}

}  // namespace

extern "C" long long gssd_dcn_packed_weight_elems_bf16(int Cout, int C) {
    if (Cout <= 0 || C <= 0 || C % BKC != 0) return -1;
    return (long long)((Cout + BN - 1) / BN) * BN * 9 * C;
}

extern "C" int gssd_dcn_pack_weight_bf16(const float* w_oihw, void* w_packed, int Cout, int C, int dg, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_oihw && w_packed && Cout > 0 && C > 0 && dg > 0 && C % dg == 0 && (C / dg) % BKC == 0);
    const long long total = gssd_dcn_packed_weight_elems_bf16(Cout, C);
    hipLaunchKernelGGL(dcn_pack_weight_bf16_kernel, dim3((int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_oihw, reinterpret_cast<u16*>(w_packed), Cout, C, dg, total);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_forward_bf16(const void* x, const float* om, const void* w_packed, const float* bias, void* out, int B, int H,
                                     int W, int C, int dg, int om_stride, int Cout, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && w_packed && out && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0 && Cout > 0 && Cout % 8 == 0);
    GSSD_CHECK_ARG(C % dg == 0 && (C / dg) % BKC == 0 && om_stride >= 27 * dg);
    GSSD_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)out % 16) == 0);
    const long long Mll = (long long)B * H * W;
    GSSD_CHECK_ARG(Mll < (1ll << 30) && Mll * C < (1ll << 32));          // 30-bit pixel index + 2 flag bits; 32-bit element offsets
    const int M = (int)Mll;
    const int ntn = (Cout + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    static unsigned attr_mask = 0;
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
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
    hipLaunchKernelGGL(dcn_bf16_kernel, dim3(blocks), dim3(256), LDS_BYTES, as_stream(stream), reinterpret_cast<const u16*>(x), om,
                       reinterpret_cast<const u16*>(w_packed), bias, reinterpret_cast<u16*>(out), M, H, W, C, dg, om_stride, Cout, ntn,
                       mtiles);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
