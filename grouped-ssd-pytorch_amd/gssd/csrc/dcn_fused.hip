// Fused modulated deformable 3x3 convolution (DCNv2, stride 1 / pad 1 / dil 1) for gfx950: sampling + contraction in ONE
// kernel -- no column buffer, no library GEMM.  Replaces what the reference gets from the external `dcn_v2` extension at
// layers/dcn_v2_custom.py:84-89 (modulated_deformable_im2col + gemm); algorithm restated in oracle/gssd_oracle.py::dcn_v2_conv.
//
//   out[m][n] = bias[n] + sum_{d, c, tap} W[n][c][tap] * mask(m,d,tap) * bilinear(x[b, :, :, c], (h-1+i+dy, w-1+j+dx))
//
// A 256-thread workgroup (4 waves, 2 x 2) owns BM = 128 output pixels x BN = 256 output channels; a wave owns 64 x 128
// (4 x 8 MFMA tiles of v_mfma_f32_16x16x4_f32, 128 accumulator registers), two workgroups share a CU: while one gathers, the
// other keeps the matrix pipe busy.  K runs in chunks of 16 input channels of ONE tap, ordered (deformable group, channel
// chunk, tap): the nine taps of a chunk re-read the same ~20 KB window of x, so the gather hits L1 / L2 and x leaves HBM about once.
//
//   per (pixel, tap) of the current deformable group, once per group:  4 bilinear weights (mask folded in, invalid corners
//       zeroed) + the clamped corner pixel index -> LDS "setup" table (22.5 KB)
//   per chunk:  B tile (256 x 16 weights, pre-packed chunk-major and pre-swizzled: memory image == LDS image) by 16-byte
//       LDS-DMA;  A tile: every thread loads the 4 corners of 2 (pixel, 4-channel) cells as float4 straight from x (issued
//       BEFORE the chunk's MFMAs, consumed after them), blends, and writes one ds_write_b128 per cell
//   fragments: one ds_read_b128 per 16 x 16 tile per chunk (lane (r, kq) holds k = 4*kq + s for MFMA step s; A and B use the same
//       permutation); 64-byte rows, slot' = slot ^ (r & 8 ? 3 : 0) makes the b128 reads conflict free
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 256, BKC = 16;          // tile, channels per K chunk
constexpr int WTM = 64, WTN = 128, MT = WTM / 16, NT = WTN / 16;
constexpr int A_STAGE = BM * BKC, B_STAGE = BN * BKC;            // floats
constexpr int LDS_FLOATS = 2 * (A_STAGE + B_STAGE) + 9 * BM * 4 + 9 * BM;

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }

// wp: [n_tiles][chunks][BN][16] (slot-swizzled rows), chunk = (d * cpg/16 + c16) * 9 + tap
__global__ __launch_bounds__(256, 2) void dcn_fused_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                          const float* __restrict__ wp, const float* __restrict__ bias,
                                                          float* __restrict__ out, int M, int H, int W, int C, int dg,
                                                          int om_stride, int Cout, int ntn, int mtiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                                   // [2][BM][16]
    float* const Bs = smem + 2 * A_STAGE;                     // [2][BN][16]
    f32x4* const setw = reinterpret_cast<f32x4*>(smem + 2 * (A_STAGE + B_STAGE));   // [9][BM]
    int* const setp = reinterpret_cast<int*>(smem + 2 * (A_STAGE + B_STAGE) + 9 * BM * 4);   // [9][BM]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
    // XCD-aware tile order: workgroup id L runs on XCD L & 7; an XCD streams ONE weight slab (N tile) when the N-tile count
    // divides 8, and walks consecutive M tiles (shared halo rows of x stay in that L2)
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
    const int HW = H * W, cpg = C / dg, cpc = cpg / BKC;       // chunks of channels per deformable group
    const int nchunks = dg * cpc * 9;
    const float* wslab = wp + (size_t)nt * nchunks * B_STAGE;

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    // gather roles: thread -> (pixel row pl = (tid >> 2) + 64*j, 4-channel quad q = tid & 3)
    const int gq = tid & 3, gp = tid >> 2;
    const int a_wr0 = gp * BKC + ((gq ^ swz(gp)) << 2);                    // LDS float offset of cell j = 0 (j = 1: + 64 rows)
    const int fo = r * BKC + ((kq ^ swz(r)) << 2);                         // fragment read offset inside a 16-row block

    auto setups = [&](int d) {
        // 9 taps x 128 pixels of deformable group d (1152 entries over 256 threads)
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
                const float msk = 1.f / (1.f + expf(-ml));                 // torch.sigmoid (dcn_v2_custom.py:83)
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
                    const int ya = y0ok ? y0 : 0, xa = x0ok ? x0 : 0;                    // corner 00 clamped into the image
                    const int yb = y1ok ? y0 + 1 : H - 1, xb = x1ok ? x0 + 1 : W - 1;    // (zero-weight corners read a valid pixel)
                    pos = (int)((unsigned)(b * HW + ya * W + xa) | ((unsigned)(xb - xa) << 30) | ((unsigned)(yb - ya) << 31));
                }
            }
            setw[e] = wv;
            setp[e] = pos;
        }
    };

    // chunk -> (d, c16, tap)
    int ch_tap = 0, ch_c = 0, ch_d = 0;          // of the NEXT chunk to stage
    f32x4 gw[2];
    f32x4 gv[2][4];

    auto gather_issue = [&]() {
        const int cb = ch_d * cpg + ch_c * BKC + gq * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = ch_tap * BM + gp + 64 * j;
            gw[j] = setw[e];
            const int pos = setp[e];
            const unsigned i00 = (unsigned)(pos & 0x3FFFFFFF);
            const unsigned dxb = ((unsigned)pos >> 30) & 1u, dyb = (unsigned)pos >> 31;
            const unsigned i10 = i00 + dyb * (unsigned)W;
            gv[j][0] = *reinterpret_cast<const f32x4*>(x + (size_t)(i00 * (unsigned)C + cb));
            gv[j][1] = *reinterpret_cast<const f32x4*>(x + (size_t)((i00 + dxb) * (unsigned)C + cb));
            gv[j][2] = *reinterpret_cast<const f32x4*>(x + (size_t)(i10 * (unsigned)C + cb));
            gv[j][3] = *reinterpret_cast<const f32x4*>(x + (size_t)((i10 + dxb) * (unsigned)C + cb));
        }
    };
    auto gather_finish = [&](int buf) {
        float* Ad = As + buf * A_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 v = gv[j][0] * gw[j][0] + gv[j][1] * gw[j][1] + gv[j][2] * gw[j][2] + gv[j][3] * gw[j][3];
            *reinterpret_cast<f32x4*>(Ad + a_wr0 + j * 64 * BKC) = v;
        }
    };
    auto b_issue = [&](int chunk, int buf) {
        const float* src = wslab + (size_t)chunk * B_STAGE + lane * 4;
        float* dst = Bs + buf * B_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = j * 4 + wave;                        // 16 pieces of 1 KiB
            dma16(src + piece * 256, dst + piece * 256);
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

    // ---- prologue: chunk 0 -----------------------------------------------------------------------------------------------
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
            if (ch_tap == 0 && ch_c == 0) {          // next chunk opens a new deformable group: new sampling table
                setups(ch_d);
                __syncthreads();
            }
            gather_issue();
            b_issue(ch + 1, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* Ab = As + buf * A_STAGE + wm * WTM * BKC + fo;
        const float* Bb = Bs + buf * B_STAGE + wn * WTN * BKC + fo;
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 16 * BKC);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 16 * BKC);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            gather_finish(buf ^ 1);
            advance();
        }
        __syncthreads();
    }

    // ---- epilogue: + bias, NHWC store ------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = nt * BN + wn * WTN + j * 16 + r;
        if (n >= Cout) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + wm * WTM + i * 16 + kq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (mb + e < M) out[(size_t)(mb + e) * Cout + n] = acc[i][j][e] + bv;
        }
    }
}

// OIHW [Cout][C][3][3] -> [n_tiles][chunks][BN][16] with the slot swizzle; rows beyond Cout are zero
__global__ void dcn_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int C, int dg, long long total) {
    const int cpg = C / dg, cpc = cpg / BKC, nchunks = dg * cpc * 9;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const int slot = (int)((i >> 2) & 3);
        const int nl = (int)((i >> 4) % BN);
        const long long t = (i >> 4) / BN;
        const int chunk = (int)(t % nchunks);
        const int nt = (int)(t / nchunks);
        const int q = slot ^ swz(nl);                 // logical quad stored in this physical slot
        const int tap = chunk % 9, cc = chunk / 9;    // cc = d * cpc + c16
        const int c = cc * BKC + q * 4 + e;
        const int n = nt * BN + nl;
        wp[i] = n < Cout ? w[((size_t)n * C + c) * 9 + tap] : 0.f;
    }
}

}  // namespace

extern "C" long long gssd_dcn_packed_weight_elems(int Cout, int C) {
    if (Cout <= 0 || C <= 0 || C % BKC != 0) return -1;
    return (long long)((Cout + BN - 1) / BN) * BN * 9 * C;
}

extern "C" int gssd_dcn_pack_weight_f32(const float* w_oihw, float* w_packed, int Cout, int C, int dg, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_oihw && w_packed && Cout > 0 && C > 0 && dg > 0 && C % dg == 0 && (C / dg) % BKC == 0);
    const long long total = gssd_dcn_packed_weight_elems(Cout, C);
    hipLaunchKernelGGL(dcn_pack_weight_kernel, dim3((int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_oihw, w_packed, Cout, C, dg, total);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_forward_f32(const float* x, const float* om, const float* w_packed, const float* bias, float* out, int B,
                                    int H, int W, int C, int dg, int om_stride, int Cout, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && w_packed && out && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0 && Cout > 0);
    GSSD_CHECK_ARG(C % dg == 0 && (C / dg) % BKC == 0 && om_stride >= 27 * dg);
    GSSD_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0);
    const long long Mll = (long long)B * H * W;
    GSSD_CHECK_ARG(Mll < (1ll << 30) && Mll * C < (1ll << 30));          // 30-bit pixel index + 2 flag bits; 32-bit byte offsets
    const int M = (int)Mll;
    const int ntn = (Cout + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    static bool attr_set[16] = {false};
    int dev = 0;
    hipGetDevice(&dev);
    constexpr int smem = LDS_FLOATS * (int)sizeof(float);
    if (dev < 16 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=
            hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", smem);
            return GSSD_ELAUNCH;
        }
        attr_set[dev] = true;
    }
    // grid: ids round-robin over the 8 XCDs; slots cover ceil(mtiles / (8 / ntn)) groups when ntn divides 8
    int blocks;
    if (8 % ntn == 0) {
        const int per = 8 / ntn;
        blocks = ((mtiles + per - 1) / per) * 8;
    } else {
        blocks = ((mtiles * ntn + 7) / 8) * 8;
    }
    hipLaunchKernelGGL(dcn_fused_kernel, dim3(blocks), dim3(256), smem, as_stream(stream), x, om, w_packed, bias, out, M, H, W, C, dg,
                       om_stride, Cout, ntn, mtiles);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
