// Weight gradient of the implicit-GEMM convolution for gfx950 (fp32 MFMA 16x16x4).
//
//   dWp[n][k] += sum_m dY[m][n] * A[m][k]        (A = the forward's im2col view of the NHWC input)
//
// The reduction runs over output pixels m, the slow index of both NHWC operands, so both tiles are staged
// pixel-major: LDS rows = 32 pixels, columns = BMW output channels of dY and BNW im2col columns k (each 16-byte quad
// of a row is one (tap, 4 input channels) piece fetched from its own shifted pixel, exactly the forward's A-loader
// addressing).  MFMA operands are read with ONE ds_read_b128 per k-step for FOUR 16-wide fragment tiles: fragment
// tile j's row r is mapped to column 4*r + j of the staged block, so the 4 consecutive floats a lane reads feed 4
// different MFMAs (the output permutation is undone in the epilogue).  Pixels are split over grid.x (split-K) and
// the partial products accumulate into the zero-filled packed gradient with fp32 atomics.
// The consumer-side fused BatchNorm+ReLU of the forward (in_scale/in_shift/in_pad) is re-applied to the A operand.
//
// Replaces autograd's conv weight-gradient kernels (cuDNN wgrad behind loss.backward(),
// train_lesion_multiphase_v2.py:247-248).
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BP = 32;      // pixels per K-chunk

__device__ __attribute__((aligned(16))) float g_zero_page_w[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct WgradParams {
    const float* in;       // forward input (NHWC), possibly raw with in_scale/in_shift/in_pad
    const float* dy;       // [M][Cout] dense NHWC gradient of the conv output
    float* dw;             // packed [Cout][K], zero-filled by the caller
    const float* in_scale;
    const float* in_shift;
    const float* in_pad;
    int B, H, W, in_stride, in_ch_off, Ho, Wo, Cout, groups, cin_g, KH, KW, stride, pad, dil, K;
    int M, pix_per_slice;
    int n_tiles, m_tiles;  // per group
};

// MT = 4: 64 output channels per wave (b128 trick on dY); MT = 1: 16 output channels per wave (scalar reads)
template <int MT, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BMW = WM * MT * 16, BNW = WN * 64;
    constexpr int QA = BMW / 4, QB = BNW / 4;            // quads per pixel row
    constexpr int ROWF = BMW + BNW;                      // floats per pixel row: [dY block | im2col block]
    constexpr int STAGE = BP * ROWF;
    constexpr int APIECES = BP * QA / 64, BPIECES = BP * QB / 64;     // 1-KiB DMA pieces per chunk
    static_assert(BP * QA % 64 == 0 && BP * QB % 64 == 0, "tile pieces");
    constexpr int AR = (APIECES + 3) / 4, BR = (BPIECES + 3) / 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, kq = lane >> 4;
    const int cout_g = p.Cout / p.groups;
    const int tiles_per_group = p.m_tiles * p.n_tiles;
    const int g = blockIdx.y / tiles_per_group;
    const int trem = blockIdx.y % tiles_per_group;
    const int co0 = (trem / p.n_tiles) * BMW, k0 = (trem % p.n_tiles) * BNW;
    const int HoWo = p.Ho * p.Wo;
    const int taps = p.KH * p.KW;
    const int m_begin = blockIdx.x * p.pix_per_slice;
    const int m_end = min(p.M, m_begin + p.pix_per_slice);
    const float* zero = g_zero_page_w;
    const bool xf = p.in_scale != nullptr;
    const float* in = p.in + p.in_ch_off + g * p.cin_g;

    // ---- DMA roles.  A piece = 64 lanes = (64/QA) pixel rows x QA quads of dY; B piece likewise for im2col. ------------
    // dY: lane -> (row offset inside the piece, quad)
    const int a_rows = 64 / (QA < 64 ? QA : 64);         // pixel rows per A piece
    const int a_prow = lane / QA, a_q = lane % QA;
    const int b_rows = (QB >= 64) ? 1 : 64 / QB;
    const int b_q_in = lane % (QB < 64 ? QB : 64);
    // im2col quad of this lane for B piece j: when QB > 64 a pixel row spans QB/64 pieces
    int b_tapdy[BR], b_tapdx[BR], b_c[BR];
    bool b_kok[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int piece = j * 4 + wave;
        const int q = (QB > 64) ? (piece % (QB / 64)) * 64 + lane : b_q_in;
        const int k = k0 + 4 * q;
        const int tap = k / p.cin_g;
        b_c[j] = k - tap * p.cin_g;
        b_tapdy[j] = (tap / p.KW) * p.dil - p.pad;
        b_tapdx[j] = (tap % p.KW) * p.dil - p.pad;
        b_kok[j] = k < p.K && tap < taps;
    }

    auto issue = [&](int m0, int buf) {
        float* S = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const int piece = j * 4 + wave;
            if (piece < APIECES) {
                const int prow = piece * a_rows + a_prow;              // pixel row inside the chunk
                const int m = m0 + prow;
                const int co = co0 + 4 * a_q;
                const bool ok = m < m_end && co < cout_g;
                const float* src = ok ? p.dy + (size_t)m * p.Cout + g * cout_g + co : zero;
                // LDS image of a piece is lane-linear: rows of QA quads packed back to back
                dma16(src, S + piece * 256);
            }
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int piece = j * 4 + wave;
            if (piece < BPIECES) {
                const int prow = (QB > 64) ? piece / (QB / 64) : piece * b_rows + lane / QB;
                const int m = m0 + prow;
                bool ok = m < m_end && b_kok[j];
                const float* src = zero;
                if (ok) {
                    const int b = m / HoWo, pix = m - b * HoWo;
                    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
                    const int iy = oy * p.stride + b_tapdy[j], ix = ox * p.stride + b_tapdx[j];
                    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                        src = in + ((size_t)(b * p.H + iy) * p.W + ix) * p.in_stride + b_c[j];
                    else if (xf)
                        src = p.in_pad + p.in_ch_off + g * p.cin_g + b_c[j];
                }
                dma16(src, S + BP * BMW + piece * 256);
            }
        }
    };

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fused producer BN + ReLU on the im2col operand: this lane's 4 columns are k = k0 + wn*64 + 4*r + {0..3}
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (xf) {
        const int k = k0 + wn * 64 + 4 * r;
        const int c = k % p.cin_g;
        if (k < p.K) {
            sc = *reinterpret_cast<const f32x4*>(p.in_scale + p.in_ch_off + g * p.cin_g + c);
            sh = *reinterpret_cast<const f32x4*>(p.in_shift + p.in_ch_off + g * p.cin_g + c);
        }
    }

    const int nchunks = (m_end - m_begin + BP - 1) / BP;
    if (nchunks > 0) issue(m_begin, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) issue(m_begin + (ch + 1) * BP, buf ^ 1);
        const float* A = smem + buf * STAGE;                 // [BP][BMW]
        const float* Bm = A + BP * BMW;                      // [BP][BNW]
        // software-pipelined by one k-step: step s+1's operands are requested before step s's MFMAs
        auto ldb = [&](int s) -> f32x4 { return *reinterpret_cast<const f32x4*>(Bm + (4 * s + kq) * BNW + wn * 64 + 4 * r); };
        auto lda4 = [&](int s) -> f32x4 { return *reinterpret_cast<const f32x4*>(A + (4 * s + kq) * BMW + wm * 64 + 4 * r); };
        auto lda1 = [&](int s) -> float { return A[(4 * s + kq) * BMW + wm * 16 + r]; };
        f32x4 bv = ldb(0), av4 = {0.f, 0.f, 0.f, 0.f}, bn_ = bv, an4 = av4;
        float av1 = 0.f, an1 = 0.f;
        if constexpr (MT == 4) av4 = lda4(0);
        else av1 = lda1(0);
#pragma unroll
        for (int s = 0; s < BP / 4; ++s) {
            if (s + 1 < BP / 4) {
                bn_ = ldb(s + 1);
                if constexpr (MT == 4) an4 = lda4(s + 1);
                else an1 = lda1(s + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (xf) {
                bv = bv * sc + sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = fmaxf(bv[e], 0.f);
            }
            if constexpr (MT == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[i], bv[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv[j], acc[0][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            bv = bn_;
            av4 = an4;
            av1 = an1;
        }
        __syncthreads();
    }

    // ---- epilogue: undo the 4*r + j column permutation, fp32 atomics into the packed gradient ----------------------------
    // Straight from the accumulator layout an atomic instruction would touch 8 cache lines with 8 floats each (lanes r are 16 bytes
    // apart, the four kq groups in different rows) -- measured on the slot kernel at 37 G atomics/s, i.e. more than the MFMAs of a
    // short reduction slice.  Each wave transposes its block through 32 x 65 floats of the (idle) stage memory, 32 rows per pass, so
    // that one instruction adds 64 consecutive floats of one row = two full lines.
    __syncthreads();                                          // every wave is done reading the stages
    constexpr int TP = 65;
    float* const tb = smem + wave * (32 * TP);
    constexpr int PASSES = MT == 4 ? 2 : 1;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e1 = 0; e1 < (MT == 4 ? 2 : 4); ++e1) {
                    const int e = MT == 4 ? 2 * ps + e1 : e1;
                    const int local = MT == 4 ? (kq * 2 + e1) * 4 + i : kq * 4 + e;
                    tb[local * TP + 4 * r + j] = acc[i][j][e];
                }
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the wave's own writes have landed (nobody else reads them)
        __builtin_amdgcn_wave_barrier();
        const int k = k0 + wn * 64 + lane;
        if (k < p.K) {
#pragma unroll 8
            for (int local = 0; local < (MT == 4 ? 32 : 16); ++local) {
                const int co = MT == 4 ? co0 + wm * 64 + 4 * (4 * (local >> 3) + 2 * ps + ((local >> 2) & 1)) + (local & 3)
                                       : co0 + wm * 16 + local;
                if (co < cout_g) unsafeAtomicAdd(p.dw + (size_t)(g * cout_g + co) * p.K + k, tb[local * TP + lane]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int MT, int WM, int WN>
int launch_wgrad(WgradParams& p, hipStream_t stream) {
    constexpr int BMW = WM * MT * 16, BNW = WN * 64;
    const int cout_g = p.Cout / p.groups;
    p.m_tiles = (cout_g + BMW - 1) / BMW;
    p.n_tiles = (p.K + BNW - 1) / BNW;
    const long long tiles = (long long)p.groups * p.m_tiles * p.n_tiles;
    // split the pixel range so that ~3 workgroups per CU exist, each with >= 16 chunks: every slice adds one full
    // [Cout][K] round of fp32 atomics, which would dominate the traffic of the small-map layers otherwise
    long long slices = (768 + tiles - 1) / tiles;
    const long long max_slices = (p.M + 16 * BP - 1) / (16 * BP);
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    p.pix_per_slice = (int)(((p.M + slices - 1) / slices + BP - 1) / BP * BP);
    const int gx = (p.M + p.pix_per_slice - 1) / p.pix_per_slice;
    constexpr size_t smem = 2 * (size_t)BP * (BMW + BNW) * sizeof(float);
    auto kern = conv_wgrad_kernel<MT, WM, WN>;
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    if (gssd_attr_needed(&attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute failed (wgrad)");
            return GSSD_ELAUNCH;
        }
    }
    gssd_attr_done(&attr_mask);
    hipLaunchKernelGGL(kern, dim3(gx, (unsigned)tiles), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

__global__ void unpack_weight_grad_kernel(const float* __restrict__ wp, float* __restrict__ w, int Cout, int cin_g,
                                          int taps, int cin_g_pad, int Kpad, int accumulate) {
    const long long total = (long long)Cout * cin_g * taps;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long long oc = i / taps;
        const int c = (int)(oc % cin_g);
        const int o = (int)(oc / cin_g);
        const float v = wp[(long long)o * Kpad + tap * cin_g_pad + c];
        w[i] = accumulate ? w[i] + v : v;
    }
}

__global__ void pack_weight_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int groups,
                                         int cin_g, int KH, int KW) {
    // rows: input channels (global), columns k' = flipped_tap * cout_g + co_local
    const int cout_g = Cout / groups, taps = KH * KW;
    const int Kd = taps * cout_g;
    const long long total = (long long)groups * cin_g * Kd;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % Kd);
        const int ci_glob = (int)(i / Kd);
        const int g = ci_glob / cin_g, ci = ci_glob - g * cin_g;
        const int tapf = kk / cout_g, co = kk - tapf * cout_g;
        const int tap = taps - 1 - tapf;                       // 180-degree flip
        wp[i] = w[((long long)(g * cout_g + co) * cin_g + ci) * taps + tap];
    }
}

}  // namespace

extern "C" int gssd_conv2d_wgrad_f32(const gssd_conv_desc* dp, const float* dy, float* dw_packed, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dp && dy && dw_packed);
    const gssd_conv_desc& d = *dp;
    GSSD_CHECK_ARG(d.in && d.groups > 0 && d.Cout % d.groups == 0 && d.cin_g % 4 == 0 && !d.m_per_image);
    GSSD_CHECK_ARG(d.K == d.KH * d.KW * d.cin_g && d.in_stride % 4 == 0 && d.in_ch_off % 4 == 0);
    GSSD_CHECK_ARG((d.Cout / d.groups) % 4 == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)d.in % 16) == 0);
    GSSD_CHECK_ARG((d.in_scale == nullptr) == (d.in_shift == nullptr) && (d.in_scale == nullptr) == (d.in_pad == nullptr));
    {
        const int rc = gssd_try_conv_thin_wgrad(d, dy, dw_packed, as_stream(stream));    // conv1_1 / conv1_2: patch-staged
        if (rc != 1) return rc;
    }
    {
        const int rc = gssd_try_conv_patch_wgrad(d, dy, dw_packed, as_stream(stream));   // conv2_1 .. conv3_3: patch-staged per group
        if (rc != 1) return rc;
    }
    {
        static const bool no_slot = getenv("GSSD_NO_GEMM_SLOT") != nullptr || getenv("GSSD_NO_WGRAD_SLOT") != nullptr;   // ablation switches
        const int rc = no_slot ? 1 : gssd_try_wgrad_slot(d, dy, dw_packed, as_stream(stream));   // large plain 1x1 / DCN contraction
        if (rc != 1) return rc;
    }
    WgradParams p;
    p.in = d.in;
    p.dy = dy;
    p.dw = dw_packed;
    p.in_scale = d.in_scale;
    p.in_shift = d.in_shift;
    p.in_pad = d.in_pad;
    p.B = d.B; p.H = d.H; p.W = d.W; p.in_stride = d.in_stride; p.in_ch_off = d.in_ch_off; p.Ho = d.Ho; p.Wo = d.Wo;
    p.Cout = d.Cout; p.groups = d.groups; p.cin_g = d.cin_g; p.KH = d.KH; p.KW = d.KW; p.stride = d.stride;
    p.pad = d.pad; p.dil = d.dil; p.K = d.K;
    const long long M = (long long)d.B * d.Ho * d.Wo;
    GSSD_CHECK_ARG(M < (1ll << 31));
    p.M = (int)M;
    const int cout_g = d.Cout / d.groups;
    hipStream_t s = as_stream(stream);
    if (cout_g >= 128) return launch_wgrad<4, 2, 2>(p, s);      // 128 co x 128 k
    if (cout_g >= 64) return launch_wgrad<4, 1, 4>(p, s);       // 64 co x 256 k
    if (cout_g > 16) return launch_wgrad<1, 2, 2>(p, s);        // 32 co x 128 k
    return launch_wgrad<1, 1, 4>(p, s);                         // 16 co x 256 k
}

extern "C" int gssd_unpack_conv_weight_grad(const float* w_packed, float* w_oihw, int Cout, int cin_g, int KH, int KW,
                                            int cin_g_pad, int Kpad, int accumulate, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_packed && w_oihw && Cout > 0 && cin_g > 0 && cin_g_pad >= cin_g && Kpad >= KH * KW * cin_g_pad);
    const long long total = (long long)Cout * cin_g * KH * KW;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unpack_weight_grad_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w_packed, w_oihw, Cout,
                       cin_g, KH * KW, cin_g_pad, Kpad, accumulate);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_pack_conv_weight_dgrad(const float* w_oihw, float* w_packed, int Cout, int groups, int cin_g, int KH,
                                           int KW, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w_oihw && w_packed && Cout > 0 && groups > 0 && Cout % groups == 0 && cin_g > 0);
    const long long total = (long long)groups * cin_g * KH * KW * (Cout / groups);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_dgrad_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w_oihw, w_packed, Cout,
                       groups, cin_g, KH, KW);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
