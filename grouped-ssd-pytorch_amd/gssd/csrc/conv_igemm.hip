// Implicit-GEMM convolution for gfx950 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   out[m][n] = epilogue( sum_k A[m][k] * Wp[n][k] )
//   m : output pixel (linear over b, oy, ox, or over one image when m_per_image)
//   n : output channel inside conv group g
//   k : (tap, input channel of the group) flattened, k = tap*cin_g + c
//
// Both operands are K-contiguous in memory (NHWC activations; K-major packed weights), so both
// tiles are staged with 16-byte loads into LDS as [row][BK+4] and read back as one ds_read_b128
// per 16x16 fragment and 16 k values: lane (r = lane&15, kq = lane>>4) holds k = 16*ks + 4*kq + s
// for MFMA step s -- A and B use the same k permutation, so the sum is unchanged.
// The +4 float row pad keeps the 16 rows of a fragment on distinct 16-byte LDS slots.
//
// A 256-thread workgroup (4 waves) owns a BM x BN output tile; global loads of K-chunk i+1 are
// issued into registers before the MFMAs of chunk i and written to the other LDS buffer after
// them (one barrier per chunk).  fp32 MFMA is exact fp32 FMA at the vector rate, so results
// match an fp32 reference to accumulation-order rounding.
//
// Replaces the implicit cuDNN/ATen kernels behind nn.Conv2d / torch.bmm on the reference path;
// see include/gssd_hip.h for the call-site map.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const gssd_conv_desc p, const int M,
                                                         const int tiles_per_group) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int AR = BM / 32;                     // A float4 loads per thread per chunk
    constexpr int BR = (BN * 8 + 255) / 256;        // B float4 loads per thread per chunk
    constexpr int STAGE = (BM + BN) * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y / tiles_per_group;
    const int n0g = (blockIdx.y % tiles_per_group) * BN;
    const int cout_g = p.Cout / p.groups;
    const int m0 = blockIdx.x * BM;
    const int img = p.m_per_image ? blockIdx.z : 0;
    const int HoWo = p.Ho * p.Wo;
    const int K = p.K;
    const int taps = p.KH * p.KW;

    const float* __restrict__ in = p.in + (size_t)img * p.in_batch_stride + p.in_ch_off + g * p.cin_g;
    const float* __restrict__ wgt =
        p.wgt + (size_t)img * p.wgt_batch_stride + (size_t)(g * cout_g) * p.wgt_row_stride;

    // ---- per-thread A rows -------------------------------------------------------------
    const int aq = tid & 7, ar = tid >> 3;
    int a_iy0[AR], a_ix0[AR], a_base[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + ar + 32 * i;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        int b = 0, pix = mm;
        if (!p.m_per_image) {
            b = mm / HoWo;
            pix = mm - b * HoWo;
        }
        const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_base[i] = b * p.H * p.W;
    }
    int a_tap = (4 * aq) / p.cin_g;
    int a_c = (4 * aq) - a_tap * p.cin_g;

    f32x4 areg[AR], breg[BR];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto gload = [&](int chunk) {
        const int ty = a_tap / p.KW, tx = a_tap - ty * p.KW;
        const int dy = ty * p.dil, dx = tx * p.dil;
        const bool tap_ok = a_tap < taps;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = a_ok[i] && tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const size_t off = (size_t)(a_base[i] + iy * p.W + ix) * p.in_stride + a_c;
            areg[i] = ok ? *reinterpret_cast<const f32x4*>(in + off) : zero4;
        }
        const int k0 = chunk * BK;
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int idx = tid + j * 256;
            const int row = idx >> 3, q = idx & 7;
            const bool ok = (idx < BN * 8) && (n0g + row < cout_g) && (k0 + 4 * q < K);
            breg[j] = ok ? *reinterpret_cast<const f32x4*>(wgt + (size_t)(n0g + row) * p.wgt_row_stride + k0 + 4 * q)
                         : zero4;
        }
        a_c += BK;
        while (a_c >= p.cin_g) {
            a_c -= p.cin_g;
            ++a_tap;
        }
    };
    auto lds_store = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + BM * LDK;
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(As + (ar + 32 * i) * LDK + 4 * aq) = areg[i];
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int idx = tid + j * 256;
            if (idx < BN * 8) *reinterpret_cast<f32x4*>(Bs + (idx >> 3) * LDK + 4 * (idx & 7)) = breg[j];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    const int nchunks = (K + BK - 1) / BK;
    gload(0);
    lds_store(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) gload(ch + 1);
        const float* As = smem + buf * STAGE + (wm * WTM + r) * LDK + kq * 4;
        const float* Bs = smem + buf * STAGE + BM * LDK + (wn * WTN + r) * LDK + kq * 4;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 16 * LDK + ks * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 16 * LDK + ks * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (ch + 1 < nchunks) lds_store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
    const float gate = p.gate ? *p.gate : 0.f;
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) ssum[j] = ssq[j] = 0.f;

#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ng = n0g + wn * WTN + j * 16 + r;
        const bool n_ok = ng < cout_g;
        const int n = g * cout_g + ng;
        const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
        const float alpha = (p.alpha && n_ok) ? p.alpha[n] : 1.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + wm * WTM + i * 16 + kq * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[i][j][e] * alpha + bias;
                if (mb + e < M && n_ok) {
                    ssum[j] += v[e];
                    ssq[j] += v[e] * v[e];
                }
            }
            if (!n_ok) continue;
            if (p.out_mode == GSSD_OUT_TRANSPOSED) {
                if (mb < p.out_stride) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[e];
                        if (p.relu) t = fmaxf(t, 0.f);
                        o[e] = (mb + e < M) ? t : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(p.out + (size_t)img * p.out_batch_stride + (size_t)n * p.out_stride + mb) = o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = mb + e;
                    if (m >= M) continue;
                    float t = v[e];
                    if (p.out_mode == GSSD_OUT_HEADS) {
                        const int b = m / HoWo, pix = m - b * HoWo;
                        if (n < p.split_n)
                            p.out[(size_t)b * p.out_batch_stride + p.out_off + (size_t)pix * p.split_n + n] = t;
                        else
                            p.out_b[(size_t)b * p.outb_batch_stride + p.outb_off +
                                    (size_t)pix * (p.Cout - p.split_n) + (n - p.split_n)] = t;
                    } else {
                        const size_t o = (size_t)img * p.out_batch_stride + (size_t)m * p.out_stride + p.out_ch_off + n;
                        if (p.gate) {
                            t *= gate;
                            if (p.out2) p.out2[o] = t;
                            if (p.resid) t += p.resid[o];
                        } else if (p.resid) {
                            t += p.resid[o];
                        }
                        if (p.relu) t = fmaxf(t, 0.f);
                        p.out[o] = t;
                    }
                }
            }
        }
    }

    if (p.stats) {
        // per-channel sum / sum^2 over this tile's rows: lanes sharing (lane&15) -> LDS over wm -> fp64 atomics
        __syncthreads();
        float* red = smem;  // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = ssum[j], q = ssq[j];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (kq == 0) {
                red[(wm * BN + wn * WTN + j * 16 + r) * 2 + 0] = s;
                red[(wm * BN + wn * WTN + j * 16 + r) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0g + tid < cout_g) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s += (double)red[(w * BN + tid) * 2 + 0];
                q += (double)red[(w * BN + tid) * 2 + 1];
            }
            const int n = g * cout_g + n0g + tid;
            unsafeAtomicAdd(p.stats + n, s);
            unsafeAtomicAdd(p.stats + p.Cout + n, q);
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const gssd_conv_desc& d, int M, int images, hipStream_t stream) {
    static bool attr_set = false;
    constexpr size_t smem = 2 * (size_t)(BM + BN) * LDK * sizeof(float);
    auto kern = conv_igemm_kernel<BM, BN, WM, WN>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess) {
            gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %zu) failed", smem);
            return GSSD_ELAUNCH;
        }
        attr_set = true;
    }
    const int cout_g = d.Cout / d.groups;
    const int tiles = (cout_g + BN - 1) / BN;
    dim3 grid((M + BM - 1) / BM, d.groups * tiles, images);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, d, M, tiles);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

}  // namespace

extern "C" int gssd_conv2d_nhwc_f32(const gssd_conv_desc* dp, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dp != nullptr);
    const gssd_conv_desc& d = *dp;
    GSSD_CHECK_ARG(d.in && d.wgt && d.out);
    GSSD_CHECK_ARG(d.B > 0 && d.H > 0 && d.W > 0 && d.Ho > 0 && d.Wo > 0);
    GSSD_CHECK_ARG(d.groups > 0 && d.Cout > 0 && d.Cout % d.groups == 0);
    GSSD_CHECK_ARG(d.cin_g > 0 && d.cin_g % 4 == 0 && d.in_stride % 4 == 0 && d.in_ch_off % 4 == 0);
    GSSD_CHECK_ARG(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.dil > 0 && d.pad >= 0);
    GSSD_CHECK_ARG(d.K == d.KH * d.KW * d.cin_g && d.wgt_row_stride >= d.K && d.wgt_row_stride % 4 == 0);
    GSSD_CHECK_ARG(((uintptr_t)d.in % 16) == 0 && ((uintptr_t)d.wgt % 16) == 0);
    GSSD_CHECK_ARG(d.out_mode >= 0 && d.out_mode <= 2);
    if (d.out_mode == GSSD_OUT_TRANSPOSED) {
        GSSD_CHECK_ARG(d.m_per_image && d.out_stride % 4 == 0 && ((uintptr_t)d.out % 16) == 0);
        GSSD_CHECK_ARG(d.out_batch_stride % 4 == 0 && !d.gate && !d.resid);
    }
    if (d.out_mode == GSSD_OUT_HEADS) GSSD_CHECK_ARG(d.out_b && d.split_n > 0 && d.split_n < d.Cout && !d.m_per_image);
    if (d.m_per_image) GSSD_CHECK_ARG(d.in_batch_stride % 4 == 0 && d.wgt_batch_stride % 4 == 0);
    // the conv arithmetic must reproduce Ho/Wo
    GSSD_CHECK_ARG((d.H + 2 * d.pad - d.dil * (d.KH - 1) - 1) / d.stride + 1 == d.Ho);
    GSSD_CHECK_ARG((d.W + 2 * d.pad - d.dil * (d.KW - 1) - 1) / d.stride + 1 == d.Wo);

    const int images = d.m_per_image ? d.B : 1;
    const long long Mll = (long long)(d.m_per_image ? 1 : d.B) * d.Ho * d.Wo;
    GSSD_CHECK_ARG(Mll < (1ll << 31) && (long long)d.B * d.H * d.W < (1ll << 31));
    const int M = (int)Mll;
    const int cout_g = d.Cout / d.groups;
    hipStream_t s = as_stream(stream);
    if (cout_g > 64) return launch_cfg<128, 128, 2, 2>(d, M, images, s);
    if (cout_g > 32) return launch_cfg<128, 64, 2, 2>(d, M, images, s);
    if (cout_g > 16) return launch_cfg<128, 32, 4, 1>(d, M, images, s);
    return launch_cfg<128, 16, 4, 1>(d, M, images, s);
}
