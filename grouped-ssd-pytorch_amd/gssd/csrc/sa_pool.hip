// Self_Attn with max_pool_factor > 1 (layers/self_attn.py:57-59, 67, 76): the keys (phi) and values (g) are average-pooled to a
// P x P grid, P = max(H / factor, 1), by F.adaptive_avg_pool2d -- cell o covers rows [floor(o H / P), ceil((o + 1) H / P)), so
// neighbouring cells overlap whenever P does not divide H.  Forward: pooled keys token-major [B][Nk][C8] and pooled values
// channel-major [B][C2][Nkp] (the layouts csrc/flash_attn.hip stages); backward: the transpose of the same averaging.
// HBM-bound elementwise passes over <= 47 MB (B = 32, 38 x 38 x 512 channels); not on the default path (factor 1 pools nothing).
#include "common.h"

namespace {

__device__ __forceinline__ int cell_start(int o, int H, int P) { return (o * H) / P; }
__device__ __forceinline__ int cell_end(int o, int H, int P) { return ((o + 1) * H + P - 1) / P; }

// keys: tp [B][N][2 C8] (theta | phi) -> kp [B][P P][C8]
__global__ __launch_bounds__(256) void sa_pool_keys_kernel(const float* __restrict__ tp, float* __restrict__ kp, long long total, int H,
                                                           int P, int C8) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C8);
    const int cell = (int)((idx / C8) % (P * P));
    const long long b = idx / ((long long)C8 * P * P);
    const int oy = cell / P, ox = cell - oy * P;
    const int y0 = cell_start(oy, H, P), y1 = cell_end(oy, H, P), x0 = cell_start(ox, H, P), x1 = cell_end(ox, H, P);
    const float* src = tp + b * (long long)H * H * 2 * C8 + C8 + c;
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += src[(long long)(y * H + x) * 2 * C8];
    kp[idx] = s / (float)((y1 - y0) * (x1 - x0));
}

// values: gT [B][C2][Np] -> gTp [B][C2][Nkp], pad columns [P P, Nkp) = 0 (the core multiplies them by probability 0)
__global__ __launch_bounds__(256) void sa_pool_values_kernel(const float* __restrict__ gT, float* __restrict__ gTp, long long total, int H,
                                                             int P, int Np, int Nkp) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cell = (int)(idx % Nkp);
    const long long row = idx / Nkp;                     // b * C2 + c
    if (cell >= P * P) {
        gTp[idx] = 0.f;
        return;
    }
    const int oy = cell / P, ox = cell - oy * P;
    const int y0 = cell_start(oy, H, P), y1 = cell_end(oy, H, P), x0 = cell_start(ox, H, P), x1 = cell_end(ox, H, P);
    const float* src = gT + row * Np;
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += src[y * H + x];
    gTp[idx] = s / (float)((y1 - y0) * (x1 - x0));
}

// backward: dkg [B][P P][CW] (d pooled phi | d pooled g, token-major) -> dst [B][N][ld] columns [0, CW):
// dst[n][j] = sum over the cells that contain token n of dkg[cell][j] / |cell|
__global__ __launch_bounds__(256) void sa_unpool_kernel(const float* __restrict__ dkg, float* __restrict__ dst, long long total, int H, int P,
                                                        int CW, int ld) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % CW);
    const int n = (int)((idx / CW) % (H * H));
    const long long b = idx / ((long long)CW * H * H);
    const int y = n / H, x = n - y * H;
    // candidates: start(o) <= y < end(o) only for o within one of floor(y P / H)
    const int oyc = (y * P) / H, oxc = (x * P) / H;
    float s = 0.f;
    for (int oy = oyc - 1; oy <= oyc + 1; ++oy) {
        if (oy < 0 || oy >= P) continue;
        const int y0 = cell_start(oy, H, P), y1 = cell_end(oy, H, P);
        if (y < y0 || y >= y1) continue;
        for (int ox = oxc - 1; ox <= oxc + 1; ++ox) {
            if (ox < 0 || ox >= P) continue;
            const int x0 = cell_start(ox, H, P), x1 = cell_end(ox, H, P);
            if (x < x0 || x >= x1) continue;
            s += dkg[(b * P * P + oy * P + ox) * CW + j] / (float)((y1 - y0) * (x1 - x0));
        }
    }
    dst[(b * H * H + n) * ld + j] = s;
}

}  // namespace

extern "C" int gssd_sa_pool_kv_f32(const float* tp, const float* gT, float* kp, float* gTp, int B, int H, int P, int C8, int C2, int Np,
                                   int Nkp, gssd_stream_t stream) {
    GSSD_CHECK_ARG(tp && gT && kp && gTp && B > 0 && H > 0 && P > 0 && P <= H && C8 > 0 && C2 > 0);
    GSSD_CHECK_ARG(Np >= H * H && Nkp >= P * P);
    hipStream_t s = as_stream(stream);
    const long long tk = (long long)B * P * P * C8, tv = (long long)B * C2 * Nkp;
    GSSD_CHECK_ARG((tk + 255) / 256 < (1ll << 31) && (tv + 255) / 256 < (1ll << 31));
    hipLaunchKernelGGL(sa_pool_keys_kernel, dim3((unsigned)((tk + 255) / 256)), dim3(256), 0, s, tp, kp, tk, H, P, C8);
    GSSD_CHECK_LAUNCH();
    hipLaunchKernelGGL(sa_pool_values_kernel, dim3((unsigned)((tv + 255) / 256)), dim3(256), 0, s, gT, gTp, tv, H, P, Np, Nkp);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_sa_unpool_f32(const float* dkg, float* dst, int B, int H, int P, int CW, int ld, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dkg && dst && B > 0 && H > 0 && P > 0 && P <= H && CW > 0 && ld >= CW);
    const long long total = (long long)B * H * H * CW;
    GSSD_CHECK_ARG((total + 255) / 256 < (1ll << 31));
    hipLaunchKernelGGL(sa_unpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), dkg, dst, total, H, P, CW,
                       ld);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
