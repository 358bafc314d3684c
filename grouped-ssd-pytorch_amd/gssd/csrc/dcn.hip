// DCNv2 sampling for gfx950: modulated, bilinearly sampled column matrix of a 3x3 / stride 1 / pad 1
// deformable convolution (the gather half of the reference's external `dcn_v2` extension called at
// layers/dcn_v2_custom.py:84-89; algorithm restated in oracle/gssd_oracle.py::dcn_v2_conv --
// parity unpinned, see there).  One wave per (pixel, tap, deformable group): the sampling position and
// the four bilinear weights are wave-uniform, each lane moves float4 channel slices, so every global
// access is a contiguous 16 B x 64 lane burst in NHWC.  The contraction with the weights is a plain
// 1x1 implicit GEMM over the columns (conv_igemm.hip).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                         float* __restrict__ cols, int B, int H, int W, int C, int dg,
                                                         int om_stride) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int HW = H * W, cpg = C / dg, cpg4 = cpg >> 2;
    const long long units = (long long)B * HW * 9 * dg;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (long long u = wave0; u < units; u += nwaves) {
        const int d = (int)(u % dg);
        long long t = u / dg;
        const int tap = (int)(t % 9);
        const long long bp = t / 9;            // b*HW + p
        const int p = (int)(bp % HW);
        const int b = (int)(bp / HW);
        const int h = p / W, w = p - h * W;
        const float* omp = om + bp * om_stride;
        const float dy = omp[d * 18 + 2 * tap];
        const float dx = omp[d * 18 + 2 * tap + 1];
        const float ml = omp[dg * 18 + d * 9 + tap];
        const float m = 1.f / (1.f + expf(-ml));            // torch.sigmoid (dcn_v2_custom.py:83)
        const float py = (float)(h - 1 + tap / 3) + dy;
        const float px = (float)(w - 1 + tap % 3) + dx;
        float* dst = cols + (bp * 9 + tap) * C + d * cpg;
        if (!(py > -1.f && px > -1.f && py < (float)H && px < (float)W)) {
            for (int c = lane; c < cpg4; c += 64) reinterpret_cast<f32x4*>(dst)[c] = zero4;
            continue;
        }
        const float y0f = floorf(py), x0f = floorf(px);
        const int y0 = (int)y0f, x0 = (int)x0f;
        const float ly = py - y0f, lx = px - x0f, hy = 1.f - ly, hx = 1.f - lx;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
        const float w00 = (y0ok && x0ok) ? hy * hx * m : 0.f;
        const float w01 = (y0ok && x1ok) ? hy * lx * m : 0.f;
        const float w10 = (y1ok && x0ok) ? ly * hx * m : 0.f;
        const float w11 = (y1ok && x1ok) ? ly * lx * m : 0.f;
        const float* xb = x + (size_t)b * HW * C + d * cpg;
        const f32x4* p00 = reinterpret_cast<const f32x4*>(xb + (size_t)((y0ok ? y0 : 0) * W + (x0ok ? x0 : 0)) * C);
        const f32x4* p01 = reinterpret_cast<const f32x4*>(xb + (size_t)((y0ok ? y0 : 0) * W + (x1ok ? x0 + 1 : 0)) * C);
        const f32x4* p10 = reinterpret_cast<const f32x4*>(xb + (size_t)((y1ok ? y0 + 1 : 0) * W + (x0ok ? x0 : 0)) * C);
        const f32x4* p11 = reinterpret_cast<const f32x4*>(xb + (size_t)((y1ok ? y0 + 1 : 0) * W + (x1ok ? x0 + 1 : 0)) * C);
        for (int c = lane; c < cpg4; c += 64) {
            const f32x4 v = p00[c] * w00 + p01[c] * w01 + p10[c] * w10 + p11[c] * w11;
            reinterpret_cast<f32x4*>(dst)[c] = v;
        }
    }
}

// Backward of the sampling: one wave per (pixel, tap, deformable group) again.  d(cols) arrives from the dgrad of the 1x1
// GEMM; the wave scatters d(x) to the four corners with fp32 atomics (contiguous 16 B x 64 lane bursts) and reduces the
// three scalars d(offset_y), d(offset_x), d(mask logit) over its channel slice.  The corner validity masks and floor()
// are constants of the differentiation, exactly as in the gather formulation (autograd of the oracle graph).
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const float* __restrict__ x, const float* __restrict__ om,
                                                         const float* __restrict__ dcols, float* __restrict__ dx,
                                                         float* __restrict__ dom, int B, int H, int W, int C, int dg,
                                                         int om_stride) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int HW = H * W, cpg = C / dg, cpg4 = cpg >> 2;
    const long long units = (long long)B * HW * 9 * dg;
    for (long long u = wave0; u < units; u += nwaves) {
        const int d = (int)(u % dg);
        long long t = u / dg;
        const int tap = (int)(t % 9);
        const long long bp = t / 9;
        const int p = (int)(bp % HW);
        const int b = (int)(bp / HW);
        const int h = p / W, w = p - h * W;
        const float* omp = om + bp * om_stride;
        float* domp = dom + bp * om_stride;
        const float oy = omp[d * 18 + 2 * tap];
        const float ox = omp[d * 18 + 2 * tap + 1];
        const float ml = omp[dg * 18 + d * 9 + tap];
        const float m = 1.f / (1.f + expf(-ml));
        const float py = (float)(h - 1 + tap / 3) + oy;
        const float px = (float)(w - 1 + tap % 3) + ox;
        if (!(py > -1.f && px > -1.f && py < (float)H && px < (float)W)) {
            if (lane == 0) {
                domp[d * 18 + 2 * tap] = 0.f;
                domp[d * 18 + 2 * tap + 1] = 0.f;
                domp[dg * 18 + d * 9 + tap] = 0.f;
            }
            continue;
        }
        const float y0f = floorf(py), x0f = floorf(px);
        const int y0 = (int)y0f, x0 = (int)x0f;
        const float ly = py - y0f, lx = px - x0f, hy = 1.f - ly, hx = 1.f - lx;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1, x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
        const float k00 = (y0ok && x0ok) ? 1.f : 0.f, k01 = (y0ok && x1ok) ? 1.f : 0.f;
        const float k10 = (y1ok && x0ok) ? 1.f : 0.f, k11 = (y1ok && x1ok) ? 1.f : 0.f;
        const size_t o00 = (size_t)((y0ok ? y0 : 0) * W + (x0ok ? x0 : 0)) * C;
        const size_t o01 = (size_t)((y0ok ? y0 : 0) * W + (x1ok ? x0 + 1 : 0)) * C;
        const size_t o10 = (size_t)((y1ok ? y0 + 1 : 0) * W + (x0ok ? x0 : 0)) * C;
        const size_t o11 = (size_t)((y1ok ? y0 + 1 : 0) * W + (x1ok ? x0 + 1 : 0)) * C;
        const size_t base = (size_t)b * HW * C + d * cpg;
        const float* xb = x + base;
        float* dxb = dx + base;
        const float* gsrc = dcols + (bp * 9 + tap) * C + d * cpg;
        float s_m = 0.f, s_y = 0.f, s_x = 0.f;
        for (int c = lane; c < cpg4; c += 64) {
            const f32x4 g = reinterpret_cast<const f32x4*>(gsrc)[c];
            const f32x4 v00 = reinterpret_cast<const f32x4*>(xb + o00)[c] * k00;
            const f32x4 v01 = reinterpret_cast<const f32x4*>(xb + o01)[c] * k01;
            const f32x4 v10 = reinterpret_cast<const f32x4*>(xb + o10)[c] * k10;
            const f32x4 v11 = reinterpret_cast<const f32x4*>(xb + o11)[c] * k11;
            const f32x4 val = v00 * (hy * hx) + v01 * (hy * lx) + v10 * (ly * hx) + v11 * (ly * lx);
            const f32x4 gy = (v10 - v00) * hx + (v11 - v01) * lx;
            const f32x4 gx = (v01 - v00) * hy + (v11 - v10) * ly;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s_m += g[e] * val[e];
                s_y += g[e] * gy[e];
                s_x += g[e] * gx[e];
            }
            const f32x4 gm = g * m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (k00 != 0.f) unsafeAtomicAdd(dxb + o00 + 4 * c + e, gm[e] * (hy * hx));
                if (k01 != 0.f) unsafeAtomicAdd(dxb + o01 + 4 * c + e, gm[e] * (hy * lx));
                if (k10 != 0.f) unsafeAtomicAdd(dxb + o10 + 4 * c + e, gm[e] * (ly * hx));
                if (k11 != 0.f) unsafeAtomicAdd(dxb + o11 + 4 * c + e, gm[e] * (ly * lx));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s_m += __shfl_xor(s_m, o);
            s_y += __shfl_xor(s_y, o);
            s_x += __shfl_xor(s_x, o);
        }
        if (lane == 0) {
            domp[d * 18 + 2 * tap] = s_y * m;
            domp[d * 18 + 2 * tap + 1] = s_x * m;
            domp[dg * 18 + d * 9 + tap] = s_m * m * (1.f - m);
        }
    }
}

}  // namespace

extern "C" int gssd_dcn_im2col_f32(const float* x, const float* om, float* cols, int B, int H, int W, int C, int dg,
                                   int om_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && cols && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0);
    GSSD_CHECK_ARG(C % (4 * dg) == 0 && om_stride >= 27 * dg);
    const long long units = (long long)B * H * W * 9 * dg;
    long long blocks = (units + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dcn_im2col_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, om, cols, B, H, W, C, dg,
                       om_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dcn_col2im_f32(const float* x, const float* om, const float* dcols, float* dx, float* dom, int B, int H,
                                   int W, int C, int dg, int om_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && om && dcols && dx && dom && B > 0 && H > 0 && W > 0 && C > 0 && dg > 0);
    GSSD_CHECK_ARG(C % (4 * dg) == 0 && om_stride >= 27 * dg);
    const long long units = (long long)B * H * W * 9 * dg;
    long long blocks = (units + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dcn_col2im_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, om, dcols, dx, dom, B, H, W, C,
                       dg, om_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
