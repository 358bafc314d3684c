// Self_Attn backward building blocks for gfx950 (backward of layers/self_attn.py:46-89, which the
// reference gets from autograd over bmm / softmax / conv2d):
//   gssd_bgemm_f32            batched fp32 GEMM with independent transpose flags on the fp32 matrix cores -- the products of the
//                             attention backward that are not "activations x K-major weights":
//                               dA = d(ag) . g      dtheta = dS . phi      dphi = dS^T . theta      dg = A^T . d(ag)
//   gssd_softmax_bwd_rows_f32 dS = A * (dA - rowsum(A * dA)) in place (one wave per row)
//   gssd_sn_weight_grad_f32   spectral-norm chain rule: dW_orig = dW_eff / s - <dW_eff, W> / s^2 * u v^T  (u, v constants of the step;
//                             two launches: a multi-workgroup dot product, then the elementwise update)
//   gssd_scaled_transpose_f32 Wd[c][n] = W[n][c] * alpha[n]: the data-gradient weights of a spectrally normalised 1x1 conv
//   gssd_dot_f32              sum a[i] * b[i] into a double (the gradient of Self_Attn's scalar gate sigma)
//   gssd_bgemm_ex_f32         the same GEMM with an epilogue: exp(acc - lse[m]) rebuilds the probabilities from the forward's
//                             log-sum-exp (no QK^T conv + softmax pass), aux (acc - D[m]) turns dA into dS where it is produced
//   gssd_rowdot_f32           D_i = <d(ag)_i, ag_i> = rowsum(A o dA)
// The forward is flash-style and keeps no attention map, only the rows' log-sum-exp; the backward re-materialises A per block.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = 64, TN = 64, TK = 16, LDS_LD = TK + 4;     // 80-byte rows: conflict-free ds_read_b128 of 16 consecutive rows

struct BgemmParams {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, lda, ldb, ldc, transA, transB, accumulate;
    long long sA, sB, sC;
    float alpha;
    int mode;                 // 0: C = alpha acc;  1: C = exp(alpha acc - rowvec[m]);  2: C = aux[m][n] (alpha acc - rowvec[m])
    const float* rowvec;      // [batch][M]
    const float* aux;         // laid out like C
};

// element (row, k) of op(X): X[row*ld + k] (not transposed) or X[k*ld + row] (transposed).  A 64 x 16 tile is one 16-byte quad per
// thread: fetched into a register (fetch_tile) one k-step ahead of the MFMAs that use it, written to LDS (commit_tile) after them.
__device__ __forceinline__ f32x4 fetch_tile(const float* __restrict__ X, int ld, int trans, int row0, int nrows, int k0, int K, int tid) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!trans) {
        // contiguous along k: thread -> (row = tid / 4, k quad = tid % 4)
        const int row = tid >> 2, kq = (tid & 3) << 2;
        const int gr = row0 + row, gk = k0 + kq;
        if (gr < nrows && gk < K) {
            const float* p = X + (size_t)gr * ld + gk;
            if (gk + 3 < K && (((uintptr_t)p) & 15) == 0) v = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gk + e < K) v[e] = p[e];
            }
        }
    } else {
        // contiguous along the row index: thread -> (k = tid / 16, row quad = tid % 16)
        const int k = tid >> 4, rq = (tid & 15) << 2;
        const int gk = k0 + k, gr = row0 + rq;
        if (gk < K && gr < nrows) {
            const float* p = X + (size_t)gk * ld + gr;
            if (gr + 3 < nrows && (((uintptr_t)p) & 15) == 0) v = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gr + e < nrows) v[e] = p[e];
            }
        }
    }
    return v;
}

__device__ __forceinline__ void commit_tile(const f32x4 v, int trans, float* lds, int tid) {
    if (!trans) {
        const int row = tid >> 2, kq = (tid & 3) << 2;
        *reinterpret_cast<f32x4*>(lds + row * LDS_LD + kq) = v;
    } else {
        const int k = tid >> 4, rq = (tid & 15) << 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[(rq + e) * LDS_LD + k] = v[e];
    }
}

__global__ __launch_bounds__(256) void bgemm_kernel(const BgemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[TM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[TN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
    const int b = blockIdx.z;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const float* A = p.A + (size_t)b * p.sA;
    const float* B = p.B + (size_t)b * p.sB;
    float* C = p.C + (size_t)b * p.sC;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // op(B)[k][n]: stored B[k*ldb + n] (not transposed) = "row index n is the contiguous one" -> the transposed loader
    f32x4 ra = fetch_tile(A, p.lda, p.transA, m0, p.M, 0, p.K, tid);
    f32x4 rb = fetch_tile(B, p.ldb, !p.transB, n0, p.N, 0, p.K, tid);
    for (int k0 = 0; k0 < p.K; k0 += TK) {
        commit_tile(ra, p.transA, As, tid);
        commit_tile(rb, !p.transB, Bs, tid);
        __syncthreads();
        if (k0 + TK < p.K) {                      // the next k-step's quads travel while this one's MFMAs run
            ra = fetch_tile(A, p.lda, p.transA, m0, p.M, k0 + TK, p.K, tid);
            rb = fetch_tile(B, p.ldb, !p.transB, n0, p.N, k0 + TK, p.K, tid);
        }
        f32x4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + (wm * 32 + i * 16 + r) * LDS_LD + 4 * kq);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * 32 + j * 16 + r) * LDS_LD + 4 * kq);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + r;
            if (n >= p.N) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * 32 + i * 16 + kq * 4 + e;
                if (m >= p.M) continue;
                float* dst = C + (size_t)m * p.ldc + n;
                float v = acc[i][j][e] * p.alpha;
                if (p.mode == 1) v = __expf(v - p.rowvec[(size_t)b * p.M + m]);
                else if (p.mode == 2) v = p.aux[(size_t)b * p.sC + (size_t)m * p.ldc + n] * (v - p.rowvec[(size_t)b * p.M + m]);
                *dst = p.accumulate ? *dst + v : v;
            }
        }
}

// dS[row][j] = A[row][j] * (dA[row][j] - sum_j A[row][j] * dA[row][j]) written over dA; pad columns [n, stride) are zeroed
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ A, float* __restrict__ dA, long long rows, int n,
                                                               int stride) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long rI = wave0; rI < rows; rI += nwaves) {
        const float* a = A + rI * stride;
        float* d = dA + rI * stride;
        float s = 0.f;
        for (int c = lane; c < n; c += 64) s += a[c] * d[c];
        s = wave_sum(s);
        for (int c = lane; c < stride; c += 64) d[c] = c < n ? a[c] * (d[c] - s) : 0.f;
    }
}

// out[row] = sum_c a[row][c] * b[row][c]: D_i = <d(ag)_i, ag_i> = rowsum(A o dA) of the attention backward (one wave per row)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                     long long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= rows) return;
    const float* pa = a + row * C;
    const float* pb = b + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += pa[c] * pb[c];
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

// spectral-norm chain rule in two launches: (1) dot = <scale * dW_eff, W> (grid-stride partial sums, one fp64 atomic per
// workgroup into a caller-zeroed scalar); (2) dW = scale * dW_eff * is - dot * is^2 * u v^T
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ dweff, int ld_dw, const float* __restrict__ w,
                                                     const float* __restrict__ scale, int rows, int cols, double* __restrict__ dot) {
    __shared__ double red[4];
    const float sc = scale ? scale[0] : 1.f;
    double acc = 0.0;
    const int total = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int rr = i / cols, cc = i - rr * cols;
        acc += (double)(sc * dweff[(size_t)rr * ld_dw + cc]) * (double)w[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(dot, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sn_apply_kernel(const float* __restrict__ dweff, int ld_dw, const float* __restrict__ u,
                                                       const float* __restrict__ v, const float* __restrict__ inv_sigma,
                                                       const float* __restrict__ scale, const double* __restrict__ dot,
                                                       float* __restrict__ dw, int rows, int cols) {
    const float sc = scale ? scale[0] : 1.f;
    const float is = inv_sigma[0];
    const float k = (float)(dot[0] * (double)is * (double)is);
    const int total = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int rr = i / cols, cc = i - rr * cols;
        dw[i] = sc * dweff[(size_t)rr * ld_dw + cc] * is - k * u[rr] * v[cc];
    }
}

__global__ void scaled_transpose_kernel(const float* __restrict__ w, const float* __restrict__ alpha, float* __restrict__ out, int rows,
                                        int cols) {
    const long long total = (long long)rows * cols;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % rows), c = (int)(i / rows);          // out[c][n]
        out[i] = w[(size_t)n * cols + c] * (alpha ? alpha[n] : 1.f);
    }
}

// y[c][r] = bf16(x[r][c]): 64 x 64 tiles through LDS, 16-byte reads and writes.  The weight gradients of the bf16 storage mode run as NT
// GEMMs on the bf16 matrix cores over the TRANSPOSED operands (reduction index = pixel, contiguous): gssd/backward.py::_wgrad_nt_bf16
__global__ __launch_bounds__(256) void transpose_cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long long rows,
                                                                   int cols, long long ld_x, long long ld_y) {
    __shared__ float tile[64][65];
    const long long r0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64, tid = threadIdx.x;
    for (int i = tid; i < 64 * 16; i += 256) {             // 64 rows x 16 float4
        const int rr = i >> 4, q = i & 15;
        const long long r = r0 + rr;
        const int c = c0 + 4 * q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            if (c + 3 < cols && ((ld_x | (long long)c0) & 3) == 0) {
                const float4 t = *reinterpret_cast<const float4*>(x + r * ld_x + c);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c + e < cols) v[e] = x[r * ld_x + c + e];
            }
        }
        for (int e = 0; e < 4; ++e) tile[rr][4 * q + e] = v[e];
    }
    __syncthreads();
    for (int i = tid; i < 64 * 8; i += 256) {              // 64 output rows (c) x 8 pieces of 8 r
        const int cc = i >> 3, p = i & 7;
        const int c = c0 + cc;
        const long long r = r0 + 8 * p;
        if (c >= cols || r >= rows) continue;
        unsigned short h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = __builtin_bit_cast(unsigned short, (__bf16)tile[8 * p + e][cc]);
        unsigned short* dst = y + (long long)c * ld_y + r;
        if (r + 7 < rows) {
            uint4 v;
            v.x = h[0] | ((unsigned)h[1] << 16);
            v.y = h[2] | ((unsigned)h[3] << 16);
            v.z = h[4] | ((unsigned)h[5] << 16);
            v.w = h[6] | ((unsigned)h[7] << 16);
            *reinterpret_cast<uint4*>(dst) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (r + e < rows) dst[e] = h[e];
        }
    }
}

__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, double* out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc += (double)a[i] * (double)b[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, long long n, float a,
                             float b) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a * x[i] + b * y[i];
}

// y[i] = scale[0] * x[i] (x fp64 column sums): bias gradients behind Self_Attn's gate
__global__ void scale_cast_kernel(const double* __restrict__ x, const float* __restrict__ scale, float* __restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (float)((double)(scale ? scale[0] : 1.f) * x[i]);
}

// d(sigma) = dot + sum_c bias[c] * colsum[c]
__global__ __launch_bounds__(256) void sigma_grad_kernel(const double* __restrict__ dot, const double* __restrict__ colsum,
                                                         const float* __restrict__ bias, int C, float* __restrict__ dsigma) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) acc += (double)bias[c] * colsum[c];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dsigma[0] = (float)(dot[0] + red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

extern "C" int gssd_axpby_f32(const float* x, const float* y, float* out, int64_t n, float a, float b, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && out && n > 0);
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(axpby_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, y, out, (long long)n, a, b);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_scale_cast_f64_f32(const double* x, const float* scale, float* y, int n, gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(scale_cast_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), x, scale, y, n);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_sa_sigma_grad_f32(const double* dot, const double* colsum, const float* bias, int C, float* dsigma,
                                      gssd_stream_t stream) {
    GSSD_CHECK_ARG(dot && colsum && bias && dsigma && C > 0);
    hipLaunchKernelGGL(sigma_grad_kernel, dim3(1), dim3(256), 0, as_stream(stream), dot, colsum, bias, C, dsigma);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                              int transB, long long strideA, long long strideB, long long strideC, int batch, float alpha, int accumulate,
                              gssd_stream_t stream) {
    GSSD_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535);
    GSSD_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
    BgemmParams p{A, B, C, M, N, K, lda, ldb, ldc, transA, transB, accumulate, strideA, strideB, strideC, alpha, 0, nullptr, nullptr};
    hipLaunchKernelGGL(bgemm_kernel, dim3((M + TM - 1) / TM, (N + TN - 1) / TN, batch), dim3(256), 0, as_stream(stream), p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_bgemm_ex_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                                 int transB, long long strideA, long long strideB, long long strideC, int batch, float alpha, int mode,
                                 const float* rowvec, const float* aux, gssd_stream_t stream) {
    GSSD_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535);
    GSSD_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
    GSSD_CHECK_ARG((mode == 0) || (mode == 1 && rowvec) || (mode == 2 && rowvec && aux));
    BgemmParams p{A, B, C, M, N, K, lda, ldb, ldc, transA, transB, 0, strideA, strideB, strideC, alpha, mode, rowvec, aux};
    hipLaunchKernelGGL(bgemm_kernel, dim3((M + TM - 1) / TM, (N + TN - 1) / TN, batch), dim3(256), 0, as_stream(stream), p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_softmax_bwd_rows_f32(const float* attn, float* dattn, int64_t rows, int n, int row_stride, gssd_stream_t stream) {
    GSSD_CHECK_ARG(attn && dattn && rows > 0 && n > 0 && row_stride >= n);
    long long blocks = (rows + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), attn, dattn, (long long)rows, n,
                       row_stride);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_sn_weight_grad_f32(const float* dw_eff, int ld_dw, const float* w_orig, const float* u, const float* v,
                                       const float* inv_sigma, const float* scale, double* dot_scratch, float* dw_orig, int rows,
                                       int cols, gssd_stream_t stream) {
    GSSD_CHECK_ARG(dw_eff && w_orig && u && v && inv_sigma && dot_scratch && dw_orig && rows > 0 && cols > 0 && ld_dw >= cols);
    GSSD_CHECK_ARG((long long)rows * cols < (1ll << 31));
    int blocks = (rows * cols + 1023) / 1024;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(sn_dot_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dw_eff, ld_dw, w_orig, scale, rows, cols, dot_scratch);
    hipLaunchKernelGGL(sn_apply_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dw_eff, ld_dw, u, v, inv_sigma, scale,
                       dot_scratch, dw_orig, rows, cols);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_scaled_transpose_f32(const float* w, const float* alpha, float* out, int rows, int cols, gssd_stream_t stream) {
    GSSD_CHECK_ARG(w && out && rows > 0 && cols > 0);
    const long long total = (long long)rows * cols;
    hipLaunchKernelGGL(scaled_transpose_kernel, dim3((int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w, alpha, out, rows, cols);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_transpose_cast_f32_bf16(const float* x, void* y, int64_t rows, int cols, int64_t ld_x, int64_t ld_y,
                                            gssd_stream_t stream) {
    GSSD_CHECK_ARG(x && y && rows > 0 && cols > 0 && ld_x >= cols && ld_y >= rows && ld_y % 8 == 0 && ((uintptr_t)y % 16) == 0 &&
                   ((uintptr_t)x % 16) == 0);
    GSSD_CHECK_ARG((rows + 63) / 64 < (1ll << 31) && (cols + 63) / 64 < 65536);
    hipLaunchKernelGGL(transpose_cast_bf16_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64)), dim3(256), 0,
                       as_stream(stream), x, reinterpret_cast<unsigned short*>(y), (long long)rows, cols, (long long)ld_x, (long long)ld_y);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_rowdot_f32(const float* a, const float* b, float* out, int64_t rows, int C, gssd_stream_t stream) {
    GSSD_CHECK_ARG(a && b && out && rows > 0 && C > 0 && (rows + 3) / 4 < (1ll << 31));
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), a, b, out, rows, C);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_dot_f32(const float* a, const float* b, int64_t n, double* out, gssd_stream_t stream) {
    GSSD_CHECK_ARG(a && b && out && n > 0);
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(dot_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), a, b, (long long)n, out);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
