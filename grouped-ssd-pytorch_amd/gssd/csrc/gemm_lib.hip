// Plain library GEMMs (rocBLAS) for the two contractions of the path that are nothing but a GEMM: the deformable conv's
// 9216-deep product over the sampled column matrix (forward, data gradient, weight gradient).  Everything with a fused
// epilogue or an implicit-GEMM loader stays on the hand-written kernels; rocBLAS reaches 137-141 TFLOP/s on these shapes
// where conv_igemm reaches 110.  One handle per process (one process per GPU), re-pointed at the caller's stream per call.
#include <rocblas/rocblas.h>

#include "common.h"

namespace {

rocblas_handle g_handle = nullptr;

int ensure_handle(hipStream_t stream) {
    if (!g_handle && rocblas_create_handle(&g_handle) != rocblas_status_success) {
        g_handle = nullptr;
        gssd_set_error("rocblas_create_handle failed");
        return GSSD_ELAUNCH;
    }
    if (rocblas_set_stream(g_handle, stream) != rocblas_status_success) {
        gssd_set_error("rocblas_set_stream failed");
        return GSSD_ELAUNCH;
    }
    return GSSD_OK;
}

__global__ void fill_rows_kernel(float* __restrict__ c, const float* __restrict__ bias, long long rows, int n, int ldc) {
    const long long total = rows * (n >> 2);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / (n >> 2);
        const int q = (int)(i - row * (n >> 2));
        reinterpret_cast<float4*>(c + row * ldc)[q] = reinterpret_cast<const float4*>(bias)[q];
    }
}

}  // namespace

// C[M][N] (row stride ldc) = A[M][K] (lda) . B[N][K]^T (ldb) (+ bias[N] broadcast over rows) (+ C if accumulate)
extern "C" int gssd_gemm_nt_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                const float* bias, int accumulate, gssd_stream_t stream) {
    GSSD_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc >= N);
    GSSD_CHECK_ARG(!(bias && accumulate) && (!bias || (N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C % 16) == 0)));
    hipStream_t s = as_stream(stream);
    const int rc = ensure_handle(s);
    if (rc != GSSD_OK) return rc;
    float beta = accumulate ? 1.f : 0.f;
    if (bias) {
        long long blocks = ((long long)M * (N / 4) + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(fill_rows_kernel, dim3((int)blocks), dim3(256), 0, s, C, bias, (long long)M, N, ldc);
        GSSD_CHECK_LAUNCH();
        beta = 1.f;
    }
    const float alpha = 1.f;
    // row-major C = A B^T  <=>  column-major C^T (N x M) = B (as op_T of its K x N image) . A's K x M image
    if (rocblas_sgemm(g_handle, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &alpha, B, ldb, A, lda, &beta, C,
                      ldc) != rocblas_status_success) {
        gssd_set_error("rocblas_sgemm (nt) failed");
        return GSSD_ELAUNCH;
    }
    return GSSD_OK;
}

// C[N][K2] (row stride ldc) (+)= A[M][N]^T (lda) . B[M][K2] (ldb): the reduction runs over the rows of both operands
// (weight gradient of a 1x1 contraction: A = dY, B = the layer input)
extern "C" int gssd_gemm_tn_f32(const float* A, const float* B, float* C, int M, int N, int K2, int lda, int ldb, int ldc,
                                int accumulate, gssd_stream_t stream) {
    GSSD_CHECK_ARG(A && B && C && M > 0 && N > 0 && K2 > 0 && lda >= N && ldb >= K2 && ldc >= K2);
    hipStream_t s = as_stream(stream);
    const int rc = ensure_handle(s);
    if (rc != GSSD_OK) return rc;
    const float alpha = 1.f, beta = accumulate ? 1.f : 0.f;
    // column-major C^T (K2 x N) = B's K2 x M image . (A's N x M image)^T
    if (rocblas_sgemm(g_handle, rocblas_operation_none, rocblas_operation_transpose, K2, N, M, &alpha, B, ldb, A, lda, &beta, C,
                      ldc) != rocblas_status_success) {
        gssd_set_error("rocblas_sgemm (tn) failed");
        return GSSD_ELAUNCH;
    }
    return GSSD_OK;
}
