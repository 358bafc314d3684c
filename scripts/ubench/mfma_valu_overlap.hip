// Micro-benchmark: how many independent VALU instructions hide in the shadow of one MFMA when a SIMD runs ONE wave
// (the situation of the 512-register fp32 kernels: conv_wino, dcn_fused)?  One workgroup of 256 threads per CU, every wave issues
//     [1 MFMA, N independent v_fma_f32] x 64, unrolled, 200 times
// and reports shader cycles per MFMA for N = 0 .. 12, for the fp32 MFMA (v_mfma_f32_16x16x4_f32, 32 cycles alone) and for a bf16 MFMA
// of the same length (v_mfma_f32_32x32x16_bf16, 32 cycles alone... 8 passes x 4).  If VALU work hides behind the MFMA the curve stays flat
// until ~7 fillers (28 of the 32 cycles); if the fp32 MFMA occupies the vector ALU's fp32 lanes it rises by ~4 cycles per filler from N = 1.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// FILL 0: v_fma_f32; 1: v_add_u32 (integer); 2 / 3: v_cndmask_b32 on VCC / on an SGPR pair (the padding selects); 4: v_mul_f32; 5: v_accvgpr_write; 6: v_max_f32
template <int N, int FILL>
__device__ __forceinline__ void fillers(float (&f)[12], float a, float b) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (FILL == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a), "v"(b));
        if (FILL == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(f[i]) : "v"(a));
        if (FILL == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(a) : "vcc");
        if (FILL == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f[i]) : "v"(a) : "s20", "s21");
        if (FILL == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "v"(a));
        if (FILL == 5) asm volatile("v_accvgpr_write_b32 a0, %0" : : "v"(f[i]) : "a0");
        if (FILL == 6) asm volatile("v_max_f32 %0, %1, %0" : "+v"(f[i]) : "v"(a));
    }
}

template <int N, bool BF16, int FILL = 0>
__global__ __launch_bounds__(256, 1) void overlap_kernel(float* __restrict__ out, unsigned long long* __restrict__ cycles, int iters) {
    const int lane = threadIdx.x & 63;
    float f[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) f[i] = (float)(lane + i);
    const float a = 1.0001f, b = 0.5f;
    f32x4 acc4[4];
    f32x16 acc16[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
    bf16x8 pa, pb;
#pragma unroll
    for (int e = 0; e < 8; ++e) pa[e] = (__bf16)(float)(lane & 3), pb[e] = (__bf16)1.f;
    const float qa = (float)(lane & 7), qb = 1.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (BF16) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[k & 1]) : "v"(pa), "v"(pb));
            } else {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[k & 3]) : "v"(qa), "v"(qb));
            }
            fillers<N, FILL>(f, a, b);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += f[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc4[i][0];
    s += acc16[0][0] + acc16[1][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int N, bool BF16, int FILL = 0>
double run(float* out, unsigned long long* cyc, int blocks) {
    const int iters = 200;
    hipLaunchKernelGGL((overlap_kernel<N, BF16, FILL>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((overlap_kernel<N, BF16, FILL>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / blocks / (iters * 64.0);
}

int main() {
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)cus * 256 * sizeof(float));
    hipMalloc(&cyc, (size_t)cus * sizeof(unsigned long long));
    printf("one wave per SIMD on %d CUs; shader cycles (s_memtime) per MFMA with N independent v_fma_f32 issued behind it\n", cus);
    printf("  N   v_mfma_f32_16x16x4_f32   v_mfma_f32_32x32x16_bf16\n");
    const double f[7] = {run<0, false>(out, cyc, cus), run<2, false>(out, cyc, cus), run<4, false>(out, cyc, cus), run<6, false>(out, cyc, cus),
                         run<8, false>(out, cyc, cus), run<10, false>(out, cyc, cus), run<12, false>(out, cyc, cus)};
    const double b[7] = {run<0, true>(out, cyc, cus), run<2, true>(out, cyc, cus), run<4, true>(out, cyc, cus), run<6, true>(out, cyc, cus),
                         run<8, true>(out, cyc, cus), run<10, true>(out, cyc, cus), run<12, true>(out, cyc, cus)};
    for (int i = 0; i < 7; ++i) printf(" %2d   %10.1f               %10.1f\n", 2 * i, f[i], b[i]);
    printf("fillers of other kinds behind the fp32 MFMA (N = 0, 4, 8, 12):\n");
    printf("  v_add_u32      %6.1f %6.1f %6.1f %6.1f\n", run<0, false, 1>(out, cyc, cus), run<4, false, 1>(out, cyc, cus), run<8, false, 1>(out, cyc, cus),
           run<12, false, 1>(out, cyc, cus));
    printf("  v_cndmask_b32  %6.1f %6.1f %6.1f %6.1f   (VCC)\n", run<0, false, 2>(out, cyc, cus), run<4, false, 2>(out, cyc, cus), run<8, false, 2>(out, cyc, cus),
           run<12, false, 2>(out, cyc, cus));
    printf("  v_cndmask e64  %6.1f %6.1f %6.1f %6.1f   (SGPR pair)\n", run<0, false, 3>(out, cyc, cus), run<4, false, 3>(out, cyc, cus), run<8, false, 3>(out, cyc, cus),
           run<12, false, 3>(out, cyc, cus));
    printf("  v_mul_f32      %6.1f %6.1f %6.1f %6.1f\n", run<0, false, 4>(out, cyc, cus), run<4, false, 4>(out, cyc, cus), run<8, false, 4>(out, cyc, cus),
           run<12, false, 4>(out, cyc, cus));
    printf("  v_max_f32      %6.1f %6.1f %6.1f %6.1f\n", run<0, false, 6>(out, cyc, cus), run<4, false, 6>(out, cyc, cus), run<8, false, 6>(out, cyc, cus),
           run<12, false, 6>(out, cyc, cus));
    printf("  v_accvgpr_write%6.1f %6.1f %6.1f %6.1f\n", run<0, false, 5>(out, cyc, cus), run<4, false, 5>(out, cyc, cus), run<8, false, 5>(out, cyc, cus),
           run<12, false, 5>(out, cyc, cus));
    printf("the same fillers behind the bf16 MFMA (N = 0, 4, 8, 12):\n");
    printf("  v_add_u32      %6.1f %6.1f %6.1f %6.1f\n", run<0, true, 1>(out, cyc, cus), run<4, true, 1>(out, cyc, cus), run<8, true, 1>(out, cyc, cus),
           run<12, true, 1>(out, cyc, cus));
    printf("  v_cndmask e64  %6.1f %6.1f %6.1f %6.1f\n", run<0, true, 3>(out, cyc, cus), run<4, true, 3>(out, cyc, cus), run<8, true, 3>(out, cyc, cus),
           run<12, true, 3>(out, cyc, cus));
    return 0;
}
