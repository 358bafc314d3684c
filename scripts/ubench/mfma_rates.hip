// Micro-benchmark (round 5): issue rate of the bf16 MFMA shapes of gfx950 on one SIMD (one wave, four independent accumulators round robin):
// is the legacy K = 16 form (v_mfma_f32_16x16x16_bf16, bf16x4 operands = 2 VGPRs) half the work at half the time of v_mfma_f32_16x16x32_bf16,
// or the same time?  Decides whether a three-plane Winograd kernel can keep conv_wino.hip's 16-channel chunk / lane = (tile, channel quad) layout.
//   hipcc --offload-arch=gfx950 -O3 mfma_rates.hip -o mfma_rates && ./mfma_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void rate_kernel(float* __restrict__ out, unsigned long long* __restrict__ cycles, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    f32x16 big[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) big[i][e] = 0.f;
    bf16x8 pa, pb;
#pragma unroll
    for (int e = 0; e < 8; ++e) pa[e] = (__bf16)(float)(lane & 3), pb[e] = (__bf16)1.f;
    s16x4 qa = {1, 2, 3, 4}, qb = {1, 1, 1, 1};
    f32x4 fa = {1.f, 2.f, 3.f, 4.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k & 3]) : "v"(pa), "v"(pb));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x16bf16_1k %0, %1, %2, %0" : "+v"(acc[k & 3]) : "v"(qa), "v"(qb));
            if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[k & 1]) : "v"(pa), "v"(pb));
            if (KIND == 3) asm volatile("v_mfma_f32_32x32x8bf16_1k %0, %1, %2, %0" : "+v"(big[k & 1]) : "v"(qa), "v"(qb));
            if (KIND == 4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[k & 3]) : "v"(fa[0]), "v"(fa[1]));
            // round 5 (dcn_x6): six MFMAs in a row into the SAME accumulator (the three-plane kernels' chains), and two chains alternating
            if (KIND == 5) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(k / 6) & 3]) : "v"(pa), "v"(pb));
            if (KIND == 6) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[((k / 12) & 1) * 2 + (k & 1)]) : "v"(pa), "v"(pb));
            if (KIND == 7) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(pa), "v"(pb));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    s += big[0][0] + big[1][0];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

static double g_last_ns = 0;       // wall-clock ns per MFMA of one wave (hipEvent over the second launch): what the cycle counter counts

template <int KIND, int THREADS>
double run(float* out, unsigned long long* cyc, int blocks) {
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<KIND, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<KIND, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, cyc, 20 * iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    g_last_ns = ms * 1e6 / (20.0 * iters * 64.0);
    hipLaunchKernelGGL((rate_kernel<KIND, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / blocks / (iters * 64.0);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)cus * 512 * 4);
    hipMalloc(&cyc, (size_t)cus * 8);
    printf("shader cycles per MFMA of one wave (one workgroup per CU on %d CUs); 256 threads = one wave per SIMD, 512 = two\n", cus);
    printf("  %-34s %8s %8s\n", "instruction", "256 thr", "512 thr");
    printf("  %-34s %8.1f %8.1f\n", "v_mfma_f32_16x16x32_bf16", run<0, 256>(out, cyc, cus), run<0, 512>(out, cyc, cus));
    printf("  %-34s %8.1f %8.1f\n", "v_mfma_f32_16x16x16_bf16 (K=16)", run<1, 256>(out, cyc, cus), run<1, 512>(out, cyc, cus));
    printf("  %-34s %8.1f %8.1f\n", "v_mfma_f32_32x32x16_bf16", run<2, 256>(out, cyc, cus), run<2, 512>(out, cyc, cus));
    printf("  %-34s %8.1f %8.1f\n", "v_mfma_f32_32x32x8_bf16_1k (K=8)", run<3, 256>(out, cyc, cus), run<3, 512>(out, cyc, cus));
    printf("  %-34s %8.1f %8.1f\n", "v_mfma_f32_16x16x4_f32", run<4, 256>(out, cyc, cus), run<4, 512>(out, cyc, cus));
    printf("  %-34s %8.1f %8.1f\n", "16x16x32_bf16, chains of 6", run<5, 256>(out, cyc, cus), run<5, 512>(out, cyc, cus));
    printf("  %-34s %8.1f %8.1f\n", "16x16x32_bf16, 2 chains alternating", run<6, 256>(out, cyc, cus), run<6, 512>(out, cyc, cus));
    printf("  %-34s %8.1f %8.1f\n", "16x16x32_bf16, one accumulator", run<7, 256>(out, cyc, cus), run<7, 512>(out, cyc, cus));
    const double c1 = run<0, 256>(out, cyc, cus), n1 = g_last_ns, c2 = run<0, 512>(out, cyc, cus), n2 = g_last_ns;
    printf("wall clock (hipEvent, 256000 MFMAs per wave): one wave per SIMD %.2f ns per MFMA (%.1f counts -> counter at %.2f GHz), two waves %.2f ns per MFMA of one wave (%.1f counts -> %.2f GHz)\n",
           n1, c1, c1 / n1, n2, c2, c2 / n2);
    printf("  -> MFMA rate per SIMD: %.2f / ns with one wave, %.2f / ns with two (peak 2.5 PFLOP/s = %.3f / ns per SIMD at 16384 FLOP each)\n", 1.0 / n1, 2.0 / n2,
           2.5e15 / 16384 / (cus * 4) / 1e9);
    return 0;
}
