// Do v_mfma_f32_16x16x32_bf16 of one wave and vector-ALU instructions of ANOTHER wave on the same SIMD overlap?  (round 6: conv_thin_x6's two
// resident workgroups ran as if MFMA time and VALU time added up.)  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue a stream
// of dependent / independent MFMAs, waves 4-7 (the second wave of each SIMD) a stream of v_fma_f32 (or v_cvt_pk / v_sub mixes).  Modes: MFMA
// waves alone, VALU waves alone, both.  If the two overlap, T(both) ~ max(T(mfma), T(valu)); if the SIMD serialises them, T(both) ~ the sum.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_two_waves mfma_valu_two_waves.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int valu_per_iter) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
        f32x4 c[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u % CHAINS], 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
        if (!(mode & 2)) return;
        float x[8];
        for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 0.01f + e;
        const float s = 1.0001f, t = 0.5f;
        // (fully unrolled per iteration: a branch every 8 instructions made the first version of this loop 10 cycles per instruction)
        if (valu_per_iter == 16) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[e & 7]) : "v"(s), "v"(t));
            }
        } else if (valu_per_iter == 32) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int e = 0; e < 32; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[e & 7]) : "v"(s), "v"(t));
            }
        } else if (valu_per_iter == 48) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int e = 0; e < 48; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[e & 7]) : "v"(s), "v"(t));
            }
        } else {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int e = 0; e < 64; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[e & 7]) : "v"(s), "v"(t));
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
    }
}

template <int CHAINS>
float run(float* out, int iters, int mode, int vpi) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<CHAINS><<<256, 512>>>(out, 100, mode, vpi);
    hipEventRecord(e0);
    k<CHAINS><<<256, 512>>>(out, iters, mode, vpi);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;         // 16 MFMAs per iteration per MFMA wave
    for (int vpi : {16, 32, 48, 64}) {
        const float tm1 = run<1>(out, iters, 1, vpi), tm4 = run<4>(out, iters, 1, vpi), tv = run<1>(out, iters, 2, vpi);
        const float tb1 = run<1>(out, iters, 3, vpi), tb4 = run<4>(out, iters, 3, vpi);
        printf("16 MFMAs (one wave) + %2d v_fma_f32 (the other wave) per iteration: MFMA alone %.0f us (1 chain) %.0f us (4 chains) = %.1f / %.1f cycles per MFMA at 2.4 GHz; "
               "VALU alone %.0f us = %.2f cycles per instruction; both %.0f us (1 chain) %.0f us (4 chains); sum %.0f, max %.0f\n",
               vpi, tm1, tm4, tm1 * 2400.f / (iters * 16.f), tm4 * 2400.f / (iters * 16.f), tv, tv * 2400.f / (iters * (float)vpi), tb1, tb4, tm4 + tv, tm4 > tv ? tm4 : tv);
    }
    return 0;
}
