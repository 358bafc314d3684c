// Micro-benchmark (round 4, for csrc/conv_x6.hip / dcn_x6.hip): which vector instructions hide in the shadow of v_mfma_f32_16x16x32_bf16
// (16 cycles per SIMD)?  Every wave issues [1 MFMA (four independent accumulators round robin), N independent fillers] x 64, unrolled,
// 200 times; one or two waves per SIMD (256 / 512 threads, one workgroup per CU).  Reports shader cycles per MFMA of ONE wave: flat at 16
// (one wave) or 32 (two waves sharing the pipe) while the fillers hide, rising by the filler's cost once they do not.
//   hipcc --offload-arch=gfx950 -O3 mfma16_valu_overlap.hip -o mfma16_valu_overlap && ./mfma16_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// FILL 0 v_fma_f32, 1 v_cvt_pk_bf16_f32, 2 v_pk_add_f32, 3 v_pk_fma_f32, 4 v_sub_f32, 5 v_lshlrev_b32, 6 v_and_b32, 7 v_cndmask_b32 (SGPR pair), 8 v_perm_b32
template <int N, int FILL>
__device__ __forceinline__ void fillers(f32x2 (&f)[8], float a, float b, f32x4 (&g_ld)[4], const float* g_ptr, const float* g_ptr2, unsigned g_lds) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (FILL == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i][0]) : "v"(a), "v"(b));
        if (FILL == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(f[i][1]) : "v"(f[i][0]), "v"(a));
        if (FILL == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (FILL == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        if (FILL == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i][0]) : "v"(a));
        if (FILL == 5) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(f[i][1]) : "v"(f[i][0]));
        if (FILL == 6) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(f[i][1]) : "v"(f[i][0]));
        if (FILL == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f[i][0]) : "v"(a) : "s20", "s21");
        if (FILL == 8) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(f[i][1]) : "v"(f[i][0]), "v"(a), "v"(b));
        // round 5: is the 8-cycle cost of the bit / convert rows the instruction or the out-of-place destination of this test?
        if (FILL == 9) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(f[i][0]));
        if (FILL == 10) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[i][1]) : "v"(f[i][0]), "v"(a));
        if (FILL == 11) asm volatile("v_and_b32 %0, %1, %0" : "+v"(f[i][0]) : "v"(a));
        if (FILL == 12) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[i][0]) : "v"(a), "v"(b));
        if (FILL == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[i][0]) : "v"(a));
        if (FILL == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i][1]) : "v"(f[i][0]));
        if (FILL == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[i][0]) : "v"(a));
        if (FILL == 16) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(f[i][0]) : "v"(a), "v"(b));
        if (FILL == 17) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f[i][0]) : "v"(a));
        if (FILL == 18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i][0]) : "v"(a));
        if (FILL == 19) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i][0]) : "v"(a));
        if (FILL == 20) asm volatile("v_fma_mix_f32 %0, %1, %2, %0" : "+v"(f[i][0]) : "v"(a), "v"(b));
        if (FILL == 21) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(f[i][0]) : "v"(a), "v"(b));
        // round 5 (dcn_x6): memory instructions behind an MFMA -- an L1-resident 1-KiB load / LDS DMA piece per wave, LDS reads / writes
        if (FILL == 22) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g_ld[i & 3]) : "v"(g_ptr) : "memory");
        if (FILL == 23) asm volatile("ds_read_b128 %0, %1" : "=v"(g_ld[i & 3]) : "v"(g_lds) : "memory");
        if (FILL == 24) asm volatile("ds_write_b64 %0, %1" : : "v"(g_lds), "v"(f[i]) : "memory");
        if (FILL == 25) asm volatile("global_load_lds_dwordx4 %0, off" : : "v"(g_ptr) : "memory");
        if (FILL == 26) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g_ld[i & 3]) : "v"(g_ptr2) : "memory");
    }
}

template <int N, int FILL, int THREADS, int EVERY = 1, int ACCS = 4>
__global__ __launch_bounds__(THREADS, 1) void overlap_kernel(float* __restrict__ out, unsigned long long* __restrict__ cycles, int iters) {
    const int lane = threadIdx.x & 63;
    f32x2 f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = f32x2{(float)(lane + i), 1.f};
    const float a = 1.0001f, b = 0.5f;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 pa, pb;
#pragma unroll
    for (int e = 0; e < 8; ++e) pa[e] = (__bf16)(float)(lane & 3), pb[e] = (__bf16)1.f;
    f32x4 g_ld[4];
    __shared__ __attribute__((aligned(16))) float lds_buf[8 * 64 * 4 * 2];
    const float* g_ptr = out + (size_t)(blockIdx.x * THREADS + threadIdx.x) * 4;                         // 1 KiB contiguous per wave
    const float* g_ptr2 = out + (size_t)blockIdx.x * THREADS * 4 + (threadIdx.x >> 6) * 8192 + (lane >> 2) * 1024 + (lane & 3) * 8;   // 128 B per quad, 4 KiB apart
    const unsigned g_lds = (unsigned)(size_t)(lds_buf) + threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %0" : : "s"(__builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_buf + (threadIdx.x >> 6) * 1024)));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[ACCS == 6 ? (k / 6) & 3 : k & (ACCS - 1)]) : "v"(pa), "v"(pb));
            if (k % EVERY == 0) fillers<N, FILL>(f, a, b, g_ld, g_ptr, g_ptr2, g_lds);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(g_ld[0]), "+v"(g_ld[1]), "+v"(g_ld[2]), "+v"(g_ld[3]));
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (FILL >= 22 && FILL != 24 && FILL != 25) ? g_ld[0][0] + g_ld[1][0] + g_ld[2][0] + g_ld[3][0] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i][0] + f[i][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int N, int FILL, int THREADS, int EVERY = 1, int ACCS = 4>
double run(float* out, unsigned long long* cyc, int blocks) {
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((overlap_kernel<N, FILL, THREADS, EVERY, ACCS>), dim3(blocks), dim3(THREADS), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / blocks / (iters * 64.0);
}

template <int FILL, int THREADS>
void row(const char* name, float* out, unsigned long long* cyc, int cus) {
    printf("  %-20s %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", name, run<0, FILL, THREADS>(out, cyc, cus), run<1, FILL, THREADS>(out, cyc, cus),
           run<2, FILL, THREADS>(out, cyc, cus), run<3, FILL, THREADS>(out, cyc, cus), run<4, FILL, THREADS>(out, cyc, cus), run<6, FILL, THREADS>(out, cyc, cus));
}

template <int THREADS>
void table(float* out, unsigned long long* cyc, int cus) {
    printf("%d wave(s) per SIMD: cycles per v_mfma_f32_16x16x32_bf16 of one wave with N = 0 1 2 3 4 6 fillers behind it\n", THREADS / 256);
    row<0, THREADS>("v_fma_f32", out, cyc, cus);
    row<1, THREADS>("v_cvt_pk_bf16_f32", out, cyc, cus);
    row<2, THREADS>("v_pk_add_f32", out, cyc, cus);
    row<3, THREADS>("v_pk_fma_f32", out, cyc, cus);
    row<4, THREADS>("v_sub_f32", out, cyc, cus);
    row<5, THREADS>("v_lshlrev_b32", out, cyc, cus);
    row<6, THREADS>("v_and_b32", out, cyc, cus);
    row<7, THREADS>("v_cndmask_b32 (sgpr)", out, cyc, cus);
    row<8, THREADS>("v_perm_b32", out, cyc, cus);
    row<9, THREADS>("v_lshlrev_b32 inplace", out, cyc, cus);
    row<10, THREADS>("v_sub_f32 outofplace", out, cyc, cus);
    row<11, THREADS>("v_and_b32 inplace", out, cyc, cus);
    row<12, THREADS>("v_dot2c_f32_bf16", out, cyc, cus);
    row<13, THREADS>("v_cvt_pk inplace", out, cyc, cus);
    row<14, THREADS>("v_mov_b32", out, cyc, cus);
    row<15, THREADS>("v_add_u32", out, cyc, cus);
    row<16, THREADS>("v_and_or_b32", out, cyc, cus);
    row<17, THREADS>("v_xor_b32", out, cyc, cus);
    row<18, THREADS>("v_max_f32", out, cyc, cus);
    row<19, THREADS>("v_mul_f32", out, cyc, cus);
    row<20, THREADS>("v_fma_mix_f32", out, cyc, cus);
    row<21, THREADS>("v_bfi_b32", out, cyc, cus);
    row<22, THREADS>("global_load_dwordx4 1KiB", out, cyc, cus);
    row<26, THREADS>("global_load_dwordx4 quad", out, cyc, cus);
    row<25, THREADS>("global_load_lds_dwordx4", out, cyc, cus);
    row<23, THREADS>("ds_read_b128", out, cyc, cus);
    row<24, THREADS>("ds_write_b64", out, cyc, cus);
}

// round 5: is v - float(bf16(v)) through v_dot2c_f32_bf16 (acc = v; acc += h * -1 + h' * 0) the exact residual the shift + subtract gives?
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void dot2_exact_kernel(const float* __restrict__ v, unsigned* __restrict__ bad, int n, unsigned ca, unsigned cb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned ph = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);
    float da = a, db = b;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(da) : "v"(ph), "v"(ca));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(db) : "v"(ph), "v"(cb));
    if (__builtin_bit_cast(unsigned, da) != __builtin_bit_cast(unsigned, ra)) atomicAdd(&bad[0], 1u);
    if (__builtin_bit_cast(unsigned, db) != __builtin_bit_cast(unsigned, rb)) atomicAdd(&bad[1], 1u);
    // second level: the residual of the residual
    const unsigned pm = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
    const float sa = ra - __builtin_bit_cast(float, pm << 16), sb = rb - __builtin_bit_cast(float, pm & 0xffff0000u);
    float ea = ra, eb = rb;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(ea) : "v"(pm), "v"(ca));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(eb) : "v"(pm), "v"(cb));
    if (__builtin_bit_cast(unsigned, ea) != __builtin_bit_cast(unsigned, sa)) atomicAdd(&bad[2], 1u);
    if (__builtin_bit_cast(unsigned, eb) != __builtin_bit_cast(unsigned, sb)) atomicAdd(&bad[3], 1u);
}

void dot2_exact() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        unsigned bits = (unsigned)(s >> 32);
        if (i % 3 == 0) {                                    // any finite bit pattern incl. subnormals
            if (((bits >> 23) & 0xff) == 0xff) bits &= ~(1u << 30);
            h[i] = __builtin_bit_cast(float, bits);
        } else {                                             // activations: ~N(0, 1) scale values
            h[i] = ((int)(bits >> 8) - (1 << 23)) * (4.0f / (1 << 23)) * ((i % 3 == 1) ? 1.f : 1e-3f);
        }
    }
    float* d;
    unsigned* bad;
    hipMalloc(&d, (size_t)n * sizeof(float));
    hipMalloc(&bad, 4 * sizeof(unsigned));
    hipMemset(bad, 0, 4 * sizeof(unsigned));
    hipMemcpy(d, h.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(dot2_exact_kernel, dim3(n / 2 / 256), dim3(256), 0, 0, d, bad, n, 0x0000BF80u, 0xBF800000u);
    unsigned hb[4];
    hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
    printf("v_dot2c_f32_bf16 residual vs shift + v_sub_f32 on %d values (1/3 arbitrary bit patterns): mismatches level 1 lo %u hi %u, level 2 lo %u hi %u\n", n, hb[0],
           hb[1], hb[2], hb[3]);
}

template <int FILL, int THREADS>
void sparse_row(const char* name, float* out, unsigned long long* cyc, int cus) {
    printf("  %-26s %6.1f %6.1f %6.1f %6.1f %6.1f\n", name, run<1, FILL, THREADS, 2>(out, cyc, cus), run<1, FILL, THREADS, 4>(out, cyc, cus),
           run<1, FILL, THREADS, 8>(out, cyc, cus), run<2, FILL, THREADS, 8>(out, cyc, cus), run<1, FILL, THREADS, 16>(out, cyc, cus));
}

template <int FILL, int ACCS>
void chain_row(const char* name, float* out, unsigned long long* cyc, int cus) {
    printf("  %-40s %6.1f %6.1f %6.1f %6.1f %6.1f\n", name, run<0, FILL, 256, 1, ACCS>(out, cyc, cus), run<1, FILL, 256, 1, ACCS>(out, cyc, cus),
           run<2, FILL, 256, 1, ACCS>(out, cyc, cus), run<3, FILL, 256, 1, ACCS>(out, cyc, cus), run<4, FILL, 256, 1, ACCS>(out, cyc, cus));
}

int main() {
    dot2_exact();
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)cus * 512 * 64 * sizeof(float));
    hipMemset(out, 0, (size_t)cus * 512 * 64 * sizeof(float));
    hipMalloc(&cyc, (size_t)cus * sizeof(unsigned long long));
    table<256>(out, cyc, cus);
    table<512>(out, cyc, cus);
    printf("one wave per SIMD, N = 0 1 2 3 4 fillers behind every MFMA, by how the MFMAs use their accumulators: cycles per MFMA\n");
    chain_row<0, 1>("v_fma_f32, ONE accumulator", out, cyc, cus);
    chain_row<0, 2>("v_fma_f32, two alternating", out, cyc, cus);
    chain_row<0, 6>("v_fma_f32, chains of six", out, cyc, cus);
    chain_row<0, 4>("v_fma_f32, four round robin", out, cyc, cus);
    chain_row<13, 1>("v_cvt_pk_bf16_f32, ONE accumulator", out, cyc, cus);
    chain_row<13, 6>("v_cvt_pk_bf16_f32, chains of six", out, cyc, cus);
    chain_row<23, 6>("ds_read_b128, chains of six", out, cyc, cus);
    printf("one wave per SIMD, memory instructions behind every 2nd / 4th / 8th / (two behind every) 8th / 16th MFMA: cycles per MFMA (16.2 = hidden)\n");
    sparse_row<22, 256>("global_load_dwordx4 1KiB", out, cyc, cus);
    sparse_row<26, 256>("global_load_dwordx4 quad", out, cyc, cus);
    sparse_row<25, 256>("global_load_lds_dwordx4", out, cyc, cus);
    sparse_row<23, 256>("ds_read_b128", out, cyc, cus);
    sparse_row<24, 256>("ds_write_b64", out, cyc, cus);
    return 0;
}
