// Micro-benchmark (round 5, for csrc/dcn_x6.hip): what does one CU get out of L2 through the vector memory path, per shader cycle?
//   mode 0  global_load_dwordx4, a wave reads 1 KiB contiguous (the weight planes' pieces as plain loads)
//   mode 1  global_load_lds_dwordx4 of the same pieces (LDS DMA, the way the x6 kernels stage their weight planes)
//   mode 2  global_load_dwordx4, quads of lanes read 128 contiguous bytes at scattered 4 KiB-strided pixels (dcn_x6's corner loads: 2 x 16 B per lane)
//   mode 3  as mode 2 with the quad's pieces contiguous: 64 B per quad and instruction (lane & 3) * 16, second instruction + 64
// Every wave issues `batch` loads, waits for all of them, repeats; the footprint per CU is `span` bytes (default 96 KiB: L2 hits, beyond the 32 KiB L1).
//   hipcc --offload-arch=gfx950 -O3 vmem_rates.hip -o vmem_rates && ./vmem_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int BATCH>
__global__ __launch_bounds__(512, 1) void vmem_kernel(const float* __restrict__ src, float* __restrict__ out, int iters, int span_bytes, int shared_src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = reinterpret_cast<const char*>(src) + (shared_src ? 0 : (size_t)blockIdx.x * span_bytes);
    f32x4 v[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    unsigned off = (unsigned)(wave * BATCH) * 1024u;
    const unsigned span = (unsigned)span_bytes;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            unsigned o = off + (unsigned)b * 1024u;
            if (o >= span) o -= span;
            if (MODE == 0) {
                const char* p = base + o + lane * 16;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[b]) : "v"(p) : "memory");
            } else if (MODE == 1) {
                const char* p = base + o + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * BATCH + b) * 256), 16, 0, 0);
            } else if (MODE == 3) {
                // the same 16 pixels per instruction, but the quad's four 16-byte pieces are CONTIGUOUS (64 B; b odd: the line's second half)
                unsigned pix = (o >> 11) + (unsigned)(lane >> 2) * 7u;
                unsigned a = (pix * 4096u) % span + (unsigned)(lane & 3) * 16u + (unsigned)(b & 1) * 64u;
                const char* p = base + a;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[b]) : "v"(p) : "memory");
            } else {
                // 16 pixels per instruction (lane >> 2), 4 KiB apart, 32 B per lane in two instructions (b even / odd)
                unsigned pix = (o >> 11) + (unsigned)(lane >> 2) * 7u;
                unsigned a = (pix * 4096u) % span + (unsigned)(lane & 3) * 32u + (unsigned)(b & 1) * 16u;
                const char* p = base + a;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[b]) : "v"(p) : "memory");
            }
        }
        off += (unsigned)(nw * BATCH) * 1024u;
        if (off >= span) off -= span;
        if (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int b = 0; b < BATCH; ++b) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[b]));
#pragma unroll
            for (int b = 0; b < BATCH; ++b) s += v[b];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + (MODE == 1 ? lds[threadIdx.x] : 0.f);
}

template <int MODE, int BATCH>
void run(const char* name, const float* src, float* out, int cus, int threads, int span, int shared_src, double ghz) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(vmem_kernel<MODE, BATCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL((vmem_kernel<MODE, BATCH>), dim3(cus), dim3(threads), 128 * 1024, 0, src, out, 50, span, shared_src);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((vmem_kernel<MODE, BATCH>), dim3(cus), dim3(threads), 128 * 1024, 0, src, out, iters, span, shared_src);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = (double)iters * (threads / 64) * BATCH * 1024.0;
    printf("  %-44s %d waves/CU batch %2d span %4d KiB %s: %6.1f B/ns per CU = %5.1f B/clk at %.2f GHz, chip %5.2f TB/s\n", name, threads / 64, BATCH, span / 1024,
           shared_src ? "shared " : "per CU ", bytes_cu / (ms * 1e6), bytes_cu / (ms * 1e6) / ghz, ghz, bytes_cu * cus / (ms * 1e6) / 1e3);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = 2.3;
    const size_t total = (size_t)cus * 1024 * 1024;
    float *src, *out;
    hipMalloc(&src, total);
    hipMemset(src, 0, total);
    hipMalloc(&out, (size_t)cus * 512 * 4);
    for (int shared_src = 0; shared_src < 2; ++shared_src)
        for (int span : {24 * 1024, 96 * 1024, 1024 * 1024}) {
            for (int threads : {256, 512}) {
                run<0, 12>("global_load_dwordx4 contiguous", src, out, cus, threads, span, shared_src, ghz);
                run<1, 12>("global_load_lds_dwordx4 contiguous", src, out, cus, threads, span, shared_src, ghz);
                run<2, 16>("global_load_dwordx4 128 B per quad, scattered", src, out, cus, threads, span, shared_src, ghz);
                run<3, 16>("global_load_dwordx4 64 B contiguous per quad", src, out, cus, threads, span, shared_src, ghz);
            }
        }
    run<0, 4>("global_load_dwordx4 contiguous", src, out, cus, 256, 96 * 1024, 0, ghz);
    run<1, 4>("global_load_lds_dwordx4 contiguous", src, out, cus, 256, 96 * 1024, 0, ghz);
    run<1, 24>("global_load_lds_dwordx4 contiguous", src, out, cus, 256, 96 * 1024, 0, ghz);
    return 0;
}
