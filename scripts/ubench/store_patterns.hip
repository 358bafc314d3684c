// Micro-benchmark: how fast can a wave's 16-byte-per-lane stores leave the CU, as a function of the lane -> address pattern?
// (epilogues of the bf16 trunk kernels write 32..128-byte runs per pixel; `fill_` writes lane-contiguous kilobytes)
//   hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Each wave-instruction writes 1 KB: RUN bytes contiguous per "pixel", pixels PITCH bytes apart.  ADJ = adjacent lanes hold adjacent
// 16-byte pieces of a run; !ADJ = the MFMA accumulator order (piece index = lane >> 4-ish, pixel = low lane bits).
template <int RUN, bool ADJ>
__global__ __launch_bounds__(256) void store_kernel(u32x4* __restrict__ out, long long pitch16, long long rows_per_wave, int iters) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int PPR = RUN / 16;            // 16-byte pieces per run
    constexpr int NPIX = 64 / PPR;           // runs (pixels) per instruction
    const int pix = ADJ ? lane / PPR : lane % NPIX;
    const int piece = ADJ ? lane % PPR : lane / NPIX;
    u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    for (int it = 0; it < iters; ++it) {
        // this wave's rows: [wave * rows_per_wave, ...): NPIX rows per instruction
        u32x4* base = out + (wave * rows_per_wave + (long long)it * NPIX + pix) * pitch16 + piece;
        *base = v;
        v.x += 64;
    }
}

template <int RUN, bool ADJ>
double run(u32x4* buf, size_t bytes, long long pitch, int reps) {
    constexpr int NPIX = 64 / (RUN / 16);
    const long long rows = (long long)(bytes / pitch);
    const int blocks = 256 * 8;
    const long long waves = blocks * 4;
    const long long rows_per_wave = rows / waves / NPIX * NPIX;
    const int iters = (int)(rows_per_wave / NPIX);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    store_kernel<RUN, ADJ><<<blocks, 256>>>(buf, pitch / 16, rows_per_wave, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) store_kernel<RUN, ADJ><<<blocks, 256>>>(buf, pitch / 16, rows_per_wave, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double written = (double)waves * iters * 1024.0 * reps;
    return written / (ms * 1e-3) / 1e9;
}

int main() {
    const size_t bytes = 1ull << 30;
    u32x4* buf;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    printf("pattern: bytes written per pixel run / pixel pitch / lane order -> GB/s of bytes actually written\n");
#define R(RUN, ADJ, PITCH) printf("run %4d B  pitch %5d B  %-9s %8.0f GB/s\n", RUN, PITCH, ADJ ? "adjacent" : "mfma", run<RUN, ADJ>(buf, bytes, PITCH, 5));
    R(1024, true, 1024)      // fully contiguous (fill-like)
    R(256, true, 256)        // contiguous, 4 rows per instruction
    R(256, true, 1024)       // 256-B runs, 1 KB apart (a quarter of each pixel vector)
    R(128, true, 128)
    R(128, true, 1024)       // full lines, pixels 1 KB apart
    R(128, true, 512)
    R(128, false, 1024)
    R(64, true, 1024)        // half lines, adjacent lanes
    R(64, false, 1024)       // half lines, accumulator order (conv_flat / conv_bf16 epilogue)
    R(64, true, 128)         // half lines of a 128-B pixel vector (the other half comes from another wave / later)
    R(64, false, 128)
    R(32, true, 128)         // conv1_2 thin: a wave owns 32 B of each 128-B pixel vector
    R(32, false, 128)
    R(32, true, 256)
    R(16, false, 128)
    return 0;
}
