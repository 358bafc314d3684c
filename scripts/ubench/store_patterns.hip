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

// The thin trunk kernels' pattern: the 4 waves of a workgroup each write ONE RUN-byte slice (a phase group's channels) of the same
// pixels' vectors (pitch = 4 * RUN), 16 pixels x (RUN / 16) pieces per instruction... QUART = true; or each wave writes whole vectors
// of a quarter of the pixels (QUART = false), same bytes per workgroup.
template <int RUN, bool QUART>
__global__ __launch_bounds__(256) void group_store_kernel(u32x4* __restrict__ out, long long rows_per_wg, int iters) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int PPR = RUN / 16, NPIX = 64 / PPR, PITCH16 = 4 * PPR;
    u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    const long long row0 = (long long)blockIdx.x * rows_per_wg;
    if (QUART) {
        const int pix = lane % NPIX, piece = lane / NPIX;
        for (int it = 0; it < iters; ++it) {
            out[(row0 + (long long)it * NPIX + pix) * PITCH16 + w * PPR + piece] = v;
            v.x += 64;
        }
    } else {
        // wave w owns rows [w * iters * NPIX / 4 ...): whole vectors, PITCH16 pieces per pixel, 64 / PITCH16 pixels per instruction
        constexpr int NP2 = 64 / PITCH16;
        const int pix = lane / PITCH16, piece = lane % PITCH16;
        for (int it = 0; it < iters; ++it) {
            out[(row0 + ((long long)w * iters + it) * NP2 + pix) * PITCH16 + piece] = v;
            v.x += 64;
        }
    }
}

template <int RUN, bool QUART>
double run_group(u32x4* buf, size_t bytes, int reps) {
    constexpr int PPR = RUN / 16, NPIX = 64 / PPR;
    const long long rows = (long long)(bytes / (4 * RUN));
    const int blocks = 256 * 8;
    const long long rows_per_wg = rows / blocks / (4 * NPIX) * (4 * NPIX);
    const int iters = QUART ? (int)(rows_per_wg / NPIX) : (int)(rows_per_wg / (64 / (4 * PPR)) / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    group_store_kernel<RUN, QUART><<<blocks, 256>>>(buf, rows_per_wg, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) group_store_kernel<RUN, QUART><<<blocks, 256>>>(buf, rows_per_wg, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 4 * iters * 1024.0 * reps / (ms * 1e-3) / 1e9;
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
    printf("4 waves of a workgroup covering the same pixel vectors:\n");
    printf("each wave one 32-B slice of every 128-B vector (thin conv1_x epilogue) %8.0f GB/s\n", run_group<32, true>(buf, bytes, 5));
    printf("each wave whole 128-B vectors of a quarter of the pixels               %8.0f GB/s\n", run_group<32, false>(buf, bytes, 5));
    printf("each wave one 64-B slice of every 256-B vector (conv2_x epilogue)      %8.0f GB/s\n", run_group<64, true>(buf, bytes, 5));
    printf("each wave whole 256-B vectors of a quarter of the pixels               %8.0f GB/s\n", run_group<64, false>(buf, bytes, 5));
    return 0;
}
