// Micro-benchmark (round 4): the DATA-MOVEMENT SKELETON of the thin bf16 trunk layers (conv1_2 / conv2_1 / conv2_2) without the
// convolution -- which launch / tile / pipeline structure lets a CU move "patch in by LDS-DMA, slices out from registers" at the rate
// of a device copy (5.2 - 5.5 TB/s on this GPU)?  The real kernels run 1.3 - 2.1 TB/s of compulsory bytes although their traffic is
// clean (1.03 x); round 3's stamps say the loss is phase overlap.  Every variant moves the same bytes as the real layer:
//   in : B x H x W pixel vectors of IN_B bytes  (read once + halo)         out: B x H x W pixel vectors of OUT_B bytes
//   each of the 4 compute waves writes ITS quarter (a phase group's channels) of every output pixel vector, 16 bytes per lane
// Variants:
//   tile<TH>      : today's structure.  Persistent 256-thread workgroups over TH x 16 tiles: all 4 waves issue the patch DMA, barrier,
//                   [NM MFMAs per row], stores, barrier.
//   strip<TH,NS>  : 320-thread workgroups = 1 LOADER wave + 4 compute waves over a strip of TH rows, walking 16-column blocks through a
//                   ring of NS LDS stages: the loader only issues LDS-DMA (counted vmcnt: loads only), the compute waves only store and
//                   never wait on vmcnt; one s_barrier per block.
//   hipcc --offload-arch=gfx950 -O3 thin_skeleton.hip -o thin_skeleton && ./thin_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) unsigned g_zero[4] = {0, 0, 0, 0};

__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base,
                                     16, 0, 0);
}

struct P {
    const char* in;
    char* out;
    int B, H, W, nm;
};

constexpr int TW = 16;

// the compute waves' work for ROWS rows of a 16-pixel-wide block: per row optional MFMAs fed from LDS, then the wave's slice stores
template <int IN_B, int OUT_B>
__device__ __forceinline__ void rows_out(const P& p, const char* lds_rows, int lds_row_pitch, int b, int y0, int x0, int rows, int g, int lane) {
    constexpr int S = OUT_B / 4;          // slice bytes per pixel and wave
    constexpr int PP = S / 16;            // 16-byte pieces per slice
    constexpr int PIX = 64 / PP;          // pixels per store instruction
    constexpr int RPI = PIX / TW;         // rows per store instruction (2 for 32-byte slices, 1 for 64-byte slices)
    static_assert(RPI >= 1, "slice too wide");
    const int pix = lane % PIX, piece = lane / PIX;
    const int rr = pix / TW, xx = pix % TW;
    for (int r0 = 0; r0 < rows; r0 += RPI) {
        const int y = y0 + r0 + rr, x = x0 + xx;
        // something that depends on the staged patch (one b128 per lane and row) and on nm MFMAs
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds_rows + (r0 + rr + 1) * lds_row_pitch + (xx + 1) * IN_B + (g * (IN_B / 4)) % IN_B);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < p.nm; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc, 0, 0, 0);
        u32x4 v = {__builtin_bit_cast(unsigned, acc[0]), __builtin_bit_cast(unsigned, acc[1]), (unsigned)lane, (unsigned)y};
        if (y < p.H && x < p.W && r0 + rr < rows)
            *reinterpret_cast<u32x4*>(p.out + ((size_t)(b * p.H + y) * p.W + x) * OUT_B + g * S + piece * 16) = v;
    }
}

template <int IN_B, int OUT_B, int TH>
__global__ __launch_bounds__(256) void tile_kernel(const P p) {
    constexpr int PW = TW + 2, PH = TH + 2, NP = PH * PW;
    constexpr int UPR = IN_B / 16, PPI = 64 / UPR, NI = (NP + PPI - 1) / PPI;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = (p.H + TH - 1) / TH, tx = (p.W + TW - 1) / TW;
    const int ntiles = p.B * ty * tx;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (ty * tx), rem = tile - b * ty * tx;
        const int y0 = (rem / tx) * TH, x0 = (rem % tx) * TW;
        for (int i = g; i < NI; i += 4) {
            const int pp = i * PPI + lane / UPR;
            const int py = pp / PW, px = pp - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool ok = pp < NP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const char* src = ok ? p.in + ((size_t)(b * p.H + iy) * p.W + ix) * IN_B + (lane % UPR) * 16 : (const char*)g_zero;
            dma16(src, lds + i * 1024);
        }
        __syncthreads();
        rows_out<IN_B, OUT_B>(p, lds, PW * IN_B, b, y0, x0, TH, g, lane);
        __syncthreads();
    }
}

// strip: loader wave 4 + compute waves 0..3.  Stage = (TH + 2) rows x 16 columns of pixel vectors, row-major, [row][px][IN_B].
template <int IN_B, int OUT_B, int TH, int NS>
__global__ __launch_bounds__(320) void strip_kernel(const P p) {
    constexpr int PH = TH + 2;
    constexpr int IPR = TW * IN_B / 1024;                 // DMA instructions per stage row (2 KB rows at 128 B per pixel)
    static_assert(IPR >= 1, "row shorter than one DMA instruction");
    constexpr int NI = PH * IPR, STAGE = PH * TW * IN_B;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sy = (p.H + TH - 1) / TH, nbx = (p.W + TW - 1) / TW;
    const int nstrips = p.B * sy;
    // global block sequence of this workgroup: strips blockIdx.x, + gridDim.x, ... each nbx blocks; the ring runs across strip boundaries
    const int my_strips = (nstrips - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nblk = my_strips * nbx;
    auto issue = [&](int t) {           // loader: DMA of block t into stage t % NS
        const int s = t / nbx, bx = t - s * nbx;
        const int strip = blockIdx.x + s * gridDim.x;
        const int b = strip / sy, y0 = (strip - b * sy) * TH, x0 = bx * TW;
        char* dst = lds + (t % NS) * STAGE;
#pragma unroll 4
        for (int i = 0; i < NI; ++i) {
            const int row = i / IPR, part = i - row * IPR;
            const int iy = y0 - 1 + row;
            const int px = (part * 1024 + lane * 16) / IN_B;
            const bool ok = (unsigned)iy < (unsigned)p.H && x0 + px < p.W;
            const char* src = ok ? p.in + ((size_t)(b * p.H + iy) * p.W + x0) * IN_B + part * 1024 + lane * 16 : (const char*)g_zero;
            dma16(src, dst + i * 1024);
        }
    };
    if (wv == 4) {
        for (int t = 0; t < NS - 1 && t < nblk; ++t) issue(t);
        for (int t = 0; t < nblk; ++t) {
            // stage t must have landed before the barrier that opens iteration t: at most (NS - 2) younger stages may be in flight
            if constexpr (NS - 2 >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI < 63 ? 2 * NI : 63) : "memory");
            else if constexpr (NS - 2 == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI < 63 ? NI : 63) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // B_t: stage t visible to the compute waves; stage t - 1 is free
            if (t + NS - 1 < nblk) issue(t + NS - 1);
        }
    } else {
        for (int t = 0; t < nblk; ++t) {
            __builtin_amdgcn_s_barrier();
            const int s = t / nbx, bx = t - s * nbx;
            const int strip = blockIdx.x + s * gridDim.x;
            const int b = strip / sy, y0 = (strip - b * sy) * TH, x0 = bx * TW;
            // (the skeleton reads its own stage only; the real kernel also reads the neighbouring stages' edge columns)
            rows_out<IN_B, OUT_B>(p, lds + (t % NS) * STAGE - IN_B, TW * IN_B, b, y0, x0, TH, wv, lane);
        }
    }
}

static double time_it(void (*launch)(const P&, int), const P& p, int grid, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch(p, grid);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(p, grid);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) return -1;
    return ms / reps;
}

template <int IN_B, int OUT_B, int TH>
void launch_tile(const P& p, int grid) {
    constexpr int PW = TW + 2, PH = TH + 2, NP = PH * PW, UPR = IN_B / 16, PPI = 64 / UPR, NI = (NP + PPI - 1) / PPI;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel<IN_B, OUT_B, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, NI * 1024);
        once = true;
    }
    hipLaunchKernelGGL((tile_kernel<IN_B, OUT_B, TH>), dim3(grid), dim3(256), NI * 1024, 0, p);
}
template <int IN_B, int OUT_B, int TH, int NS>
void launch_strip(const P& p, int grid) {
    constexpr int STAGE = (TH + 2) * TW * IN_B;
    if (NS * STAGE + 1024 > 160 * 1024) return;            // does not fit the CU's LDS: reported as 0
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(strip_kernel<IN_B, OUT_B, TH, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE + 1024);
        once = true;
    }
    hipLaunchKernelGGL((strip_kernel<IN_B, OUT_B, TH, NS>), dim3(grid), dim3(320), NS * STAGE + 1024, 0, p);
}

template <int IN_B, int OUT_B>
void layer(const char* name, int B, int H, char* in, char* out) {
    P p{in, out, B, H, H, 0};
    const double bytes = (double)B * H * H * (IN_B + OUT_B);
    printf("== %s: B %d, %d x %d, %d B in + %d B out per pixel = %.0f MB\n", name, B, H, H, IN_B, OUT_B, bytes / 1e6);
    for (int nm : {0, 40, 120}) {
        p.nm = nm;
        printf(" -- %d MFMAs (16x16x32 bf16) per wave and 16-pixel row\n", nm);
        for (int wpc : {1, 2, 3, 4}) {
            double t8 = time_it(launch_tile<IN_B, OUT_B, 8>, p, 256 * wpc, 10);
            double t16 = time_it(launch_tile<IN_B, OUT_B, 16>, p, 256 * wpc, 10);
            printf("    tile  8x16 / 16x16, %d workgroups per CU:              %7.1f us %6.0f GB/s | %7.1f us %6.0f GB/s\n", wpc, t8 * 1e3,
                   bytes / t8 / 1e6, t16 * 1e3, bytes / t16 / 1e6);
        }
        for (int wpc : {1, 2, 3, 4}) {
            double a = time_it(launch_strip<IN_B, OUT_B, 8, 3>, p, 256 * wpc, 10);
            double b = time_it(launch_strip<IN_B, OUT_B, 8, 4>, p, 256 * wpc, 10);
            double c = time_it(launch_strip<IN_B, OUT_B, 4, 4>, p, 256 * wpc, 10);
            double d = time_it(launch_strip<IN_B, OUT_B, 16, 3>, p, 256 * wpc, 10);
            printf("    strip (TH, stages) (8,3) (8,4) (4,4) (16,3), %d workgroups per CU: %7.1f us %6.0f | %7.1f us %6.0f | %7.1f us %6.0f | %7.1f us %6.0f GB/s\n",
                   wpc, a * 1e3, bytes / a / 1e6, b * 1e3, bytes / b / 1e6, c * 1e3, bytes / c / 1e6, d * 1e3, bytes / d / 1e6);
        }
    }
}

__global__ void copy_kernel(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
    const int B = 32;
    size_t in_bytes = (size_t)B * 300 * 300 * 128, out_bytes = (size_t)B * 300 * 300 * 128 + (size_t)B * 150 * 150 * 256;
    char *in, *out;
    hipMalloc(&in, in_bytes);
    hipMalloc(&out, out_bytes);
    hipMemset(in, 1, in_bytes);
    hipMemset(out, 0, out_bytes);
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const size_t n = in_bytes / 16;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, n);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, (const u32x4*)in, (u32x4*)out, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("device copy of %.0f MB: %.1f us = %.0f GB/s (read + written)\n", in_bytes / 1e6, ms * 100, 2.0 * in_bytes / (ms / 10) / 1e6);
    }
    layer<128, 128>("conv1_2-like (64 -> 64 channels bf16, 300 x 300)", B, 300, in, out);
    layer<128, 256>("conv2_1-like (64 -> 128 channels bf16, 150 x 150)", B, 150, in, out);
    layer<256, 256>("conv2_2-like (128 -> 128 channels bf16, 150 x 150)", B, 150, in, out);
    return 0;
}
