// Device-side input stage for gfx950: what the reference does on the CPU in its DataLoader for every
// 4-phase study slice -- per phase `Image.fromarray(u8).resize((size, size))` (Pillow), `-= mean`, min-max normalise over
// the whole study, and the [4, H, W, 3] -> [12, H, W] channel order (data/__init__.py:33-54, utils/augmentations.py:506-524,
// train_lesion_multiphase_v2.py:198).  The resampler restates Pillow's 8-bit path (src/libImaging/Resample.c): double
// coefficients normalised per output pixel and quantised to 22-bit fixed point ON THE HOST (gssd_resample_coeffs), integer
// accumulation from 1 << 21, arithmetic shift, clip to [0, 255]; horizontal pass, then vertical, uint8 in between.  Pure
// byte / integer work: bit-exact against Pillow.  All three kernels are HBM-streaming (100 MB in, 138 MB out at B = 32); the
// LDS-staged forms further down are the ones the 512 -> 300 path runs, the first two kernels the fallback for odd row lengths.
// Compiled with -ffp-contract=off: the host coefficient code and the fp32 normalisation must round like the reference.
#include <math.h>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

double filter_bilinear(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}
double filter_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

struct Geometry {
    double scale, filterscale, support;
    int ksize;
};
Geometry geometry(int in_size, int out_size, int filter) {
    Geometry g;
    g.scale = g.filterscale = (double)in_size / out_size;
    if (g.filterscale < 1.0) g.filterscale = 1.0;
    g.support = (filter == GSSD_FILTER_BILINEAR ? 1.0 : 2.0) * g.filterscale;
    g.ksize = (int)ceil(g.support) * 2 + 1;
    return g;
}

__device__ __forceinline__ uint8_t clip8(int acc) {
    const int v = acc >> PRECISION_BITS;                    // arithmetic shift, like the C reference
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// in [n][H][Win][C] -> out [n][H][Wout][C]; one thread per output pixel (its C bytes are contiguous in both images)
template <int C>
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                       long long rows, int Win, int Wout) {
    const long long total = rows * Wout;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wout);
        const long long row = i / Wout;
        const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
        const uint8_t* src = in + (row * Win + x0) * C;
        const int* k = kk + xo * ksize;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += __mul24((int)src[t * C + c], w);     // 8-bit x 23-bit: exact in 24-bit multiplies
        }
#pragma unroll
        for (int c = 0; c < C; ++c) out[i * C + c] = clip8(acc[c]);
    }
}

// in [n][Hin][W][C] -> out [n][Hout][W][C]; a thread owns 4 consecutive bytes of a row (one 32-bit access per tap, coalesced
// along the row; the taps are rows apart) and walks V_ROWS output rows.  Also folds the study's per-channel min / max of the
// result into mm[study][C][2] = (max of 255 - v, max of v): zero-filled by the caller, so both are integer atomicMax.
constexpr int V_ROWS = 10;

template <typename WordT>
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                       int n_img, int Hin, int Hout, int W, int C, int imgs_per_study,
                                                       int* __restrict__ mm) {
    constexpr int NB = sizeof(WordT);                                   // 4: rows are a whole number of 32-bit words; else 1
    const int rowlen = W * C, words = rowlen / NB;
    const int img = blockIdx.z;
    int lo[4] = {255, 255, 255, 255}, hi[4] = {0, 0, 0, 0};          // per channel (C <= 4), this thread's bytes
    const int wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (wi < words) {
        const int c0 = (NB * wi) % C;
        for (int yo = blockIdx.y * V_ROWS; yo < min(Hout, (blockIdx.y + 1) * V_ROWS); ++yo) {
            const int y0 = bounds[2 * yo], n = bounds[2 * yo + 1];
            const int* k = kk + yo * ksize;
            const WordT* src = reinterpret_cast<const WordT*>(in + ((long long)img * Hin + y0) * rowlen) + wi;
            int acc[NB];
#pragma unroll
            for (int e = 0; e < NB; ++e) acc[e] = 1 << (PRECISION_BITS - 1);
            for (int t = 0; t < n; ++t) {
                const uint32_t v = src[(long long)t * words];
                const int w = k[t];
#pragma unroll
                for (int e = 0; e < NB; ++e) acc[e] += __mul24((int)((v >> (8 * e)) & 255u), w);
            }
            uint32_t r = 0;
            int c = c0;
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int b = clip8(acc[e]);
                r |= (uint32_t)b << (8 * e);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q == c) {
                        lo[q] = min(lo[q], b);
                        hi[q] = max(hi[q], b);
                    }
                c = (c + 1 == C) ? 0 : c + 1;
            }
            reinterpret_cast<WordT*>(out + ((long long)img * Hout + yo) * rowlen)[wi] = (WordT)r;
        }
    }
    if (mm) {
        // one workgroup-level result, then at most 2 C atomics by 2 C lanes in parallel: with one (load, atomic) chain per wave and
        // channel the six round trips to the hot L2 lines at the end of every workgroup were most of this kernel's time (0.34 ms
        // at B = 32, growing with the number of workgroups: 0.96 ms with 3 rows per workgroup, 0.16 ms with 30)
        __shared__ int red[4][8];
        for (int q = 0; q < C; ++q) {
            int l = lo[q], h = hi[q];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                l = min(l, __shfl_xor(l, o, 64));
                h = max(h, __shfl_xor(h, o, 64));
            }
            if ((threadIdx.x & 63) == 0) {
                red[threadIdx.x >> 6][2 * q] = h >= l ? 255 - l : 0;
                red[threadIdx.x >> 6][2 * q + 1] = h >= l ? h : 0;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * C) {
            int* m = mm + (img / imgs_per_study) * C * 2 + threadIdx.x;
            const int v = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
            if (v > __atomic_load_n(m, __ATOMIC_RELAXED)) atomicMax(m, v);      // (a stale read is fine, the atomic decides)
        }
    }
}

// The horizontal pass for rows that are whole 32-bit words: a workgroup stages H_ROWS input rows (coalesced words) and the
// coefficient table in LDS; a thread then produces one 32-bit word of an output row from LDS bytes (0.25 -> 0.18 ms at B = 32).
// Measured and rejected: one thread per output pixel on the same staging (0.26 ms), and a vertical pass that first gathers a
// thread's whole input window into its own LDS column with batched independent loads (0.45 - 0.51 ms against 0.34 ms for the plain
// kernel above: the four workgroups per CU that fit beside the 32 KB column array hide less latency than the eight without it).
constexpr int H_ROWS = 16;

template <int C>
__global__ __launch_bounds__(256) void resize_h_lds_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                           long long rows, int Win, int Wout) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hsm[];
    const int inw = Win * C / 4, outw = Wout * C / 4, tid = threadIdx.x;
    const uint8_t* rowb = reinterpret_cast<const uint8_t*>(hsm);
    int* kl = reinterpret_cast<int*>(hsm + H_ROWS * inw);
    int* bl = kl + Wout * ksize;
    const long long row0 = (long long)blockIdx.x * H_ROWS;
    const int nr = (int)(rows - row0 < H_ROWS ? rows - row0 : H_ROWS);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in + row0 * Win * C);
    for (int i = tid; i < nr * inw; i += 256) hsm[i] = src[i];
    for (int i = tid; i < Wout * ksize; i += 256) kl[i] = kk[i];
    for (int i = tid; i < 2 * Wout; i += 256) bl[i] = bounds[i];
    __syncthreads();
    for (int item = tid; item < nr * outw; item += 256) {                   // one 32-bit word of an output row
        const int r = item / outw, w = item - r * outw;
        uint32_t res = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int b = 4 * w + e;
            const int xo = b / C, c = b - xo * C;
            const int x0 = bl[2 * xo], n = bl[2 * xo + 1];
            const int* k = kl + xo * ksize;
            const uint8_t* sp = rowb + (size_t)r * Win * C + x0 * C + c;
            int acc = 1 << (PRECISION_BITS - 1);
            for (int t = 0; t < n; ++t) acc += __mul24((int)sp[t * C], k[t]);      // 8-bit x 23-bit: exact in the 24-bit multiplier
            res |= (uint32_t)clip8(acc) << (8 * e);
        }
        reinterpret_cast<uint32_t*>(out + (row0 + r) * Wout * C)[w] = res;
    }
}

// u8 [B][P][S][S][C] -> fp32 NCHW [B][P*C][S][S]:  x = (float)v * scale - mean[c];  normalise: (x - lo) / (hi - lo) with
// lo / hi the study's extrema of x (fp32, same operations as the reference's numpy expression).
__global__ __launch_bounds__(256) void input_finish_kernel(const uint8_t* __restrict__ img, const int* __restrict__ mm,
                                                           float m0, float m1, float m2, float* __restrict__ out, int B,
                                                           int P, int S, int C, int normalize) {
    const int HW = S * S;
    const long long total = (long long)B * P * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long long bp = i / HW;
        const int b = (int)(bp / P);
        const float mean[3] = {m0, m1, m2};
        float lo = 0.f, den = 1.f;
        if (normalize) {
            const int* m = mm + b * C * 2;
            float hi = -INFINITY;
            lo = INFINITY;
            for (int c = 0; c < C; ++c) {
                lo = fminf(lo, (float)(255 - m[2 * c]) - mean[c]);
                hi = fmaxf(hi, (float)m[2 * c + 1] - mean[c]);
            }
            den = hi - lo;
        }
        const uint8_t* src = img + i * C;
        for (int c = 0; c < C; ++c) {
            float x = (float)src[c] - mean[c];
            if (normalize) x = __fdiv_rn(x - lo, den);
            out[(bp * C + c) * HW + pix] = x;
        }
    }
}

}  // namespace

extern "C" int gssd_resample_ksize(int in_size, int out_size, int filter) {
    if (in_size <= 0 || out_size <= 0 || (filter != GSSD_FILTER_BILINEAR && filter != GSSD_FILTER_BICUBIC)) return -1;
    return geometry(in_size, out_size, filter).ksize;
}

extern "C" int gssd_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk) {
    GSSD_CHECK_ARG(in_size > 0 && out_size > 0 && bounds && kk);
    GSSD_CHECK_ARG(filter == GSSD_FILTER_BILINEAR || filter == GSSD_FILTER_BICUBIC);
    const Geometry g = geometry(in_size, out_size, filter);
    double (*f)(double) = filter == GSSD_FILTER_BILINEAR ? filter_bilinear : filter_bicubic;
    const double ss = 1.0 / g.filterscale;
    double* w = new double[g.ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * g.scale;
        int xmin = (int)(center - g.support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + g.support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            w[x] = f((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int32_t* k = kk + (size_t)xx * g.ksize;
        for (int x = 0; x < g.ksize; ++x) {
            double v = 0.0;
            if (x < xmax) v = (ww != 0.0) ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int32_t)(-0.5 + v * (1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
    return GSSD_OK;
}

extern "C" int gssd_resize_u8_horizontal(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize,
                                         int n_img, int H, int W_in, int W_out, int C, gssd_stream_t stream) {
    GSSD_CHECK_ARG(in && out && bounds && kk && ksize > 0 && n_img > 0 && H > 0 && W_in > 0 && W_out > 0 && C > 0);
    GSSD_CHECK_ARG(C <= 4);
    const long long rows = (long long)n_img * H;
    long long blocks = (rows * W_out + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const size_t lds = (size_t)H_ROWS * W_in * C + (size_t)(W_out * ksize + 2 * W_out) * sizeof(int);
    if ((W_in * C) % 4 == 0 && (W_out * C) % 4 == 0 && ((uintptr_t)in % 4) == 0 && ((uintptr_t)out % 4) == 0 && lds <= 60 * 1024 &&
        (rows + H_ROWS - 1) / H_ROWS < (1ll << 31)) {
        auto kl = C == 1 ? resize_h_lds_kernel<1> : C == 2 ? resize_h_lds_kernel<2> : C == 3 ? resize_h_lds_kernel<3> : resize_h_lds_kernel<4>;
        hipLaunchKernelGGL(kl, dim3((unsigned)((rows + H_ROWS - 1) / H_ROWS)), dim3(256), lds, as_stream(stream), in, out, bounds, kk, ksize,
                           rows, W_in, W_out);
        GSSD_CHECK_LAUNCH();
        return GSSD_OK;
    }
    auto kern = C == 1 ? resize_h_kernel<1> : C == 2 ? resize_h_kernel<2> : C == 3 ? resize_h_kernel<3> : resize_h_kernel<4>;
    hipLaunchKernelGGL(kern, dim3((int)blocks), dim3(256), 0, as_stream(stream), in, out, bounds, kk, ksize, rows, W_in, W_out);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_resize_u8_vertical(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize,
                                       int n_img, int H_in, int H_out, int W, int C, int imgs_per_study, int32_t* minmax,
                                       gssd_stream_t stream) {
    GSSD_CHECK_ARG(in && out && bounds && kk && ksize > 0 && n_img > 0 && H_in > 0 && H_out > 0 && W > 0);
    GSSD_CHECK_ARG(C > 0 && C <= 4 && imgs_per_study > 0 && n_img % imgs_per_study == 0 && n_img <= 65535 && H_out <= 65535);
    const bool words = (W * C) % 4 == 0;                    /* rows walked in 32-bit words when they are whole, else bytes */
    const int gx = ((words ? W * C / 4 : W * C) + 255) / 256;
    auto kern = words ? resize_v_kernel<uint32_t> : resize_v_kernel<uint8_t>;
    hipLaunchKernelGGL(kern, dim3(gx, (H_out + V_ROWS - 1) / V_ROWS, n_img), dim3(256), 0, as_stream(stream), in, out, bounds, kk, ksize,
                       n_img, H_in, H_out, W, C, imgs_per_study, minmax);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" int gssd_input_finish_f32(const uint8_t* img, const int32_t* minmax, float mean0, float mean1, float mean2,
                                     float* out_nchw, int B, int phases, int S, int C, int normalize, gssd_stream_t stream) {
    GSSD_CHECK_ARG(img && out_nchw && B > 0 && phases > 0 && S > 0 && C > 0 && C <= 3 && (!normalize || minmax));
    const long long total = (long long)B * phases * S * S;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(input_finish_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), img, minmax, mean0, mean1, mean2,
                       out_nchw, B, phases, S, C, normalize);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
