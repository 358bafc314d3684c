// AP / IoBB evaluator for gfx950: test_ap_iobb.py:126-148 (consumption of the Detect output),
// :231-328 (greedy TP / FP assignment, precision / recall) and :10-41 (voc_ap), batched on the device.
//   eval_match : one wave per image.  Detect's rows are already in descending score order, and the reference's global
//                greedy loop only couples detections of the same image (through that image's `det` flags), so walking
//                each image's rows in order reproduces the global loop.  Overlaps in float64 with the reference's operation
//                order (-ffp-contract=off); first-max argmax like np.argmax.
//   eval_ap    : stable descending radix sort of all confidences (hipCUB), then one workgroup per metric: block scan of
//                the TP / FP flags, precision / recall per rank, 11-point maxima (bit-exact) or the envelope integral.
// Integer / index work (flags, ranks, counts) is bit-exact; the 11-point AP is bit-exact; the area AP differs from numpy's
// pairwise np.sum only in summation order (~1e-16).
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace {

constexpr int EV_MAX_GT = 256;       // ground-truth boxes per image
constexpr int EV_MAX_METRICS = 8;

__global__ __launch_bounds__(64) void eval_match_kernel(const float* __restrict__ det, long long img_stride, int N, int top_k,
                                                        const float* __restrict__ scales, const double* __restrict__ gt,
                                                        const int* __restrict__ gt_off, double thresh,
                                                        const double* __restrict__ thr, int n_iou, int n_iobb,
                                                        float* __restrict__ conf_out, uint8_t* __restrict__ flag_out) {
    __shared__ uint8_t used[EV_MAX_METRICS][EV_MAX_GT];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int nm = n_iou + n_iobb;
    const long long M = (long long)N * top_k;
    const int g0 = gt_off[n], ng = gt_off[n + 1] - g0;
    for (int i = lane; i < EV_MAX_METRICS * EV_MAX_GT; i += 64) (&used[0][0])[i] = 0;
    __syncthreads();
    const float* rows = det + n * img_stride;
    const float s0 = scales[4 * n], s1 = scales[4 * n + 1], s2 = scales[4 * n + 2], s3 = scales[4 * n + 3];
    for (int d = 0; d < top_k; ++d) {
        const float score = rows[5 * d];
        const long long o = (long long)n * top_k + d;
        const bool keep = score > 0.f && (double)score > thresh;          // :129 mask, :147 threshold
        if (lane == 0) conf_out[o] = keep ? score : -INFINITY;
        if (!keep || ng == 0) {                                           // no GT: neither TP nor FP (:257-258)
            if (lane < nm) flag_out[lane * M + o] = 0;
            continue;
        }
        const double b0 = (double)(rows[5 * d + 1] * s0), b1 = (double)(rows[5 * d + 2] * s1);      // fp32 multiply (:137)
        const double b2 = (double)(rows[5 * d + 3] * s2), b3 = (double)(rows[5 * d + 4] * s3);
        const double barea = (b2 - b0) * (b3 - b1);
        double best_iou = -INFINITY, best_iobb = -INFINITY;
        int j_iou = 0, j_iobb = 0;
        bool nan_iou = false, nan_iobb = false;
        for (int j = lane; j < ng; j += 64) {
            const double* G = gt + 4 * (long long)(g0 + j);
            const double ixmin = fmax(G[0], b0), iymin = fmax(G[1], b1), ixmax = fmin(G[2], b2), iymax = fmin(G[3], b3);
            const double iw = fmax(ixmax - ixmin, 0.), ih = fmax(iymax - iymin, 0.);
            const double inters = iw * ih;
            const double uni = (barea + (G[2] - G[0]) * (G[3] - G[1])) - inters;
            const double ov_iou = inters / uni, ov_iobb = inters / barea;
            nan_iou |= ov_iou != ov_iou;
            nan_iobb |= ov_iobb != ov_iobb;
            if (ov_iou > best_iou) { best_iou = ov_iou; j_iou = j; }      // strict: first maximum within the lane
            if (ov_iobb > best_iobb) { best_iobb = ov_iobb; j_iobb = j; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {                          // first maximum across lanes
            const double v1 = __shfl_xor(best_iou, off, 64), v2 = __shfl_xor(best_iobb, off, 64);
            const int i1 = __shfl_xor(j_iou, off, 64), i2 = __shfl_xor(j_iobb, off, 64);
            if (v1 > best_iou || (v1 == best_iou && i1 < j_iou)) { best_iou = v1; j_iou = i1; }
            if (v2 > best_iobb || (v2 == best_iobb && i2 < j_iobb)) { best_iobb = v2; j_iobb = i2; }
        }
        nan_iou = __any(nan_iou);
        nan_iobb = __any(nan_iobb);
        if (lane < nm) {                                                  // lane m owns metric m
            const bool is_iou = lane < n_iou;
            const double ov = is_iou ? best_iou : best_iobb;
            const int j = is_iou ? j_iou : j_iobb;
            const bool bad = is_iou ? nan_iou : nan_iobb;                 // np.max is NaN -> the comparison is False
            uint8_t f = 2;
            if (!bad && ov > thr[lane] && !used[lane][j]) {
                f = 1;
                used[lane][j] = 1;
            }
            flag_out[lane * M + o] = f;
        }
    }
}

__global__ void iota_kernel(int* idx, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) idx[i] = i;
}

// one workgroup per metric; chunked block scan over the sorted list
__global__ __launch_bounds__(1024) void eval_ap_kernel(const float* __restrict__ conf_sorted, const int* __restrict__ idx_sorted,
                                                       const uint8_t* __restrict__ flags, int M, double npos, int use_07,
                                                       double* __restrict__ ap_out) {
    __shared__ int s_tp[1024], s_fp[1024];
    __shared__ double s_max[1024];
    __shared__ double s_p11[11];
    __shared__ int s_seen[11];
    const int m = blockIdx.x, t = threadIdx.x;
    const uint8_t* fl = flags + (long long)m * M;
    const int chunk = (M + 1023) / 1024;
    const int c0 = min(M, t * chunk), c1 = min(M, c0 + chunk);
    int tp = 0, fp = 0;
    for (int i = c0; i < c1; ++i) {
        if (conf_sorted[i] == -INFINITY) break;
        const uint8_t f = fl[idx_sorted[i]];
        tp += f == 1;
        fp += f == 2;
    }
    s_tp[t] = tp;
    s_fp[t] = fp;
    __syncthreads();
    if (t == 0) {                                        // 1024 partials: a serial exclusive scan is cheap enough
        int a = 0, b = 0;
        for (int i = 0; i < 1024; ++i) {
            const int x = s_tp[i], y = s_fp[i];
            s_tp[i] = a;
            s_fp[i] = b;
            a += x;
            b += y;
        }
    }
    if (t < 11) { s_p11[t] = 0.; s_seen[t] = 0; }
    __syncthreads();
    const double eps = 2.220446049250313e-16;            // np.finfo(np.float64).eps
    if (use_07) {
        double pmax[11];
        bool seen[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) { pmax[k] = -INFINITY; seen[k] = false; }
        tp = s_tp[t];
        fp = s_fp[t];
        for (int i = c0; i < c1; ++i) {
            if (conf_sorted[i] == -INFINITY) break;
            const uint8_t f = fl[idx_sorted[i]];
            tp += f == 1;
            fp += f == 2;
            const double rec = (double)tp / npos;
            const double prec = (double)tp / fmax((double)tp + (double)fp, eps);
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const double thr = 0. + k * 0.1;         // np.arange(0., 1.1, 0.1)
                if (rec >= thr) { pmax[k] = fmax(pmax[k], prec); seen[k] = true; }
            }
        }
        for (int k = 0; k < 11; ++k) {                   // block max per threshold
            s_max[t] = seen[k] ? pmax[k] : -INFINITY;
            __syncthreads();
            for (int o = 512; o > 0; o >>= 1) {
                if (t < o) s_max[t] = fmax(s_max[t], s_max[t + o]);
                __syncthreads();
            }
            if (t == 0) { s_p11[k] = s_max[0]; s_seen[k] = s_max[0] > -INFINITY; }
            __syncthreads();
        }
        if (t == 0) {
            double ap = 0.;
            for (int k = 0; k < 11; ++k) {
                const double p = s_seen[k] ? s_p11[k] : 0.;
                ap = ap + p / 11.;
            }
            ap_out[m] = ap;
        }
        return;
    }
    // ---- area under the precision envelope -----------------------------------------------------------------------
    // suffix maximum of prec: per-thread chunk maxima, serial suffix scan over the 1024 partials, then a reverse walk
    int tpe = s_tp[t], fpe = s_fp[t];
    double cmax = 0.;                                    // mpre's trailing sentinel is 0
    int n_valid = 0;
    for (int i = c0; i < c1; ++i) {
        if (conf_sorted[i] == -INFINITY) break;
        const uint8_t f = fl[idx_sorted[i]];
        tpe += f == 1;
        fpe += f == 2;
        cmax = fmax(cmax, (double)tpe / fmax((double)tpe + (double)fpe, eps));
        ++n_valid;
    }
    s_max[t] = cmax;
    __syncthreads();
    if (t == 0) {
        double run = 0.;
        for (int i = 1023; i >= 0; --i) {                // s_max[i] <- max over chunks AFTER i
            const double x = s_max[i];
            s_max[i] = run;
            run = fmax(run, x);
        }
    }
    __syncthreads();
    double suf = s_max[t], acc = 0.;
    for (int i = c0 + n_valid - 1; i >= c0; --i) {
        const uint8_t f = fl[idx_sorted[i]];
        const double prec = (double)tpe / fmax((double)tpe + (double)fpe, eps);
        suf = fmax(suf, prec);
        const double rec = (double)tpe / npos;
        tpe -= f == 1;
        fpe -= f == 2;
        const double rec_prev = (double)tpe / npos;      // mrec[i] (0 for the sentinel: tp = 0 before the first row)
        if (rec != rec_prev) acc += (rec - rec_prev) * suf;
    }
    __syncthreads();
    s_max[t] = acc;
    __syncthreads();
    if (t == 0) {
        double ap = 0.;
        for (int i = 0; i < 1024; ++i) ap += s_max[i];
        ap_out[m] = ap;
    }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int gssd_eval_match(const float* det, long long img_stride, int N, int top_k, const float* scales, const double* gt,
                               const int* gt_off, int max_gt, double thresh, const double* thr, int n_iou, int n_iobb,
                               float* conf_out, uint8_t* flag_out, gssd_stream_t stream) {
    GSSD_CHECK_ARG(det && scales && gt_off && thr && conf_out && flag_out && N > 0 && top_k > 0 && img_stride >= 5ll * top_k);
    GSSD_CHECK_ARG(n_iou >= 0 && n_iobb >= 0 && n_iou + n_iobb > 0 && n_iou + n_iobb <= EV_MAX_METRICS);
    GSSD_CHECK_ARG(max_gt >= 0 && max_gt <= EV_MAX_GT && (gt || max_gt == 0));
    hipLaunchKernelGGL(eval_match_kernel, dim3(N), dim3(64), 0, as_stream(stream), det, img_stride, N, top_k, scales, gt, gt_off,
                       thresh, thr, n_iou, n_iobb, conf_out, flag_out);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

extern "C" long long gssd_eval_workspace_bytes(int M) {
    if (M <= 0) return -1;
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, temp, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                                 (int*)nullptr, M);
    return (long long)(align256(temp) + align256((size_t)M * 4) * 3);
}

extern "C" int gssd_eval_ap(const float* conf, const uint8_t* flags, int M, int n_metrics, double npos, int use_07_metric,
                            void* workspace, long long workspace_bytes, double* ap_out, gssd_stream_t stream) {
    GSSD_CHECK_ARG(conf && flags && M > 0 && n_metrics > 0 && n_metrics <= EV_MAX_METRICS && workspace && ap_out);
    GSSD_CHECK_ARG(workspace_bytes >= gssd_eval_workspace_bytes(M));
    char* w = static_cast<char*>(workspace);
    float* keys_out = reinterpret_cast<float*>(w);
    int* idx_in = reinterpret_cast<int*>(w + align256((size_t)M * 4));
    int* idx_out = reinterpret_cast<int*>(w + 2 * align256((size_t)M * 4));
    void* temp = w + 3 * align256((size_t)M * 4);
    size_t temp_bytes = (size_t)workspace_bytes - 3 * align256((size_t)M * 4);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(iota_kernel, dim3((M + 255) / 256), dim3(256), 0, s, idx_in, M);
    GSSD_CHECK_LAUNCH();
    if (hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, conf, keys_out, idx_in, idx_out, M, 0, 32, s) !=
        hipSuccess) {
        gssd_set_error("evaluator: radix sort failed");
        return GSSD_ELAUNCH;
    }
    hipLaunchKernelGGL(eval_ap_kernel, dim3(n_metrics), dim3(1024), 0, s, keys_out, idx_out, flags, M, npos, use_07_metric,
                       ap_out);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
