// Detect on gfx950: confidence threshold -> top_k selection -> decode -> greedy NMS, one 256-thread
// workgroup per (image, class).  Compiled with -ffp-contract=off so decode / IoU round exactly like
// the reference's separate fp32 ops (layers/box_utils.py:139-157,186-238); exp() is evaluated in
// double and rounded once (see oracle/gssd_oracle.py::decode).
//
// Selection: the reference sorts all scores ascending and walks the last top_k from the end.  Here a
// 4-pass radix select finds the top_k-th score, the <= top_k survivors are gathered and bitonic-sorted
// in LDS on the key (score desc, prior index asc), an LDS bit matrix holds "IoU > thresh" for every
// ordered pair (wave-parallel), and a single lane sweeps it serially -- the only sequential part.
#include "common.h"

namespace {

constexpr int DT = 256;
constexpr int MAXK = 256;   // top_k <= 256 (the path uses 200)

__global__ __launch_bounds__(DT) void detect_kernel(const float* __restrict__ loc, const float* __restrict__ conf,
                                                    const float* __restrict__ priors, int P, int C, int top_k,
                                                    float conf_thresh, float nms_thresh, float var0, float var1,
                                                    int conf_is_logits, int loc_is_boxes,
                                                    float* __restrict__ out, int* __restrict__ keep_idx,
                                                    int* __restrict__ keep_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* sc = reinterpret_cast<float*>(smraw);   // [P] masked scores (0 = below threshold)
    __shared__ unsigned hist[256];
    __shared__ int bcast[2];
    __shared__ unsigned long long key[MAXK];
    __shared__ float bx[MAXK][4];
    __shared__ float barea[MAXK];
    __shared__ unsigned long long supp[MAXK][MAXK / 64];
    __shared__ int s_wcnt[DT / 64];
    __shared__ int s_base[2];
    __shared__ int order[MAXK];
    __shared__ int s_nkeep;

    const int cl = blockIdx.x % (C - 1) + 1;
    const int b = blockIdx.x / (C - 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* cb = conf + (size_t)b * P * C;

    // class 0 (background) rows are all zero (detection_pytorch_ver_1point5.py:56,63)
    if (cl == 1)
        for (int i = tid; i < top_k * 5; i += DT) out[((size_t)b * C) * top_k * 5 + i] = 0.f;
    float* ob = out + ((size_t)b * C + cl) * top_k * 5;
    for (int i = tid; i < top_k * 5; i += DT) ob[i] = 0.f;
    if (keep_idx)
        for (int i = tid; i < top_k; i += DT) {
            keep_idx[((size_t)b * C + cl) * top_k + i] = -1;
            if (cl == 1) keep_idx[((size_t)b * C) * top_k + i] = -1;
        }

    int cnt = 0;
    for (int p = tid; p < P; p += DT) {
        float s;
        if (conf_is_logits) {
            float m = cb[(size_t)p * C];
            for (int c = 1; c < C; ++c) m = fmaxf(m, cb[(size_t)p * C + c]);
            float z = 0.f;
            for (int c = 0; c < C; ++c) z += (float)exp((double)(cb[(size_t)p * C + c] - m));
            s = __fdiv_rn((float)exp((double)(cb[(size_t)p * C + cl] - m)), z);
        } else {
            s = cb[(size_t)p * C + cl];
        }
        const bool ok = s > conf_thresh;   // strict (detection_pytorch_ver_1point5.py:69)
        sc[p] = ok ? s : 0.f;
        cnt += ok;
    }
    cnt = wave_sum(cnt);
    if (lane == 0) s_wcnt[wave] = cnt;
    __syncthreads();
    cnt = 0;
    for (int w = 0; w < DT / 64; ++w) cnt += s_wcnt[w];
    if (cnt == 0) {
        if (tid == 0 && keep_cnt) {
            keep_cnt[b * C + cl] = 0;
            if (cl == 1) keep_cnt[b * C] = 0;
        }
        return;
    }
    const int ncand = cnt < top_k ? cnt : top_k;

    // ---- ncand-th largest score (bit pattern order == value order for positive floats) ----------
    unsigned prefix = 0, mask = 0;
    int greater = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += DT) hist[i] = 0;
        __syncthreads();
        for (int p = tid; p < P; p += DT) {
            const unsigned u = __float_as_uint(sc[p]);
            if (u != 0 && (u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = greater, bin = 255;
            for (; bin > 0; --bin) {
                if (acc + (int)hist[bin] >= ncand) break;
                acc += (int)hist[bin];
            }
            bcast[0] = bin;
            bcast[1] = acc;
        }
        __syncthreads();
        prefix |= ((unsigned)bcast[0]) << shift;
        mask |= 255u << shift;
        greater = bcast[1];
        __syncthreads();
    }
    const unsigned kth = prefix;
    const int ties_needed = ncand - greater;

    // ---- gather survivors: score > kth, plus the lowest-index ties -------------------------------
    if (tid == 0) {
        s_base[0] = 0;   // write cursor
        s_base[1] = 0;   // ties taken so far
    }
    for (int i = tid; i < MAXK; i += DT) key[i] = ~0ull;   // sorts last
    __syncthreads();
    for (int p0 = 0; p0 < P; p0 += DT) {
        const int p = p0 + tid;
        unsigned u = 0;
        if (p < P) u = __float_as_uint(sc[p]);
        const bool gt = u > kth;
        const bool tie = (u == kth) && u != 0;
        // ordered tie count
        const unsigned long long tb = __ballot(tie);
        const int tbefore = __popcll(tb & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[wave] = __popcll(tb);
        __syncthreads();
        int tbase = s_base[1];
        for (int w = 0; w < wave; ++w) tbase += s_wcnt[w];
        const bool take = gt || (tie && tbase + tbefore < ties_needed);
        int ttot = 0;
        for (int w = 0; w < DT / 64; ++w) ttot += s_wcnt[w];
        __syncthreads();
        const unsigned long long kb = __ballot(take);
        const int kbefore = __popcll(kb & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[wave] = __popcll(kb);
        __syncthreads();
        int kbase = s_base[0];
        for (int w = 0; w < wave; ++w) kbase += s_wcnt[w];
        int ktot = 0;
        for (int w = 0; w < DT / 64; ++w) ktot += s_wcnt[w];
        if (take) key[kbase + kbefore] = ((unsigned long long)(~u) << 32) | (unsigned)p;   // score desc, index asc
        __syncthreads();
        if (tid == 0) {
            s_base[0] += ktot;
            s_base[1] += ttot;
        }
        __syncthreads();
    }

    // ---- bitonic sort of MAXK keys ----------------------------------------------------------------
    for (int k2 = 2; k2 <= MAXK; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const int i = tid;   // DT == MAXK
            const int ixj = i ^ j;
            if (ixj > i) {
                const unsigned long long a = key[i], c2 = key[ixj];
                const bool up = (i & k2) == 0;
                if ((a > c2) == up) {
                    key[i] = c2;
                    key[ixj] = a;
                }
            }
            __syncthreads();
        }
    }

    // ---- decode the candidates (box_utils.py:152-157) --------------------------------------------
    if (tid < ncand) {
        const int p = (int)(key[tid] & 0xffffffffu);
        const float4 l = reinterpret_cast<const float4*>(loc)[(size_t)b * P + p];
        const float4 pr = reinterpret_cast<const float4*>(priors)[p];
        const float cx = pr.x + (l.x * var0) * pr.z;
        const float cy = pr.y + (l.y * var0) * pr.w;
        const float w = pr.z * (float)exp((double)(l.z * var1));
        const float h = pr.w * (float)exp((double)(l.w * var1));
        float x1 = cx - w / 2.f, y1 = cy - h / 2.f;
        float x2 = w + x1, y2 = h + y1;
        if (loc_is_boxes) {   // box_utils.nms() entry: rows are already (x1, y1, x2, y2)
            x1 = l.x;
            y1 = l.y;
            x2 = l.z;
            y2 = l.w;
        }
        bx[tid][0] = x1;
        bx[tid][1] = y1;
        bx[tid][2] = x2;
        bx[tid][3] = y2;
        barea[tid] = (x2 - x1) * (y2 - y1);
    }
    for (int i = tid; i < MAXK * (MAXK / 64); i += DT) supp[i / (MAXK / 64)][i % (MAXK / 64)] = 0ull;
    __syncthreads();

    // ---- suppression bit matrix: row i, bit j (j > i) set iff NOT (IoU(j | i) <= thresh) ------------
    for (int i = wave; i < ncand; i += DT / 64) {
        const float ix1 = bx[i][0], iy1 = bx[i][1], ix2 = bx[i][2], iy2 = bx[i][3], ia = barea[i];
        for (int j0 = 0; j0 < ncand; j0 += 64) {
            const int j = j0 + lane;
            bool s = false;
            if (j < ncand && j > i) {
                const float xx1 = fmaxf(bx[j][0], ix1), yy1 = fmaxf(bx[j][1], iy1);
                const float xx2 = fminf(bx[j][2], ix2), yy2 = fminf(bx[j][3], iy2);
                const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
                const float inter = w * h;
                const float uni = (barea[j] - inter) + ia;
                const float iou = __fdiv_rn(inter, uni);
                s = !(iou <= nms_thresh);
            }
            const unsigned long long m = __ballot(s);
            if (lane == 0) supp[i][j0 >> 6] = m;
        }
    }
    __syncthreads();

    // ---- serial greedy sweep (one lane) ----------------------------------------------------------------
    if (tid == 0) {
        unsigned long long dead[MAXK / 64] = {0ull, 0ull, 0ull, 0ull};
        int nk = 0;
        for (int i = 0; i < ncand; ++i) {
            if ((dead[i >> 6] >> (i & 63)) & 1ull) continue;
            order[nk++] = i;
#pragma unroll
            for (int w = 0; w < MAXK / 64; ++w) dead[w] |= supp[i][w];
        }
        s_nkeep = nk;
        if (keep_cnt) {
            keep_cnt[b * C + cl] = nk;
            if (cl == 1) keep_cnt[b * C] = 0;
        }
    }
    __syncthreads();
    const int nk = s_nkeep;
    if (tid < nk) {
        const int i = order[tid];
        const unsigned long long kk = key[i];
        const int p = (int)(kk & 0xffffffffu);
        float* o = ob + (size_t)tid * 5;
        o[0] = __uint_as_float(~(unsigned)(kk >> 32));
        o[1] = bx[i][0];
        o[2] = bx[i][1];
        o[3] = bx[i][2];
        o[4] = bx[i][3];
        if (keep_idx) keep_idx[((size_t)b * C + cl) * top_k + tid] = p;
    }
}

}  // namespace

extern "C" int gssd_detect(const float* loc, const float* conf, const float* priors, int B, int P, int C, int top_k,
                           float conf_thresh, float nms_thresh, float var0, float var1, int conf_is_logits,
                           int loc_is_boxes, float* out, int* keep_idx, int* keep_cnt, gssd_stream_t stream) {
    GSSD_CHECK_ARG(loc && conf && priors && out);
    GSSD_CHECK_ARG(B > 0 && P > 0 && C >= 2 && top_k > 0 && top_k <= MAXK && P <= 30000);
    GSSD_CHECK_ARG(nms_thresh > 0.f);   // the reference raises ValueError (detection_pytorch_ver_1point5.py:39-40)
    GSSD_CHECK_ARG(conf_thresh >= 0.f);
    GSSD_CHECK_ARG(((uintptr_t)loc % 16) == 0 && ((uintptr_t)priors % 16) == 0);
    static unsigned attr_mask = 0;     // one bit per device (the attribute is per device)
    const size_t smem = (size_t)P * sizeof(float);
    if (gssd_attr_needed(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(detect_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            120 * 1024);
    }
    gssd_attr_done(&attr_mask);
    hipLaunchKernelGGL(detect_kernel, dim3(B * (C - 1)), dim3(DT), smem, as_stream(stream), loc, conf, priors, P, C,
                       top_k, conf_thresh, nms_thresh, var0, var1, conf_is_logits, loc_is_boxes, out, keep_idx, keep_cnt);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}
