// Weight gradient of large 1x1 convolutions (and of the deformable conv's contraction, whose "input" is the column matrix) on the
// fp32 matrix cores, as the slot-scheduled 128 x 256 MFMA stream of csrc/gemm_slot.hip with BOTH operands reduction-major:
//
//   dW[co][k] += sum_m dY[m][co] * X[m][k]        m = output pixel (the reduction), co < Cout, k < K = Cin
//
// dY rows and X rows are what memory holds, so a K-chunk of the reduction (32 pixels) is 32 contiguous rows of each: LDS-DMA lands
// them as they are ([32][128] of dY, [32][256] of X; 2 / 1 rows per 1-KiB wave piece, no swizzle needed: rows are 512 B / 1 KiB apart
// and ds_read_b128's lane groups pair lanes of two rows on complementary 16-byte slots).  A lane reads 4 consecutive co (or k) of one
// pixel row as one ds_read_b128 and uses them as the A (B) operands of FOUR tiles -- tile i of the wave's 64 rows covers
// co = 4 r + i, tile j of its 128 columns k = 64 (j >> 2) + 4 r + (j & 3) -- so a chunk needs the same 24 fragment reads as the NT
// kernel, and the epilogue holds 4 consecutive k per lane.  Split over the reduction (grid z) with fp32 atomics into the zero-filled
// dW, like conv_wgrad.  Replaces conv_wgrad<4,2,2> (0.5 of peak: transposing ds_reads, barrier at the chunk end) for these shapes.
#include <type_traits>
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 256, BK = 32;
constexpr int MT = 4, NT = 8;
constexpr int A_STAGE = BK * BM, B_STAGE = BK * BN;
constexpr int LDS_FLOATS = 2 * (A_STAGE + B_STAGE);

__device__ __attribute__((aligned(16))) float g_zero_ws[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct WsParams {
    const float* dy;      // [R][Cout]
    const float* x;       // [R][x_stride] (+ channel offset already applied); CONV: the NHWC input [B][H][W][x_stride]
    float* dw;            // [Cout][dw_stride]
    int R, Cout, K, x_stride, dw_stride, split, ntn, mtiles;
    // CONV (3x3 / strided / dilated / grouped): K = KH*KW*cin_g per group, Cout = cout_g per group in the tile arithmetic
    int H, W, Ho, Wo, KW, stride, pad, dil, cin_g, cout_g, cout_total;
};

// CONV: the B rows of tap (ty, tx) are the input pixel rows shifted by that tap -- lane l of a B piece owns k' = k0 + 4 l, i.e. one
// (tap, channel quad), and computes its own shifted source pixel and border predicate per reduction row; the group is blockIdx.y.
template <bool CONV>
__global__ __launch_bounds__(256, 1) void wgrad_slot_kernel(const WsParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                     // [2][32][BM]
    float* const Bs = smem + 2 * A_STAGE;       // [2][32][BN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
    const int nt = blockIdx.x % p.ntn, mt = blockIdx.x / p.ntn;
    const int co0 = mt * BM, k0 = nt * BN;
    const int g = CONV ? blockIdx.y : 0;
    const int nchunks_all = (p.R + BK - 1) / BK;
    const int cps = (nchunks_all + p.split - 1) / p.split;
    const int ch_begin = blockIdx.z * cps, ch_end = min(nchunks_all, ch_begin + cps);
    if (ch_begin >= ch_end) return;

    // DMA roles.  A piece (2 rows x 128 co): lane -> row lane >> 5, co quad lane & 31;  B piece (1 row x 256 k): lane -> k quad lane
    const int a_row = lane >> 5, a_q = lane & 31;
    const int a_stride = CONV ? p.cout_total : p.Cout;
    const bool a_ok = co0 + 4 * a_q < p.Cout, b_ok = k0 + 4 * lane < p.K;
    const float* a_src = p.dy + (size_t)(ch_begin * BK) * a_stride + g * p.Cout + co0 + 4 * a_q;
    const float* b_src = p.x + (CONV ? 0 : (size_t)(ch_begin * BK) * p.x_stride + k0 + 4 * lane);
    int r_next = ch_begin * BK;                  // first reduction row of the chunk being staged
    // CONV: this lane's tap offset / channel, and the (image, y, x) of the eight reduction rows its wave stages per chunk
    int l_dy = 0, l_dx = 0, l_ch = 0;
    int pb[8], py[8], px[8];
    if (CONV) {
        const int kk = b_ok ? k0 + 4 * lane : 0;
        const int tap = kk / p.cin_g, c = kk - tap * p.cin_g;
        const int ty = tap / p.KW, tx = tap - ty * p.KW;
        l_dy = ty * p.dil - p.pad;
        l_dx = tx * p.dil - p.pad;
        l_ch = g * p.cin_g + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = r_next + j * 4 + wave;
            const int hw = p.Ho * p.Wo;
            pb[j] = m / hw;
            const int rem = m - pb[j] * hw;
            py[j] = rem / p.Wo;
            px[j] = rem - py[j] * p.Wo;
        }
    }

    f32x4 acc[MT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4;

    auto stage_a = [&](int j, float* dst) {      // piece j * 4 + wave: rows 2 * piece + a_row
        const int rr = 2 * (j * 4 + wave) + a_row;
        const bool ok = a_ok && r_next + rr < p.R;
        dma16(ok ? a_src + (size_t)rr * a_stride : g_zero_ws, dst + (j * 4 + wave) * 256);
    };
    auto stage_b = [&](int j, float* dst) {      // piece j * 4 + wave: row = piece
        const int rr = j * 4 + wave;
        if (CONV) {
            const int iy = py[j] * p.stride + l_dy, ix = px[j] * p.stride + l_dx;
            const bool row_ok = b_ok && r_next + rr < p.R;
            const bool ok = row_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            dma16(ok ? b_src + ((size_t)(pb[j] * p.H + iy) * p.W + ix) * p.x_stride + l_ch : g_zero_ws, dst + (j * 4 + wave) * 256);
        } else {
            const bool ok = b_ok && r_next + rr < p.R;
            dma16(ok ? b_src + (size_t)rr * p.x_stride : g_zero_ws, dst + (j * 4 + wave) * 256);
        }
    };
    auto advance = [&]() {
        a_src += (size_t)BK * a_stride;
        if (!CONV) b_src += (size_t)BK * p.x_stride;
        r_next += BK;
        if (CONV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                px[j] += BK;
                while (px[j] >= p.Wo) {
                    px[j] -= p.Wo;
                    if (++py[j] == p.Ho) {
                        py[j] = 0;
                        ++pb[j];
                    }
                }
            }
        }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_a(j, As);
#pragma unroll
    for (int j = 0; j < 8; ++j) stage_b(j, Bs);
    advance();
    __syncthreads();

    // fragment read offsets (floats): k-step t of a chunk reads row 4 t + kq; A: co = wm * 64 + 4 r .., B: k = wn * 128 + 64 h + 4 r ..
    const int fa = kq * BM + wm * 64 + 4 * r;
    const int fb = kq * BN + wn * 128 + 4 * r;
    f32x4 af[8], bf[8][2];
    float* a_dst = nullptr;
    float* b_dst = nullptr;
    auto slot = [&](auto kc, auto stage_c, const float* Ab, const float* Bb, const float* Abn, const float* Bbn) {
        constexpr int KK = decltype(kc)::value;
        constexpr bool stage = decltype(stage_c)::value;
        constexpr int t = KK >> 5, i = (KK >> 3) & 3, j = KK & 7;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][i], bf[t][j >> 2][j & 3], acc[i][j], 0, 0, 0);
        // second half (k-steps 4..7) of this chunk's fragments
        if constexpr (KK < 4) af[4 + KK] = *reinterpret_cast<const f32x4*>(Ab + (4 + KK) * 4 * BM + fa);
        if constexpr (KK >= 4 && KK < 12) bf[4 + ((KK - 4) >> 1)][(KK - 4) & 1] = *reinterpret_cast<const f32x4*>(Bb + (4 + ((KK - 4) >> 1)) * 4 * BN + ((KK - 4) & 1) * 64 + fb);
        if constexpr (stage) {
            if constexpr (KK >= 12 && KK < 36 && ((KK - 12) & 1) == 0) {
                constexpr int n = (KK - 12) >> 1;
                if constexpr (n < 4) stage_a(n, a_dst);
                else stage_b(n - 4, b_dst);
            }
            if constexpr (KK == 36) advance();
            if constexpr (KK == 192) __syncthreads();
            if constexpr (KK >= 194 && KK < 198) af[KK - 194] = *reinterpret_cast<const f32x4*>(Abn + (KK - 194) * 4 * BM + fa);
            if constexpr (KK >= 198 && KK < 206) bf[(KK - 198) >> 1][(KK - 198) & 1] = *reinterpret_cast<const f32x4*>(Bbn + ((KK - 198) >> 1) * 4 * BN + ((KK - 198) & 1) * 64 + fb);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        af[t] = *reinterpret_cast<const f32x4*>(As + t * 4 * BM + fa);
        bf[t][0] = *reinterpret_cast<const f32x4*>(Bs + t * 4 * BN + fb);
        bf[t][1] = *reinterpret_cast<const f32x4*>(Bs + t * 4 * BN + 64 + fb);
    }
    const int nch = ch_end - ch_begin;
    for (int ch = 0; ch < nch - 1; ++ch) {
        const int buf = ch & 1;
        a_dst = As + (buf ^ 1) * A_STAGE;
        b_dst = Bs + (buf ^ 1) * B_STAGE;
        const float* Ab = As + buf * A_STAGE;
        const float* Bb = Bs + buf * B_STAGE;
        const float* Abn = As + (buf ^ 1) * A_STAGE;
        const float* Bbn = Bs + (buf ^ 1) * B_STAGE;
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 256>([&](auto kc) { slot(kc, std::true_type{}, Ab, Bb, Abn, Bbn); });
    }
    {
        const int buf = (nch - 1) & 1;
        const float* Ab = As + buf * A_STAGE;
        const float* Bb = Bs + buf * B_STAGE;
        static_for<0, 256>([&](auto kc) { slot(kc, std::false_type{}, Ab, Bb, Ab, Bb); });
    }

    // ---- epilogue: tile (i, j) element e of lane (r, kq) is dW[co0 + wm*64 + 4*(4*kq + e) + i][k0 + wn*128 + 64*(j>>2) + 4*r + (j&3)].
    // The reduction slices meet in fp32 atomics on the zero-filled dW.  Issued straight from the accumulator layout an atomic
    // instruction would touch 8 cache lines with 8 floats each (measured: 37 G atomics/s -- the epilogue of a 45-chunk slice cost
    // more than its MFMAs); each wave therefore transposes its 64 x 128 block through its own 16 KB of the (now idle) stage memory,
    // 64 columns at a time, so that one instruction adds 64 consecutive floats = two full lines.
    const int cog = g * p.Cout;
    if (p.split == 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + wm * 64 + 4 * (4 * kq + e) + i;
                if (co >= p.Cout) continue;
                float* row = p.dw + (size_t)(cog + co) * p.dw_stride;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = k0 + wn * 128 + 64 * h + 4 * r;
                    if (k >= p.K) continue;                       // K % 4 == 0: a quad is in or out as a whole
                    f32x4 v = *reinterpret_cast<f32x4*>(row + k);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] += acc[i][4 * h + c][e];
                    *reinterpret_cast<f32x4*>(row + k) = v;
                }
            }
        return;
    }
    __syncthreads();                                          // every wave is done reading the stages
    constexpr int TP = 65;                                    // row pitch (floats): column reads and row writes both conflict free
    float* const tb = smem + wave * (64 * TP);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) tb[(4 * (4 * kq + e) + i) * TP + 4 * r + c] = acc[i][4 * h + c][e];
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this wave's own writes have landed (no other wave reads them)
        __builtin_amdgcn_wave_barrier();
        const int k = k0 + wn * 128 + 64 * h + lane;
        if (k < p.K) {
#pragma unroll 8
            for (int row = 0; row < 64; ++row) {
                const int co = co0 + wm * 64 + row;
                if (co < p.Cout) unsafeAtomicAdd(p.dw + (size_t)(cog + co) * p.dw_stride + k, tb[row * TP + lane]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

static int pick_split(int tiles, int nchunks) {
    // split the reduction so that the grid is a whole number of 256-CU rounds (about), each slice >= 8 chunks
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 64 && nchunks / s >= 8; ++s) {
        const double rounds = (double)tiles * s / 256.0;
        const double eff = rounds / (double)(long long)(rounds + 0.999999);
        // a slice pays one prologue + epilogue (~10 us) per tile: prefer fewer, longer slices at equal fill
        const double score = eff - 0.01 * s * (8.0 / (nchunks / s + 8.0));
        if (score > best_eff) {
            best_eff = score;
            best = s;
        }
    }
    return best;
}

template <bool CONV>
static int launch_ws(const WsParams& p, int groups, hipStream_t stream) {
    static unsigned attr_mask = 0;
    constexpr int smem = LDS_FLOATS * (int)sizeof(float);
    auto kern = wgrad_slot_kernel<CONV>;
    if (gssd_attr_needed(&attr_mask) &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        gssd_set_error("hipFuncSetAttribute(max dynamic LDS = %d) failed", smem);
        return GSSD_ELAUNCH;
    }
    gssd_attr_done(&attr_mask);
    hipLaunchKernelGGL(kern, dim3(p.ntn * p.mtiles, groups, p.split), dim3(256), smem, stream, p);
    GSSD_CHECK_LAUNCH();
    return GSSD_OK;
}

// returns 1 when the descriptor is not a shape this kernel takes (the caller falls through to conv_wgrad / the patch kernels)
int gssd_try_wgrad_slot(const gssd_conv_desc& d, const float* dy, float* dw, hipStream_t stream) {
    if (d.m_per_image) return 1;
    const long long R = (long long)d.B * d.Ho * d.Wo;
    if (R < 4096 || R >= (1ll << 31) || ((uintptr_t)dw % 16) != 0 || d.in_stride % 4 != 0 || d.in_ch_off % 4 != 0) return 1;
    const bool plain = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 && d.groups == 1 && !d.in_scale;
    const int cout_g = d.Cout / d.groups;
    WsParams p = {};
    p.dy = dy;
    p.x = d.in + d.in_ch_off;
    p.dw = dw;
    p.R = (int)R;
    p.K = d.K;
    p.x_stride = d.in_stride;
    p.dw_stride = d.K;                          // the packed gradient matrix is [Cout][K] like conv_wgrad writes it
    p.ntn = (d.K + BN - 1) / BN;
    const int nchunks = (p.R + BK - 1) / BK;
    if (plain) {
        if (d.K % 4 != 0 || d.Cout % 4 != 0 || d.K < 192 || d.Cout < 96) return 1;
        p.Cout = d.Cout;
        p.mtiles = (d.Cout + BM - 1) / BM;
        // tile fill (ragged Cout / K tails waste MFMA work): stay on the generic kernel below 0.7
        if (((double)d.Cout / (p.mtiles * BM)) * ((double)d.K / (p.ntn * BN)) < 0.7) return 1;
        p.split = pick_split(p.ntn * p.mtiles, nchunks);
        return launch_ws<false>(p, 1, stream);
    }
    // convolutions with taps (stride / dilation / padding; the group index is a grid dimension)
    // Dense layers only (the DCN offset conv: 1.75 -> 1.06 ms).  Measured and rejected for the grouped trunk layers: conv4_2 (4 groups x
    // 128 x 1152 weights, 25 reduction slices so that 500 workgroups fill the chip) runs 765 us here against 484 us on conv_wgrad<4,2,2>
    // (112 TFLOP/s: its 128 x 128 tiles need 22 slices only and two workgroups share a CU) -- with so little output per group the
    // one-workgroup-per-CU stream spends its time in prologues and atomic epilogues.
    if (d.groups != 1 || d.in_scale || d.cin_g % 4 != 0 || cout_g % 4 != 0 || cout_g < 96 || d.K < 192) return 1;
    if ((long long)d.B * d.H * d.W * d.in_stride >= (1ll << 31)) return 1;
    p.Cout = cout_g;
    p.cout_total = d.Cout;
    p.cout_g = cout_g;
    p.cin_g = d.cin_g;
    p.mtiles = (cout_g + BM - 1) / BM;
    if (((double)cout_g / (p.mtiles * BM)) * ((double)d.K / (p.ntn * BN)) < 0.7) return 1;
    p.H = d.H; p.W = d.W; p.Ho = d.Ho; p.Wo = d.Wo; p.KW = d.KW; p.stride = d.stride; p.pad = d.pad; p.dil = d.dil;
    p.split = pick_split(p.ntn * p.mtiles * d.groups, nchunks);
    return launch_ws<true>(p, d.groups, stream);
}
