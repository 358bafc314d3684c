/* TEST INFRASTRUCTURE -- never linked into or called from the product path.
 *
 * A second, structurally independent CPU restatement of the modulated deformable convolution (DCNv2) behind
 * /root/reference/ssd_liverdet/layers/dcn_v2_custom.py:79-89 (`_DCNv2.apply(input, offset, mask, weight, bias, stride, padding,
 * dilation, deformable_groups)`).  The arithmetic lives in the un-vendored third-party extension `dcn_v2`
 * (CharlesShang/DCNv2, version not pinned by the reference) -- PARITY UNPINNED: nothing in /root/reference holds a
 * known answer for it.  oracle/gssd_oracle.py::dcn_v2_conv restates it with vectorised torch gathers; THIS file restates it the
 * other way round -- one scalar loop nest per output element, in the order of the published algorithm:
 *
 *   modulated_deformable_im2col:  for (b, c_im, h_col, w_col), for tap (i, j):
 *        group       = c_im / (C / deformable_groups)
 *        off_h       = offset[b][group*2*kh*kw + 2*(i*kw + j)    ][h_col][w_col]
 *        off_w       = offset[b][group*2*kh*kw + 2*(i*kw + j) + 1][h_col][w_col]       (row first, then column:
 *                                                                   corroborated by utils/show_offset.py:28-32)
 *        m           = mask  [b][group*kh*kw   +    i*kw + j     ][h_col][w_col]
 *        h_im, w_im  = h_col*stride - pad + i*dil + off_h,  w_col*stride - pad + j*dil + off_w
 *        val         = (h_im > -1 && w_im > -1 && h_im < H && w_im < W) ? bilinear(h_im, w_im) : 0
 *        col[(c_im*kh*kw + i*kw + j)][b][h_col][w_col] = val * m
 *   bilinear (dmcn_im2col_bilinear): floor corners; each of the four corners contributes only if it lies inside the map
 *        (h_low >= 0, w_low >= 0, h_high <= H-1, w_high <= W-1 tested per corner); weights (1-lh)(1-lw), (1-lh)lw, lh(1-lw), lh lw.
 *   output = weight[Cout][C*kh*kw] x col + bias  (dense weight, no conv groups).
 *
 * The sampling arithmetic is single precision like the extension's float instantiation; the contraction accumulates in DOUBLE
 * (so this is also the arbiter for accumulation-order differences between the torch oracle and the HIP kernel).
 * Layouts are NCHW, exactly the tensors the reference call site passes.  Two independent restatements agreeing is the most that can
 * exist in this environment; it does not lift "parity unpinned".
 */
#include <math.h>
#include <stddef.h>

static float bilinear_at(const float *plane, int H, int W, float h, float w) {
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = h - (float)h_low, lw = w - (float)w_low;
    float hh = 1.0f - lh, hw = 1.0f - lw;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h_low >= 0 && w_low >= 0) v1 = plane[(size_t)h_low * W + w_low];
    if (h_low >= 0 && w_high <= W - 1) v2 = plane[(size_t)h_low * W + w_high];
    if (h_high <= H - 1 && w_low >= 0) v3 = plane[(size_t)h_high * W + w_low];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = plane[(size_t)h_high * W + w_high];
    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* x [B][C][H][W], offset [B][dg*2*kh*kw][Ho][Wo], mask [B][dg*kh*kw][Ho][Wo], weight [Cout][C][kh][kw], bias [Cout] or NULL,
 * out [B][Cout][Ho][Wo] (float).  Returns 0, or -1 on a shape that does not divide. */
int dcn_scalar_forward(const float *x, const float *offset, const float *mask, const float *weight, const float *bias, float *out,
                       int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil, int dg) {
    if (dg <= 0 || C % dg != 0) return -1;
    int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1;
    int Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
    int cpg = C / dg, K = kh * kw;
    for (int b = 0; b < B; ++b)
        for (int h_col = 0; h_col < Ho; ++h_col)
            for (int w_col = 0; w_col < Wo; ++w_col)
                for (int co = 0; co < Cout; ++co) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int c_im = 0; c_im < C; ++c_im) {
                        int group = c_im / cpg;
                        const float *plane = x + ((size_t)b * C + c_im) * H * W;
                        const float *off_b = offset + ((size_t)b * dg + group) * 2 * K * Ho * Wo;
                        const float *msk_b = mask + ((size_t)b * dg + group) * K * Ho * Wo;
                        for (int i = 0; i < kh; ++i)
                            for (int j = 0; j < kw; ++j) {
                                int t = i * kw + j;
                                float off_h = off_b[((size_t)(2 * t) * Ho + h_col) * Wo + w_col];
                                float off_w = off_b[((size_t)(2 * t + 1) * Ho + h_col) * Wo + w_col];
                                float m = msk_b[((size_t)t * Ho + h_col) * Wo + w_col];
                                float h_im = (float)(h_col * stride - pad + i * dil) + off_h;
                                float w_im = (float)(w_col * stride - pad + j * dil) + off_w;
                                float val = 0.f;
                                if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)
                                    val = bilinear_at(plane, H, W, h_im, w_im);
                                float col = val * m;
                                acc += (double)weight[(((size_t)co * C + c_im) * kh + i) * kw + j] * (double)col;
                            }
                    }
                    out[(((size_t)b * Cout + co) * Ho + h_col) * Wo + w_col] = (float)acc;
                }
    return 0;
}
