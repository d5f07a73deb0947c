/* CPU ORACLE - test infrastructure only.  NOT part of the product path.
 *
 * Scalar restatement of the reference's modulated deformable convolution, one output element at a
 * time, following the CUDA kernels statement by statement (paths relative to /root/reference):
 *   bilinear sample with per-corner bounds ... thirdparty/deform_conv/src/deform_conv_cuda_kernel.cu:403-432
 *   im2col (offset/mask channel indexing, (-1,H)x(-1,W) test) ................................. :506-571
 *   gradient weight of one integer neighbour .................................................. :434-459
 *   coordinate (offset) derivative ............................................................ :461-503
 *   col2im (grad_x scatter over the 5x5 window) ............................................... :574-631
 *   col2im_coord (grad_offset, grad_mask) ..................................................... :634-705
 *   host: GEMM + bias, per-image loop, grad_weight / grad_bias accumulation ... deform_conv_cuda.cpp:474-549, 551-664
 *
 * PARITY UNPINNED BY EXECUTION: the reference's only implementation of this arithmetic is CUDA
 * source that cannot be built in this image (needs nvcc + ATen/THC); this file is the second,
 * independent restatement used to cross-check oracle/otpose_oracle.py::mdcn_forward/backward.
 *
 * Built by oracle/Makefile into oracle/_build/libmdcn_scalar.so (float and double entry points).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define DEFINE_MDCN(REAL, SUF)                                                                        \
static REAL bilinear_##SUF(const REAL *im, int H, int W, REAL h, REAL w) {                            \
    int h_low = (int)floor((double)h), w_low = (int)floor((double)w);                                 \
    int h_high = h_low + 1, w_high = w_low + 1;                                                       \
    REAL lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                                    \
    REAL v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                              \
    if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];                                         \
    if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];                                   \
    if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];                                   \
    if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];                             \
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;                                 \
}                                                                                                     \
static REAL grad_weight_##SUF(REAL ah, REAL aw, int h, int w, int H, int W) {                         \
    if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;                                         \
    int hl = (int)floor((double)ah), wl = (int)floor((double)aw), hh = hl + 1, wh = wl + 1;           \
    REAL wt = 0;                                                                                      \
    if (h == hl && w == wl) wt = (h + 1 - ah) * (w + 1 - aw);                                         \
    if (h == hl && w == wh) wt = (h + 1 - ah) * (aw + 1 - w);                                         \
    if (h == hh && w == wl) wt = (ah + 1 - h) * (w + 1 - aw);                                         \
    if (h == hh && w == wh) wt = (ah + 1 - h) * (aw + 1 - w);                                         \
    return wt;                                                                                        \
}                                                                                                     \
static REAL coord_weight_##SUF(REAL ah, REAL aw, int H, int W, const REAL *im, int dir) {             \
    if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;                                         \
    int hl = (int)floor((double)ah), wl = (int)floor((double)aw), hh = hl + 1, wh = wl + 1;           \
    REAL wt = 0;                                                                                      \
    if (dir == 0) {                                                                                   \
        if (hl >= 0 && wl >= 0) wt += -1 * (wl + 1 - aw) * im[hl * W + wl];                           \
        if (hl >= 0 && wh <= W - 1) wt += -1 * (aw - wl) * im[hl * W + wh];                           \
        if (hh <= H - 1 && wl >= 0) wt += (wl + 1 - aw) * im[hh * W + wl];                            \
        if (hh <= H - 1 && wh <= W - 1) wt += (aw - wl) * im[hh * W + wh];                            \
    } else {                                                                                          \
        if (hl >= 0 && wl >= 0) wt += -1 * (hl + 1 - ah) * im[hl * W + wl];                           \
        if (hl >= 0 && wh <= W - 1) wt += (hl + 1 - ah) * im[hl * W + wh];                            \
        if (hh <= H - 1 && wl >= 0) wt += -1 * (ah - hl) * im[hh * W + wl];                           \
        if (hh <= H - 1 && wh <= W - 1) wt += (ah - hl) * im[hh * W + wh];                            \
    }                                                                                                 \
    return wt;                                                                                        \
}                                                                                                     \
/* columns of one image: col[(c*K + k) * P + p] */                                                    \
static void im2col_##SUF(const REAL *x, const REAL *off, const REAL *msk, REAL *col, int C, int H,    \
                         int W, int kh, int kw, int stride, int pad, int dil, int dg, int Ho, int Wo) {\
    int K = kh * kw, P = Ho * Wo, cpg = C / dg;                                                       \
    for (int c = 0; c < C; ++c) {                                                                     \
        int g = c / cpg;                                                                              \
        const REAL *o = off + (size_t)g * 2 * K * P, *m = msk + (size_t)g * K * P;                    \
        for (int ho = 0; ho < Ho; ++ho) for (int wo = 0; wo < Wo; ++wo) {                             \
            int p = ho * Wo + wo, h_in = ho * stride - pad, w_in = wo * stride - pad;                 \
            for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j) {                               \
                int k = i * kw + j;                                                                   \
                REAL h_im = h_in + i * dil + o[(2 * k) * P + p];                                      \
                REAL w_im = w_in + j * dil + o[(2 * k + 1) * P + p];                                  \
                REAL val = 0;                                                                         \
                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                                   \
                    val = bilinear_##SUF(x + (size_t)c * H * W, H, W, h_im, w_im);                    \
                col[((size_t)c * K + k) * P + p] = val * m[k * P + p];                                \
            }                                                                                         \
        }                                                                                             \
    }                                                                                                 \
}                                                                                                     \
int mdcn_forward_##SUF(const REAL *x, const REAL *offset, const REAL *mask, const REAL *weight,       \
                       const REAL *bias, REAL *out, int N, int C, int H, int W, int Co, int kh,       \
                       int kw, int stride, int pad, int dil, int groups, int dg) {                    \
    int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;                                       \
    int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;                                       \
    int K = kh * kw, P = Ho * Wo, cg = C / groups, og = Co / groups;                                  \
    REAL *col = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * P);                                     \
    if (!col) return -1;                                                                              \
    for (int n = 0; n < N; ++n) {                                                                     \
        im2col_##SUF(x + (size_t)n * C * H * W, offset + (size_t)n * dg * 2 * K * P,                  \
                     mask + (size_t)n * dg * K * P, col, C, H, W, kh, kw, stride, pad, dil, dg, Ho, Wo);\
        for (int o = 0; o < Co; ++o) {                                                                \
            int g = o / og;                                                                           \
            for (int p = 0; p < P; ++p) {                                                             \
                REAL acc = 0;                                                                         \
                for (int ck = 0; ck < cg * K; ++ck)                                                   \
                    acc += weight[(size_t)o * cg * K + ck] * col[((size_t)g * cg * K + ck) * P + p];  \
                out[((size_t)n * Co + o) * P + p] = acc + (bias ? bias[o] : 0);                       \
            }                                                                                         \
        }                                                                                             \
    }                                                                                                 \
    free(col);                                                                                        \
    return 0;                                                                                         \
}                                                                                                     \
/* grad_x, grad_offset, grad_mask are overwritten; grad_weight, grad_bias are accumulated into. */    \
int mdcn_backward_##SUF(const REAL *x, const REAL *offset, const REAL *mask, const REAL *weight,      \
                        const REAL *gout, REAL *gx, REAL *goff, REAL *gmask, REAL *gw, REAL *gb,      \
                        int N, int C, int H, int W, int Co, int kh, int kw, int stride, int pad,      \
                        int dil, int groups, int dg) {                                                \
    int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;                                       \
    int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;                                       \
    int K = kh * kw, P = Ho * Wo, cg = C / groups, og = Co / groups, cpg = C / dg;                    \
    REAL *col = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * P);                                     \
    REAL *gcol = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * P);                                    \
    if (!col || !gcol) return -1;                                                                     \
    memset(gx, 0, sizeof(REAL) * (size_t)N * C * H * W);                                              \
    for (int n = 0; n < N; ++n) {                                                                     \
        const REAL *xn = x + (size_t)n * C * H * W, *on = offset + (size_t)n * dg * 2 * K * P;        \
        const REAL *mn = mask + (size_t)n * dg * K * P, *gn = gout + (size_t)n * Co * P;              \
        /* gcol = W^T gout (cpp:602-605) */                                                           \
        for (int c = 0; c < C; ++c) for (int k = 0; k < K; ++k) for (int p = 0; p < P; ++p) {         \
            int g = c / cg, cl = c % cg;                                                              \
            REAL acc = 0;                                                                             \
            for (int o = 0; o < og; ++o)                                                              \
                acc += weight[((size_t)(g * og + o) * cg + cl) * K + k] * gn[(size_t)(g * og + o) * P + p];\
            gcol[((size_t)c * K + k) * P + p] = acc;                                                  \
        }                                                                                             \
        /* col2im_coord: one (offset channel, pixel) at a time (.cu:634-705) */                       \
        for (int oc = 0; oc < dg * 2 * K; ++oc) for (int p = 0; p < P; ++p) {                         \
            int g = oc / (2 * K), rem = oc % (2 * K), k = rem / 2, dir = rem % 2;                     \
            int i = k / kw, j = k % kw, ho = p / Wo, wo = p % Wo;                                     \
            REAL val = 0, mval = 0;                                                                   \
            for (int cl = 0; cl < cpg; ++cl) {                                                        \
                int c = g * cpg + cl;                                                                 \
                const REAL *im = xn + (size_t)c * H * W;                                              \
                REAL ih = ho * stride - pad + i * dil + on[((size_t)g * 2 * K + 2 * k) * P + p];      \
                REAL iw = wo * stride - pad + j * dil + on[((size_t)g * 2 * K + 2 * k + 1) * P + p];  \
                REAL m = mn[((size_t)g * K + k) * P + p], gc = gcol[((size_t)c * K + k) * P + p];     \
                if (ih <= -1 || iw <= -1 || ih >= H || iw >= W) { ih = iw = -2; }                     \
                else mval += gc * bilinear_##SUF(im, H, W, ih, iw);                                   \
                val += coord_weight_##SUF(ih, iw, H, W, im, dir) * gc * m;                            \
            }                                                                                         \
            goff[((size_t)n * dg * 2 * K + oc) * P + p] = val;                                        \
            if (dir == 0) gmask[((size_t)n * dg * K + g * K + k) * P + p] = mval;                     \
        }                                                                                             \
        /* col2im: scatter into grad_x (.cu:574-631) */                                               \
        for (int c = 0; c < C; ++c) for (int k = 0; k < K; ++k) for (int p = 0; p < P; ++p) {         \
            int g = c / cpg, i = k / kw, j = k % kw, ho = p / Wo, wo = p % Wo;                        \
            REAL ih = ho * stride - pad + i * dil + on[((size_t)g * 2 * K + 2 * k) * P + p];          \
            REAL iw = wo * stride - pad + j * dil + on[((size_t)g * 2 * K + 2 * k + 1) * P + p];      \
            REAL top = gcol[((size_t)c * K + k) * P + p] * mn[((size_t)g * K + k) * P + p];           \
            int ch = (int)ih, cw = (int)iw;                                                           \
            for (int dy = -2; dy <= 2; ++dy) for (int dx = -2; dx <= 2; ++dx) {                       \
                int yy = ch + dy, xx = cw + dx;                                                       \
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && fabs((double)(ih - yy)) < 1 &&          \
                    fabs((double)(iw - xx)) < 1)                                                      \
                    gx[((size_t)n * C + c) * H * W + yy * W + xx] +=                                  \
                        grad_weight_##SUF(ih, iw, yy, xx, H, W) * top;                                \
            }                                                                                         \
        }                                                                                             \
        /* grad_weight += gout col^T ; grad_bias += gout 1 (cpp:626-650) */                           \
        im2col_##SUF(xn, on, mn, col, C, H, W, kh, kw, stride, pad, dil, dg, Ho, Wo);                 \
        for (int o = 0; o < Co; ++o) {                                                                \
            int g = o / og;                                                                           \
            for (int ck = 0; ck < cg * K; ++ck) {                                                     \
                REAL acc = 0;                                                                         \
                for (int p = 0; p < P; ++p) acc += gn[(size_t)o * P + p] * col[((size_t)g * cg * K + ck) * P + p];\
                gw[(size_t)o * cg * K + ck] += acc;                                                   \
            }                                                                                         \
            if (gb) { REAL acc = 0; for (int p = 0; p < P; ++p) acc += gn[(size_t)o * P + p]; gb[o] += acc; }\
        }                                                                                             \
    }                                                                                                 \
    free(col); free(gcol);                                                                            \
    return 0;                                                                                         \
}

DEFINE_MDCN(float, f32)
DEFINE_MDCN(double, f64)
