/* otpose_hip.h - C ABI of libotpose_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the OTPose hot path.  Plain pointers and sizes only:
 * every pointer is a DEVICE pointer to a contiguous NCHW float32 tensor unless stated otherwise,
 * `stream` is a hipStream_t passed as void*.  Functions never allocate, never synchronise, launch
 * on `stream`, and return OTP_OK (0) or a negative OTP_ERR_* code (no exceptions cross the ABI).
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   otp_mdcn_forward   <- modulated_deform_conv_cuda_forward   thirdparty/deform_conv/src/deform_conv_cuda.cpp:474-549
 *                         (+ modulated_deformable_im2col_cuda   src/deform_conv_cuda_kernel.cu:506-571, 707-737)
 *   otp_mdcn_backward  <- modulated_deform_conv_cuda_backward  src/deform_conv_cuda.cpp:551-664
 *                         (+ col2im / col2im_coord kernels      src/deform_conv_cuda_kernel.cu:574-705, 739-805)
 *   everything else    <- the ATen calls issued by model/OTPose.py:307-394, model/HRNet.py:116-152,
 *                         model/blocks.py:95-110,264-280,400-453, model/ConvVideoTransformer.py:123-184,
 *                         model/RSB.py:77-103 and model/loss.py:25-92 (the reference has no native code
 *                         for them; the C entry points below are what a native port of that graph binds).
 */
#ifndef OTPOSE_HIP_H
#define OTPOSE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTP_OK 0
#define OTP_ERR_BAD_ARG (-1)     /* null pointer / non-positive dimension */
#define OTP_ERR_UNSUPPORTED (-2) /* shape, dtype or stride combination not implemented */
#define OTP_ERR_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after the launch */
#define OTP_ERR_WORKSPACE (-4)   /* workspace_bytes too small */

#define OTP_DTYPE_F32 0          /* what OTPose uses: the fused kernels of mdcn.hip */
#define OTP_DTYPE_F16 1          /* the reference dispatches f64 / f32 / f16 (AT_DISPATCH_FLOATING_TYPES_AND_HALF,      */
#define OTP_DTYPE_BF16 2         /* deform_conv_cuda_kernel.cu:719,751,784); f16 / bf16 compute in fp32, f64 in fp64     */
#define OTP_DTYPE_F64 3

#define OTP_ACT_NONE 0
#define OTP_ACT_RELU 1
#define OTP_ACT_GELU 2           /* exact erf GELU (nn.GELU default) */

int otp_version(void);

/* ---- range guard of the 16-bit-operand kernels (csrc/range.hip) -------------------------------------------------------------
 * The split-product ("x3") kernels below carry every fp32 operand as two IEEE-half pieces, and the fp16 engine (otp_h16_*)
 * stores activations as halves: both have fp32's significand budget or less and HALF'S EXPONENT RANGE - |a| >= 65504 becomes
 * inf, its products NaN, and a ReLU or the DCN's bounds test would swallow the NaN.  The reference computes these layers in
 * fp32 (model/HRNet.py:500-530, model/blocks.py:248-254, 400-453) and has no such limit, so it is never crossed silently:
 * every such kernel tests its results before the activation and records a violation in one process-wide sticky word.
 *   otp_range_flag_read(reset): 0 = clean, else the code of a kernel family that saw a non-finite sum or a result >= 65504
 *     since the last reset (1 convx, 2 convs, 3 convs2, 4 pointx, 5 stem, 6 S8 passes, 7 mlpx, 8 densex, 9 attention,
 *     10 dcn_fused, 11 fp16 engine).  Host function; definitive once the launches in question have completed.
 *   otp_range_poison(out, n, stream): launch that fills out[0 .. n) (fp32) with NaN if the word is set - the last launch of a
 *     forward, so that a caller who never asks cannot consume finite-looking heat-maps computed from an overflowed operand. */
int otp_range_flag_read(int reset);
int otp_range_poison(void* out, size_t n, void* stream);

/* ---- fp16-storage eval kernels of the HRNet backbone (csrc/h16.hip; cfg.MODEL.DTYPE = "fp16", BASELINE.json configs[4]) --------
 * The reference's native op dispatches half (deform_conv_cuda_kernel.cu:719) but its model never runs below fp32; this is the
 * documented fp16 extension.  Operands are IEEE half - activations as stored, weights rounded once times a per-layer power of two
 * `pre` (the launch multiplies its sums by out_scale = 1 / pre) - one f16 MFMA per product, fp32 accumulation / shift / residual /
 * ReLU, one rounding to half per stored value.
 * H8 image of a logical (N, C, H, W) tensor, C % 8 == 0: [N][C / 8][H * W] records of 16 bytes = the 8 halves of channels
 * 8 g .. 8 g + 7 at pixel p (otp_h8_bytes = 2 bytes per element).  (gtot, goff) address a range of channel groups of a wider H8
 * tensor (gtot = 0: dense), so channel concatenations cost nothing.
 *   otp_h8_pack / otp_h8_unpack       fp32 NCHW channel slice <-> H8 (round to nearest even / exact widening)
 *   otp_h16_conv3x3                   out = act(conv3x3 pad 1 stride 1|2 (in) + shift (+ res)), H8 -> H8: model/HRNet.py:500-530
 *                                     (BasicBlock), :551-571 (Bottleneck conv2), :213-229 / :442-470 (stride-2 chains), :66-72
 *   otp_h16_pointwise                 out = act(W x + shift (+ res)): H8 -> H8, or -> a channel slice of an fp32 NCHW tensor
 *                                     (out_f32_nchw = 1; out_tot / out_off then count channels): :551-571 conv1 / conv3, :426-439
 *                                     (fuse 1x1), :108-114 (final_layer)
 *   otp_h16_stem                      relu(conv3x3 s2 p1 (frames of the fp32 clip (B, 3 F, H, W)) + shift) -> H8 (F B, Cout, ..): :118-120
 *   otp_h16_upsample_add              out = act(res + sum_k nearest_up_fk(low_k)), all H8: a fuse row's tail, :487-494
 * Every launch feeds the range guard (otp_range_flag_read, code 11).  Never allocates, never synchronises. */
typedef struct otp_h16_conv_desc {
    int N, Cin, H, W, Cout, stride, act;                               /* 3x3, pad 1, dilation 1; Cin % 16 == 0, Cout % 8 == 0 */
    int in_gtot, in_goff, out_gtot, out_goff, res_gtot, res_goff;      /* channel-group slices (gtot = 0: the dense tensor) */
    float out_scale;                                                   /* 1 / pre of otp_h16_conv3x3_pack_weight (0 = 1) */
} otp_h16_conv_desc;
size_t otp_h8_bytes(int N, int C, int H, int W);
int otp_h8_pack(const void* in_f32, void* out_h8, int N, int C, int H, int W, int in_ctot, int in_coff, int out_gtot, int out_goff,
                void* stream);
int otp_h8_unpack(const void* in_h8, void* out_f32, int N, int C, int H, int W, int in_gtot, int in_goff, void* stream);
int otp_h16_conv3x3_supported(const otp_h16_conv_desc* desc);
size_t otp_h16_conv3x3_weight_bytes(int Cout, int Cin);
int otp_h16_conv3x3_pack_weight(const void* weight, const void* scale, void* wpacked, int Cout, int Cin, float pre, void* stream);
int otp_h16_conv3x3(const void* in_h8, const void* wpacked, const void* shift, const void* res_h8, void* out_h8,
                    const otp_h16_conv_desc* desc, void* stream);
int otp_h16_pointwise_supported(int Cin, int Cout);
size_t otp_h16_pointwise_weight_bytes(int Cin, int Cout);
int otp_h16_pointwise_pack(const void* w, const void* scale, const void* shift, void* packed, int Cin, int Cout, float pre,
                           void* stream);
int otp_h16_pointwise(const void* in_h8, const void* packed, const void* res_h8, void* out, int out_f32_nchw, int N, int Cin, int Cout,
                      int HW, int in_gtot, int in_goff, int res_gtot, int res_goff, int out_tot, int out_off, int relu,
                      float out_scale, void* stream);
int otp_h16_stem_supported(int B, int F, int H, int W, int Cout);
size_t otp_h16_stem_weight_bytes(int Cout);
int otp_h16_stem_pack(const void* w, const void* scale, const void* shift, void* packed, int Cout, void* stream);
int otp_h16_stem(const void* in, const void* packed, void* out_h8, int B, int F, int H, int W, int Cout, void* stream);
int otp_h16_upsample_add(const void* const* lows_h8, const int* factors, int nlow, const void* res_h8, void* out_h8, int N, int C,
                         int Hh, int Wh, int relu, void* stream);
/* The temporal encoders' matrix kernels with the same arithmetic (model/blocks.py:248-254, 400-419): fp32 (B, C, T) tensors as in
 * otp_ln_mlp_x3 / otp_dense_x3 / otp_qkv_front_x3 - same arguments; otp_dense_h1 / otp_qkv_front_h1 read the hi pieces of the SAME packed
 * weights, otp_ln_mlp_h1 a hi-only image of its own (otp_mlp_h1_pack: half the LDS, two workgroups per CU at C = 204) - but every operand
 * is rounded to half ONCE, one MFMA per product, the MLP's hidden layer rounded to half behind a 6e-5 GELU; LayerNorm and accumulation fp32. */
size_t otp_mlp_h1_weight_bytes(int C, int HID);                    /* the hi-only image: half of otp_mlp_x3_weight_bytes */
int otp_mlp_h1_pack(const void* w1, const void* b1, const void* w2, void* packed, int C, int HID, void* stream);
int otp_ln_mlp_h1(const void* y, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* packed, const void* scale,
                  const void* shift, void* out, int B, int C, int HID, int T, void* stream);
int otp_dense_h1(const void* const* x, const void* const* packed, const void* const* res, void* const* out, int nprob, int B, int C,
                 int T, void* stream);
int otp_qkv_front_h1(const void* x, const void* table, const void* packed_q, const void* packed_k, const void* packed_v, void* q,
                     void* k, void* v, int B, int C, int T, float eps, void* stream);

/* ---- modulated deformable convolution ------------------------------------------------------
 * out[n,o,p] = beta*out[n,o,p] + alpha*( bias[o] + sum_{c,k} W[o,c,k] * mask[n,g(c)*K+k,p] *
 *              bilinear(x[n,c], p*stride - pad + tap_k*dil + offset[n,g(c)*2K+2k(+1),p]) ),
 * g(c) = c / (C/deformable_groups); sample zero outside the open interval (-1,H)x(-1,W), corners
 * outside the image contribute zero.  alpha=1,beta=0 is the reference operator; alpha/beta fuse the
 * reference's weighted sum over dilations (model/OTPose.py:387-392).  bias may be NULL. */
int otp_mdcn_forward(const void* x, const void* offset, const void* mask, const void* weight,
                     const void* bias, void* out,
                     int N, int C, int H, int W, int Cout, int kh, int kw,
                     int stride, int pad, int dil, int groups, int deformable_groups,
                     float alpha, float beta, int dtype, void* stream);

/* The same operator with the reference's full argument list (deform_conv_cuda.cpp:474-480): independent stride / padding /
 * dilation per axis; `dtype` selects the storage type of EVERY tensor; mask == NULL is DCN v1 (deform_conv_cuda.cpp:148-249:
 * no modulation).  otp_mdcn_forward is the isotropic special case.  fp32 3x3 isotropic calls take the fused kernels, everything
 * else the general form (csrc/mdcn_generic.hip). */
int otp_mdcn_forward_ex(const void* x, const void* offset, const void* mask, const void* weight, const void* bias, void* out,
                        int N, int C, int H, int W, int Cout, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                        int dil_h, int dil_w, int groups, int deformable_groups, float alpha, float beta, int dtype,
                        void* stream);

/* grad_x / grad_offset / grad_mask are overwritten, grad_weight / grad_bias are ACCUMULATED into
 * (the reference accumulates them over the batch, cpp:638-650; the caller zeroes them, reference
 * functions/deform_conv.py:152-156).  grad_bias may be NULL.  workspace: otp_mdcn_backward_workspace bytes. */
size_t otp_mdcn_backward_workspace(int N, int C, int H, int W, int Cout, int kh, int kw);
/* exact workspace of one otp_mdcn_backward_ex call (otp_mdcn_backward_workspace is the upper bound over every geometry with
 * that input size).  The backward holds NO order-dependent arithmetic (the reference's col2im scatters with float atomicAdd,
 * deform_conv_cuda_kernel.cu:612-629): the fused fp32 3x3 form accumulates grad_x in 64-bit fixed point and adds per-workgroup
 * partial sums of grad_x / grad_weight / grad_bias, kept in this workspace, in a fixed order - two calls on the same inputs
 * return the same bits. */
size_t otp_mdcn_backward_workspace_ex(int N, int C, int H, int W, int Cout, int kh, int kw, int stride_h, int stride_w, int pad_h,
                                      int pad_w, int dil_h, int dil_w, int groups, int deformable_groups, int dtype,
                                      int has_mask);
int otp_mdcn_backward(const void* x, const void* offset, const void* mask, const void* weight,
                      const void* grad_out, void* grad_x, void* grad_offset, void* grad_mask,
                      void* grad_weight, void* grad_bias, void* workspace, size_t workspace_bytes,
                      int N, int C, int H, int W, int Cout, int kh, int kw,
                      int stride, int pad, int dil, int groups, int deformable_groups,
                      int dtype, void* stream);
/* full argument list of modulated_deform_conv_cuda_backward (deform_conv_cuda.cpp:551-558) and, with mask == grad_mask ==
 * NULL, of deform_conv_backward_input_cuda + deform_conv_backward_parameters_cuda (cpp:251-472): any Cout, any kernel. */
int otp_mdcn_backward_ex(const void* x, const void* offset, const void* mask, const void* weight, const void* grad_out,
                         void* grad_x, void* grad_offset, void* grad_mask, void* grad_weight, void* grad_bias, void* workspace,
                         size_t workspace_bytes, int N, int C, int H, int W, int Cout, int kh, int kw, int stride_h, int stride_w,
                         int pad_h, int pad_w, int dil_h, int dil_w, int groups, int deformable_groups, int dtype, void* stream);

/* ---- dense convolution (implicit GEMM on the f32 matrix cores) --------------------------------
 * out[n, out_coff+co, ho, wo] = act( scale[co] * conv(in (+ in2))[n,co,ho,wo] + shift[co] (+ res) )
 * groups = 1.  Tensors may be channel slices of wider tensors: `*_ctot` is the channel count of the
 * tensor the pointer addresses, `*_coff` the first channel used.  `res` may alias `out` (in-place
 * accumulate).  res_up = f > 1: the conv result is nearest-upsampled by f and res/out live on the
 * (Ho*f, Wo*f) grid (HRNet fuse layers, model/HRNet.py:426-439,488-494).  frame_split = B > 0: `in`
 * is the (B, 5*Cin, H, W) clip tensor read as (5B, Cin, H, W) with image n = f*B + b taken from
 * channels [f*Cin, (f+1)*Cin) of sample b (model/OTPose.py:317).
 * `wpacked` comes from otp_conv2d_pack_weight ([kh*kw][Cin][Cout16] floats, Cout16 = Cout rounded up
 * to 16).  scale / shift may be NULL (1 / 0). */
typedef struct otp_conv_desc {
    int N, Cin, H, W, Cout, kh, kw, stride, pad, dil;
    int in_ctot, in_coff, in2_ctot, in2_coff, out_ctot, out_coff;
    int res_ctot, res_coff, res_up, act, Ho, Wo, frame_split;
    /* split-product kernels only (otp_conv2d_x3, otp_conv3x3_s8, otp_conv3x3_s2_s8): the packed weights carry a power-of-two
     * factor 2^k (folded into the `scale` vector handed to the packer) so that BOTH half pieces of every weight are normal
     * numbers - 22 significand bits per weight instead of ~17 for BatchNorm-folded weights of magnitude 0.01 - and the kernel
     * multiplies the accumulated sum by out_scale = 2^-k before shift / residual / activation.  0 means 1. */
    float out_scale;
    /* otp_conv3x3_s8 only: layout of the residual image.  0: C4 fp32 image [N][Cout / 4][H * W][4] (the default);
     * 1: S8 records [N][Cout / 8][hi | lo][H * W] - a block's input image serves as its residual (hi + lo holds it to 2^-22),
     * so a chain of BasicBlocks needs no fp32 image between its blocks. */
    int res_layout;
} otp_conv_desc;

/* tuning / test hook: force the (M-blocks, pixel-blocks, waves-in-M, waves-in-pixels) tile of otp_conv2d;
 * all zeros restores the built-in choice.  Results never depend on it. */
int otp_conv2d_set_tile(int MB, int PB, int WM, int WP);
/* tuning hook: {MB, PB, WM, WP, CK, grid, lds bytes, tiles} of the last otp_conv2d call (zeros: generic kernel) */
int otp_conv2d_last_plan(int* out8);
/* the same for a descriptor, without launching anything (host arithmetic only; usable without a GPU) */
int otp_conv2d_plan(const otp_conv_desc* desc, int* out8);
int otp_conv2d_pack_weight(const void* weight, void* wpacked, int Cout, int Cin, int kh, int kw, void* stream);
int otp_conv2d(const void* in, const void* in2, const void* wpacked, const void* scale, const void* shift,
               const void* res, void* out, const otp_conv_desc* desc, void* stream);

/* 3x3 / stride 1 / pad 1 convolutions (the HRNet BasicBlock / Bottleneck convs, model/HRNet.py:500-571) in Winograd
 * F(2x2, 3x3) form: 2.25x fewer multiplies on the same f32 matrix cores, fp32 throughout, same fused epilogue
 * (scale / shift, residual, activation, channel-sliced views) as otp_conv2d.  upacked = otp_conv2d_wino_weight_bytes(Cout, Cin)
 * bytes filled by otp_conv2d_wino_pack_weight from the (Cout, Cin, 3, 3) weights.  otp_conv2d_wino_supported tells whether
 * a descriptor qualifies (3x3, stride 1, pad 1, dilation 1, H*W % 4 == 0, no res_up / frame_split / second input). */
size_t otp_conv2d_wino_weight_bytes(int Cout, int Cin);
int otp_conv2d_wino_pack_weight(const void* weight, void* upacked, int Cout, int Cin, void* stream);
int otp_conv2d_wino_supported(const otp_conv_desc* desc);
/* tuning hook: {16-tile blocks per workgroup, channels per chunk, grid (workgroups), LDS bytes} of the last otp_conv2d_wino */
int otp_conv2d_wino_last_plan(int* out4);
int otp_conv2d_wino(const void* in, const void* upacked, const void* scale, const void* shift, const void* res, void* out,
                    const otp_conv_desc* desc, void* stream);

/* 3x3 / stride 1 / pad 1 convolutions with at most 24 input and 24 output channels and an optional pre-added second input
 * (the RSB staircase convs, model/RSB.py:80-92): one thread per output pixel, exact fp32 FMAs, same scale / shift / ReLU
 * epilogue and channel-sliced views as otp_conv2d (no residual).  `wpacked`: otp_conv3x3_small_weight_bytes bytes written by
 * otp_conv3x3_small_pack from the plain (Cout, Cin, 3, 3) tensor ([ci][tap][co] order, read through the scalar cache). */
int otp_conv3x3_small_supported(const otp_conv_desc* desc);
size_t otp_conv3x3_small_weight_bytes(int Cout, int Cin);
int otp_conv3x3_small_pack(const void* weight, void* wpacked, int Cout, int Cin, void* stream);
int otp_conv3x3_small(const void* in, const void* in2, const void* wpacked, const void* scale, const void* shift, void* out,
                      const otp_conv_desc* desc, void* stream);

/* fp32 3x3 (stride 1 or 2) and 1x1 (stride 1) convolutions on the 16-bit matrix cores with split ("f16x3": two IEEE-half pieces per operand, csrc/common.h) products: every fp32 operand is the sum
 * of two IEEE-half pieces (hi = rne(a), lo = rne(a - hi): 22 significand bits while lo is a normal number - see out_scale in
 * otp_conv_desc for the weights), a product is lo*hi + hi*lo + hi*hi accumulated in fp32, storage stays fp32 NCHW.  Same descriptor and fused epilogue (shift, residual, activation,
 * channel-sliced views) as otp_conv2d_wino; the per-channel scale is folded into the packed weights.  Replaces the cuDNN
 * convs behind model/HRNet.py:500-571 (BasicBlock / Bottleneck conv2) and the 3x3 transition / dilated convs.
 * otp_conv2d_x3_supported: 3x3 with stride 1 (Cin % 16 == 0) or 2 (Cin % 8 == 0), any pad / dilation whose window fits the
 * LDS, or 1x1 with stride 1 and no padding (Cin % 32 == 0); H*W % 4 == 0, Ho*Wo % 4 == 0, no second input / res_up / GELU. */
/* the packed layout depends on the kernel size k (1 or 3) and the stride (channels per chunk: 16 / 8 for 3x3 stride 1 / 2,
 * 32 for 1x1) */
size_t otp_conv2d_x3_weight_bytes(int Cout, int Cin, int k, int stride);
int otp_conv2d_x3_pack_weight(const void* weight, const void* scale, void* wpacked, int Cout, int Cin, int k, int stride,
                              void* stream);
int otp_conv2d_x3_supported(const otp_conv_desc* desc);
int otp_conv2d_x3(const void* in, const void* wpacked, const void* shift, const void* res, void* out,
                  const otp_conv_desc* desc, void* stream);

/* ---- split-record ("S8") activations + LDS-DMA fed 3x3 convolutions (csrc/convs.hip) ----------------------------------------
 * The 3x3 / stride 1 / pad 1 convs of the HRNet BasicBlocks (model/HRNet.py:500-530: conv1 + bn1 + relu, conv2 + bn2 +
 * residual + relu; 73 % of the forward's MACs) with the split-product arithmetic of otp_conv2d_x3, on activations the
 * PRODUCER already stored as MFMA operand records:
 *   S8 image of a logical (N, C, H, W) fp32 tensor, C % 8 == 0:  [N][C/8][2][H*W] records of 16 bytes = 8 bf16;
 *   record (n, g, part, p): part 0 = hi = rne_bf16(x), part 1 = lo = rne_bf16(x - hi) of channels 8g .. 8g+7 at pixel p
 *   (4 bytes per element, otp_s8_bytes);
 *   C4 image of the same tensor: [N][C/4][H*W][4] fp32 - the layout a lane's accumulators have (4 output channels of one
 *   pixel), used for the residual chain inside a branch: no transposition in any epilogue.
 * otp_s8_pack converts a channel slice of an fp32 NCHW tensor to S8 (and, with out_c4 != NULL, to C4); otp_s8_unpack gives
 * hi + lo back as fp32 NCHW, otp_c4_unpack the C4 image (tests).
 * otp_conv3x3_s8: out = act(conv3x3(in) [scale folded into wpacked] + shift (+ res)); `in_s8` an S8 image, `wpacked` from
 * otp_conv3x3_s8_pack_weight, `res_c4` a C4 image of (N, Cout, H, W) or NULL, `out_f32` NULL or the fp32 result as a C4 image
 * (out_f32_layout = OTP_S8_F32_C4) or as a channel slice of an NCHW tensor (OTP_S8_F32_NCHW, desc->out_ctot / out_coff),
 * `out_s8` NULL or the S8 image of the result.  desc: kh = kw = 3, stride 1, pad 1, dil 1, Cin % 16 == 0, Cout % 8 == 0,
 * H*W % 4 == 0, act NONE / RELU.  Never allocates, never synchronises. */
#define OTP_S8_F32_C4 1
#define OTP_S8_F32_NCHW 2
size_t otp_s8_bytes(int N, int C, int H, int W);
int otp_s8_pack(const void* in_f32, void* out_s8, void* out_c4, int N, int C, int H, int W, int in_ctot, int in_coff, void* stream);
/* otp_upsample_add_multi (a fuse row's upsampled terms, model/HRNet.py:487-494) writing the S8 image of its result - what the
 * next module's branch reads - the C4 image when out_c4 != NULL (a branch that takes its residual from an fp32 image:
 * otp_conv_desc.res_layout = 0) and the NCHW tensor only when out_nchw != NULL; same additions in the same order */
int otp_s8_upsample_add(const void* const* lows, const int* factors, int nlow, const void* res, void* out_nchw, void* out_s8,
                        void* out_c4, int N, int C, int Hh, int Wh, int relu, int res_ctot, int res_coff, int out_ctot,
                        int out_coff, void* stream);
/* the same with the residual given as fp32 NCHW (res_layout 0) or as its S8 image (res_layout 1: hi + lo of the records; the
 * res_ctot / res_coff arguments are then unused) - a branch output that every consumer reads as S8 needs no NCHW tensor */
int otp_s8_upsample_add_ex(const void* const* lows, const int* factors, int nlow, const void* res, int res_layout, void* out_nchw,
                           void* out_s8, void* out_c4, int N, int C, int Hh, int Wh, int relu, int res_ctot, int res_coff,
                           int out_ctot, int out_coff, void* stream);
int otp_s8_unpack(const void* in_s8, void* out_f32, int N, int C, int H, int W, void* stream);
int otp_c4_unpack(const void* in_c4, void* out_f32, int N, int C, int H, int W, void* stream);
int otp_conv3x3_s8_supported(const otp_conv_desc* desc);
size_t otp_conv3x3_s8_weight_bytes(int Cout, int Cin);
int otp_conv3x3_s8_pack_weight(const void* weight, const void* scale, void* wpacked, int Cout, int Cin, void* stream);
int otp_conv3x3_s8(const void* in_s8, const void* wpacked, const void* shift, const void* res_c4, void* out_f32,
                   int out_f32_layout, void* out_s8, const otp_conv_desc* desc, void* stream);
/* The STRIDE 2 form (csrc/convs2.hip): the down-sampling chains of the fuse layers (model/HRNet.py:442-470), the transition
 * layers (:192-231) and the stem's second conv (:66-72) from an S8 image, same packed weights (otp_conv3x3_s8_pack_weight).
 * The window is staged with its columns de-interleaved by parity, so the fragment reads of 16 consecutive output pixels stay
 * conflict-free.  Exactly one output form per call: `out_s8` = the S8 image of act(conv + shift) (a chain's intermediate:
 * conv + BN + ReLU, no residual), or `out_nchw` = a channel slice of an fp32 NCHW tensor (desc->out_ctot / out_coff) holding
 * act(conv + shift + res), `res_nchw` NULL or a channel slice (desc->res_ctot / res_coff) that may be the output itself (a
 * fuse row accumulates in place).  desc: kh = kw = 3, stride 2, pad 1, dil 1, H and W even, Cin % 16 == 0, Cout % 16 == 0,
 * Ho*Wo % 4 == 0, act NONE / RELU. */
int otp_conv3x3_s2_s8_supported(const otp_conv_desc* desc, int nchw_out);
int otp_conv3x3_s2_s8(const void* in_s8, const void* wpacked, const void* shift, const void* res_nchw, void* out_nchw,
                      void* out_s8, const otp_conv_desc* desc, void* stream);

/* ---- training-step building blocks for the convolutional layers (script/Common.py:91,136-144 run the reference under
 * model.train(): BatchNorm2d uses batch statistics, every conv needs both gradients) -----------------------------
 * Gradient w.r.t. the input of a stride-1 conv = otp_conv2d of grad_out with the weights packed by
 * otp_conv2d_pack_weight_dgrad (flipped taps, channels transposed: a Cout -> Cin conv, pad' = dil*(k-1) - pad);
 * for stride s > 1 grad_out is first zero-inserted to the input resolution with otp_dilate. */
int otp_conv2d_pack_weight_dgrad(const void* weight, void* wpacked, int Cout, int Cin, int kh, int kw, void* stream);
/* out (planes, H, W) = 0 except out[:, y*s, x*s] = in[:, y, x]; in (planes, Hi, Wi) */
int otp_dilate(const void* in, void* out, int planes, int Hi, int Wi, int s, int H, int W, void* stream);
/* grad_weight (Cout, Cin, kh, kw) += sum over batch and pixels of grad_out x shifted input (ACCUMULATES: zero it first).
 * x (N, x_ctot, H, W) channels [x_coff, x_coff+Cin); grad_out (N, dy_ctot, Ho, Wo) channels [dy_coff, dy_coff+Cout).
 * 1x1 and 3x3 kernels, any stride / padding / dilation.
 * workspace (otp_conv2d_wgrad_workspace(Cin, Cout) bytes, contents irrelevant) holds the per-workgroup partial sums
 * that a second launch folds into grad_weight; workspace == NULL selects one float atomic per weight and workgroup
 * instead (same result up to summation order, several times slower at the HRNet sizes). */
size_t otp_conv2d_wgrad_workspace(int Cin, int Cout);
int otp_conv2d_wgrad(const void* x, const void* grad_out, void* grad_weight, int N, int Cin, int H, int W, int Cout,
                     int kh, int kw, int stride, int pad, int dil, int x_ctot, int x_coff, int dy_ctot, int dy_coff,
                     void* workspace, size_t workspace_bytes, void* stream);
/* BatchNorm2d, training mode (torch semantics: biased variance for normalisation, unbiased for running_var,
 * running = (1-momentum)*running + momentum*batch), fused with the residual add and ReLU that follow it:
 *   y = relu?( (x - mean_c) * rstd_c * gamma_c + beta_c (+ res) );  save_mean / save_rstd (C) feed the backward.
 * running_mean / running_var may be NULL.  workspace: otp_bn_workspace bytes. */
size_t otp_bn_workspace(int N, int C, int HW);
int otp_bn_train_forward(const void* x, const void* gamma, const void* beta, const void* res, void* y, void* save_mean,
                         void* save_rstd, void* running_mean, void* running_var, void* workspace, size_t workspace_bytes,
                         int N, int C, int HW, float eps, float momentum, int relu, int x_ctot, int x_coff, int res_ctot,
                         int res_coff, int y_ctot, int y_coff, void* stream);
/* g = grad_y * (y_relu > 0) (y_relu NULL: no ReLU);  grad_res (optional, dense (N,C,HW)) = g;
 * grad_gamma = sum g*xhat, grad_beta = sum g (overwritten);  grad_x (dense) = gamma*rstd*(g - grad_beta/n - xhat*grad_gamma/n) */
int otp_bn_train_backward(const void* grad_y, const void* x, const void* y_relu, const void* save_mean,
                          const void* save_rstd, const void* gamma, void* grad_x, void* grad_res, void* grad_gamma,
                          void* grad_beta, void* workspace, size_t workspace_bytes, int N, int C, int HW, int dy_ctot,
                          int dy_coff, int x_ctot, int x_coff, int y_ctot, int y_coff, void* stream);
/* out[c] = sum over (n, p) of a[n, a_coff + c, p] (conv bias gradients); workspace: otp_bn_workspace + C*4 bytes */
int otp_channel_sum(const void* a, void* out, void* workspace, size_t workspace_bytes, int N, int C, int HW, int a_ctot,
                    int a_coff, void* stream);

/* ---- offset / mask convolutions fused into the deformable-convolution gather ------------------
 * All ND dilations of the warping head (model/OTPose.py:381-392) in one launch:
 *   out = alpha * sum_i [ ModulatedDeformConv_i(x, Conv_off_i(trans), Conv_mask_i(trans)) + bias_i ]
 * trans (B, 32, H, W), x = def_heatmaps (B, J, H, W), out (B, J, H, W), all fp32; 3x3 kernels, stride 1, padding = dilation,
 * deformable_groups = J, groups = 1 (the only form model/OTPose.py:141-157 builds).  The 27 J offset / mask channels per pixel
 * and dilation exist only in registers / LDS; the 32 -> 27 J convolutions run on the 16-bit matrix cores with split products
 * (see otp_conv2d_x3), the sampling follows deform_conv_cuda_kernel.cu:403-432, 549-556 exactly.
 * packed: otp_dcn_fused_weight_bytes(ND, J) bytes written by otp_dcn_fused_pack from DEVICE arrays of ND device pointers
 * (w_off[i] (18 J, 32, 3, 3), w_mask[i] (9 J, 32, 3, 3), w_dcn[i] (J, J, 3, 3), bias[i] (J) or NULL).
 * workspace: otp_dcn_fused_workspace(B, H, W) bytes (the bf16 hi / lo NHWC copy of trans).
 * otp_dcn_fused_supported: Cin == 32, J == 17, H * W % 128 == 0, ND <= 8. */
int otp_dcn_fused_supported(int Cin, int J, int H, int W, int ND);
size_t otp_dcn_fused_weight_bytes(int ND, int J);
int otp_dcn_fused_pack(const void* const* w_off, const void* const* w_mask, const void* const* w_dcn, const void* const* bias,
                       void* packed, int ND, int J, void* stream);
size_t otp_dcn_fused_workspace(int B, int H, int W);
int otp_dcn_fused_forward(const void* trans, const void* x, const void* packed, void* out, void* workspace,
                          size_t workspace_bytes, int B, int Cin, int J, int H, int W, const int* dilations, int ND, float alpha,
                          void* stream);

/* ---- OTPose glue (model/OTPose.py:317-359) ------------------------------------------------------ */
/* rough (5B,J,HW) -> total (B,J,HW), squeezed (B,J,HW), intersection (B,J,HW), flow_in = total + pe (pe: (J,HW)) */
int otp_glue_total(const void* rough, void* total, void* squeezed, void* inter, void* flow_in, const void* pe,
                   int B, int J, int HW, void* stream);
/* builds x1, x2 (B, 8J, HW) (+ pe (8J,HW)) and prev_b (B,J,HW); margin is (B,4) float32 */
int otp_glue_stack(const void* rough, const void* margin, const void* squeezed, const void* inter,
                   const void* ctx, const void* pe1, const void* pe2, void* x1, void* x2, void* prev_b,
                   int B, int J, int HW, void* stream);

/* the same for a window of F = 5 (reference) or 7 frames (BASELINE configs[4] extension: frames cur, prev_1, next_1, ...,
 * margin (B, F-1), 12 stacked maps per joint; the map list is oracle/otpose_oracle.py:window_maps) */
int otp_glue_total_n(const void* rough, void* total, void* squeezed, void* inter, void* flow_in, const void* pe,
                     int B, int J, int HW, int F, void* stream);
int otp_glue_stack_n(const void* rough, const void* margin, const void* squeezed, const void* inter,
                     const void* ctx, const void* pe1, const void* pe2, void* x1, void* x2, void* prev_b,
                     int B, int J, int HW, int F, void* stream);

/* ---- ConvTransformer pieces (model/blocks.py) ---------------------------------------------------- */
/* channel LayerNorm over C of (B,C,T) (blocks.py:95-110); optional second output
 * pool = MaxPool1d(3,2,1)(x) (blocks.py:234-238) when pool != NULL (then T must be the input length) */
int otp_ln_channel(const void* x, const void* gamma, const void* beta, void* y, void* pool,
                   int B, int C, int T, float eps, void* stream);
/* three depthwise k=3 convs (stride s, pad 1, no bias) of the same input, each followed by a channel
 * LayerNorm (blocks.py:406-415): x (B,C,T) -> q,k,v (B,C,To), To = (T+2-3)/s+1.  dw* are (C,3). */
int otp_dwconv_ln3(const void* x, const void* dwq, const void* dwk, const void* dwv,
                   const void* gq, const void* bq, const void* gk, const void* bk, const void* gv, const void* bv,
                   void* q, void* k, void* v, int B, int C, int T, int stride, float eps, void* stream);
/* channel attention (blocks.py:427-447): per (b, head): S = (q*scale) k^T over T (hs x hs), P = softmax(S),
 * O = P v, written as out[b][head][t][ch] (the transpose(2,3).contiguous().view(B,C,T) layout). */
size_t otp_chan_attn_workspace(int B, int C, int T, int n_head);
int otp_chan_attn(const void* q, const void* k, const void* v, void* out, void* workspace, size_t workspace_bytes,
                  int B, int C, int T, int n_head, float scale, void* stream);
/* nn.Upsample(scale_factor=f, mode='linear', align_corners=False) on (B,C,T) -> out (B, out_ctot, T*f) at
 * channel offset out_coff (f = 1 copies) (ConvVideoTransformer.py:108,179; OTPose.py:362-369) */
int otp_upsample_linear(const void* x, void* out, int B, int C, int T, int f, int out_ctot, int out_coff, void* stream);

/* TransformerBlock MLP in one launch (model/blocks.py:248-254 applied at :277-279, eval mode):
 *   out = res + scale * (W2 . gelu(W1 . x + b1)) + shift,   x / res / out (B, C, T), W1 (HID, C), W2 (C, HID),
 * scale = AffineDropPath scale (C), shift = b2 * scale (C); the hidden (B, HID, T) activation never reaches HBM.
 * `packed` = otp_mlp_fused_pack's image of (W1, b1, W2) (otp_mlp_fused_weight_bytes bytes, 16-byte aligned).
 * otp_mlp_fused_supported: 1 when a kernel is instantiated for (C, HID) and T qualifies (C = 136, HID = 544, T even);
 * otherwise the caller issues the two otp_conv2d launches.  `res` may alias `out`. */
int otp_mlp_fused_supported(int C, int HID, int T);
size_t otp_mlp_fused_weight_bytes(int C, int HID);
int otp_mlp_fused_pack(const void* w1, const void* b1, const void* w2, void* packed, int C, int HID, void* stream);
int otp_mlp_fused(const void* x, const void* packed, const void* scale, const void* shift, const void* res, void* out,
                  int B, int C, int HID, int T, void* stream);
/* the same with TransformerBlock.ln2 in front (model/blocks.py:277: out = y + drop_path_mlp(mlp(ln2(y)))):
 *   out = y + scale * (W2 . gelu(W1 . LN(y) + b1)) + shift, LN over channels with gamma / beta (C), eps inside the root. */
int otp_ln_mlp_fused(const void* y, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* packed,
                     const void* scale, const void* shift, void* out, int B, int C, int HID, int T, void* stream);

/* otp_dense_cc / otp_qkv_front (csrc/dense.hip) with split-half ("f16x3": two IEEE-half pieces per fp32 operand, csrc/common.h) products on the 16-bit matrix cores
 * (csrc/densex.hip): same arguments; the packed images come from otp_dense_x3_pack (otp_dense_x3_weight_bytes bytes each),
 * the parameter table of otp_qkv_front_x3 is the one otp_qkv_front_pack_table writes. */
int otp_dense_x3_supported(int C, int T);
size_t otp_dense_x3_weight_bytes(int C);
int otp_dense_x3_pack(const void* w, const void* scale, const void* shift, void* packed, int C, void* stream);
int otp_dense_x3(const void* const* x, const void* const* packed, const void* const* res, void* const* out, int nprob, int B,
                 int C, int T, void* stream);
/* The same projection with BFLOAT16 operand pieces (csrc/densex_grad.hip), for operands of unknown magnitude: the training
 * backward's input gradient dx = W^T dy (otpose_amd/train_ops.py).  An IEEE-half piece flushes 1e-8 to zero and holds 1e-5 to
 * 8 bits; a bfloat16 pair keeps 16-17 bits at any magnitude.  Same arguments; images from otp_dense_x3_pack_bf16p. */
int otp_dense_x3_pack_bf16p(const void* w, const void* scale, const void* shift, void* packed, int C, void* stream);
int otp_dense_x3_bf16p(const void* const* x, const void* const* packed, const void* const* res, void* const* out, int nprob, int B,
                 int C, int T, void* stream);
int otp_qkv_front_x3(const void* x, const void* table, const void* packed_q, const void* packed_k, const void* packed_v,
                     void* q, void* k, void* v, int B, int C, int T, float eps, void* stream);

/* HRNet's first convolution (model/HRNet.py:33-36, :118-120) on the frames of the clip tensor (model/OTPose.py:317), csrc/stem.hip:
 *   out (F * B, Cout, Ho, Wo) = relu(scale * conv3x3_stride2_pad1(frame n of in) + shift), frame n = f * B + b = channels
 *   [3 f, 3 f + 3) of clip b of in (B, 3 F, H, W) fp32; Ho = (H - 1) / 2 + 1, Wo likewise (a multiple of 4), Cout <= 64.
 * Split-bf16 products like otp_conv2d_x3; packed: otp_stem_conv_x3_weight_bytes(Cout) bytes from otp_stem_conv_x3_pack
 * (w: (Cout, 3, 3, 3) fp32, scale / shift: the folded BatchNorm, NULL = 1 / 0). */
int otp_stem_conv_x3_supported(int B, int F, int H, int W, int Cout);
size_t otp_stem_conv_x3_weight_bytes(int Cout);
int otp_stem_conv_x3_pack(const void* w, const void* scale, const void* shift, void* packed, int Cout, void* stream);
int otp_stem_conv_x3(const void* in, const void* packed, void* out, int B, int F, int H, int W, int Cout, void* stream);

/* Pointwise (1x1, stride 1) convolution Cin -> Cout on fp32 NCHW channel slices with split-half products (csrc/pointx.hip):
 * HRNet layer1's Bottleneck convs (model/HRNet.py:551-571: 256 -> 64, 64 -> 256 + residual, the shortcut folded over the
 * concatenation 128 -> 256), what nn.Conv2d(k = 1) + folded BatchNorm2d + ReLU compute there:
 *   out[b, out_coff + o, t] = act(scale[o] * sum_c w[o, c] * x[b, x_coff + c, t] + shift[o] (+ res[b, res_coff + o, t]))
 * x / res / out: (B, ctot, T) fp32, T = H * W pixels; 16 <= Cin <= 256 (held by the 64 / 128 / 256-channel instantiation, zero
 * weights past Cin), Cout <= 256, T even;
 * packed: otp_pointwise_x3_weight_bytes(Cin, Cout) bytes from otp_pointwise_x3_pack (w: (Cout, Cin) fp32; scale / shift may
 * be NULL = 1 / 0); relu != 0 clamps at zero.  OTP_ERR_UNSUPPORTED for other shapes (the caller keeps otp_conv2d_x3). */
int otp_pointwise_x3_supported(int Cin, int Cout, int T);
size_t otp_pointwise_x3_weight_bytes(int Cin, int Cout);
int otp_pointwise_x3_pack(const void* w, const void* scale, const void* shift, void* packed, int Cin, int Cout, void* stream);
int otp_pointwise_x3(const void* x, const void* packed, const void* res, void* out, int B, int Cin, int Cout, int T, int x_ctot,
                     int x_coff, int res_ctot, int res_coff, int out_ctot, int out_coff, int relu, void* stream);
/* The same convolution writing the S8 image of its result ([B][Cout / 8][hi | lo][T] records, otp_s8_bytes(B, Cout, H, W) bytes:
 * the input format of otp_conv3x3_s8) instead of an fp32 tensor - a Bottleneck's conv1 in front of its 3x3 conv2.  Cin in
 * {64, 256}, Cout a multiple of 32 (<= 256), T a multiple of 4; packed from otp_pointwise_x3_s8_pack (its own row order). */
int otp_pointwise_x3_s8_supported(int Cin, int Cout, int T);
size_t otp_pointwise_x3_s8_weight_bytes(int Cin, int Cout);
int otp_pointwise_x3_s8_pack(const void* w, const void* scale, const void* shift, void* packed, int Cin, int Cout, void* stream);
int otp_pointwise_x3_s8(const void* x, const void* packed, void* out_s8, int B, int Cin, int Cout, int T, int x_ctot, int x_coff,
                        int relu, void* stream);
/* + an fp32 NCHW residual (channel slice r_coff .. r_coff + Cout of r_ctot) added before the activation: a Bottleneck's conv3
 * (model/HRNet.py:566-571) whose result is read as S8 records only */
int otp_pointwise_x3_s8_res(const void* x, const void* packed, const void* res, void* out_s8, int B, int Cin, int Cout, int T,
                            int x_ctot, int x_coff, int r_ctot, int r_coff, int relu, void* stream);

/* The same operator with split-half ("f16x3": two IEEE-half pieces per fp32 operand, csrc/common.h) products on the 16-bit matrix cores (csrc/mlpx.hip): fp32 storage, fp32
 * accumulation, LayerNorm / bias / GELU in fp32; each product is lo*hi + hi*lo + hi*hi of two bf16 pieces per operand.
 * Own packed image (otp_mlp_x3_weight_bytes / otp_mlp_x3_pack); same arguments and aliasing rules as otp_mlp_fused /
 * otp_ln_mlp_fused. */
int otp_mlp_x3_supported(int C, int HID, int T);
size_t otp_mlp_x3_weight_bytes(int C, int HID);
int otp_mlp_x3_pack(const void* w1, const void* b1, const void* w2, void* packed, int C, int HID, void* stream);
int otp_mlp_x3(const void* x, const void* packed, const void* scale, const void* shift, const void* res, void* out, int B,
               int C, int HID, int T, void* stream);
int otp_ln_mlp_x3(const void* y, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* packed,
                  const void* scale, const void* shift, void* out, int B, int C, int HID, int T, void* stream);

/* TransformerBlock of the flow encoder (C = num_joints = 17 channels, stride 1; model/blocks.py:264-280, 400-453 as built
 * by model/OTPose.py:209-216) around the channel attention as two launches (csrc/flowenc.hip): a thread owns a time step
 * and its 17 channels, weights arrive through the scalar cache.
 *   otp_flow_front: q, k, v (B, C, T) = Conv1d_1x1(LayerNorm(dwconv3(ln1(x))))  [blocks.py:411-419 behind :272]
 *   otp_flow_back : out = y + s_m * (W_2 gelu(W_1 ln2(y) + b_1) + b_2),  y = x + s_a * (W_p att + b_p)   [blocks.py:450, 272-279]
 * with `att` the (B, C, T) view of otp_chan_attn's output.  Parameter blocks (fp32, device memory):
 *   front: ln1 gamma[C], beta[C]; for q, k, v: dw[C][3], norm gamma[C], norm beta[C], W[C][C] (out, in), bias[C]
 *   back : W_p[C][C] * s_a[out], b_p[C] * s_a; ln2 gamma[C], beta[C]; W_1[H][C], b_1[H]; W_2^T[H][C] * s_m[out], b_2[C] * s_m
 * (otp_flow_*_param_floats give their sizes; 0 = unsupported C). */
int otp_flow_block_supported(int C, int hidden, int T);
size_t otp_flow_front_param_floats(int C);
size_t otp_flow_back_param_floats(int C, int hidden);
int otp_flow_front(const void* x, const void* params, void* q, void* k, void* v, int B, int C, int T, float eps, void* stream);
int otp_flow_back(const void* x, const void* att, const void* params, void* out, int B, int C, int hidden, int T, float eps,
                  void* stream);

/* The C -> C pointwise projections of MaskedMHCA (query / key / value / proj: model/blocks.py:383-386, applied at :417-419
 * and :450) on (B, C, T) tensors, nprob (1..3) independent problems of one shape per launch:
 *   out[p] = scale[p] * (W[p] . x[p]) + shift[p] (+ res[p]);   res may be NULL, or hold NULL entries.
 * `packed[p]` = otp_dense_cc_pack's image of (W (C, C), scale (C) or NULL = 1, shift (C) or NULL = 0), 16-byte aligned,
 * otp_dense_cc_weight_bytes(C) bytes.  otp_dense_cc_supported: 1 when a kernel is instantiated (C = 136, T even);
 * otherwise the caller uses otp_conv2d.  x / res / out are arrays of nprob device pointers (host memory). */
int otp_dense_cc_supported(int C, int T);
size_t otp_dense_cc_weight_bytes(int C);
int otp_dense_cc_pack(const void* w, const void* scale, const void* shift, void* packed, int C, void* stream);
int otp_dense_cc(const void* const* x, const void* const* packed, const void* const* res, void* const* out, int nprob,
                 int B, int C, int T, void* stream);

/* Attention front end of MaskedMHCA.forward for stride 1 (model/blocks.py:406-419): q, k, v = W_p . LN_p(dwconv3_p(x)) + b_p,
 * i.e. otp_dwconv_ln3 followed by the three projections, in one launch without the (B, C, T) intermediates.  x, q, k, v
 * (B, C, T); `table` = otp_qkv_front_pack_table's image of the three depthwise weights (C, 3) and LayerNorm gamma / beta (C)
 * (otp_qkv_front_table_bytes bytes); packed_q/k/v = otp_dense_cc_pack(W_p, NULL, b_p).  Supported when
 * otp_dense_cc_supported(C, T). */
size_t otp_qkv_front_table_bytes(int C);
int otp_qkv_front_pack_table(const void* dwq, const void* dwk, const void* dwv, const void* gq, const void* bq,
                             const void* gk, const void* bk, const void* gv, const void* bv, void* table, int C,
                             void* stream);
int otp_qkv_front(const void* x, const void* table, const void* packed_q, const void* packed_k, const void* packed_v,
                  void* q, void* k, void* v, int B, int C, int T, float eps, void* stream);

/* out = act(res + nearest_upsample_f(low)) on channel slices (HRNet fuse layers with f >= 4, model/HRNet.py:426-439,
 * 488-494; `res` may alias `out`); low (N, low_ctot, Hl, Wl), res / out (N, *_ctot, Hl*f, Wl*f); relu != 0 applies ReLU */
int otp_upsample_add(const void* low, const void* res, void* out, int N, int C, int Hl, int Wl, int f, int relu,
                     int low_ctot, int low_coff, int res_ctot, int res_coff, int out_ctot, int out_coff, void* stream);
/* a whole fuse row's upsampled terms in one pass: out = act(res + up_f0(low0) + up_f1(low1) (+ up_f2(low2))), summed in that
 * order (bit-identical to chaining otp_upsample_add); lows are dense (N, C, Hh / f_k, Wh / f_k), f_k powers of two >= 2,
 * Wh % 4 == 0; `lows` / `factors` are host arrays of nlow (1..3) entries */
int otp_upsample_add_multi(const void* const* lows, const int* factors, int nlow, const void* res, void* out, int N, int C,
                           int Hh, int Wh, int relu, int res_ctot, int res_coff, int out_ctot, int out_coff, void* stream);

/* ---- ConvTransformer backward pieces (training step; model/blocks.py:95-110,234-254,359-381,400-453) -------------
 * Channel attention backward is assembled on the host side from these (otpose_amd/train_ops.py): with O^T = the
 * transposed-contiguous image otp_chan_attn writes, dO = transpose(d_out), dP = dO v^T (scores + slab sum),
 * dS = softmax'(P, dP), dq = scale * dS k, dk = scale * dS^T q, dv = P^T dO (three otp_chan_attn_apply + transposes). */
/* Arithmetic of the q.k^T score products of otp_chan_attn / otp_chan_attn_scores: 1 (default) = split-half products on the
 * bf16 matrix cores (fp32 accumulation, see otp_conv2d_x3), 0 = the f32 MFMA kernel.  Process-wide switch. */
int otp_chan_attn_set_split(int on);
int otp_chan_attn_splits(int BH, int T);
int otp_chan_attn_scores(const void* a, const void* b, void* slabs, int BH, int hs, int T, void* stream);
int otp_chan_attn_apply(const void* v, const void* M, void* out, int BH, int hs, int T, void* stream);
/* The two products with bfloat16 operand pieces (csrc/transformer_grad.hip) for the attention backward, whose operands are
 * gradients (dS = dO v^T; dq, dk, dv): same arguments, same otp_chan_attn_set_split switch. */
int otp_chan_attn_scores_bf16p(const void* a, const void* b, void* slabs, int BH, int hs, int T, void* stream);
int otp_chan_attn_apply_bf16p(const void* v, const void* M, void* out, int BH, int hs, int T, void* stream);
/* per batch: in (R, Cc) row-major -> out (Cc, R) row-major times scale */
int otp_transpose_scale(const void* in, void* out, int batches, int R, int Cc, float scale, void* stream);
/* slabs (BH, NS, HSP, HSP) partial dP, P (BH, HSP, HSP) -> dS, dS^T, P^T (BH, HSP, HSP), HSP = hs rounded up to 16 */
int otp_softmax_backward(const void* slabs, const void* P, void* dS, void* dST, void* PT, int BH, int hs, int NS,
                         void* stream);
/* channel LayerNorm backward: grad_x, and dy_xhat = grad_y * xhat so that grad_gamma = otp_channel_sum(dy_xhat),
 * grad_beta = otp_channel_sum(grad_y) */
int otp_ln_channel_backward(const void* x, const void* grad_y, const void* gamma, void* grad_x, void* dy_xhat, int B, int C,
                            int T, float eps, void* stream);
/* the same with the parameter gradients from the same pass (C <= 136): grad_gamma = sum_{b,t} dy * xhat, grad_beta =
 * sum_{b,t} dy, reduced per workgroup in the kernel and folded by a second tiny launch (fixed order, fp64); no dy * xhat
 * tensor.  Workspace: otp_ln_channel_backward_workspace bytes (0 = shape not supported, use the form above). */
size_t otp_ln_channel_backward_workspace(int B, int C, int T);
int otp_ln_channel_backward_params(const void* x, const void* grad_y, const void* gamma, void* grad_x, void* grad_gamma,
                                   void* grad_beta, void* workspace, size_t workspace_bytes, int B, int C, int T, float eps,
                                   void* stream);
/* depthwise Conv1d(C, C, 3, stride, padding 1, groups C, bias False): w (C,3); grad_w is ACCUMULATED into */
int otp_dwconv3_forward(const void* x, const void* w, void* y, int B, int C, int T, int stride, void* stream);
int otp_dwconv3_backward(const void* x, const void* w, const void* grad_y, void* grad_x, void* grad_w, int B, int C, int T,
                         int stride, void* stream);
int otp_gelu_forward(const void* x, void* y, size_t n, void* stream);
int otp_gelu_backward(const void* x, const void* grad_y, void* grad_x, size_t n, void* stream);
/* MaxPool1d(3, 2, 1) on `rows` rows of length T; backward sends the gradient to the first maximum of each window */
int otp_maxpool3s2_forward(const void* x, void* y, int rows, int T, void* stream);
int otp_maxpool3s2_backward(const void* x, const void* grad_y, void* grad_x, int rows, int T, void* stream);
/* backward of otp_upsample_linear: grad_out is the channel slice [out_coff, out_coff+C) of a (B, out_ctot, T*f) tensor */
int otp_upsample_linear_backward(const void* grad_out, void* grad_x, int B, int C, int T, int f, int out_ctot, int out_coff,
                                 void* stream);

/* backward of otp_upsample_add on dense (planes, Hl*f, Wl*f) tensors: g = grad_out * (out_relu > 0) (out_relu NULL: no ReLU),
 * grad_res (optional) = g, grad_low (planes, Hl, Wl) = sum of g over each f x f cell */
int otp_upsample_add_backward(const void* grad_out, const void* out_relu, void* grad_res, void* grad_low, int planes, int Hl,
                              int Wl, int f, void* stream);

/* y = alpha*x + beta*y over n floats */
int otp_axpby(const void* x, void* y, float alpha, float beta, size_t n, void* stream);

/* ---- heat-map decode (the step right after the path: utils/heatmap.py:143-171 get_max_preds, :108-132 get_final_preds,
 * called per iteration at script/Common.py:147,371,424 after a full device-to-host copy) ------------------------------
 * heatmaps (N,J,H,W) -> preds (N,J,2) = (x, y) of the first maximum times (max > 0), maxvals (N,J).  refine != 0 adds
 * the reference's +-0.25 px shift towards the higher neighbour (strict 1 < px < W-1, 1 < py < H-1 test included).
 * center / scale ((N,2) each, both or neither): transform_preds for rot = 0, i.e. the similarity
 * center + (p - (W/2, H/2)) * 200 * scale_x / W  (utils/transform.py:76-105 feeds cv2.getAffineTransform three points
 * of exactly that map). */
int otp_heatmap_decode(const void* heatmaps, void* preds, void* maxvals, const void* center, const void* scale,
                       int N, int J, int H, int W, int refine, void* stream);

/* PCK accuracy of the training / validation loops (utils/evaluate.py:384-415 accuracy, :352-381 calc_dists / dist_acc;
 * called per iteration at script/Common.py:147-150) from the otp_heatmap_decode(refine = 0) coordinates (N,J,2) of the
 * predicted and target heat-maps of size H x W: acc (J+1) float32 = [mean over joints with a valid sample, per-joint
 * fraction of valid samples with distance < thr (-1: no valid sample)], cnt (1) int32 = joints with a valid sample. */
int otp_pck_accuracy(const void* pred_coords, const void* target_coords, void* acc, void* cnt, int N, int J, int H, int W,
                     float thr, void* stream);

/* ---- optimizer step of the training loop (script/Common.py:136-143: clip_grad_norm_ then optimizer.step(); AdamW built by
 * thirdparty/utils/train_utils.py:129-133) over flat, 16-byte aligned fp32 buffers ------------------------------------
 * otp_grad_sumsq: acc_f64[0] += sum(grad^2) (zero it first; call once per flat gradient segment).  acc_f64 addresses
 * 1 + otp_grad_sumsq_scratch() doubles: the words behind the accumulator hold the per-workgroup partial sums, which a
 * second launch adds in a fixed order (no float atomics: the clip coefficient is the same bits on every run).
 * otp_adamw_step: torch.optim.AdamW update of one hyper-parameter group, step >= 1 = the 1-based step count; the
 * gradient is first scaled by min(1, max_norm / (sqrt(*gradnorm_sq_f64) + 1e-6)) read on the device
 * (gradnorm_sq_f64 == NULL or max_norm <= 0: no clipping). */
size_t otp_grad_sumsq_scratch(void);
int otp_grad_sumsq(const void* grad, size_t n, void* acc_f64, void* stream);
int otp_adamw_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, size_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, const void* gradnorm_sq_f64, float max_norm,
                   void* stream);

/* ---- input assembly (the step before the path: dataset/PoseTrackDataset.py:397-406 transform per frame =
 * utils/transform.py:7-15 ToTensor + Normalize, script/Common.py:117 torch.cat over the frames) ---------------------
 * frames_u8 (B, F, H, W, 3) uint8 RGB, HWC like the cv2.warpAffine output -> out (B, 3F, H, W) float32,
 * out[b, 3f + c] = (u8 / 255 - mean_c) / std_c in float32, bit-identical to torchvision's two steps.  H*W % 4 == 0. */
int otp_frames_u8_to_clip(const void* frames_u8, void* out, int B, int F, int H, int W, float mean_r, float mean_g,
                          float mean_b, float std_r, float std_g, float std_b, void* stream);

/* ---- heatmap losses (model/loss.py) ------------------------------------------------------------- */
/* ST_OHKW_MSELoss.forward (loss.py:25-92): s,t,g (B,J,HW), w (B,J); flags (J) int32 in/out: when
 * flags_given == 0 the kernel computes flags[j] = (max_b,p g[b,j,p] == 1) itself; result[0..2] =
 * {ohkm_loss_s, mse_loss_s, final_loss}.  grad_s / grad_t (optional, may be NULL) receive
 * d final_loss / d s and d final_loss / d t. */
size_t otp_loss_workspace(int B, int J);
/* the same with d final_loss / d g as well (the second criterion call of script/Common.py:128-130 feeds a target that
 * depends on the model); grad_s / grad_t / grad_g each optional */
int otp_loss_st_ohkw_grads(const void* s, const void* t, const void* g, const void* w, void* flags, void* result,
                           void* grad_s, void* grad_t, void* grad_g, void* workspace, size_t workspace_bytes, int B, int J,
                           int HW, int topk, int flags_given, void* stream);
int otp_loss_st_ohkw(const void* s, const void* t, const void* g, const void* w, void* flags,
                     void* result, void* grad_s, void* grad_t, void* workspace, size_t workspace_bytes,
                     int B, int J, int HW, int topk, int flags_given, void* stream);
/* JointsMSE_OHKMMSELoss.forward (loss.py:115-148; ohkm != 0: result = {ohkm_loss, mse_loss / effective_num_joints,
 * final_loss}) and JointMSELoss.forward (loss.py:158-182; ohkm == 0: result = {0, loss, loss}).  o, g (B,J,HW);
 * w (B,J) or NULL for use_target_weight = False; effective_num_joints <= 0 means J (the reference's None);
 * grad_o (optional) receives d final_loss / d o (d loss / d o for the plain form).  Workspace: otp_loss_workspace. */
int otp_loss_joints_mse(const void* o, const void* g, const void* w, void* result, void* grad_o, void* workspace,
                        size_t workspace_bytes, int B, int J, int HW, int topk, int ohkm, int effective_num_joints,
                        void* stream);


/* ---- bf16 training path of the backbone (BASELINE configs[2]; reference step script/Common.py:118-144 over
 * model/HRNet.py:116-152).  Activations are NHWC bfloat16 with the channel stride rounded up to 8 (padding channels
 * zero); contractions run on the bf16 matrix cores with fp32 accumulation; statistics, parameter gradients and the
 * master weights are fp32.  `desc` describes the convolution as seen by the kernel (Cin -> Cout, true channel counts). */
typedef struct otp_nhwc_conv_desc {
    int N, H, W, Cin, Cout, kh, kw, stride, pad, dil;
    int out_mode;                /* 0: NHWC bf16 (+ per-tile channel statistics), 1: NCHW fp32 (+ bias), no statistics */
} otp_nhwc_conv_desc;
/* bytes of the packed bf16 weights / rows of the per-tile statistics buffer ([rows][2][CoutS] floats: sum, sum of squares
 * of the bf16-rounded outputs) / {MB, NB, CK, chunks, m-tiles, grid, LDS bytes, k-steps} of the launch (host only) */
size_t otp_nhwc_conv_weight_bytes(const otp_nhwc_conv_desc* desc);
int otp_nhwc_conv_stats_rows(const otp_nhwc_conv_desc* desc);
int otp_nhwc_conv_plan(const otp_nhwc_conv_desc* desc, int* out8);
/* fp32 (Cout, Cin, kh, kw) weights -> packed bf16.  dgrad != 0: `weight` is the (desc->Cin, desc->Cout, kh, kw) tensor of
 * the FORWARD conv and the packed operator is its input gradient (channels transposed, taps flipped; the caller sets
 * desc->pad = dil*(k-1) - pad and feeds a zero-inserted gradient for stride > 1, otp_nhwc_dilate). */
int otp_nhwc_conv_pack(const void* weight, void* wpacked, const otp_nhwc_conv_desc* desc, int dgrad, void* stream);
/* The same re-layouts batched: a training step repacks every fp32 master weight twice (forward operator + input-gradient
 * operator, ~670 launches of a few microseconds at cfg2).  otp_nhwc_conv_pack_job writes the otp_nhwc_conv_pack_job_bytes()
 * bytes of one job (pointers, strides, launch plan; opaque to the caller) to HOST memory; the caller keeps the jobs of a step
 * as one contiguous table in device memory and otp_nhwc_conv_pack_batch(table, n) performs all of them in one launch. */
size_t otp_nhwc_conv_pack_job_bytes(void);
int otp_nhwc_conv_pack_job(const void* weight, void* wpacked, const otp_nhwc_conv_desc* desc, int dgrad, void* job_host);
int otp_nhwc_conv_pack_batch(const void* jobs_device, int n_jobs, void* stream);
/* out = conv(x) (+ bias, fp32 (Cout), may be NULL); stats may be NULL */
int otp_nhwc_conv_bf16(const void* x, const void* wpacked, const void* bias, void* out, void* stats,
                       const otp_nhwc_conv_desc* desc, void* stream);
/* out = bf16(conv(x) + bias) + res: `res` (N, Ho, Wo, CoutS) bf16 is added to the ROUNDED conv result and the sum rounded
 * again - bit for bit what a separate bf16 add of the two tensors gives (out_mode 0 only, stats must be NULL).  The input-
 * gradient conv of a residual block's first layer takes the gradient of the skip connection here (a BasicBlock's dL/dx is
 * dgrad(conv1) + dL/dres: one pass over the tensor less than summing afterwards); res == NULL is otp_nhwc_conv_bf16. */
int otp_nhwc_conv_bf16_res(const void* x, const void* wpacked, const void* bias, const void* res, void* out, void* stats,
                           const otp_nhwc_conv_desc* desc, void* stream);
/* grad_weight (Cout, Cin, kh, kw) fp32 is OVERWRITTEN with sum over batch and pixels of grad_out x shifted input;
 * x (N, H, W, CinS) and grad_out (N, Ho, Wo, CoutS) bf16; 1x1 and 3x3 kernels */
size_t otp_nhwc_wgrad_workspace(const otp_nhwc_conv_desc* desc);
int otp_nhwc_wgrad_bf16(const void* x, const void* grad_out, void* grad_weight, void* workspace, size_t workspace_bytes,
                        const otp_nhwc_conv_desc* desc, void* stream);
/* BatchNorm2d with batch statistics (nn.BatchNorm2d in training mode, model/HRNet.py:500-571): the conv's per-tile sums
 * -> mean / rstd / scale = gamma*rstd / shift = beta - mean*scale (C-padded arrays of CS floats) and the running
 * statistics update; y = act(x*scale + shift (+ res)); backward in two passes (sums of g and g*xhat, then gx / gres).
 * relu_mask (may be NULL): one bit per element (one byte per 8-channel group, pixels * CS / 8 bytes) = y > 0, which is all
 * the backward needs of y; otp_nhwc_bn_backward takes either y (relu = 1) or that mask (relu = 2) in its `y` argument. */
int otp_nhwc_bn_finalize(const void* partials, int rows, int C, int CS, float count, const void* gamma, const void* beta,
                         void* mean, void* rstd, void* scale, void* shift, void* running_mean, void* running_var, float eps,
                         float momentum, void* stream);
int otp_nhwc_bn_apply(const void* x, const void* scale, const void* shift, const void* res, void* y, void* relu_mask,
                      size_t pixels, int CS, int relu, void* stream);
/* conv (no bias) + BatchNorm2d with batch statistics (+ residual) (+ ReLU) of one HRNet layer as ONE call - the three launches above
 * in sequence (host time: the training forward issues ~290 of these).  conv_out (N, Ho, Wo, CoutS) bf16 and stats
 * (otp_nhwc_conv_stats_rows x 2 x CoutS floats) receive the convolution and its per-tile sums, vec (4 x CoutS floats) mean | rstd |
 * scale | shift, y / relu_mask the layer's result as otp_nhwc_bn_apply writes them. */
int otp_nhwc_conv_bn_bf16(const void* x, const void* wpacked, const void* res, void* conv_out, void* stats, void* vec,
                          const void* gamma, const void* beta, void* running_mean, void* running_var, float eps, float momentum,
                          void* y, void* relu_mask, int relu, const otp_nhwc_conv_desc* desc, void* stream);
size_t otp_nhwc_bn_backward_workspace(size_t pixels, int CS);
int otp_nhwc_bn_backward(const void* gy, const void* y, const void* x, const void* mean, const void* rstd, const void* gamma,
                         void* gx, void* gres, void* dgamma, void* dbeta, void* workspace, size_t workspace_bytes,
                         size_t pixels, int C, int CS, int relu, void* stream);
/* out = act(res + nearest_upsample_f(low)) on NHWC bf16 (HRNet fuse rows, model/HRNet.py:426-439,488-494) and its backward */
int otp_nhwc_upsample_add(const void* low, const void* res, void* out, int N, int H, int W, int CS, int f, int relu,
                          void* stream);
int otp_nhwc_upsample_add_backward(const void* gy, const void* out, void* gres, void* glow, int N, int Hl, int Wl, int CS,
                                   int f, int relu, void* stream);
/* layout / precision hand-over between the fp32 NCHW tensors of the module boundary and the NHWC bf16 interior;
 * frame_split = B > 0 reads the (B, 5*C, H, W) clip as (5B, C, H, W) like otp_conv2d does (model/OTPose.py:317) */
int otp_nchw_f32_to_nhwc_bf16(const void* in, void* out, int N, int C, int H, int W, int frame_split, void* stream);
int otp_nhwc_bf16_to_nchw_f32(const void* in, void* out, int N, int C, int H, int W, void* stream);
int otp_nhwc_dilate(const void* in, void* out, int N, int Hi, int Wi, int s, int H, int W, int CS, void* stream);


/* ---- TransformerBlock residual update in training (model/blocks.py:277-279 with AffineDropPath, :283-316):
 * out = x + mask[b] * scale[c] * a on (B, C, T) fp32 tensors; `mask` (B floats: Bernoulli(keep) / keep) may be NULL.
 * Backward: grad_a = mask[b]*scale[c]*grad_out, grad_scale[c] = sum_{b,t} mask[b]*a*grad_out (grad_x is grad_out). */
int otp_scale_residual(const void* x, const void* a, const void* scale, const void* mask, void* out, int B, int C, int T,
                       void* stream);
size_t otp_scale_residual_backward_workspace(int B, int C, int T);
int otp_scale_residual_backward(const void* grad_out, const void* a, const void* scale, const void* mask, void* grad_a,
                                void* grad_scale, void* workspace, size_t workspace_bytes, int B, int C, int T, void* stream);


/* bf16 interior of the TransformerBlock MLP (model/blocks.py:248-254) on (B, 1, T, CS) NHWC views of the (B, C, T)
 * sequences: bias gradients (per-channel sums over the pixels) and the exact-erf GELU with fp32 arithmetic. */
size_t otp_nhwc_channel_sum_workspace(size_t pixels, int CS);
int otp_nhwc_channel_sum(const void* g, void* out, void* workspace, size_t workspace_bytes, size_t pixels, int C, int CS,
                         void* stream);
int otp_gelu_bf16_forward(const void* x, void* y, size_t n, void* stream);
int otp_gelu_bf16_backward(const void* x, const void* grad_y, void* grad_x, size_t n, void* stream);
/* nn.GELU followed by nn.Dropout(p) of the same MLP (model/blocks.py:250-251; training mode) as one pass over n bf16 elements
 * (n % 8 == 0): y = keep ? gelu(x) / (1 - p') : 0 with p' = round(65536 p) / 65536, rounded once; keep_bits (n / 8 bytes) receives
 * the decisions, bit j of byte u = element 8 u + j.  The draws are a counter-based hash of (element index, seed): the same seed
 * repeats them.  The backward takes the bits: grad_x = keep ? grad_y * gelu'(x) / (1 - p') : 0. */
/* The same MLP interior with those passes folded into the projections' launches (1x1 convolutions of (N, 1, T, C) sequences):
 * otp_nhwc_mlp_up_bf16: pre = bf16(conv(x) + bias) and act = dropout(gelu(pre), p) + keep_bits in one launch (gelu through a 6e-5 fit of
 * the normal CDF - the value is rounded to bfloat16); otp_nhwc_mlp_down_dgrad_bf16: grad_pre = keep ? dgrad(gy) * gelu'(pre) / (1 - p') : 0
 * in one launch, `desc` / `wpacked` being the input-gradient convolution of the down-projection (otp_nhwc_conv_pack(.., dgrad = 1)).
 * otp_nhwc_mlp_fused_supported: the shape runs on the kernel that has these epilogues (else the caller keeps the separate launches). */
int otp_nhwc_mlp_fused_supported(const otp_nhwc_conv_desc* desc);
int otp_nhwc_mlp_up_bf16(const void* x, const void* wpacked, const void* bias, void* pre, void* act, void* keep_bits, float p,
                         unsigned long long seed, const otp_nhwc_conv_desc* desc, void* stream);
int otp_nhwc_mlp_down_dgrad_bf16(const void* gy, const void* wpacked, const void* pre, const void* keep_bits, void* grad_pre, float p,
                                 const otp_nhwc_conv_desc* desc, void* stream);
int otp_gelu_dropout_bf16_forward(const void* x, void* y, void* keep_bits, size_t n, float p, unsigned long long seed, void* stream);
int otp_gelu_dropout_bf16_backward(const void* x, const void* grad_y, const void* keep_bits, void* grad_x, size_t n, float p,
                                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OTPOSE_HIP_H */
