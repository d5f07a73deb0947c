// csrc/hb.hip (= csrc/h16.hip compiled for bfloat16, NHWC addressing): the training step's 3x3 convolutions on the window /
// weight-stream kernel of the fp16 engine.  Internal interface between csrc/nhwc.hip (which owns the C ABI: otp_nhwc_conv_*)
// and that translation unit, plus the layout of the packed weights both sides must agree on.
#pragma once
#include "common.h"

constexpr int OTP_HB_KS = 5;              // k-steps per 16-channel chunk: 18 (tap, channel group) slots of 8 channels in 5 x 4 (2 empty)
// packed weights of a (cout block, 16-channel chunk): 4 full k-steps x NTW tiles x 1 KB, then the half-filled fifth (k-slots 16, 17
// on lanes 0 .. 31: 512 bytes per tile)
__host__ __device__ constexpr int otp_hb_wb(int ntw) { return ntw * 4608; }
// Output-channel row of an MFMA tile <-> channel: cout tiles go in pairs (2 tp, 2 tp + 1) whose rows are permuted so that lane
// (pixel, kl) ends up with 8 CONSECUTIVE channels 32 tp + 8 kl .. + 7 of its pixel = one 16-byte record; a tile without a partner
// keeps the identity (4 consecutive channels per lane = half a record).
__host__ __device__ inline bool otp_hb_paired(int co_blk, int t, int ntw, int Cout) {
    const int tb = t | 1;
    return tb < ntw && co_blk + 16 * tb < Cout;
}
__host__ __device__ inline int otp_hb_row2ch(int co_blk, int t, int row, int ntw, int Cout) {
    return otp_hb_paired(co_blk, t, ntw, Cout) ? co_blk + 32 * (t >> 1) + 8 * (row >> 2) + 4 * (t & 1) + (row & 3) : co_blk + 16 * t + row;
}
inline int otp_hb_ntw(int Cout) {
    const int c16 = (Cout + 15) / 16;
    return (c16 % 3 == 0) ? 3 : (c16 % 2 == 0 || c16 <= 2 ? 2 : 3);
}
// 16-byte unit u of a (cout block, chunk) image -> (k-step, tile, lane); lane = (row i16 = lane & 15, k-group kl = lane >> 4),
// k-slot q = 4 s + kl: tap q / 2, input channels 16 chunk + 8 (q % 2) .. + 7
__host__ __device__ inline void otp_hb_unit(int u, int ntw, int* s, int* t, int* lane) {
    const int full = (OTP_HB_KS - 1) * ntw * 64;
    if (u < full) {
        *s = u / (ntw * 64), *t = (u >> 6) % ntw, *lane = u & 63;
    } else {
        u -= full;
        *s = OTP_HB_KS - 1, *t = u >> 5, *lane = u & 31;
    }
}

// ---- 1x1 convolutions (csrc/h16.hip: h16_pointwise_kernel) -------------------------------------------------------------------------
// KS k-steps of 32 input channels (Cin padded with zero weights and masked loads); the packed operator is ceil(Cout / 32) blocks of
// [tile of the pair (2)][k-step][lane] 16-byte A fragments - row r16 of tile m of block p = output channel 32 p + 8 (r16 >> 2) + 4 m +
// (r16 & 3), lane = (r16 = lane & 15, kq = lane >> 4): input channels 32 ks + 8 kq .. + 7 - each block padded to whole 4 KB
__host__ __device__ constexpr int otp_hbpw_blkb(int KS) { return ((2 * KS * 1024) + 4095) / 4096 * 4096; }
inline int otp_hbpw_ks(int Cin) {               // the instantiated k-step counts of the bf16 build; 0: not covered
    const int k = (Cin + 31) / 32;
    return k <= 2 ? 2 : (k <= 4 ? 4 : (k == 5 ? 5 : (k <= 8 ? 8 : (k <= 12 ? 12 : (k <= 17 ? 17 : 0)))));
}
// Dropout draws of the bf16 training path (csrc/nhwc.hip: gelu_dropout_bf16_*, csrc/hb.hip: the MLP up-projection's epilogue): a
// counter-based hash of (element pair, seed), two 16-bit uniforms per 32-bit hash (lowbias32 finaliser), keep <=> u16 >= round(65536 p).
// Unit u = elements 8 u .. 8 u + 7 of the tensor; bit j of the result = element 8 u + j is kept.
__device__ __forceinline__ uint32_t otp_drop_hash(size_t pair, uint32_t s0, uint32_t s1) {
    uint32_t h = (uint32_t)pair * 0x9E3779B1u + s0;
    h ^= ((uint32_t)(pair >> 32) + s1) * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x21F0AAADu; h ^= h >> 15; h *= 0x735A2D97u; h ^= h >> 15;
    return h;
}
__device__ __forceinline__ unsigned otp_drop_keep8(size_t u, uint32_t s0, uint32_t s1, uint32_t thr) {
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t h = otp_drop_hash(u * 4 + k, s0, s1);
        bits |= ((h & 0xFFFFu) >= thr ? 1u : 0u) << (2 * k);
        bits |= ((h >> 16) >= thr ? 1u : 0u) << (2 * k + 1);
    }
    return bits;
}
// Phi(x) to 6e-5 (csrc/mlpx.hip: mx_gelu_h's fit, 11 plain vector instructions): 0.5 + xc R(xc^2), xc = clamp(x, +-4.2) - for values that
// are rounded to bfloat16 (2^-9) afterwards
__device__ __forceinline__ float otp_phi_fast(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.2f, 4.2f), s2 = xc * xc;
    float r = 4.711542158e-11f;
    r = r * s2 + -4.627826316e-09f;
    r = r * s2 + 2.017249231e-07f;
    r = r * s2 + -5.191738396e-06f;
    r = r * s2 + 8.882890530e-05f;
    r = r * s2 + -1.080400373e-03f;
    r = r * s2 + 9.720675326e-03f;
    r = r * s2 + -6.618899545e-02f;
    r = r * s2 + 3.988157481e-01f;
    return xc * r + 0.5f;
}

// epilogue of the pointwise kernel for the TransformerBlock MLP (model/blocks.py:248-254): mode 1 = the up-projection, which stores its
// result (`out`, the pre-activation the backward needs) AND dropout(gelu(result)) (`out2`, + the keep bits); mode 2 = the down-projection's
// input gradient, multiplied by gelu'(pre-activation `aux`) and the kept / (1 - p) factor before it is stored
struct otp_hbpw_epi {
    int mode;
    void* out2;
    unsigned char* keep;
    const void* aux;
    uint32_t s0, s1, thr;
    float scale;
};
bool otp_hbpw_supported(const otp_nhwc_conv_desc* d);              /* 1x1 / stride 1 / pad 0, Cin % 8 == 0, Cout % 8 == 0, NHWC bf16 or NCHW fp32 result */
int otp_hbpw_stats_rows(const otp_nhwc_conv_desc* d);              /* rows of the statistics buffer [rows][2][Cout] = 128-pixel tiles */
int otp_hbpw_conv(const void* x, const void* wpacked, const void* bias, const void* res, void* out, void* stats,
                  const otp_nhwc_conv_desc* d, hipStream_t stream, const otp_hbpw_epi* epi = nullptr);

// implemented in csrc/hb.hip
bool otp_hb_supported(const otp_nhwc_conv_desc* d);                /* 3x3 / pad 1 / dilation 1 / stride 1 or 2, NHWC bf16 result, Cin % 16 == 0, Cout % 8 == 0 */
bool otp_hb_pays(const otp_nhwc_conv_desc* d);                     /* the shapes on which it beats nhwc_conv_kernel (measured) */
int otp_hb_stats_rows(const otp_nhwc_conv_desc* d);                /* rows of the statistics buffer [rows][2][Cout] = pixel tiles of the launch */
int otp_hb_conv(const void* x, const void* wpacked, const void* bias, const void* res, void* out, void* stats,
                const otp_nhwc_conv_desc* d, hipStream_t stream);
