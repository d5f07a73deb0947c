// The TransformerBlock MLP of the temporal encoders (reference model/blocks.py:248-254, applied at :277-279) as ONE launch
// on the bf16 matrix cores with split ("bf16x3") products - the same operator as csrc/mlp.hip:
//     out = res + scale * (W2 . gelu(W1 . LN?(x) + b1)) + shift        x, res, out (B, C, T); W1 (HID, C); W2 (C, HID)
// fp32 storage, fp32 accumulation, fp32 LayerNorm / bias / GELU; every product a*b of the two GEMMs is computed as
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on v_mfma_f32_16x16x32_bf16 with a = a_hi + a_lo (+ <= 2^-18 |a|), see csrc/convx.hip.
// Three bf16 MFMAs replace eight f32 ones (16x16x32 against 16x16x4 at half the cycles), 4.9x fewer matrix-pipe cycles.
//
// Register-resident like csrc/mlp.hip: a wave owns NT x 16 tokens and keeps their C input channels as split B-operand
// fragments (k-step = 32 channels: lane (token n, kq) holds channels 32 ks + 8 kq .. + 7).  The hidden dimension is walked 32
// channels at a time: phase 1 produces two 16 x 16 hidden tiles (accumulator register i of lane (n, kq) = hidden channel
// 4 kq + i of the tile), bias + erf-GELU run on the accumulators, and the 8 values a lane holds ARE a B-operand k-slot of
// phase 2 once W2's fragments are packed to contract slot (kq, j) with hidden channel 4 kq + j (j < 4, first tile) or
// 16 + 4 kq + j - 4 (second tile) - so the 4C-wide hidden activation never leaves the registers and never crosses lanes.
// Weights: one 40 KB image per 32 hidden channels (W1 / W2 fragments hi + lo, b1), copied global -> LDS by the LDS-DMA
// (global_load_lds, no staging registers), double buffered, one barrier per block.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef otp_x3x8 h16x8;              // 8 operand pieces of the split products (common.h: IEEE half since round 4)
typedef otp_x3x2 h16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// erf-GELU on a pair of values, branch-free, packed f32 (the rational fit of csrc/mlp.hip: 7e-7 max abs error)
__device__ __forceinline__ f32x2 mx_gelu2(f32x2 x) {
    f32x2 t = x * 0.70710678118654752440f;
    t.x = __builtin_amdgcn_fmed3f(t.x, -4.f, 4.f);
    t.y = __builtin_amdgcn_fmed3f(t.y, -4.f, 4.f);
    const f32x2 t2 = t * t;
    f32x2 p = t2 * -2.72614225801306e-10f + 2.77068142495902e-08f;
    p = p * t2 + -2.10102402082508e-06f;
    p = p * t2 + -5.69250639462346e-05f;
    p = p * t2 + -7.34990630326855e-04f;
    p = p * t2 + -2.95459980854025e-03f;
    p = p * t2 + -1.60960333262415e-02f;
    p = p * t;
    f32x2 q = t2 * -1.45660718464996e-05f + -2.13374055278905e-04f;
    q = q * t2 + -1.68282697438203e-03f;
    q = q * t2 + -7.37332916720468e-03f;
    q = q * t2 + -1.42647390514189e-02f;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(q.x);
    r.y = __builtin_amdgcn_rcpf(q.y);
    const f32x2 e = p * r, hx = x * 0.5f;
    return hx * e + hx;
}

// erf-GELU for the HALF-operand form (H1 below): the result is rounded to half (2^-11 relative) on its way into the second GEMM,
// so 6e-5 absolute is enough: x * (0.5 + xc R(xc^2)), xc = clamp(x, +-4.2), R = a degree-8 fit of erf(sqrt(s / 2)) / (2 sqrt(s)) -
// 12 plain vector instructions, no reciprocal (the fp32 form above: 18 + v_rcp_f32)
__device__ __forceinline__ float mx_gelu_h(float x) {
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
    return x * (xc * r + 0.5f);
}

__device__ __forceinline__ float mx_kslot_sum(float v) {   // sum over the four k-slot lane groups (lanes n, n+16, n+32, n+48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// 8 floats -> bf16 hi / lo vectors (hi = rne(a), lo = rne(a - hi))
__device__ __forceinline__ void mx_split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 a = {v[2 * i], v[2 * i + 1]};
        const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, h16x2));
        const f32x2 af = otp_x3_widen(hb);
        h[i] = hb;
        l[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a - af, h16x2));
    }
    hi = __builtin_bit_cast(h16x8, (u32x4){h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(h16x8, (u32x4){l[0], l[1], l[2], l[3]});
}
// 8 floats -> halves, rounded once (the fp16 engine's operands)
__device__ __forceinline__ h16x8 mx_half8(const float (&v)[8]) {
    uint32_t h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v[2 * i], v[2 * i + 1]}, h16x2));
    return __builtin_bit_cast(h16x8, (u32x4){h[0], h[1], h[2], h[3]});
}

constexpr int mx_ks1(int C) { return (C + 31) / 32; }
constexpr int mx_mt(int C) { return (C + 15) / 16; }
// bytes of one 32-hidden-channel block: W1 fragments [2 tiles][ks][hi, lo][1 KB], W2 fragments [mt][hi, lo][1 KB], b1[32],
// rounded up to whole KB (one LDS-DMA instruction of a wave); the last copy pass of a workgroup may be partial
// (PS = pieces per fragment: 2 = hi | lo, the split products; 1 = hi only, the half-operand form H1 - half the image, so that two
//  workgroups of the C = 204 instantiation share a CU's LDS)
constexpr int mx_w1_bytes(int C, int PS = 2) { return 2 * mx_ks1(C) * PS * 1024; }
constexpr int mx_w2_bytes(int C, int PS = 2) { return mx_mt(C) * PS * 1024; }
constexpr int mx_block_bytes(int C, int PS = 2) { return (mx_w1_bytes(C, PS) + mx_w2_bytes(C, PS) + 128 + 1023) / 1024 * 1024; }

__global__ void mlpx_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                 unsigned char* __restrict__ packed, int C, int HID, int PS) {
    const int KS1 = mx_ks1(C), W1B = mx_w1_bytes(C, PS), W2B = mx_w2_bytes(C, PS), BLKB = mx_block_bytes(C, PS);
    const int units = BLKB / 16;                                   // 16-byte units per block
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((HID + 31) / 32) * units) return;                  // (a ragged last block - HID = 816 - is padded with zero weights)
    const int hb = idx / units, u = idx - hb * units;
    u32x4 o = {0u, 0u, 0u, 0u};
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (u < (W1B + W2B) / 16) {
        const int frag = u >> 6, lane = u & 63, row = lane & 15, kq = lane >> 4;
        const bool lo_part = PS == 2 && (frag & 1);
        if (u < W1B / 16) {
            const int f2 = frag / PS, tile = f2 / KS1, ks = f2 - tile * KS1;
            const int hid = 32 * hb + 16 * tile + row;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 32 * ks + 8 * kq + j;
                if (c < C && hid < HID) v[j] = w1[(size_t)hid * C + c];
            }
        } else {
            const int mt = (frag - W1B / 1024) / PS, c = 16 * mt + row;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int hid = 32 * hb + (j < 4 ? 4 * kq + j : 16 + 4 * kq + j - 4);
                if (c < C && hid < HID) v[j] = w2[(size_t)c * HID + hid];
            }
        }
        h16x8 hi, lo;
        mx_split8(v, hi, lo);
        o = __builtin_bit_cast(u32x4, lo_part ? lo : hi);
    } else if (u < (W1B + W2B) / 16 + 8) {
        const int k = (u - (W1B + W2B) / 16) * 4;
        const bool in = 32 * hb + k < HID;                         // (HID % 4 == 0)
        o = in ? (u32x4){__builtin_bit_cast(uint32_t, b1[32 * hb + k]), __builtin_bit_cast(uint32_t, b1[32 * hb + k + 1]),
                         __builtin_bit_cast(uint32_t, b1[32 * hb + k + 2]), __builtin_bit_cast(uint32_t, b1[32 * hb + k + 3])}
               : (u32x4){0u, 0u, 0u, 0u};
    }
    reinterpret_cast<u32x4*>(packed)[idx] = o;
}

// copy one weight block global -> LDS with the LDS-DMA: unit u (16 bytes) of the block lands at lds + 16 u
template <int NTHR, int BLKB>
__device__ __forceinline__ void mx_stage(const unsigned char* __restrict__ src, unsigned char* lds) {
    constexpr int NST = (BLKB / 16 + NTHR - 1) / NTHR;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int u0 = i * NTHR + wave * 64;                       // wave-uniform first unit of this wave-instruction
        if (u0 * 16 >= BLKB) break;                                // (whole KB: a wave-instruction is inside the block or past it)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(u0 + lane) * 16),
                                         (__attribute__((address_space(3))) void*)(lds + u0 * 16), 16, 0, 0);
    }
}

// One pass of a wave over NT (1 or 2) column tiles of 16 tokens starting at token tok0, all HID / 32 hidden blocks; every wave
// of the workgroup calls it together (the weight blocks go through LDS behind one barrier per block).
// NT == 2: column n of tile j is token tok0 + 2n + j (8-byte accesses); NT == 1: token tok0 + n.
// H1: the fp16 engine's arithmetic (cfg.MODEL.DTYPE = "fp16") - every operand a half, rounded ONCE (the hi pieces of the same packed
// weight image; activations and the hidden layer rounded to half), one MFMA per product, fp32 accumulation, LayerNorm / bias in fp32
template <int C, int HID, int NTHR, bool LN, int NT, bool H1 = false>
__device__ __forceinline__ void mlpx_pass(const float* __restrict__ x, const unsigned char* __restrict__ packed,
                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                          const float* __restrict__ res, float* __restrict__ out, int T, size_t base, int tok0,
                                          unsigned char* lds, const float* __restrict__ ln_gamma,
                                          const float* __restrict__ ln_beta, float ln_eps, unsigned* rflag) {
    constexpr int KS1 = mx_ks1(C), MT = mx_mt(C), HB = (HID + 31) / 32, PS = H1 ? 1 : 2;
    constexpr int W1B = mx_w1_bytes(C, PS), W2B = mx_w2_bytes(C, PS), BLKB = mx_block_bytes(C, PS);
    const int lane = threadIdx.x & 63, kq = lane >> 4, n = lane & 15;
    const int tok = tok0 + NT * n;
    const bool valid = tok < T;                       // NT == 2: T is even, a token pair is inside or outside together

    // weights of hidden block 0 -> LDS buffer 0 (every wave is past the previous pass: the barrier that ended its loop)
    mx_stage<NTHR, BLKB>(packed, lds);

    // the wave's input columns: X[ks][j][t] = x[32 ks + 8 kq + j][token of column n of tile t] (0 past C)
    float X[KS1][8][NT];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 32 * ks + 8 * kq + j;
            const bool live = 32 * ks + 24 + j < C || c < C;       // compile-time true except in the last k-step
            const float* src = x + base + (size_t)(live ? c : 0) * T + (valid ? tok : T - NT);
            if (NT == 2) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(src);
                X[ks][j][0] = live ? v.x : 0.f;
                X[ks][j][NT - 1] = live ? v.y : 0.f;
            } else {
                X[ks][j][0] = live ? *src : 0.f;
            }
        }
    if (LN) {
        constexpr float inv_c = 1.f / (float)C;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float s0 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) s0 += X[ks][j][t];
            const float m0 = mx_kslot_sum(s0) * inv_c;
            float v0 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool live = 32 * ks + 24 + j < C || 32 * ks + 8 * kq + j < C;
                    X[ks][j][t] = live ? X[ks][j][t] - m0 : 0.f;
                    v0 += X[ks][j][t] * X[ks][j][t];
                }
            const float r0 = 1.f / sqrtf(mx_kslot_sum(v0) * inv_c + ln_eps);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 32 * ks + 8 * kq + j;
                    const bool live = 32 * ks + 24 + j < C || c < C;
                    X[ks][j][t] = live ? X[ks][j][t] * r0 * ln_gamma[live ? c : 0] + ln_beta[live ? c : 0] : 0.f;
                }
        }
    }
    h16x8 Xh[KS1][NT], Xl[H1 ? 1 : KS1][NT];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = X[ks][j][t];
            if constexpr (H1) Xh[ks][t] = mx_half8(v);
            else mx_split8(v, Xh[ks][t], Xl[H1 ? 0 : ks][t]);
        }
    f32x4 Y[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) Y[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                  // (drains the LDS-DMA of block 0)

    for (int hb = 0; hb < HB; ++hb) {
        // next block's weights on their way (LDS-DMA into the other buffer) while this one is consumed
        if (hb + 1 < HB) mx_stage<NTHR, BLKB>(packed + (size_t)(hb + 1) * BLKB, lds + ((hb + 1) & 1) * BLKB);
        const unsigned char* P1 = lds + (hb & 1) * BLKB + lane * 16;
        const unsigned char* P2 = P1 + W1B;
        const float* PB = reinterpret_cast<const float*>(lds + (hb & 1) * BLKB + W1B + W2B);
        // phase 1: two hidden tiles (16 channels x 16 NT tokens) = W1[32 hb + 16 tile ..][:] . X, accumulators start from b1
        f32x4 H[2][NT];
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
#pragma unroll
            for (int t = 0; t < NT; ++t) H[tile][t] = *reinterpret_cast<const f32x4*>(PB + 16 * tile + 4 * kq);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const h16x8 ah = *reinterpret_cast<const h16x8*>(P1 + ((tile * KS1 + ks) * PS) * 1024);
                if constexpr (H1) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) H[tile][t] = OTP_X3_MFMA(ah, Xh[ks][t], H[tile][t], 0, 0, 0);
                } else {
                    const h16x8 al = *reinterpret_cast<const h16x8*>(P1 + ((tile * KS1 + ks) * 2 + 1) * 1024);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        H[tile][t] = OTP_X3_MFMA(al, Xh[ks][t], H[tile][t], 0, 0, 0);
                        H[tile][t] = OTP_X3_MFMA(ah, Xl[H1 ? 0 : ks][t], H[tile][t], 0, 0, 0);
                        H[tile][t] = OTP_X3_MFMA(ah, Xh[ks][t], H[tile][t], 0, 0, 0);
                    }
                }
            }
        }
        h16x8 Gh[NT], Gl[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (H1) {
                const float v[8] = {mx_gelu_h(H[0][t][0]), mx_gelu_h(H[0][t][1]), mx_gelu_h(H[0][t][2]), mx_gelu_h(H[0][t][3]),
                                    mx_gelu_h(H[1][t][0]), mx_gelu_h(H[1][t][1]), mx_gelu_h(H[1][t][2]), mx_gelu_h(H[1][t][3])};
                Gh[t] = mx_half8(v);
            } else {
                const f32x2 g0 = mx_gelu2(f32x2{H[0][t][0], H[0][t][1]}), g1 = mx_gelu2(f32x2{H[0][t][2], H[0][t][3]});
                const f32x2 g2 = mx_gelu2(f32x2{H[1][t][0], H[1][t][1]}), g3 = mx_gelu2(f32x2{H[1][t][2], H[1][t][3]});
                const float v[8] = {g0.x, g0.y, g1.x, g1.y, g2.x, g2.y, g3.x, g3.y};
                mx_split8(v, Gh[t], Gl[t]);
            }
        }
        // phase 2: Y += W2[:, 32 hb ..] . hidden tiles (k-slot (kq, j) = hidden channel 4 kq + j / 16 + 4 kq + j - 4)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const h16x8 ah = *reinterpret_cast<const h16x8*>(P2 + (mt * PS) * 1024);
            if constexpr (H1) {
#pragma unroll
                for (int t = 0; t < NT; ++t) Y[mt][t] = OTP_X3_MFMA(ah, Gh[t], Y[mt][t], 0, 0, 0);
            } else {
                const h16x8 al = *reinterpret_cast<const h16x8*>(P2 + (mt * 2 + 1) * 1024);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    Y[mt][t] = OTP_X3_MFMA(al, Gh[t], Y[mt][t], 0, 0, 0);
                    Y[mt][t] = OTP_X3_MFMA(ah, Gl[t], Y[mt][t], 0, 0, 0);
                    Y[mt][t] = OTP_X3_MFMA(ah, Gh[t], Y[mt][t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                              // every wave is done with this block; the next one has landed
    }

    {   // range guard (common.h): an input or hidden value beyond a half's range has made these sums NaN
        bool bad = false;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) bad |= otp_out_of_range(Y[mt][t][i]);
        otp_range_report(rflag, bad, OTP_RANGE_MLPX);
    }
    if (!valid) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 16 * mt + 4 * kq + i;
            if (c < C) {
                const size_t o = base + (size_t)c * T + tok;
                const float sc = scale[c], sh = shift[c];
                if (NT == 2) {
                    const f32x2 r = *reinterpret_cast<const f32x2*>(res + o);
                    f32x2 v;
                    v.x = r.x + Y[mt][0][i] * sc + sh;
                    v.y = r.y + Y[mt][NT - 1][i] * sc + sh;
                    *reinterpret_cast<f32x2*>(out + o) = v;
                } else {
                    out[o] = res[o] + Y[mt][0][i] * sc + sh;
                }
            }
        }
    }
}

// NT token tiles of 16 per wave: 2 at C = 136; 1 at C = 204 (7 + 7 input fragments and 13 accumulator tiles per token tile)
template <int C, int HID, int WAVES, bool LN, int NT, bool H1 = false>
__global__ __launch_bounds__(WAVES * 64, (NT == 1 && (C <= 136 || H1)) ? 4 : 2) void mlpx_kernel(
    const float* __restrict__ x, const unsigned char* __restrict__ packed, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ out, int T, int tiles_per_b,
    const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, float ln_eps, unsigned* rflag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mx_lds[];
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    mlpx_pass<C, HID, WAVES * 64, LN, NT, H1>(x, packed, scale, shift, res, out, T, (size_t)b * C * T,
                                          tile * (WAVES * 16 * NT) + wave * 16 * NT, mx_lds, ln_gamma, ln_beta, ln_eps, rflag);
}

// Balanced form for token counts that are multiples of 27 column tiles per workgroup (T = 6912 = 16 x 27 x 16), as
// mlp_fused_balanced_kernel: one 8-wave workgroup owns 27 tiles and walks them in two passes (7 tiles per SIMD, not 8).
template <int C, int HID, bool LN>
__global__ __launch_bounds__(512, 2) void mlpx_balanced_kernel(
    const float* __restrict__ x, const unsigned char* __restrict__ packed, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ out, int T, int wgs_per_b,
    const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, float ln_eps, unsigned* rflag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mx_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / wgs_per_b, part = blockIdx.x - b * wgs_per_b;
    const size_t base = (size_t)b * C * T;
    const int t0 = part * (27 * 16);
    mlpx_pass<C, HID, 512, LN, 2>(x, packed, scale, shift, res, out, T, base, t0 + wave * 32, mx_lds, ln_gamma, ln_beta, ln_eps, rflag);
    if (wave < 3) {
        mlpx_pass<C, HID, 512, LN, 2>(x, packed, scale, shift, res, out, T, base, t0 + (16 + 2 * wave) * 16, mx_lds, ln_gamma,
                                      ln_beta, ln_eps, rflag);
    } else {
        // waves 4, 5, 6 -> tiles 22, 23, 24; wave 3 -> 25; wave 7 -> 26
        const int tile = wave == 3 ? 25 : (wave == 7 ? 26 : 18 + wave);
        mlpx_pass<C, HID, 512, LN, 1>(x, packed, scale, shift, res, out, T, base, t0 + tile * 16, mx_lds, ln_gamma, ln_beta,
                                      ln_eps, rflag);
    }
}

}  // namespace

extern "C" int otp_mlp_x3_supported(int C, int HID, int T) {
    return (((C == 136 && HID == 544) || (C == 204 && HID == 816)) && T > 0 && T % 2 == 0) ? 1 : 0;
}

extern "C" size_t otp_mlp_x3_weight_bytes(int C, int HID) {
    if (C <= 0 || C % 4 || HID <= 0 || HID % 16) return 0;   // (partial k-slots / row tiles are masked: C = 204 = 6 x 32 + 12)
    return (size_t)((HID + 31) / 32) * mx_block_bytes(C);
}

/* the image of otp_ln_mlp_h1: the hi pieces only (weights rounded to half once), half the bytes */
extern "C" size_t otp_mlp_h1_weight_bytes(int C, int HID) {
    if (C <= 0 || C % 4 || HID <= 0 || HID % 16) return 0;
    return (size_t)((HID + 31) / 32) * mx_block_bytes(C, 1);
}

extern "C" int otp_mlp_x3_pack(const void* w1, const void* b1, const void* w2, void* packed, int C, int HID, void* stream) {
    if (!w1 || !b1 || !w2 || !packed) return OTP_ERR_BAD_ARG;
    const size_t bytes = otp_mlp_x3_weight_bytes(C, HID);
    if (!bytes) return OTP_ERR_UNSUPPORTED;
    const int total = (int)(bytes / 16);
    hipLaunchKernelGGL(mlpx_pack_kernel, dim3(otp_ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w1), static_cast<const float*>(b1), static_cast<const float*>(w2),
                       static_cast<unsigned char*>(packed), C, HID, 2);
    return otp_launch_status();
}

extern "C" int otp_mlp_h1_pack(const void* w1, const void* b1, const void* w2, void* packed, int C, int HID, void* stream) {
    if (!w1 || !b1 || !w2 || !packed) return OTP_ERR_BAD_ARG;
    const size_t bytes = otp_mlp_h1_weight_bytes(C, HID);
    if (!bytes) return OTP_ERR_UNSUPPORTED;
    const int total = (int)(bytes / 16);
    hipLaunchKernelGGL(mlpx_pack_kernel, dim3(otp_ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w1), static_cast<const float*>(b1), static_cast<const float*>(w2),
                       static_cast<unsigned char*>(packed), C, HID, 1);
    return otp_launch_status();
}

namespace {
int mlpx_launch(const void* x, const void* packed, const void* scale, const void* shift, const void* res, void* out,
                const void* ln_gamma, const void* ln_beta, float ln_eps, int B, int C, int HID, int T, void* stream, bool h1 = false) {
    if (!x || !packed || !scale || !shift || !res || !out || B <= 0) return OTP_ERR_BAD_ARG;
    if (h1 && !ln_gamma) return OTP_ERR_UNSUPPORTED;              // (the half-operand form exists with the fused LayerNorm only)
    if (!otp_mlp_x3_supported(C, HID, T)) return OTP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out)) & 7 ||
        reinterpret_cast<uintptr_t>(packed) & 15)
        return OTP_ERR_BAD_ARG;
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    auto pk = static_cast<const unsigned char*>(packed);
    if (C == 204) {                                              // 7-frame window: one token tile per wave, 128 tokens per workgroup
        constexpr int WAVES = 8;
        const size_t lds204 = 2 * (size_t)mx_block_bytes(204, h1 ? 1 : 2);
        const int tiles = otp_ceil_div(T, WAVES * 16);
        auto kern = h1 ? mlpx_kernel<204, 816, WAVES, true, 1, true>
                       : (ln_gamma ? mlpx_kernel<204, 816, WAVES, true, 1> : mlpx_kernel<204, 816, WAVES, false, 1>);
        OTP_ALLOW_BIG_LDS(kern, lds204);
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * tiles)), dim3(WAVES * 64), lds204, static_cast<hipStream_t>(stream), f(x), pk,
                           f(scale), f(shift), f(res), static_cast<float*>(out), T, tiles, f(ln_gamma), f(ln_beta), ln_eps, otp_range_word());
        return otp_launch_status();
    }
    const size_t lds = 2 * (size_t)mx_block_bytes(136, h1 ? 1 : 2);
    // Default: one token tile per wave (NT = 1), 128 tokens per 8-wave workgroup, <= 128 VGPRs, i.e. TWO 80 KB workgroups per CU
    // - 132 us at cfg2 against 149 for the balanced form.  (Opt-in until the packed-fp32 op_sel hazard of DESIGN.md section 3.1d
    // was found: the forward's replays differed with it - through the kernels of csrc/densex.hip next to it, not through this
    // one.)  OTP_MLP_NT1=0: the older forms - two token tiles per wave, or the balanced two-pass kernel from B * T / 432 >= 192;
    // OTP_MLP_BALANCED "0": never balanced, "2": balanced whenever the shape allows (tests).
    const char* bal = getenv("OTP_MLP_BALANCED");
    const bool bal_off = bal && bal[0] == '0', bal_force = bal && bal[0] == '2';
    const char* e1 = getenv("OTP_MLP_NT1");
    const bool nt1 = !(e1 && e1[0] == '0');
    if (h1 || (nt1 && !bal_force)) {
        constexpr int WAVES = 8;
        const int tiles = otp_ceil_div(T, WAVES * 16);
        auto kern = h1 ? mlpx_kernel<136, 544, WAVES, true, 1, true>
                       : (ln_gamma ? mlpx_kernel<136, 544, WAVES, true, 1> : mlpx_kernel<136, 544, WAVES, false, 1>);
        OTP_ALLOW_BIG_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * tiles)), dim3(WAVES * 64), lds, static_cast<hipStream_t>(stream), f(x), pk,
                           f(scale), f(shift), f(res), static_cast<float*>(out), T, tiles, f(ln_gamma), f(ln_beta), ln_eps, otp_range_word());
        return otp_launch_status();
    }
    if (!bal_off && T % (27 * 16) == 0 && (bal_force || (long)B * (T / (27 * 16)) >= 192)) {
        auto bk = ln_gamma ? mlpx_balanced_kernel<136, 544, true> : mlpx_balanced_kernel<136, 544, false>;
        OTP_ALLOW_BIG_LDS(bk, lds);
        const int per_b = T / (27 * 16);
        hipLaunchKernelGGL(bk, dim3((unsigned)(B * per_b)), dim3(512), lds, static_cast<hipStream_t>(stream), f(x), pk,
                           f(scale), f(shift), f(res), static_cast<float*>(out), T, per_b, f(ln_gamma), f(ln_beta), ln_eps, otp_range_word());
        return otp_launch_status();
    }
    constexpr int WAVES = 8;
    const int tiles = otp_ceil_div(T, WAVES * 32);
    auto kern = ln_gamma ? mlpx_kernel<136, 544, WAVES, true, 2> : mlpx_kernel<136, 544, WAVES, false, 2>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * tiles)), dim3(WAVES * 64), lds, static_cast<hipStream_t>(stream), f(x), pk,
                       f(scale), f(shift), f(res), static_cast<float*>(out), T, tiles, f(ln_gamma), f(ln_beta), ln_eps, otp_range_word());
    return otp_launch_status();
}
}  // namespace

extern "C" int otp_mlp_x3(const void* x, const void* packed, const void* scale, const void* shift, const void* res, void* out,
                          int B, int C, int HID, int T, void* stream) {
    return mlpx_launch(x, packed, scale, shift, res, out, nullptr, nullptr, 0.f, B, C, HID, T, stream);
}

extern "C" int otp_ln_mlp_x3(const void* y, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* packed,
                             const void* scale, const void* shift, void* out, int B, int C, int HID, int T, void* stream) {
    if (!ln_gamma || !ln_beta) return OTP_ERR_BAD_ARG;
    return mlpx_launch(y, packed, scale, shift, y, out, ln_gamma, ln_beta, ln_eps, B, C, HID, T, stream);
}

/* the same launch with the fp16 engine's arithmetic: half operands rounded once, one MFMA per product, the hidden layer rounded to
 * half behind a 6e-5 GELU; fp32 tensors, LayerNorm and accumulation.  `packed`: the hi-only image of otp_mlp_h1_pack */
extern "C" int otp_ln_mlp_h1(const void* y, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* packed,
                             const void* scale, const void* shift, void* out, int B, int C, int HID, int T, void* stream) {
    if (!ln_gamma || !ln_beta) return OTP_ERR_BAD_ARG;
    return mlpx_launch(y, packed, scale, shift, y, out, ln_gamma, ln_beta, ln_eps, B, C, HID, T, stream, true);
}
