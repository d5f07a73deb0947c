// The TransformerBlock MLP of the temporal encoders (reference model/blocks.py:248-254, applied at :277-279) as ONE launch:
//     out = res + scale * (W2 . gelu(W1 . x + b1)) + shift          x, res, out (B, C, T); W1 (HID, C); W2 (C, HID)
// (eval mode: both Dropouts and the drop-path are identities; `scale` is AffineDropPath's per-channel scale, `shift` =
// b2 * scale).  The 4C-wide hidden activation never leaves the register file: a wave owns 32 tokens, keeps their C input
// channels as MFMA B-operand fragments (C/4 k-steps x 2 column tiles), and walks the hidden dimension 16 channels at a
// time - phase 1 produces a 16 x 32 hidden tile with v_mfma_f32_16x16x4_f32, bias + erf-GELU are applied to the
// accumulator registers, and phase 2 feeds those same registers back as the B operand of the second GEMM.  No cross-lane
// movement is needed for that: accumulator register i of lane (g = lane / 16, n = lane % 16) is hidden channel 4g + i of
// column n, and a B operand wants "k-slot g, column n" - so the i-th k-step of phase 2 simply contracts over the hidden
// channels {4g + i}, and the packed W2 fragments are laid out in that order.  Columns are permuted too (column n of tile j
// is token 2n + j) so every global access is an 8-byte load / store of two neighbouring tokens.
// Weights are pre-packed per 16-channel hidden block in LDS fragment order and streamed L2 -> registers -> LDS, double
// buffered, one barrier per block, shared by the waves of the workgroup.
// Roofline: MFMA-bound.  2*2*C*HID flop/token (C = 136: 295.9 kflop) against 8 bytes/token/channel of HBM traffic (x, res
// in, out: 1.6 kB/token).
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// erf-GELU on a pair of accumulator values, branch-free and in packed-f32 form (v_pk_fma_f32): erf(t) = t P(t^2) / Q(t^2)
// on |t| <= 4 (the classic single-precision rational fit; erf is 1 - 1.5e-8 beyond), max abs error 4.2e-7 on erf and
// 7e-7 on gelu(x) over the whole real line - checked against fp64 in tests/test_gpu_ops.py.  The libm erff the other
// epilogues call has two data-dependent paths, and both would run for every accumulator register of a wave here.
__device__ __forceinline__ f32x2 mlp_gelu2(f32x2 x) {
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

__device__ __forceinline__ float kslot_sum(float v) {      // sum over the four k-slot lane groups (lanes n, n+16, n+32, n+48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// floats per hidden block (a multiple of 4: the block is copied as 16-byte pieces)
constexpr int mlp_block_floats(int C) { return ((C / 4 + 3) / 4 + (C + 15) / 16) * 256 + 16; }

// packed[h][...]: phase-1 A fragments [kgroup][lane][4 k-steps], phase-2 A fragments [mtile][lane][4 steps], b1[16]
__global__ void mlp_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                float* __restrict__ packed, int C, int HID) {
    const int KS = C / 4, KG = (KS + 3) / 4, MT = (C + 15) / 16, BLK = mlp_block_floats(C);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (HID / 16) * BLK) return;
    const int h = idx / BLK, r = idx % BLK;
    float v = 0.f;
    if (r < KG * 256) {
        const int sg = r / 256, l = (r % 256) / 4, q = r % 4, s = 4 * sg + q;
        if (s < KS) v = w1[(size_t)(16 * h + (l & 15)) * C + 4 * s + (l >> 4)];
    } else if (r < (KG + MT) * 256) {
        const int r2 = r - KG * 256, mt = r2 / 256, l = (r2 % 256) / 4, i = r2 % 4;
        const int row = 16 * mt + (l & 15);
        if (row < C) v = w2[(size_t)row * HID + 16 * h + 4 * (l >> 4) + i];
    } else if (r < (KG + MT) * 256 + 16) {
        v = b1[16 * h + r - (KG + MT) * 256];
    }
    packed[idx] = v;
}

// One pass of a wave over NT (1 or 2) column tiles of 16 tokens starting at token tok0, all HID / 16 hidden blocks; every wave
// of the workgroup calls it together (the weight blocks go through LDS behind one barrier per block).
// NT == 2: column n of tile j is token tok0 + 2n + j (8-byte accesses); NT == 1: token tok0 + n.
template <int C, int HID, int NTHR, bool LN, int NT>
__device__ __forceinline__ void mlp_pass(const float* __restrict__ x, const float* __restrict__ packed,
                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                         const float* __restrict__ res, float* __restrict__ out, int T, size_t base, int tok0,
                                         float* lds, const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta,
                                         float ln_eps) {
    constexpr int KS = C / 4, KG = (KS + 3) / 4, HT = HID / 16, MT = (C + 15) / 16;
    constexpr int BLK = mlp_block_floats(C), BLK4 = BLK / 4, NST = (BLK4 + NTHR - 1) / NTHR;
    const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, n = lane & 15;
    const int tok = tok0 + NT * n;
    const bool valid = tok < T;                       // NT == 2: T is even, a token pair is inside or outside together

    // weights of hidden block 0 -> LDS buffer 0 (every wave is past the previous pass: the barrier that ended its loop)
    const f32x4* pk = reinterpret_cast<const f32x4*>(packed);
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int i = 0; i < NST; ++i)
        if (BLK4 % NTHR == 0 || tid + i * NTHR < BLK4) l4[tid + i * NTHR] = pk[tid + i * NTHR];

    // the wave's input columns as B fragments: X[s][j] = x[4s + kq][token of column n of tile j]
    float X[KS][NT];
#pragma unroll
    for (int s = 0; s < KS; ++s) {     // columns past T read the last ones instead (their results are never stored)
        const float* src = x + base + (size_t)(4 * s + kq) * T + (valid ? tok : T - NT);
        if (NT == 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(src);
            X[s][0] = v.x;
            X[s][NT - 1] = v.y;
        } else {
            X[s][0] = *src;
        }
    }
    if (LN) {
        constexpr float inv_c = 1.f / (float)C;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s0 = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) s0 += X[s][j];
            const float m0 = kslot_sum(s0) * inv_c;
            float v0 = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                X[s][j] -= m0;
                v0 += X[s][j] * X[s][j];
            }
            const float r0 = 1.f / sqrtf(kslot_sum(v0) * inv_c + ln_eps);
#pragma unroll
            for (int s = 0; s < KS; ++s) X[s][j] = X[s][j] * r0 * ln_gamma[4 * s + kq] + ln_beta[4 * s + kq];
        }
    }
    f32x4 Y[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) Y[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int h = 0; h < HT; ++h) {
        // next block's weights on their way while this one is consumed
        f32x4 stage[NST];
        if (h + 1 < HT) {
            const f32x4* src = pk + (size_t)(h + 1) * BLK4;
#pragma unroll
            for (int i = 0; i < NST; ++i)
                if (BLK4 % NTHR == 0 || tid + i * NTHR < BLK4) stage[i] = src[tid + i * NTHR];
        }
        const float* P1 = lds + (h & 1) * BLK;
        const float* P2 = P1 + KG * 256;
        const float* PB = P2 + MT * 256;
        // phase 1: hidden tile (16 channels x 16 NT tokens) = W1[16h .. 16h+15][:] . X
        // (the accumulators start from b1: register i of lane group kq is hidden channel 16h + 4 kq + i)
        f32x4 H[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) H[j] = *reinterpret_cast<const f32x4*>(PB + 4 * kq);
#pragma unroll
        for (int sg = 0; sg < KG; ++sg) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(P1 + (sg * 64 + lane) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = 4 * sg + q;
                if (s < KS) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) H[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], X[s][j], H[j], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const f32x2 g0 = mlp_gelu2(f32x2{H[j][0], H[j][1]}), g1 = mlp_gelu2(f32x2{H[j][2], H[j][3]});
            H[j] = f32x4{g0.x, g0.y, g1.x, g1.y};
        }
        // phase 2: Y += W2[:, 16h .. 16h+15] . hidden tile, contraction step i over the hidden channels {4g + i}
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(P2 + (mt * 64 + lane) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) Y[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], H[j][i], Y[mt][j], 0, 0, 0);
        }
        if (h + 1 < HT) {
            f32x4* dst = reinterpret_cast<f32x4*>(lds + ((h + 1) & 1) * BLK);
#pragma unroll
            for (int i = 0; i < NST; ++i)
                if (BLK4 % NTHR == 0 || tid + i * NTHR < BLK4) dst[tid + i * NTHR] = stage[i];
        }
        __syncthreads();
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

// LN: x is the block's un-normalised input and the kernel applies the channel LayerNorm in front of the MLP itself
// (TransformerBlock.ln2, model/blocks.py:95-110: biased variance, eps inside the root) - a token's channels are the wave's
// k-steps x the four k-slot lane groups, so its statistics are an in-lane sum and two cross-lane adds.
template <int C, int HID, int WAVES, bool LN>
__global__ __launch_bounds__(WAVES * 64, 2) void mlp_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ packed, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ out, int T, int tiles_per_b,
    const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, float ln_eps) {
    extern __shared__ float lds[];                    // 2 x BLK floats
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    mlp_pass<C, HID, WAVES * 64, LN, 2>(x, packed, scale, shift, res, out, T, (size_t)b * C * T,
                                        tile * (WAVES * 32) + wave * 32, lds, ln_gamma, ln_beta, ln_eps);
}

// Balanced form for token counts that are multiples of 27 column tiles per workgroup (T = 6912 = 16 x 27 x 16): 6912 x B / 16
// column tiles over the 1024 SIMDs of the chip are 6.75 per SIMD, so equal workgroups of 4 x 2 tiles need two rounds of
// four tiles per SIMD (8) where 7 would do.  Here one 8-wave workgroup per CU owns 27 tiles and walks them in two passes:
// every wave takes two tiles in the first, then the waves of a SIMD (w, w + 4) take 2 + 1, 2 + 1, 2 + 1 and 1 + 1.
template <int C, int HID, bool LN>
__global__ __launch_bounds__(512, 2) void mlp_fused_balanced_kernel(
    const float* __restrict__ x, const float* __restrict__ packed, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ out, int T, int wgs_per_b,
    const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, float ln_eps) {
    extern __shared__ float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / wgs_per_b, part = blockIdx.x - b * wgs_per_b;
    const size_t base = (size_t)b * C * T;
    const int t0 = part * (27 * 16);
    mlp_pass<C, HID, 512, LN, 2>(x, packed, scale, shift, res, out, T, base, t0 + wave * 32, lds, ln_gamma, ln_beta, ln_eps);
    if (wave < 3) {
        mlp_pass<C, HID, 512, LN, 2>(x, packed, scale, shift, res, out, T, base, t0 + (16 + 2 * wave) * 16, lds, ln_gamma,
                                     ln_beta, ln_eps);
    } else {
        // waves 4, 5, 6 -> tiles 22, 23, 24; wave 3 -> 25; wave 7 -> 26
        const int tile = wave == 3 ? 25 : (wave == 7 ? 26 : 18 + wave);
        mlp_pass<C, HID, 512, LN, 1>(x, packed, scale, shift, res, out, T, base, t0 + tile * 16, lds, ln_gamma, ln_beta,
                                     ln_eps);
    }
}

#ifndef OTP_MLP_WAVES
#define OTP_MLP_WAVES 4
#endif
constexpr int MLP_WAVES = OTP_MLP_WAVES;

}  // namespace

extern "C" int otp_mlp_fused_supported(int C, int HID, int T) {
    return (C == 136 && HID == 544 && T > 0 && T % 2 == 0) ? 1 : 0;
}

extern "C" size_t otp_mlp_fused_weight_bytes(int C, int HID) {
    if (C <= 0 || C % 4 || HID <= 0 || HID % 16) return 0;
    return (size_t)(HID / 16) * mlp_block_floats(C) * sizeof(float);
}

extern "C" int otp_mlp_fused_pack(const void* w1, const void* b1, const void* w2, void* packed, int C, int HID,
                                  void* stream) {
    if (!w1 || !b1 || !w2 || !packed) return OTP_ERR_BAD_ARG;
    const size_t bytes = otp_mlp_fused_weight_bytes(C, HID);
    if (!bytes) return OTP_ERR_UNSUPPORTED;
    const int total = (int)(bytes / sizeof(float));
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(otp_ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w1), static_cast<const float*>(b1), static_cast<const float*>(w2),
                       static_cast<float*>(packed), C, HID);
    return otp_launch_status();
}

namespace {
int mlp_launch(const void* x, const void* packed, const void* scale, const void* shift, const void* res, void* out,
               const void* ln_gamma, const void* ln_beta, float ln_eps, int B, int C, int HID, int T, void* stream) {
    if (!x || !packed || !scale || !shift || !res || !out || B <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_mlp_fused_supported(C, HID, T)) return OTP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out)) & 7 ||
        reinterpret_cast<uintptr_t>(packed) & 15)
        return OTP_ERR_BAD_ARG;
    const int tiles = otp_ceil_div(T, MLP_WAVES * 32);
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    const size_t lds = 2 * (size_t)mlp_block_floats(136) * sizeof(float);
    const char* bal = getenv("OTP_MLP_BALANCED");               // "0": never, "2": whenever the shape allows (tests)
    const bool bal_off = bal && bal[0] == '0', bal_force = bal && bal[0] == '2';
    if (!bal_off && T % (27 * 16) == 0 && (bal_force || (long)B * (T / (27 * 16)) >= 192)) {
        // one 8-wave workgroup per CU and two passes: 7 column tiles per SIMD instead of 8
        auto bk = ln_gamma ? mlp_fused_balanced_kernel<136, 544, true> : mlp_fused_balanced_kernel<136, 544, false>;
        const int per_b = T / (27 * 16);
        hipLaunchKernelGGL(bk, dim3((unsigned)(B * per_b)), dim3(512), lds, static_cast<hipStream_t>(stream), f(x), f(packed),
                           f(scale), f(shift), f(res), static_cast<float*>(out), T, per_b, f(ln_gamma), f(ln_beta), ln_eps);
        return otp_launch_status();
    }
    auto kern = ln_gamma ? mlp_fused_kernel<136, 544, MLP_WAVES, true> : mlp_fused_kernel<136, 544, MLP_WAVES, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * tiles)), dim3(MLP_WAVES * 64), lds, static_cast<hipStream_t>(stream),
                       f(x), f(packed), f(scale), f(shift), f(res), static_cast<float*>(out), T, tiles, f(ln_gamma),
                       f(ln_beta), ln_eps);
    return otp_launch_status();
}
}  // namespace

extern "C" int otp_mlp_fused(const void* x, const void* packed, const void* scale, const void* shift, const void* res,
                             void* out, int B, int C, int HID, int T, void* stream) {
    return mlp_launch(x, packed, scale, shift, res, out, nullptr, nullptr, 0.f, B, C, HID, T, stream);
}

extern "C" int otp_ln_mlp_fused(const void* y, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* packed,
                                const void* scale, const void* shift, void* out, int B, int C, int HID, int T,
                                void* stream) {
    if (!ln_gamma || !ln_beta) return OTP_ERR_BAD_ARG;
    return mlp_launch(y, packed, scale, shift, y, out, ln_gamma, ln_beta, ln_eps, B, C, HID, T, stream);
}
