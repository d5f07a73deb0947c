// The C -> C pointwise projections of MaskedMHCA (query / key / value / proj, reference model/blocks.py:383-386 applied at
// :417-419 and :450) on (B, C, T) tensors, up to three independent problems per launch (q, k and v share a shape):
//     out = scale * (W . x) + shift (+ res)            W (C, C); scale / shift per output channel
// Same register-resident-input scheme as csrc/mlp.hip: a wave owns 32 tokens and holds their C channels as MFMA B-operand
// fragments, loaded with 8-byte accesses (column n of tile j is token 2n + j); the weights stream through LDS one 16-row
// output block at a time (double buffered, one barrier per block) and each block's 16 x 32 result is scaled, shifted,
// added to the residual and stored as soon as its 2 x C/4 MFMAs are done, so only 8 accumulator registers are live.
// Roofline: HBM-bound - 8 bytes / token / channel (x in, out), + 4 with a residual, against 2 C^2 flop / token.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// floats per 16-row output block: A fragments [kgroup][lane][4 k-steps], scale[16], shift[16]; whole 256 x 16-byte passes
constexpr int dense_block_floats(int C) { return (((C / 4 + 3) / 4) * 256 + 32 + 1023) / 1024 * 1024; }

__global__ void dense_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                  float* __restrict__ packed, int C) {
    const int KS = C / 4, KG = (KS + 3) / 4, MT = (C + 15) / 16, BLK = dense_block_floats(C);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= MT * BLK) return;
    const int mt = idx / BLK, r = idx % BLK;
    float v = 0.f;
    if (r < KG * 256) {
        const int sg = r / 256, l = (r % 256) / 4, q = r % 4, s = 4 * sg + q, row = 16 * mt + (l & 15);
        if (s < KS && row < C) v = w[(size_t)row * C + 4 * s + (l >> 4)];
    } else if (r < KG * 256 + 16) {
        const int c = 16 * mt + r - KG * 256;
        if (c < C) v = scale ? scale[c] : 1.f;
    } else if (r < KG * 256 + 32) {
        const int c = 16 * mt + r - KG * 256 - 16;
        if (c < C && shift) v = shift[c];
    }
    packed[idx] = v;
}

struct DenseArgs {
    const float* x[3];
    const float* packed[3];
    const float* res[3];
    float* out[3];
};

template <int C>
__global__ __launch_bounds__(256, 3) void dense_cc_kernel(DenseArgs A, int T, int tiles_per_b) {
    constexpr int KS = C / 4, KG = (KS + 3) / 4, MT = (C + 15) / 16;
    constexpr int BLK = dense_block_floats(C), BLK4 = BLK / 4, NST = BLK4 / 256;
    __shared__ float lds[2 * BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    const int tok = tile * 128 + wave * 32 + 2 * n;
    const bool valid = tok < T;
    const size_t base = (size_t)b * C * T;
    const float* __restrict__ x = A.x[blockIdx.y];
    const float* __restrict__ res = A.res[blockIdx.y];
    float* __restrict__ out = A.out[blockIdx.y];
    const f32x4* pk = reinterpret_cast<const f32x4*>(A.packed[blockIdx.y]);

    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int i = 0; i < NST; ++i) l4[tid + i * 256] = pk[tid + i * 256];
    f32x2 X[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)       // columns past T read the last pair instead (their results are never stored)
        X[s] = *reinterpret_cast<const f32x2*>(x + base + (size_t)(4 * s + kq) * T + (valid ? tok : T - 2));
    __syncthreads();

    for (int mt = 0; mt < MT; ++mt) {
        f32x4 stage[NST];
        if (mt + 1 < MT) {
            const f32x4* src = pk + (size_t)(mt + 1) * BLK4;
#pragma unroll
            for (int i = 0; i < NST; ++i) stage[i] = src[tid + i * 256];
        }
        const float* P = lds + (mt & 1) * BLK;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sg = 0; sg < KG; ++sg) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(P + (sg * 64 + lane) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = 4 * sg + q;
                if (s < KS) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], X[s].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], X[s].y, acc1, 0, 0, 0);
                }
            }
        }
        const f32x4 sc = *reinterpret_cast<const f32x4*>(P + KG * 256 + 4 * kq);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(P + KG * 256 + 16 + 4 * kq);
        if (valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 16 * mt + 4 * kq + i;
                if (c < C) {
                    const size_t o = base + (size_t)c * T + tok;
                    f32x2 v = {acc0[i] * sc[i] + sh[i], acc1[i] * sc[i] + sh[i]};
                    if (res) v += *reinterpret_cast<const f32x2*>(res + o);
                    *reinterpret_cast<f32x2*>(out + o) = v;
                }
            }
        }
        if (mt + 1 < MT) {
            f32x4* dst = reinterpret_cast<f32x4*>(lds + ((mt + 1) & 1) * BLK);
#pragma unroll
            for (int i = 0; i < NST; ++i) dst[tid + i * 256] = stage[i];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int otp_dense_cc_supported(int C, int T) { return (C == 136 && T > 0 && T % 2 == 0) ? 1 : 0; }

extern "C" size_t otp_dense_cc_weight_bytes(int C) {
    if (C <= 0 || C % 4) return 0;
    return (size_t)((C + 15) / 16) * dense_block_floats(C) * sizeof(float);
}

extern "C" int otp_dense_cc_pack(const void* w, const void* scale, const void* shift, void* packed, int C, void* stream) {
    if (!w || !packed) return OTP_ERR_BAD_ARG;
    const size_t bytes = otp_dense_cc_weight_bytes(C);
    if (!bytes) return OTP_ERR_UNSUPPORTED;
    const int total = (int)(bytes / sizeof(float));
    hipLaunchKernelGGL(dense_pack_kernel, dim3(otp_ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w), static_cast<const float*>(scale), static_cast<const float*>(shift),
                       static_cast<float*>(packed), C);
    return otp_launch_status();
}

extern "C" int otp_dense_cc(const void* const* x, const void* const* packed, const void* const* res, void* const* out,
                            int nprob, int B, int C, int T, void* stream) {
    if (!x || !packed || !out || nprob < 1 || nprob > 3 || B <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_dense_cc_supported(C, T)) return OTP_ERR_UNSUPPORTED;
    DenseArgs a = {};
    for (int i = 0; i < nprob; ++i) {
        a.x[i] = static_cast<const float*>(x[i]);
        a.packed[i] = static_cast<const float*>(packed[i]);
        a.res[i] = res ? static_cast<const float*>(res[i]) : nullptr;
        a.out[i] = static_cast<float*>(out[i]);
        if (!a.x[i] || !a.packed[i] || !a.out[i]) return OTP_ERR_BAD_ARG;
        if ((reinterpret_cast<uintptr_t>(a.x[i]) | reinterpret_cast<uintptr_t>(a.res[i]) | reinterpret_cast<uintptr_t>(a.out[i])) & 7 ||
            reinterpret_cast<uintptr_t>(a.packed[i]) & 15)
            return OTP_ERR_BAD_ARG;
    }
    const int tiles = otp_ceil_div(T, 128);
    hipLaunchKernelGGL(dense_cc_kernel<136>, dim3((unsigned)(B * tiles), (unsigned)nprob), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a, T, tiles);
    return otp_launch_status();
}
