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

// ---- attention front end: q, k, v = W_p . LN_p(dwconv3_p(x)) + b_p in one launch (stride 1) -------------------------------
// MaskedMHCA.forward up to the attention product (model/blocks.py:406-419 with the modules of :359-386): three depthwise
// k = 3 convolutions over T (zero padding, no bias) of the same input, a channel LayerNorm after each, then the three
// pointwise projections.  The depthwise outputs only ever exist as the B-operand fragments of the projection: a wave owns
// 32 tokens, reads each channel's 4-token window (token pair + one neighbour either side), and the LayerNorm statistics of
// a token are an in-lane sum over its k-steps plus two cross-lane adds over the four k-slots.  Saves the 3 x (B, C, T)
// round trip through HBM between otp_dwconv_ln3 and otp_dense_cc.
// table[p][s][kq] = {dw0, dw1, dw2, gamma, beta, 0, 0, 0} for channel 4s + kq of problem p.
__global__ void qkv_table_kernel(const float* __restrict__ dwq, const float* __restrict__ dwk, const float* __restrict__ dwv,
                                 const float* __restrict__ gq, const float* __restrict__ bq, const float* __restrict__ gk,
                                 const float* __restrict__ bk, const float* __restrict__ gv, const float* __restrict__ bv,
                                 float* __restrict__ table, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * C * 8) return;
    const int p = idx / (C * 8), c = (idx / 8) % C, j = idx % 8;
    const float* dw = p == 0 ? dwq : p == 1 ? dwk : dwv;
    const float* g = p == 0 ? gq : p == 1 ? gk : gv;
    const float* b = p == 0 ? bq : p == 1 ? bk : bv;
    table[idx] = j < 3 ? dw[c * 3 + j] : j == 3 ? g[c] : j == 4 ? b[c] : 0.f;   // (c = 4s + kq: [s][kq] order is c order)
}

struct QkvArgs {
    const float* packed[3];
    float* out[3];
};

__device__ __forceinline__ float kslot_sum(float v) {      // sum over the four k-slot lane groups (lanes n, n+16, n+32, n+48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int C>
__global__ __launch_bounds__(256, 2) void qkv_front_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                           QkvArgs A, int T, int tiles_per_b, float eps) {
    constexpr int KS = C / 4, KG = (KS + 3) / 4, MT = (C + 15) / 16;
    constexpr int BLK = dense_block_floats(C), BLK4 = BLK / 4, NST = BLK4 / 256, TAB = 3 * C * 8;
    __shared__ float lds[2 * BLK];
    __shared__ float tab[TAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    const int tok = tile * 128 + wave * 32 + 2 * n;
    const bool valid = tok < T;
    const int tokc = valid ? tok : T - 2;
    const bool l_ok = tokc >= 1, r_ok = tokc + 2 < T;
    const size_t base = (size_t)b * C * T;
    // one buffer resource per batch element; per-lane byte offset of (channel kq, token tokc), k-step offsets are uniform
    const otp_rsrc rx = make_rsrc(x + base, (size_t)C * T * sizeof(float));
    const int voff = (kq * T + tokc) * 4;
    constexpr float inv_c = 1.f / (float)C;
    for (int i = tid; i < TAB / 4; i += 256)
        reinterpret_cast<f32x4*>(tab)[i] = reinterpret_cast<const f32x4*>(table)[i];
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);

    for (int p = 0; p < 3; ++p) {
        const f32x4* pk = reinterpret_cast<const f32x4*>(A.packed[p]);
        float* __restrict__ out = A.out[p];
        // (every wave is past the last block of the previous problem: the barrier that ends its loop)
#pragma unroll
        for (int i = 0; i < NST; ++i) l4[tid + i * 256] = pk[tid + i * 256];
        if (p == 0) __syncthreads();                 // parameter table visible
        const float* tp = tab + p * C * 8 + kq * 8;
        f32x2 X[KS];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int so = 4 * s * T * 4;
            const f32x2 m = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, voff, so, 0));
            const float la = bload(rx, voff - 4, so), ld = bload(rx, voff + 8, so);
            const float a = l_ok ? la : 0.f, d = r_ok ? ld : 0.f;
            const f32x4 w = *reinterpret_cast<const f32x4*>(tp + s * 32);
            if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // bound the loads in flight (registers)
            X[s].x = w[0] * a + w[1] * m.x + w[2] * m.y;
            X[s].y = w[0] * m.x + w[1] * m.y + w[2] * d;
            s0 += X[s].x;
            s1 += X[s].y;
        }
        const float m0 = kslot_sum(s0) * inv_c, m1 = kslot_sum(s1) * inv_c;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            X[s].x -= m0;
            X[s].y -= m1;
            v0 += X[s].x * X[s].x;
            v1 += X[s].y * X[s].y;
        }
        const float r0 = 1.f / sqrtf(kslot_sum(v0) * inv_c + eps), r1 = 1.f / sqrtf(kslot_sum(v1) * inv_c + eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float g = tp[s * 32 + 3], be = tp[s * 32 + 4];
            X[s].x = X[s].x * r0 * g + be;
            X[s].y = X[s].y * r1 * g + be;
        }
        __syncthreads();                             // weight block 0 of this problem visible

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
                        const f32x2 v = {acc0[i] * sc[i] + sh[i], acc1[i] * sc[i] + sh[i]};
                        *reinterpret_cast<f32x2*>(out + base + (size_t)c * T + tok) = v;
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

extern "C" size_t otp_qkv_front_table_bytes(int C) { return C > 0 ? (size_t)3 * C * 8 * sizeof(float) : 0; }

extern "C" int otp_qkv_front_pack_table(const void* dwq, const void* dwk, const void* dwv, const void* gq, const void* bq,
                                        const void* gk, const void* bk, const void* gv, const void* bv, void* table, int C,
                                        void* stream) {
    if (!dwq || !dwk || !dwv || !gq || !bq || !gk || !bk || !gv || !bv || !table || C <= 0) return OTP_ERR_BAD_ARG;
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    hipLaunchKernelGGL(qkv_table_kernel, dim3(otp_ceil_div(3 * C * 8, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       f(dwq), f(dwk), f(dwv), f(gq), f(bq), f(gk), f(bk), f(gv), f(bv), static_cast<float*>(table), C);
    return otp_launch_status();
}

extern "C" int otp_qkv_front(const void* x, const void* table, const void* packed_q, const void* packed_k,
                             const void* packed_v, void* q, void* k, void* v, int B, int C, int T, float eps, void* stream) {
    if (!x || !table || !packed_q || !packed_k || !packed_v || !q || !k || !v || B <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_dense_cc_supported(C, T)) return OTP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
         reinterpret_cast<uintptr_t>(v)) & 7 ||
        (reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(packed_q) | reinterpret_cast<uintptr_t>(packed_k) |
         reinterpret_cast<uintptr_t>(packed_v)) & 15)
        return OTP_ERR_BAD_ARG;
    QkvArgs a;
    a.packed[0] = static_cast<const float*>(packed_q); a.packed[1] = static_cast<const float*>(packed_k);
    a.packed[2] = static_cast<const float*>(packed_v);
    a.out[0] = static_cast<float*>(q); a.out[1] = static_cast<float*>(k); a.out[2] = static_cast<float*>(v);
    const int tiles = otp_ceil_div(T, 128);
    hipLaunchKernelGGL(qkv_front_kernel<136>, dim3((unsigned)(B * tiles)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(x), static_cast<const float*>(table), a, T, tiles, eps);
    return otp_launch_status();
}
