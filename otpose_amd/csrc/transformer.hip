// ConvTransformer kernels for gfx950: channel LayerNorm (+ MaxPool1d skip), fused 3x depthwise-conv +
// LayerNorm, channel attention (hs x hs scores, contraction over T) on the f32 matrix cores, linear
// up-sampling.  Reference: model/blocks.py:95-110 (LayerNorm), :234-238 (pool_skip), :400-453
// (MaskedMHCA.forward), model/ConvVideoTransformer.py:108,163-184 (nn.Upsample linear).
//
// All tensors are (B, C, T) with T contiguous, so a thread owns one time step t and walks the C
// channels: every global access of a wave is a coalesced 256-byte run along T.
//
// This file is compiled twice.  transformer.o: everything, split products on IEEE-half pieces (csrc/common.h).
// transformer_grad.o (transformer_grad.hip: -DOTP_X3_BF16 -DOTP_X3_GRAD_COPY): only otp_chan_attn_scores_bf16p /
// otp_chan_attn_apply_bf16p, the two attention products with bfloat16 pieces, for the training backward, whose operands are
// GRADIENTS of unknown magnitude (a half piece flushes 1e-8 to zero; a bfloat16 pair keeps 16-17 bits at any magnitude).
#include "common.h"
#include <cstdlib>
#ifndef OTP_ENTRY
#define OTP_ENTRY(name) name
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- channel LayerNorm ---------------------------------------------------------------------------
// CREG > 0: the C <= CREG channel values of a time step are held in registers (one HBM read).
template <int CREG>
__global__ __launch_bounds__(256) void ln_channel_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          int C, int T, float eps) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const size_t base = (size_t)blockIdx.y * C * T + t;
    const float inv_c = 1.f / (float)C;
    if (CREG > 0) {
        float v[CREG > 0 ? CREG : 1];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            v[c] = c < C ? x[base + (size_t)c * T] : 0.f;
            s += v[c];
        }
        const float mu = s * inv_c;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            v[c] -= mu;
            q += c < C ? v[c] * v[c] : 0.f;
        }
        const float rs = 1.f / sqrtf(q * inv_c + eps);
#pragma unroll
        for (int c = 0; c < CREG; ++c)
            if (c < C) y[base + (size_t)c * T] = v[c] * rs * gamma[c] + beta[c];
    } else {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += x[base + (size_t)c * T];
        const float mu = s * inv_c;
        float q = 0.f;
        for (int c = 0; c < C; ++c) {
            float dlt = x[base + (size_t)c * T] - mu;
            q += dlt * dlt;
        }
        const float rs = 1.f / sqrtf(q * inv_c + eps);
        for (int c = 0; c < C; ++c) y[base + (size_t)c * T] = (x[base + (size_t)c * T] - mu) * rs * gamma[c] + beta[c];
    }
}

// The same LayerNorm for wide channel counts (C = 136 of the temporal encoders): workgroup = 64 time steps x 4 waves that
// split the channels (lane = time step: 256-byte coalesced rows), statistics reduced across the waves through LDS, mean first,
// then the biased variance of the centred values.  One thread walking all 136 channels (ln_channel_kernel<136>) left the chip
// with 432 workgroups of dependent loads: 34 us for a 120 MB pass; this form runs 1728 workgroups.
template <int CW>
__global__ __launch_bounds__(256) void ln_channel_split_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y,
                                                                int C, int T, float eps) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const size_t base = (size_t)blockIdx.y * C * T + (live ? t : 0);
    const float inv_c = 1.f / (float)C;
    const int cw = (C + 3) / 4, cbeg = wave * cw;
    float v[CW];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        v[i] = (i < cw && c < C) ? x[base + (size_t)c * T] : 0.f;
        s += v[i];
    }
    red[0][wave][lane] = s;
    __syncthreads();
    const float mu = ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane])) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) {
            v[i] -= mu;
            q += v[i] * v[i];
        }
    }
    red[1][wave][lane] = q;
    __syncthreads();
    const float rs = 1.f / sqrtf(((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane])) * inv_c + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) y[base + (size_t)c * T] = v[i] * rs * gamma[c] + beta[c];
    }
}

// MaxPool1d(kernel 3, stride 2, padding 1): To = (T + 2 - 3) / 2 + 1
__global__ void maxpool3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int To) {
    const int to = blockIdx.x * blockDim.x + threadIdx.x;
    if (to >= To) return;
    const float* xr = x + (size_t)blockIdx.y * T;
    const int t = 2 * to;
    float m = xr[t];
    if (t - 1 >= 0) m = fmaxf(m, xr[t - 1]);
    if (t + 1 < T) m = fmaxf(m, xr[t + 1]);
    y[(size_t)blockIdx.y * To + to] = m;
}

// ---- three depthwise convs (k=3, pad 1, stride s) each followed by a channel LayerNorm -------------
// Workgroup = 64 output time steps (lane = time step: 256-byte coalesced rows) x 4 waves that split the channels;
// every conv output is computed once and kept in registers (CW channels x 3 per lane), the channel statistics are
// reduced across the four waves through LDS: mean first, then the biased variance of the centred values
// (the two-pass form of blocks.py:100-103).
template <int CW>
__global__ __launch_bounds__(256) void dwconv_ln3_kernel(
    const float* __restrict__ x, const float* __restrict__ dwq, const float* __restrict__ dwk,
    const float* __restrict__ dwv, const float* __restrict__ gq, const float* __restrict__ bq,
    const float* __restrict__ gk, const float* __restrict__ bk, const float* __restrict__ gv,
    const float* __restrict__ bv, float* __restrict__ q, float* __restrict__ k, float* __restrict__ v,
    int C, int T, int To, int stride, float eps) {
    __shared__ float red[3][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int to = blockIdx.x * 64 + lane;
    const bool live = to < To;
    const float* xb = x + (size_t)blockIdx.y * C * T;
    const size_t ob = (size_t)blockIdx.y * C * To + to;
    const int t0 = (live ? to : 0) * stride - 1;
    const bool l_ok = t0 >= 0, r_ok = t0 + 2 < T;
    const float inv_c = 1.f / (float)C;
    const int cw = (C + 3) / 4;                 // channels per wave (<= CW)
    const int cbeg = wave * cw;
    float dq[CW], dk[CW], dv[CW];
    float sq = 0.f, sk = 0.f, sv = 0.f;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        dq[i] = dk[i] = dv[i] = 0.f;
        if (i < cw && c < C) {
            const float* xr = xb + (size_t)c * T + t0;
            const float a = l_ok ? xr[0] : 0.f, b = xr[1], cc = r_ok ? xr[2] : 0.f;
            dq[i] = dwq[c * 3] * a + dwq[c * 3 + 1] * b + dwq[c * 3 + 2] * cc;
            dk[i] = dwk[c * 3] * a + dwk[c * 3 + 1] * b + dwk[c * 3 + 2] * cc;
            dv[i] = dwv[c * 3] * a + dwv[c * 3 + 1] * b + dwv[c * 3 + 2] * cc;
            sq += dq[i]; sk += dk[i]; sv += dv[i];
        }
    }
    red[0][wave][lane] = sq; red[1][wave][lane] = sk; red[2][wave][lane] = sv;
    __syncthreads();
    const float mq = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) * inv_c;
    const float mk = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) * inv_c;
    const float mv = (red[2][0][lane] + red[2][1][lane] + red[2][2][lane] + red[2][3][lane]) * inv_c;
    __syncthreads();
    float vq = 0.f, vk = 0.f, vv = 0.f;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) {
            dq[i] -= mq; dk[i] -= mk; dv[i] -= mv;
            vq += dq[i] * dq[i]; vk += dk[i] * dk[i]; vv += dv[i] * dv[i];
        }
    }
    red[0][wave][lane] = vq; red[1][wave][lane] = vk; red[2][wave][lane] = vv;
    __syncthreads();
    const float rq = 1.f / sqrtf((red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) * inv_c + eps);
    const float rk = 1.f / sqrtf((red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) * inv_c + eps);
    const float rv = 1.f / sqrtf((red[2][0][lane] + red[2][1][lane] + red[2][2][lane] + red[2][3][lane]) * inv_c + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) {
            q[ob + (size_t)c * To] = dq[i] * rq * gq[c] + bq[c];
            k[ob + (size_t)c * To] = dk[i] * rk * gk[c] + bk[c];
            v[ob + (size_t)c * To] = dv[i] * rv * gv[c] + bv[c];
        }
    }
}

// any channel count: one thread per time step, three passes over the channels
__global__ __launch_bounds__(256) void dwconv_ln3_generic_kernel(
    const float* __restrict__ x, const float* __restrict__ dwq, const float* __restrict__ dwk,
    const float* __restrict__ dwv, const float* __restrict__ gq, const float* __restrict__ bq,
    const float* __restrict__ gk, const float* __restrict__ bk, const float* __restrict__ gv,
    const float* __restrict__ bv, float* __restrict__ q, float* __restrict__ k, float* __restrict__ v,
    int C, int T, int To, int stride, float eps) {
    const int to = blockIdx.x * blockDim.x + threadIdx.x;
    if (to >= To) return;
    const float* xb = x + (size_t)blockIdx.y * C * T;
    const size_t ob = (size_t)blockIdx.y * C * To + to;
    const int t0 = to * stride - 1;
    const bool l_ok = t0 >= 0, r_ok = t0 + 2 < T;
    const float inv_c = 1.f / (float)C;
    float sq = 0.f, sk = 0.f, sv = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* xr = xb + (size_t)c * T + t0;
        const float a = l_ok ? xr[0] : 0.f, b = xr[1], cc = r_ok ? xr[2] : 0.f;
        sq += dwq[c * 3] * a + dwq[c * 3 + 1] * b + dwq[c * 3 + 2] * cc;
        sk += dwk[c * 3] * a + dwk[c * 3 + 1] * b + dwk[c * 3 + 2] * cc;
        sv += dwv[c * 3] * a + dwv[c * 3 + 1] * b + dwv[c * 3 + 2] * cc;
    }
    const float mq = sq * inv_c, mk = sk * inv_c, mv = sv * inv_c;
    float vq = 0.f, vk = 0.f, vv = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* xr = xb + (size_t)c * T + t0;
        const float a = l_ok ? xr[0] : 0.f, b = xr[1], cc = r_ok ? xr[2] : 0.f;
        float d1 = dwq[c * 3] * a + dwq[c * 3 + 1] * b + dwq[c * 3 + 2] * cc - mq;
        float d2 = dwk[c * 3] * a + dwk[c * 3 + 1] * b + dwk[c * 3 + 2] * cc - mk;
        float d3 = dwv[c * 3] * a + dwv[c * 3 + 1] * b + dwv[c * 3 + 2] * cc - mv;
        vq += d1 * d1; vk += d2 * d2; vv += d3 * d3;
    }
    const float rq = 1.f / sqrtf(vq * inv_c + eps), rk = 1.f / sqrtf(vk * inv_c + eps), rv = 1.f / sqrtf(vv * inv_c + eps);
    for (int c = 0; c < C; ++c) {
        const float* xr = xb + (size_t)c * T + t0;
        const float a = l_ok ? xr[0] : 0.f, b = xr[1], cc = r_ok ? xr[2] : 0.f;
        float d1 = dwq[c * 3] * a + dwq[c * 3 + 1] * b + dwq[c * 3 + 2] * cc - mq;
        float d2 = dwk[c * 3] * a + dwk[c * 3 + 1] * b + dwk[c * 3 + 2] * cc - mk;
        float d3 = dwv[c * 3] * a + dwv[c * 3 + 1] * b + dwv[c * 3 + 2] * cc - mv;
        q[ob + (size_t)c * To] = d1 * rq * gq[c] + bq[c];
        k[ob + (size_t)c * To] = d2 * rk * gk[c] + bk[c];
        v[ob + (size_t)c * To] = d3 * rv * gv[c] + bv[c];
    }
}

// ---- channel attention ---------------------------------------------------------------------------
// Phase 1: partial scores.  grid (B*nh, NS); a workgroup (4 waves) owns a chunk of T, stages q and k
// tiles [HSP][64] through LDS (coalesced rows) and accumulates the NB x NB 16x16 score tiles with
// v_mfma_f32_16x16x4_f32 (A = q rows, B = k rows, K = time).  Wave w owns tiles w, w+4, ...
constexpr int ATT_TC = 64;            // time steps per staged tile
constexpr int ATT_TCP = ATT_TC + 2;   // LDS row pitch: (row*2 + k) mod 32 distinct inside each half wave

template <int NB>                     // NB = HSP / 16 (5 for hs = 68, 2 for hs = 17)
__global__ __launch_bounds__(256) void attn_scores_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           float* __restrict__ slabs, int hs, int T, int chunk) {
    constexpr int HSP = NB * 16, NT = NB * NB, TPW = (NT + 3) / 4;
    // float4 items of one [HSP][64] tile: 16 per row; NQ per thread and matrix
    constexpr int NQ = (HSP * (ATT_TC / 4) + 255) / 256;
    __shared__ float ql[HSP * ATT_TCP];
    __shared__ float kl[HSP * ATT_TCP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, s = blockIdx.y;
    const float* qb = q + (size_t)bh * hs * T;
    const float* kb = k + (size_t)bh * hs * T;
    const int t_begin = s * chunk, t_end = min(T, t_begin + chunk);
    f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kk = lane >> 4;
    // rows are T floats apart; 16-byte vectors need T % 4 == 0 (and the chunk start is a multiple of 64)
    const bool vec = (T & 3) == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0;
    f32x4 pq[NQ], pk[NQ];
    auto load_tile = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int idx = tid + j * 256;
            const int row = idx / (ATT_TC / 4), t = t0 + 4 * (idx - row * (ATT_TC / 4));
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (row < hs && idx < HSP * (ATT_TC / 4)) {
                const float* qp = qb + (size_t)row * T + t;
                const float* kp = kb + (size_t)row * T + t;
                if (vec && t + 3 < t_end) {
                    a = *reinterpret_cast<const f32x4*>(qp);
                    b = *reinterpret_cast<const f32x4*>(kp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (t + e < t_end) { a[e] = qp[e]; b[e] = kp[e]; }
                }
            }
            pq[j] = a;
            pk[j] = b;
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int idx = tid + j * 256;
            if (idx < HSP * (ATT_TC / 4)) {
                const int row = idx / (ATT_TC / 4), tt = 4 * (idx - row * (ATT_TC / 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {                  // pitch ATT_TCP = 66 floats: rows are not 16-byte aligned
                    ql[row * ATT_TCP + tt + e] = pq[j][e];
                    kl[row * ATT_TCP + tt + e] = pk[j][e];
                }
            }
        }
    };
    // tile pipeline: the global loads of tile i+1 are in flight while tile i is multiplied
    load_tile(t_begin);
    for (int t0 = t_begin; t0 < t_end; t0 += ATT_TC) {
        __syncthreads();                                       // previous tile fully read
        store_tile();
        __syncthreads();
        if (t0 + ATT_TC < t_end) load_tile(t0 + ATT_TC);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int tile = wave + 4 * i;
            if (tile < NT) {
                const int ib = tile / NB, jb = tile - ib * NB;
                const float* qa = ql + (ib * 16 + r16) * ATT_TCP + kk;
                const float* ka = kl + (jb * 16 + r16) * ATT_TCP + kk;
#pragma unroll
                for (int ks = 0; ks < ATT_TC; ks += 4)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[ks], ka[ks], acc[i], 0, 0, 0);
            }
        }
    }
    float* slab = slabs + ((size_t)bh * gridDim.y + s) * HSP * HSP;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + 4 * i;
        if (tile < NT) {
            const int ib = tile / NB, jb = tile - ib * NB;
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(ib * 16 + kk * 4 + r) * HSP + jb * 16 + r16] = acc[i][r];
        }
    }
}

// Phase 1 with split-bf16 products: the same partial scores on v_mfma_f32_16x16x32_bf16.  The staged
// q / k tiles are split once into bf16 hi / lo images [row][64 tokens] (row pitch 144 B: an odd multiple of 16 B, so the 16
// rows of a fragment read hit 16 distinct bank groups); a fragment is one ds_read_b128 (8 consecutive tokens of a row), a
// 16 x 16 score tile takes 3 MFMAs per 32 tokens instead of 8 f32 ones at twice the cycles.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef otp_x3x8 sx_h16x8;
typedef otp_x3x2 sx_h16x2;
constexpr int SX_PITCH = ATT_TC * 2 + 16;          // bytes per LDS row

template <int NB>
__global__ __launch_bounds__(256) void attn_scores_x3_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              float* __restrict__ slabs, int hs, int T, int chunk) {
    constexpr int HSP = NB * 16, NT = NB * NB, TPW = (NT + 3) / 4;
    constexpr int NQ = (HSP * (ATT_TC / 4) + 255) / 256;
    // (dynamic: 4 HSP SX_PITCH bytes - 64.5 KB at HSP = 112, the 102-wide heads of the 7-frame window, past the static limit)
    extern __shared__ __attribute__((aligned(16))) unsigned char sxs[];
    unsigned char *qh = sxs, *ql = sxs + HSP * SX_PITCH, *kh = sxs + 2 * HSP * SX_PITCH, *kl_ = sxs + 3 * HSP * SX_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, s = blockIdx.y;
    const float* qb = q + (size_t)bh * hs * T;
    const float* kb = k + (size_t)bh * hs * T;
    const int t_begin = s * chunk, t_end = min(T, t_begin + chunk);
    f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kk = lane >> 4;
    const bool vec = (T & 3) == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0;
    f32x4 pq[NQ], pk[NQ];
    auto load_tile = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int idx = tid + j * 256;
            const int row = idx / (ATT_TC / 4), t = t0 + 4 * (idx - row * (ATT_TC / 4));
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (row < hs && idx < HSP * (ATT_TC / 4)) {
                const float* qp = qb + (size_t)row * T + t;
                const float* kp = kb + (size_t)row * T + t;
                if (vec && t + 3 < t_end) {
                    a = *reinterpret_cast<const f32x4*>(qp);
                    b = *reinterpret_cast<const f32x4*>(kp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (t + e < t_end) { a[e] = qp[e]; b[e] = kp[e]; }
                }
            }
            pq[j] = a;
            pk[j] = b;
        }
    };
    auto split4 = [](f32x4 v, unsigned long long& hi, unsigned long long& lo) __attribute__((always_inline)) {
        uint32_t h[2], l[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x2 a = {v[2 * i], v[2 * i + 1]};
            const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, sx_h16x2));
            const f32x2 af = otp_x3_widen(hb);
            h[i] = hb;
            l[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a - af, sx_h16x2));
        }
        hi = (unsigned long long)h[0] | ((unsigned long long)h[1] << 32);
        lo = (unsigned long long)l[0] | ((unsigned long long)l[1] << 32);
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int idx = tid + j * 256;
            if (idx < HSP * (ATT_TC / 4)) {
                const int row = idx / (ATT_TC / 4), tt = 4 * (idx - row * (ATT_TC / 4));
                unsigned long long h, l;
                split4(pq[j], h, l);
                *reinterpret_cast<unsigned long long*>(qh + row * SX_PITCH + tt * 2) = h;
                *reinterpret_cast<unsigned long long*>(ql + row * SX_PITCH + tt * 2) = l;
                split4(pk[j], h, l);
                *reinterpret_cast<unsigned long long*>(kh + row * SX_PITCH + tt * 2) = h;
                *reinterpret_cast<unsigned long long*>(kl_ + row * SX_PITCH + tt * 2) = l;
            }
        }
    };
    load_tile(t_begin);
    for (int t0 = t_begin; t0 < t_end; t0 += ATT_TC) {
        __syncthreads();                                       // previous tile fully read
        store_tile();
        __syncthreads();
        if (t0 + ATT_TC < t_end) load_tile(t0 + ATT_TC);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int tile = wave + 4 * i;
            if (tile < NT) {
                const int ib = tile / NB, jb = tile - ib * NB;
                const int qo = (ib * 16 + r16) * SX_PITCH + kk * 16, ko = (jb * 16 + r16) * SX_PITCH + kk * 16;
#pragma unroll
                for (int ks = 0; ks < ATT_TC / 32; ++ks) {
                    const sx_h16x8 a_h = *reinterpret_cast<const sx_h16x8*>(qh + qo + ks * 64);
                    const sx_h16x8 a_l = *reinterpret_cast<const sx_h16x8*>(ql + qo + ks * 64);
                    const sx_h16x8 b_h = *reinterpret_cast<const sx_h16x8*>(kh + ko + ks * 64);
                    const sx_h16x8 b_l = *reinterpret_cast<const sx_h16x8*>(kl_ + ko + ks * 64);
                    acc[i] = OTP_X3_MFMA(a_l, b_h, acc[i], 0, 0, 0);
                    acc[i] = OTP_X3_MFMA(a_h, b_l, acc[i], 0, 0, 0);
                    acc[i] = OTP_X3_MFMA(a_h, b_h, acc[i], 0, 0, 0);
                }
            }
        }
    }
    float* slab = slabs + ((size_t)bh * gridDim.y + s) * HSP * HSP;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + 4 * i;
        if (tile < NT) {
            const int ib = tile / NB, jb = tile - ib * NB;
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(ib * 16 + kk * 4 + r) * HSP + jb * 16 + r16] = acc[i][r];
        }
    }
}

// Phase 2: sum the slabs, scale, row softmax (one wave per row, shuffles for max / sum).
// P (B*nh, HSP, HSP) with zero padding columns.  grid (B*nh, ceil(HSP / 4)): every wave of the chip gets a row
// (a grid of B*nh workgroups alone left 7/8 of the CUs idle for 120 us).
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float* __restrict__ slabs, float* __restrict__ P,
                                                            int hs, int HSP, int NS, float scale) {
    const int bh = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.y * 4 + wave;
    if (row >= HSP) return;
    const float* sb = slabs + (size_t)bh * NS * HSP * HSP;
    float* pb = P + (size_t)bh * HSP * HSP;
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = lane + 64 * h;
        // the NS partial slabs: independent loads, eight in flight (one after the other they were a 40 us latency chain);
        // a masked lane re-reads element 0 of its slab, same summation order as before for the live ones
        const bool live = row < hs && col < hs;
        const float* sp = sb + (live ? (size_t)row * HSP + col : 0);
        float a = 0.f;
        int s = 0;
        for (; s + 8 <= NS; s += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = sp[(size_t)(s + j) * HSP * HSP];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += t[j];
        }
        for (; s < NS; ++s) a += sp[(size_t)s * HSP * HSP];
        v[h] = live ? a * scale : -INFINITY;
    }
    const float m = wave_max(fmaxf(v[0], v[1]));
    float e0 = (row < hs && lane < hs) ? expf(v[0] - m) : 0.f;
    float e1 = (row < hs && lane + 64 < hs) ? expf(v[1] - m) : 0.f;
    const float den = wave_sum(e0 + e1);
    if (lane < HSP) pb[row * HSP + lane] = row < hs ? e0 / den : 0.f;
    if (lane + 64 < HSP) pb[row * HSP + lane + 64] = row < hs ? e1 / den : 0.f;
}

// Phase 3: O^T[t, i] = sum_j v[j, t] P[i, j], stored as out[bh][t][i] (i contiguous) - exactly the memory
// image of `out.transpose(2,3).contiguous()` (blocks.py:447).  grid (B*nh, ceil(T / 256)); each wave owns
// 64 time steps (4 row blocks) x all NB column blocks.  A = v^T from an LDS tile, B = P^T from LDS.
// time steps per workgroup: 256, 128 once HSP > 96 (the v tile must fit in LDS)
constexpr int pv_tt(int nb) { return nb > 6 ? 128 : 256; }
template <int NB>
__global__ __launch_bounds__(256) void attn_pv_kernel(const float* __restrict__ v, const float* __restrict__ P,
                                                       float* __restrict__ out, int hs, int T) {
    constexpr int HSP = NB * 16, TT = pv_tt(NB), TB = TT / 64;   // TB 16-row time blocks per wave
    constexpr int PS = HSP + 2;       // P row pitch: (i*2 + k) distinct banks
    constexpr int VS = TT + 16;    // v row pitch == 16 (mod 32)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* pl = smem;                 // [HSP][PS]
    float* vl = smem + HSP * PS;      // [HSP][VS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, t0 = blockIdx.y * TT;
    const float* vb = v + (size_t)bh * hs * T;
    const float* pb = P + (size_t)bh * HSP * HSP;
    // 16-byte vector loads, all issued before the first LDS write (one memory round trip for the whole tile)
    {
        constexpr int NP4 = (HSP * HSP / 4 + 255) / 256, NV4 = (HSP * (TT / 4) + 255) / 256;
        const bool vecv = (T & 3) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0;
        f32x4 rp[NP4], rv[NV4];
#pragma unroll
        for (int j = 0; j < NP4; ++j) {
            const int idx = tid + j * 256;
            rp[j] = idx < HSP * HSP / 4 ? reinterpret_cast<const f32x4*>(pb)[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int idx = tid + j * 256;
            const int row = idx / (TT / 4), t = t0 + 4 * (idx - row * (TT / 4));
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (row < hs && idx < HSP * (TT / 4)) {
                const float* vp = vb + (size_t)row * T + t;
                if (vecv && t + 3 < T) {
                    a = *reinterpret_cast<const f32x4*>(vp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (t + e < T) a[e] = vp[e];
                }
            }
            rv[j] = a;
        }
#pragma unroll
        for (int j = 0; j < NP4; ++j) {
            const int idx = tid + j * 256;
            if (idx < HSP * HSP / 4) {
                const int i = (4 * idx) / HSP, jj = 4 * idx - i * HSP;    // HSP % 4 == 0: a float4 stays in one row
#pragma unroll
                for (int e = 0; e < 4; ++e) pl[i * PS + jj + e] = rp[j][e];
            }
        }
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int idx = tid + j * 256;
            if (idx < HSP * (TT / 4)) {
                const int row = idx / (TT / 4), tt = 4 * (idx - row * (TT / 4));
                *reinterpret_cast<f32x4*>(vl + row * VS + tt) = rv[j];
            }
        }
    }
    __syncthreads();
    const int r16 = lane & 15, kk = lane >> 4;
    f32x4 acc[TB][NB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) acc[tb][ib] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* va = vl + kk * VS + wave * (TT / 4) + r16;          // A[t][j]: lane (t = r16, k = kk)
    const float* pa = pl + r16 * PS + kk;                      // B[j][i]: lane (k = kk, i = r16)
    for (int j0 = 0; j0 < HSP; j0 += 4) {
        float a[TB], b[NB];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) a[tb] = va[j0 * VS + tb * 16];
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) b[ib] = pa[ib * 16 * PS + j0];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
                acc[tb][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tb], b[ib], acc[tb][ib], 0, 0, 0);
    }
    float* ob = out + (size_t)bh * T * hs;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + wave * (TT / 4) + tb * 16 + kk * 4 + r;
            if (t < T) {
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) {
                    const int i = ib * 16 + r16;
                    if (i < hs) ob[(size_t)t * hs + i] = acc[tb][ib][r];
                }
            }
        }
}

// Phase 3 with split-bf16 products: O^T[t, i] = sum_j v[j, t] P[i, j] on v_mfma_f32_16x16x32_bf16 (M = 256 tokens per
// workgroup, N = i, K = j).  Both operands want 8 consecutive j per lane: P rows have them; the v tile is transposed while it
// is staged (a thread loads 8 rows j x 4 tokens and writes four per-token records of 8 j each, hi and lo), like the window
// staging of csrc/convx.hip.  Records: [HSP j hi | HSP j lo | 16 B pad] (an odd multiple of 16 bytes).
// Eight waves own 256 tokens and ONE workgroup owns its CU: the launch asks for PVX_LDS bytes of LDS whatever the records need.
// Sharing a CU with a workgroup of this kernel changed results of the LDS-DMA fed projection kernel (csrc/densex.hip: one
// accumulator row of one 16-token tile, a few times per launch, only while the two temporal encoders ran side by side - found
// as replay-to-replay differences of the batch-16 forward, tools/dbg notes in DESIGN.md section 4); with the CU to itself
// neither kernel's results depend on what else is running.
// PVX_WAVES: 8 up to HSP = 80; 4 for the 96 / 112-wide heads of the 7-frame window (the records of 256 tokens would not fit)
constexpr size_t PVX_LDS = 138 * 1024;
constexpr int pvx_waves(int NB) { return NB <= 5 ? 8 : 4; }
constexpr size_t pvx_lds(int NB) {
    return NB <= 5 ? PVX_LDS : (size_t)(32 * pvx_waves(NB) + NB * 16) * (NB * 16 * 4 + 16) + 16;
}
template <int NB>
__global__ __launch_bounds__(64 * pvx_waves(NB), 2) void attn_pv_x3_kernel(const float* __restrict__ v, const float* __restrict__ P,
                                                                      float* __restrict__ out, int hs, int T, unsigned* rflag) {
    constexpr int PVX_WAVES = pvx_waves(NB), PVX_TH = 64 * PVX_WAVES, PVX_TT = 32 * PVX_WAVES;
    constexpr int HSP = NB * 16, KS = (HSP + 31) / 32, G = HSP / 8;
    constexpr int REC = HSP * 4 + 16;                              // bytes of a token (v) / row (P) record
    constexpr int NVI = (G * (PVX_TT / 4) + PVX_TH - 1) / PVX_TH;            // v items (8 rows x 4 tokens) per thread
    constexpr int NP4 = (HSP * HSP / 4 + PVX_TH - 1) / PVX_TH;
    extern __shared__ __attribute__((aligned(16))) unsigned char pvs[];
    unsigned char* vrec = pvs;                                     // [PVX_TT][REC]
    unsigned char* prec = pvs + PVX_TT * REC;                      // [HSP][REC]
    unsigned char* zrec = prec + HSP * REC;                        // 16 zero bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, t0 = blockIdx.y * PVX_TT;
    const float* vb = v + (size_t)bh * hs * T;
    const float* pb = P + (size_t)bh * HSP * HSP;
    auto split4 = [](f32x4 x, unsigned long long& hi, unsigned long long& lo) __attribute__((always_inline)) {
        uint32_t h[2], l[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x2 a = {x[2 * i], x[2 * i + 1]};
            const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, sx_h16x2));
            const f32x2 af = otp_x3_widen(hb);
            h[i] = hb;
            l[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a - af, sx_h16x2));
        }
        hi = (unsigned long long)h[0] | ((unsigned long long)h[1] << 32);
        lo = (unsigned long long)l[0] | ((unsigned long long)l[1] << 32);
    };
    if (tid < 4) reinterpret_cast<uint32_t*>(zrec)[tid] = 0u;
    // P (HSP x HSP fp32, zero padded) -> rows of bf16 hi / lo
    {
        f32x4 rp[NP4];
#pragma unroll
        for (int j = 0; j < NP4; ++j) {
            const int idx = tid + j * PVX_TH;
            rp[j] = idx < HSP * HSP / 4 ? reinterpret_cast<const f32x4*>(pb)[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NP4; ++j) {
            const int idx = tid + j * PVX_TH;
            if (idx < HSP * HSP / 4) {
                const int i = (4 * idx) / HSP, jj = 4 * idx - i * HSP;
                unsigned long long h, l;
                split4(rp[j], h, l);
                *reinterpret_cast<unsigned long long*>(prec + i * REC + jj * 2) = h;
                *reinterpret_cast<unsigned long long*>(prec + i * REC + HSP * 2 + jj * 2) = l;
            }
        }
    }
    // v tile, transposed: item = (8-row group g, 4 tokens)
    const bool vecv = (T & 3) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0;
#pragma unroll
    for (int it = 0; it < NVI; ++it) {
        const int idx = tid + it * PVX_TH;
        const int g = idx / (PVX_TT / 4), f4 = idx - g * (PVX_TT / 4);
        const int t = t0 + 4 * f4;
        f32x4 x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = 8 * g + e;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (idx < G * (PVX_TT / 4) && row < hs) {
                const float* vp = vb + (size_t)row * T + t;
                if (vecv && t + 3 < T) {
                    a = *reinterpret_cast<const f32x4*>(vp);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (t + c < T) a[c] = vp[c];
                }
            }
            x[e] = a;
        }
        if (idx < G * (PVX_TT / 4)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned long long h0, l0, h1, l1;
                split4((f32x4){x[0][c], x[1][c], x[2][c], x[3][c]}, h0, l0);
                split4((f32x4){x[4][c], x[5][c], x[6][c], x[7][c]}, h1, l1);
                unsigned char* r = vrec + (4 * f4 + c) * REC + g * 16;
                *reinterpret_cast<unsigned long long*>(r) = h0;
                *reinterpret_cast<unsigned long long*>(r + 8) = h1;
                *reinterpret_cast<unsigned long long*>(r + HSP * 2) = l0;
                *reinterpret_cast<unsigned long long*>(r + HSP * 2 + 8) = l1;
            }
        }
    }
    __syncthreads();
    const int r16 = lane & 15, kk = lane >> 4;
    f32x4 acc[2][NB];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NB; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int kg = 4 * ks + kk;                                // 8-wide j group of this lane
        const bool kv = kg < G;
        sx_h16x8 ah[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const unsigned char* a = kv ? vrec + ((wave * 2 + mt) * 16 + r16) * REC + kg * 16 : zrec;
            ah[mt] = *reinterpret_cast<const sx_h16x8*>(a);
            al[mt] = *reinterpret_cast<const sx_h16x8*>(kv ? a + HSP * 2 : zrec);
        }
#pragma unroll
        for (int nt = 0; nt < NB; ++nt) {
            const unsigned char* b = kv ? prec + (nt * 16 + r16) * REC + kg * 16 : zrec;
            const sx_h16x8 b_h = *reinterpret_cast<const sx_h16x8*>(b);
            const sx_h16x8 b_l = *reinterpret_cast<const sx_h16x8*>(kv ? b + HSP * 2 : zrec);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt][nt] = OTP_X3_MFMA(al[mt], b_h, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = OTP_X3_MFMA(ah[mt], b_l, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = OTP_X3_MFMA(ah[mt], b_h, acc[mt][nt], 0, 0, 0);
            }
        }
    }
    {   // range guard (common.h): q, k or v beyond a half's range has made these sums NaN (through the scores and the softmax)
        bool bad = false;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NB; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bad |= otp_out_of_range(acc[mt][nt][r]);
        otp_range_report(rflag, bad, OTP_RANGE_ATTN);
    }
    float* ob = out + (size_t)bh * T * hs;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + (wave * 2 + mt) * 16 + kk * 4 + r;
            if (t < T) {
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) {
                    const int i = nt * 16 + r16;
                    if (i < hs) ob[(size_t)t * hs + i] = acc[mt][nt][r];
                }
            }
        }
}

// ---- nn.Upsample(scale_factor = f, mode = 'linear', align_corners = False) on (B, C, T) ------------
__global__ void upsample_linear_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int T, int f,
                                       int out_ctot, int out_coff) {
    const int To = T * f;
    const int to = blockIdx.x * blockDim.x + threadIdx.x;
    if (to >= To) return;
    const int c = blockIdx.y % C, b = blockIdx.y / C;
    const float* xr = x + ((size_t)b * C + c) * T;
    float r;
    if (f == 1) {
        r = xr[to];
    } else {
        float src = ((float)to + 0.5f) * (1.f / (float)f) - 0.5f;
        src = src < 0.f ? 0.f : src;
        int i0 = (int)src;
        int i1 = i0 + (i0 < T - 1 ? 1 : 0);
        float l1 = src - (float)i0, l0 = 1.f - l1;
        r = l0 * xr[i0] + l1 * xr[i1];
    }
    out[((size_t)b * out_ctot + out_coff + c) * To + to] = r;
}

// four consecutive outputs per thread, one 16-byte store (rows of To = T f floats, To % 4 == 0, 16-byte aligned rows): the
// scalar form above moves the 60 MB pyramid levels of a temporal encoder at 2 TB/s
__global__ void upsample_linear4_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int T, int f,
                                        int out_ctot, int out_coff) {
    const int To = T * f;
    const int t4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (t4 >= To) return;
    const int c = blockIdx.y % C, b = blockIdx.y / C;
    const float* xr = x + ((size_t)b * C + c) * T;
    f32x4 r;
    if (f == 1) {
        r = *reinterpret_cast<const f32x4*>(xr + t4);
    } else {
        const float inv = 1.f / (float)f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float src = ((float)(t4 + k) + 0.5f) * inv - 0.5f;
            src = src < 0.f ? 0.f : src;
            const int i0 = (int)src;
            const int i1 = i0 + (i0 < T - 1 ? 1 : 0);
            const float l1 = src - (float)i0, l0 = 1.f - l1;
            r[k] = l0 * xr[i0] + l1 * xr[i1];
        }
    }
    *reinterpret_cast<f32x4*>(out + ((size_t)b * out_ctot + out_coff + c) * To + t4) = r;
}

}  // namespace

#ifndef OTP_X3_GRAD_COPY
extern "C" int otp_ln_channel(const void* x, const void* gamma, const void* beta, void* y, void* pool, int B,
                              int C, int T, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || B <= 0 || C <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    auto st = static_cast<hipStream_t>(stream);
    auto xf = static_cast<const float*>(x);
    auto gf = static_cast<const float*>(gamma);
    auto bf = static_cast<const float*>(beta);
    auto yf = static_cast<float*>(y);
    dim3 grid(otp_ceil_div(T, 256), B);
    if (C <= 17) hipLaunchKernelGGL(ln_channel_kernel<17>, grid, dim3(256), 0, st, xf, gf, bf, yf, C, T, eps);
    else if (C <= 136)
        hipLaunchKernelGGL(ln_channel_split_kernel<34>, dim3(otp_ceil_div(T, 64), B), dim3(256), 0, st, xf, gf, bf, yf, C, T, eps);
    else if (C <= 204)                                   // 12 x 17 stacked maps: the 7-frame window of BASELINE configs[4]
        hipLaunchKernelGGL(ln_channel_split_kernel<51>, dim3(otp_ceil_div(T, 64), B), dim3(256), 0, st, xf, gf, bf, yf, C, T, eps);
    else hipLaunchKernelGGL(ln_channel_kernel<0>, grid, dim3(256), 0, st, xf, gf, bf, yf, C, T, eps);
    if (pool) {
        int To = (T + 2 - 3) / 2 + 1;
        hipLaunchKernelGGL(maxpool3s2_kernel, dim3(otp_ceil_div(To, 256), B * C), dim3(256), 0, st, xf,
                           static_cast<float*>(pool), T, To);
    }
    return otp_launch_status();
}

extern "C" int otp_dwconv_ln3(const void* x, const void* dwq, const void* dwk, const void* dwv, const void* gq,
                              const void* bq, const void* gk, const void* bk, const void* gv, const void* bv,
                              void* q, void* k, void* v, int B, int C, int T, int stride, float eps, void* stream) {
    if (!x || !dwq || !dwk || !dwv || !gq || !bq || !gk || !bk || !gv || !bv || !q || !k || !v) return OTP_ERR_BAD_ARG;
    if (B <= 0 || C <= 0 || T <= 0 || stride <= 0) return OTP_ERR_BAD_ARG;
    const int To = (T + 2 - 3) / stride + 1;
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    auto st = static_cast<hipStream_t>(stream);
#define OTP_DW_ARGS f(x), f(dwq), f(dwk), f(dwv), f(gq), f(bq), f(gk), f(bk), f(gv), f(bv), static_cast<float*>(q), \
                    static_cast<float*>(k), static_cast<float*>(v), C, T, To, stride, eps
    if (C <= 4 * 5)
        hipLaunchKernelGGL(dwconv_ln3_kernel<5>, dim3(otp_ceil_div(To, 64), B), dim3(256), 0, st, OTP_DW_ARGS);
    else if (C <= 4 * 34)
        hipLaunchKernelGGL(dwconv_ln3_kernel<34>, dim3(otp_ceil_div(To, 64), B), dim3(256), 0, st, OTP_DW_ARGS);
    else if (C <= 4 * 51)                                          // C = 204: the 7-frame window (the generic kernel took 370-450 us)
        hipLaunchKernelGGL(dwconv_ln3_kernel<51>, dim3(otp_ceil_div(To, 64), B), dim3(256), 0, st, OTP_DW_ARGS);
    else
        hipLaunchKernelGGL(dwconv_ln3_generic_kernel, dim3(otp_ceil_div(To, 256), B), dim3(256), 0, st, OTP_DW_ARGS);
#undef OTP_DW_ARGS
    return otp_launch_status();
}
#endif  // OTP_X3_GRAD_COPY

// 1: score products as split products (default), 0: f32 MFMA (otp_chan_attn_set_split); one flag for both copies of this file
#ifndef OTP_X3_GRAD_COPY
std::atomic<int> otp_g_attn_split{1};
#else
extern std::atomic<int> otp_g_attn_split;
#endif
#define g_attn_split otp_g_attn_split

namespace {
int attn_splits(int BH, int T) {
    int ns = 1;
    while (BH * ns < 768 && T / (ns * 2) >= 2 * ATT_TC) ns *= 2;
    return ns;
}
}  // namespace

#ifndef OTP_X3_GRAD_COPY
extern "C" size_t otp_chan_attn_workspace(int B, int C, int T, int n_head) {
    if (B <= 0 || C <= 0 || T <= 0 || n_head <= 0 || C % n_head) return 0;
    const int hs = C / n_head, HSP = (hs + 15) & ~15;
    const size_t BH = (size_t)B * n_head;
    return (BH * attn_splits((int)BH, T) + BH) * HSP * HSP * sizeof(float);
}

extern "C" int otp_chan_attn(const void* q, const void* k, const void* v, void* out, void* workspace,
                             size_t workspace_bytes, int B, int C, int T, int n_head, float scale, void* stream) {
    if (!q || !k || !v || !out || !workspace || B <= 0 || C <= 0 || T <= 0 || n_head <= 0) return OTP_ERR_BAD_ARG;
    if (C % n_head) return OTP_ERR_BAD_ARG;
    const int hs = C / n_head, HSP = (hs + 15) & ~15, NB = HSP / 16;
    if (NB > 7) return OTP_ERR_UNSUPPORTED;                              // hs <= 112
    if (workspace_bytes < otp_chan_attn_workspace(B, C, T, n_head)) return OTP_ERR_WORKSPACE;
    const int BH = B * n_head, NS = attn_splits(BH, T);
    const int chunk = otp_ceil_div(otp_ceil_div(T, NS), ATT_TC) * ATT_TC;
    auto st = static_cast<hipStream_t>(stream);
    float* slabs = static_cast<float*>(workspace);
    float* P = slabs + (size_t)BH * NS * HSP * HSP;
    auto qf = static_cast<const float*>(q);
    auto kf = static_cast<const float*>(k);
    auto vf = static_cast<const float*>(v);
    auto of = static_cast<float*>(out);
    const int TT = pv_tt(NB);
    const size_t pv_lds = ((size_t)HSP * (HSP + 2) + (size_t)HSP * (TT + 16)) * sizeof(float);
    dim3 g1(BH, NS), g3(BH, otp_ceil_div(T, TT));
#define OTP_ATT(NB_)                                                                                           \
    {                                                                                                          \
        bool split_ = false;                                                                                   \
        if (g_attn_split.load(std::memory_order_relaxed)) {                                                    \
            split_ = true;                                                                                     \
            auto ks_ = attn_scores_x3_kernel<NB_>;                                                             \
            const size_t ls_ = (size_t)4 * NB_ * 16 * SX_PITCH;                                                \
            OTP_ALLOW_BIG_LDS(ks_, ls_);                                                                       \
            hipLaunchKernelGGL(ks_, g1, dim3(256), ls_, st, qf, kf, slabs, hs, T, chunk);                      \
        }                                                                                                      \
        if (!split_) hipLaunchKernelGGL(attn_scores_kernel<NB_>, g1, dim3(256), 0, st, qf, kf, slabs, hs, T, chunk); \
        hipLaunchKernelGGL(attn_softmax_kernel, dim3(BH, otp_ceil_div(HSP, 4)), dim3(256), 0, st, slabs, P, hs, HSP, NS, scale);      \
        if (split_) {                                                                                          \
            auto kx = attn_pv_x3_kernel<NB_>;                                                                  \
            const size_t lx = pvx_lds(NB_);                 /* >= (tokens + HSP) * (HSP * 4 + 16) + 16 */                                                   \
            OTP_ALLOW_BIG_LDS(kx, lx);                                                                         \
            hipLaunchKernelGGL(kx, dim3(BH, otp_ceil_div(T, 32 * pvx_waves(NB_))), dim3(64 * pvx_waves(NB_)), lx, st, vf, P, of, hs, T, otp_range_word()); \
        } else {                                                                                               \
            auto kern = attn_pv_kernel<NB_>;                                                                   \
            OTP_ALLOW_BIG_LDS(kern, pv_lds);                                                                   \
            hipLaunchKernelGGL(kern, g3, dim3(256), pv_lds, st, vf, P, of, hs, T);                             \
        }                                                                                                      \
    }
    switch (NB) {
        case 1: OTP_ATT(1) break;
        case 2: OTP_ATT(2) break;
        case 3: OTP_ATT(3) break;
        case 4: OTP_ATT(4) break;
        case 5: OTP_ATT(5) break;
        case 6: OTP_ATT(6) break;
        default: OTP_ATT(7) break;
    }
#undef OTP_ATT
    return otp_launch_status();
}

// ---- pieces of the channel attention exposed for its backward (otpose_amd/train_ops.py) -------------------------
// number of T-splits / score slabs the kernels use for (BH, T), and where otp_chan_attn leaves P inside its workspace
// 1 (default): q.k^T products as split bf16 on the bf16 matrix cores; 0: the f32 MFMA kernel.  Process-wide.
extern "C" int otp_chan_attn_set_split(int on) {
    g_attn_split.store(on ? 1 : 0, std::memory_order_relaxed);
    return OTP_OK;
}

extern "C" int otp_chan_attn_splits(int BH, int T) { return (BH > 0 && T > 0) ? attn_splits(BH, T) : 0; }
#endif  // OTP_X3_GRAD_COPY

// slabs[bh][s] (HSP x HSP, zero padded) = partial a . b^T over the s-th slice of T, no scale: sum the slabs for a . b^T
extern "C" int OTP_ENTRY(otp_chan_attn_scores)(const void* a, const void* b, void* slabs, int BH, int hs, int T, void* stream) {
    if (!a || !b || !slabs || BH <= 0 || hs <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    const int HSP = (hs + 15) & ~15, NB = HSP / 16;
    if (NB > 7) return OTP_ERR_UNSUPPORTED;
    const int NS = attn_splits(BH, T);
    const int chunk = otp_ceil_div(otp_ceil_div(T, NS), ATT_TC) * ATT_TC;
    auto st = static_cast<hipStream_t>(stream);
    dim3 g1(BH, NS);
    auto af = static_cast<const float*>(a);
    auto bf = static_cast<const float*>(b);
    auto sf = static_cast<float*>(slabs);
#define OTP_SC(NB_)                                                                                       \
    {                                                                                                     \
        bool split_ = false;                                                                              \
        if (g_attn_split.load(std::memory_order_relaxed)) {                                               \
            split_ = true;                                                                                \
            auto ks_ = attn_scores_x3_kernel<NB_>;                                                        \
            const size_t ls_ = (size_t)4 * NB_ * 16 * SX_PITCH;                                           \
            OTP_ALLOW_BIG_LDS(ks_, ls_);                                                                  \
            hipLaunchKernelGGL(ks_, g1, dim3(256), ls_, st, af, bf, sf, hs, T, chunk);                    \
        }                                                                                                 \
        if (!split_) hipLaunchKernelGGL(attn_scores_kernel<NB_>, g1, dim3(256), 0, st, af, bf, sf, hs, T, chunk); \
    }
    switch (NB) {
        case 1: OTP_SC(1) break;
        case 2: OTP_SC(2) break;
        case 3: OTP_SC(3) break;
        case 4: OTP_SC(4) break;
        case 5: OTP_SC(5) break;
        case 6: OTP_SC(6) break;
        default: OTP_SC(7) break;
    }
#undef OTP_SC
    return otp_launch_status();
}

// out[bh][t][i] = sum_j M[bh][i][j] * v[bh][j][t]   (M: HSP x HSP zero padded; the transposed-contiguous output image)
extern "C" int OTP_ENTRY(otp_chan_attn_apply)(const void* v, const void* M, void* out, int BH, int hs, int T, void* stream) {
    if (!v || !M || !out || BH <= 0 || hs <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    const int HSP = (hs + 15) & ~15, NB = HSP / 16;
    if (NB > 7) return OTP_ERR_UNSUPPORTED;
    auto st = static_cast<hipStream_t>(stream);
    const int TT = pv_tt(NB);
    const size_t pv_lds = ((size_t)HSP * (HSP + 2) + (size_t)HSP * (TT + 16)) * sizeof(float);
    dim3 g3(BH, otp_ceil_div(T, TT));
    auto vf = static_cast<const float*>(v);
    auto mf = static_cast<const float*>(M);
    auto of = static_cast<float*>(out);
#define OTP_APPLY(NB_)                                                                                      \
    {                                                                                                       \
        bool split_ = false;                                                                                \
        if (g_attn_split.load(std::memory_order_relaxed)) {                                                 \
            split_ = true;                                                                                  \
            auto kx = attn_pv_x3_kernel<NB_>;                                                               \
            const size_t lx = pvx_lds(NB_);                                                                 \
            OTP_ALLOW_BIG_LDS(kx, lx);                                                                      \
            hipLaunchKernelGGL(kx, dim3(BH, otp_ceil_div(T, 32 * pvx_waves(NB_))), dim3(64 * pvx_waves(NB_)), lx, st, vf, mf, of, hs, T, otp_range_word()); \
        }                                                                                                   \
        if (!split_) {                                                                                      \
            auto kern = attn_pv_kernel<NB_>;                                                                \
            OTP_ALLOW_BIG_LDS(kern, pv_lds);                                                                \
            hipLaunchKernelGGL(kern, g3, dim3(256), pv_lds, st, vf, mf, of, hs, T);                         \
        }                                                                                                   \
    }
    switch (NB) {
        case 1: OTP_APPLY(1) break;
        case 2: OTP_APPLY(2) break;
        case 3: OTP_APPLY(3) break;
        case 4: OTP_APPLY(4) break;
        case 5: OTP_APPLY(5) break;
        case 6: OTP_APPLY(6) break;
        default: OTP_APPLY(7) break;
    }
#undef OTP_APPLY
    return otp_launch_status();
}

#ifndef OTP_X3_GRAD_COPY
extern "C" int otp_maxpool3s2_forward(const void* x, void* y, int rows, int T, void* stream) {
    if (!x || !y || rows <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    const int To = (T + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(otp_ceil_div(To, 256), rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(x), static_cast<float*>(y), T, To);
    return otp_launch_status();
}

extern "C" int otp_upsample_linear(const void* x, void* out, int B, int C, int T, int f, int out_ctot, int out_coff,
                                   void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || T <= 0 || f <= 0 || out_ctot < out_coff + C) return OTP_ERR_BAD_ARG;
    const bool vec = (T * f) % 4 == 0 && (f > 1 || T % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(upsample_linear4_kernel, dim3(otp_ceil_div(T * f / 4, 256), B * C), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<const float*>(x), static_cast<float*>(out), C, T,
                           f, out_ctot, out_coff);
    else
        hipLaunchKernelGGL(upsample_linear_kernel, dim3(otp_ceil_div(T * f, 256), B * C), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<const float*>(x), static_cast<float*>(out), C, T,
                           f, out_ctot, out_coff);
    return otp_launch_status();
}
#endif  // OTP_X3_GRAD_COPY
