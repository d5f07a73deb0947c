// Backward kernels of the ConvTransformer pieces (training step; forward counterparts in transformer.hip):
// channel LayerNorm, depthwise k=3 conv (forward + both gradients), exact-erf GELU, MaxPool1d(3,2,1), linear
// up-sampling, and the helpers the channel-attention backward is assembled from (layout transpose, slab sum,
// softmax backward).  Reference: model/blocks.py:95-110, 234-254, 359-381, 400-453; ConvVideoTransformer.py:108.
// All tensors (B, C, T) fp32 with T contiguous; a thread owns one time step, so wave accesses are 256-byte rows.
#include "common.h"

namespace {

// ---- channel LayerNorm backward ------------------------------------------------------------------------
// dx = r * (dyg - mean_c(dyg) - xhat * mean_c(dyg * xhat)), dyg = dy * gamma;  dyxh = dy * xhat is written out so that
// dgamma = channel_sum(dyxh), dbeta = channel_sum(dy) (otp_channel_sum)
__global__ __launch_bounds__(256) void ln_channel_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ gamma, float* __restrict__ dx,
                                                              float* __restrict__ dyxh, int C, int T, float eps) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const size_t base = (size_t)blockIdx.y * C * T + t;
    const float inv_c = 1.f / (float)C;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += x[base + (size_t)c * T];
    const float mu = s * inv_c;
    float q = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = x[base + (size_t)c * T] - mu;
        q += d * d;
    }
    const float r = 1.f / sqrtf(q * inv_c + eps);
    float m1 = 0.f, m2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const float xh = (x[base + (size_t)c * T] - mu) * r, g = dy[base + (size_t)c * T] * gamma[c];
        m1 += g;
        m2 += g * xh;
    }
    m1 *= inv_c;
    m2 *= inv_c;
    for (int c = 0; c < C; ++c) {
        const float xh = (x[base + (size_t)c * T] - mu) * r, d = dy[base + (size_t)c * T];
        dx[base + (size_t)c * T] = r * (d * gamma[c] - m1 - xh * m2);
        dyxh[base + (size_t)c * T] = d * xh;
    }
}

// The same for C <= 4 * CW with every value read once: a workgroup owns 64 tokens, wave w keeps channels [w * cw, (w+1) * cw)
// of x and dy in registers (cw = ceil(C / 4)), the three per-token reductions (mean, variance, the two dy moments) cross the
// waves through LDS.  One HBM read of x and dy, one write of dx and dy * xhat (the one-thread-per-token form above re-reads
// its 136-channel columns four times out of L2: 145 us -> the traffic bound at (B, 136, 6912) is ~75 us).
// SUMS: instead of writing dy * xhat for a later channel_sum, the workgroup reduces sum_t dy * xhat and sum_t dy of its 64
// tokens per channel (wavefront shuffles) and stores them as partials[2][C][workgroups]; ln_param_reduce_kernel folds them
// into dgamma / dbeta - saves the dy * xhat round trip and two more passes over dy per LayerNorm.
template <int CW, bool SUMS>
__global__ __launch_bounds__(256) void ln_channel_bwd_split_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    const float* __restrict__ gamma, float* __restrict__ dx,
                                                                    float* __restrict__ dyxh, int C, int T, float eps) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const size_t base = (size_t)blockIdx.y * C * T + (live ? t : T - 1);
    const float inv_c = 1.f / (float)C;
    const int cw = (C + 3) / 4, cbeg = wave * cw;
    float xv[CW], dv[CW];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        // unconditional loads (channel index clamped, value masked afterwards): a branch around each load makes hipcc wait
        // for every outstanding load at every branch, i.e. 2 * CW dependent HBM round trips instead of one batch
        const int c = cbeg + i;
        const bool ok = i < cw && c < C;
        const size_t o = base + (size_t)(ok ? c : C - 1) * T;
        const float xl = x[o], dl = dy[o];
        xv[i] = ok ? xl : 0.f;
        dv[i] = ok ? dl : 0.f;
        s += xv[i];
    }
    red[0][wave][lane] = s;
    __syncthreads();
    const float mu = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) * inv_c;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) {
            xv[i] -= mu;
            q += xv[i] * xv[i];
        }
    }
    red[0][wave][lane] = q;
    __syncthreads();
    const float r = 1.f / sqrtf((red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) * inv_c + eps);
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) {
            xv[i] *= r;                                   // xhat
            const float g = dv[i] * gamma[c];
            m1 += g;
            m2 += g * xv[i];
        }
    }
    red[0][wave][lane] = m1;
    red[1][wave][lane] = m2;
    __syncthreads();
    m1 = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) * inv_c;
    m2 = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) * inv_c;
    if (SUMS) {
        const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            const int c = cbeg + i;
            if (i < cw && c < C) {                       // (wave-uniform condition)
                const float pg = wave_sum(live ? dv[i] * xv[i] : 0.f), pb = wave_sum(live ? dv[i] : 0.f);
                if (lane == 0) {
                    dyxh[(size_t)c * nwg + wg] = pg;      // here `dyxh` is the partial-sum workspace [2][C][nwg]
                    dyxh[((size_t)C + c) * nwg + wg] = pb;
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) {
            dx[base + (size_t)c * T] = r * (dv[i] * gamma[c] - m1 - xv[i] * m2);
            if (!SUMS) dyxh[base + (size_t)c * T] = dv[i] * xv[i];
        }
    }
}

// Wide form of the kernel above for T % 4 == 0: a workgroup of 8 waves owns 256 consecutive tokens, a lane 4 of them
// (16-byte accesses: every wave instruction moves 1 KB of one channel row instead of 256 B - the 64-token form spread its
// traffic over 136 rows in 256-byte pieces and reached 1.6 TB/s), wave w keeps channels [w*CW, (w+1)*CW) of x and dy in
// registers (CW = ceil(C / 8)).  Same arithmetic, same partial-sum layout [2][C][workgroups].
typedef float lnf4 __attribute__((ext_vector_type(4)));
template <int CW>
__global__ __launch_bounds__(512) void ln_channel_bwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const float* __restrict__ gamma, float* __restrict__ dx,
                                                                   float* __restrict__ part, int C, int T, float eps) {
    __shared__ lnf4 red[2][8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t4 = blockIdx.x * 64 + lane, T4 = T >> 2;
    const bool live = t4 < T4;
    const size_t base = ((size_t)blockIdx.y * C * T4) + (live ? t4 : T4 - 1);      // in float4 units
    const lnf4* x4 = reinterpret_cast<const lnf4*>(x);
    const lnf4* d4 = reinterpret_cast<const lnf4*>(dy);
    const float inv_c = 1.f / (float)C;
    const int cw = (C + 7) / 8, cbeg = wave * cw;
    lnf4 xv[CW], dv[CW];
    lnf4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        const bool ok = i < cw && c < C;
        const size_t o = base + (size_t)(ok ? c : C - 1) * T4;
        const lnf4 xl = x4[o], dl = d4[o];
        xv[i] = ok ? xl : lnf4{0.f, 0.f, 0.f, 0.f};
        dv[i] = ok ? dl : lnf4{0.f, 0.f, 0.f, 0.f};
        s += xv[i];
    }
    auto all_waves = [&](int slot, lnf4 v) {
        red[slot][wave][lane] = v;
        __syncthreads();
        lnf4 r = red[slot][0][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) r += red[slot][w][lane];
        __syncthreads();
        return r;
    };
    const lnf4 mu = all_waves(0, s) * inv_c;
    lnf4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const bool ok = i < cw && cbeg + i < C;
        if (ok) {
            xv[i] -= mu;
            q += xv[i] * xv[i];
        }
    }
    const lnf4 var = all_waves(0, q) * inv_c;
    lnf4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = 1.f / sqrtf(var[j] + eps);
    lnf4 m1 = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) {
            xv[i] *= r;                                          // xhat
            const lnf4 g = dv[i] * gamma[c];
            m1 += g;
            m2 += g * xv[i];
        }
    }
    red[0][wave][lane] = m1;
    red[1][wave][lane] = m2;
    __syncthreads();
    m1 = red[0][0][lane];
    m2 = red[1][0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) {
        m1 += red[0][w][lane];
        m2 += red[1][w][lane];
    }
    m1 *= inv_c;
    m2 *= inv_c;
    const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) {                                   // (wave-uniform condition)
            const lnf4 a = dv[i] * xv[i];
            const float pg = wave_sum(live ? (a[0] + a[1]) + (a[2] + a[3]) : 0.f);
            const float pb = wave_sum(live ? (dv[i][0] + dv[i][1]) + (dv[i][2] + dv[i][3]) : 0.f);
            if (lane == 0) {
                part[(size_t)c * nwg + wg] = pg;
                part[((size_t)C + c) * nwg + wg] = pb;
            }
        }
    }
    if (!live) return;
    lnf4* o4 = reinterpret_cast<lnf4*>(dx);
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int c = cbeg + i;
        if (i < cw && c < C) o4[base + (size_t)c * T4] = r * (dv[i] * gamma[c] - m1 - xv[i] * m2);
    }
}

// grid (2 * C): out[k] = sum over the nwg partials of row k (fp64 accumulation, fixed order)
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int C, int nwg) {
    __shared__ double red[256];
    const float* src = part + (size_t)blockIdx.x * nwg;
    double s = 0.0;
    for (int i = threadIdx.x; i < nwg; i += 256) s += (double)src[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if ((int)blockIdx.x < C) dgamma[blockIdx.x] = (float)red[0];
        else dbeta[blockIdx.x - C] = (float)red[0];
    }
}

// ---- depthwise conv, k = 3, pad 1, stride s, no bias ------------------------------------------------------
__global__ void dwconv3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int C,
                                   int T, int To, int stride) {
    const int to = blockIdx.x * blockDim.x + threadIdx.x;
    if (to >= To) return;
    const int c = blockIdx.y % C;
    const float* xr = x + (size_t)blockIdx.y * T;
    const int t0 = to * stride - 1;
    const float a = t0 >= 0 ? xr[t0] : 0.f, b = xr[t0 + 1], cc = t0 + 2 < T ? xr[t0 + 2] : 0.f;
    y[(size_t)blockIdx.y * To + to] = w[c * 3] * a + w[c * 3 + 1] * b + w[c * 3 + 2] * cc;
}

// dx[t] = sum_k w[k] * dy[to] over the outputs with to*s - 1 + k == t
__global__ void dwconv3_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                     int C, int T, int To, int stride) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int c = blockIdx.y % C;
    const float* dr = dy + (size_t)blockIdx.y * To;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int num = t + 1 - k;                 // to * stride
        if (num >= 0 && num % stride == 0 && num / stride < To) s += w[c * 3 + k] * dr[num / stride];
    }
    dx[(size_t)blockIdx.y * T + t] = s;
}

// stride 1, rows of T % 4 == 0 floats (every block of the temporal encoders but the two strided ones): four consecutive time steps per
// thread - one float4 of dy and its two neighbours in, one float4 out; dx[t] = w0 dy[t + 1] + w1 dy[t] + w2 dy[t - 1].  The
// one-element kernel above pays two integer divisions per tap and ran at 2.2 TB/s (54 us at 16 x 136 x 6912).
__global__ __launch_bounds__(256) void dwconv3_bwd_x_s1_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                float* __restrict__ dx, int C, int T, size_t groups) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int q4 = T >> 2;
    for (size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gidx < groups; gidx += (size_t)gridDim.x * blockDim.x) {
        const size_t row = gidx / q4;
        const int q = (int)(gidx - row * q4), c = (int)(row % C);
        const float* dr = dy + row * T + 4 * q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(dr);
        const float lft = q > 0 ? dr[-1] : 0.f, rgt = q + 1 < q4 ? dr[4] : 0.f;
        const float w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];
        f32x4 o;
        o[0] = w0 * v[1] + w1 * v[0] + w2 * lft;
        o[1] = w0 * v[2] + w1 * v[1] + w2 * v[0];
        o[2] = w0 * v[3] + w1 * v[2] + w2 * v[1];
        o[3] = w0 * rgt + w1 * v[3] + w2 * v[2];
        *reinterpret_cast<f32x4*>(dx + row * T + 4 * q) = o;
    }
}

// dw[c][k] += sum_{b, to} dy[b,c,to] * x[b,c,to*s-1+k]; grid (C): ONE workgroup of 1024 threads per channel, so the sum has a
// fixed order (strided per-thread sums, shuffle tree, waves in order) and the result is the same bits on every run - the
// (C, splits) grid with one float atomic per split that this replaces was not (round 4).  A channel is 2 * B * To floats
// (0.9 MB at 16 x 6912): C = 136 workgroups stream it at the same rate the split grid did.
constexpr int DWW_THREADS = 1024;
__global__ __launch_bounds__(DWW_THREADS) void dwconv3_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ dw, int B, int C, int T, int To,
                                                                     int stride) {
    __shared__ float red[3][DWW_THREADS / 64];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (stride == 1 && (T & 3) == 0) {
        // rows of T = To floats, 16-byte aligned: a thread takes four consecutive time steps per pass (one float4 of dy, one of x
        // and the two neighbours), two passes in flight - the one-element loop below ran at 1.3 TB/s (91 us at 16 x 136 x 6912)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int q4 = T >> 2, groups = B * q4;
        auto one = [&](int i) __attribute__((always_inline)) {
            const int b = i / q4, t = (i - b * q4) << 2;
            const float* xr = x + ((size_t)b * C + c) * T + t;
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((size_t)b * C + c) * T + t);
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr);
            const float left = t > 0 ? xr[-1] : 0.f, right = t + 4 < T ? xr[4] : 0.f;
            s0 += g[0] * left + g[1] * v[0] + g[2] * v[1] + g[3] * v[2];
            s1 += g[0] * v[0] + g[1] * v[1] + g[2] * v[2] + g[3] * v[3];
            s2 += g[0] * v[1] + g[1] * v[2] + g[2] * v[3] + g[3] * right;
        };
        int i = threadIdx.x;
        for (; i + DWW_THREADS < groups; i += 2 * DWW_THREADS) {
            one(i);
            one(i + DWW_THREADS);
        }
        if (i < groups) one(i);
    } else {
        const size_t total = (size_t)B * To;
        for (size_t i = threadIdx.x; i < total; i += DWW_THREADS) {
            const int b = (int)(i / To), to = (int)(i - (size_t)b * To);
            const float* xr = x + ((size_t)b * C + c) * T;
            const float g = dy[((size_t)b * C + c) * To + to];
            const int t0 = to * stride - 1;
            if (t0 >= 0) s0 += g * xr[t0];
            s1 += g * xr[t0 + 1];
            if (t0 + 2 < T) s2 += g * xr[t0 + 2];
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < DWW_THREADS / 64; ++i) s += red[threadIdx.x][i];
        dw[c * 3 + threadIdx.x] += s;
    }
}

// ---- GELU (exact erf) --------------------------------------------------------------------------------------
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
        dx[i] = dy[i] * (cdf + v * pdf);
    }
}

// ---- MaxPool1d(kernel 3, stride 2, padding 1) backward: the gradient goes to the FIRST maximum of each window ---------
__device__ __forceinline__ int pool_argmax(const float* xr, int to, int T) {
    const int t = 2 * to;
    int best = t - 1 >= 0 ? t - 1 : t;
    float bv = xr[best];
    if (best != t && xr[t] > bv) { bv = xr[t]; best = t; }
    if (t + 1 < T && xr[t + 1] > bv) best = t + 1;
    return best;
}
__global__ void maxpool3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                      int T, int To) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float* xr = x + (size_t)blockIdx.y * T;
    const float* dr = dy + (size_t)blockIdx.y * To;
    float s = 0.f;
    // windows containing t: to = t/2 (t even or odd) and, for odd t, also (t+1)/2
    const int a = t >> 1;
    if (a < To && pool_argmax(xr, a, T) == t) s += dr[a];
    if (t & 1) {
        const int b = (t + 1) >> 1;
        if (b < To && pool_argmax(xr, b, T) == t) s += dr[b];
    }
    dx[(size_t)blockIdx.y * T + t] = s;
}

// ---- nn.Upsample(scale f, linear, align_corners=False) backward: dy is a channel slice of a wider tensor -------------
__global__ void upsample_linear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int C, int T, int f,
                                           int dy_ctot, int dy_coff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const int c = blockIdx.y % C, b = blockIdx.y / C;
    const int To = T * f;
    const float* dr = dy + ((size_t)b * dy_ctot + dy_coff + c) * To;
    float s = 0.f;
    if (f == 1) {
        s = dr[i];
    } else {
        const int lo = max(0, f * i - f), hi = min(To - 1, f * i + 2 * f);
        for (int to = lo; to <= hi; ++to) {
            float src = ((float)to + 0.5f) * (1.f / (float)f) - 0.5f;
            src = src < 0.f ? 0.f : src;
            const int i0 = (int)src;
            const int i1 = i0 + (i0 < T - 1 ? 1 : 0);
            const float l1 = src - (float)i0, l0 = 1.f - l1;
            if (i0 == i) s += l0 * dr[to];
            if (i1 == i) s += l1 * dr[to];
        }
    }
    dx[((size_t)b * C + c) * T + i] = s;
}

// ---- attention backward helpers ------------------------------------------------------------------------
// per (b, head): in (R, Cc) row-major -> out (Cc, R) row-major, scaled (the O <-> out.transpose(2,3).contiguous() image)
__global__ void transpose_scale_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc, float scale) {
    __shared__ float tile[32][33];
    const float* ib = in + (size_t)blockIdx.z * R * Cc;
    float* ob = out + (size_t)blockIdx.z * R * Cc;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < Cc) ? ib[(size_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (c < Cc && r < R) ob[(size_t)c * R + r] = tile[threadIdx.x][j] * scale;
    }
}

// dP = sum of the NS score slabs (padded HSP x HSP); dS = P o (dP - rowsum(dP o P)); also writes dS^T and P^T
__global__ __launch_bounds__(64) void softmax_bwd_kernel(const float* __restrict__ slabs, const float* __restrict__ P,
                                                          float* __restrict__ dS, float* __restrict__ dST,
                                                          float* __restrict__ PT, int hs, int HSP, int NS) {
    const int bh = blockIdx.x, row = blockIdx.y, lane = threadIdx.x;
    const float* sb = slabs + (size_t)bh * NS * HSP * HSP;
    const float* pb = P + (size_t)bh * HSP * HSP;
    float dp[2], pv[2], acc = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = lane + 64 * h;
        dp[h] = 0.f;
        pv[h] = 0.f;
        if (row < hs && col < hs) {
            for (int s = 0; s < NS; ++s) dp[h] += sb[((size_t)s * HSP + row) * HSP + col];
            pv[h] = pb[row * HSP + col];
        }
        acc += dp[h] * pv[h];
    }
    acc = wave_sum(acc);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = lane + 64 * h;
        if (col < HSP) {
            const float v = (row < hs && col < hs) ? pv[h] * (dp[h] - acc) : 0.f;
            dS[((size_t)bh * HSP + row) * HSP + col] = v;
            dST[((size_t)bh * HSP + col) * HSP + row] = v;
            PT[((size_t)bh * HSP + col) * HSP + row] = (row < hs && col < hs) ? pv[h] : 0.f;
        }
    }
}

}  // namespace

extern "C" int otp_ln_channel_backward(const void* x, const void* grad_y, const void* gamma, void* grad_x, void* dy_xhat,
                                       int B, int C, int T, float eps, void* stream) {
    if (!x || !grad_y || !gamma || !grad_x || !dy_xhat || B <= 0 || C <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    if (C <= 4 * 34)
        hipLaunchKernelGGL((ln_channel_bwd_split_kernel<34, false>), dim3(otp_ceil_div(T, 64), B), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<const float*>(x), static_cast<const float*>(grad_y),
                           static_cast<const float*>(gamma), static_cast<float*>(grad_x), static_cast<float*>(dy_xhat), C, T,
                           eps);
    else
        hipLaunchKernelGGL(ln_channel_bwd_kernel, dim3(otp_ceil_div(T, 256), B), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const float*>(x), static_cast<const float*>(grad_y), static_cast<const float*>(gamma),
                           static_cast<float*>(grad_x), static_cast<float*>(dy_xhat), C, T, eps);
    return otp_launch_status();
}

extern "C" int otp_dwconv3_forward(const void* x, const void* w, void* y, int B, int C, int T, int stride, void* stream) {
    if (!x || !w || !y || B <= 0 || C <= 0 || T <= 0 || stride <= 0) return OTP_ERR_BAD_ARG;
    const int To = (T + 2 - 3) / stride + 1;
    hipLaunchKernelGGL(dwconv3_fwd_kernel, dim3(otp_ceil_div(To, 256), B * C), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(x), static_cast<const float*>(w), static_cast<float*>(y), C, T, To, stride);
    return otp_launch_status();
}

extern "C" int otp_dwconv3_backward(const void* x, const void* w, const void* grad_y, void* grad_x, void* grad_w, int B,
                                    int C, int T, int stride, void* stream) {
    if (!x || !w || !grad_y || !grad_x || !grad_w || B <= 0 || C <= 0 || T <= 0 || stride <= 0) return OTP_ERR_BAD_ARG;
    const int To = (T + 2 - 3) / stride + 1;
    auto st = static_cast<hipStream_t>(stream);
    if (stride == 1 && (T & 3) == 0 && ((reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x)) & 15) == 0) {
        const size_t groups = (size_t)B * C * (T >> 2);
        const size_t blocks = (groups + 255) / 256;
        hipLaunchKernelGGL(dwconv3_bwd_x_s1_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, st,
                           static_cast<const float*>(grad_y), static_cast<const float*>(w), static_cast<float*>(grad_x), C, T, groups);
    } else {
        hipLaunchKernelGGL(dwconv3_bwd_x_kernel, dim3(otp_ceil_div(T, 256), B * C), dim3(256), 0, st,
                           static_cast<const float*>(grad_y), static_cast<const float*>(w), static_cast<float*>(grad_x), C, T, To,
                           stride);
    }
    hipLaunchKernelGGL(dwconv3_bwd_w_kernel, dim3(C), dim3(DWW_THREADS), 0, st, static_cast<const float*>(x),
                       static_cast<const float*>(grad_y), static_cast<float*>(grad_w), B, C, T, To, stride);
    return otp_launch_status();
}

extern "C" int otp_gelu_forward(const void* x, void* y, size_t n, void* stream) {
    if (!x || !y) return OTP_ERR_BAD_ARG;
    if (n == 0) return OTP_OK;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks > 8192 ? 8192 : (unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(x), static_cast<float*>(y), n);
    return otp_launch_status();
}

extern "C" int otp_gelu_backward(const void* x, const void* grad_y, void* grad_x, size_t n, void* stream) {
    if (!x || !grad_y || !grad_x) return OTP_ERR_BAD_ARG;
    if (n == 0) return OTP_OK;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks > 8192 ? 8192 : (unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(x), static_cast<const float*>(grad_y),
                       static_cast<float*>(grad_x), n);
    return otp_launch_status();
}

extern "C" int otp_maxpool3s2_backward(const void* x, const void* grad_y, void* grad_x, int rows, int T, void* stream) {
    if (!x || !grad_y || !grad_x || rows <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    const int To = (T + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(otp_ceil_div(T, 256), rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(x), static_cast<const float*>(grad_y), static_cast<float*>(grad_x), T, To);
    return otp_launch_status();
}

extern "C" int otp_upsample_linear_backward(const void* grad_out, void* grad_x, int B, int C, int T, int f, int out_ctot,
                                            int out_coff, void* stream) {
    if (!grad_out || !grad_x || B <= 0 || C <= 0 || T <= 0 || f <= 0 || out_ctot < out_coff + C) return OTP_ERR_BAD_ARG;
    hipLaunchKernelGGL(upsample_linear_bwd_kernel, dim3(otp_ceil_div(T, 256), B * C), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(grad_out), static_cast<float*>(grad_x), C, T,
                       f, out_ctot, out_coff);
    return otp_launch_status();
}

extern "C" int otp_transpose_scale(const void* in, void* out, int batches, int R, int Cc, float scale, void* stream) {
    if (!in || !out || batches <= 0 || R <= 0 || Cc <= 0) return OTP_ERR_BAD_ARG;
    hipLaunchKernelGGL(transpose_scale_kernel, dim3(otp_ceil_div(Cc, 32), otp_ceil_div(R, 32), batches), dim3(32, 8), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(in), static_cast<float*>(out), R, Cc, scale);
    return otp_launch_status();
}

extern "C" int otp_softmax_backward(const void* slabs, const void* P, void* dS, void* dST, void* PT, int BH, int hs, int NS,
                                    void* stream) {
    if (!slabs || !P || !dS || !dST || !PT || BH <= 0 || hs <= 0 || NS <= 0) return OTP_ERR_BAD_ARG;
    const int HSP = (hs + 15) & ~15;
    if (HSP > 128) return OTP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(BH, HSP), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(slabs), static_cast<const float*>(P), static_cast<float*>(dS),
                       static_cast<float*>(dST), static_cast<float*>(PT), hs, HSP, NS);
    return otp_launch_status();
}

extern "C" size_t otp_ln_channel_backward_workspace(int B, int C, int T) {
    if (B <= 0 || C <= 0 || T <= 0 || C > 4 * 34) return 0;
    return (size_t)2 * C * B * otp_ceil_div(T, 64) * sizeof(float);
}

extern "C" int otp_ln_channel_backward_params(const void* x, const void* grad_y, const void* gamma, void* grad_x,
                                              void* grad_gamma, void* grad_beta, void* workspace, size_t workspace_bytes,
                                              int B, int C, int T, float eps, void* stream) {
    if (!x || !grad_y || !gamma || !grad_x || !grad_gamma || !grad_beta || !workspace || B <= 0 || C <= 0 || T <= 0)
        return OTP_ERR_BAD_ARG;
    if (C > 4 * 34) return OTP_ERR_UNSUPPORTED;
    if (workspace_bytes < otp_ln_channel_backward_workspace(B, C, T)) return OTP_ERR_WORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    int gx = otp_ceil_div(T, 64);
    if ((T & 3) == 0 && T >= 1024) {                          // 256 tokens x all channels per workgroup, 16-byte accesses
        gx = otp_ceil_div(T, 256);
        hipLaunchKernelGGL((ln_channel_bwd_wide_kernel<17>), dim3(gx, B), dim3(512), 0, st, static_cast<const float*>(x),
                           static_cast<const float*>(grad_y), static_cast<const float*>(gamma), static_cast<float*>(grad_x),
                           static_cast<float*>(workspace), C, T, eps);
    } else
    hipLaunchKernelGGL((ln_channel_bwd_split_kernel<34, true>), dim3(gx, B), dim3(256), 0, st, static_cast<const float*>(x),
                       static_cast<const float*>(grad_y), static_cast<const float*>(gamma), static_cast<float*>(grad_x),
                       static_cast<float*>(workspace), C, T, eps);
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(2 * C), dim3(256), 0, st, static_cast<const float*>(workspace),
                       static_cast<float*>(grad_gamma), static_cast<float*>(grad_beta), C, gx * B);
    return otp_launch_status();
}
