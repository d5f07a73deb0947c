// TransformerBlock of the flow encoder (reference model/blocks.py:264-280, 400-453 with C = num_joints = 17 channels,
// stride 1; model/OTPose.py:209-216 builds it, :334 runs it on total_b) around the channel attention, as TWO launches
// instead of nine: 17 channels fit in the registers of the thread that owns a time step, every weight matrix is at most
// 68 x 17 floats and is read with uniform addresses through the scalar cache, so the chain of per-token layers needs no
// LDS, no barriers and no intermediate tensors.
//
//   otp_flow_front:  q, k, v = Conv1d_1x1(LayerNorm(dwconv3(ln1(x))))             (ln1 + dwconv_ln3 + three 1x1 convs)
//   [otp_chan_attn]  att = softmax(q k^T * scale) v  in the reference's transposed memory image (global over T: stays a
//                    separate operator)
//   otp_flow_back:   y = x + s_a * (W_p att + b_p);  out = y + s_m * (W_2 gelu(W_1 ln2(y) + b_1) + b_2)
//                                                                                  (proj + residual, ln2, MLP + residual)
//
// The generic kernels these replace are latency bound at this size (12 launches of 15-30 us per block, 6 blocks on the
// serial path between the backbone and the temporal encoders: 1.2 ms); here a block costs the attention plus two ~10 us
// launches.  Arithmetic: exact fp32 FMAs (the layers' own formulas; sums run over channels in index order).
// Parameter blocks (floats, built by the host - otpose_amd/ops.py pack_flow_front / pack_flow_back):
//   front: ln1 gamma[C], beta[C]; then for q, k, v:  dw[C][3], norm gamma[C], norm beta[C], W[C][C] (out, in), bias[C]
//   back:  W_p[C][C] * s_a[out], b_p[C] * s_a; ln2 gamma[C], beta[C]; W_1[H][C], b_1[H]; W_2^T[H][C] * s_m[out], b_2[C] * s_m
#include "common.h"

namespace {

__device__ __forceinline__ float fe_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

template <int C>
__device__ __forceinline__ void fe_layer_norm(float (&v)[C], const float* __restrict__ g, const float* __restrict__ b, float eps) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) s += v[c];
    const float mu = s * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        v[c] -= mu;
        q += v[c] * v[c];
    }
    const float rs = 1.f / sqrtf(q * (1.f / C) + eps);
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = v[c] * rs * g[c] + b[c];
}

template <int C>
constexpr int fe_front_floats() { return 2 * C + 3 * (3 * C + 2 * C + C * C + C); }
template <int C>
constexpr int fe_back_floats(int H) { return C * C + C + 2 * C + H * C + H + H * C + C; }

// one wave per projection (q / k / v) over the same 64 time steps: the wave index is uniform, so the branch's parameters
// still arrive through the scalar cache, and three times as many waves share the latency-bound work
template <int C>
__global__ __launch_bounds__(192) void flow_front_kernel(const float* __restrict__ x, const float* __restrict__ prm,
                                                         float* __restrict__ q, float* __restrict__ k, float* __restrict__ v,
                                                         int T, float eps) {
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int br = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t >= T) return;
    const size_t base = (size_t)blockIdx.y * C * T + t;
    // ln1 of the token and of its two neighbours (the depthwise conv pads the ln1 OUTPUT with zeros)
    float n[3][C];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tt = t - 1 + s;
        const bool ok = tt >= 0 && tt < T;
#pragma unroll
        for (int c = 0; c < C; ++c) n[s][c] = ok ? x[base + (size_t)c * T + (s - 1)] : 0.f;
        fe_layer_norm<C>(n[s], prm, prm + C, eps);
        if (!ok) {
#pragma unroll
            for (int c = 0; c < C; ++c) n[s][c] = 0.f;
        }
    }
    float* out = br == 0 ? q : (br == 1 ? k : v);
    const float* p = prm + 2 * C + br * (3 * C + 2 * C + C * C + C);
    float d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = p[c * 3] * n[0][c] + p[c * 3 + 1] * n[1][c] + p[c * 3 + 2] * n[2][c];
    fe_layer_norm<C>(d, p + 3 * C, p + 4 * C, eps);
    const float* W = p + 5 * C;
    const float* bias = W + C * C;
#pragma unroll
    for (int o = 0; o < C; ++o) {
        float a = bias[o];
#pragma unroll
        for (int c = 0; c < C; ++c) a = fmaf(W[o * C + c], d[c], a);
        out[base + (size_t)o * T] = a;
    }
}

// four waves over the same 64 time steps, each with a quarter of the hidden units (m = wave, wave + 4, ...: uniform per
// wave, weights through the scalar cache); the partial sums of the down-projection meet in LDS
template <int C>
__global__ __launch_bounds__(256) void flow_back_kernel(const float* __restrict__ x, const float* __restrict__ att,
                                                        const float* __restrict__ prm, float* __restrict__ out, int H, int T,
                                                        float eps) {
    __shared__ float part[3][C][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const size_t base = (size_t)blockIdx.y * C * T + (live ? t : 0);
    float a[C], y[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        a[c] = att[base + (size_t)c * T];
        y[c] = x[base + (size_t)c * T];
    }
    const float* Wp = prm;
    const float* bp = Wp + C * C;
#pragma unroll
    for (int o = 0; o < C; ++o) {
        float s = bp[o];
#pragma unroll
        for (int c = 0; c < C; ++c) s = fmaf(Wp[o * C + c], a[c], s);
        y[o] += s;
    }
    float nn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) nn[c] = y[c];
    fe_layer_norm<C>(nn, bp + C, bp + 2 * C, eps);
    const float* W1 = bp + 3 * C;
    const float* b1 = W1 + (size_t)H * C;
    const float* W2t = b1 + H;
    const float* b2 = W2t + (size_t)H * C;
    float acc[C];
#pragma unroll
    for (int o = 0; o < C; ++o) acc[o] = 0.f;
    // hidden unit by hidden unit: h_m is consumed at once, the 4C-wide activation never exists
#pragma unroll 2
    for (int m = wave; m < H; m += 4) {
        float h = b1[m];
#pragma unroll
        for (int c = 0; c < C; ++c) h = fmaf(W1[m * C + c], nn[c], h);
        h = fe_gelu(h);
#pragma unroll
        for (int o = 0; o < C; ++o) acc[o] = fmaf(W2t[m * C + o], h, acc[o]);
    }
    if (wave > 0) {
#pragma unroll
        for (int o = 0; o < C; ++o) part[wave - 1][o][lane] = acc[o];
    }
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int o = 0; o < C; ++o)
            out[base + (size_t)o * T] = y[o] + b2[o] + ((acc[o] + part[0][o][lane]) + (part[1][o][lane] + part[2][o][lane]));
    }
}

}  // namespace

extern "C" int otp_flow_block_supported(int C, int hidden, int T) { return (C == 17 && hidden > 0 && T > 0) ? 1 : 0; }

extern "C" size_t otp_flow_front_param_floats(int C) { return C == 17 ? (size_t)fe_front_floats<17>() : 0; }

extern "C" size_t otp_flow_back_param_floats(int C, int hidden) {
    return (C == 17 && hidden > 0) ? (size_t)fe_back_floats<17>(hidden) : 0;
}

extern "C" int otp_flow_front(const void* x, const void* params, void* q, void* k, void* v, int B, int C, int T, float eps,
                              void* stream) {
    if (!x || !params || !q || !k || !v || B <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    if (C != 17) return OTP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(flow_front_kernel<17>, dim3(otp_ceil_div(T, 64), B), dim3(192), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(x), static_cast<const float*>(params), static_cast<float*>(q),
                       static_cast<float*>(k), static_cast<float*>(v), T, eps);
    return otp_launch_status();
}

extern "C" int otp_flow_back(const void* x, const void* att, const void* params, void* out, int B, int C, int hidden, int T,
                             float eps, void* stream) {
    if (!x || !att || !params || !out || B <= 0 || T <= 0 || hidden <= 0) return OTP_ERR_BAD_ARG;
    if (C != 17) return OTP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(flow_back_kernel<17>, dim3(otp_ceil_div(T, 64), B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(x), static_cast<const float*>(att), static_cast<const float*>(params),
                       static_cast<float*>(out), hidden, T, eps);
    return otp_launch_status();
}
