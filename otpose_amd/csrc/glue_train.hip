// Training-step glue of the ConvTransformer blocks (model/blocks.py:264-316): the residual update
//     out = x + drop_path(scale * a)       (AffineDropPath: per-channel scale (1, C, 1), per-sample Bernoulli mask / keep)
// forward in one pass and backward in one pass (grad_a, grad_scale; grad_x is grad_out itself), instead of the four
// element-wise passes forward and six backward that the same expression costs as separate tensor ops.  HBM-bound:
// 12 B/element forward, 12 B/element backward.
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void scale_residual_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ scale, const float* __restrict__ mask,
                                                              float* __restrict__ out, int C, int T, size_t total4) {
    // one thread = 4 consecutive t of one (b, c) row (T % 4 == 0)
    const int T4 = T / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / T4;
        const int c = (int)(row % C), b = (int)(row / C);
        const float k = scale[c] * (mask ? mask[b] : 1.f);
        const f4 xv = reinterpret_cast<const f4*>(x)[i], av = reinterpret_cast<const f4*>(a)[i];
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = xv[j] + k * av[j];
        reinterpret_cast<f4*>(out)[i] = o;
    }
}

__global__ __launch_bounds__(256) void scale_residual_scalar_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ mask, float* __restrict__ out,
                                                                     int C, int T, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / T;
        const int c = (int)(row % C), b = (int)(row / C);
        out[i] = x[i] + scale[c] * (mask ? mask[b] : 1.f) * a[i];
    }
}

// grid (C, S): channel c, split s of the B*T elements.  ga = m[b]*scale[c]*g, partial[c][s] = sum m[b]*a*g
__global__ __launch_bounds__(256) void scale_residual_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                                  const float* __restrict__ scale, const float* __restrict__ mask,
                                                                  float* __restrict__ ga, float* __restrict__ part, int B, int C,
                                                                  int T) {
    __shared__ float red[4];
    const int c = blockIdx.x, S = gridDim.y, s = blockIdx.y;
    const float sc = scale[c];
    float acc = 0.f;
    if ((T & 3) == 0) {
        const int T4 = T / 4;
        const size_t total = (size_t)B * T4;
        for (size_t i = (size_t)s * 256 + threadIdx.x; i < total; i += (size_t)S * 256) {
            const int b = (int)(i / T4), t4 = (int)(i - (size_t)b * T4);
            const size_t o = ((size_t)b * C + c) * T4 + t4;
            const float m = mask ? mask[b] : 1.f;
            const f4 gv = reinterpret_cast<const f4*>(g)[o], av = reinterpret_cast<const f4*>(a)[o];
            f4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = m * sc * gv[j];
                acc += m * av[j] * gv[j];
            }
            reinterpret_cast<f4*>(ga)[o] = r;
        }
    } else {
        const size_t total = (size_t)B * T;
        for (size_t i = (size_t)s * 256 + threadIdx.x; i < total; i += (size_t)S * 256) {
            const int b = (int)(i / T), t = (int)(i - (size_t)b * T);
            const size_t o = ((size_t)b * C + c) * T + t;
            const float m = mask ? mask[b] : 1.f;
            ga[o] = m * sc * g[o];
            acc += m * a[o] * g[o];
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)c * S + s] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void scale_residual_finish_kernel(const float* __restrict__ part, float* __restrict__ gscale, int C, int S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int i = 0; i < S; ++i) s += (double)part[(size_t)c * S + i];
    gscale[c] = (float)s;
}

int bwd_splits(int B, int C, int T) {
    const size_t per_channel = (size_t)B * T / 4;
    int s = (int)(2048 / (C > 0 ? C : 1));
    if (s < 1) s = 1;
    const size_t cap = (per_channel + 1023) / 1024;              // at least ~1024 float4 per workgroup
    if ((size_t)s > cap) s = (int)(cap ? cap : 1);
    return s;
}

}  // namespace

extern "C" int otp_scale_residual(const void* x, const void* a, const void* scale, const void* mask, void* out, int B, int C,
                                  int T, void* stream) {
    if (!x || !a || !scale || !out || B <= 0 || C <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t total = (size_t)B * C * T;
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    if ((T & 3) == 0) {
        const size_t t4 = total / 4, blocks = (t4 + 255) / 256;
        scale_residual_kernel<<<(unsigned)(blocks > 8192 ? 8192 : blocks), 256, 0, st>>>(f(x), f(a), f(scale), f(mask),
                                                                                      static_cast<float*>(out), C, T, t4);
    } else {
        const size_t blocks = (total + 255) / 256;
        scale_residual_scalar_kernel<<<(unsigned)(blocks > 8192 ? 8192 : blocks), 256, 0, st>>>(
            f(x), f(a), f(scale), f(mask), static_cast<float*>(out), C, T, total);
    }
    return otp_launch_status();
}

extern "C" size_t otp_scale_residual_backward_workspace(int B, int C, int T) {
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    return (size_t)C * bwd_splits(B, C, T) * sizeof(float);
}

extern "C" int otp_scale_residual_backward(const void* grad_out, const void* a, const void* scale, const void* mask, void* grad_a,
                                           void* grad_scale, void* workspace, size_t workspace_bytes, int B, int C, int T,
                                           void* stream) {
    if (!grad_out || !a || !scale || !grad_a || !grad_scale || !workspace || B <= 0 || C <= 0 || T <= 0) return OTP_ERR_BAD_ARG;
    if (workspace_bytes < otp_scale_residual_backward_workspace(B, C, T)) return OTP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int S = bwd_splits(B, C, T);
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    scale_residual_bwd_kernel<<<dim3(C, S), 256, 0, st>>>(f(grad_out), f(a), f(scale), f(mask), static_cast<float*>(grad_a),
                                                           static_cast<float*>(workspace), B, C, T);
    scale_residual_finish_kernel<<<otp_ceil_div(C, 64), 64, 0, st>>>(static_cast<const float*>(workspace),
                                                                     static_cast<float*>(grad_scale), C, S);
    return otp_launch_status();
}
