// Optimizer step of the reference training loop (script/Common.py:136-143): global-norm gradient clipping
// (torch.nn.utils.clip_grad_norm_, max_norm = TRAIN.CLIP_GRAD_L2NORM) followed by AdamW
// (thirdparty/utils/train_utils.py:129-133, torch.optim.AdamW semantics) over FLAT parameter / gradient / moment buffers.
// Two HBM-bound passes: sum of squares (fp64 partials per workgroup, added in a fixed order: no atomics, the same bits on
// every run) and the fused clip + update, which reads the
// clip coefficient from device memory - no host synchronisation between backward and the next forward.
#include "common.h"

namespace {

constexpr int SUMSQ_PARTS = 1024;       // most workgroups of one sum-of-squares launch = scratch doubles behind the accumulator

// part[block] = this workgroup's share of sum(g^2) (fp64, fixed order: strided per-thread sums, shuffle tree, waves in order)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n4, size_t n,
                                                     double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    const otp_f32x4* g4 = reinterpret_cast<const otp_f32x4*>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const otp_f32x4 v = g4[i];
        s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += (double)g[i] * (double)g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// *acc += sum of the `parts` partials in a fixed order (one workgroup; no atomics: the same bits on every run)
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const double* __restrict__ part, int parts, double* __restrict__ acc) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < parts; i += 256) s += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *acc += (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float clip, float lr_wd, float b1, float b2,
                                          float step_size, float inv_sqrt_bc2, float eps) {
    g *= clip;
    p *= lr_wd;                                       // p * (1 - lr * weight_decay)
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, size_t n4, size_t n, float lr, float b1, float b2,
                                                     float eps, float wd, float bc1, float bc2,
                                                     const double* __restrict__ gradnorm_sq, float max_norm) {
    float clip = 1.f;
    if (gradnorm_sq && max_norm > 0.f) {
        const float c = max_norm / ((float)sqrt(*gradnorm_sq) + 1e-6f);      // clip_grad_norm_: clamped to 1
        clip = c < 1.f ? c : 1.f;
    }
    const float lr_wd = 1.f - lr * wd, step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    otp_f32x4* p4 = reinterpret_cast<otp_f32x4*>(p);
    otp_f32x4* m4 = reinterpret_cast<otp_f32x4*>(m);
    otp_f32x4* v4 = reinterpret_cast<otp_f32x4*>(v);
    const otp_f32x4* g4 = reinterpret_cast<const otp_f32x4*>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        otp_f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
        const otp_f32x4 gg = g4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pp[e], me = mm[e], ve = vv[e];
            adamw_one(pe, gg[e], me, ve, clip, lr_wd, b1, b2, step_size, inv_sqrt_bc2, eps);
            pp[e] = pe; mm[e] = me; vv[e] = ve;
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256)
            adamw_one(p[i], g[i], m[i], v[i], clip, lr_wd, b1, b2, step_size, inv_sqrt_bc2, eps);
}

unsigned grid_for(size_t n4) {
    const size_t b = (n4 + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" size_t otp_grad_sumsq_scratch(void) { return SUMSQ_PARTS; }

extern "C" int otp_grad_sumsq(const void* grad, size_t n, void* acc_f64, void* stream) {
    if (!grad || !acc_f64 || n == 0) return OTP_ERR_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(grad) & 15) return OTP_ERR_UNSUPPORTED;
    unsigned grid = grid_for(n / 4);
    if (grid > SUMSQ_PARTS) grid = SUMSQ_PARTS;
    double* acc = static_cast<double*>(acc_f64);
    auto st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, st, static_cast<const float*>(grad), n / 4, n, acc + 1);
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, acc + 1, (int)grid, acc);
    return otp_launch_status();
}

extern "C" int otp_adamw_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, size_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, const void* gradnorm_sq_f64,
                              float max_norm, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n == 0 || step < 1) return OTP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
        return OTP_ERR_UNSUPPORTED;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<float*>(param), static_cast<const float*>(grad), static_cast<float*>(exp_avg),
                       static_cast<float*>(exp_avg_sq), n / 4, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                       static_cast<const double*>(gradnorm_sq_f64), max_norm);
    return otp_launch_status();
}
