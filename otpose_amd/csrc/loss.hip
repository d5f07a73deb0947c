// Heat-map loss of the reference's default criterion ST_OHKW_MSELoss (model/loss.py:25-92) as three
// small launches instead of ~200: per-(sample, joint) squared-error sums with wavefront-shuffle
// reductions, a one-workgroup finish (per-joint "ground truth has an exact-1 peak" flags, top-k of 17
// per sample, means), and an optional elementwise gradient pass.
#include "common.h"

namespace {

// grid (B*J): stats[bj] = {sum (a-gg)^2, sum (a-tt)^2, max g}, a = s*w, gg = g*w, tt = t*w
__global__ __launch_bounds__(256) void loss_stats_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                          const float* __restrict__ g, const float* __restrict__ w,
                                                          float* __restrict__ stats, int HW) {
    __shared__ float red[3][4];
    const int bj = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float wt = w[bj];
    const size_t base = (size_t)bj * HW;
    float sg = 0.f, stt = 0.f, mx = -INFINITY;
    for (int p = tid; p < HW; p += 256) {
        const float a = s[base + p] * wt, gv = g[base + p];
        const float dg = a - gv * wt, dt = a - t[base + p] * wt;
        sg += dg * dg;
        stt += dt * dt;
        mx = fmaxf(mx, gv);
    }
    sg = wave_sum(sg); stt = wave_sum(stt); mx = wave_max(mx);
    if (lane == 0) { red[0][wave] = sg; red[1][wave] = stt; red[2][wave] = mx; }
    __syncthreads();
    if (tid == 0) {
        stats[bj * 3 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        stats[bj * 3 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        stats[bj * 3 + 2] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    }
}

// one workgroup: flags, per-sample top-k, scalars; coef[bj] = {dL/da scale of the (a-gg) term, of the (a-tt) term}
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ stats, int* __restrict__ flags,
                                                           float* __restrict__ result, float* __restrict__ coef,
                                                           int B, int J, int HW, int topk, int flags_given) {
    extern __shared__ float sm[];
    float* l = sm;                    // [B*J] per-sample per-joint loss
    int* fl = reinterpret_cast<int*>(sm + B * J);     // [J]
    float* acc = sm + B * J + J;      // [2]: ohkm sum, mse sum
    int* sel = reinterpret_cast<int*>(acc + 2);       // [B*J] 1 when (b,j) is in the sample's top-k
    float* mj = reinterpret_cast<float*>(sel + B * J);      // [J] per-joint mse terms
    float* ob = mj + J;                                     // [B] per-sample ohkm terms
    const int tid = threadIdx.x;
    for (int j = tid; j < J; j += blockDim.x) {
        int f;
        if (flags_given) {
            f = flags[j];
        } else {
            float mx = -INFINITY;
            for (int b = 0; b < B; ++b) mx = fmaxf(mx, stats[(b * J + j) * 3 + 2]);
            f = (mx == 1.0f) ? 1 : 0;                 // exact compare, loss.py:47
            flags[j] = f;
        }
        fl[j] = f;
    }
    __syncthreads();
    const float inv_hw = 1.f / (float)HW;
    for (int i = tid; i < B * J; i += blockDim.x) {
        const int j = i % J;
        const float sg = stats[i * 3], st = stats[i * 3 + 1];
        l[i] = 0.5f * (fl[j] ? sg : sg + st) * inv_hw;
        sel[i] = 0;
    }
    __syncthreads();
    // mse term: sum_j mean_{b,p}  (loss.py:52-64)
    if (tid < J) {
        float m = 0.f;
        for (int b = 0; b < B; ++b) m += stats[(b * J + tid) * 3] + (fl[tid] ? 0.f : stats[(b * J + tid) * 3 + 1]);
        mj[tid] = m / ((float)B * (float)HW);
    }
    // ohkm: per sample the k largest joint losses (loss.py:13-23); ties resolved by lower joint index
    for (int b = tid; b < B; b += blockDim.x) {
        float sum = 0.f;
        for (int k = 0; k < topk; ++k) {
            int best = -1;
            float bv = -INFINITY;
            for (int j = 0; j < J; ++j)
                if (!sel[b * J + j] && l[b * J + j] > bv) { bv = l[b * J + j]; best = j; }
            if (best < 0) break;
            sel[b * J + best] = 1;
            sum += bv;
        }
        ob[b] = sum / (float)topk;
    }
    __syncthreads();
    if (tid == 0) {                   // sums in index order: the same bits on every run (no atomics)
        float a0 = 0.f, a1 = 0.f;
        for (int b = 0; b < B; ++b) a0 += ob[b];
        for (int j = 0; j < J; ++j) a1 += mj[j];
        acc[0] = a0; acc[1] = a1;
        const float ohkm = acc[0] / (float)B;
        result[0] = ohkm;
        result[1] = acc[1] / (float)J;
        result[2] = ohkm + acc[1];
    }
    if (coef) {
        for (int i = tid; i < B * J; i += blockDim.x) {
            const int j = i % J;
            // d final / d a = [sel/(topk*B*HW) + 2/(B*HW)] * ((a-gg) + nf*(a-tt))
            const float c = (sel[i] ? 1.f / ((float)topk * B * HW) : 0.f) + 2.f / ((float)B * HW);
            coef[i * 2] = c;
            coef[i * 2 + 1] = fl[j] ? 0.f : c;
        }
    }
}

__global__ void loss_grad_kernel(const float* __restrict__ s, const float* __restrict__ t, const float* __restrict__ g,
                                 const float* __restrict__ w, const float* __restrict__ coef, float* __restrict__ gs,
                                 float* __restrict__ gt, float* __restrict__ gg, int HW) {
    const int bj = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const size_t i = (size_t)bj * HW + p;
    const float wt = w[bj], a = s[i] * wt;
    const float dg = a - g[i] * wt, dt = a - t[i] * wt;
    const float cg = coef[bj * 2], ct = coef[bj * 2 + 1];
    if (gs) gs[i] = wt * (cg * dg + ct * dt);
    if (gt) gt[i] = -wt * ct * dt;
    if (gg) gg[i] = -wt * cg * dg;              // the 2nd criterion call's target depends on the model (Common.py:128-130)
}

// ---- JointsMSE_OHKMMSELoss (loss.py:95-148) and JointMSELoss (loss.py:151-182): one prediction, one target ----
// grid (B*J): ss[bj] = sum (o*w - g*w)^2
__global__ __launch_bounds__(256) void joints_stats_kernel(const float* __restrict__ o, const float* __restrict__ g,
                                                            const float* __restrict__ w, float* __restrict__ ss, int HW) {
    __shared__ float red[4];
    const int bj = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float wt = w ? w[bj] : 1.f;
    const size_t base = (size_t)bj * HW;
    float acc = 0.f;
    for (int p = tid; p < HW; p += 256) {
        const float d = o[base + p] * wt - g[base + p] * wt;
        acc += d * d;
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) ss[bj] = red[0] + red[1] + red[2] + red[3];
}

// one workgroup; ohkm != 0: result = {ohkm, mse/eff, ohkm + mse}; else result = {0, mse/eff, mse/eff}
__global__ __launch_bounds__(256) void joints_finish_kernel(const float* __restrict__ ss, float* __restrict__ result,
                                                             float* __restrict__ coef, int B, int J, int HW, int topk,
                                                             int ohkm, float eff) {
    extern __shared__ float sm[];
    float* l = sm;                                    // [B*J]
    int* sel = reinterpret_cast<int*>(sm + B * J);    // [B*J]
    float* acc = sm + 2 * B * J;                      // [2]
    float* mj = acc + 2;                              // [J] per-joint mse terms
    float* ob = mj + J;                               // [B] per-sample ohkm terms
    const int tid = threadIdx.x;
    const float inv_hw = 1.f / (float)HW;
    for (int i = tid; i < B * J; i += blockDim.x) { l[i] = 0.5f * ss[i] * inv_hw; sel[i] = 0; }
    __syncthreads();
    for (int j = tid; j < J; j += blockDim.x) {
        float m = 0.f;
        for (int b = 0; b < B; ++b) m += ss[b * J + j];
        mj[j] = m / ((float)B * (float)HW);
    }
    if (ohkm)
        for (int b = tid; b < B; b += blockDim.x) {
            float sum = 0.f;
            for (int k = 0; k < topk; ++k) {
                int best = -1;
                float bv = -INFINITY;
                for (int j = 0; j < J; ++j)
                    if (!sel[b * J + j] && l[b * J + j] > bv) { bv = l[b * J + j]; best = j; }
                if (best < 0) break;
                sel[b * J + best] = 1;
                sum += bv;
            }
            ob[b] = sum / (float)topk;
        }
    __syncthreads();
    if (tid == 0) {                                   // sums in index order (no atomics)
        float a0 = 0.f, a1 = 0.f;
        if (ohkm)
            for (int b = 0; b < B; ++b) a0 += ob[b];
        for (int j = 0; j < J; ++j) a1 += mj[j];
        acc[0] = a0; acc[1] = a1;
        if (ohkm) {
            const float v = acc[0] / (float)B;
            result[0] = v; result[1] = acc[1] / eff; result[2] = v + acc[1];
        } else {
            result[0] = 0.f; result[1] = acc[1] / eff; result[2] = acc[1] / eff;
        }
    }
    if (coef)
        for (int i = tid; i < B * J; i += blockDim.x)
            coef[i] = ohkm ? (sel[i] ? 1.f / ((float)topk * B * HW) : 0.f) + 2.f / ((float)B * HW)
                           : 2.f / ((float)B * HW * eff);
}

__global__ void joints_grad_kernel(const float* __restrict__ o, const float* __restrict__ g, const float* __restrict__ w,
                                   const float* __restrict__ coef, float* __restrict__ go, int HW) {
    const int bj = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const size_t i = (size_t)bj * HW + p;
    const float wt = w ? w[bj] : 1.f;
    go[i] = wt * coef[bj] * (o[i] * wt - g[i] * wt);
}

}  // namespace

extern "C" size_t otp_loss_workspace(int B, int J) {
    if (B <= 0 || J <= 0) return 0;
    return (size_t)B * J * 5 * sizeof(float);
}

extern "C" int otp_loss_st_ohkw_grads(const void* s, const void* t, const void* g, const void* w, void* flags, void* result,
                                void* grad_s, void* grad_t, void* grad_g, void* workspace, size_t workspace_bytes, int B, int J,
                                int HW, int topk, int flags_given, void* stream) {
    if (!s || !t || !g || !w || !flags || !result || !workspace || B <= 0 || J <= 0 || HW <= 0 || topk <= 0 || topk > J)
        return OTP_ERR_BAD_ARG;
    if (workspace_bytes < otp_loss_workspace(B, J)) return OTP_ERR_WORKSPACE;
    const size_t lds = ((size_t)B * J * 2 + 2 * J + B + 2) * sizeof(float);
    if (lds > 64 * 1024) return OTP_ERR_UNSUPPORTED;
    auto st = static_cast<hipStream_t>(stream);
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    float* stats = static_cast<float*>(workspace);
    float* coef = stats + (size_t)B * J * 3;
    const bool want_grad = grad_s || grad_t || grad_g;
    hipLaunchKernelGGL(loss_stats_kernel, dim3(B * J), dim3(256), 0, st, f(s), f(t), f(g), f(w), stats, HW);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), lds, st, stats, static_cast<int*>(flags),
                       static_cast<float*>(result), want_grad ? coef : nullptr, B, J, HW, topk, flags_given);
    if (want_grad)
        hipLaunchKernelGGL(loss_grad_kernel, dim3(otp_ceil_div(HW, 256), B * J), dim3(256), 0, st, f(s), f(t), f(g), f(w),
                           coef, static_cast<float*>(grad_s), static_cast<float*>(grad_t), static_cast<float*>(grad_g), HW);
    return otp_launch_status();
}

extern "C" int otp_loss_st_ohkw(const void* s, const void* t, const void* g, const void* w, void* flags, void* result,
                                void* grad_s, void* grad_t, void* workspace, size_t workspace_bytes, int B, int J,
                                int HW, int topk, int flags_given, void* stream) {
    return otp_loss_st_ohkw_grads(s, t, g, w, flags, result, grad_s, grad_t, nullptr, workspace, workspace_bytes, B, J, HW,
                                  topk, flags_given, stream);
}

extern "C" int otp_loss_joints_mse(const void* o, const void* g, const void* w, void* result, void* grad_o, void* workspace,
                                   size_t workspace_bytes, int B, int J, int HW, int topk, int ohkm,
                                   int effective_num_joints, void* stream) {
    if (!o || !g || !result || !workspace || B <= 0 || J <= 0 || HW <= 0) return OTP_ERR_BAD_ARG;
    if (ohkm && (topk <= 0 || topk > J)) return OTP_ERR_BAD_ARG;
    if (workspace_bytes < otp_loss_workspace(B, J)) return OTP_ERR_WORKSPACE;
    const size_t lds = ((size_t)B * J * 2 + J + B + 2) * sizeof(float);
    if (lds > 64 * 1024) return OTP_ERR_UNSUPPORTED;
    auto st = static_cast<hipStream_t>(stream);
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    float* ss = static_cast<float*>(workspace);
    float* coef = ss + (size_t)B * J;
    const float eff = (float)(effective_num_joints > 0 ? effective_num_joints : J);
    hipLaunchKernelGGL(joints_stats_kernel, dim3(B * J), dim3(256), 0, st, f(o), f(g), f(w), ss, HW);
    hipLaunchKernelGGL(joints_finish_kernel, dim3(1), dim3(256), lds, st, ss, static_cast<float*>(result),
                       grad_o ? coef : nullptr, B, J, HW, topk, ohkm, eff);
    if (grad_o)
        hipLaunchKernelGGL(joints_grad_kernel, dim3(otp_ceil_div(HW, 256), B * J), dim3(256), 0, st, f(o), f(g), f(w), coef,
                           static_cast<float*>(grad_o), HW);
    return otp_launch_status();
}
