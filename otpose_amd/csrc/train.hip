// Training-step building blocks for the convolutional part of the OTPose path (HRNet, RSB heads, offset / mask
// convs; reference model/HRNet.py, model/RSB.py run under model.train(): script/Common.py:91,136-144):
//   * conv backward w.r.t. the input = a forward convolution of grad_out with the flipped, channel-transposed
//     weights (otp_conv2d_pack_weight_dgrad feeds otp_conv2d; stride-2 layers first zero-insert grad_out, otp_dilate)
//   * conv backward w.r.t. the weights (otp_conv2d_wgrad): implicit GEMM over the pixels on the f32 matrix cores,
//     partial sums per workgroup in registers, one float atomic per weight per workgroup at the end
//   * BatchNorm2d in training mode (batch statistics over N*H*W per channel, running-stat update) fused with the
//     residual add and ReLU that follow it in every HRNet / RSB block, and its backward
//   * per-channel sums (bias gradients)
// Everything is fp32 NCHW like the forward path; statistics are combined in fp64.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// dgrad helpers
// ------------------------------------------------------------------------------------------------
// packed weights of the transposed convolution: conv(Cout -> Cin) with w'[ci][co][flip(tap)] = w[co][ci][tap],
// in the [tap][Cin' = Cout][Cout16' = Cin16] layout otp_conv2d expects
__global__ void pack_weight_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                         int KK, int Cin16) {
    const int total = KK * Cout * Cin16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % Cin16, r = i / Cin16;
        const int co = r % Cout, tap = r / Cout;
        wp[i] = ci < Cin ? w[((size_t)co * Cin + ci) * KK + (KK - 1 - tap)] : 0.f;
    }
}

// out (N*C, H, W) = zeros with out[y*s][x*s] = in[y][x]  (grad_out of a stride-s conv, ready for a stride-1 dgrad)
__global__ void dilate_kernel(const float* __restrict__ in, float* __restrict__ out, int Hi, int Wi, int s, int H,
                              int W, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t r = i / W;
        const int y = (int)(r % H);
        const size_t plane = r / H;
        float v = 0.f;
        if (y % s == 0 && x % s == 0 && y / s < Hi && x / s < Wi) v = in[(plane * Hi + y / s) * Wi + x / s];
        out[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[co][ci][tap] += sum_{n, y, x} dY[n, co, y, x] * X[n, ci, y*s - pad + ti*dil, x*s - pad + tj*dil]
// ------------------------------------------------------------------------------------------------
// grid (workgroups over (image, row-block) tiles, co groups of 48, ci groups of 48); 4 waves.  GEMM view per tap:
// D[co][ci] += A[co][pixel] * B[pixel][ci], contraction over the pixels in steps of 4 (v_mfma_f32_16x16x4_f32).
// LDS: dY rows of the tile [48][PS] and the input rows it touches [48][NRX][LWP] with a ZERO halo (so the taps
// that fall off the image multiply zeros), plus a per-pixel table of input offsets (any width, any stride).
// A wave owns up to 7 (tap, co-block) groups x 3 ci-blocks = 21 accumulator tiles, kept in registers over every tile
// the workgroup walks; they leave through one float atomic per weight at the end.
constexpr int WG_CO = 48, WG_CI = 48, WG_GROUPS = 7;

struct WgradPlan {
    int N, Cin, H, W, Cout, KS, stride, pad, dil, Ho, Wo;
    int x_ctot, x_coff, dy_ctot, dy_coff;
    int RT, WT, PT, PS;       // output rows x columns per tile, pixels (rounded to 4), dY row pitch (PS % 32 == 2)
    int NRX, LWP, CSX;        // staged input rows, staged row pitch, channel stride (CSX % 32 == 2)
    int sparse;               // dilated kernels: stage only the KS rows a tap row touches per output row (r = ti*RT + yl)
    int tiles_x, tiles_per_img, ntiles;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, const WgradPlan P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;                                   // [48][PS]
    float* xs = smem + WG_CO * P.PS;                     // [48][CSX]
    int* poff = reinterpret_cast<int*>(xs + WG_CI * P.CSX);   // [PT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kl = lane >> 4;
    const int co0 = blockIdx.y * WG_CO, ci0 = blockIdx.z * WG_CI;
    const int KK = P.KS * P.KS;
    const int ngroups = KK * 3;                          // (tap, co-block) groups, dealt round-robin to the waves

    f32x4 acc[WG_GROUPS][3];
#pragma unroll
    for (int g = 0; g < WG_GROUPS; ++g)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[g][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // pixel -> input offset table (tile-invariant: local row * stride * LWP + column * stride)
    for (int p = tid; p < P.PT; p += 256) {
        const int yl = p / P.WT, xq = p - yl * P.WT;
        poff[p] = yl < P.RT ? yl * (P.sparse ? 1 : P.stride) * P.LWP + xq * P.stride : 0;   // rounding pixels: any staged word
    }

    for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const int n = tile / P.tiles_per_img;
        const int tin = tile - n * P.tiles_per_img;
        const int yb = tin / P.tiles_x, xb = tin - yb * P.tiles_x;
        const int y0 = yb * P.RT, x0 = xb * P.WT;                      // first output row / column of the tile
        const int rows = min(P.RT, P.Ho - y0), cols = min(P.WT, P.Wo - x0);
        __syncthreads();                                               // previous tile fully consumed
        // ---- stage dY rows: [co][p], zero past the image rows / channels --------------------------------------
        {
            const otp_rsrc rdy = make_rsrc32(dy + ((size_t)n * P.dy_ctot + P.dy_coff) * P.Ho * P.Wo,
                                             (unsigned)P.Cout * (unsigned)(P.Ho * P.Wo) * 4u);
            for (int i = tid; i < WG_CO * P.PT; i += 256) {
                const int c = i / P.PT, p = i - c * P.PT;
                const int yl = p / P.WT, xq = p - yl * P.WT;
                const int co = co0 + c;
                const bool ok = co < P.Cout && yl < rows && xq < cols;
                dys[c * P.PS + p] = bload(rdy, ok ? (co * P.Ho * P.Wo + (y0 + yl) * P.Wo + x0 + xq) * 4 : -1, 0);
            }
        }
        // ---- stage the input rows with a zero halo: [ci][r][col], col 0 = image column -pad ------------------
        {
            const otp_rsrc rx = make_rsrc32(x + ((size_t)n * P.x_ctot + P.x_coff) * P.H * P.W,
                                            (unsigned)P.Cin * (unsigned)(P.H * P.W) * 4u);
            const int r0 = y0 * P.stride - P.pad;
            const int per_c = P.NRX * P.LWP;
            for (int i = tid; i < WG_CI * per_c; i += 256) {
                const int c = i / per_c, rem = i - c * per_c;
                const int r = rem / P.LWP, col = rem - r * P.LWP;
                const int yy = P.sparse ? (y0 + r % P.RT) * P.stride - P.pad + (r / P.RT) * P.dil : r0 + r;
                const int xx = x0 * P.stride - P.pad + col, ci = ci0 + c;
                const bool ok = ci < P.Cin && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
                xs[c * P.CSX + rem] = bload(rx, ok ? (ci * P.H * P.W + yy * P.W + xx) * 4 : -1, 0);
            }
        }
        __syncthreads();
        // ---- MFMA over the tile's pixels --------------------------------------------------------------------
        const int steps = (rows * P.WT + 3) >> 2;
#pragma unroll
        for (int g = 0; g < WG_GROUPS; ++g) {
            const int grp = wave + 4 * g;
            if (grp < ngroups) {
                const int tap = grp / 3, cb = grp - tap * 3;
                const int ti = tap / P.KS, tj = tap - ti * P.KS;
                const float* arow = dys + (cb * 16 + i16) * P.PS + kl;
                const float* brow = xs + i16 * P.CSX + ti * (P.sparse ? P.RT : P.dil) * P.LWP + tj * P.dil;
#pragma unroll 2
                for (int s = 0; s < steps; ++s) {
                    const int p = 4 * s + kl;
                    const float a = arow[4 * s];                       // dY[co][p]  (zero past the tile)
                    const int xo = poff[p < P.PT ? p : 0];
                    const float b0 = brow[xo], b1 = brow[16 * P.CSX + xo], b2 = brow[32 * P.CSX + xo];
                    acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[g][1], 0, 0, 0);
                    acc[g][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2, acc[g][2], 0, 0, 0);
                }
            }
        }
    }
    // ---- flush: D[row = co (kl*4 + r)][col = ci (i16)] -> dW (Cout, Cin, kh, kw), one atomic per weight ---------
#pragma unroll
    for (int g = 0; g < WG_GROUPS; ++g) {
        const int grp = wave + 4 * g;
        if (grp < ngroups) {
            const int tap = grp / 3, cb = grp - tap * 3;
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + cb * 16 + kl * 4 + r, ci = ci0 + b * 16 + i16;
                    if (co < P.Cout && ci < P.Cin) atomicAdd(&dw[((size_t)co * P.Cin + ci) * KK + tap], acc[g][b][r]);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm2d, training mode
// ------------------------------------------------------------------------------------------------
// per-channel partial sums of (a, a*b) [b == nullptr: (a, a*a)] over a slice of the N*HW elements, combined in fp64
// x / y may be channel slices of wider tensors (ctot / coff)
template <bool PAIR>
__global__ __launch_bounds__(256) void channel_sums_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ relu_y,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            double* __restrict__ part, int N, int C, int HW, int a_ctot,
                                                            int a_coff, int b_ctot, int b_coff, int y_ctot, int y_coff) {
    // PAIR == false: sums of a and a*a (forward statistics)
    // PAIR == true : g = a * (relu_y > 0 if relu_y), sums of g and g * (b - mean) * rstd (backward reductions)
    __shared__ double red[2][4];
    const int c = blockIdx.x, S = gridDim.y, s = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t total = (size_t)N * HW;
    float s0 = 0.f, s1 = 0.f;
    double d0 = 0.0, d1 = 0.0;
    int run = 0;
    const float mu = PAIR ? mean[c] : 0.f, rs = PAIR ? rstd[c] : 0.f;
    for (size_t i = (size_t)s * 256 + threadIdx.x; i < total; i += (size_t)S * 256) {
        const int n = (int)(i / HW), p = (int)(i - (size_t)n * HW);
        float v = a[((size_t)n * a_ctot + a_coff + c) * HW + p];
        if (PAIR) {
            if (relu_y && !(relu_y[((size_t)n * y_ctot + y_coff + c) * HW + p] > 0.f)) v = 0.f;
            const float xh = (b[((size_t)n * b_ctot + b_coff + c) * HW + p] - mu) * rs;
            s0 += v;
            s1 += v * xh;
        } else {
            s0 += v;
            s1 += v * v;
        }
        if (++run == 64) {                       // short fp32 runs, fp64 across runs
            d0 += s0; d1 += s1; s0 = s1 = 0.f; run = 0;
        }
    }
    d0 += s0; d1 += s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        d0 += __shfl_xor(d0, o, 64);
        d1 += __shfl_xor(d1, o, 64);
    }
    if (lane == 0) { red[0][wave] = d0; red[1][wave] = d1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)c * S + s) * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[((size_t)c * S + s) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// forward statistics: mean, 1/sqrt(biased var + eps), running-stat update (torch: unbiased variance, momentum m)
__global__ void bn_finish_stats_kernel(const double* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd,
                                       float* running_mean, float* running_var, int C, int S, double count, float eps,
                                       float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < S; ++i) { s += part[((size_t)c * S + i) * 2]; q += part[((size_t)c * S + i) * 2 + 1]; }
    const double mu = s / count;
    double var = q / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    if (running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// backward reductions -> dgamma = sum g*xhat, dbeta = sum g
__global__ void bn_finish_grads_kernel(const double* __restrict__ part, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int C, int S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < S; ++i) { s += part[((size_t)c * S + i) * 2]; q += part[((size_t)c * S + i) * 2 + 1]; }
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

// y = relu?( (x - mean) * rstd * gamma + beta (+ res) )
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* res, float* y, int C,
                                                        int HW, int relu, int x_ctot, int x_coff, int r_ctot, int r_coff,
                                                        int y_ctot, int y_coff, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const size_t r = i / HW;
        const int c = (int)(r % C), n = (int)(r / C);
        float v = (x[((size_t)n * x_ctot + x_coff + c) * HW + p] - mean[c]) * rstd[c] * gamma[c] + beta[c];
        if (res) v += res[((size_t)n * r_ctot + r_coff + c) * HW + p];
        if (relu) v = fmaxf(v, 0.f);
        y[((size_t)n * y_ctot + y_coff + c) * HW + p] = v;
    }
}

// g = dy * (y > 0 if relu);  dres = g;  dx = gamma * rstd * (g - dbeta / n - xhat * dgamma / n)
__global__ __launch_bounds__(256) void bn_backward_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y_relu,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta, float* dx, float* dres, int C, int HW,
    float inv_count, int dy_ctot, int dy_coff, int x_ctot, int x_coff, int y_ctot, int y_coff, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const size_t r = i / HW;
        const int c = (int)(r % C), n = (int)(r / C);
        float g = dy[((size_t)n * dy_ctot + dy_coff + c) * HW + p];
        if (y_relu && !(y_relu[((size_t)n * y_ctot + y_coff + c) * HW + p] > 0.f)) g = 0.f;
        const float xh = (x[((size_t)n * x_ctot + x_coff + c) * HW + p] - mean[c]) * rstd[c];
        if (dres) dres[i] = g;
        dx[i] = gamma[c] * rstd[c] * (g - dbeta[c] * inv_count - xh * dgamma[c] * inv_count);
    }
}

int sums_splits(int C, size_t count) {
    int s = 1;
    while ((long)C * s < 2048 && count / ((size_t)s * 2) >= 4096) s *= 2;
    return s;
}

int pad2(int n) {                                // smallest r >= n with r % 32 == 2
    int r = (n + 31) / 32 * 32 + 2;
    while (r - 32 >= n) r -= 32;
    return r;
}

}  // namespace

extern "C" int otp_conv2d_pack_weight_dgrad(const void* weight, void* wpacked, int Cout, int Cin, int kh, int kw,
                                            void* stream) {
    if (!weight || !wpacked || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0) return OTP_ERR_BAD_ARG;
    const int Cin16 = (Cin + 15) & ~15, total = kh * kw * Cout * Cin16;
    hipLaunchKernelGGL(pack_weight_dgrad_kernel, dim3(otp_ceil_div(total, 256) > 1024 ? 1024 : otp_ceil_div(total, 256)),
                       dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float*>(weight),
                       static_cast<float*>(wpacked), Cout, Cin, kh * kw, Cin16);
    return otp_launch_status();
}

extern "C" int otp_dilate(const void* in, void* out, int planes, int Hi, int Wi, int s, int H, int W, void* stream) {
    if (!in || !out || planes <= 0 || Hi <= 0 || Wi <= 0 || s <= 0 || H < (Hi - 1) * s + 1 || W < (Wi - 1) * s + 1)
        return OTP_ERR_BAD_ARG;
    const size_t total = (size_t)planes * H * W, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dilate_kernel, dim3(blocks > 8192 ? 8192 : (unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(in), static_cast<float*>(out), Hi, Wi, s,
                       H, W, total);
    return otp_launch_status();
}

extern "C" int otp_conv2d_wgrad(const void* x, const void* grad_out, void* grad_weight, int N, int Cin, int H, int W,
                                int Cout, int kh, int kw, int stride, int pad, int dil, int x_ctot, int x_coff,
                                int dy_ctot, int dy_coff, void* stream) {
    if (!x || !grad_out || !grad_weight || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || stride <= 0 ||
        pad < 0 || dil <= 0)
        return OTP_ERR_BAD_ARG;
    if (kh != kw || (kh != 1 && kh != 3)) return OTP_ERR_UNSUPPORTED;
    if (x_ctot < x_coff + Cin || dy_ctot < dy_coff + Cout) return OTP_ERR_BAD_ARG;
    WgradPlan P{};
    P.N = N; P.Cin = Cin; P.H = H; P.W = W; P.Cout = Cout; P.KS = kh; P.stride = stride; P.pad = pad; P.dil = dil;
    P.Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    P.Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    if (P.Ho <= 0 || P.Wo <= 0) return OTP_ERR_BAD_ARG;
    if ((long)Cin * H * W >= (1l << 29) || (long)Cout * P.Ho * P.Wo >= (1l << 29)) return OTP_ERR_UNSUPPORTED;
    P.x_ctot = x_ctot; P.x_coff = x_coff; P.dy_ctot = dy_ctot; P.dy_coff = dy_coff;
    // tile = RT output rows x WT output columns: whole rows when they are short, 128-column pieces of long rows
    // (the (B, C, 1, T) tensors of the ConvTransformers), as many rows as fit ~72 KB of LDS (two workgroups per CU)
    P.sparse = (dil > 1 && kh > 1) ? 1 : 0;
    P.WT = P.Wo <= 160 ? P.Wo : 128;
    P.tiles_x = otp_ceil_div(P.Wo, P.WT);
    P.LWP = (P.WT - 1) * stride + (kw - 1) * dil + 1;
    size_t lds = 0;
    for (int rt = P.Ho; rt >= 1; --rt) {
        P.RT = rt;
        P.PT = (rt * P.WT + 3) & ~3;
        P.PS = pad2(P.PT + 4);
        P.NRX = P.sparse ? kh * rt : (rt - 1) * stride + (kh - 1) * dil + 1;
        P.CSX = pad2(P.NRX * P.LWP + 4);
        lds = ((size_t)WG_CO * P.PS + (size_t)WG_CI * P.CSX + P.PT) * sizeof(float);
        if (lds <= 72 * 1024) break;
        if (rt == 1 && lds > OTP_LDS_LIMIT) return OTP_ERR_UNSUPPORTED;
    }
    P.tiles_per_img = otp_ceil_div(P.Ho, P.RT) * P.tiles_x;
    P.ntiles = N * P.tiles_per_img;
    const int gy = otp_ceil_div(Cout, WG_CO), gz = otp_ceil_div(Cin, WG_CI);
    int gx = 512 / (gy * gz);
    if (gx < 8) gx = 8;
    if (gx > P.ntiles) gx = P.ntiles;
    auto kern = conv_wgrad_kernel;
    OTP_ALLOW_BIG_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(256), lds, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(x), static_cast<const float*>(grad_out), static_cast<float*>(grad_weight), P);
    return otp_launch_status();
}

extern "C" size_t otp_bn_workspace(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return (size_t)C * sums_splits(C, (size_t)N * HW) * 2 * sizeof(double);
}

extern "C" int otp_bn_train_forward(const void* x, const void* gamma, const void* beta, const void* res, void* y,
                                    void* save_mean, void* save_rstd, void* running_mean, void* running_var,
                                    void* workspace, size_t workspace_bytes, int N, int C, int HW, float eps, float momentum,
                                    int relu, int x_ctot, int x_coff, int res_ctot, int res_coff, int y_ctot, int y_coff,
                                    void* stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !workspace || N <= 0 || C <= 0 || HW <= 0)
        return OTP_ERR_BAD_ARG;
    if (workspace_bytes < otp_bn_workspace(N, C, HW)) return OTP_ERR_WORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    const int S = sums_splits(C, (size_t)N * HW);
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    hipLaunchKernelGGL(channel_sums_kernel<false>, dim3(C, S), dim3(256), 0, st, f(x), nullptr, nullptr, nullptr, nullptr,
                       static_cast<double*>(workspace), N, C, HW, x_ctot, x_coff, 0, 0, 0, 0);
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(otp_ceil_div(C, 64)), dim3(64), 0, st,
                       static_cast<const double*>(workspace), static_cast<float*>(save_mean), static_cast<float*>(save_rstd),
                       static_cast<float*>(running_mean), static_cast<float*>(running_var), C, S, (double)N * HW, eps,
                       momentum);
    const size_t total = (size_t)N * C * HW, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks > 16384 ? 16384 : (unsigned)blocks), dim3(256), 0, st, f(x),
                       f(save_mean), f(save_rstd), f(gamma), f(beta), f(res), static_cast<float*>(y), C, HW, relu, x_ctot,
                       x_coff, res_ctot, res_coff, y_ctot, y_coff, total);
    return otp_launch_status();
}

extern "C" int otp_bn_train_backward(const void* grad_y, const void* x, const void* y_relu, const void* save_mean,
                                     const void* save_rstd, const void* gamma, void* grad_x, void* grad_res,
                                     void* grad_gamma, void* grad_beta, void* workspace, size_t workspace_bytes, int N,
                                     int C, int HW, int dy_ctot, int dy_coff, int x_ctot, int x_coff, int y_ctot,
                                     int y_coff, void* stream) {
    if (!grad_y || !x || !save_mean || !save_rstd || !gamma || !grad_x || !grad_gamma || !grad_beta || !workspace)
        return OTP_ERR_BAD_ARG;
    if (N <= 0 || C <= 0 || HW <= 0) return OTP_ERR_BAD_ARG;
    if (workspace_bytes < otp_bn_workspace(N, C, HW)) return OTP_ERR_WORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    const int S = sums_splits(C, (size_t)N * HW);
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    hipLaunchKernelGGL(channel_sums_kernel<true>, dim3(C, S), dim3(256), 0, st, f(grad_y), f(x), f(y_relu), f(save_mean),
                       f(save_rstd), static_cast<double*>(workspace), N, C, HW, dy_ctot, dy_coff, x_ctot, x_coff, y_ctot,
                       y_coff);
    hipLaunchKernelGGL(bn_finish_grads_kernel, dim3(otp_ceil_div(C, 64)), dim3(64), 0, st,
                       static_cast<const double*>(workspace), static_cast<float*>(grad_gamma), static_cast<float*>(grad_beta),
                       C, S);
    const size_t total = (size_t)N * C * HW, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(blocks > 16384 ? 16384 : (unsigned)blocks), dim3(256), 0, st,
                       f(grad_y), f(x), f(y_relu), f(save_mean), f(save_rstd), f(gamma), f(grad_gamma), f(grad_beta),
                       static_cast<float*>(grad_x), static_cast<float*>(grad_res), C, HW, 1.f / ((float)N * (float)HW),
                       dy_ctot, dy_coff, x_ctot, x_coff, y_ctot, y_coff, total);
    return otp_launch_status();
}

// sums over (N, HW) per channel of a (N, ctot, HW) slice: bias gradients.  out (C) float.
extern "C" int otp_channel_sum(const void* a, void* out, void* workspace, size_t workspace_bytes, int N, int C, int HW,
                               int a_ctot, int a_coff, void* stream) {
    if (!a || !out || !workspace || N <= 0 || C <= 0 || HW <= 0) return OTP_ERR_BAD_ARG;
    if (workspace_bytes < otp_bn_workspace(N, C, HW) + (size_t)C * sizeof(float)) return OTP_ERR_WORKSPACE;
    auto st = static_cast<hipStream_t>(stream);
    const int S = sums_splits(C, (size_t)N * HW);
    double* part = static_cast<double*>(workspace);
    float* scratch = reinterpret_cast<float*>(part + (size_t)C * S * 2);      // the unused sum of squares
    hipLaunchKernelGGL(channel_sums_kernel<false>, dim3(C, S), dim3(256), 0, st, static_cast<const float*>(a), nullptr,
                       nullptr, nullptr, nullptr, part, N, C, HW, a_ctot, a_coff, 0, 0, 0, 0);
    hipLaunchKernelGGL(bn_finish_grads_kernel, dim3(otp_ceil_div(C, 64)), dim3(64), 0, st, part, scratch,
                       static_cast<float*>(out), C, S);
    return otp_launch_status();
}
