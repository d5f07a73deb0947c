// OTPose glue between the backbone, the temporal encoders and the warp (model/OTPose.py:317-359):
// elementwise / small-reduction kernels, one thread per heat-map pixel walking the 17 joints, so all
// accesses are coalesced along the pixel axis and each input is read once.
#include "common.h"

namespace {

// rough (5B, J, HW): frames [cur | prev | next | pprev | nnext] in blocks of B (OTPose.py:317-321)
// F = frames of the window (5: reference; 7: the BASELINE configs[4] extension, oracle window_maps)
__global__ void glue_total_kernel(const float* __restrict__ rough, float* __restrict__ total,
                                  float* __restrict__ squeezed, float* __restrict__ inter,
                                  float* __restrict__ flow_in, const float* __restrict__ pe, int B, int J, int HW, int F) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (p >= HW) return;
    const size_t fs = (size_t)B * J * HW;                  // frame stride
    const size_t base = (size_t)b * J * HW + p;
    float sq = 0.f;
    for (int j = 0; j < J; ++j) {
        const size_t i = base + (size_t)j * HW;
        // same association as the reference: (((cur + prev) + next) + pprev) + nnext  (OTPose.py:324)
        float t = rough[i];
        for (int f = 1; f < F; ++f) t += rough[(size_t)f * fs + i];
        total[i] = t;
        flow_in[i] = t + pe[(size_t)j * HW + p];
        sq += t;                                            // torch.sum over joints (OTPose.py:325)
    }
    for (int j = 0; j < J; ++j) {
        const size_t i = base + (size_t)j * HW;
        squeezed[i] = sq;
        inter[i] = total[i] * sq;                           // OTPose.py:330
    }
}

// x1/x2 channel = joint*M + feature (OTPose.py:356-359); positional table added here (ConvVideoTransformer.py:144/155).
// R = rings of the window (frames cur, prev_1, next_1, ..., prev_R, next_R); M = 8 (R = 2, reference) or 12 (R = 3).
template <int R>
__global__ void glue_stack_kernel(const float* __restrict__ rough, const float* __restrict__ margin,
                                  const float* __restrict__ squeezed, const float* __restrict__ inter,
                                  const float* __restrict__ ctx, const float* __restrict__ pe1,
                                  const float* __restrict__ pe2, float* __restrict__ x1, float* __restrict__ x2,
                                  float* __restrict__ prev_b_out, int B, int J, int HW) {
    constexpr int NB = R == 2 ? 3 : 5, M = 2 + 2 * NB;       // "b" maps per encoder, stacked maps per joint
    // one thread per (clip, joint, pixel): 7344 workgroups at cfg2 instead of 432 threads walking the 17 joints one after the
    // other through dependent loads (145 -> 40 us)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y / J, j = blockIdx.y - b * J;
    if (p >= HW) return;
    const size_t fs = (size_t)B * J * HW;
    const size_t base = (size_t)b * J * HW + p;
    float mg[2 * R];
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) mg[k] = margin[b * 2 * R + k] + 1.f;
    const float sq = squeezed[base];
    {
        const size_t i = base + (size_t)j * HW;
        const float cur = rough[i];
        float prev[R], next[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {                       // OTPose.py:339-342
            prev[r] = rough[(size_t)(1 + 2 * r) * fs + i] / mg[2 * r];
            next[r] = rough[(size_t)(2 + 2 * r) * fs + i] / mg[2 * r + 1];
        }
        float sp = prev[0], sn = next[0];
#pragma unroll
        for (int r = 1; r < R; ++r) { sp += prev[r]; sn += next[r]; }
        const float prev_b = cur + sp, next_b = cur + sn;   // OTPose.py:345-346: cur + (prev + pprev)
        float sym[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sym[r] = cur + (next[r] + prev[r]);      // close_b, far_b (:347-349), wide_b
        float b1[NB], b2[NB];
        b1[0] = prev_b;
        b2[0] = next_b;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            b1[1 + r] = sym[R - 1 - r];                     // prev_b, far_b, close_b          (:356)
            b2[1 + r] = sym[r];                             // next_b, close_b, far_b          (:358)
        }
        if (R == 3) {
            b1[NB - 1] = cur + (prev[1] + prev[2]);
            b2[NB - 1] = cur + (next[1] + next[2]);
        }
        const float in = inter[i], cx = ctx[i];
        prev_b_out[i] = prev_b;
        const size_t o = ((size_t)b * J * M + (size_t)j * M) * HW + p;
        const size_t pj = (size_t)j * M * HW + p;
        float f1[M], f2[M];
        f1[0] = f2[0] = in;
        f1[1] = f2[1] = cx;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            f1[2 + k] = b1[k];
            f2[2 + k] = b2[k];
            f1[2 + NB + k] = b1[k] * sq;                    // :351-354
            f2[2 + NB + k] = b2[k] * sq;
        }
#pragma unroll
        for (int f = 0; f < M; ++f) {
            x1[o + (size_t)f * HW] = f1[f] + pe1[pj + (size_t)f * HW];
            x2[o + (size_t)f * HW] = f2[f] + pe2[pj + (size_t)f * HW];
        }
    }
}

__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, float alpha, float beta, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = beta == 0.f ? alpha * x[i] : alpha * x[i] + beta * y[i];
}


// ---- heat-map decode: get_max_preds + the quarter-pixel shift of get_final_preds (utils/heatmap.py:143-171, 108-125) ----
// One wave per (sample, joint) plane.  argmax = FIRST maximum (numpy argmax), carried through the wavefront
// reduction as a (value, index) pair; a NaN anywhere makes the first NaN the arg-maximum, as numpy does.
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > bv || (v == bv && i < bi);
}

__global__ __launch_bounds__(256) void heatmap_decode_kernel(const float* __restrict__ hm, float* __restrict__ preds,
                                                              float* __restrict__ maxvals, const float* __restrict__ center,
                                                              const float* __restrict__ scale, int NJ, int J, int H, int W,
                                                              int refine) {
    const int lane = threadIdx.x & 63, plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= NJ) return;
    const int HW = H * W;
    const float* p = hm + (size_t)plane * HW;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < HW; i += 64) {
        const float v = p[i];
        if (bi == 0x7fffffff || better(v, i, bv, bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || better(ov, oi, bv, bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        const float m = bv > 0.f ? 1.f : 0.f;                      // pred_mask = maxvals > 0 (NaN -> 0)
        float x = (float)(bi % W) * m, y = (float)(bi / W) * m;     // idx % width, floor(idx / width)
        if (refine) {
            const int px = (int)floorf(x + 0.5f), py = (int)floorf(y + 0.5f);
            if (1 < px && px < W - 1 && 1 < py && py < H - 1) {      // the reference's strict bounds (heatmap.py:120)
                const float dx = p[py * W + px + 1] - p[py * W + px - 1];
                const float dy = p[(py + 1) * W + px] - p[(py - 1) * W + px];
                x += dx > 0.f ? 0.25f : (dx < 0.f ? -0.25f : 0.f);    // np.sign(diff) * .25
                y += dy > 0.f ? 0.25f : (dy < 0.f ? -0.25f : 0.f);
            }
        }
        if (center && scale) {
            // transform_preds with rot = 0: cv2.getAffineTransform of the three points of get_affine_transform(inv=1)
            // (utils/transform.py:76-105) is the similarity  src = center + (dst - (W/2, H/2)) * (200 * scale_x / W)
            const int n = plane / J;
            const float k = 200.f * scale[2 * n] / (float)W;
            x = center[2 * n] + (x - 0.5f * (float)W) * k;
            y = center[2 * n + 1] + (y - 0.5f * (float)H) * k;
        }
        preds[2 * plane] = x;
        preds[2 * plane + 1] = y;
        maxvals[plane] = bv;
    }
}

// ---- nearest-upsample accumulate: out = act(res + up_f(low)) (HRNet fuse layers, model/HRNet.py:426-439,488-494) --------
// For f >= 4 the upsampled tensor is f*f >= 16 times larger than the conv result, so the accumulate is a streaming
// kernel of its own (one float4 of out / res per thread, 256 B per wave-instruction) instead of a conv epilogue.
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ low, const float* res,
                                                            float* out, int C, int Hl, int Wl, int f, int relu,
                                                            int low_ctot, int low_coff, int res_ctot, int res_coff,
                                                            int out_ctot, int out_coff, size_t total4) {
    const int Wh = Wl * f, Hh = Hl * f, Wh4 = Wh >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % Wh4);
        size_t r = i / Wh4;
        const int y = (int)(r % Hh);
        r /= Hh;
        const int c = (int)(r % C), n = (int)(r / C);
        const size_t hi = ((size_t)y * Wh + 4 * x4);
        const float* lrow = low + (((size_t)n * low_ctot + low_coff + c) * Hl + y / f) * Wl;
        const otp_f32x4 rv = *reinterpret_cast<const otp_f32x4*>(res + ((size_t)n * res_ctot + res_coff + c) * Hh * Wh + hi);
        otp_f32x4 o;
        if (f >= 4) {
            const float l = lrow[(4 * x4) / f];
            o = rv + l;
        } else {                                                     // f == 2: two low-resolution pixels per float4
            const float l0 = lrow[2 * x4], l1 = lrow[2 * x4 + 1];
            o = rv + (otp_f32x4){l0, l0, l1, l1};
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<otp_f32x4*>(out + ((size_t)n * out_ctot + out_coff + c) * Hh * Wh + hi) = o;
    }
}

// out = act(res + up_f0(low0) + up_f1(low1) (+ up_f2(low2))), added in that order: a whole fuse row's upsampled terms in one
// pass over the high-resolution tensor instead of one read-modify-write pass per term (dense lows, f_k >= 2, W % 4 == 0)
struct UpMulti {
    const float* low[3];
    int f[3];
    int n;
};
__global__ __launch_bounds__(256) void upsample_add_multi_kernel(UpMulti U, const float* res, float* out, int C, int Hh, int Wh,
                                                                  int relu, int res_ctot, int res_coff, int out_ctot,
                                                                  int out_coff, size_t total4) {
    const int Wh4 = Wh >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % Wh4);
        size_t r = i / Wh4;
        const int y = (int)(r % Hh);
        r /= Hh;
        const int c = (int)(r % C), n = (int)(r / C);
        const size_t hi = ((size_t)y * Wh + 4 * x4);
        otp_f32x4 o = *reinterpret_cast<const otp_f32x4*>(res + ((size_t)n * res_ctot + res_coff + c) * Hh * Wh + hi);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < U.n) {
                const int f = U.f[k], Wl = Wh / f, Hl = Hh / f;
                const float* lrow = U.low[k] + (((size_t)n * C + c) * Hl + y / f) * Wl;
                if (f >= 4) {
                    o = o + lrow[(4 * x4) / f];
                } else {
                    const float l0 = lrow[2 * x4], l1 = lrow[2 * x4 + 1];
                    o = o + (otp_f32x4){l0, l0, l1, l1};
                }
            }
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<otp_f32x4*>(out + ((size_t)n * out_ctot + out_coff + c) * Hh * Wh + hi) = o;
    }
}

// the same, one element per thread (rows that are not whole float4)
__global__ __launch_bounds__(256) void upsample_add_scalar_kernel(const float* __restrict__ low, const float* res, float* out,
                                                                   int C, int Hl, int Wl, int f, int relu, int low_ctot,
                                                                   int low_coff, int res_ctot, int res_coff, int out_ctot,
                                                                   int out_coff, size_t total) {
    const int Wh = Wl * f, Hh = Hl * f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wh);
        size_t r = i / Wh;
        const int y = (int)(r % Hh);
        r /= Hh;
        const int c = (int)(r % C), n = (int)(r / C);
        const size_t hi = (size_t)y * Wh + x;
        float v = res[((size_t)n * res_ctot + res_coff + c) * Hh * Wh + hi] +
                  low[(((size_t)n * low_ctot + low_coff + c) * Hl + y / f) * Wl + x / f];
        if (relu) v = fmaxf(v, 0.f);
        out[((size_t)n * out_ctot + out_coff + c) * Hh * Wh + hi] = v;
    }
}

// backward of out = relu?(res + up_f(low)): g = dy * (out > 0 if relu); grad_res = g; grad_low = sum of g over each f x f cell
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out_relu,
                                                                float* __restrict__ dres, float* __restrict__ dlow, int Hl,
                                                                int Wl, int f, size_t total_low) {
    const int Wh = Wl * f, Hh = Hl * f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_low; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wl);
        size_t r = i / Wl;
        const int y = (int)(r % Hl);
        const size_t plane = r / Hl;
        float s = 0.f;
        for (int dyy = 0; dyy < f; ++dyy)
            for (int dxx = 0; dxx < f; ++dxx) {
                const size_t o = (plane * Hh + (size_t)y * f + dyy) * Wh + (size_t)x * f + dxx;
                float g = dy[o];
                if (out_relu && !(out_relu[o] > 0.f)) g = 0.f;
                if (dres) dres[o] = g;
                s += g;
            }
        dlow[i] = s;
    }
}


// ------------------------------------------------------------------------------------------------
// PCK accuracy of the training / validation loop (utils/evaluate.py:352-415: calc_dists, dist_acc, accuracy), from the
// argmax coordinates of the predicted and the target heat-maps: per joint the fraction of samples whose normalised
// distance is below thr, samples whose target argmax has x <= 1 or y <= 1 ignored; acc[0] = mean over the joints that
// have a valid sample.  The reference divides (x, y) by (H/10, W/10) in that order (evaluate.py:396, 357-358) - kept.
// One workgroup; a thread per joint walks the batch; float64 like the numpy reference.
__global__ __launch_bounds__(256) void pck_accuracy_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                            float* __restrict__ acc, int* __restrict__ cnt, int N, int J,
                                                            int H, int W, float thr) {
    __shared__ double s_acc[256];
    __shared__ int s_ok[256];
    const double nx = (double)H / 10.0, ny = (double)W / 10.0;
    double avg = 0.0;
    int c = 0;
    for (int j0 = 0; j0 < J; j0 += 256) {
        const int j = j0 + threadIdx.x;
        double a = -1.0;
        if (j < J) {
            int valid = 0, hit = 0;
            for (int n = 0; n < N; ++n) {
                const float tx = tgt[((size_t)n * J + j) * 2], ty = tgt[((size_t)n * J + j) * 2 + 1];
                if (tx > 1.f && ty > 1.f) {
                    const double dx = (double)pred[((size_t)n * J + j) * 2] / nx - (double)tx / nx;
                    const double dy = (double)pred[((size_t)n * J + j) * 2 + 1] / ny - (double)ty / ny;
                    ++valid;
                    if (sqrt(dx * dx + dy * dy) < (double)thr) ++hit;
                }
            }
            if (valid > 0) a = (double)hit / (double)valid;
            acc[j + 1] = (float)a;
        }
        s_acc[threadIdx.x] = a >= 0.0 ? a : 0.0;
        s_ok[threadIdx.x] = a >= 0.0 ? 1 : 0;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int t = 0; t < 256 && j0 + t < J; ++t) { avg += s_acc[t]; c += s_ok[t]; }   // joint order, like the reference loop
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        acc[0] = c ? (float)(avg / c) : 0.f;
        *cnt = c;
    }
}

// ------------------------------------------------------------------------------------------------
// input assembly (the step before the path: dataset/PoseTrackDataset.py:397-406 + utils/transform.py:7-15 ToTensor /
// Normalize per frame, script/Common.py:117 channel concat): uint8 HWC frames (B, F, H, W, 3) ->
// float32 (B, 3F, H, W) with v = (u8 / 255 - mean_c) / std_c, the exact float32 operation order of torchvision.
// A thread converts 4 pixels: three aligned dword loads, one float4 store per colour plane.
__global__ __launch_bounds__(256) void frames_u8_kernel(const uint32_t* __restrict__ in, float* __restrict__ out, int HW4,
                                                         size_t total4, float m0, float m1, float m2, float s0, float s1,
                                                         float s2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t img = i / HW4;                                      // (b, f) image
        const int q = (int)(i - img * HW4);                              // group of 4 pixels
        const uint32_t w0 = in[i * 3], w1 = in[i * 3 + 1], w2 = in[i * 3 + 2];
        // bytes: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        const float r[4] = {(float)(w0 & 255), (float)(w0 >> 24), (float)((w1 >> 16) & 255), (float)((w2 >> 8) & 255)};
        const float g[4] = {(float)((w0 >> 8) & 255), (float)(w1 & 255), (float)(w1 >> 24), (float)((w2 >> 16) & 255)};
        const float b[4] = {(float)((w0 >> 16) & 255), (float)((w1 >> 8) & 255), (float)(w2 & 255), (float)(w2 >> 24)};
        otp_f32x4 vr, vg, vb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vr[e] = (r[e] / 255.f - m0) / s0;
            vg[e] = (g[e] / 255.f - m1) / s1;
            vb[e] = (b[e] / 255.f - m2) / s2;
        }
        float* o = out + img * 3 * (size_t)HW4 * 4 + (size_t)q * 4;
        *reinterpret_cast<otp_f32x4*>(o) = vr;
        *reinterpret_cast<otp_f32x4*>(o + (size_t)HW4 * 4) = vg;
        *reinterpret_cast<otp_f32x4*>(o + (size_t)HW4 * 8) = vb;
    }
}

}  // namespace

extern "C" int otp_glue_total_n(const void* rough, void* total, void* squeezed, void* inter, void* flow_in, const void* pe,
                                int B, int J, int HW, int F, void* stream) {
    if (!rough || !total || !squeezed || !inter || !flow_in || !pe) return OTP_ERR_BAD_ARG;
    if (B <= 0 || J <= 0 || HW <= 0 || F < 1) return OTP_ERR_BAD_ARG;
    hipLaunchKernelGGL(glue_total_kernel, dim3(otp_ceil_div(HW, 256), B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(rough), static_cast<float*>(total), static_cast<float*>(squeezed),
                       static_cast<float*>(inter), static_cast<float*>(flow_in), static_cast<const float*>(pe), B, J, HW, F);
    return otp_launch_status();
}

extern "C" int otp_glue_total(const void* rough, void* total, void* squeezed, void* inter, void* flow_in, const void* pe,
                              int B, int J, int HW, void* stream) {
    return otp_glue_total_n(rough, total, squeezed, inter, flow_in, pe, B, J, HW, 5, stream);
}

extern "C" int otp_glue_stack_n(const void* rough, const void* margin, const void* squeezed, const void* inter,
                                const void* ctx, const void* pe1, const void* pe2, void* x1, void* x2, void* prev_b,
                                int B, int J, int HW, int F, void* stream) {
    if (!rough || !margin || !squeezed || !inter || !ctx || !pe1 || !pe2 || !x1 || !x2 || !prev_b) return OTP_ERR_BAD_ARG;
    if (B <= 0 || J <= 0 || HW <= 0) return OTP_ERR_BAD_ARG;
    if (F != 5 && F != 7) return OTP_ERR_UNSUPPORTED;
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    const dim3 grid(otp_ceil_div(HW, 256), B * J);
    if (F == 5)
        hipLaunchKernelGGL(glue_stack_kernel<2>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), f(rough), f(margin),
                           f(squeezed), f(inter), f(ctx), f(pe1), f(pe2), static_cast<float*>(x1), static_cast<float*>(x2),
                           static_cast<float*>(prev_b), B, J, HW);
    else
        hipLaunchKernelGGL(glue_stack_kernel<3>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), f(rough), f(margin),
                           f(squeezed), f(inter), f(ctx), f(pe1), f(pe2), static_cast<float*>(x1), static_cast<float*>(x2),
                           static_cast<float*>(prev_b), B, J, HW);
    return otp_launch_status();
}

extern "C" int otp_glue_stack(const void* rough, const void* margin, const void* squeezed, const void* inter,
                              const void* ctx, const void* pe1, const void* pe2, void* x1, void* x2, void* prev_b,
                              int B, int J, int HW, void* stream) {
    return otp_glue_stack_n(rough, margin, squeezed, inter, ctx, pe1, pe2, x1, x2, prev_b, B, J, HW, 5, stream);
}

extern "C" int otp_axpby(const void* x, void* y, float alpha, float beta, size_t n, void* stream) {
    if (!x || !y) return OTP_ERR_BAD_ARG;
    if (n == 0) return OTP_OK;
    size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks > 2048 ? 2048 : (unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(x), static_cast<float*>(y), alpha, beta, n);
    return otp_launch_status();
}

extern "C" int otp_upsample_add(const void* low, const void* res, void* out, int N, int C, int Hl, int Wl, int f, int relu,
                                int low_ctot, int low_coff, int res_ctot, int res_coff, int out_ctot, int out_coff,
                                void* stream) {
    if (!low || !res || !out || N <= 0 || C <= 0 || Hl <= 0 || Wl <= 0) return OTP_ERR_BAD_ARG;
    if (f < 1) return OTP_ERR_BAD_ARG;
    if (low_ctot < low_coff + C || res_ctot < res_coff + C || out_ctot < out_coff + C) return OTP_ERR_BAD_ARG;
    const bool vec = (f == 2 || f == 4 || f == 8 || f == 16) && ((Wl * f) & 3) == 0 &&
                     !(reinterpret_cast<uintptr_t>(res) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15);
    if (!vec) {
        const size_t total = (size_t)N * C * Hl * f * Wl * f, nb = (total + 255) / 256;
        hipLaunchKernelGGL(upsample_add_scalar_kernel, dim3(nb > 8192 ? 8192 : (unsigned)nb), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<const float*>(low), static_cast<const float*>(res),
                           static_cast<float*>(out), C, Hl, Wl, f, relu, low_ctot, low_coff, res_ctot, res_coff, out_ctot,
                           out_coff, total);
        return otp_launch_status();
    }
    const size_t total4 = (size_t)N * C * Hl * f * (Wl * f / 4);
    const size_t blocks = (total4 + 255) / 256;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(blocks > 8192 ? 8192 : (unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(low), static_cast<const float*>(res),
                       static_cast<float*>(out), C, Hl, Wl, f, relu, low_ctot, low_coff, res_ctot, res_coff, out_ctot,
                       out_coff, total4);
    return otp_launch_status();
}

extern "C" int otp_upsample_add_multi(const void* const* lows, const int* factors, int nlow, const void* res, void* out, int N,
                                      int C, int Hh, int Wh, int relu, int res_ctot, int res_coff, int out_ctot, int out_coff,
                                      void* stream) {
    if (!lows || !factors || !res || !out || nlow < 1 || nlow > 3 || N <= 0 || C <= 0 || Hh <= 0 || Wh <= 0) return OTP_ERR_BAD_ARG;
    if (Wh % 4 || res_ctot < res_coff + C || out_ctot < out_coff + C) return OTP_ERR_UNSUPPORTED;
    UpMulti U{};
    U.n = nlow;
    for (int k = 0; k < nlow; ++k) {
        const int f = factors[k];
        if (!lows[k]) return OTP_ERR_BAD_ARG;
        if (f < 2 || (f & (f - 1)) || Hh % f || Wh % f) return OTP_ERR_UNSUPPORTED;
        U.low[k] = static_cast<const float*>(lows[k]);
        U.f[k] = f;
    }
    if ((reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out)) & 15 || ((size_t)Hh * Wh) % 4) return OTP_ERR_UNSUPPORTED;
    const size_t total4 = (size_t)N * C * Hh * (Wh / 4);
    const size_t blocks = (total4 + 255) / 256;
    hipLaunchKernelGGL(upsample_add_multi_kernel, dim3(blocks > 8192 ? 8192 : (unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), U, static_cast<const float*>(res), static_cast<float*>(out), C, Hh, Wh,
                       relu, res_ctot, res_coff, out_ctot, out_coff, total4);
    return otp_launch_status();
}

extern "C" int otp_upsample_add_backward(const void* grad_out, const void* out_relu, void* grad_res, void* grad_low,
                                         int planes, int Hl, int Wl, int f, void* stream) {
    if (!grad_out || !grad_low || planes <= 0 || Hl <= 0 || Wl <= 0 || f <= 0) return OTP_ERR_BAD_ARG;
    const size_t total = (size_t)planes * Hl * Wl, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(blocks > 8192 ? 8192 : (unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(grad_out),
                       static_cast<const float*>(out_relu), static_cast<float*>(grad_res), static_cast<float*>(grad_low), Hl, Wl,
                       f, total);
    return otp_launch_status();
}

extern "C" int otp_heatmap_decode(const void* heatmaps, void* preds, void* maxvals, const void* center, const void* scale,
                                  int N, int J, int H, int W, int refine, void* stream) {
    if (!heatmaps || !preds || !maxvals || N <= 0 || J <= 0 || H <= 0 || W <= 0) return OTP_ERR_BAD_ARG;
    if ((center == nullptr) != (scale == nullptr)) return OTP_ERR_BAD_ARG;
    const int NJ = N * J;
    hipLaunchKernelGGL(heatmap_decode_kernel, dim3(otp_ceil_div(NJ, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(heatmaps), static_cast<float*>(preds), static_cast<float*>(maxvals),
                       static_cast<const float*>(center), static_cast<const float*>(scale), NJ, J, H, W, refine);
    return otp_launch_status();
}

extern "C" int otp_pck_accuracy(const void* pred_coords, const void* target_coords, void* acc, void* cnt, int N, int J,
                                int H, int W, float thr, void* stream) {
    if (!pred_coords || !target_coords || !acc || !cnt || N <= 0 || J <= 0 || H <= 0 || W <= 0) return OTP_ERR_BAD_ARG;
    hipLaunchKernelGGL(pck_accuracy_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(pred_coords), static_cast<const float*>(target_coords),
                       static_cast<float*>(acc), static_cast<int*>(cnt), N, J, H, W, thr);
    return otp_launch_status();
}

extern "C" int otp_frames_u8_to_clip(const void* frames_u8, void* out, int B, int F, int H, int W, float mean_r,
                                     float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream) {
    if (!frames_u8 || !out || B <= 0 || F <= 0 || H <= 0 || W <= 0) return OTP_ERR_BAD_ARG;
    if (((size_t)H * W) % 4 != 0 || std_r == 0.f || std_g == 0.f || std_b == 0.f) return OTP_ERR_UNSUPPORTED;
    const int HW4 = H * W / 4;
    const size_t total4 = (size_t)B * F * HW4, blocks = (total4 + 255) / 256;
    hipLaunchKernelGGL(frames_u8_kernel, dim3(blocks > 16384 ? 16384 : (unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint32_t*>(frames_u8), static_cast<float*>(out),
                       HW4, total4, mean_r, mean_g, mean_b, std_r, std_g, std_b);
    return otp_launch_status();
}

extern "C" int otp_version(void) { return 1; }
