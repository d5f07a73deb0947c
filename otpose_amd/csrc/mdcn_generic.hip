// General form of the (modulated) deformable convolution - every case the reference's operator accepts and the fused
// fp32 kernels of mdcn.hip do not specialise:
//   * storage types f64 / f32 / f16 (the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF,
//     thirdparty/deform_conv/src/deform_conv_cuda_kernel.cu:719-737, 751-769, 784-805) and bf16, with fp32 arithmetic
//     (fp64 for f64);
//   * any kernel size, any Cout / groups / deformable groups in BOTH directions, independent stride / padding / dilation
//     per axis (deform_conv_cuda.cpp:474-480, 551-558 take *_h and *_w);
//   * mask == NULL: DCN v1 (deform_conv_cuda.cpp:148-472) - no mask stream is read, no mask gradient is written.
// Same sampling rules as mdcn.hip (open interval (-1, H) x (-1, W), corners outside the image contribute zero,
// kernel.cu:403-432, 549-556) and the same structure: the input plane is staged into LDS with a zero border, grad_x is
// accumulated in an LDS copy of the plane.  It favours generality over speed: weights are read through the scalar
// cache, Cout is walked in chunks of 16, and the backward recomputes the sample once per chunk of output channels.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

constexpr int PADL = 4;
constexpr int THREADS = 256;

struct GGeom {
    int N, C, H, W, Co, K, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo, P;
    int dg, cpg_dg, cin_g, cout_g;
    int LW, plane;
};

template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };

template <typename T> __device__ __forceinline__ typename Acc<T>::type ld(const T* p, size_t i) { return (typename Acc<T>::type)p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p, size_t i) { return __half2float(p[i]); }
template <> __device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16* p, size_t i) { return __bfloat162float(p[i]); }
template <typename T, typename A> __device__ __forceinline__ void st(T* p, size_t i, A v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void st<__half, float>(__half* p, size_t i, float v) { p[i] = __float2half(v); }
template <> __device__ __forceinline__ void st<__hip_bfloat16, float>(__hip_bfloat16* p, size_t i, float v) { p[i] = __float2bfloat16(v); }

template <typename A>
struct GTap {
    A v1, v2, v3, v4, lh, lw;
    int addr;
    bool inside;
};

template <typename A>
__device__ __forceinline__ GTap<A> gtap(const A* __restrict__ plane, A h_im, A w_im, const GGeom& g) {
    GTap<A> t;
    t.inside = (h_im > (A)-1) && (w_im > (A)-1) && (h_im < (A)g.H) && (w_im < (A)g.W);
    A hc = h_im < (A)-2 ? (A)-2 : (h_im > (A)(g.H + 1) ? (A)(g.H + 1) : h_im);
    A wc = w_im < (A)-2 ? (A)-2 : (w_im > (A)(g.W + 1) ? (A)(g.W + 1) : w_im);
    if (!(hc == hc)) hc = (A)-2;                               // NaN coordinates: outside
    if (!(wc == wc)) wc = (A)-2;
    const A hf = floor(hc), wf = floor(wc);
    t.lh = hc - hf;
    t.lw = wc - wf;
    const int hl = min(max((int)hf, -1), g.H - 1), wl = min(max((int)wf, -1), g.W - 1);
    t.addr = (hl + 1) * g.LW + wl + PADL;
    t.v1 = plane[t.addr];
    t.v2 = plane[t.addr + 1];
    t.v3 = plane[t.addr + g.LW];
    t.v4 = plane[t.addr + g.LW + 1];
    return t;
}

template <typename T, typename A>
__device__ __forceinline__ void gstage(A* __restrict__ plane, const T* __restrict__ src, const GGeom& g, int tid) {
    for (int i = tid; i < g.H * g.W; i += THREADS) {
        const int y = i / g.W, xx = i - y * g.W;
        plane[(y + 1) * g.LW + PADL + xx] = ld<T>(src, i);
    }
}

// grid (ceil(P / 256), N, ceil(Cout / 16)); LDS: one padded plane of A
template <typename T>
__global__ __launch_bounds__(THREADS) void mdcn_generic_fwd_kernel(const T* __restrict__ x, const T* __restrict__ off,
                                                                    const T* __restrict__ msk, const T* __restrict__ w,
                                                                    const T* __restrict__ bias, T* __restrict__ out, GGeom g,
                                                                    float alpha, float beta) {
    typedef typename Acc<T>::type A;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    A* plane = reinterpret_cast<A*>(smem_raw);
    const int tid = threadIdx.x, n = blockIdx.y, co0 = blockIdx.z * 16;
    const int p = blockIdx.x * THREADS + tid;
    const bool valid = p < g.P;
    const int pp = valid ? p : 0;
    const int ho = pp / g.Wo, wo = pp - ho * g.Wo;
    A acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = (A)0;
    for (int i = tid; i < g.plane; i += THREADS) plane[i] = (A)0;
    const T* xn = x + (size_t)n * g.C * g.H * g.W;
    const T* offn = off + (size_t)n * g.dg * 2 * g.K * g.P;
    const T* mskn = msk ? msk + (size_t)n * g.dg * g.K * g.P : nullptr;
    for (int c = 0; c < g.C; ++c) {
        __syncthreads();
        gstage<T, A>(plane, xn + (size_t)c * g.H * g.W, g, tid);
        __syncthreads();
        const int grp = c / g.cpg_dg, wgrp = c / g.cin_g, cl = c - wgrp * g.cin_g;
        for (int k = 0; k < g.K; ++k) {
            const int ki = k / g.kw, kj = k - ki * g.kw;
            const A o_h = ld<T>(offn, (size_t)(grp * 2 * g.K + 2 * k) * g.P + pp);
            const A o_w = ld<T>(offn, (size_t)(grp * 2 * g.K + 2 * k + 1) * g.P + pp);
            const A m = mskn ? ld<T>(mskn, (size_t)(grp * g.K + k) * g.P + pp) : (A)1;
            const GTap<A> t = gtap<A>(plane, (A)(ho * g.sh - g.ph + ki * g.dh) + o_h, (A)(wo * g.sw - g.pw + kj * g.dw) + o_w, g);
            const A hh = (A)1 - t.lh, hw = (A)1 - t.lw;
            const A col = t.inside ? (hh * hw * t.v1 + hh * t.lw * t.v2 + t.lh * hw * t.v3 + t.lh * t.lw * t.v4) * m : (A)0;
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                const int oo = co0 + o;                         // uniform: the weight is a scalar load
                if (oo < g.Co && oo / g.cout_g == wgrp) acc[o] += ld<T>(w, ((size_t)oo * g.cin_g + cl) * g.K + k) * col;
            }
        }
    }
    if (!valid) return;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        const int oo = co0 + o;
        if (oo < g.Co) {
            const size_t idx = ((size_t)n * g.Co + oo) * g.P + p;
            A r = (A)alpha * (acc[o] + (bias ? ld<T>(bias, oo) : (A)0));
            if (beta != 0.f) r += (A)beta * ld<T>(out, idx);
            st<T, A>(out, idx, r);
        }
    }
}

template <typename A>
__device__ __forceinline__ A wsum(A v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid (deformable groups, N): a workgroup owns every pixel of its (image, deformable group) - grad_x is complete in the
// LDS plane, no cross-workgroup accumulation.  gw / gb: fp32 (fp64 for f64) accumulators `gwa` / `gba` (zeroed by the
// host entry point, added to grad_weight / grad_bias afterwards).  LDS: plane + grad plane (A) + 4 reduction slots.
template <typename T>
__global__ __launch_bounds__(THREADS) void mdcn_generic_bwd_kernel(const T* __restrict__ x, const T* __restrict__ off,
                                                                    const T* __restrict__ msk, const T* __restrict__ w,
                                                                    const T* __restrict__ gout, T* __restrict__ gx,
                                                                    T* __restrict__ goff, T* __restrict__ gmsk,
                                                                    typename Acc<T>::type* __restrict__ gwa,
                                                                    typename Acc<T>::type* __restrict__ gba, GGeom g) {
    typedef typename Acc<T>::type A;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    A* plane = reinterpret_cast<A*>(smem_raw);
    A* gplane = plane + g.plane;
    A* red = gplane + g.plane;                               // [4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.x, n = blockIdx.y;
    const T* xn = x + (size_t)n * g.C * g.H * g.W;
    const T* offg = off + ((size_t)n * g.dg + grp) * 2 * g.K * g.P;
    const T* mskg = msk ? msk + ((size_t)n * g.dg + grp) * g.K * g.P : nullptr;
    T* goffg = goff + ((size_t)n * g.dg + grp) * 2 * g.K * g.P;
    T* gmskg = gmsk ? gmsk + ((size_t)n * g.dg + grp) * g.K * g.P : nullptr;
    const T* gon = gout + (size_t)n * g.Co * g.P;
    for (int i = tid; i < g.plane; i += THREADS) plane[i] = (A)0;

    auto block_add = [&](A v, A* dst) {                       // dst += sum over the workgroup of v
        v = wsum<A>(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) atomicAdd(dst, (red[0] + red[1]) + (red[2] + red[3]));
    };

    for (int cl = 0; cl < g.cpg_dg; ++cl) {
        const int c = grp * g.cpg_dg + cl;
        const int wgrp = c / g.cin_g, ci = c - wgrp * g.cin_g;
        __syncthreads();
        for (int i = tid; i < g.plane; i += THREADS) gplane[i] = (A)0;
        gstage<T, A>(plane, xn + (size_t)c * g.H * g.W, g, tid);
        __syncthreads();
        for (int k = 0; k < g.K; ++k) {
            const int ki = k / g.kw, kj = k - ki * g.kw;
            // output channels of this input channel's conv group, 16 at a time; the first chunk also produces the
            // gradients w.r.t. input / offset / mask, which need (W^T grad_out) over the WHOLE group
            for (int oc = 0; oc < g.cout_g; oc += 16) {
                A gwacc[16];
#pragma unroll
                for (int o = 0; o < 16; ++o) gwacc[o] = (A)0;
                for (int p = tid; p < g.P; p += THREADS) {
                    const int ho = p / g.Wo, wo = p - ho * g.Wo;
                    const A o_h = ld<T>(offg, (size_t)(2 * k) * g.P + p), o_w = ld<T>(offg, (size_t)(2 * k + 1) * g.P + p);
                    const A m = mskg ? ld<T>(mskg, (size_t)k * g.P + p) : (A)1;
                    const GTap<A> t = gtap<A>(plane, (A)(ho * g.sh - g.ph + ki * g.dh) + o_h,
                                              (A)(wo * g.sw - g.pw + kj * g.dw) + o_w, g);
                    const A in = t.inside ? (A)1 : (A)0;
                    const A hh = (A)1 - t.lh, hw = (A)1 - t.lw;
                    const A bil = hh * hw * t.v1 + hh * t.lw * t.v2 + t.lh * hw * t.v3 + t.lh * t.lw * t.v4;
                    if (oc == 0) {
                        A gcol = (A)0;                          // (W^T grad_out)[c, k, p]   (cpp:602-605)
                        for (int o = 0; o < g.cout_g; ++o) {
                            const int oo = wgrp * g.cout_g + o;
                            gcol += ld<T>(w, ((size_t)oo * g.cin_g + ci) * g.K + k) * ld<T>(gon, (size_t)oo * g.P + p);
                        }
                        A gm = in * gcol * bil;
                        const A gc_m = in * gcol * m;
                        A d_h = gc_m * (hw * (t.v3 - t.v1) + t.lw * (t.v4 - t.v2));
                        A d_w = gc_m * (hh * (t.v2 - t.v1) + t.lh * (t.v4 - t.v3));
                        if (cl > 0) {                           // several channels share one offset group: accumulate
                            if (gmskg) gm += ld<T>(gmskg, (size_t)k * g.P + p);
                            d_h += ld<T>(goffg, (size_t)(2 * k) * g.P + p);
                            d_w += ld<T>(goffg, (size_t)(2 * k + 1) * g.P + p);
                        }
                        if (gmskg) st<T, A>(gmskg, (size_t)k * g.P + p, gm);
                        st<T, A>(goffg, (size_t)(2 * k) * g.P + p, d_h);
                        st<T, A>(goffg, (size_t)(2 * k + 1) * g.P + p, d_w);
                        atomicAdd(&gplane[t.addr], gc_m * hh * hw);
                        atomicAdd(&gplane[t.addr + 1], gc_m * hh * t.lw);
                        atomicAdd(&gplane[t.addr + g.LW], gc_m * t.lh * hw);
                        atomicAdd(&gplane[t.addr + g.LW + 1], gc_m * t.lh * t.lw);
                    }
                    const A col = in * bil * m;
#pragma unroll
                    for (int o = 0; o < 16; ++o)
                        if (oc + o < g.cout_g) gwacc[o] += ld<T>(gon, (size_t)(wgrp * g.cout_g + oc + o) * g.P + p) * col;
                }
#pragma unroll
                for (int o = 0; o < 16; ++o)
                    if (oc + o < g.cout_g)                      // uniform
                        block_add(gwacc[o], &gwa[((size_t)(wgrp * g.cout_g + oc + o) * g.cin_g + ci) * g.K + k]);
            }
        }
        __syncthreads();
        T* gxc = gx + ((size_t)n * g.C + c) * g.H * g.W;
        for (int i = tid; i < g.H * g.W; i += THREADS) {
            const int y = i / g.W, xx = i - y * g.W;
            st<T, A>(gxc, i, gplane[(y + 1) * g.LW + PADL + xx]);
        }
    }
    if (gba != nullptr && grp == 0) {                         // grad_bias[o] += sum_p grad_out[n, o, p]
        for (int o = 0; o < g.Co; ++o) {
            A s = (A)0;
            for (int p = tid; p < g.P; p += THREADS) s += ld<T>(gon, (size_t)o * g.P + p);
            block_add(s, &gba[o]);
        }
    }
}

template <typename T>
__global__ void add_into_kernel(T* __restrict__ dst, const typename Acc<T>::type* __restrict__ src, size_t n) {
    typedef typename Acc<T>::type A;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) st<T, A>(dst, i, ld<T>(dst, i) + src[i]);
}

bool make_ggeom(GGeom& g, int N, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                int groups, int dg) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 ||
        dh <= 0 || dw <= 0 || groups <= 0 || dg <= 0)
        return false;
    if (C % groups || Co % groups || C % dg) return false;
    g.N = N, g.C = C, g.H = H, g.W = W, g.Co = Co, g.K = kh * kw, g.kh = kh, g.kw = kw;
    g.sh = sh, g.sw = sw, g.ph = ph, g.pw = pw, g.dh = dh, g.dw = dw;
    g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    if (g.Ho <= 0 || g.Wo <= 0) return false;
    g.P = g.Ho * g.Wo;
    g.dg = dg, g.cpg_dg = C / dg, g.cin_g = C / groups, g.cout_g = Co / groups;
    g.LW = (W + PADL + 1 + 3) & ~3;
    g.plane = (H + 2) * g.LW;
    return true;
}

template <typename T>
int fwd_t(const void* x, const void* off, const void* msk, const void* w, const void* bias, void* out, const GGeom& g,
          float alpha, float beta, hipStream_t st) {
    typedef typename Acc<T>::type A;
    const size_t lds = (size_t)g.plane * sizeof(A);
    if (lds > OTP_LDS_LIMIT) return OTP_ERR_UNSUPPORTED;
    auto kern = mdcn_generic_fwd_kernel<T>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    kern<<<dim3(otp_ceil_div(g.P, THREADS), g.N, otp_ceil_div(g.Co, 16)), THREADS, lds, st>>>(
        static_cast<const T*>(x), static_cast<const T*>(off), static_cast<const T*>(msk), static_cast<const T*>(w),
        static_cast<const T*>(bias), static_cast<T*>(out), g, alpha, beta);
    return otp_launch_status();
}

template <typename T>
int bwd_t(const void* x, const void* off, const void* msk, const void* w, const void* gout, void* gx, void* goff, void* gmsk,
          void* gw, void* gb, void* ws, const GGeom& g, hipStream_t st) {
    typedef typename Acc<T>::type A;
    const size_t lds = ((size_t)2 * g.plane + 4) * sizeof(A);
    if (lds > OTP_LDS_LIMIT) return OTP_ERR_UNSUPPORTED;
    const size_t nw = (size_t)g.Co * g.cin_g * g.K, nb = gb ? (size_t)g.Co : 0;
    A* gwa = static_cast<A*>(ws);
    A* gba = gb ? gwa + nw : nullptr;
    if (hipMemsetAsync(ws, 0, (nw + nb) * sizeof(A), st) != hipSuccess) return OTP_ERR_LAUNCH;
    auto kern = mdcn_generic_bwd_kernel<T>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    kern<<<dim3(g.dg, g.N), THREADS, lds, st>>>(static_cast<const T*>(x), static_cast<const T*>(off), static_cast<const T*>(msk),
                                                static_cast<const T*>(w), static_cast<const T*>(gout), static_cast<T*>(gx),
                                                static_cast<T*>(goff), static_cast<T*>(gmsk), gwa, gba, g);
    if (otp_launch_status() != OTP_OK) return OTP_ERR_LAUNCH;
    add_into_kernel<T><<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(static_cast<T*>(gw), gwa, nw);
    if (gb) add_into_kernel<T><<<(unsigned)((nb + 255) / 256), 256, 0, st>>>(static_cast<T*>(gb), gba, nb);
    return otp_launch_status();
}

}  // namespace

// entry points used by mdcn.hip's dispatch (not part of the C ABI themselves)
int otp_mdcn_generic_forward(const void* x, const void* off, const void* msk, const void* w, const void* bias, void* out, int N,
                             int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                             int groups, int dg, float alpha, float beta, int dtype, hipStream_t st) {
    GGeom g;
    if (!make_ggeom(g, N, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg)) return OTP_ERR_BAD_ARG;
    switch (dtype) {
        case OTP_DTYPE_F32: return fwd_t<float>(x, off, msk, w, bias, out, g, alpha, beta, st);
        case OTP_DTYPE_F16: return fwd_t<__half>(x, off, msk, w, bias, out, g, alpha, beta, st);
        case OTP_DTYPE_BF16: return fwd_t<__hip_bfloat16>(x, off, msk, w, bias, out, g, alpha, beta, st);
        case OTP_DTYPE_F64: return fwd_t<double>(x, off, msk, w, bias, out, g, alpha, beta, st);
    }
    return OTP_ERR_UNSUPPORTED;
}

size_t otp_mdcn_generic_backward_workspace(int C, int Co, int kh, int kw, int groups) {
    if (groups <= 0 || C <= 0 || Co <= 0) return 0;
    return ((size_t)Co * (C / groups) * kh * kw + Co) * sizeof(double);
}

int otp_mdcn_generic_backward(const void* x, const void* off, const void* msk, const void* w, const void* gout, void* gx,
                              void* goff, void* gmsk, void* gw, void* gb, void* ws, size_t ws_bytes, int N, int C, int H, int W,
                              int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg,
                              int dtype, hipStream_t st) {
    GGeom g;
    if (!make_ggeom(g, N, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg)) return OTP_ERR_BAD_ARG;
    if (!ws || ws_bytes < otp_mdcn_generic_backward_workspace(C, Co, kh, kw, groups)) return OTP_ERR_WORKSPACE;
    switch (dtype) {
        case OTP_DTYPE_F32: return bwd_t<float>(x, off, msk, w, gout, gx, goff, gmsk, gw, gb, ws, g, st);
        case OTP_DTYPE_F16: return bwd_t<__half>(x, off, msk, w, gout, gx, goff, gmsk, gw, gb, ws, g, st);
        case OTP_DTYPE_BF16: return bwd_t<__hip_bfloat16>(x, off, msk, w, gout, gx, goff, gmsk, gw, gb, ws, g, st);
        case OTP_DTYPE_F64: return bwd_t<double>(x, off, msk, w, gout, gx, goff, gmsk, gw, gb, ws, g, st);
    }
    return OTP_ERR_UNSUPPORTED;
}
