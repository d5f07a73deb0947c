// Modulated deformable convolution for gfx950 - forward and backward, fused (no im2col buffer).
//
// Replaces modulated_deform_conv_cuda_forward / _backward of the reference
// (thirdparty/deform_conv/src/deform_conv_cuda.cpp:474-549, 551-664 and the kernels at
// src/deform_conv_cuda_kernel.cu:403-432, 434-503, 506-571, 574-631, 634-705).  Not a translation:
//   * forward: one launch for the whole batch.  A workgroup owns a contiguous range of output pixels
//     of one image and walks the input planes; each plane (H x W floats, 27.6 KB at 96x72) is staged
//     into LDS once with a one-pixel ZERO BORDER, so the data-dependent bilinear gather hits LDS and
//     the per-corner bounds tests of the reference become plain reads of the border.  The 18 offset +
//     9 mask streams of a plane (93 % of the op's bytes) are read exactly once, coalesced, straight
//     into registers; the (Cout x C*K) contraction and the bias are accumulated in registers, so the
//     reference's `columns` round trip (C*K*P floats per image), its per-image GEMM launches, its
//     output.zero_() and its separate bias add disappear.  alpha/beta fold the weighted sum over
//     dilations of model/OTPose.py:387-392 into the store.
//   * backward: a workgroup owns (image, deformable group, pixel chunk).  grad_x is accumulated in an
//     LDS copy of the plane as 64-bit fixed point (ds_add_u64: order-independent, so the backward is
//     bit-reproducible - the reference's float atomics are not); grad_offset / grad_mask are written once,
//     coalesced; grad_weight / grad_bias partial sums live in registers, are reduced with wavefront
//     shuffles and leave as per-workgroup partials that a second launch adds in a fixed order.
//
// HBM roofline: algorithmic bytes per (image, call) = (C + 2*dg*K + dg*K + Cout) * H*W * 4
// (13,630,464 B at C=Cout=dg=17, K=9, 96x72).
#include "common.h"

namespace {

constexpr int PADL = 4;      // LDS column PADL-1 holds x = -1 (zero); data starts 16-byte aligned at PADL
// 512 threads: 14 tiles x 16 images = 224 workgroups of 8 waves - one per CU, two waves per SIMD - at cfg2; a workgroup stages
// every input plane whole, so twice the pixels per workgroup halve the staging per pixel (64.3 -> 58.3 us against 256 threads,
// 432 workgroups, 1.7 per CU)
#ifndef OTP_DCN_THREADS
#define OTP_DCN_THREADS 512
#endif
constexpr int FWD_THREADS = OTP_DCN_THREADS;

struct Geom {
    int N, C, H, W, Co, K, kh, kw, stride, pad, dil, Ho, Wo, P;
    int dg, cpg_dg, cin_g, cout_g;
    int LW, plane;           // LDS row pitch (floats) and plane size (H+2)*LW
};

struct Tap {                 // one bilinear sample of the staged plane
    float v1, v2, v3, v4;    // corner values (zero border supplies out-of-image corners)
    float lh, lw;            // fractional parts
    int addr;                // LDS index of the (h_low, w_low) corner
    bool inside;             // h in (-1,H) and w in (-1,W)  (kernel.cu:556)
};

__device__ __forceinline__ Tap fetch_tap(const float* __restrict__ plane, float h_im, float w_im,
                                         const Geom& g) {
    Tap t;
    t.inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)g.H) && (w_im < (float)g.W);
    // Inside samples are untouched by the clamps below (floor is already in [-1, H-1]); outside / NaN
    // samples only need a finite fraction and an address inside the padded plane - their value is dropped.
    float hc = fminf(fmaxf(h_im, -2.f), (float)(g.H + 1));
    float wc = fminf(fmaxf(w_im, -2.f), (float)(g.W + 1));
    float hf = floorf(hc), wf = floorf(wc);
    t.lh = hc - hf;
    t.lw = wc - wf;
    int hl = min(max((int)hf, -1), g.H - 1);
    int wl = min(max((int)wf, -1), g.W - 1);
    t.addr = (hl + 1) * g.LW + wl + PADL;
    t.v1 = plane[t.addr];
    t.v2 = plane[t.addr + 1];
    t.v3 = plane[t.addr + g.LW];
    t.v4 = plane[t.addr + g.LW + 1];
    return t;
}

// Forward-only sample: the coordinate is clamped into the zero border ([-1, H] x [-1, W]), where the bilinear blend is
// exactly zero, so no inside test and no select are needed; inside the image the arithmetic is the reference's
// (hh*hw*v1 + hh*lw*v2 + lh*hw*v3 + lh*lw*v4, kernel.cu:403-432).  NaN offsets clamp like -inf (value 0).
__device__ __forceinline__ float sample_zero_border(const float* __restrict__ plane, float h_im, float w_im,
                                                    const Geom& g) {
    const float hc = fminf(fmaxf(h_im, -1.f), (float)g.H);
    const float wc = fminf(fmaxf(w_im, -1.f), (float)g.W);
    const float hf = fminf(floorf(hc), (float)(g.H - 1)), wf = fminf(floorf(wc), (float)(g.W - 1));
    const float lh = hc - hf, lw = wc - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int addr = ((int)hf + 1) * g.LW + (int)wf + PADL;
    const float v1 = plane[addr], v2 = plane[addr + 1], v3 = plane[addr + g.LW], v4 = plane[addr + g.LW + 1];
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

__device__ __forceinline__ float bilinear(const Tap& t) {
    float hh = 1.f - t.lh, hw = 1.f - t.lw;
    return hh * hw * t.v1 + hh * t.lw * t.v2 + t.lh * hw * t.v3 + t.lh * t.lw * t.v4;
}

// zero the padded plane once; the interior is overwritten per input plane, the border stays zero
__device__ __forceinline__ void zero_plane(float* plane, int n, int tid, int nthreads) {
    for (int i = tid; i < n; i += nthreads) plane[i] = 0.f;
}

__device__ __forceinline__ void stage_plane(float* __restrict__ plane, const float* __restrict__ src,
                                            const Geom& g, int tid, int nthreads) {
    if ((g.W & 3) == 0) {
        const int wq = g.W >> 2, nq = g.H * wq;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        for (int i = tid; i < nq; i += nthreads) {
            int y = i / wq, xq = i - y * wq;
            float4 v = s4[i];
            *reinterpret_cast<float4*>(&plane[(y + 1) * g.LW + PADL + 4 * xq]) = v;
        }
    } else {
        const int n = g.H * g.W;
        for (int i = tid; i < n; i += nthreads) {
            int y = i / g.W, xx = i - y * g.W;
            plane[(y + 1) * g.LW + PADL + xx] = src[i];
        }
    }
}

// nearest 64-bit integer of a fixed-point contribution; out-of-range / NaN contributions add nothing HERE - a workgroup that met a
// non-finite grad_out / mask / weight writes NaN planes instead (mdcn_bwd_kernel), as the reference's float atomicAdd would have
// carried the NaN / Inf into grad_x (deform_conv_cuda_kernel.cu:612-629)
__device__ __forceinline__ long long fx_round(double v) {
    return fabs(v) < 9.0e18 ? __double2ll_rn(v) : 0ll;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// CO_T : output channels accumulated per workgroup pass (blockIdx.z walks Cout in chunks of CO_T)
// VEC  : output pixels per thread (pixel p = (tile*VEC + v)*FWD_THREADS + tid: coalesced dword streams,
//        conflict-free LDS gathers for smooth offset fields)
// K9   : kernel is 3x3 (taps unrolled, offset/mask streams of a plane prefetched into registers)
template <int CO_T, int VEC, bool K9>
__global__ __launch_bounds__(FWD_THREADS, 2) void mdcn_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ off, const float* __restrict__ msk,
    const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out, Geom g,
    float alpha, float beta) {
    constexpr int COP = (CO_T + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* plane = smem;                    // g.plane floats (K9: two buffers, the next plane is written while this one is read)
    float* wl = smem + (K9 ? 2 : 1) * g.plane;   // [C*K][COP] transposed weights, zero where groups differ
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int co0 = blockIdx.z * CO_T;
    const int nco = min(CO_T, g.Co - co0);
    const int K = K9 ? 9 : g.K;

    zero_plane(plane, (K9 ? 2 : 1) * g.plane, tid, FWD_THREADS);
    for (int i = tid; i < g.C * K * COP; i += FWD_THREADS) {
        int ck = i / COP, o = i - ck * COP;
        int c = ck / K, k = ck - c * K;
        float v = 0.f;
        if (o < nco) {
            int oo = co0 + o;
            int grp = oo / g.cout_g;
            if (c / g.cin_g == grp) v = w[((size_t)oo * g.cin_g + (c - grp * g.cin_g)) * K + k];
        }
        wl[i] = v;
    }

    int p[VEC], hin[VEC], win[VEC];
    bool valid[VEC];
    float acc[VEC][CO_T];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        p[v] = (blockIdx.x * VEC + v) * FWD_THREADS + tid;
        valid[v] = p[v] < g.P;
        int pp = valid[v] ? p[v] : 0;
        int ho = pp / g.Wo, wo = pp - ho * g.Wo;
        hin[v] = ho * g.stride - g.pad;
        win[v] = wo * g.stride - g.pad;
        p[v] = pp;
#pragma unroll
        for (int o = 0; o < CO_T; ++o) acc[v][o] = 0.f;
    }

    const float* xn = x + (size_t)n * g.C * g.H * g.W;
    // per-image buffer resources: stream address = SGPR plane offset + one VGPR pixel offset
    const otp_rsrc roff = make_rsrc(off + (size_t)n * g.dg * 2 * K * g.P, (size_t)g.dg * 2 * K * g.P * 4);
    const otp_rsrc rmsk = make_rsrc(msk + (size_t)n * g.dg * K * g.P, (size_t)g.dg * K * g.P * 4);
    int pb[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) pb[v] = p[v] * 4;
    const int P4 = g.P * 4;

    if (K9) {
        // One "step" = one kernel row (3 taps) of one input plane.  The 9*VEC stream loads of steps s+1 and s+2 are in
        // flight while step s is computed: ~27 dword streams per thread keep enough bytes in the air to cover the
        // HBM round trip at 2-3 waves per SIMD.
        float r0[9][VEC], r1[9][VEC], r2[9][VEC];
        auto load_step = [&](float (&dst)[9][VEC], int step) __attribute__((always_inline)) {
            const int c = step / 3, i = step - c * 3;
            const int grp = c / g.cpg_dg;
            const int so = (grp * 18 + 6 * i) * P4, sm = (grp * 9 + 3 * i) * P4;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    dst[3 * j + 0][v] = bload(roff, pb[v], so + (2 * j) * P4);
                    dst[3 * j + 1][v] = bload(roff, pb[v], so + (2 * j + 1) * P4);
                    dst[3 * j + 2][v] = bload(rmsk, pb[v], sm + j * P4);
                }
        };
        // Plane pipeline: while plane c is gathered out of one LDS buffer, plane c+1 travels global -> registers ->
        // the other buffer, so a plane change costs one barrier and no exposed memory round trip.  (W % 4 == 0 and
        // planes of at most 2048 float4; otherwise the plane is staged in place between two barriers.)
        constexpr int MAXQ = 8 * 256 / FWD_THREADS;
        const int wq = g.W >> 2, nq = g.H * wq;
        const bool piped = (g.W & 3) == 0 && nq <= MAXQ * FWD_THREADS;
        otp_f32x4 pq[MAXQ];                                    // (ext-vector type: stays in registers)
#define OTP_LOAD_PLANE(c_)                                                                                   \
    {                                                                                                        \
        const otp_f32x4* s4_ = reinterpret_cast<const otp_f32x4*>(xn + (size_t)(c_) * g.H * g.W);            \
        _Pragma("unroll") for (int k_ = 0; k_ < MAXQ; ++k_) {                                                \
            const int i_ = tid + k_ * FWD_THREADS;                                                           \
            pq[k_] = s4_[i_ < nq ? i_ : 0];              /* unconditional: keeps pq in registers */          \
        }                                                                                                    \
    }
#define OTP_STORE_PLANE(dst_)                                                                                \
    {                                                                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < MAXQ; ++k_) {                                                \
            const int i_ = tid + k_ * FWD_THREADS;                                                           \
            if (i_ < nq) {                                                                                   \
                const int y_ = i_ / wq, xq_ = i_ - y_ * wq;                                                  \
                *reinterpret_cast<otp_f32x4*>(&(dst_)[(y_ + 1) * g.LW + PADL + 4 * xq_]) = pq[k_];           \
            }                                                                                                \
        }                                                                                                    \
    }
        auto compute_step = [&](const float (&cur)[9][VEC], int step) __attribute__((always_inline)) {
            const int c = step / 3, i = step - c * 3;
            const float* pl = plane + (piped ? (c & 1) * g.plane : 0);
            if (i == 0 && !piped) {
                __syncthreads();                               // every wave finished gathering plane c-1
                stage_plane(plane, xn + (size_t)c * g.H * g.W, g, tid, FWD_THREADS);
                __syncthreads();
            }
            const int di = i * g.dil;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float* wrow = wl + (c * 9 + i * 3 + j) * COP;
                float wr[COP];
#pragma unroll
                for (int q = 0; q < COP / 4; ++q)
                    *reinterpret_cast<float4*>(&wr[4 * q]) = *reinterpret_cast<const float4*>(&wrow[4 * q]);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float col = sample_zero_border(pl, (float)(hin[v] + di) + cur[3 * j][v],
                                                         (float)(win[v] + j * g.dil) + cur[3 * j + 1][v], g) *
                                      cur[3 * j + 2][v];
                    // packed f32 FMAs (v_pk_fma_f32): two output channels per instruction
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 col2 = {col, col};
#pragma unroll
                    for (int o = 0; o + 1 < CO_T; o += 2) {
                        f32x2 a2 = {acc[v][o], acc[v][o + 1]};
                        const f32x2 w2 = {wr[o], wr[o + 1]};
                        a2 = __builtin_elementwise_fma(w2, col2, a2);
                        acc[v][o] = a2.x;
                        acc[v][o + 1] = a2.y;
                    }
                    if (CO_T & 1) acc[v][CO_T - 1] = fmaf(wr[CO_T - 1], col, acc[v][CO_T - 1]);
                }
            }
        };
        const int nsteps = g.C * 3;                            // a multiple of 3: the rotation below is static
        load_step(r0, 0);
        if (nsteps > 1) load_step(r1, 1);
        if (piped) {
            OTP_LOAD_PLANE(0)
            __syncthreads();                                   // borders zeroed, weights staged
            OTP_STORE_PLANE(plane)
            __syncthreads();
        }
#pragma unroll 1
        for (int step = 0; step < nsteps; step += 3) {         // one trip = one input plane
            const int c = step / 3;
            const bool next = piped && c + 1 < g.C;
            if (next) OTP_LOAD_PLANE(c + 1)                    // in flight during the three steps of plane c
            if (step + 2 < nsteps) load_step(r2, step + 2);
            compute_step(r0, step);
            if (step + 3 < nsteps) load_step(r0, step + 3);
            if (step + 1 < nsteps) compute_step(r1, step + 1);
            if (step + 4 < nsteps) load_step(r1, step + 4);
            if (step + 2 < nsteps) compute_step(r2, step + 2);
            if (next) {
                float* other = plane + ((c + 1) & 1) * g.plane;   // last read before the previous barrier
                OTP_STORE_PLANE(other)
                __syncthreads();
            }
        }
#undef OTP_LOAD_PLANE
#undef OTP_STORE_PLANE
    } else {
        for (int c = 0; c < g.C; ++c) {
            const int grp = c / g.cpg_dg;
            const int so = grp * 2 * K * P4, sm = grp * K * P4;
            __syncthreads();
            stage_plane(plane, xn + (size_t)c * g.H * g.W, g, tid, FWD_THREADS);
            __syncthreads();
            for (int k = 0; k < K; ++k) {
                const float* wrow = wl + (c * K + k) * COP;
                const int di = (k / g.kw) * g.dil, dj = (k % g.kw) * g.dil;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float o_h = bload(roff, pb[v], so + (2 * k) * P4);
                    float o_w = bload(roff, pb[v], so + (2 * k + 1) * P4);
                    float m = bload(rmsk, pb[v], sm + k * P4);
                    Tap t = fetch_tap(plane, (float)(hin[v] + di) + o_h, (float)(win[v] + dj) + o_w, g);
                    float col = t.inside ? bilinear(t) * m : 0.f;
#pragma unroll
                    for (int o = 0; o < CO_T; ++o) acc[v][o] = fmaf(wrow[o], col, acc[v][o]);
                }
            }
        }
    }

#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        if (!valid[v]) continue;
#pragma unroll
        for (int o = 0; o < CO_T; ++o) {
            if (o < nco) {
                size_t idx = ((size_t)n * g.Co + co0 + o) * g.P + p[v];
                float r = alpha * (acc[v][o] + (bias ? bias[co0 + o] : 0.f));
                if (beta != 0.f) r = fmaf(beta, out[idx], r);
                out[idx] = r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// grid = (pixel chunks S, deformable groups, N), 512 threads, one workgroup per CU.  LDS: grad_x accumulation plane (64-bit
// integers) + x plane (both with the zero border) + [K][COP] weights of the current channel + reduction scratch.
// CO_T must cover Cout (host restricts the fast path to Cout <= CO_T).
//
// DETERMINISM (round 4).  The reference scatters grad_x with float atomicAdd (kernel.cu:612-629), whose result depends on the
// order the adds arrive in; round 3 traced the run-to-run spread of the bf16 training backward (one step in five off by
// 1e-3 .. 6e-3) to exactly such order noise (3e-8 here) being amplified by the 60 bf16-rounded layers behind it.  This kernel
// has NO order-dependent arithmetic:
//   * grad_x: every contribution v is rounded ONCE to a multiple of q = 2^(e-40), where 2^e > B >= |v| and
//     B = max_k sum_o |W[o,c,k]| * max|grad_out| * max|mask| over the workgroup's pixels (maxima are order-independent), and
//     added to the LDS plane as a 64-bit integer (ds_add_u64): integer addition is associative, so the plane is bit-identical
//     whatever order the waves arrive in.  |v|/q < 2^40 and a cell receives < 2^20 contributions, so nothing overflows;
//     the rounding error per contribution is <= 2^-41 B (a float atomic rounds the running sum to 2^-24 of itself per add).
//   * a plane shared by S > 1 pixel chunks leaves as S partial planes (workspace) that a second launch adds in chunk order;
//   * grad_weight / grad_bias: per-workgroup partial sums (fixed shuffle tree) into the workspace, added in workgroup order by
//     the same second launch.
constexpr int BWD_THREADS = 512;
constexpr int BWD_WAVES = BWD_THREADS / 64;
constexpr int BWD_SMAX = 8;                    // most pixel chunks per (image, deformable group)

template <int CO_T>
__global__ __launch_bounds__(BWD_THREADS, 1) void mdcn_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ off, const float* __restrict__ msk,
    const float* __restrict__ w, const float* __restrict__ gout, float* __restrict__ gxdst,
    float* __restrict__ goff, float* __restrict__ gmsk, float* __restrict__ gwpart, float* __restrict__ gbpart,
    Geom g, int chunk_px) {
    constexpr int COP = (CO_T + 3) & ~3;
    constexpr int K = 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned long long* gpl = reinterpret_cast<unsigned long long*>(smem);   // grad_x accumulation plane, fixed point
    float* plane = smem + 2 * g.plane;         // x plane
    float* wl = plane + g.plane;               // [K][COP] weights W[o][c][k] of channel c
    float* red = wl + K * COP;                 // [BWD_WAVES][CO_T] reduction scratch (also the maxima of the bound)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.y, n = blockIdx.z;
    const int p_begin = blockIdx.x * chunk_px;
    const int p_end = min(g.P, p_begin + chunk_px);
    const int wslot = n * gridDim.x + blockIdx.x;          // this workgroup's row of the grad_weight / grad_bias partials

    zero_plane(plane, g.plane, tid, BWD_THREADS);
    const float* xn = x + (size_t)n * g.C * g.H * g.W;
    const int P4 = g.P * 4;
    const otp_rsrc rog = make_rsrc(off + ((size_t)n * g.dg + grp) * 2 * K * g.P, (size_t)2 * K * P4);
    const otp_rsrc rmg = make_rsrc(msk + ((size_t)n * g.dg + grp) * K * g.P, (size_t)K * P4);
    const otp_rsrc rgog = make_rsrc(goff + ((size_t)n * g.dg + grp) * 2 * K * g.P, (size_t)2 * K * P4);
    const otp_rsrc rgmg = make_rsrc(gmsk + ((size_t)n * g.dg + grp) * K * g.P, (size_t)K * P4);
    const otp_rsrc rgon = make_rsrc(gout + (size_t)n * g.Co * g.P, (size_t)g.Co * P4);

    // ---- the two data maxima of the contribution bound, over this workgroup's pixels ------------------------------
    float gomax = 0.f, mmax = 0.f;
    int nfl = 0;                               // a non-finite grad_out / mask value among this workgroup's pixels (fmaxf drops NaN)
    for (int p = p_begin + tid; p < p_end; p += BWD_THREADS) {
#pragma unroll
        for (int o = 0; o < CO_T; ++o) {
            const float v = fabsf(bload(rgon, p * 4, o * P4));                                     // rows >= Cout read 0
            nfl |= !(v < INFINITY);
            gomax = fmaxf(gomax, v);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float v = fabsf(bload(rmg, p * 4, k * P4));
            nfl |= !(v < INFINITY);
            mmax = fmaxf(mmax, v);
        }
    }
    gomax = wave_max(gomax);
    mmax = wave_max(mmax);
    nfl = __any(nfl);                                                          // (per wave)
    if (lane == 0) { red[wave] = gomax; red[BWD_WAVES + wave] = mmax; red[2 * BWD_WAVES + wave] = nfl ? 1.f : 0.f; }
    __syncthreads();
    bool nonfinite = false;                          // (uniform)
#pragma unroll
    for (int i = 0; i < BWD_WAVES; ++i) nonfinite |= red[2 * BWD_WAVES + i] != 0.f;
#pragma unroll
    for (int i = 0; i < BWD_WAVES; ++i) { gomax = fmaxf(gomax, red[i]); mmax = fmaxf(mmax, red[BWD_WAVES + i]); }

    for (int cl = 0; cl < g.cpg_dg; ++cl) {
        const int c = grp * g.cpg_dg + cl;
        const int wgrp = c / g.cin_g;          // conv group of this input channel
        __syncthreads();
        for (int i = tid; i < g.plane; i += BWD_THREADS) gpl[i] = 0ull;
        stage_plane(plane, xn + (size_t)c * g.H * g.W, g, tid, BWD_THREADS);
        for (int i = tid; i < K * COP; i += BWD_THREADS) {
            int k = i / COP, o = i - k * COP;
            float v = 0.f;
            if (o < g.Co && o / g.cout_g == wgrp) v = w[((size_t)o * g.cin_g + (c - wgrp * g.cin_g)) * K + k];
            wl[i] = v;
        }
        __syncthreads();
        if (tid < K) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < CO_T; ++o) s += fabsf(wl[tid * COP + o]);
            red[tid] = s;
        }
        __syncthreads();
        float wmax = 0.f;
        bool wnf = false;                      // a non-finite weight of this channel (its |w| sums are then Inf / NaN)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            wnf |= !(red[k] < INFINITY);
            wmax = fmaxf(wmax, red[k]);
        }
        // |gcol * mask * (bilinear weight <= 1)| <= bound (1.001: the fma chain of gcol rounds); 2^e > bound
        const float bound = wmax * gomax * mmax * 1.001f;
        double sc = 0.0, inv = 0.0;            // bound == 0: every contribution is 0; not finite: nothing sensible to add
        if (bound > 0.f && bound < INFINITY) {
            int e;
            (void)frexpf(bound, &e);
            sc = ldexp(1.0, 40 - e);
            inv = ldexp(1.0, e - 40);
        }
        __syncthreads();                       // red is reused below

        float gbacc[CO_T];                     // sum_p gout[o,p] (reduced only by the channel-0 workgroups)
#pragma unroll
        for (int o = 0; o < CO_T; ++o) gbacc[o] = 0.f;
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            float gwacc[CO_T];                 // sum_p gout[o,p] * col[c,k,p]
            float wk[COP];
#pragma unroll
            for (int o = 0; o < CO_T; ++o) gwacc[o] = 0.f;
#pragma unroll
            for (int q = 0; q < COP / 4; ++q)
                *reinterpret_cast<float4*>(&wk[4 * q]) = *reinterpret_cast<const float4*>(&wl[k * COP + 4 * q]);
            const int di = (k / 3) * g.dil, dj = (k % 3) * g.dil;
#pragma unroll 1
            for (int p = p_begin + tid; p < p_end; p += BWD_THREADS) {
                const float o_h = bload(rog, p * 4, (2 * k) * P4);
                const float o_w = bload(rog, p * 4, (2 * k + 1) * P4);
                const float m = bload(rmg, p * 4, k * P4);
                float go[CO_T];
#pragma unroll
                for (int o = 0; o < CO_T; ++o) go[o] = bload(rgon, p * 4, o * P4);   // rows >= Cout read 0
                float gcol = 0.f;              // (W^T gout)[c,k,p]   (cpp:602-605)
#pragma unroll
                for (int o = 0; o < CO_T; ++o) gcol = fmaf(wk[o], go[o], gcol);
                if (k == 0) {
#pragma unroll
                    for (int o = 0; o < CO_T; ++o) gbacc[o] += go[o];
                }
                const int ho = p / g.Wo, wo = p - ho * g.Wo;
                Tap t = fetch_tap(plane, (float)(ho * g.stride - g.pad + di) + o_h,
                                  (float)(wo * g.stride - g.pad + dj) + o_w, g);
                const float in = t.inside ? 1.f : 0.f;
                const float hh = 1.f - t.lh, hw = 1.f - t.lw;
                const float bil = bilinear(t);
                // grad_mask, grad_offset (kernel.cu:634-705; the zero border supplies the dropped corners)
                float gm = in * gcol * bil;
                const float gc_m = in * gcol * m;
                float d_h = gc_m * (hw * (t.v3 - t.v1) + t.lw * (t.v4 - t.v2));
                float d_w = gc_m * (hh * (t.v2 - t.v1) + t.lh * (t.v4 - t.v3));
                if (cl > 0) {                  // several channels share one offset group: accumulate
                    gm += bload(rgmg, p * 4, k * P4);
                    d_h += bload(rgog, p * 4, (2 * k) * P4);
                    d_w += bload(rgog, p * 4, (2 * k + 1) * P4);
                }
                bstore(gm, rgmg, p * 4, k * P4);
                bstore(d_h, rgog, p * 4, (2 * k) * P4);
                bstore(d_w, rgog, p * 4, (2 * k + 1) * P4);
                // grad_x: scatter to the four corners of the LDS plane (border cells are dropped later), fixed point
                if (gc_m != 0.f) {
                    const double gs = (double)gc_m * sc;
                    const double a1 = gs * (double)(hh * hw), a2 = gs * (double)(hh * t.lw);
                    const double a3 = gs * (double)(t.lh * hw), a4 = gs * (double)(t.lh * t.lw);
                    atomicAdd(&gpl[t.addr], (unsigned long long)fx_round(a1));
                    atomicAdd(&gpl[t.addr + 1], (unsigned long long)fx_round(a2));
                    atomicAdd(&gpl[t.addr + g.LW], (unsigned long long)fx_round(a3));
                    atomicAdd(&gpl[t.addr + g.LW + 1], (unsigned long long)fx_round(a4));
                }
                const float col = in * bil * m;
#pragma unroll
                for (int o = 0; o < CO_T; ++o) gwacc[o] = fmaf(go[o], col, gwacc[o]);
            }
            // grad_weight[:, c, k] partial: wave shuffle reduce, cross-wave through LDS in wave order
#pragma unroll
            for (int o = 0; o < CO_T; ++o) {
                float sred = wave_sum(gwacc[o]);
                if (lane == 0) red[wave * CO_T + o] = sred;
            }
            __syncthreads();
            if (tid < CO_T) {
                float sred = 0.f;
#pragma unroll
                for (int i = 0; i < BWD_WAVES; ++i) sred += red[i * CO_T + tid];
                gwpart[(((size_t)wslot * g.C + c) * K + k) * CO_T + tid] = sred;
            }
            __syncthreads();
        }
        // ---- grad_x plane -> global (the tensor itself, or this chunk's partial plane) -----------------------
        float* gxc = gxdst + (((size_t)blockIdx.x * g.N + n) * g.C + c) * g.H * g.W;
        const bool poison = nonfinite || wnf;  // (uniform) non-finite operands: the fixed-point plane cannot carry them - NaN, loudly
        for (int i = tid; i < g.H * g.W; i += BWD_THREADS) {
            int y = i / g.W, xx = i - y * g.W;
            gxc[i] = poison ? __builtin_nanf("") : (float)((double)(long long)gpl[(y + 1) * g.LW + PADL + xx] * inv);
        }
        // ---- grad_bias partial (channel-0 workgroups only) ---------------------------------------------------
        if (gbpart != nullptr && c == 0) {
#pragma unroll
            for (int o = 0; o < CO_T; ++o) {
                float sred = wave_sum(gbacc[o]);
                if (lane == 0) red[wave * CO_T + o] = sred;
            }
            __syncthreads();
            if (tid < CO_T) {
                float sred = 0.f;
#pragma unroll
                for (int i = 0; i < BWD_WAVES; ++i) sred += red[i * CO_T + tid];
                gbpart[(size_t)wslot * CO_T + tid] = sred;
            }
        }
    }
}

// gx = sum over the S partial planes, in chunk order (S > 1 only)
__global__ __launch_bounds__(256) void mdcn_bwd_gx_reduce_kernel(const float* __restrict__ part, float* __restrict__ gx,
                                                                  size_t total, int S) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float s = part[i];
        for (int j = 1; j < S; ++j) s += part[(size_t)j * total + i];
        gx[i] = s;
    }
}

// grad_weight[o][ci][k] += sum over the `slots` workgroup rows (row order) of the partials; same for grad_bias
template <int CO_T>
__global__ __launch_bounds__(256) void mdcn_bwd_param_reduce_kernel(const float* __restrict__ gwpart,
                                                                     const float* __restrict__ gbpart, float* __restrict__ gw,
                                                                     float* __restrict__ gb, Geom g, int slots) {
    constexpr int K = 9;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nw = g.Co * g.cin_g * K;
    if (i < nw) {
        const int k = i % K, ci = (i / K) % g.cin_g, o = i / (K * g.cin_g);
        const int c = (o / g.cout_g) * g.cin_g + ci;
        const float* src = gwpart + ((size_t)c * K + k) * CO_T + o;
        float s = 0.f;
        for (int r = 0; r < slots; ++r) s += src[(size_t)r * g.C * K * CO_T];
        gw[i] += s;
    } else if (gb != nullptr && i < nw + g.Co) {
        const int o = i - nw;
        float s = 0.f;
        for (int r = 0; r < slots; ++r) s += gbpart[(size_t)r * CO_T + o];
        gb[o] += s;
    }
}

bool make_geom(Geom& g, int N, int C, int H, int W, int Co, int kh, int kw, int stride, int pad, int dil,
               int groups, int dg) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 ||
        dil <= 0 || groups <= 0 || dg <= 0)
        return false;
    if (C % groups || Co % groups || C % dg) return false;
    g.N = N; g.C = C; g.H = H; g.W = W; g.Co = Co; g.K = kh * kw; g.kh = kh; g.kw = kw;
    g.stride = stride; g.pad = pad; g.dil = dil;
    g.Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    g.Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    if (g.Ho <= 0 || g.Wo <= 0) return false;
    g.P = g.Ho * g.Wo;
    g.dg = dg; g.cpg_dg = C / dg; g.cin_g = C / groups; g.cout_g = Co / groups;
    g.LW = (W + PADL + 1 + 3) & ~3;
    g.plane = (H + 2) * g.LW;
    return true;
}

template <int CO_T, int VEC, bool K9>
int launch_fwd(const float* x, const float* off, const float* msk, const float* w, const float* bias,
               float* out, const Geom& g, float alpha, float beta, hipStream_t st) {
    constexpr int COP = (CO_T + 3) & ~3;
    size_t lds = ((size_t)(K9 ? 2 : 1) * g.plane + (size_t)g.C * g.K * COP) * sizeof(float);
    if (lds > OTP_LDS_LIMIT) return OTP_ERR_UNSUPPORTED;
    auto kern = mdcn_fwd_kernel<CO_T, VEC, K9>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    dim3 grid(otp_ceil_div(g.P, FWD_THREADS * VEC), g.N, otp_ceil_div(g.Co, CO_T));
    hipLaunchKernelGGL(kern, grid, dim3(FWD_THREADS), lds, st, x, off, msk, w, bias, out, g, alpha, beta);
    return otp_launch_status();
}

}  // namespace

// general form (any dtype / kernel / Cout / per-axis geometry, optional mask): mdcn_generic.hip
int otp_mdcn_generic_forward(const void* x, const void* off, const void* msk, const void* w, const void* bias, void* out, int N,
                             int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                             int groups, int dg, float alpha, float beta, int dtype, hipStream_t st);
size_t otp_mdcn_generic_backward_workspace(int C, int Co, int kh, int kw, int groups);
int otp_mdcn_generic_backward(const void* x, const void* off, const void* msk, const void* w, const void* gout, void* gx,
                              void* goff, void* gmsk, void* gw, void* gb, void* ws, size_t ws_bytes, int N, int C, int H, int W,
                              int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg,
                              int dtype, hipStream_t st);

extern "C" int otp_mdcn_forward_ex(const void* x, const void* offset, const void* mask, const void* weight, const void* bias,
                                   void* out, int N, int C, int H, int W, int Cout, int kh, int kw, int stride_h, int stride_w,
                                   int pad_h, int pad_w, int dil_h, int dil_w, int groups, int deformable_groups, float alpha,
                                   float beta, int dtype, void* stream) {
    if (!x || !offset || !weight || !out) return OTP_ERR_BAD_ARG;
    auto st = static_cast<hipStream_t>(stream);
    const bool iso = stride_h == stride_w && pad_h == pad_w && dil_h == dil_w;
    if (dtype != OTP_DTYPE_F32 || !iso || !mask)
        return otp_mdcn_generic_forward(x, offset, mask, weight, bias, out, N, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h,
                                        pad_w, dil_h, dil_w, groups, deformable_groups, alpha, beta, dtype, st);
    Geom g;
    if (!make_geom(g, N, C, H, W, Cout, kh, kw, stride_h, pad_h, dil_h, groups, deformable_groups)) return OTP_ERR_BAD_ARG;
    auto xf = static_cast<const float*>(x);
    auto of = static_cast<const float*>(offset);
    auto mf = static_cast<const float*>(mask);
    auto wf = static_cast<const float*>(weight);
    auto bf = static_cast<const float*>(bias);
    auto outf = static_cast<float*>(out);
    const bool k9 = (kh == 3 && kw == 3);
    int rc;
    if (k9 && Cout == 17) rc = launch_fwd<17, 1, true>(xf, of, mf, wf, bf, outf, g, alpha, beta, st);
    else if (k9) rc = launch_fwd<16, 1, true>(xf, of, mf, wf, bf, outf, g, alpha, beta, st);
    else rc = launch_fwd<16, 1, false>(xf, of, mf, wf, bf, outf, g, alpha, beta, st);
    if (rc == OTP_ERR_UNSUPPORTED)            // e.g. the transposed weights of a wide layer do not fit LDS: general form
        rc = otp_mdcn_generic_forward(x, offset, mask, weight, bias, out, N, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h,
                                      pad_w, dil_h, dil_w, groups, deformable_groups, alpha, beta, dtype, st);
    return rc;
}

extern "C" int otp_mdcn_forward(const void* x, const void* offset, const void* mask, const void* weight,
                                const void* bias, void* out, int N, int C, int H, int W, int Cout, int kh,
                                int kw, int stride, int pad, int dil, int groups, int deformable_groups,
                                float alpha, float beta, int dtype, void* stream) {
    return otp_mdcn_forward_ex(x, offset, mask, weight, bias, out, N, C, H, W, Cout, kh, kw, stride, stride, pad, pad, dil, dil,
                               groups, deformable_groups, alpha, beta, dtype, stream);
}

namespace {

// Launch plan of the fused fp32 3x3 backward: pixel chunks per (image, deformable group) and the workspace it needs
// ([S partial grad_x planes when S > 1][grad_weight partials per workgroup row][grad_bias partials per workgroup row]).
constexpr int BWD_CO_T = 17;
struct BwdPlan { int S, chunk; size_t gx_part, gw_part, gb_part; };

size_t bwd_need(const Geom& g, int S) {
    const size_t rows = (size_t)g.N * S;
    return ((S > 1 ? (size_t)S * g.N * g.C * g.H * g.W : 0) + rows * g.C * 9 * BWD_CO_T + rows * BWD_CO_T) * sizeof(float);
}

BwdPlan bwd_plan(const Geom& g, size_t workspace_bytes) {
    // pixel chunks: enough workgroups (one per CU) to fill 256 CUs a few times over
    const int wgs = g.N * g.dg;
    int S = 1;
    while (S < BWD_SMAX && wgs * S < 1024 && g.P / (S * 2) >= BWD_THREADS) S *= 2;
    while (S > 1 && workspace_bytes < bwd_need(g, S)) S /= 2;          // a small workspace costs parallelism, not the result
    BwdPlan p;
    p.chunk = otp_ceil_div(otp_ceil_div(g.P, S), 64) * 64;
    p.S = otp_ceil_div(g.P, p.chunk);
    p.gx_part = p.S > 1 ? (size_t)p.S * g.N * g.C * g.H * g.W : 0;
    p.gw_part = (size_t)g.N * p.S * g.C * 9 * BWD_CO_T;
    p.gb_part = (size_t)g.N * p.S * BWD_CO_T;
    return p;
}

bool bwd_fast(Geom& g, int N, int C, int H, int W, int Cout, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
              int dil_h, int dil_w, int groups, int dg, int dtype, bool has_mask, size_t* lds) {
    const bool iso = stride_h == stride_w && pad_h == pad_w && dil_h == dil_w;
    if (!(dtype == OTP_DTYPE_F32 && iso && has_mask && kh == 3 && kw == 3 && Cout <= BWD_CO_T)) return false;
    if (!make_geom(g, N, C, H, W, Cout, kh, kw, stride_h, pad_h, dil_h, groups, dg)) return false;
    constexpr int COP = (BWD_CO_T + 3) & ~3;
    *lds = ((size_t)3 * g.plane + 9 * COP + BWD_WAVES * BWD_CO_T) * sizeof(float);
    return *lds <= OTP_LDS_LIMIT;
}

}  // namespace

// exact size for one call (the fused fp32 3x3 form chooses its pixel chunks from the geometry)
extern "C" size_t otp_mdcn_backward_workspace_ex(int N, int C, int H, int W, int Cout, int kh, int kw, int stride_h, int stride_w,
                                                 int pad_h, int pad_w, int dil_h, int dil_w, int groups,
                                                 int deformable_groups, int dtype, int has_mask) {
    Geom g;
    size_t lds = 0;
    const size_t generic = otp_mdcn_generic_backward_workspace(C, Cout, kh, kw, groups > 0 ? groups : 1);
    if (!bwd_fast(g, N, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups, deformable_groups, dtype,
                  has_mask != 0, &lds))
        return generic;
    const BwdPlan p = bwd_plan(g, (size_t)-1);
    const size_t need = (p.gx_part + p.gw_part + p.gb_part) * sizeof(float);
    return need > generic ? need : generic;
}

// upper bound over every geometry with this input size (most pixel chunks, most workgroup rows)
extern "C" size_t otp_mdcn_backward_workspace(int N, int C, int H, int W, int Cout, int kh, int kw) {
    const size_t generic = otp_mdcn_generic_backward_workspace(C, Cout, kh, kw, 1);      // groups = 1 is the largest case
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return generic;
    const size_t rows = (size_t)N * BWD_SMAX;
    const size_t need = ((size_t)BWD_SMAX * N * C * H * W + rows * C * 9 * BWD_CO_T + rows * BWD_CO_T) * sizeof(float);
    return need > generic ? need : generic;
}

extern "C" int otp_mdcn_backward_ex(const void* x, const void* offset, const void* mask, const void* weight,
                                    const void* grad_out, void* grad_x, void* grad_offset, void* grad_mask, void* grad_weight,
                                    void* grad_bias, void* workspace, size_t workspace_bytes, int N, int C, int H, int W,
                                    int Cout, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                                    int dil_w, int groups, int deformable_groups, int dtype, void* stream) {
    if (!x || !offset || !weight || !grad_out || !grad_x || !grad_offset || !grad_weight) return OTP_ERR_BAD_ARG;
    if (mask && !grad_mask) return OTP_ERR_BAD_ARG;
    auto st = static_cast<hipStream_t>(stream);
    Geom g;
    size_t lds = 0;
    if (!bwd_fast(g, N, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups, deformable_groups, dtype,
                  mask != nullptr, &lds))
        return otp_mdcn_generic_backward(x, offset, mask, weight, grad_out, grad_x, grad_offset, grad_mask, grad_weight, grad_bias,
                                         workspace, workspace_bytes, N, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w,
                                         dil_h, dil_w, groups, deformable_groups, dtype, st);
    const BwdPlan p = bwd_plan(g, workspace ? workspace_bytes : 0);
    if (!workspace || workspace_bytes < (p.gx_part + p.gw_part + p.gb_part) * sizeof(float)) return OTP_ERR_WORKSPACE;
    float* gxpart = static_cast<float*>(workspace);
    float* gwpart = gxpart + p.gx_part;
    float* gbpart = gwpart + p.gw_part;
    auto kern = mdcn_bwd_kernel<BWD_CO_T>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(p.S, deformable_groups, N), dim3(BWD_THREADS), lds, st,
                       static_cast<const float*>(x), static_cast<const float*>(offset),
                       static_cast<const float*>(mask), static_cast<const float*>(weight),
                       static_cast<const float*>(grad_out), p.S > 1 ? gxpart : static_cast<float*>(grad_x),
                       static_cast<float*>(grad_offset), static_cast<float*>(grad_mask), gwpart,
                       grad_bias ? gbpart : nullptr, g, p.chunk);
    if (p.S > 1) {
        const size_t total = (size_t)N * C * H * W;
        const size_t blocks = (total + 255) / 256;
        hipLaunchKernelGGL(mdcn_bwd_gx_reduce_kernel, dim3(blocks > 4096 ? 4096 : (unsigned)blocks), dim3(256), 0, st, gxpart,
                           static_cast<float*>(grad_x), total, p.S);
    }
    const int nw = g.Co * g.cin_g * 9 + g.Co;
    hipLaunchKernelGGL(mdcn_bwd_param_reduce_kernel<BWD_CO_T>, dim3(otp_ceil_div(nw, 256)), dim3(256), 0, st, gwpart,
                       grad_bias ? gbpart : nullptr, static_cast<float*>(grad_weight), static_cast<float*>(grad_bias), g,
                       N * p.S);
    return otp_launch_status();
}

extern "C" int otp_mdcn_backward(const void* x, const void* offset, const void* mask, const void* weight,
                                 const void* grad_out, void* grad_x, void* grad_offset, void* grad_mask,
                                 void* grad_weight, void* grad_bias, void* workspace, size_t workspace_bytes,
                                 int N, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad,
                                 int dil, int groups, int deformable_groups, int dtype, void* stream) {
    return otp_mdcn_backward_ex(x, offset, mask, weight, grad_out, grad_x, grad_offset, grad_mask, grad_weight, grad_bias,
                                workspace, workspace_bytes, N, C, H, W, Cout, kh, kw, stride, stride, pad, pad, dil, dil, groups,
                                deformable_groups, dtype, stream);
}
