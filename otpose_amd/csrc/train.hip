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
#include <stdlib.h>

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
// 3x3: a wave owns up to 7 of the 27 (tap, ci-block) combinations x 3 co-blocks = 21 accumulator tiles; per 4-pixel
// step it reads the three dY fragments once and one input fragment per combination (11 LDS reads : 21 MFMAs).
// 1x1: every wave owns all (ci-block, co-block) tiles and takes every fourth pixel step.
// The accumulators stay in registers over every tile the workgroup walks and leave through one float atomic per
// weight at the end.
constexpr int WG_CO = 48, WG_CI = 48, WG_GROUPS = 7, SU = 8, SV = 6;

struct WgradPlan {
    int N, Cin, H, W, Cout, KS, stride, pad, dil, Ho, Wo;
    int x_ctot, x_coff, dy_ctot, dy_coff;
    int RT, WT, PT, PS;       // output rows x columns per tile, pixels (rounded to 4), dY row pitch (PS % 32 == 2)
    int NRX, LWP, CSX;        // staged input rows, staged row pitch, channel stride (CSX % 32 == 2)
    int sparse;               // dilated kernels: stage only the KS rows a tap row touches per output row (r = ti*RT + yl)
    int tiles_x, tiles_per_img, ntiles;
    int vec, GX, GD;          // 16-byte staging path; float4 groups per staged input row / per dY row
    int pipe, NPD, NPX;       // software-pipelined staging (full-width tiles): float4 items per thread for dY / x
    unsigned magicRT, magicNRX, magicWT, magicLWP, magicGX, magicGD;   // ceil(2^32 / d): exact quotients of the small (< 2^16) staging indices
};

__device__ __forceinline__ int magic_div(int n, unsigned magic, int d) {
    return d == 1 ? n : (int)__umulhi((unsigned)n, magic);
}

// the MFMA loop of one wave over one staged tile: NG (tap, ci-block) combinations x up to 3 co-blocks
template <int NG>
__device__ __forceinline__ void wgrad_steps(f32x4 (&acc)[WG_GROUPS][3], const float* __restrict__ dys,
                                            const float* __restrict__ xs, const int* __restrict__ poff,
                                            const int (&bbase)[WG_GROUPS], int abase, int PS16, int s0, int ds, int steps,
                                            int ncob) {
    const int kl = (threadIdx.x & 63) >> 4;
    for (int s = s0; s < steps; s += ds) {
        const int xo = poff[4 * s + kl];
        const int ao = abase + 4 * s;
        const float a0 = dys[ao];
        const float a1 = ncob > 1 ? dys[ao + PS16] : 0.f;
        const float a2 = ncob > 2 ? dys[ao + 2 * PS16] : 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float b = xs[bbase[g] + xo];
            acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[g][0], 0, 0, 0);
            if (ncob > 1) acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[g][1], 0, 0, 0);
            if (ncob > 2) acc[g][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, acc[g][2], 0, 0, 0);
        }
    }
}

// partial sums: part != nullptr -> each workgroup stores its accumulators in fragment order
// [workgroup][wave][g][co-block][r][lane] (coalesced) and wgrad_reduce_kernel folds them into dW; part == nullptr ->
// one float atomic per weight and workgroup (512 atomics per weight cost more than the whole MFMA loop at the HRNet sizes)
constexpr int WG_SLOTS = 4 * WG_GROUPS * 3 * 4 * 64;

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ part,
                                                              const WgradPlan P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;                                   // [48][PS]
    float* xs = smem + WG_CO * P.PS;                     // [48][CSX]
    int* poff = reinterpret_cast<int*>(xs + WG_CI * P.CSX);   // [PT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kl = lane >> 4;
    const int co0 = blockIdx.y * WG_CO, ci0 = blockIdx.z * WG_CI;
    const int KK = P.KS * P.KS;
    const int ncib = min(3, (P.Cin - ci0 + 15) >> 4), ncob = min(3, (P.Cout - co0 + 15) >> 4);
    // combinations (tap, ci-block) of this wave: 3x3 -> round-robin over the waves; 1x1 -> all of them, split pixel steps
    const bool split_steps = KK == 1;
    const int ncombo = KK * ncib;
    const int ng = split_steps ? ncombo : (ncombo > wave ? (ncombo - wave + 3) >> 2 : 0);
    const int s0 = split_steps ? wave : 0, ds = split_steps ? 4 : 1;

    f32x4 acc[WG_GROUPS][3];
    int bbase[WG_GROUPS];                                 // LDS word of the input fragment of combination g, pixel offset 0
#pragma unroll
    for (int g = 0; g < WG_GROUPS; ++g) {
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[g][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int combo = split_steps ? g : wave + 4 * g;
        const int tap = combo / ncib, cib = combo - tap * ncib;
        const int ti = tap / P.KS, tj = tap - ti * P.KS;
        bbase[g] = (cib * 16 + i16) * P.CSX + ti * (P.sparse ? P.RT : P.dil) * P.LWP + tj * P.dil;
    }
    const int abase = i16 * P.PS + kl, PS16 = 16 * P.PS;

    // pixel -> input offset table (tile-invariant: local row * stride * LWP + column * stride)
    for (int p = tid; p < P.PT; p += 256) {
        const int yl = p / P.WT, xq = p - yl * P.WT;
        poff[p] = yl < P.RT ? yl * (P.sparse ? 1 : P.stride) * P.LWP + xq * P.stride : 0;   // rounding pixels: any staged word
    }
    // the rounding pixels of the dY rows (PT - RT*WT <= 3 per channel) are never staged: zero them once
    for (int i = tid; i < WG_CO * (P.PT - P.RT * P.WT); i += 256) {
        const int c = i / (P.PT - P.RT * P.WT), q = i - c * (P.PT - P.RT * P.WT);
        dys[c * P.PS + P.RT * P.WT + q] = 0.f;
    }

    if (P.pipe) {
        // ---- software-pipelined tile loop (16-byte staging, full-width tiles): the loads of tile t+1 are in flight in
        // registers while tile t is multiplied; a thread's items are the same for every tile up to the tile's row offset,
        // so each is one packed (channel, row, group) word decoded at issue and at store time --------------------------------
        constexpr int MPD = 6, MPX = 12;
        otp_f32x4 pd[MPD], px[MPX];
        int kd[MPD], kx[MPX];                                          // (c << 16) | (row << 8) | group, -1: no item
        const int totd = WG_CO * P.RT * P.GD, totx = WG_CI * P.NRX * P.GX;
#pragma unroll
        for (int u = 0; u < MPD; ++u) {
            const int i = tid + 256 * u;
            const int cr = magic_div(i, P.magicGD, P.GD), g = i - cr * P.GD;
            const int c = magic_div(cr, P.magicRT, P.RT), yl = cr - c * P.RT;
            kd[u] = (u < P.NPD && i < totd) ? (c << 16) | (yl << 8) | g : -1;
        }
#pragma unroll
        for (int u = 0; u < MPX; ++u) {
            const int i = tid + 256 * u;
            const int cr = magic_div(i, P.magicGX, P.GX), g = i - cr * P.GX;
            const int c = magic_div(cr, P.magicNRX, P.NRX), r = cr - c * P.NRX;
            kx[u] = (u < P.NPX && i < totx) ? (c << 16) | (r << 8) | g : -1;
        }
        const int xg0 = (0 - P.pad) & ~3;                              // x0 == 0: aligned first image column of group 0
        const int colb = xg0 + P.pad;                                  // staged column of element 0 of group 0 (-3 .. 0)
        auto issue_dy = [&](int tile) __attribute__((always_inline)) {
            const int n = tile / P.tiles_per_img;
            const int y0 = (tile - n * P.tiles_per_img) * P.RT;
            const int rows = min(P.RT, P.Ho - y0);
            const otp_rsrc rdy = make_rsrc32(dy + ((size_t)n * P.dy_ctot + P.dy_coff) * P.Ho * P.Wo,
                                             (unsigned)P.Cout * (unsigned)(P.Ho * P.Wo) * 4u);
#pragma unroll
            for (int u = 0; u < MPD; ++u) {
                const int c = kd[u] >> 16, yl = (kd[u] >> 8) & 255, g = kd[u] & 255;
                const bool ok = kd[u] >= 0 && co0 + c < P.Cout && yl < rows;
                pd[u] = bload4(rdy, ok ? (((co0 + c) * P.Ho + y0 + yl) * P.Wo + 4 * g) * 4 : -1);
            }
        };
        auto issue = [&](int tile) __attribute__((always_inline)) {
            const int n = tile / P.tiles_per_img;
            const int y0 = (tile - n * P.tiles_per_img) * P.RT;
            const int r0 = y0 - P.pad;
            const otp_rsrc rx = make_rsrc32(x + ((size_t)n * P.x_ctot + P.x_coff) * P.H * P.W,
                                            (unsigned)P.Cin * (unsigned)(P.H * P.W) * 4u);
#pragma unroll
            for (int u = 0; u < MPX; ++u) {
                const int c = kx[u] >> 16, r = (kx[u] >> 8) & 255, g = kx[u] & 255;
                const int yy = r0 + r, xg = xg0 + 4 * g;
                const bool ok = kx[u] >= 0 && ci0 + c < P.Cin && yy >= 0 && yy < P.H && xg >= 0 && xg < P.W;
                px[u] = bload4(rx, ok ? (((ci0 + c) * P.H + yy) * P.W + xg) * 4 : -1);
            }
            asm volatile("" ::: "memory");
        };
        auto store = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < MPD; ++u)
                if (kd[u] >= 0) {
                    float* d = dys + (kd[u] >> 16) * P.PS + ((kd[u] >> 8) & 255) * P.WT + 4 * (kd[u] & 255);
                    *reinterpret_cast<float2*>(d) = make_float2(pd[u][0], pd[u][1]);
                    *reinterpret_cast<float2*>(d + 2) = make_float2(pd[u][2], pd[u][3]);
                }
#pragma unroll
            for (int u = 0; u < MPX; ++u)
                if (kx[u] >= 0) {
                    const int col = colb + 4 * (kx[u] & 255);
                    float* d = xs + (kx[u] >> 16) * P.CSX + ((kx[u] >> 8) & 255) * P.LWP + col;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e >= 0 && col + e < P.LWP) d[e] = px[u][e];
                }
        };
        int tile = blockIdx.x;
        if (tile < P.ntiles) issue(tile);
        for (; tile < P.ntiles; tile += gridDim.x) {
            const int y0 = (tile - (tile / P.tiles_per_img) * P.tiles_per_img) * P.RT;
            const int rows = min(P.RT, P.Ho - y0);
            // keep the packed item words opaque per iteration: otherwise their decoded addresses are hoisted out of the tile loop
            // into ~50 more registers (spills)
#pragma unroll
            for (int u = 0; u < MPD; ++u) asm volatile("" : "+v"(kd[u]));
#pragma unroll
            for (int u = 0; u < MPX; ++u) asm volatile("" : "+v"(kx[u]));
            issue_dy(tile);                                            // the small operand: loaded here, its latency overlaps the barrier
            __syncthreads();                                           // previous tile fully consumed
            store();
            __syncthreads();
            if (tile + (int)gridDim.x < P.ntiles) issue(tile + (int)gridDim.x);   // the large operand of the next tile: under the MFMAs
            const int steps = (rows * P.WT + 3) >> 2;
            switch (ng) {
                case 7: wgrad_steps<7>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
                case 6: wgrad_steps<6>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
                case 5: wgrad_steps<5>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
                case 4: wgrad_steps<4>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
                case 3: wgrad_steps<3>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
                case 2: wgrad_steps<2>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
                case 1: wgrad_steps<1>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
                default: break;
            }
        }
    } else
    for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const int n = tile / P.tiles_per_img;
        const int tin = tile - n * P.tiles_per_img;
        const int yb = tin / P.tiles_x, xb = tin - yb * P.tiles_x;
        const int y0 = yb * P.RT, x0 = xb * P.WT;                      // first output row / column of the tile
        const int rows = min(P.RT, P.Ho - y0), cols = min(P.WT, P.Wo - x0);
        __syncthreads();                                               // previous tile fully consumed
        if (P.vec) {
            // ---- 16-byte staging (stride 1, W % 4 == 0): one buffer load per 4 columns, SV groups in flight per lane ----
            const otp_rsrc rdy = make_rsrc32(dy + ((size_t)n * P.dy_ctot + P.dy_coff) * P.Ho * P.Wo,
                                             (unsigned)P.Cout * (unsigned)(P.Ho * P.Wo) * 4u);
            const otp_rsrc rx = make_rsrc32(x + ((size_t)n * P.x_ctot + P.x_coff) * P.H * P.W,
                                            (unsigned)P.Cin * (unsigned)(P.H * P.W) * 4u);
            const int totd = WG_CO * P.RT * P.GD, totx = WG_CI * P.NRX * P.GX;
            const int r0 = y0 * P.stride - P.pad, xx0 = x0 - P.pad;
            const int xg0 = (x0 - P.pad) & ~3;                         // first (aligned) image column of group 0; may be < 0
            for (int base = tid; base < totd; base += 256 * SV) {
                otp_f32x4 v[SV];
                int dst[SV];
#pragma unroll
                for (int u = 0; u < SV; ++u) {
                    const int i = base + u * 256;
                    const int cr = magic_div(i, P.magicGD, P.GD), xq = (i - cr * P.GD) * 4;
                    const int c = magic_div(cr, P.magicRT, P.RT), yl = cr - c * P.RT;
                    const bool ok = i < totd && co0 + c < P.Cout && yl < rows && xq < cols;
                    v[u] = bload4(rdy, ok ? (((co0 + c) * P.Ho + y0 + yl) * P.Wo + x0 + xq) * 4 : -1);
                    dst[u] = i < totd ? c * P.PS + yl * P.WT + xq : -1;
                }
#pragma unroll
                for (int u = 0; u < SV; ++u)
                    if (dst[u] >= 0) {
                        *reinterpret_cast<float2*>(dys + dst[u]) = make_float2(v[u][0], v[u][1]);
                        *reinterpret_cast<float2*>(dys + dst[u] + 2) = make_float2(v[u][2], v[u][3]);
                    }
            }
            for (int base = tid; base < totx; base += 256 * SV) {
                otp_f32x4 v[SV];
                int dst[SV];
#pragma unroll
                for (int u = 0; u < SV; ++u) {
                    const int i = base + u * 256;
                    const int cr = magic_div(i, P.magicGX, P.GX), j = i - cr * P.GX;
                    const int c = magic_div(cr, P.magicNRX, P.NRX), r = cr - c * P.NRX;
                    const int yy = r0 + r, xg = xg0 + 4 * j;
                    const bool ok = i < totx && ci0 + c < P.Cin && yy >= 0 && yy < P.H && xg >= 0 && xg < P.W;
                    v[u] = bload4(rx, ok ? (((ci0 + c) * P.H + yy) * P.W + xg) * 4 : -1);
                    dst[u] = i < totx ? c * P.CSX + r * P.LWP + (xg - xx0) : -(1 << 20);   // LDS word of element 0 (col may be < 0)
                }
#pragma unroll
                for (int u = 0; u < SV; ++u)
                    if (dst[u] > -(1 << 19)) {
                        const int i = base + u * 256;
                        const int cr = magic_div(i, P.magicGX, P.GX), j = i - cr * P.GX;
                        const int col = xg0 - xx0 + 4 * j;                 // staged column of element 0 (-3 .. LWP-1)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e >= 0 && col + e < P.LWP) xs[dst[u] + e] = v[u][e];
                    }
            }
        } else {
        // ---- stage dY rows: [co][yl * WT + xq], zero past the image rows / columns / channels ---------------------
        // (flat item index, SU loads in flight per lane before the first LDS store: the loop is latency-bound otherwise)
        {
            const otp_rsrc rdy = make_rsrc32(dy + ((size_t)n * P.dy_ctot + P.dy_coff) * P.Ho * P.Wo,
                                             (unsigned)P.Cout * (unsigned)(P.Ho * P.Wo) * 4u);
            const int total = WG_CO * P.RT * P.WT;
            for (int base = tid; base < total; base += 256 * SU) {
                float v[SU];
                int dst[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = base + u * 256;
                    const int cr = magic_div(i, P.magicWT, P.WT), xq = i - cr * P.WT;
                    const int c = magic_div(cr, P.magicRT, P.RT), yl = cr - c * P.RT;
                    const bool ok = i < total && co0 + c < P.Cout && yl < rows && xq < cols;
                    v[u] = bload(rdy, ok ? (((co0 + c) * P.Ho + y0 + yl) * P.Wo + x0 + xq) * 4 : -1, 0);
                    dst[u] = i < total ? c * P.PS + yl * P.WT + xq : -1;
                }
#pragma unroll
                for (int u = 0; u < SU; ++u)
                    if (dst[u] >= 0) dys[dst[u]] = v[u];
            }
        }
        // ---- stage the input rows with a zero halo: [ci][r][col], col 0 = image column x0*stride - pad ----------------
        {
            const otp_rsrc rx = make_rsrc32(x + ((size_t)n * P.x_ctot + P.x_coff) * P.H * P.W,
                                            (unsigned)P.Cin * (unsigned)(P.H * P.W) * 4u);
            const int r0 = y0 * P.stride - P.pad, xx0 = x0 * P.stride - P.pad;
            const int total = WG_CI * P.NRX * P.LWP;
            for (int base = tid; base < total; base += 256 * SU) {
                float v[SU];
                int dst[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = base + u * 256;
                    const int cr = magic_div(i, P.magicLWP, P.LWP), col = i - cr * P.LWP;
                    const int c = magic_div(cr, P.magicNRX, P.NRX), r = cr - c * P.NRX;
                    int yy = r0 + r;
                    if (P.sparse) {
                        const int rt = magic_div(r, P.magicRT, P.RT);          // tap row
                        yy = (y0 + r - rt * P.RT) * P.stride - P.pad + rt * P.dil;
                    }
                    const int xx = xx0 + col;
                    const bool ok = i < total && ci0 + c < P.Cin && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
                    v[u] = bload(rx, ok ? (((ci0 + c) * P.H + yy) * P.W + xx) * 4 : -1, 0);
                    dst[u] = i < total ? c * P.CSX + cr * P.LWP - c * P.NRX * P.LWP + col : -1;
                }
#pragma unroll
                for (int u = 0; u < SU; ++u)
                    if (dst[u] >= 0) xs[dst[u]] = v[u];
            }
        }
        }
        __syncthreads();
        // ---- MFMA over the tile's pixels --------------------------------------------------------------------
        const int steps = (rows * P.WT + 3) >> 2;
        switch (ng) {
            case 7: wgrad_steps<7>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
            case 6: wgrad_steps<6>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
            case 5: wgrad_steps<5>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
            case 4: wgrad_steps<4>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
            case 3: wgrad_steps<3>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
            case 2: wgrad_steps<2>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
            case 1: wgrad_steps<1>(acc, dys, xs, poff, bbase, abase, PS16, s0, ds, steps, ncob); break;
            default: break;
        }
    }
    // ---- flush: D[row = co (kl*4 + r)][col = ci (i16)] -> partial buffer (fragment order) or dW atomics ----------
    if (part) {
        float* dst = part + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) *
                                (WG_GROUPS * 3 * 4 * 64) + lane;
#pragma unroll
        for (int g = 0; g < WG_GROUPS; ++g)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((g * 3 + b) * 4 + r) * 64] = acc[g][b][r];
        return;
    }
#pragma unroll
    for (int g = 0; g < WG_GROUPS; ++g) {
        if (g < ng) {
            const int combo = split_steps ? g : wave + 4 * g;
            const int tap = combo / ncib, cib = combo - tap * ncib;
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + b * 16 + kl * 4 + r, ci = ci0 + cib * 16 + i16;
                    if (b < ncob && co < P.Cout && ci < P.Cin)
                        atomicAdd(&dw[((size_t)co * P.Cin + ci) * KK + tap], acc[g][b][r]);
                }
        }
    }
}

// dW[co][ci][tap] += sum over the gx workgroups of one (co-group, ci-group) of their partial accumulators, in a FIXED order
// (round 4: no float atomics - the result is the same bits on every run).  grid (blocks of 32 per-wave fragment slots,
// co-group * ci-group); 1024 threads = 32 slots x 8 segments of the workgroup range x the four waves' copies of a slot; the
// segments meet in LDS and are added in order.  With 1x1 kernels the waves split the pixel steps of the same (co, ci) tile, so
// their sums are added (wave order) into ONE weight; otherwise each wave's slot is a weight of its own.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int gx,
                                                             int gy, int Cout, int Cin, int KS) {
    __shared__ float red[4][8][32];
    constexpr int WSLOTS = WG_SLOTS / 4;                                // slots of one wave: [g][b][r][lane]
    // 1024 threads = 32 slots x 8 segments x the 4 waves' copies (256 threads walking the four copies one after the other were a
    // latency chain: 35 us per launch beside the backward's other kernels)
    const int sl = threadIdx.x & 31, sg = (threadIdx.x >> 5) & 7, wave = threadIdx.x >> 8;
    const int sub = blockIdx.x * 32 + sl;
    const int seg = (gx + 7) >> 3;
    const int w0 = sg * seg, w1 = min(gx, w0 + seg);
    // which of the four waves' copies of this slot hold a weight at all (most of the 48 x 48 x 9 slot space is empty for the
    // narrow layers this fp32 path serves: their partials are never read)
    const int lane_ = sub & 63, r_ = (sub >> 6) & 3, gb_ = sub >> 8;
    const int b_ = gb_ % 3, g_ = gb_ / 3;
    const int by_ = blockIdx.y % gy, bz_ = blockIdx.y / gy;
    const int ncib_ = min(3, (Cin - bz_ * WG_CI + 15) >> 4), ncob_ = min(3, (Cout - by_ * WG_CO + 15) >> 4);
    const int KK_ = KS * KS, ncombo_ = KK_ * ncib_;
    const bool row_ok = sub < WSLOTS && b_ < ncob_ && by_ * WG_CO + b_ * 16 + (lane_ >> 4) * 4 + r_ < Cout;
    {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const int combo_ = KK_ == 1 ? g_ : wave + 4 * g_;
        const bool live = row_ok && combo_ < ncombo_ &&
                          bz_ * WG_CI + (combo_ - (combo_ / ncib_) * ncib_) * 16 + (lane_ & 15) < Cin;
        if (live && w1 > w0) {
            const float* src = part + ((size_t)blockIdx.y * gx + w0) * WG_SLOTS + wave * WSLOTS + sub;
            int w = w0;
            for (; w + 3 < w1; w += 4, src += 4 * (size_t)WG_SLOTS) {
                s0 += src[0]; s1 += src[WG_SLOTS]; s2 += src[2 * (size_t)WG_SLOTS]; s3 += src[3 * (size_t)WG_SLOTS];
            }
            for (; w < w1; ++w, src += WG_SLOTS) s0 += src[0];
        }
        red[wave][sg][sl] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (threadIdx.x >= 32 || sub >= WSLOTS) return;
    const int lane = sub & 63, r = (sub >> 6) & 3, gb = sub >> 8;       // gb = g * 3 + b
    const int b = gb % 3, g = gb / 3;
    const int by = blockIdx.y % gy, bz = blockIdx.y / gy;
    const int co0 = by * WG_CO, ci0 = bz * WG_CI, KK = KS * KS;
    const int ncib = min(3, (Cin - ci0 + 15) >> 4), ncob = min(3, (Cout - co0 + 15) >> 4);
    const bool split_steps = KK == 1;
    const int ncombo = KK * ncib;
    const int co = co0 + b * 16 + (lane >> 4) * 4 + r;
    if (b >= ncob || co >= Cout) return;
    float total = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
        float v = red[wv][0][sl];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += red[wv][k][sl];
        if (split_steps) { total += v; continue; }
        const int combo = wv + 4 * g;
        if (combo >= ncombo) continue;
        const int tap = combo / ncib, cib = combo - tap * ncib;
        const int ci = ci0 + cib * 16 + (lane & 15);
        if (ci < Cin) dw[((size_t)co * Cin + ci) * KK + tap] += v;
    }
    if (split_steps && g < ncombo) {
        const int ci = ci0 + g * 16 + (lane & 15);                      // tap 0, ci block g
        if (ci < Cin) dw[(size_t)co * Cin + ci] += total;
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

// ------------------------------------------------------------------------------------------------
// 1x1 / stride-1 wgrad (the pointwise layers of the temporal encoders, the HRNet bottleneck and fuse 1x1s):
//     dW[co][ci] += sum_{n, p} dY[n, co, p] * X[n, ci, p]
// The generic kernel above walks 48 x 48 blocks of dW, so a 136 x 136 layer reads both operands three times and feeds
// every MFMA from 4-byte LDS reads.  Here a workgroup owns up to 144 x 144 of dW (9 x 9 MFMA tiles: wave w takes column
// tiles w, w + 4, w + 8 of all nine row tiles) over one chunk of pixels of one image, and both operands go straight from
// L2 into MFMA operand registers: lane (m, kq) loads the four pixels p0 + 4 kq .. + 3 of row m with one 16-byte buffer
// load (k-step q of the group contracts over the pixels {p0 + 4 kq + q}, the same permutation on both operands), rows past
// the tensor and pixels past the chunk fall off the buffer descriptor and read zeros.  The next group's loads are in flight
// while this one is multiplied.  Partial sums leave in fragment order; w1x1_reduce_kernel folds the chunks.
struct W1Plan {
    int N, Cin, Cout, HW, x_ctot, x_coff, dy_ctot, dy_coff;
    int chunk, cpi, S;          // pixels per chunk (multiple of 16), chunks per image, S = N * cpi
    int MT, NT;                 // 16-row / 16-column tiles of dW in total
};

__global__ __launch_bounds__(256, 2) void wgrad1x1_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ part, const W1Plan P) {
    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.x, n = s / P.cpi, p_begin = (s - n * P.cpi) * P.chunk;
    const int p_end = min(P.HW, p_begin + P.chunk);
    const int mt0 = blockIdx.y * 9, nt0 = blockIdx.z * 9;
    const int nmt = min(9, P.MT - mt0);
    int ntile[3], nnt = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ntile[j] = nt0 + wave + 4 * j;
        if (wave + 4 * j < 9 && ntile[j] < P.NT) nnt = j + 1;
    }
    // rows past Cout / Cin start past the descriptor's range -> zeros
    const otp_rsrc rdy = make_rsrc(dy + ((size_t)n * P.dy_ctot + P.dy_coff) * P.HW, (size_t)P.Cout * P.HW * sizeof(float));
    const otp_rsrc rx = make_rsrc(x + ((size_t)n * P.x_ctot + P.x_coff) * P.HW, (size_t)P.Cin * P.HW * sizeof(float));
    const int arow = ((mt0 * 16 + m) * P.HW + 4 * kq) * 4;            // byte offset of (row, pixel 4 kq) for row tile 0
    int brow[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) brow[j] = ((ntile[j] * 16 + m) * P.HW + 4 * kq) * 4;
    const int tstep = 16 * P.HW * 4;                                   // next row tile

    f32x4 acc[9][3];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 A[9], B[3], An[9], Bn[3];
    auto load = [&](f32x4 (&a)[9], f32x4 (&b)[3], int p0) __attribute__((always_inline)) {
        const bool ok = p0 + 4 * kq < p_end;                           // HW % 4 == 0: a pixel quad is inside or outside
        const int po = p0 * 4;                                         // offset -1 = past every descriptor -> zeros
#pragma unroll
        for (int i = 0; i < 9; ++i) a[i] = bload4(rdy, (ok && i < nmt) ? arow + po + i * tstep : -1);
#pragma unroll
        for (int j = 0; j < 3; ++j) b[j] = bload4(rx, (ok && j < nnt) ? brow[j] + po : -1);
    };
    load(A, B, p_begin);
    for (int p0 = p_begin; p0 < p_end; p0 += 16) {
        if (p0 + 16 < p_end) load(An, Bn, p0 + 16);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (i < nmt && j < nnt)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i][q], B[j][q], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 9; ++i) A[i] = An[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) B[j] = Bn[j];
    }
    // part[s][row tile][column tile][lane][4]
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (i < nmt && j < nnt)
                *reinterpret_cast<f32x4*>(part + ((((size_t)s * P.MT + mt0 + i) * P.NT + ntile[j]) * 64 + lane) * 4) = acc[i][j];
}

// grid (tiles, 4 quarters of a tile); 256 threads = 16 float4 lanes x 16 chunk groups
__global__ __launch_bounds__(256) void w1x1_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int MT,
                                                           int NT, int Cout, int Cin) {
    __shared__ f32x4 red[16][16];
    const int tile = blockIdx.x, mt = tile / NT, nt = tile - mt * NT;
    const int l4 = threadIdx.x & 15, sg = threadIdx.x >> 4, lane = blockIdx.y * 16 + l4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)MT * NT * 256;
    const float* src = part + ((size_t)tile * 64 + lane) * 4;
    for (int s = sg; s < S; s += 16) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * stride);
    red[sg][l4] = v;
    __syncthreads();
    if (sg == 0) {
#pragma unroll
        for (int k = 1; k < 16; ++k) v += red[k][l4];
        const int ci = nt * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = mt * 16 + (lane >> 4) * 4 + r;
            if (co < Cout && ci < Cin) dw[(size_t)co * Cin + ci] += v[r];
        }
    }
}

static bool w1x1_plan(W1Plan& P, int N, int Cin, int Cout, int HW) {
    P.N = N; P.Cin = Cin; P.Cout = Cout; P.HW = HW;
    P.MT = otp_ceil_div(Cout, 16); P.NT = otp_ceil_div(Cin, 16);
    const int blocks = otp_ceil_div(P.MT, 9) * otp_ceil_div(P.NT, 9);
    int want = 512 / blocks;                                 // chunks in total: two workgroups per CU
    if (want < 1) want = 1;
    int cpi = otp_ceil_div(want, N);
    int chunk = otp_ceil_div(otp_ceil_div(HW, cpi), 16) * 16;
    if (chunk < 64) chunk = 64;
    cpi = otp_ceil_div(HW, chunk);
    P.chunk = chunk; P.cpi = cpi; P.S = N * cpi;
    return true;
}
constexpr size_t W1_MAX_PART_BYTES = 96u << 20;

static int wgrad_grid_x(int Cout, int Cin) {
    const int gy = otp_ceil_div(Cout, WG_CO), gz = otp_ceil_div(Cin, WG_CI);
    int gx = 512 / (gy * gz);
    return gx < 8 ? 8 : gx;
}

extern "C" size_t otp_conv2d_wgrad_workspace(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    const size_t generic = (size_t)wgrad_grid_x(Cout, Cin) * otp_ceil_div(Cout, WG_CO) * otp_ceil_div(Cin, WG_CI) * WG_SLOTS * sizeof(float);
    return generic > W1_MAX_PART_BYTES ? generic : W1_MAX_PART_BYTES;      // the 1x1 path keeps one 1 KB tile per chunk
}

extern "C" int otp_conv2d_wgrad(const void* x, const void* grad_out, void* grad_weight, int N, int Cin, int H, int W,
                                int Cout, int kh, int kw, int stride, int pad, int dil, int x_ctot, int x_coff,
                                int dy_ctot, int dy_coff, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !grad_out || !grad_weight || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || stride <= 0 ||
        pad < 0 || dil <= 0)
        return OTP_ERR_BAD_ARG;
    if (kh != kw || (kh != 1 && kh != 3)) return OTP_ERR_UNSUPPORTED;
    if (x_ctot < x_coff + Cin || dy_ctot < dy_coff + Cout) return OTP_ERR_BAD_ARG;
    if (kh == 1 && stride == 1 && pad == 0 && (H * W) % 4 == 0 && workspace && Cin >= 16 && Cout >= 16 &&
        (long)x_ctot * H * W < (1l << 28) && (long)dy_ctot * H * W < (1l << 28) &&
        !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_out)) & 15)) {
        static const bool off = [] { const char* e = getenv("OTP_WGRAD_1X1"); return e && e[0] == '0'; }();
        W1Plan Q{};
        w1x1_plan(Q, N, Cin, Cout, H * W);
        Q.x_ctot = x_ctot; Q.x_coff = x_coff; Q.dy_ctot = dy_ctot; Q.dy_coff = dy_coff;
        const size_t need = (size_t)Q.S * Q.MT * Q.NT * 256 * sizeof(float);
        // measured against the generic kernel (tools/wgrad_bench.py): it wins from 3 x 12 tiles of dW upwards (136 -> 136
        // 0.166 -> 0.106 ms, 136 <-> 544 0.64 -> 0.35, 256 -> 64 1.07 -> 0.86, 384 -> 48 0.053 -> 0.028) and loses on the
        // thin ones (96 -> 48 0.066 -> 0.104, 408 -> 17 0.16 -> 0.21)
        const bool pays = (Q.MT < Q.NT ? Q.MT : Q.NT) >= 3 && Q.MT * Q.NT >= 36;
        if (!off && pays && need <= workspace_bytes && ((x_coff * H * W) % 4 == 0) && ((dy_coff * H * W) % 4 == 0)) {
            auto st = static_cast<hipStream_t>(stream);
            hipLaunchKernelGGL(wgrad1x1_kernel, dim3(Q.S, otp_ceil_div(Q.MT, 9), otp_ceil_div(Q.NT, 9)), dim3(256), 0, st,
                               static_cast<const float*>(x), static_cast<const float*>(grad_out),
                               static_cast<float*>(workspace), Q);
            hipLaunchKernelGGL(w1x1_reduce_kernel, dim3(Q.MT * Q.NT, 4), dim3(256), 0, st,
                               static_cast<const float*>(workspace), static_cast<float*>(grad_weight), Q.S, Q.MT, Q.NT, Cout,
                               Cin);
            return otp_launch_status();
        }
    }
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
    size_t lds = 0;
    auto size_for = [&](int wt, int rt) {
        P.WT = wt; P.RT = rt;
        P.tiles_x = otp_ceil_div(P.Wo, wt);
        P.LWP = (wt - 1) * stride + (kw - 1) * dil + 1;
        P.PT = (rt * wt + 3) & ~3;
        P.PS = pad2(P.PT + 4);
        P.NRX = P.sparse ? kh * rt : (rt - 1) * stride + (kh - 1) * dil + 1;
        P.CSX = pad2(P.NRX * P.LWP + 4);
        return ((size_t)WG_CO * P.PS + (size_t)WG_CI * P.CSX + P.PT) * sizeof(float);
    };
    int wt = P.Wo <= 160 ? P.Wo : 128;
    while (wt > 16 && size_for(wt, 1) > 72 * 1024) wt = (wt / 2 + 3) & ~3;   // wide strided rows (the 384x288 stem)
    for (int rt = P.Ho; rt >= 1; --rt) {
        lds = size_for(wt, rt);
        if (lds <= 72 * 1024) break;
        if (rt == 1 && lds > OTP_LDS_LIMIT) return OTP_ERR_UNSUPPORTED;
    }
    {   // equal row blocks: 24 rows as 4 x 6 instead of 7 + 7 + 7 + 3
        const int nyt = otp_ceil_div(P.Ho, P.RT);
        lds = size_for(wt, otp_ceil_div(P.Ho, nyt));
    }
    P.magicRT = (unsigned)((0x100000000ull + P.RT - 1) / P.RT);
    P.magicNRX = (unsigned)((0x100000000ull + P.NRX - 1) / P.NRX);
    P.magicWT = (unsigned)((0x100000000ull + P.WT - 1) / P.WT);
    P.magicLWP = (unsigned)((0x100000000ull + P.LWP - 1) / P.LWP);
    // 16-byte staging: stride 1, rows and tiles that are multiples of 4 columns, halo of at most 4 columns either side
    P.vec = (stride == 1 && !P.sparse && W % 4 == 0 && P.Wo % 4 == 0 && P.WT % 4 == 0 && pad <= 4 &&
             (kw - 1) * dil - pad <= 4 && (x_coff * H * W) % 4 == 0)
                ? 1 : 0;
    P.GD = P.WT / 4 > 0 ? P.WT / 4 : 1;
    P.pipe = 0;
    P.GX = (((4 - pad % 4) % 4) + P.LWP + 3) / 4;
    P.magicGD = (unsigned)((0x100000000ull + P.GD - 1) / P.GD);
    P.magicGX = (unsigned)((0x100000000ull + P.GX - 1) / P.GX);
    P.tiles_per_img = otp_ceil_div(P.Ho, P.RT) * P.tiles_x;
    P.NPD = otp_ceil_div(WG_CO * P.RT * P.GD, 256);
    P.NPX = otp_ceil_div(WG_CI * P.NRX * P.GX, 256);
    P.pipe = (P.vec && P.tiles_x == 1 && P.NPD <= 6 && P.NPX <= 12 && P.RT < 256 && P.NRX < 256 && P.GX < 256) ? 1 : 0;
    P.ntiles = N * P.tiles_per_img;
    const int gy = otp_ceil_div(Cout, WG_CO), gz = otp_ceil_div(Cin, WG_CI);
    int gx = wgrad_grid_x(Cout, Cin);
    if (gx > P.ntiles) gx = P.ntiles;
    // workspace given: per-workgroup partial sums + a fixed-order reduction (bit-reproducible); NULL: float atomics straight
    // into dW (result depends on their order in the last bits)
    float* part = static_cast<float*>(workspace);
    if (part && workspace_bytes < (size_t)gx * gy * gz * WG_SLOTS * sizeof(float)) return OTP_ERR_WORKSPACE;
    auto kern = conv_wgrad_kernel;
    OTP_ALLOW_BIG_LDS(kern, lds);
    auto st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(256), lds, st, static_cast<const float*>(x),
                       static_cast<const float*>(grad_out), static_cast<float*>(grad_weight), part, P);
    if (part)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(otp_ceil_div(WG_SLOTS / 4, 32), gy * gz), dim3(1024), 0, st, part,
                           static_cast<float*>(grad_weight), gx, gy, Cout, Cin, kh);
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
