// Dense convolution as an implicit GEMM on the gfx950 f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Covers every dense contraction of the OTPose forward: HRNet 3x3/1x1/stride-2 convs with folded
// BatchNorm (+ residual, + ReLU, + nearest-upsample-accumulate of the fuse layers), the RSB heads
// (bias + BN, pre-added inputs, channel-sliced split/concat), the dilated offset/mask convs, the
// final 1x1 layers and every Conv1d(k=1) of the ConvTransformers (q/k/v/proj with the residual-scale
// epilogue, MLP with exact GELU).  Reference call sites: model/HRNet.py:116-152,478-496,514-571;
// model/RSB.py:77-103; model/OTPose.py:372-383; model/blocks.py:248-254,418-420,450.
//
// GEMM view:  D[co, pixel] = sum_{tap, ci} Wp[tap][ci][co] * X[ci][pixel shifted by tap]
//   M = Cout (16-row MFMA blocks), N = output pixels (16-column blocks, lane = pixel so that stores
//   of one accumulator register are 64-byte runs along W), K = taps x input channels in steps of 4.
//
// Kernel structure (conv_win_kernel, 1x1 and 3x3):
//   * A workgroup owns a CONTIGUOUS range of the flattened output pixels of one image times a slab of
//     output channels.  Per chunk of CK input channels the LDS holds, for every channel, one
//     contiguous WINDOW of the flattened (H*W) input plane - the rows the pixel range needs - with NO
//     column halo: positions outside [0, H*W) are zero (rows above / below the image), and taps that
//     cross the left / right image edge are masked per lane when the B fragment is read.  A window
//     is a straight 16-byte-vector copy of global memory (NCHW planes are contiguous), so staging is a
//     handful of dwordx4 loads per thread instead of one dword per element.
//   * Software pipeline: the global loads of chunk c+1 (input window + weight slab) are issued into
//     registers BEFORE the MFMA loop of chunk c and written to LDS after it, so HBM/L2 latency hides
//     under the matrix work; two workgroups per CU (<= 256 registers, <= 80 KB LDS) cover the rest.
//   * Block ids are remapped so that the M-tiles that share one input window run on the same XCD
//     (blocks b and b+8 share an XCD) right after each other: the window is fetched into that L2 once.
// f32-in/f32-accumulate MFMA is bit-identical to an fmaf chain, so results differ from a CPU conv
// only by summation order.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// floor(i / d) for i * d < 2^32 with magic = floor(2^32 / d) + 1 (0 encodes d == 1)
__device__ __forceinline__ uint32_t fdiv(uint32_t i, uint32_t magic) { return magic ? __umulhi(i, magic) : i; }
uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }

constexpr int MAXJI = 8, MAXJW = 8;   // float4 prefetch registers per thread: input windows / weight slab of a chunk

struct WinPlan {
    // problem
    int N, Cin, H, W, HW, Cout, Cout16, stride, pad, dil, Ho, Wo, HoWo;
    int in_ctot, in_coff, in2_ctot, in2_coff, out_ctot, out_coff, res_ctot, res_coff, res_up, act, frame_split;
    // tiling
    int WP, Mtile, Ptile, tiles_per_img, nP, nM, nthreads;
    int CK, flat, vec;
    int L4;                      // window length per channel in float4
    int G;                       // guard floats in front of every channel window (>= pad, multiple of 4)
    int CS, MS, M4;              // LDS channel stride, weight-row stride, Mtile / 4
    int NI4, NW4;                // float4 items per chunk: input, weights
    uint32_t magicL4, magicM4, magicCK, magicWo;
};

template <int MB, int PB, int KS>
__global__ __launch_bounds__(256, 2) void conv_win_kernel(
    const float* __restrict__ in, const float* __restrict__ in2, const float* __restrict__ wp,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* res, float* out,
    const WinPlan P) {
    constexpr int KK = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* inp = smem;                          // [CK][CS]
    float* wts = smem + P.CK * P.CS;            // [KK*CK][MS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kl = lane >> 4;

    // ---- block id -> (pixel tile, M tile): the nM tiles of one window share an XCD ------------------
    int pt, mt;
    {
        const int per = 8 * P.nM;
        const int g = blockIdx.x / per, r = blockIdx.x - g * per;
        mt = r >> 3;
        pt = g * 8 + (r & 7);
        if (pt >= P.nP) return;                 // uniform per workgroup
    }
    const int n = pt / P.tiles_per_img;
    const int q0 = (pt - n * P.tiles_per_img) * P.Ptile;
    const int wm = wave / P.WP, wpi = wave - wm * P.WP;
    const int m_wg = mt * P.Mtile;
    const int m_wave = wm * 16 * MB;
    const int pix_wave = q0 + wpi * 16 * PB;

    // image n of the (possibly frame-split) input
    size_t in_base, in2_base = 0;
    if (P.frame_split > 0) {
        const int b = n % P.frame_split, f = n / P.frame_split;
        in_base = ((size_t)b * P.in_ctot + P.in_coff + (size_t)f * P.Cin) * P.HW;
    } else {
        in_base = ((size_t)n * P.in_ctot + P.in_coff) * P.HW;
    }
    if (in2) in2_base = ((size_t)n * P.in2_ctot + P.in2_coff) * P.HW;

    // ---- window of the flattened input plane this pixel range needs ------------------------------------
    const int y_first = P.flat ? 0 : (int)fdiv((uint32_t)q0, P.magicWo);
    const int f0 = P.flat ? q0 : (y_first * P.stride - P.pad) * P.W;   // first needed position (may be < 0)
    const int f0a = f0 & ~3;                                          // 16-byte aligned window start
    // LDS offset (floats, inside a channel slot) of tap (0,0) of each of this lane's pixels + column masks
    int poff[PB];
    uint32_t cm0 = 0, cm1 = 0, cm2 = 0;          // bit pb of cm<tj>: tap column tj of pixel pb is inside the image
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        int q = pix_wave + pb * 16 + i16;
        q = q < P.HoWo ? q : q0;                 // padding lanes compute a valid pixel and are never stored
        if (P.flat) {
            poff[pb] = P.G + (q - q0);
            cm0 |= 1u << pb;
        } else {
            const int y = (int)fdiv((uint32_t)q, P.magicWo), x = q - y * P.Wo;
            poff[pb] = P.G + (f0 - f0a) - P.pad + (y - y_first) * P.stride * P.W + x * P.stride;
            const int xi = x * P.stride - P.pad;
            if (xi >= 0 && xi < P.W) cm0 |= 1u << pb;
            if (xi + P.dil >= 0 && xi + P.dil < P.W) cm1 |= 1u << pb;
            if (xi + 2 * P.dil >= 0 && xi + 2 * P.dil < P.W) cm2 |= 1u << pb;
        }
    }

    f32x4 acc[MB][PB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- chunk staging: item i of a chunk is one float4 of the input windows or of the weight slab --------
    // Buffer (SRSRC) loads: one 32-bit byte offset per item, out-of-range offsets (rows outside the image,
    // channels past Cin, columns past Cout16) return 0 from the hardware range check - no branches, no
    // 64-bit addresses.  The descriptors are rebuilt per chunk so that the range check is exact.
    f32x4 pfi[MAXJI], pfw[MAXJW];
    const float* in_img = in + in_base;
    const int wts_off = P.CK * P.CS;
    // item -> (byte offset in the chunk's tensor slice or -1, LDS float offset or -1); straight-line, no branches
    auto item_in = [&](uint32_t i, int& voff, int& dst) {
        const uint32_t c = fdiv(i, P.magicL4), r = i - c * P.L4;
        const int f = f0a + 4 * (int)r;
        const bool live = i < (uint32_t)P.NI4;
        voff = (live && f >= 0 && f < P.HW) ? (int)(c * P.HW + f) * 4 : -1;
        dst = live ? (int)(c * P.CS + 4 * r) + P.G : -1;
    };
    auto item_w = [&](uint32_t w, int& voff, int& dst) {
        const uint32_t row = fdiv(w, P.magicM4), m4 = w - row * P.M4;
        const uint32_t tap = fdiv(row, P.magicCK), c = row - tap * P.CK;
        const int m = m_wg + 4 * (int)m4;
        const bool live = w < (uint32_t)P.NW4;
        voff = (live && m < P.Cout16) ? (int)((tap * P.Cin + c) * P.Cout16 + m) * 4 : -1;
        dst = live ? wts_off + (int)(row * P.MS + 4 * m4) : -1;
    };
    auto load_items = [&](int c0) {
        const otp_rsrc rin = make_rsrc(in_img + (size_t)c0 * P.HW, (size_t)(P.Cin - c0) * P.HW * 4);
        const otp_rsrc rw = make_rsrc(wp + (size_t)c0 * P.Cout16, ((size_t)KK * P.Cin - c0) * P.Cout16 * 4);
        uint32_t t0 = tid;
        asm volatile("" : "+v"(t0));             // keep the (chunk-invariant) item arithmetic out of the MFMA loop's registers
#pragma unroll
        for (int j = 0; j < MAXJI; ++j) {
            int voff, dst;
            item_in(t0 + j * P.nthreads, voff, dst);
            pfi[j] = bload4(rin, voff);
        }
#pragma unroll
        for (int j = 0; j < MAXJW; ++j) {
            int voff, dst;
            item_w(t0 + j * P.nthreads, voff, dst);
            pfw[j] = bload4(rw, voff);
        }
    };
    auto store_items = [&](int c0) {
        uint32_t t1 = tid;
        asm volatile("" : "+v"(t1));
        if (in2) {                               // pre-added second input (RSB staircase): fetched here, unpipelined
            const otp_rsrc rin2 = make_rsrc(in2 + in2_base + (size_t)c0 * P.HW, (size_t)(P.Cin - c0) * P.HW * 4);
#pragma unroll
            for (int j = 0; j < MAXJI; ++j) {
                int voff, dst;
                item_in(t1 + j * P.nthreads, voff, dst);
                pfi[j] += bload4(rin2, voff);
            }
        }
#pragma unroll
        for (int j = 0; j < MAXJI; ++j) {
            int voff, dst;
            item_in(t1 + j * P.nthreads, voff, dst);
            if (dst >= 0) *reinterpret_cast<f32x4*>(smem + dst) = pfi[j];
        }
#pragma unroll
        for (int j = 0; j < MAXJW; ++j) {
            int voff, dst;
            item_w(t1 + j * P.nthreads, voff, dst);
            if (dst >= 0) *reinterpret_cast<f32x4*>(smem + dst) = pfw[j];
        }
    };

    const float* wbase = wts + kl * P.MS + m_wave + i16;
    const float* ibase = inp + kl * P.CS;
    const int dW = P.dil * P.W;

    load_items(0);
    store_items(0);
    __syncthreads();
    for (int c0 = 0; c0 < P.Cin; c0 += P.CK) {
        const bool more = c0 + P.CK < P.Cin;
        if (more) load_items(c0 + P.CK);        // in flight while this chunk is multiplied
        // ---- MFMA over the chunk in LDS ------------------------------------------------------------
#pragma unroll 1
        for (int tap = 0; tap < KK; ++tap) {
            const int ti = tap / KS, tj = tap - ti * KS;
            const uint32_t cm = tj == 0 ? cm0 : (tj == 1 ? cm1 : cm2);
            const float* wrow = wbase + tap * P.CK * P.MS;
            const float* irow = ibase + (KS == 1 ? 0 : ti * dW + tj * P.dil);
            auto step = [&](int kc) {
                float a[MB], b[PB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a[mb] = wrow[kc * P.MS + mb * 16];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    const float v = irow[kc * P.CS + poff[pb]];
                    // all-ones / zero word from bit pb of the column mask: 0 for taps that cross the image edge
                    const int keep = __builtin_amdgcn_sbfe(cm, pb, 1);
                    b[pb] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & keep);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[pb], acc[mb][pb], 0, 0, 0);
            };
            int kc = 0;
            for (; kc + 8 <= P.CK; kc += 8) {
                step(kc);
                step(kc + 4);
            }
            if (kc < P.CK) step(kc);
        }
        if (more) {
            __syncthreads();                    // every wave finished reading this chunk
            store_items(c0 + P.CK);
            __syncthreads();
        }
    }

    // ---- epilogue: scale/shift (+res) (+act), optional nearest-upsample accumulate -------------------
    const int f = P.res_up > 1 ? P.res_up : 1;
    const int HWo_hi = P.HoWo * f * f, Wo_hi = P.Wo * f;
    int qhi[PB];                                 // index of the (dy=0, dx=0) target pixel on the output grid
    bool qok[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int q = pix_wave + pb * 16 + i16;
        qok[pb] = q < P.HoWo && q < q0 + P.Ptile;
        if (f == 1) {
            qhi[pb] = q;
        } else {
            const int y = q / P.Wo, x = q - y * P.Wo;
            qhi[pb] = y * f * Wo_hi + x * f;
        }
    }
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
            const int sub = dy * Wo_hi + dx;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = m_wg + m_wave + mb * 16 + kl * 4 + r;
                    const bool co_ok = co < P.Cout && m_wave + mb * 16 < P.Mtile;
                    const float sc = (co_ok && scale) ? scale[co] : 1.f;
                    const float sh = (co_ok && shift) ? shift[co] : 0.f;
                    const size_t obase = ((size_t)n * P.out_ctot + P.out_coff + co) * HWo_hi + sub;
                    const size_t rbase = ((size_t)n * P.res_ctot + P.res_coff + co) * HWo_hi + sub;
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        if (co_ok && qok[pb]) {
                            float v = fmaf(acc[mb][pb][r], sc, sh);
                            if (res) v += res[rbase + qhi[pb]];
                            if (P.act == OTP_ACT_RELU) v = fmaxf(v, 0.f);
                            else if (P.act == OTP_ACT_GELU) v = gelu_erf(v);
                            out[obase + qhi[pb]] = v;
                        }
                    }
                }
            }
        }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                   int KK, int Cout16) {
    const int total = KK * Cin * Cout16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int co = i % Cout16, r = i / Cout16;
        int ci = r % Cin, tap = r / Cin;
        wp[i] = co < Cout ? w[((size_t)co * Cin + ci) * KK + tap] : 0.f;
    }
}

// smallest r >= n with r % 32 == want_mod32
int pad_stride(int n, int want_mod32) {
    int r = ((n + 31) / 32) * 32 + want_mod32;
    while (r - 32 >= n) r -= 32;
    return r;
}

int g_force[4] = {0, 0, 0, 0};   // test / tuning hook: forced (MB, PB, WM, WP)


// ------------------------------------------------------------------------------------------------
// generic fallback (any kernel size, element-wise staging): used only when the window kernel does not
// apply (kernel sides other than 1 / 3, or no window plan fits the LDS / prefetch budget)
// ------------------------------------------------------------------------------------------------
struct ConvPlan {
    otp_conv_desc d;
    int Cout16, KK;
    int WM, WP, Mtile, Ptile, tiles_per_img, HoWo;
    int CK;            // input channels per LDS chunk (multiple of 4)
    int flat;          // 1x1 / stride 1 / no padding: patch is the pixel range itself
    int NRmax, LW;     // staged rows per channel and row pitch (floats)
    int CS, MS;        // LDS channel stride / weight-row stride (floats), chosen against bank conflicts
    int nthreads;
};

template <int MB, int PB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const float* __restrict__ in, const float* __restrict__ in2, const float* __restrict__ wp,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* res,
    float* out, ConvPlan P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const otp_conv_desc& d = P.d;
    float* inp = smem;                          // [CK][CS]
    float* wts = smem + P.CK * P.CS;            // [KK*CK][MS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = P.nthreads >> 6;
    const int n = blockIdx.x / P.tiles_per_img;
    const int q0 = (blockIdx.x - n * P.tiles_per_img) * P.Ptile;
    const int wm = wave / P.WP, wpi = wave - wm * P.WP;
    const int m_wg = blockIdx.y * P.Mtile;
    const int m_wave = wm * 16 * MB;            // within the workgroup slab
    const int pix_wave = q0 + wpi * 16 * PB;
    const int HW = d.H * d.W;

    // image n of the (possibly frame-split) input
    size_t in_base, in2_base = 0;
    if (d.frame_split > 0) {
        int b = n % d.frame_split, f = n / d.frame_split;
        in_base = ((size_t)b * d.in_ctot + d.in_coff + (size_t)f * d.Cin) * HW;
    } else {
        in_base = ((size_t)n * d.in_ctot + d.in_coff) * HW;
    }
    if (in2) in2_base = ((size_t)n * d.in2_ctot + d.in2_coff) * HW;

    // rows of the input this pixel range needs (2-D mode)
    const int y_first = q0 / d.Wo;
    const int q_last = min(q0 + P.Ptile, P.HoWo) - 1;
    const int rows_out = q_last / d.Wo - y_first + 1;
    const int NR = (rows_out - 1) * d.stride + (d.kh - 1) * d.dil + 1;
    const int r_in0 = y_first * d.stride - d.pad;
    const int LWused = d.W + 2 * d.pad;

    // per-lane LDS offsets of this wave's pixels
    int poff[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        int q = pix_wave + pb * 16 + (lane & 15);
        q = q < P.HoWo ? q : q0;                // padding lanes compute a valid pixel and are never stored
        if (P.flat) {
            poff[pb] = q - q0;
        } else {
            int y = q / d.Wo, x = q - y * d.Wo;
            poff[pb] = (y - y_first) * d.stride * P.LW + x * d.stride;
        }
    }
    const int kl = lane >> 4;                   // k index of this lane inside a K-step of 4

    f32x4 acc[MB][PB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < d.Cin; c0 += P.CK) {
        __syncthreads();
        // ---- stage the input patch ------------------------------------------------------------
        if (P.flat) {
            const int total = P.CK * P.Ptile;
            for (int idx = tid; idx < total; idx += P.nthreads) {
                int c = idx / P.Ptile, j = idx - c * P.Ptile;
                int q = q0 + j;
                float v = 0.f;
                if (c0 + c < d.Cin && q < HW) {
                    size_t o = (size_t)(c0 + c) * HW + q;
                    v = in[in_base + o];
                    if (in2) v += in2[in2_base + o];
                }
                inp[c * P.CS + j] = v;
            }
        } else {
            const int nrows = P.CK * NR;
            for (int rid = wave; rid < nrows; rid += nwaves) {
                int c = rid / NR, r = rid - c * NR;
                int y = r_in0 + r;
                bool row_ok = (c0 + c < d.Cin) && y >= 0 && y < d.H;
                size_t o = (size_t)(c0 + c) * HW + (size_t)(row_ok ? y : 0) * d.W;
                float* dst = inp + c * P.CS + r * P.LW;
                for (int col = lane; col < LWused; col += 64) {
                    int x = col - d.pad;
                    float v = 0.f;
                    if (row_ok && x >= 0 && x < d.W) {
                        v = in[in_base + o + x];
                        if (in2) v += in2[in2_base + o + x];
                    }
                    dst[col] = v;
                }
            }
        }
        // ---- stage the weight slab: wts[(tap*CK + c)][m] = Wp[tap][c0+c][m_wg+m] -----------------
        {
            const int rows = P.KK * P.CK;
            const int total = rows * P.Mtile;
            for (int idx = tid; idx < total; idx += P.nthreads) {
                int row = idx / P.Mtile, m = idx - row * P.Mtile;
                int tap = row / P.CK, c = row - tap * P.CK;
                float v = 0.f;
                if (c0 + c < d.Cin && m_wg + m < P.Cout16)
                    v = wp[((size_t)tap * d.Cin + c0 + c) * P.Cout16 + m_wg + m];
                wts[row * P.MS + m] = v;
            }
        }
        __syncthreads();
        // ---- MFMA over this chunk ------------------------------------------------------------------
        for (int tap = 0; tap < P.KK; ++tap) {
            int ti = tap / d.kw, tj = tap - ti * d.kw;
            const int tap_off = P.flat ? 0 : (ti * d.dil) * P.LW + tj * d.dil;
            const float* wrow = wts + (tap * P.CK + kl) * P.MS + m_wave + (lane & 15);
            const float* irow = inp + kl * P.CS + tap_off;
            for (int kc = 0; kc < P.CK; kc += 4) {
                float a[MB], b[PB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a[mb] = wrow[kc * P.MS + mb * 16];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) b[pb] = irow[kc * P.CS + poff[pb]];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[pb], acc[mb][pb], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: scale/shift (+res) (+act), optional nearest-upsample accumulate -------------------
    const int f = d.res_up > 1 ? d.res_up : 1;
    const int HWo_hi = P.HoWo * f * f, Wo_hi = d.Wo * f;
    int qhi[PB];                                 // index of the (dy=0, dx=0) target pixel on the output grid
    bool qok[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int q = pix_wave + pb * 16 + (lane & 15);
        qok[pb] = q < P.HoWo;
        if (f == 1) {
            qhi[pb] = q;
        } else {
            const int y = q / d.Wo, x = q - y * d.Wo;
            qhi[pb] = y * f * Wo_hi + x * f;
        }
    }
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
            const int sub = dy * Wo_hi + dx;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = m_wg + m_wave + mb * 16 + kl * 4 + r;
                    const bool co_ok = co < d.Cout;
                    const float sc = (co_ok && scale) ? scale[co] : 1.f;
                    const float sh = (co_ok && shift) ? shift[co] : 0.f;
                    const size_t obase = ((size_t)n * d.out_ctot + d.out_coff + co) * HWo_hi + sub;
                    const size_t rbase = ((size_t)n * d.res_ctot + d.res_coff + co) * HWo_hi + sub;
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        if (co_ok && qok[pb]) {
                            float v = fmaf(acc[mb][pb][r], sc, sh);
                            if (res) v += res[rbase + qhi[pb]];
                            if (d.act == OTP_ACT_RELU) v = fmaxf(v, 0.f);
                            else if (d.act == OTP_ACT_GELU) v = gelu_erf(v);
                            out[obase + qhi[pb]] = v;
                        }
                    }
                }
            }
        }
}


struct Tile { int MB, PB, WM, WP; };

// Fill every tiling-dependent field of P for a candidate tile; false when it does not fit (LDS / prefetch registers).
bool fill_plan(WinPlan& P, const Tile& t, int KS, size_t& lds_bytes) {
    const int KK = KS * KS;
    P.WP = t.WP;
    P.Mtile = 16 * t.MB * t.WM;
    P.Ptile = 16 * t.PB * t.WP;
    P.nthreads = 64 * t.WM * t.WP;
    P.tiles_per_img = (P.HoWo + P.Ptile - 1) / P.Ptile;
    P.nP = P.N * P.tiles_per_img;
    P.nM = (P.Cout16 + P.Mtile - 1) / P.Mtile;
    P.M4 = P.Mtile / 4;
    P.MS = pad_stride(P.Mtile, 16);
    P.G = ((P.pad + 3) & ~3) < 16 ? 16 : ((P.pad + 3) & ~3);
    int L;
    if (P.flat) {
        L = P.Ptile;
    } else {
        int rows = (P.Ptile % P.Wo == 0) ? P.Ptile / P.Wo : (P.Ptile + P.Wo - 2) / P.Wo + 1;
        if (rows > P.Ho) rows = P.Ho;
        const int NR = (rows - 1) * P.stride + (KS - 1) * P.dil + 1;
        L = NR * P.W + 3;                                  // + alignment slack of the window start
    }
    P.L4 = (L + 3) / 4;
    P.CS = pad_stride(P.G + 4 * P.L4 + P.G, 16);           // == 16 (mod 32): the 4 k-rows of a fragment hit distinct banks
    const int cin4 = (P.Cin + 3) & ~3;
    // chunk size: a small Cin is one chunk; otherwise 16 / 8 / 4 channels, preferring the one that wastes the
    // fewest zero-padded channels in the last chunk (Cin = 136 -> 17 x 8), then the larger
    const int cands[4] = {cin4 <= 32 ? cin4 : 16, 16, 8, 4};
    int best_c = 0;
    long best_waste = 0;
    size_t best_lds = 0;
    for (int ci = 0; ci < 4; ++ci) {
        const int c = cands[ci] > cin4 ? cin4 : cands[ci];
        const size_t lds = ((size_t)c * P.CS + (size_t)KK * c * P.MS) * sizeof(float);
        if (lds > 80 * 1024 || (long)c * P.L4 > (long)MAXJI * P.nthreads || (long)KK * c * P.M4 > (long)MAXJW * P.nthreads) continue;
        const long waste = (long)((cin4 + c - 1) / c) * c - cin4;
        if (!best_c || waste < best_waste) { best_c = c; best_waste = waste; best_lds = lds; }
    }
    if (!best_c) return false;
    P.CK = best_c;
    P.NI4 = best_c * P.L4;
    P.NW4 = KK * best_c * P.M4;
    lds_bytes = best_lds;
    P.magicL4 = magic_of(P.L4);
    P.magicM4 = magic_of(P.M4);
    P.magicCK = magic_of(P.CK);
    P.magicWo = magic_of(P.Wo);
    return true;
}

// Relative time of a candidate: MFMA tile-steps on the busiest SIMD, inflated by what the tile cannot hide.
double tile_cost(const WinPlan& P, const Tile& t, int KS, size_t lds) {
    const int wpw = t.WM * t.WP;                                   // waves per workgroup
    const long nwg = (long)P.nP * P.nM;
    const long wg_per_cu = (nwg + 255) / 256;                      // busiest CU
    const long waves_per_simd = (wg_per_cu * wpw + 3) / 4;
    double cost = (double)waves_per_simd * t.MB * t.PB;
    // concurrency available to hide LDS latency and the chunk hand-over: resident waves per SIMD
    int resident = (int)((160 * 1024) / (lds ? lds : 1));
    if (resident > 8 / wpw) resident = 8 / wpw;                    // <= 2 waves per SIMD (256 registers)
    if (resident < 1) resident = 1;
    long conc = resident < wg_per_cu ? resident : wg_per_cu;
    const double simd_waves = (double)conc * wpw / 4.0;
    const double chunk_cycles = 32.0 * KS * KS * (P.CK / 4.0) * t.MB * t.PB;
    const double handover = 900.0 / chunk_cycles;                  // barriers + LDS write of the next chunk
    cost *= 1.0 + handover / (simd_waves >= 2.0 ? 3.0 : 1.0);
    cost *= 1.0 + 0.10 / (simd_waves >= 2.0 ? 2.0 : 1.0) * (12.0 / (t.MB * t.PB));   // exposed LDS latency per step
    cost *= 1.0 + 0.03 / t.MB;                                     // A-fragment reuse
    if (wpw == 3) cost *= 1.15;                                    // one SIMD idles
    return cost;
}

template <int MB, int PB, int KS>
int launch_win(const float* in, const float* in2, const float* wp, const float* scale, const float* shift,
               const float* res, float* out, const WinPlan& P, size_t lds, hipStream_t st) {
    auto kern = conv_win_kernel<MB, PB, KS>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    const int groups = (P.nP + 7) / 8;
    dim3 grid(groups * 8 * P.nM);
    hipLaunchKernelGGL(kern, grid, dim3(P.nthreads), lds, st, in, in2, wp, scale, shift, res, out, P);
    return otp_launch_status();
}

template <int KS>
int dispatch_win(int MB, int PB, const float* in, const float* in2, const float* wp, const float* scale,
                 const float* shift, const float* res, float* out, const WinPlan& P, size_t lds, hipStream_t st) {
#define OTP_CASE(M_, P_) if (MB == M_ && PB == P_) return launch_win<M_, P_, KS>(in, in2, wp, scale, shift, res, out, P, lds, st);
    OTP_CASE(1, 7) OTP_CASE(1, 8) OTP_CASE(1, 9)
    OTP_CASE(2, 7) OTP_CASE(2, 8) OTP_CASE(2, 9)
    OTP_CASE(3, 7) OTP_CASE(3, 8) OTP_CASE(3, 9)
    OTP_CASE(4, 7)
#undef OTP_CASE
    return OTP_ERR_UNSUPPORTED;
}

// ---- generic fallback plan (element-wise staging kernel) ---------------------------------------------
bool choose_generic(ConvPlan& P) {
    const otp_conv_desc& d = P.d;
    P.KK = d.kh * d.kw;
    P.Cout16 = (d.Cout + 15) & ~15;
    P.HoWo = d.Ho * d.Wo;
    P.flat = (d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad == 0) ? 1 : 0;
    const int mblk = P.Cout16 / 16;
    const int MB = mblk >= 3 ? 3 : mblk;
    P.WM = 1; P.WP = 4;
    P.Mtile = 16 * MB;
    P.Ptile = 16 * 7 * P.WP;
    P.tiles_per_img = (P.HoWo + P.Ptile - 1) / P.Ptile;
    P.nthreads = 256;
    P.MS = pad_stride(P.Mtile, 16);
    if (P.flat) {
        P.NRmax = 1; P.LW = P.Ptile;
        P.CS = pad_stride(P.Ptile, 16);
    } else {
        int rows_out = (P.Ptile + d.Wo - 1) / d.Wo + 1;
        if (rows_out > d.Ho) rows_out = d.Ho;
        P.NRmax = (rows_out - 1) * d.stride + (d.kh - 1) * d.dil + 1;
        P.LW = d.W + 2 * d.pad;
        P.CS = pad_stride(P.NRmax * P.LW, d.stride == 1 ? 16 : 17);
    }
    const size_t hard = OTP_LDS_LIMIT;
    int ck = 16;
    auto lds_of = [&](int c) { return ((size_t)c * P.CS + (size_t)P.KK * c * P.MS) * sizeof(float); };
    while (ck > 4 && lds_of(ck) > 72 * 1024) ck >>= 1;
    if (lds_of(ck) > hard) return false;
    int cin4 = (d.Cin + 3) & ~3;
    if (ck > cin4) ck = cin4;
    P.CK = ck;
    return true;
}

template <int MB>
int launch_generic(const float* in, const float* in2, const float* wp, const float* scale, const float* shift,
                   const float* res, float* out, const ConvPlan& P, hipStream_t st) {
    size_t lds = ((size_t)P.CK * P.CS + (size_t)P.KK * P.CK * P.MS) * sizeof(float);
    auto kern = conv_igemm_kernel<MB, 7>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    dim3 grid(P.d.N * P.tiles_per_img, (P.Cout16 + P.Mtile - 1) / P.Mtile);
    hipLaunchKernelGGL(kern, grid, dim3(P.nthreads), lds, st, in, in2, wp, scale, shift, res, out, P);
    return otp_launch_status();
}

}  // namespace

extern "C" int otp_conv2d_set_tile(int MB, int PB, int WM, int WP) {
    g_force[0] = MB; g_force[1] = PB; g_force[2] = WM; g_force[3] = WP;
    return OTP_OK;
}

extern "C" int otp_conv2d_pack_weight(const void* weight, void* wpacked, int Cout, int Cin, int kh, int kw,
                                      void* stream) {
    if (!weight || !wpacked || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0) return OTP_ERR_BAD_ARG;
    int Cout16 = (Cout + 15) & ~15, total = kh * kw * Cin * Cout16;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(otp_ceil_div(total, 256) > 1024 ? 1024 : otp_ceil_div(total, 256)),
                       dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float*>(weight),
                       static_cast<float*>(wpacked), Cout, Cin, kh * kw, Cout16);
    return otp_launch_status();
}

extern "C" int otp_conv2d(const void* in, const void* in2, const void* wpacked, const void* scale,
                          const void* shift, const void* res, void* out, const otp_conv_desc* desc, void* stream) {
    if (!in || !wpacked || !out || !desc) return OTP_ERR_BAD_ARG;
    const otp_conv_desc& d = *desc;
    if (d.N <= 0 || d.Cin <= 0 || d.H <= 0 || d.W <= 0 || d.Cout <= 0 || d.kh <= 0 || d.kw <= 0 || d.stride <= 0 ||
        d.pad < 0 || d.dil <= 0)
        return OTP_ERR_BAD_ARG;
    int Ho = (d.H + 2 * d.pad - (d.dil * (d.kh - 1) + 1)) / d.stride + 1;
    int Wo = (d.W + 2 * d.pad - (d.dil * (d.kw - 1) + 1)) / d.stride + 1;
    if (Ho != d.Ho || Wo != d.Wo || Ho <= 0 || Wo <= 0) return OTP_ERR_BAD_ARG;
    if (d.res_up > 1 && d.act == OTP_ACT_GELU) return OTP_ERR_UNSUPPORTED;
    auto st = static_cast<hipStream_t>(stream);
    auto a = static_cast<const float*>(in);
    auto b = static_cast<const float*>(in2);
    auto w = static_cast<const float*>(wpacked);
    auto sc = static_cast<const float*>(scale);
    auto sh = static_cast<const float*>(shift);
    auto r = static_cast<const float*>(res);
    auto o = static_cast<float*>(out);

    // the window kernel stages 16-byte vectors: channel planes must be 16-byte aligned (H*W % 4 == 0, aligned bases)
    const bool vec_ok = ((d.H * d.W) & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
                        (!in2 || (reinterpret_cast<uintptr_t>(in2) & 15) == 0) &&
                        (reinterpret_cast<uintptr_t>(wpacked) & 15) == 0;
    const bool win_ok = vec_ok && d.kh == d.kw && (d.kh == 1 || d.kh == 3) && (long)d.H * d.W < (1l << 24);
    if (win_ok) {
        WinPlan P{};
        P.N = d.N; P.Cin = d.Cin; P.H = d.H; P.W = d.W; P.HW = d.H * d.W; P.Cout = d.Cout;
        P.Cout16 = (d.Cout + 15) & ~15; P.stride = d.stride; P.pad = d.pad; P.dil = d.dil;
        P.Ho = d.Ho; P.Wo = d.Wo; P.HoWo = d.Ho * d.Wo;
        P.in_ctot = d.in_ctot; P.in_coff = d.in_coff; P.in2_ctot = d.in2_ctot; P.in2_coff = d.in2_coff;
        P.out_ctot = d.out_ctot; P.out_coff = d.out_coff; P.res_ctot = d.res_ctot; P.res_coff = d.res_coff;
        P.res_up = d.res_up; P.act = d.act; P.frame_split = d.frame_split;
        P.flat = (d.kh == 1 && d.stride == 1 && d.pad == 0) ? 1 : 0;
        P.vec = 1;
        const int mblk = P.Cout16 / 16, G = (P.HoWo + 15) / 16;
        Tile best{0, 0, 0, 0};
        double best_cost = 1e300;
        static const int wms[] = {1, 2, 4}, pbs[] = {7, 8, 9};
        for (int MB = 1; MB <= 4; ++MB)
            for (int PB : pbs)
                for (int WM : wms)
                    for (int WP = 1; WM * WP <= 4; ++WP) {
                        if (MB * PB > 28) continue;
                        const Tile t{MB, PB, WM, WP};
                        if (g_force[0]) {
                            if (MB != g_force[0] || PB != g_force[1] || WM != g_force[2] || WP != g_force[3]) continue;
                        } else {
                            if (WM > 1 && MB * (WM - 1) >= mblk) continue;        // whole waves of padding
                            if (WP > 1 && PB * (WP - 1) >= G) continue;
                        }
                        WinPlan C = P;
                        size_t lds = 0;
                        if (!fill_plan(C, t, d.kh, lds)) continue;
                        const double cost = tile_cost(C, t, d.kh, lds);
                        if (cost < best_cost) { best_cost = cost; best = t; }
                    }
        if (best.MB) {
            size_t lds = 0;
            fill_plan(P, best, d.kh, lds);
            if (d.kh == 1) return dispatch_win<1>(best.MB, best.PB, a, b, w, sc, sh, r, o, P, lds, st);
            return dispatch_win<3>(best.MB, best.PB, a, b, w, sc, sh, r, o, P, lds, st);
        }
    }
    ConvPlan P;
    P.d = d;
    if (!choose_generic(P)) return OTP_ERR_UNSUPPORTED;
    const int MB = P.Mtile / 16;
    if (MB == 1) return launch_generic<1>(a, b, w, sc, sh, r, o, P, st);
    if (MB == 2) return launch_generic<2>(a, b, w, sc, sh, r, o, P, st);
    return launch_generic<3>(a, b, w, sc, sh, r, o, P, st);
}
