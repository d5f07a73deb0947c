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

constexpr int MAXJI_LIN = 10, MAXJI_GEN = 10, MAXJW = 5;   // float4 prefetch registers per thread: input windows / weight slab of a chunk

struct WinPlan {
    // problem
    int N, Cin, H, W, HW, Cout, Cout16, stride, pad, dil, Ho, Wo, HoWo;
    int in_ctot, in_coff, in2_ctot, in2_coff, out_ctot, out_coff, res_ctot, res_coff, res_up, act, frame_split;
    // tiling
    int WP, Mtile, Ptile, tiles_per_img, nP, nM, nthreads, ntiles;
    int CK, flat, lin;
    int L4;                      // window length per channel in float4
    int G;                       // guard floats in front of every channel window (>= pad, multiple of 4)
    int CS, MS, M4;              // LDS channel stride, weight-row stride, Mtile / 4
    int CKW, JR, NJI;            // input staging: channels per wave, 64-float4 pieces per window, items per thread
    int NW4, NJW;                // weight staging: float4 per chunk, items per thread
    int prio;                    // tuning: static wave priority by dispatch round (see kernel)
    int ep_vec;                  // epilogue may use 16-byte vectors (no upsample, Ho*Wo % 4 == 0, aligned out / res)
    uint32_t magicM4, magicCK, magicWo, magicTpi;
};

// One workgroup = (a contiguous range of Ptile output pixels of one image) x (Mtile output channels); workgroups are
// persistent and walk the tile list with stride gridDim.x.  The kernel is one flat pipeline over (tile, chunk)
// steps: while step s is multiplied out of LDS, the global loads of step s+1 - the next chunk of this tile or the
// first chunk of the next tile - are in flight into registers.
//
// On gfx950 the f32 MFMA shares the SIMD's vector issue with ordinary VALU work (measured: every VALU
// instruction of either resident wave adds its ~4 cycles to the MFMA time), so the multiply loop is written to
// need almost none.  LIN = the LDS offset of a pixel is linear in its flattened index (1x1 "flat" mode and every
// stride-1 same-size conv): the 16-pixel blocks of a lane are then a compile-time 64 bytes apart, so one address
// register + immediate offsets serve all B fragments, the edge-column masks are precomputed all-ones / zero words
// applied with one v_and, and the centre tap column needs no mask at all.
template <int MB, int PB, int KS, bool LIN>
__global__ __launch_bounds__(256, 2) void conv_win_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* res, float* out, const WinPlan P) {
    constexpr int KK = KS * KS;
    constexpr int MAXJI = LIN ? MAXJI_LIN : MAXJI_GEN;   // strided windows are ~4x larger per output pixel
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* inp = smem;                          // [CK][CS]
    float* wts = smem + P.CK * P.CS;            // [KK*CK][MS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kl = lane >> 4;
    const int wm = wave / P.WP, wpi = wave - wm * P.WP;
    const int m_wave = wm * 16 * MB;

    // ---- tile id -> (image, first pixel, M tile): the nM tiles of one window are 8 ids apart = same XCD ---------
    struct TileId { int n, q0, m_wg; };
    auto tile_ok = [&](int t) {
        const int per = 8 * P.nM;
        const int g = t / per, r = t - g * per;
        return t < P.ntiles && g * 8 + (r & 7) < P.nP;
    };
    auto decode = [&](int t) {
        TileId T;
        const int per = 8 * P.nM;
        const int g = t / per, r = t - g * per;
        const int pt = g * 8 + (r & 7);
        T.n = (int)fdiv((uint32_t)pt, P.magicTpi);
        T.q0 = (pt - T.n * P.tiles_per_img) * P.Ptile;
        T.m_wg = (r >> 3) * P.Mtile;
        return T;
    };

    // ---- chunk staging ---------------------------------------------------------------------------------------
    // Input: wave w owns channels [w*CKW, (w+1)*CKW) of the chunk; item (jc, jr) is float4 number lane + 64*jr of that
    // channel's window.  One buffer descriptor per channel (base = the channel plane, size = H*W*4 or 0 past Cin):
    // the hardware range check zero-fills rows above / below the image and missing channels, so a load costs one
    // v_add.  Weights: item w = tid + j*nthreads of the [KK*CK][M4] slab of this M tile; byte offset inside the
    // (chunk, M tile) slice of the packed tensor and LDS offset are computed once per kernel.
    f32x4 pfi[MAXJI], pfw[MAXJW];
    int wvoff[MAXJW], wdst[MAXJW];
#pragma unroll
    for (int j = 0; j < MAXJW; ++j) {
        const uint32_t w = tid + j * P.nthreads;
        const uint32_t row = fdiv(w, P.magicM4), m4 = w - row * P.M4;
        const uint32_t tap = fdiv(row, P.magicCK), c = row - tap * P.CK;
        const bool live = j < P.NJW && w < (uint32_t)P.NW4;
        wvoff[j] = live ? (int)((tap * P.Cin + c) * P.Cout16 + 4 * m4) * 4 : -1;
        wdst[j] = live ? (int)(P.CK * P.CS + row * P.MS + 4 * m4) : -1;
    }
    auto load_items = [&](int t, int c0) {
        const TileId T = decode(t);
        const float* img;
        if (P.frame_split > 0) {
            const int b = T.n % P.frame_split, f = T.n / P.frame_split;
            img = in + ((size_t)b * P.in_ctot + P.in_coff + (size_t)f * P.Cin) * P.HW;
        } else {
            img = in + ((size_t)T.n * P.in_ctot + P.in_coff) * P.HW;
        }
        // first needed position of the flattened plane, rounded down to 16 bytes (may be negative)
        const int y_first = P.flat ? 0 : (int)fdiv((uint32_t)T.q0, P.magicWo);
        const int f0a = (P.flat ? T.q0 : (y_first * P.stride - P.pad) * P.W) & ~3;
        const int vbase = (f0a + 4 * lane) * 4;
        int jc = 0, jr = 0;
#pragma unroll
        for (int j = 0; j < MAXJI; ++j) {
            if (j < P.NJI) {
                const int c = wave * P.CKW + jc, ch = c0 + c;
                const bool live = c < P.CK && ch < P.Cin;
                const otp_rsrc r = make_rsrc32(img + (size_t)(live ? ch : 0) * P.HW, live ? (unsigned)P.HW * 4u : 0u);
                pfi[j] = bload4(r, vbase + jr * 1024);
                if (++jr == P.JR) { jr = 0; ++jc; }
            }
        }
        // weight slice of (chunk, M tile): rows past Cin alias later rows (finite, multiplied by zero input) or fall
        // off the end of the tensor (range check -> 0)
        const otp_rsrc rw = make_rsrc32(wp + (size_t)c0 * P.Cout16 + T.m_wg,
                                        ((unsigned)(KK * P.Cin - c0) * (unsigned)P.Cout16 - (unsigned)T.m_wg) * 4u);
#pragma unroll
        for (int j = 0; j < MAXJW; ++j)
            if (j < P.NJW) pfw[j] = bload4(rw, wvoff[j]);
        asm volatile("" ::: "memory");           // the loads are issued HERE (not sunk towards their use after the MFMAs)
    };
    auto store_items = [&]() {
        int jc = 0, jr = 0;
#pragma unroll
        for (int j = 0; j < MAXJI; ++j) {
            if (j < P.NJI) {
                const int c = wave * P.CKW + jc, r4 = lane + 64 * jr;
                if (c < P.CK && r4 < P.L4) *reinterpret_cast<f32x4*>(inp + c * P.CS + P.G + 4 * r4) = pfi[j];
                if (++jr == P.JR) { jr = 0; ++jc; }
            }
        }
#pragma unroll
        for (int j = 0; j < MAXJW; ++j)
            if (j < P.NJW && wdst[j] >= 0) *reinterpret_cast<f32x4*>(smem + wdst[j]) = pfw[j];
    };

    // ---- lane geometry of the current tile ------------------------------------------------------------------------
    // LIN: pl0 = LDS offset of tap (0,0) of pixel block 0 (block pb adds 16 floats); keepL / keepR = all-ones words
    // unless the left / right tap column of the block's pixel falls off the image.  General: one offset per block
    // and three bit masks (tap columns 0, 1, 2).
    int pl0 = 0;
    int keepL[PB], keepR[PB];
    int poff[PB];
    uint32_t cm0 = 0, cm1 = 0, cm2 = 0;

    f32x4 acc[MB][PB];

    // ---- epilogue: scale/shift (+res) (+act), optional nearest-upsample accumulate --------------------------
    // The accumulators of one 16-row block go through a per-wave LDS tile [16][RS] (lane = pixel column, so a
    // register is a 64-byte run; LDS turns it into rows), then leave as 16-byte vectors along the pixel axis in a
    // short runtime loop: few instructions, few live registers, full-line stores.
    constexpr int RS = 16 * PB + 4;              // row pitch: == 4 (mod 8) floats -> conflict-free transposing writes
    auto epilogue = [&](const TileId& T) {
        float* ep = smem + wave * (16 * RS);
        const int pix_wave = T.q0 + wpi * 16 * PB;
        const int f = P.res_up > 1 ? P.res_up : 1;
        // scale / shift of this wave's 16*MB output channels: lane l holds channel l (one global round trip per tile;
        // the store loops fetch them with a lane shuffle)
        constexpr int NSC = (16 * MB + 63) / 64;
        float sc_l[NSC], sh_l[NSC];
#pragma unroll
        for (int i = 0; i < NSC; ++i) {
            const int co = T.m_wg + m_wave + lane + 64 * i;
            sc_l[i] = 1.f; sh_l[i] = 0.f;
            if (lane + 64 * i < 16 * MB && co < P.Cout) {
                if (scale) sc_l[i] = scale[co];
                if (shift) sh_l[i] = shift[co];
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(kl * 4 + r) * RS + pb * 16 + i16] = acc[mb][pb][r];
            const int co_base = T.m_wg + m_wave + mb * 16;
            if (P.ep_vec) {
                // lane = (row = lane / 4, float4 column = lane % 4 + 4 * it): one output channel per lane, so its
                // scale / shift are fetched once and the LDS / global offsets advance by constants (64 B per step).
                // Buffer descriptors over this image's out / res slices: channels past Cout fall off the end and are
                // dropped by the range check; pixels past Ho*Wo are masked to an out-of-range offset.
                const int row = lane >> 2, c40 = lane & 3;
                const float sc = __shfl(sc_l[mb >> 2], (mb & 3) * 16 + row, 64), sh = __shfl(sh_l[mb >> 2], (mb & 3) * 16 + row, 64);
                const int co = co_base + row;
                const unsigned obytes = co < P.Cout ? (unsigned)P.HoWo * 4u : 0u;
                const float* ep_row = ep + row * RS + 4 * c40;
                const int q_lane = pix_wave + 4 * c40;
                float* orow = out + ((size_t)T.n * P.out_ctot + P.out_coff + co) * P.HoWo;
                const float* rrow = res ? res + ((size_t)T.n * P.res_ctot + P.res_coff + co) * P.HoWo : nullptr;
                constexpr int EB = 2;                  // loads of a batch are issued before its first store
#pragma unroll
                for (int it0 = 0; it0 < PB; it0 += EB) {
                    f32x4 v[EB], rv[EB];
#pragma unroll
                    for (int e = 0; e < EB; ++e) {
                        if (it0 + e < PB) {
                            const int q = q_lane + 16 * (it0 + e);
                            v[e] = *reinterpret_cast<const f32x4*>(ep_row + 16 * (it0 + e));
                            if (res && obytes && q < P.HoWo) rv[e] = *reinterpret_cast<const f32x4*>(rrow + q);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < EB; ++e) {
                        if (it0 + e < PB) {
                            const int q = q_lane + 16 * (it0 + e);
                            const bool ok = obytes && q < P.HoWo;
                            f32x4 o = v[e] * sc + sh;
                            if (res && ok) o += rv[e];
                            if (P.act == OTP_ACT_RELU) {
                                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                            } else if (P.act == OTP_ACT_GELU) {
                                o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w);
                            }
                            if (ok) *reinterpret_cast<f32x4*>(orow + q) = o;
                        }
                    }
                }
            } else {
                const int HWo_hi = P.HoWo * f * f, Wo_hi = P.Wo * f;
#pragma unroll 1
                for (int i = lane; i < 16 * 16 * PB; i += 64) {
                    const int row = i / (16 * PB), col = i - row * (16 * PB);
                    const int co = co_base + row, q = pix_wave + col;
                    const float sc = __shfl(sc_l[mb >> 2], (mb & 3) * 16 + row, 64), sh = __shfl(sh_l[mb >> 2], (mb & 3) * 16 + row, 64);
                    if (co < P.Cout && q < P.HoWo) {
                        const float v0 = fmaf(ep[row * RS + col], sc, sh);
                        int qhi = q;
                        if (f > 1) {
                            const int y = q / P.Wo, x = q - y * P.Wo;
                            qhi = y * f * Wo_hi + x * f;
                        }
                        const size_t ob = ((size_t)T.n * P.out_ctot + P.out_coff + co) * HWo_hi + qhi;
                        const size_t rb = ((size_t)T.n * P.res_ctot + P.res_coff + co) * HWo_hi + qhi;
                        for (int dy = 0; dy < f; ++dy)
                            for (int dx = 0; dx < f; ++dx) {
                                const int sub = dy * Wo_hi + dx;
                                float v = v0;
                                if (res) v += res[rb + sub];
                                if (P.act == OTP_ACT_RELU) v = fmaxf(v, 0.f);
                                else if (P.act == OTP_ACT_GELU) v = gelu_erf(v);
                                out[ob + sub] = v;
                            }
                    }
                }
            }
        }
    };

    const float* wbase = wts + kl * P.MS + m_wave + i16;
    const float* ibase = inp + kl * P.CS;
    const int dW = P.dil * P.W;

    int tile = blockIdx.x, c0 = 0;
    if (!tile_ok(tile)) return;                  // uniform per workgroup
    // Workgroups b and b + 256 land on the same CU and their waves share SIMDs.  With equal priority the two
    // waves of a SIMD interleave MFMA by MFMA, drift into lockstep and then wait on LDS / barriers together; a
    // static priority for the second dispatch round lets one wave run its MFMA batch while the other fills the gaps.
    if (P.prio && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_setprio(1);
    load_items(tile, 0);
    store_items();
    __syncthreads();
    bool new_tile = true;
    while (true) {
        if (new_tile) {
            // ---- per-tile lane geometry and fresh accumulators ------------------------------------------------------
            const TileId T = decode(tile);
            const int pix_wave = T.q0 + wpi * 16 * PB;
            const int y_first = P.flat ? 0 : (int)fdiv((uint32_t)T.q0, P.magicWo);
            const int f0 = P.flat ? T.q0 : (y_first * P.stride - P.pad) * P.W;
            const int f0a = f0 & ~3;
            if (LIN) {
                // offset(q) = G + (f0 - f0a) - pad + (q - y_first * W)   (flat: G + q - q0)
                pl0 = P.flat ? P.G + (pix_wave + i16 - T.q0)
                             : P.G + (f0 - f0a) - P.pad + (pix_wave + i16 - y_first * P.W);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    if (KS == 1) {
                        keepL[pb] = keepR[pb] = -1;
                    } else {
                        const int q = pix_wave + pb * 16 + i16;
                        const int y = (int)fdiv((uint32_t)q, P.magicWo), x = q - y * P.Wo;
                        keepL[pb] = (x - P.dil >= 0) ? -1 : 0;
                        keepR[pb] = (x + P.dil < P.W) ? -1 : 0;
                    }
                }
            } else {
                uint32_t m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    int q = pix_wave + pb * 16 + i16;
                    q = q < P.HoWo ? q : T.q0;   // padding lanes compute a valid pixel and are never stored
                    const int y = (int)fdiv((uint32_t)q, P.magicWo), x = q - y * P.Wo;
                    poff[pb] = P.G + (f0 - f0a) - P.pad + (y - y_first) * P.stride * P.W + x * P.stride;
                    const int xi = x * P.stride - P.pad;
                    m0 |= (xi >= 0 && xi < P.W) ? 1u << pb : 0u;
                    m1 |= (xi + P.dil >= 0 && xi + P.dil < P.W) ? 1u << pb : 0u;
                    m2 |= (xi + 2 * P.dil >= 0 && xi + 2 * P.dil < P.W) ? 1u << pb : 0u;
                }
                cm0 = m0; cm1 = m1; cm2 = m2;
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // ---- next step: next chunk of this tile, else first chunk of the next tile, else none ---------------------
        const bool last_chunk = c0 + P.CK >= P.Cin;
        const int tile_n = last_chunk ? tile + (int)gridDim.x : tile;
        const int c0_n = last_chunk ? 0 : c0 + P.CK;
        const bool have_next = !last_chunk || tile_ok(tile_n);
        if (have_next) load_items(tile_n, c0_n);  // in flight while this chunk is multiplied

        // ---- MFMA over the chunk in LDS --------------------------------------------------------------------
        if (LIN) {
            // (hand-pipelining the fragment reads one step ahead behind sched_barriers measured 5-10 % SLOWER than
            // letting the compiler interleave reads, masks and MFMAs of one step: profiles/r01c_conv_notes.txt)
            const int aS = 4 * P.MS, bS = 4 * P.CS;                  // one k-step = 4 channels
            const float* ip0 = ibase + pl0;
#pragma unroll 1
            for (int ti = 0; ti < KS; ++ti) {
#pragma unroll
                for (int tj = 0; tj < KS; ++tj) {
                    const float* wrow = wbase + (ti * KS + tj) * P.CK * P.MS;
                    const float* irow = ip0 + (KS == 1 ? 0 : ti * dW + tj * P.dil);
#pragma unroll 1
                    for (int kc = 0; kc < P.CK; kc += 4) {
                        float a[MB], b[PB];
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) a[mb] = wrow[mb * 16];
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb) {
                            const float v = irow[pb * 16];
                            if (KS == 3 && tj == 0) b[pb] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & keepL[pb]);
                            else if (KS == 3 && tj == 2) b[pb] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & keepR[pb]);
                            else b[pb] = v;
                        }
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                            for (int pb = 0; pb < PB; ++pb)
                                acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[pb], acc[mb][pb], 0, 0, 0);
                        wrow += aS;
                        irow += bS;
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int tap = 0; tap < KK; ++tap) {
                const int ti = tap / KS, tj = tap - ti * KS;
                const uint32_t cm = tj == 0 ? cm0 : (tj == 1 ? cm1 : cm2);
                const float* wrow = wbase + tap * P.CK * P.MS;
                const float* irow = ibase + (KS == 1 ? 0 : ti * dW + tj * P.dil);
#pragma unroll 1
                for (int kc = 0; kc < P.CK; kc += 4) {
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
                }
            }
        }
        __syncthreads();                         // every wave finished reading this chunk
        if (last_chunk) {
            epilogue(decode(tile));
            if (!have_next) break;
            __syncthreads();                     // epilogue tiles are read before the next chunk overwrites the LDS
        }
        store_items();
        __syncthreads();
        new_tile = last_chunk;
        tile = tile_n;
        c0 = c0_n;
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
int g_prio = 0;                  // tuning hook (OTPOSE_CONV_PRIO=0 disables the static wave priority)
int g_last[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // tuning hook: plan of the last otp_conv2d call


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


struct Tile { int MB, PB, WM, WP, CK; };

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
    // a ragged last M tile is fine: its surplus weight columns alias the next row of the slab (finite values) or fall off
    // the tensor (range check -> 0), and rows >= Cout are never stored; only the whole-workgroup waste is priced
    P.M4 = P.Mtile / 4;
    P.MS = pad_stride(P.Mtile, 16);
    P.G = ((P.pad + 3) & ~3) < 16 ? 16 : ((P.pad + 3) & ~3);
    int L;
    if (P.flat) {
        L = P.Ptile;
    } else {
        int rows = (P.Ptile % P.Wo == 0) ? P.Ptile / P.Wo : (P.Ptile + P.Wo - 2) / P.Wo + 1;
        if (rows > P.Ho && !P.lin) rows = P.Ho;            // LIN reads the (zero) rows of its padding pixels too
        const int NR = (rows - 1) * P.stride + (KS - 1) * P.dil + 1;
        L = NR * P.W + 3;                                  // + alignment slack of the window start
    }
    P.L4 = (L + 3) / 4;
    P.CS = pad_stride(P.G + 4 * P.L4 + P.G, 16);           // == 16 (mod 32): the 4 k-rows of a fragment hit distinct banks
    // chunk of t.CK input channels (a multiple of 4, at most Cin rounded up): must fit the LDS budget of two
    // workgroups per CU and the prefetch registers
    const int cin4 = (P.Cin + 3) & ~3;
    const int c = t.CK > cin4 ? cin4 : t.CK;
    const int nwaves = P.nthreads / 64;
    P.JR = (P.L4 + 63) / 64;
    const size_t best_lds = ((size_t)c * P.CS + (size_t)KK * c * P.MS) * sizeof(float);
    const int ckw = (c + nwaves - 1) / nwaves;
    if (c < 4 || (c & 3) || best_lds > 80 * 1024 || ckw * P.JR > (P.lin ? MAXJI_LIN : MAXJI_GEN) ||
        (long)KK * c * P.M4 > (long)MAXJW * P.nthreads)
        return false;
    P.CK = c;
    P.CKW = ckw;
    P.NJI = P.CKW * P.JR;
    P.NW4 = KK * c * P.M4;
    P.NJW = (P.NW4 + P.nthreads - 1) / P.nthreads;
    P.ntiles = ((P.nP + 7) / 8) * 8 * P.nM;
    const size_t ep_lds = (size_t)nwaves * 16 * (16 * t.PB + 4) * sizeof(float);
    lds_bytes = best_lds > ep_lds ? best_lds : ep_lds;
    P.magicM4 = magic_of(P.M4);
    P.magicCK = magic_of(P.CK);
    P.magicWo = magic_of(P.Wo);
    P.magicTpi = magic_of(P.tiles_per_img);
    return true;
}

// Persistent grid of a plan: two waves per SIMD resident (<= 256 registers), bounded by LDS, a multiple of 8 so that a
// workgroup keeps its XCD label while it walks its tiles.
int resident_wgs(const WinPlan& P, size_t lds) {
    int resident = 256 * (8 / (P.nthreads / 64));
    const int by_lds = 256 * (int)((160 * 1024) / (lds ? lds : 1));
    if (by_lds < resident) resident = by_lds;
    resident &= ~7;
    return P.ntiles < resident ? P.ntiles : resident;
}

// Relative time of a candidate, in cycles of the busiest SIMD.  On gfx950 the f32 MFMA and the VALU work of the
// waves resident on a SIMD do not overlap (measured: profiles/r01c_*), so the time is the sum of the MFMA
// cycles and ~4 cycles per vector instruction; what is left of barrier / memory latency is hidden when two waves
// share the SIMD and exposed when one wave has it alone.
double tile_cost(const WinPlan& P, const Tile& t, int KS, size_t lds) {
    const int wpw = t.WM * t.WP;                                   // waves per workgroup
    const int R = resident_wgs(P, lds);
    // workgroup b walks tiles b, b+R, ...; workgroups b, b+256, ... share a CU and run concurrently, so the busiest
    // SIMD hosts ceil(workgroups x waves / 4) waves that take turns on its pipe
    const long T = P.ntiles;
    const long tiles_per_wg = (T + R - 1) / R;
    const int cu_wgs = (R + 255) / 256;
    const int simd_waves = (cu_wgs * wpw + 3) / 4;
    const double waves_per_simd = simd_waves;
    const int nchunks = (P.Cin + P.CK - 1) / P.CK, steps = KS * KS * (P.CK / 4);
    double step_valu = 2.0;                                        // address bumps
    if (!P.lin) step_valu += 2.0 * t.PB;                           // per-block address + mask
    else if (KS == 3) step_valu += t.PB * (2.0 / 3.0);             // edge-column masks on 2 of 3 tap columns
    // per step: the MFMAs, the vector instructions, and what is left of the LDS round trip of the fragment reads
    // (fitted to the tile sweeps in profiles/: ~200 cycles with a second wave on the SIMD, ~330 without)
    const double step = 32.0 * t.MB * t.PB + 4.0 * step_valu + (waves_per_simd >= 2.0 ? 200.0 : 330.0);
    double chunk = steps * step + 4.0 * (6.0 * P.NJI + 3.0 * P.NJW + 60.0) + 16.0 * (P.NJI + P.NJW) +
                         (waves_per_simd >= 2.0 ? 400.0 : 1500.0);   // barriers + load wait
    const double ep_items = 16.0 * 4 * t.PB * t.MB / 64.0;         // float4 stores per lane
    const double tile_fixed = 4.0 * (ep_items * (P.ep_vec ? 45.0 : 160.0) + 4.0 * t.MB * t.PB + 300.0);
    // a CU pulls the staged bytes of all its workgroups through one ~16 B/clk load path (L2-resident windows and
    // weights); stride-2 windows with narrow M tiles are bound by it
    // the loads of a chunk are issued one chunk ahead: a chunk shorter than the ~2500-cycle L2/HBM round trip leaves
    // the rest of it exposed at the LDS write (half of it when a second workgroup shares the CU)
    chunk += fmax(0.0, 2500.0 - chunk) * (waves_per_simd >= 2.0 ? 0.5 : 1.0);
    const double chunk_bytes = 16.0 * ((double)P.CK * P.L4 + (double)P.NW4);
    const double chunk_cu = fmax(simd_waves * chunk, cu_wgs * chunk_bytes / 16.0);
    double cost = (double)tiles_per_wg * (nchunks * chunk_cu + simd_waves * tile_fixed);
    // measured on the sweeps: at equal tile shape 2-wave workgroups run 7-24 % and 1-wave workgroups 17-26 % behind
    // 4-wave ones (the weight slab is staged once per workgroup)
    if (!P.lin) cost *= 1.0 + 0.2 * (P.nM - 1);                   // strided windows: re-staged by every M tile
    // tall 1x1 tiles read the input once instead of once per 48 output channels.  Stand-alone that only pays for 136->544
    // (tile-count quantisation hides it elsewhere); inside the forward graph, where the two temporal encoders share the GPU and
    // the L2, preferring them everywhere they fit measured 66.4 -> 65.0 ms per forward
    if (t.MB > 3) cost *= 0.7;
    if (wpw == 2) cost *= 1.12;
    else if (wpw == 3) cost *= 1.2;
    else if (wpw == 1) cost *= 2.0;                                // measured 2.4x on 384->48 1x1 @12x9
    return cost;
}

template <int MB, int PB, int KS, bool LIN>
int launch_win(const float* in, const float* wp, const float* scale, const float* shift,
               const float* res, float* out, const WinPlan& P, size_t lds, hipStream_t st) {
    auto kern = conv_win_kernel<MB, PB, KS, LIN>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    dim3 grid(resident_wgs(P, lds));
    g_last[4] = P.CK; g_last[5] = (int)grid.x; g_last[6] = (int)lds; g_last[7] = P.ntiles;
    hipLaunchKernelGGL(kern, grid, dim3(P.nthreads), lds, st, in, wp, scale, shift, res, out, P);
    return otp_launch_status();
}

template <int KS>
int dispatch_win(int MB, int PB, const float* in, const float* wp, const float* scale,
                 const float* shift, const float* res, float* out, const WinPlan& P, size_t lds, hipStream_t st) {
#define OTP_CASE(M_, P_)                                                                              \
    if (MB == M_ && PB == P_)                                                                         \
        return P.lin ? launch_win<M_, P_, KS, true>(in, wp, scale, shift, res, out, P, lds, st)       \
                     : launch_win<M_, P_, KS, false>(in, wp, scale, shift, res, out, P, lds, st);
    OTP_CASE(1, 7) OTP_CASE(1, 9)
    OTP_CASE(2, 7) OTP_CASE(2, 9)
    OTP_CASE(3, 7)
#undef OTP_CASE
    // tall tiles for the 1x1 GEMMs of the temporal encoders (one pass over the pixels instead of one per 48 channels)
    if constexpr (KS == 1) {
        if (P.lin && MB == 9 && PB == 3) return launch_win<9, 3, 1, true>(in, wp, scale, shift, res, out, P, lds, st);
    }
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

// Choose the window-kernel plan for a descriptor; false when the window kernel does not apply (then the generic
// kernel runs).  Pure host arithmetic: also exported as otp_conv2d_plan for tuning without a GPU.
bool plan_window(const otp_conv_desc& d, const void* in, const void* in2, const void* wpacked, const void* res,
                 const void* out, WinPlan& P, Tile& best, size_t& lds) {
    // the window kernel stages 16-byte vectors: channel planes must be 16-byte aligned (H*W % 4 == 0, aligned bases);
    // a pre-added second input (the RSB staircase, tiny convs) takes the generic kernel
    const bool vec_ok = ((d.H * d.W) & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(wpacked) & 15) == 0;
    if (!vec_ok || in2 || d.kh != d.kw || (d.kh != 1 && d.kh != 3) || (long)d.H * d.W >= (1l << 24)) return false;
    P = WinPlan{};
    P.N = d.N; P.Cin = d.Cin; P.H = d.H; P.W = d.W; P.HW = d.H * d.W; P.Cout = d.Cout;
    P.Cout16 = (d.Cout + 15) & ~15; P.stride = d.stride; P.pad = d.pad; P.dil = d.dil;
    P.Ho = d.Ho; P.Wo = d.Wo; P.HoWo = d.Ho * d.Wo;
    P.in_ctot = d.in_ctot; P.in_coff = d.in_coff; P.in2_ctot = d.in2_ctot; P.in2_coff = d.in2_coff;
    P.out_ctot = d.out_ctot; P.out_coff = d.out_coff; P.res_ctot = d.res_ctot; P.res_coff = d.res_coff;
    P.res_up = d.res_up; P.act = d.act; P.frame_split = d.frame_split;
    P.prio = g_prio;
    P.flat = (d.kh == 1 && d.stride == 1 && d.pad == 0) ? 1 : 0;
    P.lin = (P.flat || (d.stride == 1 && d.Wo == d.W && 2 * d.pad == d.dil * (d.kh - 1))) ? 1 : 0;
    P.ep_vec = (d.res_up <= 1 && ((d.Ho * d.Wo) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                (!res || (reinterpret_cast<uintptr_t>(res) & 15) == 0)) ? 1 : 0;
    const int mblk = P.Cout16 / 16, G = (P.HoWo + 15) / 16;
    best = Tile{0, 0, 0, 0, 0};
    double best_cost = 1e300;
    static const int wms[] = {1, 2, 4};
    // accumulator shapes (MB x PB 16x16 blocks per wave): the 7- / 9-block wide ones for everything, the tall one for
    // 1x1 GEMMs whose input would otherwise be re-staged once per 48 output channels (temporal-encoder projections / MLP)
    static const int shapes[][2] = {{1, 7}, {1, 9}, {2, 7}, {2, 9}, {3, 7}, {9, 3}};
    // strict pass: M tiles divide Cout16 and no wave is pure padding.  Tiny maps with few output channels (the 12x9 fuse
    // convs) then only admit single-wave workgroups, which measure 2.4x slower than a 4-wave workgroup with one idle
    // wave: a second, relaxed pass (one padding wave, one ragged M tile covering Cout) runs when that happens.
    for (int relaxed = 0; relaxed < 2; ++relaxed) {
        if (relaxed && (g_force[0] || (best.MB && best.WM * best.WP > 1))) break;
        for (const auto& sh : shapes)
            for (int WM : wms)
                for (int WP = 1; WM * WP <= 4; ++WP) {
                    const int MB = sh[0], PB = sh[1];
                    const bool tall = MB > 3;
                    if (tall && !(d.kh == 1 && P.lin)) continue;
                    if (g_force[0]) {
                        if (MB != g_force[0] || PB != g_force[1] || WM != g_force[2] || WP != g_force[3]) continue;
                    } else {
                        if (WM > 1 && MB * (WM - 1) >= mblk + relaxed) continue;   // whole waves of padding (one allowed when relaxed)
                        if (WP > 1 && PB * (WP - 1) >= G) continue;
                        if (tall ? (WM != 1 || WP != 4 || mblk < MB)
                                 : ((P.Cout16 % (16 * MB * WM)) != 0 && !(relaxed && 16 * MB * WM > P.Cout16))) continue;
                    }
                    const int cin4 = (d.Cin + 3) & ~3;
                    int last_ck = 0;
                    for (int ck : {cin4 <= 40 ? cin4 : 32, 32, 24, 16, 12, 8, 4}) {
                        if (ck > cin4) ck = cin4;
                        if (ck == last_ck) continue;
                        last_ck = ck;
                        const Tile t{MB, PB, WM, WP, ck};
                        WinPlan C = P;
                        size_t l = 0;
                        if (!fill_plan(C, t, d.kh, l)) continue;
                        const double cost = tile_cost(C, t, d.kh, l);
                        if (cost < best_cost) { best_cost = cost; best = t; }
                    }
                }
    }
    if (!best.MB) return false;
    fill_plan(P, best, d.kh, lds);
    return true;
}

}  // namespace

extern "C" int otp_conv2d_set_tile(int MB, int PB, int WM, int WP) {
    if (MB < 0) { g_prio = PB; return OTP_OK; }      // tuning: otp_conv2d_set_tile(-1, prio, 0, 0)
    g_force[0] = MB; g_force[1] = PB; g_force[2] = WM; g_force[3] = WP;
    return OTP_OK;
}

extern "C" int otp_conv2d_plan(const otp_conv_desc* desc, int* out8) {
    if (!desc || !out8) return OTP_ERR_BAD_ARG;
    WinPlan P{};
    Tile best{0, 0, 0, 0, 0};
    size_t lds = 0;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    // alignment is assumed (aligned dummy pointers): this is the plan the engine's own buffers get
    if (!plan_window(*desc, nullptr, nullptr, nullptr, nullptr, nullptr, P, best, lds)) return OTP_OK;
    out8[0] = best.MB; out8[1] = best.PB; out8[2] = best.WM; out8[3] = best.WP;
    out8[4] = P.CK; out8[5] = resident_wgs(P, lds); out8[6] = (int)lds; out8[7] = P.ntiles;
    return OTP_OK;
}

extern "C" int otp_conv2d_last_plan(int* out8) {
    if (!out8) return OTP_ERR_BAD_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = g_last[i];
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

    WinPlan P{};
    Tile best{0, 0, 0, 0, 0};
    size_t lds = 0;
    if (plan_window(d, in, in2, wpacked, res, out, P, best, lds)) {
        g_last[0] = best.MB; g_last[1] = best.PB; g_last[2] = best.WM; g_last[3] = best.WP;
        if (d.kh == 1) return dispatch_win<1>(best.MB, best.PB, a, w, sc, sh, r, o, P, lds, st);
        return dispatch_win<3>(best.MB, best.PB, a, w, sc, sh, r, o, P, lds, st);
    }
    ConvPlan G;
    G.d = d;
    for (int i = 0; i < 8; ++i) g_last[i] = 0;       // generic kernel
    if (!choose_generic(G)) return OTP_ERR_UNSUPPORTED;
    const int MB = G.Mtile / 16;
    if (MB == 1) return launch_generic<1>(a, b, w, sc, sh, r, o, G, st);
    if (MB == 2) return launch_generic<2>(a, b, w, sc, sh, r, o, G, st);
    return launch_generic<3>(a, b, w, sc, sh, r, o, G, st);
}
