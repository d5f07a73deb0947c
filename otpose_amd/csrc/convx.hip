// fp32 convolution on the gfx950 bf16 matrix cores: "split" (bf16x3) products, fp32 storage and accumulation.
//
// The HRNet convs (reference model/HRNet.py:500-571, 442-470) are fp32 NCHW tensors.  The f32 MFMA
// (v_mfma_f32_16x16x4_f32, 157 TFLOP/s) bounds the whole OTPose forward at ~42 ms; the bf16 MFMA
// (v_mfma_f32_16x16x32_bf16) is 16x faster per instruction.  Every fp32 operand is split into two bf16 pieces
//     a = a_hi + a_lo + r,   a_hi = bf16_rne(a),  a_lo = bf16_rne(a - a_hi),   |r| <= 2^-18 |a|
// and a product is accumulated in fp32 as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  (three MFMAs; the dropped a_lo*b_lo and
// remainder terms are <= 3 * 2^-18 |a b|, typically 2^-19: the same order as the fp32 rounding of a K = 432 accumulation).
// Three bf16 MFMAs cost 3/16 of the f32 MFMA they replace, so the convolution turns HBM-bound.
//
// Implicit GEMM: M = 256 consecutive output pixels (flattened over the images of the batch: NCHW planes are contiguous, so
// a lane's 4 accumulator rows are one aligned float4 of the output), N = 16 * NTW output channels, K = (tap, input channel)
// in chunks of 16 channels.  Per chunk the input WINDOW (the image rows the 256 pixels touch, all columns, zero rows between
// images / zero columns left and right) travels global fp32 NCHW -> registers -> split -> LDS as per-pixel records
// [16 ch hi | 16 ch lo | 16 B pad] (80 B: an odd multiple of 16 B, so the 16 lanes of a fragment read hit 16 distinct bank
// groups), the packed weights arrive in MFMA fragment order (lane-linear, conflict-free).  One k-step = 2 taps x 16 channels;
// the loads of chunk c+1 are in flight under the MFMAs of chunk c.
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef __bf16 bf16;
typedef otp_x3x8 h16x8;              // 8 operand pieces of the split products (common.h: IEEE half since round 4)
typedef otp_x3x2 h16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Two shapes.  Stride 1: chunks of CK = 16 channels (k-step = 2 taps x 16 channels, 5 steps, the tenth tap has zero
// weights), 256 pixels (4 m-tiles per wave).  Stride 2 reads 4x the input per output pixel: chunks of 8 channels (k-step =
// 4 taps x 8 channels, 3 steps) and 128 pixels (2 m-tiles per wave) keep the window inside half the LDS.
// Pointwise (1x1, stride 1) convolutions: chunks of 32 channels = one k-step, 256 pixels, no halo.
constexpr int xck(int stride, int taps = 9) { return taps == 1 ? 32 : (stride == 2 ? 8 : 16); }   // input channels per chunk
constexpr int xmtw(int stride) { return stride == 2 ? 2 : 4; }            // m-tiles (16 pixels) per wave
constexpr int xpix(int ck) { return ck * 4 + 16; }                        // bytes of a window pixel record: hi | lo | 16 B pad
constexpr int xks(int ck, int taps = 9) { return (taps * (ck / 8) + 3) / 4; }   // k-steps per chunk
constexpr int XOOB = -16;      // buffer offset past any descriptor: the load returns zeros

#ifdef OTP_CONVX_TIMING
// development build only (tools/convx_timing.py): per-workgroup phase stamps, never in libotpose_hip.so
__device__ unsigned long long otp_convx_stamps[8192 * 16];
#define XSTAMP(slot)                                                                                  \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 8192) otp_convx_stamps[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define XSTAMP(slot)
#endif

__device__ __forceinline__ uint32_t xdiv(uint32_t i, uint32_t magic) { return magic ? __umulhi(i, magic) : i; }
uint32_t xmagic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }   // exact while i * d < 2^32

struct XPlan {
    int N, Cin, H, W, HW, Cout, Ho, Wo, HoWo, total;
    int in_ctot, in_coff, out_ctot, out_coff, res_ctot, res_coff, act;
    int stride, pad, dil, taps;
    float post;                            // otp_conv_desc.out_scale (1 when unset)
    unsigned* rflag;                       // range-guard word (common.h)
    int NTW, nN, nTiles, nChunks, tpx;
    int VR, WPp, CS, rowsMax, S, NI;
    int winBytes, wBytes;
    uint32_t mHoWo, mWo, mW;
};

// split 8 floats into bf16 hi / lo vectors
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 a = {v[2 * i], v[2 * i + 1]};
        const h16x2 ah = __builtin_convertvector(a, h16x2);
        const uint32_t hb = __builtin_bit_cast(uint32_t, ah);
        const f32x2 af = otp_x3_widen(hb);
        const h16x2 al = __builtin_convertvector(a - af, h16x2);
        h[i] = hb;
        l[i] = __builtin_bit_cast(uint32_t, al);
    }
    hi = (u32x4){h[0], h[1], h[2], h[3]};
    lo = (u32x4){l[0], l[1], l[2], l[3]};
}

// (Cout, Cin, 3, 3) fp32 (x scale[cout]) -> [cout block][chunk][k-step][n-tile][hi, lo][lane][8] bf16: the B fragments of
// v_mfma_f32_16x16x32_bf16 (lane = (cout & 15) + 16 * kl; G = CK / 8 channel groups: kl -> tap (4 / G) s + kl / G, channels
// 8 (kl % G) .. + 7 of the chunk)
__global__ void convx_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, u32x4* __restrict__ out, int Cout,
                                  int Cin, int NTW, int nN, int nChunks, int CK, int TAPS) {
    const int G = CK / 8, XKS = xks(CK, TAPS);
    const int total = nN * nChunks * XKS * NTW * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63;
        int r = idx >> 6;
        const int t = r % NTW; r /= NTW;
        const int s = r % XKS; r /= XKS;
        const int chunk = r % nChunks, cb = r / nChunks;
        const int cout = (cb * NTW + t) * 16 + (lane & 15), kl = lane >> 4;
        const int q = 4 * s + kl, tap = q / G, ci0 = chunk * CK + 8 * (q % G);      // k-slot q of the chunk -> (tap, channel group)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = ci0 + j;
            v[j] = (tap < TAPS && cout < Cout && ci < Cin) ? w[((size_t)cout * Cin + ci) * TAPS + tap] * (scale ? scale[cout] : 1.f) : 0.f;
        }
        u32x4 hi, lo;
        split8(v, hi, lo);
        const size_t o = ((((size_t)(cb * nChunks + chunk) * XKS + s) * NTW + t) * 2) * 64 + lane;
        out[o] = hi;
        out[o + 64] = lo;
    }
}

template <int CK, int MTW, int NI, int NTW, int TAPS>
__global__ __launch_bounds__(256, 2) void convx_kernel(const float* __restrict__ in, const u32x4* __restrict__ wpk,
                                                        const float* __restrict__ shift, const float* res, float* out,
                                                        const XPlan P) {
    constexpr int G = CK / 8, XKS = xks(CK, TAPS), XPIX = xpix(CK), XBM = 64 * MTW, LO = CK * 2;
    constexpr int WUNITS = XKS * NTW * 2 * 64;                     // 16-byte units of a chunk's weights
    constexpr int NWL = (WUNITS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* win = smem;
    unsigned char* wl = smem + P.winBytes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kl = lane >> 4;

    // workgroup -> (pixel tile, output-channel block).  Block ids are dealt round-robin to the 8 XCDs: XCD x walks the
    // contiguous tile range [x * tpx, (x + 1) * tpx), the cout blocks of a tile back to back, so that the window rows shared
    // by neighbouring tiles and the re-read window of the next cout block hit that XCD's L2.
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int tl = j / P.nN, cb = j - tl * P.nN;
    const int tile = xcd * P.tpx + tl;
    if (tile >= P.nTiles) return;
    XSTAMP(0);
#ifdef OTP_CONVX_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 8192) otp_convx_stamps[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime();
#endif
    const int P0 = tile * XBM;
    const int n0 = P0 / P.HoWo, p0 = P0 - n0 * P.HoWo;
    const int yo0 = p0 / P.Wo;
    const int Vfirst = n0 * P.VR + yo0 * P.stride;                // first virtual row (image n: rows n*VR .. n*VR+pad-1 are zero rows)

    // ---- window items: (channel group g, float4 f of an image plane) -> 8 channel loads + 4 pixel records ------------------
    // The window rows of image n are one contiguous run of its NCHW plane, so the items of a channel group walk aligned
    // float4s of that run (any W; H*W % 4 == 0); each of the 4 pixels finds its own record (row, de-interleaved column).
    int goff[NI], ldst[NI][4];
    if constexpr (TAPS == 1) {
        // pointwise: no halo, the window is the tile - XBM / 4 float4 items per channel group, item q of a group = flat output
        // pixels P0 + 4 q .. + 3 (one image: HW % 4 == 0), record index = flat pixel - first pixel of the tile's first row.
        // (The general walk below covers whole rows: one more item than a thread per group can hold in one pass, and 4 k cycles
        // of index arithmetic per workgroup - a third of a 64 -> 256 workgroup's prologue.)
        constexpr int TQ = XBM / 4;
        const int row0 = n0 * P.HW + yo0 * P.W;                    // flat index of the first pixel of the tile's first row
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid + 256 * j, g = i / TQ, q = i - g * TQ;
            const int flat = P0 + 4 * q;
            const bool ok = g < G && flat < P.total;
            const int n = (int)xdiv((uint32_t)(ok ? flat - n0 * P.HW : 0), P.mHoWo);      // images past the tile's first
            const int pin = (ok ? flat - n0 * P.HW : 0) - n * P.HW;
            goff[j] = ok ? ((n * P.in_ctot + 8 * g) * P.HW + pin) * 4 : XOOB;
#pragma unroll
            for (int k = 0; k < 4; ++k) ldst[j][k] = ok ? (flat + k - row0) * XPIX + g * 16 : -1;
        }
    } else {
        int T = 0;                                                     // items per channel group of this tile (uniform)
        for (int sl = 0; sl < P.S; ++sl) {
            const int n = n0 + sl, ya = max(0, Vfirst - n * P.VR - P.pad), yb = min(P.H, Vfirst + P.rowsMax - n * P.VR - P.pad);
            T += (n < P.N && yb > ya) ? ((yb * P.W + 3) >> 2) - ((ya * P.W) >> 2) : 0;
        }
    #pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid + 256 * j;
            int g = 0;
    #pragma unroll
            for (int gg = 1; gg < G; ++gg) g += i >= gg * T ? 1 : 0;
            int rem = i - g * T, f = -1, n = 0;
            if (i >= G * T) rem = -1;
            for (int sl = 0; sl < P.S; ++sl) {
                const int ns = n0 + sl, ya = max(0, Vfirst - ns * P.VR - P.pad), yb = min(P.H, Vfirst + P.rowsMax - ns * P.VR - P.pad);
                const int c = (ns < P.N && yb > ya) ? ((yb * P.W + 3) >> 2) - ((ya * P.W) >> 2) : 0;
                if (f < 0 && rem >= 0 && rem < c) { f = ((ya * P.W) >> 2) + rem; n = ns; }
                rem -= (f < 0) ? c : 0;
            }
            goff[j] = f >= 0 ? ((((n - n0) * P.in_ctot + 8 * g) * P.HW) + 4 * f) * 4 : XOOB;
            const int y0 = (int)xdiv((uint32_t)(f >= 0 ? 4 * f : 0), P.mW);
            int y = y0, x = (f >= 0 ? 4 * f : 0) - y0 * P.W;
    #pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = n * P.VR + P.pad + y - Vfirst;
                const int xw = x + P.pad;
                const int col = P.stride == 2 ? (xw & 1) * P.CS + (xw >> 1) : xw;
                ldst[j][k] = (f >= 0 && r >= 0 && r < P.rowsMax && y < P.H) ? (r * P.WPp + col) * XPIX + g * 16 : -1;
                if (++x == P.W) { x = 0; ++y; }
            }
        }
    }
    const size_t in_base = ((size_t)n0 * P.in_ctot + P.in_coff) * P.HW;
    const size_t in_left = ((size_t)P.N * P.in_ctot) * P.HW - in_base;
    // (descriptor size capped below XOOB so that the masked offset is always out of range)
    const otp_rsrc rin = make_rsrc32(in + in_base, in_left * 4 > 0x7fffffffull ? 0x7fffffffu : (unsigned)(in_left * 4));
    const otp_rsrc rw = make_rsrc32(wpk, (unsigned)((size_t)P.nN * P.nChunks * WUNITS * 16));

    // staging registers: one set (the loads of chunk c + 1 fly under the MFMAs of chunk c); pointwise launches - one item per
    // thread, short MFMA phases, bound by the HBM round trip - keep two sets and load two chunks ahead
    constexpr int NBUF = TAPS == 1 ? 2 : 1;
    f32x4 xv[NBUF][NI][8];
    u32x4 wv[NBUF][NWL];
    auto load_chunk = [&](int c, auto bufc) __attribute__((always_inline)) {
        constexpr int b = decltype(bufc)::value;
        const int cs = c * CK * P.HW * 4;                          // scalar byte offset of the chunk's first channel
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                xv[b][j][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, goff[j], cs + e * P.HW * 4, 0));
        const int wb = ((cb * P.nChunks + c) * WUNITS) * 16;
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + 256 * j;
            wv[b][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, i < WUNITS ? wb + i * 16 : XOOB, 0, 0));
        }
    };
    auto store_chunk = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int b = decltype(bufc)::value;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = xv[b][j][e][k];
                u32x4 hi, lo;
                split8(v, hi, lo);
                if (ldst[j][k] >= 0) {
                    *reinterpret_cast<u32x4*>(win + ldst[j][k]) = hi;
                    *reinterpret_cast<u32x4*>(win + ldst[j][k] + LO) = lo;
                }
            }
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + 256 * j;
            if (i < WUNITS) reinterpret_cast<u32x4*>(wl)[i] = wv[b][j];
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, NBUF - 1>;

    XSTAMP(11);
    load_chunk(0, B0{});
    if (NBUF == 2 && P.nChunks > 1) load_chunk(1, B1{});
    XSTAMP(12);
    // zero the window once: the zero rows between / around the images, the left / right padding columns and the slack records
    // behind the last row are never written by the staging
    // (a pointwise window has no such records: every record a fragment reads is written, the rows of a ragged last tile read
    // a clamped valid record)
    if constexpr (TAPS != 1)
        for (int i = tid; i < P.winBytes / 16; i += 256) reinterpret_cast<u32x4*>(win)[i] = (u32x4){0u, 0u, 0u, 0u};

    XSTAMP(13);
    // ---- fragment addresses ---------------------------------------------------------------------------------------------------
    int mbase[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        int m = (wave * MTW + mt) * 16 + i16;
        if (P0 + m >= P.total) m = P.total - 1 - P0;               // tail tile: a finite address, the result is dropped
        const int p = p0 + m;
        const int dn = (int)xdiv((uint32_t)p, P.mHoWo), pi = p - dn * P.HoWo;
        const int yo = (int)xdiv((uint32_t)pi, P.mWo), xo = pi - yo * P.Wo;
        const int r = (n0 + dn) * P.VR + yo * P.stride - Vfirst;
        mbase[mt] = (r * P.WPp + xo) * XPIX;                        // (columns are de-interleaved by x mod stride)
    }
    int toff[XKS];
#pragma unroll
    for (int s = 0; s < XKS; ++s) {
        const int q = 4 * s + kl;
        int tap = q / G;
        if (tap > TAPS - 1) tap = TAPS - 1;                        // zero weights: any finite data
        const int dy = tap / 3, dx = tap - dy * 3;
        const int xs = dx * P.dil;
        toff[s] = ((dy * P.dil) * P.WPp + (P.stride == 2 ? (xs & 1) * P.CS + (xs >> 1) : xs)) * XPIX + (q % G) * 16;
    }

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Fragment pipeline: the LDS reads of the next n-tile (and, at the last n-tile of a k-step, of the next step's pixel
    // fragments) are issued before the 12 MFMAs of the current one, so no MFMA waits on an LDS round trip.
    auto mfma_phase = [&]() __attribute__((always_inline)) {
        h16x8 ah[2][MTW], al[2][MTW], bh[2], bl[2];
        auto load_a = [&](int buf, int s) __attribute__((always_inline)) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const unsigned char* a = win + mbase[mt] + toff[s];
                ah[buf][mt] = *reinterpret_cast<const h16x8*>(a);
                al[buf][mt] = *reinterpret_cast<const h16x8*>(a + LO);
            }
        };
        auto load_b = [&](int buf, int s, int t) __attribute__((always_inline)) {
            const unsigned char* b = wl + ((s * NTW + t) * 2) * 1024 + lane * 16;
            bh[buf] = *reinterpret_cast<const h16x8*>(b);
            bl[buf] = *reinterpret_cast<const h16x8*>(b + 1024);
        };
        load_a(0, 0);
        load_b(0, 0, 0);
#pragma unroll
        for (int s = 0; s < XKS; ++s)
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int cur = (s * NTW + t) & 1, sa = s & 1;
                if (t + 1 < NTW) {
                    load_b(cur ^ 1, s, t + 1);
                } else if (s + 1 < XKS) {
                    load_b(cur ^ 1, s + 1, 0);
                    load_a(sa ^ 1, s + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    acc[mt][t] = OTP_X3_MFMA(al[sa][mt], bh[cur], acc[mt][t], 0, 0, 0);
                    acc[mt][t] = OTP_X3_MFMA(ah[sa][mt], bl[cur], acc[mt][t], 0, 0, 0);
                    acc[mt][t] = OTP_X3_MFMA(ah[sa][mt], bh[cur], acc[mt][t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    XSTAMP(14);
    __syncthreads();                                               // window zeroed
    XSTAMP(1);
    // chunk c sits in register set c % NBUF; once it is in LDS the set takes chunk c + NBUF
    auto step = [&](int c, auto bufc) __attribute__((always_inline)) {
        store_chunk(bufc);
        if (c == 0) XSTAMP(2);
        __syncthreads();
        if (c == 0) XSTAMP(3);
        if (c + NBUF < P.nChunks) load_chunk(c + NBUF, bufc);      // in flight under the MFMAs
        __builtin_amdgcn_sched_barrier(0);                         // (hipcc otherwise sinks the loads below the MFMAs)
        mfma_phase();
        if (c == 0) XSTAMP(4);
        __syncthreads();                                           // every wave is done with the LDS image of chunk c
        if (c == 0) XSTAMP(5);
    };
    int c = 0;
    if constexpr (NBUF == 2) {
        for (; c + 2 <= P.nChunks - 1; c += 2) {
            step(c, B0{});
            step(c + 1, B1{});
        }
        if (c < P.nChunks - 1) {
            step(c, B0{});
            store_chunk(B1{});
        } else {
            store_chunk(B0{});
        }
    } else {
        for (; c < P.nChunks - 1; ++c) step(c, B0{});
        store_chunk(B0{});
    }
    __syncthreads();
    XSTAMP(6);

    // ---- epilogue addresses + residual prefetch (the staging registers are free now) -------------------------------------------
    const int co0 = (cb * NTW) * 16 + i16;
    int eo[MTW], ro[MTW];
    bool ev[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int m = (wave * MTW + mt) * 16 + 4 * kl;
        ev[mt] = P0 + m < P.total;
        const int p = p0 + (ev[mt] ? m : 0);
        const int dn = (int)xdiv((uint32_t)p, P.mHoWo), pi = p - dn * P.HoWo;
        eo[mt] = ((n0 + dn) * P.out_ctot + P.out_coff) * P.HoWo + pi;
        ro[mt] = ((n0 + dn) * P.res_ctot + P.res_coff) * P.HoWo + pi;
    }
    // residual: unconditional loads (a masked element re-reads a valid address of the output's own first pixel row)
    f32x4 rv[MTW][NTW];
    const float* rp = res ? res : out;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int co = co0 + 16 * t;
            const bool ok = res && ev[mt] && co < P.Cout;
            const size_t o = ok ? (size_t)ro[mt] + (size_t)co * P.HoWo : 0;
            rv[mt][t] = *reinterpret_cast<const f32x4*>(rp + o);
            if (!ok) rv[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    float shv[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int co = co0 + 16 * t;
        const bool ok = shift && co < P.Cout;
        shv[t] = (shift ? shift : rp)[ok ? co : 0];
        if (!ok) shv[t] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_phase();
    XSTAMP(7);

    const float lo = P.act == OTP_ACT_RELU ? 0.f : -INFINITY;
    bool bad = false;                                            // range guard (common.h): tested before the ReLU swallows a NaN
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int co = co0 + 16 * t;
        const float sh = shv[t];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            f32x4 y = acc[mt][t] * P.post + sh + rv[mt][t];      // post = 2^-k: the packed weights carry 2^k (out_scale)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bad |= otp_out_of_range(y[r]);
                y[r] = fmaxf(y[r], lo);
            }
            if (ev[mt] && co < P.Cout) *reinterpret_cast<f32x4*>(out + (size_t)eo[mt] + (size_t)co * P.HoWo) = y;
        }
    }
    otp_range_report(P.rflag, bad, OTP_RANGE_CONVX);
    XSTAMP(8);
#ifdef OTP_CONVX_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 8192) otp_convx_stamps[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
#endif
}

int x3_ntw(int Cout, int stride, int taps);

bool convx_plan(const otp_conv_desc& d, XPlan& P) {
    const bool pointwise = d.kh == 1 && d.kw == 1;
    if (!(pointwise || (d.kh == 3 && d.kw == 3)) || d.res_up > 1 || d.frame_split > 0 || d.in2_ctot > 0) return false;
    if ((d.stride != 1 && d.stride != 2) || d.act == OTP_ACT_GELU) return false;
    if (pointwise && (d.stride != 1 || d.pad != 0)) return false;
    const int TAPS = pointwise ? 1 : 9, KE = pointwise ? 0 : 2 * d.dil;       // kernel extent - 1
    const int CK = xck(d.stride, TAPS), XBM = 64 * xmtw(d.stride), XKS = xks(CK, TAPS), XPIX = xpix(CK);
    if (d.Cin % CK || ((d.H * d.W) & 3)) return false;
    const int Ho = (d.H + 2 * d.pad - KE - 1) / d.stride + 1, Wo = (d.W + 2 * d.pad - KE - 1) / d.stride + 1;
    if (Ho != d.Ho || Wo != d.Wo || Ho <= 0 || Wo <= 0) return false;
    if ((Ho * Wo) & 3) return false;                               // a lane's 4 pixels stay inside one image, 16-byte aligned
    P.N = d.N; P.Cin = d.Cin; P.H = d.H; P.W = d.W; P.HW = d.H * d.W; P.Cout = d.Cout;
    P.Ho = Ho; P.Wo = Wo; P.HoWo = Ho * Wo; P.total = d.N * P.HoWo;
    P.in_ctot = d.in_ctot; P.in_coff = d.in_coff; P.out_ctot = d.out_ctot; P.out_coff = d.out_coff;
    P.res_ctot = d.res_ctot; P.res_coff = d.res_coff; P.act = d.act;
    P.post = d.out_scale > 0.f ? d.out_scale : 1.f;
    P.stride = d.stride; P.pad = d.pad; P.dil = d.dil;
    const int c16 = (d.Cout + 15) / 16;
    P.NTW = x3_ntw(d.Cout, d.stride, TAPS);
    P.nN = (c16 + P.NTW - 1) / P.NTW;
    P.nTiles = (P.total + XBM - 1) / XBM;
    P.nChunks = d.Cin / CK;
    P.tpx = (P.nTiles + 7) / 8;
    P.VR = d.H + d.pad;                                            // virtual rows per image: pad zero rows, then the H image rows
    if ((Ho - 1) * d.stride >= P.VR) return false;
    P.CS = (d.W + 2 * d.pad + d.stride - 1) / d.stride;            // records per column-parity class
    P.WPp = P.CS * d.stride;
    int rows = 0;
    for (int t = 0; t < P.nTiles; ++t) {
        const int a = t * XBM, b = (a + XBM < P.total ? a + XBM : P.total) - 1;
        const int na = a / P.HoWo, ya = (a % P.HoWo) / Wo, nb = b / P.HoWo, yb = (b % P.HoWo) / Wo;
        const int r = (nb * P.VR + yb * d.stride + KE) - (na * P.VR + ya * d.stride) + 1;
        if (r > rows) rows = r;
    }
    P.rowsMax = rows;
    int Tmax = 0, S = 1;
    for (int t = 0; t < P.nTiles; ++t) {                           // window items / image slots of every tile (as the kernel counts them)
        const int a = t * XBM, n0 = a / P.HoWo, Vfirst = n0 * P.VR + ((a % P.HoWo) / Wo) * d.stride;
        int T = 0, sl = 0;
        for (;; ++sl) {
            const int n = n0 + sl, lo = Vfirst - n * P.VR - d.pad, hi = Vfirst + rows - n * P.VR - d.pad;
            if (hi <= 0 || n >= d.N) break;
            const int ya = lo > 0 ? lo : 0, yb = hi < d.H ? hi : d.H;
            if (yb > ya) T += ((yb * d.W + 3) >> 2) - ((ya * d.W) >> 2);
        }
        if (T > Tmax) Tmax = T;
        if (sl > S) S = sl;
    }
    P.S = S;
    P.NI = pointwise ? ((CK / 8) * (XBM / 4) + 255) / 256 : ((CK / 8) * Tmax + 255) / 256;   // pointwise: exactly the tile
    P.winBytes = (rows * P.WPp + KE + 4) * XPIX;                   // + slack: the clamped tenth tap / tail pixels stay inside
    P.winBytes = (P.winBytes + 15) & ~15;
    P.wBytes = XKS * P.NTW * 2 * 1024;
    P.mHoWo = xmagic(P.HoWo); P.mWo = xmagic(Wo); P.mW = xmagic(d.W);
    // exactness of the magic divisions (numerators < 2^32 / divisor) and 31-bit byte offsets
    if ((long)(P.HoWo + XBM) * P.HoWo >= (1l << 32) || (long)P.HW * d.W >= (1l << 32)) return false;
    if ((long)(S + 1) * d.in_ctot * P.HW * 4 >= (1l << 31)) return false;
    if ((long)P.N * d.out_ctot * P.HoWo >= (1l << 31) || (long)P.N * (d.res_ctot > 0 ? d.res_ctot : 1) * P.HoWo >= (1l << 31)) return false;
    if ((size_t)P.nN * P.nChunks * XKS * P.NTW * 2 * 1024 >= (1ull << 31)) return false;
    P.taps = TAPS;
    if (P.NI > 2 || S > 64) return false;
    return P.winBytes + P.wBytes <= OTP_LDS_LIMIT;
}

template <int CK, int MTW, int NI, int NTW, int TAPS>
int convx_launch(const float* in, const u32x4* wpk, const float* shift, const float* res, float* out, const XPlan& P, hipStream_t st) {
    auto kern = convx_kernel<CK, MTW, NI, NTW, TAPS>;
    const size_t lds = (size_t)P.winBytes + P.wBytes;
    OTP_ALLOW_BIG_LDS(kern, lds);
    const dim3 grid(8 * P.tpx * P.nN);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, in, wpk, shift, res, out, P);
    return otp_launch_status();
}

template <int NTW>
int convx_dispatch(const float* in, const u32x4* wpk, const float* shift, const float* res, float* out, const XPlan& P, hipStream_t st) {
    if (P.taps == 1)
        return P.NI <= 1 ? convx_launch<xck(1, 1), xmtw(1), 1, NTW, 1>(in, wpk, shift, res, out, P, st)
                         : convx_launch<xck(1, 1), xmtw(1), 2, NTW, 1>(in, wpk, shift, res, out, P, st);
    if (P.stride == 2)
        return P.NI <= 1 ? convx_launch<xck(2), xmtw(2), 1, NTW, 9>(in, wpk, shift, res, out, P, st)
                         : convx_launch<xck(2), xmtw(2), 2, NTW, 9>(in, wpk, shift, res, out, P, st);
    return P.NI <= 1 ? convx_launch<xck(1), xmtw(1), 1, NTW, 9>(in, wpk, shift, res, out, P, st)
                     : convx_launch<xck(1), xmtw(1), 2, NTW, 9>(in, wpk, shift, res, out, P, st);
}

int x3_ntw(int Cout, int stride, int taps) {
    const int c16 = (Cout + 15) / 16;
    if (stride == 2 || taps == 1) return (c16 % 3 == 0) ? 3 : (c16 % 4 == 0 ? 4 : (c16 <= 2 ? 2 : (c16 % 2 == 0 ? 2 : 3)));
    // stride 1: 3 n-tiles per workgroup when they divide Cout, else 2: with 4 the accumulators + staging registers spill and
    // the LDS image (window + 40 KB of weights) leaves one workgroup per CU (64 -> 64 @96x72: 220 us with 4, 194 with 2)
    return (c16 % 3 == 0) ? 3 : (c16 % 2 == 0 || c16 <= 2 ? 2 : 3);
}

}  // namespace

#ifdef OTP_CONVX_TIMING
extern "C" int otp_convx_read_stamps(void* host_out, size_t bytes) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(otp_convx_stamps), bytes) == hipSuccess ? OTP_OK : OTP_ERR_LAUNCH;
}
#endif

extern "C" int otp_conv2d_x3_supported(const otp_conv_desc* desc) {
    if (!desc) return 0;
    XPlan P{};
    return convx_plan(*desc, P) ? 1 : 0;
}

extern "C" size_t otp_conv2d_x3_weight_bytes(int Cout, int Cin, int k, int stride) {
    if (Cout <= 0 || Cin <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2) || (k == 1 && stride != 1)) return 0;
    const int TAPS = k * k, CK = xck(stride, TAPS);
    if (Cin % CK) return 0;
    const int NTW = x3_ntw(Cout, stride, TAPS), nN = ((Cout + 15) / 16 + NTW - 1) / NTW;
    return (size_t)nN * (Cin / CK) * xks(CK, TAPS) * NTW * 2 * 1024;
}

extern "C" int otp_conv2d_x3_pack_weight(const void* weight, const void* scale, void* wpacked, int Cout, int Cin, int k,
                                         int stride, void* stream) {
    if (!weight || !wpacked || Cout <= 0 || Cin <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_conv2d_x3_weight_bytes(Cout, Cin, k, stride)) return OTP_ERR_UNSUPPORTED;
    const int TAPS = k * k, CK = xck(stride, TAPS), NTW = x3_ntw(Cout, stride, TAPS);
    const int nN = ((Cout + 15) / 16 + NTW - 1) / NTW, nChunks = Cin / CK;
    const int total = nN * nChunks * xks(CK, TAPS) * NTW * 64;
    hipLaunchKernelGGL(convx_pack_kernel, dim3(otp_ceil_div(total, 256) > 2048 ? 2048 : otp_ceil_div(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(weight), static_cast<const float*>(scale),
                       static_cast<u32x4*>(wpacked), Cout, Cin, NTW, nN, nChunks, CK, TAPS);
    return otp_launch_status();
}

extern "C" int otp_conv2d_x3(const void* in, const void* wpacked, const void* shift, const void* res, void* out,
                             const otp_conv_desc* desc, void* stream) {
    if (!in || !wpacked || !out || !desc) return OTP_ERR_BAD_ARG;
    const otp_conv_desc& d = *desc;
    if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0) return OTP_ERR_BAD_ARG;
    if (d.in_ctot < d.in_coff + d.Cin || d.out_ctot < d.out_coff + d.Cout || (res && d.res_ctot < d.res_coff + d.Cout))
        return OTP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(res)) & 15)
        return OTP_ERR_UNSUPPORTED;
    XPlan P{};
    if (!convx_plan(d, P)) return OTP_ERR_UNSUPPORTED;
    P.rflag = otp_range_word();
    auto st = static_cast<hipStream_t>(stream);
    auto fi = static_cast<const float*>(in);
    auto fw = static_cast<const u32x4*>(wpacked);
    auto fs = static_cast<const float*>(shift);
    auto fr = static_cast<const float*>(res);
    auto fo = static_cast<float*>(out);
    switch (P.NTW) {
        case 2: return convx_dispatch<2>(fi, fw, fs, fr, fo, P, st);
        case 3: return convx_dispatch<3>(fi, fw, fs, fr, fo, P, st);
        default: return convx_dispatch<4>(fi, fw, fs, fr, fo, P, st);
    }
}
