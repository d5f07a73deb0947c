// 3x3 stride-1 convolution in Winograd F(2x2, 3x3) form on the gfx950 f32 matrix cores.
//
// The HRNet branch convs (reference model/HRNet.py:500-530 BasicBlock, :533-571 Bottleneck conv2; 73 % of the MACs of
// the OTPose forward) are 3x3 / stride 1 / pad 1.  Y = A^T [ (G g G^T) . (B^T d B) ] A turns each 2x2 output tile into
// 16 independent products instead of 36, i.e. 16 GEMMs  M_xy[co][tile] = sum_ci U_xy[co][ci] * V_xy[ci][tile]  - 2.25x
// fewer multiplies, all in fp32 (v_mfma_f32_16x16x4_f32); the transforms only add and subtract (G carries the 1/2 s into
// the pre-packed weights), so the result differs from the direct form by fp32 rounding of a few extra additions.
//
// Workgroup = 4 waves = (NT = 48 consecutive output tiles of one image, flattened row-major over the tile grid) x
// (48 output channels).  Per chunk of 8 input channels:
//   1. the input WINDOW (the rows the 48 tiles touch, contiguous in the NCHW plane) and the 16 x 8 x 48 slab of packed
//      weights U travel global -> registers -> LDS (buffer loads: rows above / below the image and channels past Cin read
//      as zero through the descriptor's range check; the loads of chunk c+1 are in flight under the work on chunk c);
//   2. every thread transforms (tile, channel) patches 4x4 -> 4x4 (32 adds; the left / right image edge is masked here)
//      and writes V[xy][ci][tile];
//   3. wave w owns the four coordinates xy = (w, 0..3): 4 x 3 x 3 accumulator tiles, fragments at compile-time LDS offsets.
// Epilogue: a wave reduces its row of M to P_w[j] = sum_y A^T[j][y] M[w][y] in registers, the four waves meet in LDS, and
// all threads finish Y[i][j] = sum_x A^T[i][x] P_x[j], apply scale / shift (+ residual) (+ ReLU / GELU) and store float2
// pairs (a tile's two columns) - coalesced along the tile row.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wino_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ uint32_t wdiv(uint32_t i, uint32_t magic) { return magic ? __umulhi(i, magic) : i; }
uint32_t wmagic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }

constexpr int WMB = 3;            // 16-row output-channel blocks per workgroup
constexpr int WMS = 16 * WMB;     // 48: LDS row of the weight slab (== 16 mod 32: the 4 k-rows of a fragment hit distinct banks)
constexpr int WG_ = 4;            // guard floats in front of every channel window (column -1 of the first row)

struct WinoPlan {
    int N, Cin, H, W, HW, Cout, Cout16;
    int in_ctot, in_coff, out_ctot, out_coff, res_ctot, res_coff, act;
    int TX, tpi, bpi, nM;          // tile columns, tiles per image, tile blocks per image, 48-channel tiles
    int L4, WS, JR;                // window float4 per channel, LDS channel stride, 64-float4 pieces per window
    int w_even;                    // W % 2 == 0: a tile's two columns are an aligned float2
    int TXB;                       // 2-D tile blocks: blocks per tile row
    uint32_t magicTX, magicBpi, magicTXB;
};

// (Cout, Cin, 3, 3) -> U[16][Cin][Cout16] = G g G^T, G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int Cout16) {
    const int total = Cin * Cout16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i / Cout16, co = i - ci * Cout16;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + a * 3 + b] : 0.f;
        float t[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                        u3 = t[a][2];
            u[((size_t)(a * 4 + 0) * Cin + ci) * Cout16 + co] = u0;
            u[((size_t)(a * 4 + 1) * Cin + ci) * Cout16 + co] = u1;
            u[((size_t)(a * 4 + 2) * Cin + ci) * Cout16 + co] = u2;
            u[((size_t)(a * 4 + 3) * Cin + ci) * Cout16 + co] = u3;
        }
    }
}

// TB x WCK: 3 x 8 (48 tiles, 8-channel chunks) is the general shape; 2 x 16 (32 tiles, 16-channel chunks: the same 144
// accumulator + 24/48 fragment registers, two patches per thread and chunk either way) fills the tile blocks of the small
// maps better (12x9: 30 tiles per image, 24x18: 108).
// BC > 0: the NT tiles of a workgroup form a (NT / BC) x BC rectangle of the tile grid instead of a row-major run, and the
// staged window is that rectangle's (2 BR + 2) x (2 BC + 2) pixel patch with true zeros outside the image (needs W % 4 == 0):
// on wide maps (96x72: 36 tiles per row) a run of 32 tiles drags 6 full image rows through LDS, 3.4x the input it uses.
template <int TB, int WCK, bool WEVEN, int BC>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const float* __restrict__ in, const float* __restrict__ up,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* res, float* out,
                                                             const WinoPlan P) {
    constexpr int NT = 16 * TB;
    constexpr int VS = (NT % 32 == 16) ? NT : NT + 16;            // == 16 (mod 32)
    constexpr int SLOTS = (WCK * NT + 255) / 256;                 // (tile, channel) patches per thread and chunk
    constexpr int PST = NT + 2;                                   // epilogue row pitch
    constexpr bool TWOD = BC > 0;
    constexpr int BCc = TWOD ? BC : 1, BR = NT / BCc;             // tile rectangle
    constexpr int NR = 2 * BR + 2, GW = (2 * BCc + 2 + 3 + 3) / 4, WC = 4 * GW;   // window rows, float4 per row, row pitch
    constexpr int NI2 = (WCK * NR * GW + 255) / 256;             // float4 window items per thread (2-D)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* raw = smem;                                            // [WCK][WS]
    float* V = smem + WCK * P.WS;                                 // [16][WCK][VS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kl = lane >> 4;

    // ---- block -> (image, first tile, output-channel tile) -----------------------------------------------------------
    // the nM output-channel tiles of one tile block are 8 block ids apart (= the same XCD under round-robin dispatch) and
    // run back to back, so the window is fetched into that L2 once
    const int per = 8 * P.nM;
    const int grp = (int)blockIdx.x / per, rem = (int)blockIdx.x - grp * per;
    const int bp = grp * 8 + (rem & 7), m0 = (rem >> 3) * WMS;
    if (bp >= P.N * P.bpi) return;                                // uniform per workgroup (padding of the last group of 8)
    const int n = (int)wdiv((uint32_t)bp, P.magicBpi);
    const int bi = bp - n * P.bpi;
    // row-major run: tiles t0 .. t0 + nt - 1;  rectangle: tile rows ty0 .. ty0 + BR - 1, tile columns tx0 .. tx0 + BC - 1
    const int t0 = TWOD ? 0 : bi * NT;
    const int nt = TWOD ? NT : min(NT, P.tpi - t0);
    const int by = TWOD ? (int)wdiv((uint32_t)bi, P.magicTXB) : 0;
    const int ty0 = TWOD ? by * BR : (int)wdiv((uint32_t)t0, P.magicTX);
    const int tx0 = TWOD ? (bi - by * P.TXB) * BCc : 0;
    const int f0 = (2 * ty0 - 1) * P.W;                           // first flattened input position the block touches (< 0 at the top)
    const int f0a = f0 & ~3;
    const int x0c = 2 * tx0 - 1, xa = x0c & ~3;                   // 2-D: first window column and its 16-byte aligned start
    // tile j of the block -> (tile row, tile column, exists)
    auto tile_of = [&](int j, int& ty, int& tx) __attribute__((always_inline)) {
        if (TWOD) {
            const int jr = j / BCc, jc = j - jr * BCc;
            ty = ty0 + jr; tx = tx0 + jc;
            return ty * 2 < P.H && tx * 2 < P.W;
        }
        const int t = t0 + (j < nt ? j : 0);
        ty = (int)wdiv((uint32_t)t, P.magicTX); tx = t - ty * P.TX;
        return j < nt;
    };
    const float* img = in + ((size_t)n * P.in_ctot + P.in_coff) * P.HW;

    // ---- staging ---------------------------------------------------------------------------------------------------------
    constexpr int NJI = TWOD ? NI2 : (WCK == 8 ? 6 : 8);
    f32x4 pfi[NJI];
    int goff[TWOD ? NI2 : 1], ldst[TWOD ? NI2 : 1];               // 2-D: byte offset inside the chunk's channels / LDS word
    if (TWOD) {
#pragma unroll
        for (int j = 0; j < NI2; ++j) {
            const int i = tid + 256 * j;
            const int c = i / (NR * GW), rem = i - c * (NR * GW);
            const int r = rem / GW, g = rem - r * GW;
            const int y = 2 * ty0 - 1 + r, xg = xa + 4 * g;
            const bool live = i < WCK * NR * GW;
            goff[j] = (live && y >= 0 && y < P.H && xg >= 0 && xg < P.W) ? ((c * P.H + y) * P.W + xg) * 4 : -1;
            ldst[j] = live ? (c * NR + r) * WC + 4 * g : -1;
        }
    }
    // Weight fragments: wave w multiplies only its own four coordinates, so its A operands never pass through LDS - lane
    // (i16, kl) loads U[4w + nu][c0 + 4k + kl][m0 + 16 mb + i16] straight into the register the MFMA reads (24 dwords per
    // chunk, L2-resident; rows past Cin alias finite values that meet zero inputs, or fall off the tensor -> 0).
    float areg[4][WCK / 4][WMB];
    const otp_rsrc ru = make_rsrc32(up, (unsigned)(16 * P.Cin) * (unsigned)P.Cout16 * 4u);
    const int ubase = (((wave * 4) * P.Cin + kl) * P.Cout16 + m0 + i16) * 4;
    auto load_weights = [&](int c0) __attribute__((always_inline)) {
        const int v = ubase + c0 * P.Cout16 * 4;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int k = 0; k < WCK / 4; ++k) {
                const int so = (nu * P.Cin + 4 * k) * P.Cout16 * 4;
#pragma unroll
                for (int mb = 0; mb < WMB; ++mb)
                    areg[nu][k][mb] = bload(ru, v + mb * 64, so);
            }
    };
    const int vbase = (f0a + 4 * lane) * 4;
    const otp_rsrc rimg = make_rsrc32(img, (unsigned)P.Cin * (unsigned)P.HW * 4u);   // 2-D: channels past Cin fall off the end -> 0
    auto load_chunk = [&](int c0) __attribute__((always_inline)) {
        if (TWOD) {
            const int cb = c0 * P.HW * 4;
#pragma unroll
            for (int j = 0; j < NI2; ++j) pfi[j] = bload4(rimg, goff[j] >= 0 ? goff[j] + cb : -1);
        } else {
            int jc = 0, jr = 0;
#pragma unroll
            for (int j = 0; j < NJI; ++j) {
                if (j < (WCK / 4) * P.JR) {
                    const int c = wave * (WCK / 4) + jc, ch = c0 + c;  // wave w stages WCK / 4 channels of the chunk
                    const bool live = ch < P.Cin;
                    const otp_rsrc r = make_rsrc32(img + (size_t)(live ? ch : 0) * P.HW, live ? (unsigned)P.HW * 4u : 0u);
                    pfi[j] = bload4(r, vbase + jr * 1024);
                    if (++jr == P.JR) { jr = 0; ++jc; }
                }
            }
        }
        asm volatile("" ::: "memory");
    };
    auto store_window = [&]() __attribute__((always_inline)) {
        if (TWOD) {
#pragma unroll
            for (int j = 0; j < NI2; ++j)
                if (ldst[j] >= 0) *reinterpret_cast<f32x4*>(raw + ldst[j]) = pfi[j];
        } else {
            int jc = 0, jr = 0;
#pragma unroll
            for (int j = 0; j < NJI; ++j) {
                if (j < (WCK / 4) * P.JR) {
                    const int c = wave * (WCK / 4) + jc, r4 = lane + 64 * jr;
                    if (r4 < P.L4) *reinterpret_cast<f32x4*>(raw + c * P.WS + WG_ + 4 * r4) = pfi[j];
                    if (++jr == P.JR) { jr = 0; ++jc; }
                }
            }
        }
    };
    // ---- per-thread patch geometry (the same for every chunk) -----------------------------------------------------------
    // A thread transforms SLOTS patches per chunk side by side (packed f32 math: one v_pk_* per pair of patches).  Edge
    // columns and dead patches are multiplied by 0 (the guard floats a masked read can touch are zeroed below, everything
    // else in the window is image data or range-check zeros, so 0 * x is exact).
    static_assert(SLOTS == 2, "the packed transform pairs two patches per thread");
    int toff[SLOTS], vdst[SLOTS];
    f32x2 cm[4];                                                  // cm[jj] = (slot 0, slot 1) keep factors of patch column jj
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = tid + 256 * s;
        const int c = i / NT, j = i - c * NT;
        const bool live = i < WCK * NT;
        int ty, tx;
        const bool ok = tile_of(j, ty, tx);
        const int x0 = 2 * tx - 1;
        if (TWOD) {
            // window rows / columns outside the image hold zeros (range-checked loads), so no column masks: a tile that does
            // not exist reads zeros only and contributes V = 0
            const int jr = j / BCc, jc = j - jr * BCc;
            toff[s] = ((live ? c : 0) * NR + 2 * jr) * WC + 2 * jc + (x0c - xa);
        } else {
            toff[s] = (live ? c : 0) * P.WS + WG_ + (f0 - f0a) + 2 * (ty - ty0) * P.W + x0;
        }
        vdst[s] = live ? c * VS + j : -1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) cm[jj][s] = (live && ok && x0 + jj >= 0 && x0 + jj < P.W) ? 1.f : 0.f;
    }
    if (!TWOD)
        for (int c = tid; c < WCK * WG_; c += 256) raw[(c / WG_) * P.WS + (c % WG_)] = 0.f;   // guard floats

    f32x4 acc[4][WMB][TB];

    const float* Vw = V + (wave * 4) * WCK * VS + kl * VS + i16;

    load_chunk(0);
    load_weights(0);
    store_window();
    __syncthreads();
    for (int c0 = 0; c0 < P.Cin; c0 += WCK) {
        const bool more = c0 + WCK < P.Cin;
        // ---- input transform: V = B^T d B, B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] -------------------------
        {
            const int RS_ = TWOD ? WC : P.W;                       // row pitch of the staged window
            const float* s0 = raw + toff[0];
            const float* s1 = raw + toff[1];
            f32x2 t[4][4];
            if (WEVEN) {
                // even W: patch columns 1, 2 of every row are an 8-byte aligned pair -> 3 LDS reads per row instead of 4, and
                // the paired read is conflict-free (the single reads step 2 floats per lane: 2-way)
                {                                                  // columns 1 and 2 together
                    f32x2 e[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x2 m0 = *reinterpret_cast<const f32x2*>(s0 + i * RS_ + 1);
                        const f32x2 m1 = *reinterpret_cast<const f32x2*>(s1 + i * RS_ + 1);
                        e[i][0] = (f32x2){m0[0], m1[0]} * (TWOD ? (f32x2){1.f, 1.f} : cm[1]);
                        e[i][1] = (f32x2){m0[1], m1[1]} * (TWOD ? (f32x2){1.f, 1.f} : cm[2]);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        t[0][1 + h] = e[0][h] - e[2][h];
                        t[1][1 + h] = e[1][h] + e[2][h];
                        t[2][1 + h] = e[2][h] - e[1][h];
                        t[3][1 + h] = e[1][h] - e[3][h];
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 4; jj += 3) {               // columns 0 and 3
                    const f32x2 d0 = (f32x2){s0[jj], s1[jj]} * (TWOD ? (f32x2){1.f, 1.f} : cm[jj]);
                    const f32x2 d1 = (f32x2){s0[RS_ + jj], s1[RS_ + jj]} * (TWOD ? (f32x2){1.f, 1.f} : cm[jj]);
                    const f32x2 d2 = (f32x2){s0[2 * RS_ + jj], s1[2 * RS_ + jj]} * (TWOD ? (f32x2){1.f, 1.f} : cm[jj]);
                    const f32x2 d3 = (f32x2){s0[3 * RS_ + jj], s1[3 * RS_ + jj]} * (TWOD ? (f32x2){1.f, 1.f} : cm[jj]);
                    t[0][jj] = d0 - d2;
                    t[1][jj] = d1 + d2;
                    t[2][jj] = d2 - d1;
                    t[3][jj] = d1 - d3;
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {                   // one patch column at a time: 4 loads -> 4 results
                    const f32x2 d0 = (f32x2){s0[jj], s1[jj]} * (TWOD ? (f32x2){1.f, 1.f} : cm[jj]);
                    const f32x2 d1 = (f32x2){s0[RS_ + jj], s1[RS_ + jj]} * (TWOD ? (f32x2){1.f, 1.f} : cm[jj]);
                    const f32x2 d2 = (f32x2){s0[2 * RS_ + jj], s1[2 * RS_ + jj]} * (TWOD ? (f32x2){1.f, 1.f} : cm[jj]);
                    const f32x2 d3 = (f32x2){s0[3 * RS_ + jj], s1[3 * RS_ + jj]} * (TWOD ? (f32x2){1.f, 1.f} : cm[jj]);
                    t[0][jj] = d0 - d2;
                    t[1][jj] = d1 + d2;
                    t[2][jj] = d2 - d1;
                    t[3][jj] = d1 - d3;
                }
            }
            float* q0 = V + vdst[0];
            float* q1 = V + (vdst[1] >= 0 ? vdst[1] : 0);
            const bool w1 = vdst[1] >= 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 v0 = t[i][0] - t[i][2], v1 = t[i][1] + t[i][2], v2 = t[i][2] - t[i][1], v3 = t[i][1] - t[i][3];
                q0[(i * 4 + 0) * WCK * VS] = v0[0]; q0[(i * 4 + 1) * WCK * VS] = v1[0];
                q0[(i * 4 + 2) * WCK * VS] = v2[0]; q0[(i * 4 + 3) * WCK * VS] = v3[0];
                if (w1) {
                    q1[(i * 4 + 0) * WCK * VS] = v0[1]; q1[(i * 4 + 1) * WCK * VS] = v1[1];
                    q1[(i * 4 + 2) * WCK * VS] = v2[1]; q1[(i * 4 + 3) * WCK * VS] = v3[1];
                }
            }
        }
        if (more) load_chunk(c0 + WCK);                           // window of the next chunk: in flight under the MFMAs
        __syncthreads();                                          // V complete (U was complete at the previous barrier)
        // ---- 4 coordinates x 2 k-steps x (WMB x TB) MFMAs ------------------------------------------------------------------
        // (the very first MFMA of every accumulator takes a literal zero as C: no 144-register clear per workgroup)
        auto mfma_phase = [&](auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                for (int k = 0; k < WCK / 4; ++k) {
                    float a[WMB], b[TB];
#pragma unroll
                    for (int mb = 0; mb < WMB; ++mb) a[mb] = areg[nu][k][mb];
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) b[tb] = Vw[(nu * WCK + 4 * k) * VS + tb * 16];
#pragma unroll
                    for (int mb = 0; mb < WMB; ++mb)
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb) {
                            const f32x4 cin = (FIRST && k == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[nu][mb][tb];
                            acc[nu][mb][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[tb], cin, 0, 0, 0);
                        }
                }
        };
        if (c0 == 0) mfma_phase(std::true_type{});
        else mfma_phase(std::false_type{});
        if (more) {
            load_weights(c0 + WCK);                               // the fragments of this chunk are spent; in flight under the next transform
            store_window();                                       // raw is free since the barrier above
        }
        __syncthreads();                                          // V free, raw of the next chunk complete
    }

    // ---- output transform + epilogue, one 16-channel block at a time -----------------------------------------------------
    // thread = (channel row = tid / 16, 16 consecutive tiles = tid % 16 + 16 * tb): stores of a wave are 128-byte runs
    float* Pb = smem;                                             // [4 waves][2][16][PST] (aliases the window and V: both are dead)
    const int HoWo = P.HW;
    const int erow = tid >> 4, el = tid & 15;
    int eoff[TB];                                                 // pixel offset of the tile's top-left output, -1: no such tile
    bool etwo[TB], ebot[TB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
        const int j = tb * 16 + el;
        int ty, tx;
        const bool ok = tile_of(j, ty, tx);
        eoff[tb] = ok ? 2 * ty * P.W + 2 * tx : -1;
        etwo[tb] = 2 * tx + 1 < P.W;
        ebot[tb] = 2 * ty + 1 < P.H;
    }
#pragma unroll
    for (int mb = 0; mb < WMB; ++mb) {                                // fully unrolled: static accumulator indices, no selects
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0v = acc[0][mb][tb][r], m1v = acc[1][mb][tb][r], m2v = acc[2][mb][tb][r], m3v = acc[3][mb][tb][r];
                Pb[((wave * 2 + 0) * 16 + kl * 4 + r) * PST + tb * 16 + i16] = m0v + m1v + m2v;
                Pb[((wave * 2 + 1) * 16 + kl * 4 + r) * PST + tb * 16 + i16] = m1v - m2v - m3v;
            }
        __syncthreads();
        const int co = m0 + mb * 16 + erow;
        if (co < P.Cout) {
            const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
            float* orow = out + ((size_t)n * P.out_ctot + P.out_coff + co) * HoWo;
            const float* rrow = res ? res + ((size_t)n * P.res_ctot + P.res_coff + co) * HoWo : nullptr;
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                if (eoff[tb] < 0) continue;
                const float* pr = Pb + erow * PST + tb * 16 + el;
                const float p00 = pr[0 * 16 * PST], p01 = pr[1 * 16 * PST], p10 = pr[2 * 16 * PST], p11 = pr[3 * 16 * PST],
                            p20 = pr[4 * 16 * PST], p21 = pr[5 * 16 * PST], p30 = pr[6 * 16 * PST], p31 = pr[7 * 16 * PST];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i == 1 && !ebot[tb]) continue;
                    float y0 = i == 0 ? p00 + p10 + p20 : p10 - p20 - p30;
                    float y1 = i == 0 ? p01 + p11 + p21 : p11 - p21 - p31;
                    y0 = fmaf(y0, sc, sh);
                    y1 = fmaf(y1, sc, sh);
                    const int o = eoff[tb] + i * P.W;
                    const bool two = etwo[tb];
                    if (rrow) {
                        if (two && WEVEN) {
                            const f32x2 rv = *reinterpret_cast<const f32x2*>(rrow + o);
                            y0 += rv.x; y1 += rv.y;
                        } else {
                            y0 += rrow[o];
                            if (two) y1 += rrow[o + 1];
                        }
                    }
                    if (P.act == OTP_ACT_RELU) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
                    else if (P.act == OTP_ACT_GELU) { y0 = wino_gelu(y0); y1 = wino_gelu(y1); }
                    if (two && WEVEN) {
                        *reinterpret_cast<f32x2*>(orow + o) = (f32x2){y0, y1};
                    } else {
                        orow[o] = y0;
                        if (two) orow[o + 1] = y1;
                    }
                }
            }
        }
        __syncthreads();
    }
}

bool wino_plan(const otp_conv_desc& d, WinoPlan& P, size_t& lds, int NT, int WCK) {
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1 || d.res_up > 1 || d.frame_split > 0) return false;
    if (d.Ho != d.H || d.Wo != d.W || ((d.H * d.W) & 3)) return false;
    P.N = d.N; P.Cin = d.Cin; P.H = d.H; P.W = d.W; P.HW = d.H * d.W; P.Cout = d.Cout; P.Cout16 = (d.Cout + 15) & ~15;
    P.in_ctot = d.in_ctot; P.in_coff = d.in_coff; P.out_ctot = d.out_ctot; P.out_coff = d.out_coff;
    P.res_ctot = d.res_ctot; P.res_coff = d.res_coff; P.act = d.act;
    P.TX = (d.W + 1) / 2;
    const int TY = (d.H + 1) / 2;
    P.tpi = P.TX * TY;
    P.bpi = (P.tpi + NT - 1) / NT;
    P.nM = (P.Cout16 + WMS - 1) / WMS;
    int span = (NT + P.TX - 2) / P.TX + 1;                        // tile rows NT consecutive tiles can touch
    if (span > TY) span = TY;
    const int rows_in = 2 * span + 2;
    P.L4 = (rows_in * d.W + 3 + 3) / 4 + 1;
    P.JR = (P.L4 + 63) / 64;
    P.WS = WG_ + 4 * P.L4 + 4;
    if ((P.WS & 31) == 0) P.WS += 4;
    P.w_even = (d.W & 1) == 0 ? 1 : 0;
    P.magicTX = wmagic(P.TX);
    P.magicBpi = wmagic(P.bpi);
    const int VS = (NT % 32 == 16) ? NT : NT + 16;
    lds = ((size_t)WCK * P.WS + (size_t)16 * WCK * VS) * sizeof(float);
    const size_t ep = (size_t)4 * 2 * 16 * (NT + 2) * sizeof(float);      // epilogue tiles alias the window + V
    if (ep > lds) lds = ep;
    if ((long)16 * d.Cin * P.Cout16 * 4 >= (1l << 31) || (long)P.HW * 4 >= (1l << 30)) return false;
    if ((WCK / 4) * P.JR > (WCK == 8 ? 6 : 8)) return false;      // window too long for the staging registers
    return lds <= 80 * 1024;
}

// 2-D tile blocks (BR x BC tiles per workgroup): only the block geometry and the LDS window differ from wino_plan
bool wino_plan2d(const otp_conv_desc& d, WinoPlan& P, size_t& lds, int NT, int WCK, int BC) {
    size_t dummy = 0;
    (void)wino_plan(d, P, dummy, NT, WCK);                         // fills the shape-independent fields (may itself not fit)
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1 || d.res_up > 1 || d.frame_split > 0) return false;
    if (d.Ho != d.H || d.Wo != d.W || (d.W & 3)) return false;
    const int BR = NT / BC, TY = (d.H + 1) / 2;
    P.TXB = (P.TX + BC - 1) / BC;
    P.bpi = P.TXB * ((TY + BR - 1) / BR);
    const int NR = 2 * BR + 2, GW = (2 * BC + 2 + 3 + 3) / 4;
    P.WS = NR * 4 * GW;
    P.magicTXB = wmagic(P.TXB);
    P.magicBpi = wmagic(P.bpi);
    const int VS = (NT % 32 == 16) ? NT : NT + 16;
    lds = ((size_t)WCK * P.WS + (size_t)16 * WCK * VS) * sizeof(float);
    const size_t ep = (size_t)4 * 2 * 16 * (NT + 2) * sizeof(float);
    if (ep > lds) lds = ep;
    if ((long)d.Cin * P.HW * 4 >= (1l << 31)) return false;
    return lds <= 80 * 1024;
}

// variant 0 = 48 tiles x 8-channel chunks, 1 = 32 tiles x 16-channel chunks.  Measured (tools/conv_bench.py --wino with
// OTP_WINO_VARIANT=0/1): the second is faster wherever its tile blocks are at least as full (48->48 @96x72 0.219 -> 0.200 ms,
// 192->192 @24x18 0.201 -> 0.175, 384->384 @12x9 0.282 -> 0.203, 256->48 @96x72 0.941 -> 0.779); 96->96 @48x36, whose 432
// tiles per image split into 9 full 48-tile blocks but 13.5 32-tile ones, prefers the first (0.164 vs 0.178).
int wino_choose(const otp_conv_desc& d, WinoPlan& P, size_t& lds) {
    WinoPlan A{}, B{};
    size_t la = 0, lb = 0;
    const bool oka = wino_plan(d, A, la, 48, 8), okb = wino_plan(d, B, lb, 32, 16);
    if (!oka && !okb) return -1;
    const double fa = oka ? (double)A.tpi / (A.bpi * 48.0) : 0.0, fb = okb ? (double)B.tpi / (B.bpi * 32.0) : 0.0;
    const char* force = getenv("OTP_WINO_VARIANT");
    if (force && force[0] == '1' && okb) { P = B; lds = lb; return 1; }
    if (force && force[0] == '0' && oka) { P = A; lds = la; return 0; }
    if (okb && (!oka || fb >= fa)) { P = B; lds = lb; return 1; }
    P = A; lds = la;
    return 0;
}

template <int TB, int WCK, int BC>
int wino_launch(const float* in, const float* up, const float* scale, const float* shift, const float* res, float* out,
                const WinoPlan& P, size_t lds, hipStream_t st) {
    const dim3 grid(((P.N * P.bpi + 7) / 8) * 8 * P.nM);
    if (P.w_even) {
        auto kern = conv_wino_kernel<TB, WCK, true, BC>;
        OTP_ALLOW_BIG_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, in, up, scale, shift, res, out, P);
    } else {
        auto kern = conv_wino_kernel<TB, WCK, false, BC>;
        OTP_ALLOW_BIG_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, in, up, scale, shift, res, out, P);
    }
    return otp_launch_status();
}

int g_wino_last[4] = {0, 0, 0, 0};      // tuning hook: {tile blocks per workgroup, chunk channels, grid, lds bytes} of the last launch

}  // namespace

extern "C" int otp_conv2d_wino_last_plan(int* out4) {
    if (!out4) return OTP_ERR_BAD_ARG;
    for (int i = 0; i < 4; ++i) out4[i] = g_wino_last[i];
    return OTP_OK;
}

extern "C" size_t otp_conv2d_wino_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)16 * Cin * ((Cout + 15) & ~15) * sizeof(float);
}

extern "C" int otp_conv2d_wino_pack_weight(const void* weight, void* upacked, int Cout, int Cin, void* stream) {
    if (!weight || !upacked || Cout <= 0 || Cin <= 0) return OTP_ERR_BAD_ARG;
    const int Cout16 = (Cout + 15) & ~15, total = Cin * Cout16;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(otp_ceil_div(total, 256) > 1024 ? 1024 : otp_ceil_div(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(weight), static_cast<float*>(upacked), Cout,
                       Cin, Cout16);
    return otp_launch_status();
}

extern "C" int otp_conv2d_wino_supported(const otp_conv_desc* desc) {
    if (!desc) return 0;
    WinoPlan P{};
    size_t lds = 0;
    return wino_choose(*desc, P, lds) >= 0 ? 1 : 0;
}

extern "C" int otp_conv2d_wino(const void* in, const void* upacked, const void* scale, const void* shift, const void* res,
                               void* out, const otp_conv_desc* desc, void* stream) {
    if (!in || !upacked || !out || !desc) return OTP_ERR_BAD_ARG;
    const otp_conv_desc& d = *desc;
    if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0) return OTP_ERR_BAD_ARG;
    if (d.in_ctot < d.in_coff + d.Cin || d.out_ctot < d.out_coff + d.Cout || (res && d.res_ctot < d.res_coff + d.Cout))
        return OTP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(upacked)) & 15) return OTP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(out) & 7) || (res && (reinterpret_cast<uintptr_t>(res) & 7))) return OTP_ERR_UNSUPPORTED;
    WinoPlan P{};
    size_t lds = 0;
    const int variant = wino_choose(d, P, lds);
    if (variant < 0) return OTP_ERR_UNSUPPORTED;
    auto st = static_cast<hipStream_t>(stream);
    g_wino_last[0] = variant == 1 ? 2 : 3; g_wino_last[1] = variant == 1 ? 16 : 8;
    g_wino_last[2] = ((P.N * P.bpi + 7) / 8) * 8 * P.nM; g_wino_last[3] = (int)lds;
    auto f = [](const void* p) { return static_cast<const float*>(p); };
    {
        // wide maps whose tile grid splits into full 4 x 12 rectangles (96x72: 48 x 36 tiles): 2-D tile blocks, 8-channel chunks.
        // Measured against the row-major shapes: 48->48 0.196 -> 0.180 ms, 256->48 0.766 -> 0.692, 64->64 0.460 -> 0.413;
        // 96->96 @48x36 (18 tile columns) 0.164 -> 0.192, so narrower maps keep the row-major run.
        WinoPlan Q{};
        size_t l2 = 0;
        const char* e2 = getenv("OTP_WINO_2D");
        const bool allow2d = !e2 || e2[0] != '0';
        if (allow2d && !(d.W >= 64 && ((d.W + 1) / 2) % 12 == 0 && ((d.H + 1) / 2) % 4 == 0) && ((d.W + 1) / 2) % 6 == 0 &&
            ((d.H + 1) / 2) % 8 == 0 && wino_plan2d(d, Q, l2, 48, 8, 6)) {
            // 8 x 6 rectangles (48x36 maps: 24 x 18 tiles): same window volume as the row-major run but no edge masks and
            // 22 fewer registers (no spills): 96->96 0.164 -> 0.150 ms
            g_wino_last[0] = 3; g_wino_last[1] = 8; g_wino_last[2] = ((Q.N * Q.bpi + 7) / 8) * 8 * Q.nM; g_wino_last[3] = (int)l2;
            return wino_launch<3, 8, 6>(f(in), f(upacked), f(scale), f(shift), f(res), static_cast<float*>(out), Q, l2, st);
        }
        if (allow2d && d.W >= 64 && ((d.W + 1) / 2) % 12 == 0 && ((d.H + 1) / 2) % 4 == 0 && wino_plan2d(d, Q, l2, 48, 8, 12)) {
            g_wino_last[0] = 3; g_wino_last[1] = 8; g_wino_last[2] = ((Q.N * Q.bpi + 7) / 8) * 8 * Q.nM; g_wino_last[3] = (int)l2;
            return wino_launch<3, 8, 12>(f(in), f(upacked), f(scale), f(shift), f(res), static_cast<float*>(out), Q, l2, st);
        }
    }
    return variant == 1 ? wino_launch<2, 16, 0>(f(in), f(upacked), f(scale), f(shift), f(res), static_cast<float*>(out), P, lds, st)
                        : wino_launch<3, 8, 0>(f(in), f(upacked), f(scale), f(shift), f(res), static_cast<float*>(out), P, lds, st);
}
