// 3x3 / STRIDE 2 / pad 1 convolutions of HRNet (reference model/HRNet.py:442-470 fuse-layer down-sampling chains, :192-231
// transition layers, :66-72 stem conv2) on activations in the split record (S8) format of csrc/convs.hip, fed by the LDS-DMA.
//
// Same arithmetic (split bf16x3 products, fp32 accumulation), same packed weights (otp_conv3x3_s8_pack_weight) and the same
// MFMA loop as convs_kernel; what differs is the window.  Output pixel (y, x) reads input pixels (2 y + dy - 1, 2 x + dx - 1), so
// the records of 16 consecutive output pixels of one tap are 32 bytes apart in an input row - a 2-way bank conflict on every
// fragment read.  The LDS-DMA writes lane-linear but READS a per-lane source address, so the window is staged with its columns
// DE-INTERLEAVED BY PARITY:
//
//     virtual row (one per input row, plus one zero row above every image), W + 1 records:
//         [ 0 | O_0 O_1 .. O_{Wo-1} | E_0 E_1 .. E_{Wo-1} ]        O_k = input column 2 k + 1, E_k = input column 2 k
//     tap dx = 0 (column 2 x - 1) reads O_{x-1} = record x, dx = 1 (column 2 x) reads E_x = record Wo + 1 + x,
//     dx = 2 (column 2 x + 1) reads O_x = record 1 + x      - consecutive output pixels read consecutive records.
//
// (round 3 ran these layers on convx_kernel<8, 2, ..> from fp32 NCHW input with the split in the consumer: 145 - 180 algorithmic
// TFLOP/s against 356 for the S8 kernel; the largest remaining family of the forward - VERDICT r03 "missing" item 3.)
// Epilogue: S8 records straight from the accumulators (chain intermediates: conv + BN + ReLU, model/HRNet.py:452-461), or the
// workgroup's [channels][pixels] tile through the LDS and out as 16-byte stores into a channel slice of an fp32 NCHW tensor,
// with an NCHW residual added on the way (a fuse row accumulates its terms in place: HRNet.py:488-494) and ReLU after it.
#include "common.h"
#include <cstdlib>

namespace {

typedef otp_x3x8 h16x8;              // 8 operand pieces of the split products (common.h: IEEE half since round 4)
typedef otp_x3x2 h16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---- shared conventions of csrc/convs.hip (weight image, row permutation of a cout-tile pair) -----------------------------------
__host__ __device__ constexpr int swch(int ntw) { return 8 * ntw + ntw; }     // 1 KB pieces of a chunk's packed weights
constexpr int SKS = 5;                    // k-steps per 16-channel chunk: 18 (tap, group) slots of 8 channels in 5 x 4
constexpr int SOOB = -16;                 // buffer offset outside every descriptor: the load returns / writes zeros
constexpr int MAXJ = 4;                   // 64-record window pieces per wave and plane (window planes of at most 1024 records)

__device__ __forceinline__ uint32_t sdiv(uint32_t i, uint32_t magic) { return magic ? __umulhi(i, magic) : i; }
uint32_t smagic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }   // exact while i * d < 2^32

__device__ __forceinline__ void ssplit8(const float (&v)[8], u32x4& hi, u32x4& lo) {
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

__host__ __device__ inline bool stile_paired(int co_blk, int t, int ntw, int Cout) {
    const int tb = t | 1;
    return tb < ntw && co_blk + 16 * tb < Cout;
}
__host__ __device__ inline int srow2ch(int co_blk, int t, int row, int ntw, int Cout) {
    return stile_paired(co_blk, t, ntw, Cout) ? co_blk + 32 * (t >> 1) + 8 * (row >> 2) + 4 * (t & 1) + (row & 3)
                                               : co_blk + 16 * t + row;
}

template <int NM, int NR>
__device__ __forceinline__ void sblock_sched() {
    if constexpr (NR == 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    } else if constexpr (NR >= NM - 1) {
#pragma unroll
        for (int g = 0; g < NM - 1; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (NR > NM - 1) __builtin_amdgcn_sched_group_barrier(0x100, NR - (NM - 1), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM / 2 - 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR - 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1 - NM / 2, 0);
    }
}

struct S2Plan {
    int N, C, H, W, HW, Ho, Wo, HWo, Cout, total;      // total = N * Ho * Wo output pixels
    int out_ctot, out_coff, res_ctot, res_coff, act;
    float pre, post;                                   // weights carry 2^k = pre; the sum is multiplied by post = 2^-k (out_scale)
    int NTW, nN, nTiles, nChunks, tpx, NPT;
    int VR, W1, NIW, NV, pl;                           // virtual rows per image (H + 1), records per virtual row (W + 1), 64-record
                                                       // pieces per plane, records / bytes of a window plane
    uint32_t mHWo, mWo, mW1, mVR;
    unsigned* rflag;                                   // range-guard word (common.h)
};

// NCHW: fp32 result (+ residual) into a channel slice of an NCHW tensor; otherwise the S8 image of the result only
template <int NTW, bool NCHW, int NPT>
__global__ __launch_bounds__(256, 2) void convs2_kernel(const unsigned char* __restrict__ xs, const u32x4* __restrict__ wpk,
                                                         const float* __restrict__ shift, const float* res, float* outf,
                                                         u32x4* outs, const S2Plan P) {
    constexpr int BM = 64 * NPT;
    constexpr int WCH = swch(NTW);
    constexpr int WBYTES = WCH * 1024;
    constexpr int NBLK = SKS * NPT;
    constexpr int NM = 3 * NTW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PL = P.pl;
    unsigned char* win = smem;                                     // 4 planes: (group 0, hi), (group 0, lo), (group 1, hi), (group 1, lo)
    unsigned char* wl = smem + 4 * PL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kl = lane >> 4;
    const bool upper = kl >= 2;

    const int xcd = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
    const int tl_ = jb / P.nN, cb = jb - tl_ * P.nN;
    const int tile = xcd * P.tpx + tl_;
    if (tile >= P.nTiles) return;
    const int P0 = tile * BM;
    const int n0 = P0 / P.HWo, p0 = P0 - n0 * P.HWo;              // (uniform, once per workgroup)
    const int y0 = (int)sdiv((uint32_t)p0, P.mWo);
    const int Vf = n0 * P.VR + 2 * y0;                             // first virtual row of the window: input row 2 y0 - 1
    const int imgB = P.C * P.HW * 4;                               // bytes of one image of the S8 input
    const int co_blk = cb * NTW * 16;

    // ---- window pieces of this wave: piece k = wave + 4 j covers window records 64 k .. 64 k + 63 of every plane ------------------
    int voff[MAXJ];
    bool vlive[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int v = 64 * (wave + 4 * j) + lane;
        vlive[j] = v < P.NV;
        const int r = (int)sdiv((uint32_t)v, P.mW1), i = v - r * P.W1;
        const int V = Vf + r;
        const int n = (int)sdiv((uint32_t)V, P.mVR), yy = V - n * P.VR;
        const int col = i <= P.Wo ? 2 * (i - 1) + 1 : 2 * (i - P.Wo - 1);      // odd columns first, then the even ones
        const bool ok = i >= 1 && yy >= 1 && n < P.N;
        voff[j] = ok ? (n - n0) * imgB + ((yy - 1) * P.W + col) * 16 : SOOB;
    }
    const size_t left = (size_t)(P.N - n0) * imgB;
    const otp_rsrc rin = make_rsrc32(xs + (size_t)n0 * imgB, left > 0x7fffff00ull ? 0x7fffff00u : (unsigned)left);
    const otp_rsrc rw = make_rsrc32(wpk, (unsigned)((size_t)P.nN * P.nChunks * WBYTES));
    const int woff = lane * 16;

    auto stage = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            const int so = (((2 * c + (pl >> 1)) * 2 + (pl & 1)) * P.HW) * 16;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int k = wave + 4 * j;
                if (k < P.NIW && vlive[j])
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(win + pl * PL + k * 1024),
                                                             16, voff[j], so, 0, 0);
            }
        }
        const int wb = (cb * P.nChunks + c) * WBYTES;
#pragma unroll
        for (int j = 0; j < (WCH + 3) / 4; ++j) {
            const int k = wave + 4 * j;
            if (k < WCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(wl + k * 1024), 16, woff,
                                                         wb + k * 1024, 0, 0);
        }
    };
    stage(0);

    const size_t obytes = (size_t)P.N * P.Cout * P.HWo * 4;        // S8 image of the (N, Cout, Ho, Wo) result
    const otp_rsrc rs8 = make_rsrc32(outs, outs ? (unsigned)obytes : 0u);
    const otp_rsrc rsh = make_rsrc32(shift ? shift : reinterpret_cast<const float*>(xs), shift ? (unsigned)(P.Cout * 4) : 0u);
    int pb[NPT], toff[SKS], offS[NPT], ch0[NTW];
    f32x4 acc[NTW][NPT];
    {
#pragma unroll
        for (int s = 0; s < SKS; ++s) {
            const int q = 4 * s + kl;
            int tap = q >> 1;
            if (tap > 8) tap = 8;                                  // zero weights: any finite data
            const int dy = tap / 3, dx = tap - dy * 3;
            const int rx = dx == 0 ? 0 : (dx == 1 ? P.Wo + 1 : 1); // record of the tap relative to record x of its virtual row
            toff[s] = (dy * P.W1 + rx) * 16 + (q & 1) * (2 * PL);
        }
        f32x4 sh[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            ch0[t] = srow2ch(co_blk, t, 4 * kl, NTW, P.Cout);
            sh[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsh, co_blk + 16 * t < P.Cout ? ch0[t] * 4 : SOOB, 0, 0));
        }
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
            int m = (wave * NPT + p) * 16 + i16;
            const bool pv = P0 + m < P.total;
            if (!pv) m = P.total - 1 - P0;                         // tail tile: a finite address, the result is dropped
            const int q = p0 + m;
            const int dn = (int)sdiv((uint32_t)q, P.mHWo), pi = q - dn * P.HWo;
            const int y = (int)sdiv((uint32_t)pi, P.mWo), x = pi - y * P.Wo;
            pb[p] = (((n0 + dn) * P.VR + 2 * y - Vf) * P.W1 + x) * 16;    // record x of the virtual row of tap dy = 0
            offS[p] = pv ? ((n0 + dn) * (P.Cout >> 2) * P.HWo + pi) * 16 : SOOB;   // S8 image: (((img Go + ch / 8) 2 + part) HWo + pi) 16
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t][p] = sh[t] * P.pre;
        }
    }

    auto mfma_phase = [&]() __attribute__((always_inline)) {
        h16x8 ah[2][NTW], al[2][NTW], bh[3], bl[3];
        auto load_a = [&](int buf, int s) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (s < SKS - 1) {
                    const unsigned char* a = wl + ((s * NTW + t) * 2) * 1024 + lane * 16;
                    ah[buf][t] = *reinterpret_cast<const h16x8*>(a);
                    al[buf][t] = *reinterpret_cast<const h16x8*>(a + 1024);
                } else {
                    const unsigned char* a = wl + (SKS - 1) * NTW * 2048 + t * 1024 + (lane & 31) * 16;
                    const h16x8 h = *reinterpret_cast<const h16x8*>(a), l = *reinterpret_cast<const h16x8*>(a + 512);
                    const h16x8 z = __builtin_bit_cast(h16x8, (u32x4){0u, 0u, 0u, 0u});
                    ah[buf][t] = upper ? z : h;
                    al[buf][t] = upper ? z : l;
                }
            }
        };
        auto load_b = [&](int buf, int blk) __attribute__((always_inline)) {
            const unsigned char* b = win + (pb[blk % NPT] + toff[blk / NPT]);
            bh[buf] = *reinterpret_cast<const h16x8*>(b);
            bl[buf] = *reinterpret_cast<const h16x8*>(b + PL);
        };
        load_a(0, 0);
        load_b(0, 0);
        if (NBLK > 1) load_b(1, 1);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const int s = blk / NPT, p = blk % NPT, cur = blk % 3, sa = s & 1;
            const bool nb = blk + 2 < NBLK;
            const bool na = (NPT >= 2 ? p == NPT - 2 : true) && s + 1 < SKS;
            if (nb) load_b((blk + 2) % 3, blk + 2);
            if (na) load_a(sa ^ 1, s + 1);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                acc[t][p] = OTP_X3_MFMA(al[sa][t], bh[cur], acc[t][p], 0, 0, 0);
                acc[t][p] = OTP_X3_MFMA(ah[sa][t], bl[cur], acc[t][p], 0, 0, 0);
                acc[t][p] = OTP_X3_MFMA(ah[sa][t], bh[cur], acc[t][p], 0, 0, 0);
            }
            if (!nb && !na) sblock_sched<NM, 0>();
            else if (nb && na) sblock_sched<NM, 2 + 2 * NTW>();
            else if (na) sblock_sched<NM, 2 * NTW>();
            else sblock_sched<NM, 2>();
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (int c = 0; c < P.nChunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of chunk c have landed
        __syncthreads();                                           // ... and everybody else's
        mfma_phase();
        if (c + 1 < P.nChunks) {
            __syncthreads();                                       // every wave is done with the LDS image of chunk c
            stage(c + 1);
        }
    }

    if (P.post != 1.f) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < NPT; ++p) acc[t][p] = acc[t][p] * P.post;
    }
    if (!NCHW) {
        // ---- S8 records of act(result) straight from the accumulators (no residual in this form) -------------------------------
        {   // range guard (common.h), before the ReLU that would swallow a NaN
            bool bad = false;
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int p = 0; p < NPT; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bad |= otp_out_of_range(acc[t][p][r]);
            otp_range_report(P.rflag, bad, OTP_RANGE_CONVS2);
        }
        if (P.act == OTP_ACT_RELU) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int p = 0; p < NPT; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][p][r] = otp_relu(acc[t][p][r]);
        }
        u32x4 rec[(NTW + 1) / 2][NPT][2];
#pragma unroll
        for (int t = 0; t < NTW; t += 2) {
            const bool paired = stile_paired(co_blk, t, NTW, P.Cout);
            const int t1 = t + 1 < NTW ? t + 1 : t;
#pragma unroll
            for (int p = 0; p < NPT; ++p) {
                const float f[8] = {acc[t][p][0], acc[t][p][1], acc[t][p][2], acc[t][p][3],
                                    paired ? acc[t1][p][0] : 0.f, paired ? acc[t1][p][1] : 0.f,
                                    paired ? acc[t1][p][2] : 0.f, paired ? acc[t1][p][3] : 0.f};
                ssplit8(f, rec[t >> 1][p][0], rec[t >> 1][p][1]);
            }
        }
#pragma unroll
        for (int t = 0; t < NTW; t += 2) {
            const bool tav = co_blk + 16 * t < P.Cout;
            const int so = (ch0[t] >> 3) * 2 * P.HWo * 16;
            if (stile_paired(co_blk, t, NTW, P.Cout)) {
#pragma unroll
                for (int p = 0; p < NPT; ++p) {
                    const int o = offS[p] != SOOB ? offS[p] + so : SOOB;
                    __builtin_amdgcn_raw_buffer_store_b128(rec[t >> 1][p][0], rs8, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(rec[t >> 1][p][1], rs8, o, P.HWo * 16, 0);
                }
            } else {
                const int half = (ch0[t] >> 2) & 1;
#pragma unroll
                for (int p = 0; p < NPT; ++p) {
                    const int o = (tav && offS[p] != SOOB) ? offS[p] + so + 8 * half : SOOB;
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){rec[t >> 1][p][0][0], rec[t >> 1][p][0][1]}, rs8, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){rec[t >> 1][p][1][0], rec[t >> 1][p][1][1]}, rs8, o, P.HWo * 16, 0);
                }
            }
        }
        return;
    }

    // ---- fp32 NCHW slice: the [16 NTW channels][BM pixels] tile through the LDS, out as 16-byte stores (1 KB of a channel row
    //      per wave instruction), residual added and ReLU applied on the way ------------------------------------------------------
    const otp_rsrc rof = make_rsrc32(outf, (unsigned)((size_t)P.N * P.out_ctot * P.HWo * 4));
    const otp_rsrc rres = make_rsrc32(res ? res : reinterpret_cast<const float*>(xs),
                                      res ? (unsigned)((size_t)P.N * P.res_ctot * P.HWo * 4) : 0u);
    constexpr int RS = BM + 4;                                       // floats per channel row of the tile
    float* tl = reinterpret_cast<float*>(smem);
    __syncthreads();                                                 // every wave is done with the window / weight images
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int p = 0; p < NPT; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) tl[(ch0[t] - co_blk + r) * RS + (wave * NPT + p) * 16 + i16] = acc[t][p][r];
    __syncthreads();
    constexpr int G = BM / 4, CPI = 256 / G;                         // 4-pixel groups per channel row, channel rows per pass
    const int g = tid % G, c0 = tid / G;
    const bool gv = P0 + 4 * g < P.total;                            // (a group of 4 stays inside one image: Ho Wo % 4 == 0)
    const int q = gv ? p0 + 4 * g : p0;
    const int qn = (int)sdiv((uint32_t)q, P.mHWo), qi = q - qn * P.HWo;
    const int ob = gv ? (((n0 + qn) * P.out_ctot + P.out_coff + co_blk) * P.HWo + qi) * 4 : SOOB;
    const int rb = (gv && res) ? (((n0 + qn) * P.res_ctot + P.res_coff + co_blk) * P.HWo + qi) * 4 : SOOB;
    const bool relu = P.act == OTP_ACT_RELU;
    f32x4 rv[NTW * 16 / CPI];
#pragma unroll
    for (int k = 0; k < NTW * 16 / CPI; ++k) {
        const int ch = c0 + CPI * k;
        rv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rres, (rb != SOOB && co_blk + ch < P.Cout) ? rb + ch * P.HWo * 4 : SOOB, 0, 0));
    }
    bool bad = false;                                                // range guard (common.h), before the ReLU
#pragma unroll
    for (int k = 0; k < NTW * 16 / CPI; ++k) {
        const int ch = c0 + CPI * k;
        f32x4 v = *reinterpret_cast<const f32x4*>(tl + ch * RS + 4 * g) + rv[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= otp_out_of_range(v[e]);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rof,
                                               (ob != SOOB && co_blk + ch < P.Cout) ? ob + ch * P.HWo * 4 : SOOB, 0, 0);
    }
    otp_range_report(P.rflag, bad, OTP_RANGE_CONVS2);
}

int s8_ntw(int Cout) {
    const int c16 = (Cout + 15) / 16;
    return (c16 % 3 == 0) ? 3 : (c16 % 2 == 0 || c16 <= 2 ? 2 : 3);          // the rule of csrc/convs.hip (shared weight image)
}

int window_records(const S2Plan& P, int bm) {
    int NV = 0;
    for (int t = 0; t < P.nTiles; ++t) {
        const int a = t * bm, b = (a + bm < P.total ? a + bm : P.total) - 1;
        const int na = a / P.HWo, ya = (a % P.HWo) / P.Wo;
        const int nb = b / P.HWo, yb = (b % P.HWo) / P.Wo;
        const int rows = (nb * P.VR + 2 * yb + 2) - (na * P.VR + 2 * ya) + 1;   // virtual rows from tap dy = 0 of the first pixel
        const int v = rows * P.W1;                                              // to tap dy = 2 of the last
        if (v > NV) NV = v;
    }
    return NV;
}

bool convs2_plan(const otp_conv_desc& d, S2Plan& P, bool nchw) {
    if (d.kh != 3 || d.kw != 3 || d.stride != 2 || d.pad != 1 || d.dil != 1) return false;
    if (d.res_up > 1 || d.frame_split > 0 || d.in2_ctot > 0 || d.act == OTP_ACT_GELU) return false;
    if (d.Cin % 16 || d.Cout % 16 || (d.H & 1) || (d.W & 1)) return false;
    const int Ho = d.H / 2, Wo = d.W / 2;
    if (d.Ho != Ho || d.Wo != Wo || ((Ho * Wo) & 3)) return false;
    P.N = d.N; P.C = d.Cin; P.H = d.H; P.W = d.W; P.HW = d.H * d.W; P.Ho = Ho; P.Wo = Wo; P.HWo = Ho * Wo; P.Cout = d.Cout;
    P.total = d.N * P.HWo;
    P.out_ctot = d.out_ctot; P.out_coff = d.out_coff; P.res_ctot = d.res_ctot; P.res_coff = d.res_coff; P.act = d.act;
    P.post = d.out_scale > 0.f ? d.out_scale : 1.f;
    P.pre = 1.f / P.post;
    P.NTW = s8_ntw(d.Cout);
    P.nN = ((d.Cout + 15) / 16 + P.NTW - 1) / P.NTW;
    P.nChunks = d.Cin / 16;
    P.VR = d.H + 1;
    P.W1 = d.W + 1;
    // the largest pixel tile whose window planes + weights let two workgroups share a CU (<= 80 KB) and fit the piece registers
    bool found = false;
    for (int npt = 4; npt >= 1; npt >>= 1) {
        const int bm = 64 * npt;
        P.NPT = npt;
        P.nTiles = (P.total + bm - 1) / bm;
        const int NV = window_records(P, bm);
        size_t lds = (size_t)4 * NV * 16 + swch(P.NTW) * 1024;
        if (nchw && lds < (size_t)P.NTW * 16 * (bm + 4) * 4) lds = (size_t)P.NTW * 16 * (bm + 4) * 4;
        if (NV <= 64 * 4 * MAXJ && lds <= 80 * 1024) {
            // small launches: prefer more, smaller workgroups while that still leaves >= 1 tile per CU pair
            P.NV = NV;
            found = true;
            if ((long)P.nTiles * P.nN >= 2 * 256 || npt == 1) break;
        }
    }
    if (!found) return false;
    {   // re-derive for the chosen NPT (the loop may have stepped past the last valid one)
        const int bm = 64 * P.NPT;
        P.nTiles = (P.total + bm - 1) / bm;
        P.NV = window_records(P, bm);
        if (P.NV > 64 * 4 * MAXJ) return false;
    }
    P.tpx = (P.nTiles + 7) / 8;
    P.pl = P.NV * 16;
    P.NIW = (P.NV + 63) / 64;
    P.mHWo = smagic(P.HWo); P.mWo = smagic(Wo); P.mW1 = smagic(P.W1); P.mVR = smagic(P.VR);
    if ((long)(P.HWo + 256) * P.HWo >= (1l << 32) || (long)P.HWo * Wo >= (1l << 32)) return false;
    if ((long)(d.N + 2) * P.VR * P.VR >= (1l << 32) || (long)(64 * 4 * MAXJ) * P.W1 >= (1l << 32)) return false;
    {   // images a tile's window may touch (a 256-pixel tile of a small map spans many): (n - n0) * imgB stays below 2^31
        const long span = (long)(64 * P.NPT) / P.HWo + 2;
        if (span * d.Cin * P.HW * 4 >= (1l << 31)) return false;
    }
    if ((size_t)d.N * d.out_ctot * P.HWo * 4 >= (1ull << 31) || (size_t)d.N * (d.res_ctot > 0 ? d.res_ctot : 1) * P.HWo * 4 >= (1ull << 31))
        return false;
    if ((size_t)P.nN * P.nChunks * swch(P.NTW) * 1024 >= (1ull << 31)) return false;
    if (P.HWo < 16) return false;
    return true;
}

template <int NTW, bool NCHW, int NPT>
int convs2_launch(const void* xs, const void* wpk, const float* shift, const float* res, float* outf, void* outs, const S2Plan& P,
                  hipStream_t st) {
    auto kern = convs2_kernel<NTW, NCHW, NPT>;
    size_t need = (size_t)4 * P.pl + swch(NTW) * 1024;
    if (NCHW && need < (size_t)NTW * 16 * (64 * NPT + 4) * 4) need = (size_t)NTW * 16 * (64 * NPT + 4) * 4;
    OTP_ALLOW_BIG_LDS(kern, need);
    hipLaunchKernelGGL(kern, dim3(8 * P.tpx * P.nN), dim3(256), need, st, static_cast<const unsigned char*>(xs),
                       static_cast<const u32x4*>(wpk), shift, res, outf, static_cast<u32x4*>(outs), P);
    return otp_launch_status();
}

}  // namespace

extern "C" int otp_conv3x3_s2_s8_supported(const otp_conv_desc* desc, int nchw_out) {
    if (!desc) return 0;
    S2Plan P{};
    return convs2_plan(*desc, P, nchw_out != 0) ? 1 : 0;
}

extern "C" int otp_conv3x3_s2_s8(const void* in_s8, const void* wpacked, const void* shift, const void* res_nchw, void* out_nchw,
                                 void* out_s8, const otp_conv_desc* desc, void* stream) {
    if (!in_s8 || !wpacked || !desc || (!out_nchw) == (!out_s8)) return OTP_ERR_BAD_ARG;      // exactly one output form
    if (res_nchw && !out_nchw) return OTP_ERR_BAD_ARG;
    const otp_conv_desc& d = *desc;
    if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0) return OTP_ERR_BAD_ARG;
    if (out_nchw && d.out_ctot < d.out_coff + d.Cout) return OTP_ERR_BAD_ARG;
    if (res_nchw && d.res_ctot < d.res_coff + d.Cout) return OTP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in_s8) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out_nchw) |
         reinterpret_cast<uintptr_t>(out_s8) | reinterpret_cast<uintptr_t>(res_nchw) | reinterpret_cast<uintptr_t>(shift)) & 15)
        return OTP_ERR_UNSUPPORTED;
    S2Plan P{};
    const bool nchw = out_nchw != nullptr;
    if (!convs2_plan(d, P, nchw)) return OTP_ERR_UNSUPPORTED;
    P.rflag = otp_range_word();
    auto st = static_cast<hipStream_t>(stream);
    auto fs = static_cast<const float*>(shift);
    auto fr = static_cast<const float*>(res_nchw);
    auto fo = static_cast<float*>(out_nchw);
#define OTP_C2_GO(NTW_, NCHW_, NPT_) return convs2_launch<NTW_, NCHW_, NPT_>(in_s8, wpacked, fs, fr, fo, out_s8, P, st)
#define OTP_C2_NPT(NPT_)                                                     \
    if (P.NPT == NPT_) {                                                     \
        if (P.NTW == 2) { if (nchw) OTP_C2_GO(2, true, NPT_); OTP_C2_GO(2, false, NPT_); } \
        if (nchw) OTP_C2_GO(3, true, NPT_);                                  \
        OTP_C2_GO(3, false, NPT_);                                           \
    }
    OTP_C2_NPT(4)
    OTP_C2_NPT(2)
    OTP_C2_NPT(1)
#undef OTP_C2_NPT
#undef OTP_C2_GO
    return OTP_ERR_UNSUPPORTED;
}
