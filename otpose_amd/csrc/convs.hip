// 3x3 / stride 1 / pad 1 convolutions of the HRNet BasicBlocks (reference model/HRNet.py:500-530, 73 % of the forward's
// MACs) on activations kept in the SPLIT RECORD format, fed by the LDS-DMA.
//
// Arithmetic: the split ("bf16x3") products of csrc/convx.hip - a = hi + lo (+ r, |r| <= 2^-18 |a|), products accumulated in
// fp32 as lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_bf16.  What changes is WHERE the split happens and how operands reach
// the matrix cores.  convx.hip reads fp32 NCHW, splits every window element in the consumer (3.1 vector instructions per
// MFMA, 4-pixel-stride LDS writes with 44 % bank-conflict cycles, staging phases the matrix pipe idles through: VERDICT r02
// item 4).  Here the PRODUCER's epilogue emits, next to (or instead of) the fp32 NCHW tensor, the record image the consumer's
// MFMA operands are made of:
//
//     S8 format of a logical (N, C, H, W) fp32 tensor, C % 8 == 0:   [N][C/8][2][H*W] records of 16 bytes
//     record (n, g, part, p) = 8 bf16: part 0 = hi, part 1 = lo of channels 8g .. 8g+7 at pixel p     (4 bytes / element)
//
// A pixel's record is exactly one lane's B operand of a k-slot, and a plane (n, g, part) is contiguous in the pixel index,
// so a consumer stages its input window with `buffer_load_dwordx4 ... lds` only: 64 lanes x 16 bytes = 64 consecutive window
// records per instruction, destination lane-linear in LDS, source offset per lane (computed ONCE per tile; the chunk / plane
// is the instruction's scalar offset), padding columns / rows between images = lanes whose offset is out of the descriptor's
// range (the hardware writes zeros).  No staging registers, no vector instructions, no ds_write in the main loop.
//
// LDS window: per (channel group of the chunk, part) a plane of 512 records; record index = (virtual row) * (W + 1) + 1 + x
// with one zero record between rows (the right neighbour of x = W - 1 IS the left neighbour of the next row's x = 0) and one
// zero row above every image.  Sixteen consecutive output pixels are sixteen consecutive records (17 across a row end), so
// the ds_read_b128 of a B fragment is conflict-free; the two channel groups of a k-slot pair sit 2 planes = 16 KB apart.
// Orientation: A = weights (M = 16 output channels), B = pixels (N = 16 pixels): a lane's accumulator registers are 4
// output channels of ONE pixel.  Epilogue: accumulators (+ shift) -> LDS [cout][pixel] -> rows of 256 pixels: residual add,
// ReLU, fp32 NCHW store as 1 KB runs, then per pixel 8 channels -> hi / lo records -> S8 store as 1 KB runs.
#include "common.h"
#include <cstdlib>

namespace {

typedef otp_x3x8 h16x8;              // 8 operand pieces of the split products (common.h: IEEE half since round 4)
typedef otp_x3x2 h16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 1 KB pieces of a chunk's packed weights: 4 full k-steps x NTW tiles x (hi, lo) + the half-filled fifth (512 bytes per fragment)
__host__ __device__ constexpr int swch(int ntw) { return 8 * ntw + ntw; }
constexpr int SKS = 5;                    // k-steps per chunk: 18 (tap, group) slots of 8 channels in 5 x 4 (2 zero-weight slots)
constexpr int SOOB = -16;                 // buffer offset outside every descriptor: the load returns / writes zeros

__device__ __forceinline__ uint32_t sdiv(uint32_t i, uint32_t magic) { return magic ? __umulhi(i, magic) : i; }
// a * b of the per-lane index arithmetic, both operands below 2^24 (convs_plan checks): v_mul_u32_u24 issues at full rate,
// v_mul_lo_u32 at a quarter of it
__device__ __forceinline__ int smul(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }
uint32_t smagic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }   // exact while i * d < 2^32

#ifdef OTP_CONVS_TIMING
// development build only (tools/convs_timing.sh): per-workgroup phase stamps, never in libotpose_hip.so
__device__ unsigned long long otp_convs_stamps[8192 * 32];
#define SSTAMP(slot)                                                                                  \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 8192) otp_convs_stamps[blockIdx.x * 32 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define SSTAMP(slot)
#endif

struct SPlan {
    int N, C, H, W, HW, Cout, total;
    int out_ctot, out_coff, act, f32_mode, res_s8;
    float pre, post;                      // weights carry 2^k = pre (otp_conv_desc.out_scale = post = 2^-k): see the epilogue
    int NTW, nN, nTiles, nChunks, tpx;
    int NPT;                              // pixel tiles of 16 per wave (workgroup tile = 64 NPT pixels)
    int VR, W1, NIW, NV, pl;              // virtual rows per image (H + 1), records per window row (W + 1), 64-record pieces per plane,
                                          // records / bytes of a window plane
    uint32_t mHW, mW, mW1, mVR;
    unsigned* rflag;                      // range-guard word (common.h)
};

// 8 floats -> bf16 hi / lo records
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

// fp32 NCHW (channel-sliced view) -> S8 (+ the C4 image [N][C/4][H*W][4] of the same values, the residual layout of
// otp_conv3x3_s8).  A thread owns ONE pixel of one 8-channel group: 8 dword loads (256-byte runs per channel row and wave)
// and one 16-byte store per image and part - every store instruction of a wave writes 1 KB of consecutive records.  (Round 3's
// form gave a thread 4 consecutive pixels: its stores were 16-byte pieces 64 bytes apart, and the pass ran at 2.4 TB/s;
// "store-run length is worth a factor on this chip")
__global__ __launch_bounds__(256) void s8_pack_kernel(const float* __restrict__ in, u32x4* __restrict__ out, float* __restrict__ c4,
                                                       int N, int C, int HW, int ctot, int coff, unsigned* rflag) {
    const int G8 = C >> 3;
    const size_t items = (size_t)N * G8 * HW;
    bool bad = false;                                               // range guard (common.h): a value its half pieces cannot hold
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const size_t r = i / HW;
        const int g = (int)(r % G8), n = (int)(r / G8);
        const float* src = in + ((size_t)n * ctot + coff + 8 * g) * HW + p;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = src[(size_t)e * HW];
#pragma unroll
        for (int e = 0; e < 8; ++e) bad |= otp_out_of_range(f[e]);
        u32x4 hi, lo;
        ssplit8(f, hi, lo);
        u32x4* dst = out + ((size_t)(n * G8 + g) * 2) * HW + p;
        dst[0] = hi;
        dst[HW] = lo;
        if (c4) {
            f32x4* d4 = reinterpret_cast<f32x4*>(c4) + ((size_t)n * (C >> 2) + 2 * g) * HW + p;
            d4[0] = (f32x4){f[0], f[1], f[2], f[3]};
            d4[HW] = (f32x4){f[4], f[5], f[6], f[7]};
        }
    }
    otp_range_report(rflag, bad, OTP_RANGE_S8PASS);
}

// A fuse row's upsampled terms (HRNet.py:487-494; otp_upsample_add_multi: out = act(res + up_f0(low0) + up_f1(low1) + ...), added
// in that order) written as the images the next module's branch reads: S8 + C4 (and the NCHW tensor only when somebody else
// needs it).  Same additions in the same order as upsample_add_multi_kernel: bit-identical values.  A thread owns 4
// consecutive pixels of one 8-channel group.
struct S8Up {
    const float* low[3];
    int f[3];
    int n;
};
// the round-3 form: a thread owns 4 consecutive pixels of one 8-channel group - a quarter of the low-resolution loads per pixel
// (kept for rows with three upsampled terms, where those loads outweigh the short store runs: 117 against 138 us at 48 channels
// @96x72 with terms at 1/2, 1/4, 1/8 resolution; with one or two terms the one-pixel form below is 25 % faster)
// `res`: the fp32 NCHW residual, or (res_s8 != 0) its S8 image - hi + lo of the records, the same 4 bytes per element
__global__ __launch_bounds__(256) void s8_upsample_add4_kernel(S8Up U, const float* __restrict__ res, float* out_nchw,
                                                               u32x4* __restrict__ out_s8, float* __restrict__ out_c4, int N, int C,
                                                               int Hh, int Wh, int relu, int res_ctot, int res_coff, int out_ctot,
                                                               int out_coff, int res_s8, unsigned* rflag) {
    const int HW = Hh * Wh, q4 = HW >> 2, G8 = C >> 3, Wh4 = Wh >> 2;
    const size_t items = (size_t)N * G8 * q4;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % q4);
        const size_t r = i / q4;
        const int g = (int)(r % G8), n = (int)(r / G8);
        const int y = q / Wh4, x4 = q - y * Wh4;
        f32x4 v[8];
        if (res_s8) {                                               // (uniform) records of the four pixels -> v[channel][pixel]
            const u32x4* rs = reinterpret_cast<const u32x4*>(res) + ((size_t)(n * G8 + g) * 2) * HW + 4 * q;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 h = rs[k], l = rs[(size_t)HW + k];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const otp_f32x2 a = otp_x3_widen(h[e2]) + otp_x3_widen(l[e2]);
                    v[2 * e2][k] = a.x;
                    v[2 * e2 + 1][k] = a.y;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * g + e;
            f32x4 o = res_s8 ? v[e] : *reinterpret_cast<const f32x4*>(res + ((size_t)n * res_ctot + res_coff + c) * HW + 4 * q);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < U.n) {
                    const int f = U.f[k], Wl = Wh / f, Hl = Hh / f;
                    const float* lrow = U.low[k] + (((size_t)n * C + c) * Hl + y / f) * Wl;
                    if (f >= 4) {
                        o = o + lrow[(4 * x4) / f];
                    } else {
                        const float l0 = lrow[2 * x4], l1 = lrow[2 * x4 + 1];
                        o = o + (f32x4){l0, l0, l1, l1};
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bad |= otp_out_of_range(o[j]);
            if (relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
            }
            v[e] = o;
            if (out_nchw) *reinterpret_cast<f32x4*>(out_nchw + ((size_t)n * out_ctot + out_coff + c) * HW + 4 * q) = o;
        }
        u32x4* dst = out_s8 + ((size_t)(n * G8 + g) * 2) * HW + 4 * q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = v[e][k];
            u32x4 hi, lo;
            ssplit8(f, hi, lo);
            dst[k] = hi;
            dst[(size_t)HW + k] = lo;
            if (out_c4) {
                f32x4* d4 = reinterpret_cast<f32x4*>(out_c4) + ((size_t)n * (C >> 2) + 2 * g) * HW + 4 * q + k;
                d4[0] = (f32x4){f[0], f[1], f[2], f[3]};
                d4[HW] = (f32x4){f[4], f[5], f[6], f[7]};
            }
        }
    }
    otp_range_report(rflag, bad, OTP_RANGE_S8PASS);
}

__global__ __launch_bounds__(256) void s8_upsample_add_kernel(S8Up U, const float* __restrict__ res, float* out_nchw,
                                                               u32x4* __restrict__ out_s8, float* __restrict__ out_c4, int N, int C,
                                                               int Hh, int Wh, int relu, int res_ctot, int res_coff, int out_ctot,
                                                               int out_coff, int res_s8, unsigned* rflag) {
    // a thread owns ONE pixel of one 8-channel group (see s8_pack_kernel: 1 KB store runs per wave instruction)
    const int HW = Hh * Wh, G8 = C >> 3;
    const size_t items = (size_t)N * G8 * HW;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const size_t r = i / HW;
        const int g = (int)(r % G8), n = (int)(r / G8);
        const int y = p / Wh, x = p - y * Wh;
        int lo_off[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int f = k < U.n ? U.f[k] : 1;
            lo_off[k] = (y / f) * (Wh / f) + x / f;                 // pixel of term k's low-resolution map
        }
        float f8[8];
        if (res_s8) {                                               // (uniform) the pixel's record pair of this channel group
            const u32x4* rs = reinterpret_cast<const u32x4*>(res) + ((size_t)(n * G8 + g) * 2) * HW + p;
            const u32x4 h = rs[0], l = rs[HW];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const otp_f32x2 a = otp_x3_widen(h[e2]) + otp_x3_widen(l[e2]);
                f8[2 * e2] = a.x;
                f8[2 * e2 + 1] = a.y;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * g + e;
            float o = res_s8 ? f8[e] : res[((size_t)n * res_ctot + res_coff + c) * HW + p];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < U.n) {
                    const int f = U.f[k];
                    o = o + U.low[k][((size_t)n * C + c) * (size_t)((Hh / f) * (Wh / f)) + lo_off[k]];
                }
            }
            bad |= otp_out_of_range(o);
            if (relu) o = fmaxf(o, 0.f);
            f8[e] = o;
            if (out_nchw) out_nchw[((size_t)n * out_ctot + out_coff + c) * HW + p] = o;
        }
        u32x4 hi, lo;
        ssplit8(f8, hi, lo);
        u32x4* dst = out_s8 + ((size_t)(n * G8 + g) * 2) * HW + p;
        dst[0] = hi;
        dst[HW] = lo;
        if (out_c4) {                                               // (uniform; NULL when the consumer reads its residual as S8)
            f32x4* d4 = reinterpret_cast<f32x4*>(out_c4) + ((size_t)n * (C >> 2) + 2 * g) * HW + p;
            d4[0] = (f32x4){f8[0], f8[1], f8[2], f8[3]};
            d4[HW] = (f32x4){f8[4], f8[5], f8[6], f8[7]};
        }
    }
    otp_range_report(rflag, bad, OTP_RANGE_S8PASS);
}

// C4 -> fp32 NCHW: test / debugging aid
__global__ __launch_bounds__(256) void c4_unpack_kernel(const f32x4* __restrict__ in, float* __restrict__ out, int N, int C, int HW) {
    const size_t items = (size_t)N * (C >> 2) * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const size_t r = i / HW;
        const int c4 = (int)(r % (C >> 2)), n = (int)(r / (C >> 2));
        const f32x4 v = in[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) out[((size_t)n * C + 4 * c4 + e) * HW + p] = v[e];
    }
}

// S8 -> fp32 NCHW (hi + lo in fp32): test / debugging aid, not on the forward path
__global__ __launch_bounds__(256) void s8_unpack_kernel(const u32x4* __restrict__ in, float* __restrict__ out, int N, int C, int HW) {
    const int G8 = C >> 3;
    const size_t items = (size_t)N * G8 * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const size_t r = i / HW;
        const int g = (int)(r % G8), n = (int)(r / G8);
        const u32x4 hi = in[((size_t)(n * G8 + g) * 2) * HW + p], lo = in[((size_t)(n * G8 + g) * 2 + 1) * HW + p];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x2 fh = otp_x3_widen(hi[e >> 1]), fl = otp_x3_widen(lo[e >> 1]);
            out[((size_t)n * C + 8 * g + e) * HW + p] = fh[e & 1] + fl[e & 1];
        }
    }
}

// Output-channel row of an MFMA tile <-> channel.  A lane's accumulator registers of a tile are rows 4 kl .. 4 kl + 3 (kl =
// lane / 16) of one pixel.  Cout tiles go in pairs (2 tp, 2 tp + 1): row 4 kl + r of the even tile is channel 8 kl + r of the
// pair's 32, of the odd tile channel 8 kl + 4 + r - a lane then holds 8 CONSECUTIVE channels of its pixel = one S8 record
// group, and the epilogue splits and stores them without any cross-lane traffic.  A tile without a partner (odd tile count, or
// the partner past Cout) keeps the identity: 4 consecutive channels per lane, stored as half records.
__host__ __device__ inline bool stile_paired(int co_blk, int t, int ntw, int Cout) {
    const int tb = t | 1;
    return tb < ntw && co_blk + 16 * tb < Cout;
}
__host__ __device__ inline int srow2ch(int co_blk, int t, int row, int ntw, int Cout) {
    return stile_paired(co_blk, t, ntw, Cout) ? co_blk + 32 * (t >> 1) + 8 * (row >> 2) + 4 * (t & 1) + (row & 3)
                                               : co_blk + 16 * t + row;
}

// packed weights of otp_conv3x3_s8: the image of otp_conv2d_x3_pack_weight (k = 3, stride 1) with the rows of every 16-row
// tile in srow2ch order: [cout block][chunk][k-step][cout tile][hi, lo][lane][8] bf16, lane (i16, kl): row i16 of the tile =
// channel srow2ch(block, tile, i16), k-slot q = 4 s + kl -> tap q / 2, input channels 16 chunk + 8 (q % 2) .. + 7; the fifth
// k-step holds k-slots 16, 17 only (lanes 0 .. 31 of a fragment): [cout tile][hi, lo][32 lanes] - 9 NTW KB per chunk, not 10
__global__ void s8_wpack_kernel(const float* __restrict__ w, const float* __restrict__ scale, u32x4* __restrict__ out, int Cout,
                                int Cin, int NTW, int nN, int nChunks) {
    const int total = nN * nChunks * SKS * NTW * 64;
    const int WU = swch(NTW) * 64;                                 // 16-byte units of one (cout block, chunk) image
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63;
        int r = idx >> 6;
        const int t = r % NTW; r /= NTW;
        const int s = r % SKS; r /= SKS;
        const int chunk = r % nChunks, cb = r / nChunks;
        const int cout = srow2ch(cb * NTW * 16, t, lane & 15, NTW, Cout), kl = lane >> 4;
        const int q = 4 * s + kl, tap = q >> 1, ci0 = chunk * 16 + 8 * (q & 1);
        if (tap > 8) continue;                                     // k-slots 18, 19: not stored (the kernel multiplies zeros there)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = ci0 + j;
            v[j] = (cout < Cout && ci < Cin) ? w[((size_t)cout * Cin + ci) * 9 + tap] * (scale ? scale[cout] : 1.f) : 0.f;
        }
        u32x4 hi, lo;
        ssplit8(v, hi, lo);
        const size_t base = (size_t)(cb * nChunks + chunk) * WU;
        if (s < SKS - 1) {
            const size_t o = base + ((s * NTW + t) * 2) * 64 + lane;
            out[o] = hi;
            out[o + 64] = lo;
        } else {
            const size_t o = base + (SKS - 1) * NTW * 128 + t * 64 + lane;   // lanes 0 .. 31: hi, then lo, 512 bytes each
            out[o] = hi;
            out[o + 32] = lo;
        }
    }
}

// (The two halves of a wave trade accumulator rows with __shfl_xor(v, 32) = ds_bpermute_b32.  v_permlane32_swap would do it in
// one instruction, but with hipcc / ROCm 7.2 the second result of __builtin_amdgcn_permlane32_swap comes back as a copy of the
// first (both extractvalue indices are 0 in the emitted IR), and the instruction in inline asm gave results that changed when
// other kernels shared the CU - hipcc inserts no wait states around an asm statement.)

// instruction order of one (k-step, pixel tile) block: NM MFMAs and NR LDS reads - [MFMA, read] pairs while reads remain
// (two MFMAs first when there are few), then the remaining MFMAs
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

// residual / fp32 output layouts of otp_conv3x3_s8
enum { S_F32_NONE = 0, S_F32_C4 = 1, S_F32_NCHW = 2 };

template <int NTW, bool NCHW, int NPT>
__global__ __launch_bounds__(256, NCHW ? 2 : 3) void convs_kernel(const unsigned char* __restrict__ xs, const u32x4* __restrict__ wpk,
                                                        const float* __restrict__ shift, const float* res, float* outf,
                                                        u32x4* outs, const SPlan P) {
    // NPT pixel tiles of 16 per wave: workgroup tiles of 256 pixels, or of 128 for launches that would leave CUs with fewer than
    // three workgroups (the small maps: a workgroup's set-up, waits and epilogue only overlap with ANOTHER workgroup's MFMAs)
    constexpr int BM = 64 * NPT;
    constexpr int WCH = swch(NTW);                                 // 1 KB pieces of a chunk's weights
    constexpr int WBYTES = WCH * 1024;
    constexpr int NBLK = SKS * NPT;                                // (k-step, pixel tile) blocks of a chunk: 3 NTW MFMAs each
    constexpr int NM = 3 * NTW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PL = P.pl;                                           // bytes between the planes of the window (no padding: three
    unsigned char* win = smem;                                     // workgroups of 4 planes + 27 KB of weights share a CU's 160 KB)
    unsigned char* wl = smem + 4 * PL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kl = lane >> 4;
    // lane (i16, kl) of pixel tile p: pixel m = (NPT wave + p) 16 + i16 of the tile; accumulator register r of cout tile t =
    // channel ch0[t] + r of that pixel (srow2ch: the row permutation of the packed weights)
    const bool upper = kl >= 2;

    // workgroup -> (pixel tile, output-channel block); XCD x (block id mod 8) walks a contiguous tile range, the blocks of a
    // tile back to back: the window rows neighbouring tiles share and the re-read window of the next block hit that L2
    const int xcd = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
    const int tl = jb / P.nN, cb = jb - tl * P.nN;
    const int tile = xcd * P.tpx + tl;
    if (tile >= P.nTiles) return;
    SSTAMP(0);
#ifdef OTP_CONVS_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 8192) otp_convs_stamps[blockIdx.x * 32 + 30] = __builtin_amdgcn_s_memrealtime();
#endif
    const int P0 = tile * BM;
    const int n0 = P0 / P.HW, p0 = P0 - n0 * P.HW;            // (uniform, once per workgroup)
    const int y0 = (int)sdiv((uint32_t)p0, P.mW);
    const int x0 = p0 - y0 * P.W;                                  // the window starts at the first record any tap reads: (row above, x0 - 1)
    const int Vf = n0 * P.VR + y0;                                 // first virtual row of the window (one above the first pixel's)
    const int imgB = P.C * P.HW * 4;                               // bytes of one image of the S8 tensor
    const int co_blk = cb * NTW * 16;
    const int C4o = P.Cout >> 2;

    // ---- window pieces of this wave: piece k = wave + 4 j covers window records 64 k .. 64 k + 63 ------------------------------
    // window record w <-> record x0 + w of the row-major frame (virtual row r, column cp): r = (x0 + w) / (W + 1)
    int voff[2];
    bool vlive[2];                                                 // the last piece of a plane is partial: lanes past the plane's
#pragma unroll                                                     // last record stay out of the DMA (they would write the next plane)
    for (int j = 0; j < 2; ++j) {
        vlive[j] = 64 * (wave + 4 * j) + lane < P.NV;
        const int v = 64 * (wave + 4 * j) + lane + x0;
        const int r = (int)sdiv((uint32_t)v, P.mW1), cp = v - smul(r, P.W1);
        const int V = Vf + r;
        const int n = (int)sdiv((uint32_t)V, P.mVR), yy = V - smul(n, P.VR);
        const bool ok = cp >= 1 && yy >= 1 && n < P.N;
        voff[j] = ok ? (n - n0) * imgB + (smul(yy - 1, P.W) + cp - 1) * 16 : SOOB;
    }
    const size_t left = (size_t)(P.N - n0) * imgB;
    const otp_rsrc rin = make_rsrc32(xs + (size_t)n0 * imgB, left > 0x7fffff00ull ? 0x7fffff00u : (unsigned)left);
    const otp_rsrc rw = make_rsrc32(wpk, (unsigned)((size_t)P.nN * P.nChunks * WBYTES));
    const int woff = lane * 16;

    auto stage = [&](int c) __attribute__((always_inline)) {
        // window: planes (group gl, part) of chunk c; plane index gl * 2 + part
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            const int so = (((2 * c + (pl >> 1)) * 2 + (pl & 1)) * P.HW) * 16;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
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
    SSTAMP(1);

    // ---- per pixel tile: fragment address, lane offsets into the output / residual images; accumulators start from
    //      residual + shift (loaded while the first chunk's DMA is in flight: no registers of their own, no epilogue add) -----
    const size_t obytes = (size_t)P.N * P.Cout * P.HW * 4;         // C4 and S8 images of the (N, Cout, H, W) result / residual
    const otp_rsrc rres = make_rsrc32(res ? res : reinterpret_cast<const float*>(xs), res ? (unsigned)obytes : 0u);
    const otp_rsrc rs8 = make_rsrc32(outs, outs ? (unsigned)obytes : 0u);
    const otp_rsrc rof = make_rsrc32(outf, !outf ? 0u : (P.f32_mode == S_F32_C4 ? (unsigned)obytes
                                                                               : (unsigned)((size_t)P.N * P.out_ctot * P.HW * 4)));
    const otp_rsrc rsh = make_rsrc32(shift ? shift : reinterpret_cast<const float*>(xs), shift ? (unsigned)(P.Cout * 4) : 0u);
    // (NCHW = false: the fp32 output, if any, is a C4 image, whose pixel offsets equal the S8 image's: Cout / 4 = 2 Cout / 8)
    int pb[NPT], toff[SKS], offN[NCHW ? NPT : 1], offS[NPT], ch0[NTW];
    f32x4 acc[NTW][NPT];
    {
#pragma unroll
        for (int s = 0; s < SKS; ++s) {
            const int q = 4 * s + kl;
            int tap = q >> 1;
            if (tap > 8) tap = 8;                                  // zero weights: any finite data
            const int dy = tap / 3, dx = tap - dy * 3;
            toff[s] = (dy * P.W1 + dx - x0) * 16 + (q & 1) * (2 * PL);
        }
        f32x4 sh[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            ch0[t] = srow2ch(co_blk, t, 4 * kl, NTW, P.Cout);      // (a channel past Cout for tiles past it: never loaded / stored)
            sh[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsh, co_blk + 16 * t < P.Cout ? ch0[t] * 4 : SOOB, 0, 0));
        }
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
            int m = (wave * NPT + p) * 16 + i16;
            const bool pv = P0 + m < P.total;
            if (!pv) m = P.total - 1 - P0;                         // tail tile: a finite address, the result is dropped
            const int q = p0 + m;
            const int dn = (int)sdiv((uint32_t)q, P.mHW), pi = q - smul(dn, P.HW);
            const int y = (int)sdiv((uint32_t)pi, P.mW), x = pi - smul(y, P.W);
            pb[p] = (smul(smul(n0 + dn, P.VR) + y - Vf, P.W1) + x) * 16;   // record of tap (0, 0): one row up, one column left (+ x0)
            const int img = n0 + dn;
            // pixel part of the byte offsets; the channel part (ch0[t]) is added where it is used
            //   C4 image   ((img C4o + ch / 4) HW + pi) 16          NCHW slice ((img ctot + coff + ch) HW + pi) 4
            //   S8 image   (((img Go + ch / 8) 2 + part) HW + pi) 16
            const int c4o = (smul(smul(img, C4o), P.HW) + pi) * 16;
            if (NCHW) offN[NCHW ? p : 0] = pv ? (smul(smul(img, P.out_ctot) + P.out_coff, P.HW) + pi) * 4 : SOOB;
            offS[p] = pv ? c4o : SOOB;
            // residual (C4 image; out-of-range offsets read zeros) + shift
            if (!P.res_s8) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const bool tv = co_blk + 16 * t < P.Cout;
                    acc[t][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rres, (pv && tv) ? c4o + smul(ch0[t] >> 2, P.HW) * 16 : SOOB, 0, 0));
                }
            }
        }
        if (P.res_s8) {
            // (a tile past Cout behind an unpaired one gets no residual below: zero, not whatever the registers held - nothing
            //  stores it, but the range guard of the epilogue reads every accumulator)
#pragma unroll
            for (int t = 1; t < NTW; t += 2)
#pragma unroll
                for (int p = 0; p < NPT; ++p) acc[t][p] = f32x4{0.f, 0.f, 0.f, 0.f};
            // residual as S8 records (otp_conv_desc.res_layout = 1): the block input's operand image IS the residual - hi + lo holds
            // it to 2^-22 - so no fp32 (C4) image of it has to exist.  The records are read the way the epilogue writes them: a
            // lane's registers of a tile pair are the 8 channels of one record group (srow2ch), a tile without a partner takes
            // the lower or upper half of one.
#pragma unroll
            for (int t = 0; t < NTW; t += 2) {
                const bool tav = co_blk + 16 * t < P.Cout;
                const int so = smul(ch0[t] >> 3, P.HW) * 32;
                if (stile_paired(co_blk, t, NTW, P.Cout)) {          // (uniform)
#pragma unroll
                    for (int p = 0; p < NPT; ++p) {
                        const int o = (tav && offS[p] != SOOB) ? offS[p] + so : SOOB;
                        const u32x4 h = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, o, 0, 0));
                        const u32x4 l = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, o, P.HW * 16, 0));
                        const otp_f32x2 a0 = otp_x3_widen(h[0]) + otp_x3_widen(l[0]), a1 = otp_x3_widen(h[1]) + otp_x3_widen(l[1]);
                        const otp_f32x2 a2 = otp_x3_widen(h[2]) + otp_x3_widen(l[2]), a3 = otp_x3_widen(h[3]) + otp_x3_widen(l[3]);
                        acc[t][p] = f32x4{a0.x, a0.y, a1.x, a1.y};
                        acc[t + 1 < NTW ? t + 1 : t][p] = f32x4{a2.x, a2.y, a3.x, a3.y};
                    }
                } else {
                    const int half = (ch0[t] >> 2) & 1;
#pragma unroll
                    for (int p = 0; p < NPT; ++p) {
                        const int o = (tav && offS[p] != SOOB) ? offS[p] + so + 8 * half : SOOB;
                        const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, o, 0, 0));
                        const u32x2 l = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, o, P.HW * 16, 0));
                        const otp_f32x2 a0 = otp_x3_widen(h[0]) + otp_x3_widen(l[0]), a1 = otp_x3_widen(h[1]) + otp_x3_widen(l[1]);
                        acc[t][p] = f32x4{a0.x, a0.y, a1.x, a1.y};
                    }
                }
            }
        }
        // the weights carry a factor pre = 2^k (both half pieces of every weight normal: otp_conv_desc.out_scale), so the sum
        // starts from (residual + shift) * 2^k - exact - and is multiplied by post = 2^-k in the epilogue
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const f32x4 shp = sh[t] * P.pre;
#pragma unroll
            for (int p = 0; p < NPT; ++p) acc[t][p] = acc[t][p] * P.pre + shp;
        }
    }

    // One chunk: NBLK blocks of 3 NTW MFMAs.  B fragments are read two blocks ahead (ring of three), the weight fragments of
    // the next k-step two blocks before it starts, the reads spread between the MFMAs (tools/micro/mfma_loop.hip: 16.8 cycles
    // per MFMA for one wave per SIMD, against 20.8 with reads one block ahead, clustered, and addresses computed in the loop).
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
                    // last k-step: k-slots 16, 17 (tap 8) on the lanes kl = 0, 1; kl = 2, 3 (tap 9) multiply zeros - their half of
                    // the fragment is not stored (512-byte half pieces behind the full ones)
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
        load_b(1, 1);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const int s = blk / NPT, p = blk % NPT, cur = blk % 3, sa = s & 1;
            const bool nb = blk + 2 < NBLK, na = p == NPT - 2 && s + 1 < SKS;
            if (nb) load_b((blk + 2) % 3, blk + 2);
            if (na) load_a(sa ^ 1, s + 1);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                acc[t][p] = OTP_X3_MFMA(al[sa][t], bh[cur], acc[t][p], 0, 0, 0);
                acc[t][p] = OTP_X3_MFMA(ah[sa][t], bl[cur], acc[t][p], 0, 0, 0);
                acc[t][p] = OTP_X3_MFMA(ah[sa][t], bh[cur], acc[t][p], 0, 0, 0);
            }
            if (!nb) sblock_sched<NM, 0>();
            else if (na) sblock_sched<NM, 2 + 2 * NTW>();
            else sblock_sched<NM, 2>();
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    SSTAMP(2);
    for (int c = 0; c < P.nChunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of chunk c have landed
        if (c < 3) SSTAMP(3 + 4 * c);
        __syncthreads();                                           // ... and everybody else's
        if (c < 3) SSTAMP(4 + 4 * c);
        mfma_phase();
        if (c < 3) SSTAMP(5 + 4 * c);
        if (c + 1 < P.nChunks) {
            __syncthreads();                                       // every wave is done with the LDS image of chunk c
            stage(c + 1);
        }
        if (c < 3) SSTAMP(6 + 4 * c);
    }
    SSTAMP(16);

    // ---- epilogue: no LDS, no barrier, no cross-lane traffic - every lane stores what its accumulators hold ---------------------
    // All arithmetic first, all stores last: no register a store reads is written again before the wave ends (the S8 records of
    // one pixel tile used to be rebuilt in the registers the previous tile's stores were still reading - results then changed
    // with the load on the memory pipeline, i.e. with who else was resident on the CU).
    if (P.post != 1.f) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < NPT; ++p) acc[t][p] = acc[t][p] * P.post;
    }
    {   // range guard (common.h): NaN = an operand piece overflowed, |v| >= 65504 = the next consumer could not split it - before
        // the ReLU, which would swallow the NaN
        bool bad = false;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < NPT; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) bad |= otp_out_of_range(acc[t][p][r]);
        otp_range_report(P.rflag, bad, OTP_RANGE_CONVS);
    }
    if (P.act == OTP_ACT_RELU) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < NPT; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][p][r] = otp_relu(acc[t][p][r]);
    }
    // S8 records straight from the accumulators: a lane's registers of a tile PAIR are 8 consecutive channels of its pixel
    // (srow2ch) - split, one hi and one lo record per (pair, pixel tile); a tile without a partner gives 4 consecutive channels
    // = the lower or upper half of a record, stored as 8 bytes (the lane 16 away writes the other half)
    u32x4 rec[(NTW + 1) / 2][NPT][2];
    if (outs) {
#pragma unroll
        for (int t = 0; t < NTW; t += 2) {
            const bool paired = stile_paired(co_blk, t, NTW, P.Cout);       // (uniform)
            const int t1 = t + 1 < NTW ? t + 1 : t;
#pragma unroll
            for (int p = 0; p < NPT; ++p) {
                const float f[8] = {acc[t][p][0], acc[t][p][1], acc[t][p][2], acc[t][p][3],
                                    paired ? acc[t1][p][0] : 0.f, paired ? acc[t1][p][1] : 0.f,
                                    paired ? acc[t1][p][2] : 0.f, paired ? acc[t1][p][3] : 0.f};
                ssplit8(f, rec[t >> 1][p][0], rec[t >> 1][p][1]);
            }
        }
    }
    SSTAMP(17);
    if (!NCHW && P.f32_mode == S_F32_C4) {
        // [N][Cout/4][H*W][4]: one float4 per (cout tile, pixel tile); the 16 lanes of a row write 256 contiguous bytes
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < NPT; ++p)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t][p]), rof,
                                                       (co_blk + 16 * t < P.Cout && offS[p] != SOOB) ? offS[p] + smul(ch0[t] >> 2, P.HW) * 16 : SOOB,
                                                       0, 0);
    } else if (NCHW) {
        // channel slice of an NCHW tensor (the tensor a fuse layer / another kernel family reads).  Straight from the accumulators
        // a store instruction would write 64-byte runs (16 pixels of one channel per lane group) - measured at a third of the
        // rate of long runs (csrc/stem.hip) - so the workgroup's [16 NTW channels][BM pixels] tile passes through the LDS (free
        // once every wave has left the chunk loop) and leaves as 16-byte stores, 64 lanes = 1 KB of one channel row.
        constexpr int RS = BM + 4;                                   // floats per channel row of the tile
        float* tl = reinterpret_cast<float*>(smem);
        __syncthreads();                                             // every wave is done with the window / weight images
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int p = 0; p < NPT; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    tl[(ch0[t] - co_blk + r) * RS + (wave * NPT + p) * 16 + i16] = acc[t][p][r];
        __syncthreads();
        constexpr int G = BM / 4, CPI = 256 / G;                     // 4-pixel groups per channel row, channel rows per pass
        const int g = tid % G, c0 = tid / G;
        const bool gv = P0 + 4 * g < P.total;                        // (a group of 4 stays inside one image: H W % 4 == 0)
        const int q = gv ? p0 + 4 * g : p0;                          // relative to image n0, like the pixel tiles above
        const int qn = (int)sdiv((uint32_t)q, P.mHW), qi = q - smul(qn, P.HW);
        const int ob = gv ? (smul(smul(n0 + qn, P.out_ctot) + P.out_coff + co_blk, P.HW) + qi) * 4 : SOOB;
#pragma unroll
        for (int k = 0; k < NTW * 16 / CPI; ++k) {
            const int ch = c0 + CPI * k;
            const u32x4 v = *reinterpret_cast<const u32x4*>(tl + ch * RS + 4 * g);
            __builtin_amdgcn_raw_buffer_store_b128(v, rof, (ob != SOOB && co_blk + ch < P.Cout) ? ob + smul(ch, P.HW) * 4 : SOOB, 0, 0);
        }
    }
    if (outs) {
#pragma unroll
        for (int t = 0; t < NTW; t += 2) {
            const bool tav = co_blk + 16 * t < P.Cout;
            const int so = smul(ch0[t] >> 3, P.HW) * 32;           // record group of the lane's channels, part 0
            if (stile_paired(co_blk, t, NTW, P.Cout)) {
#pragma unroll
                for (int p = 0; p < NPT; ++p) {
                    const int o = offS[p] != SOOB ? offS[p] + so : SOOB;
                    __builtin_amdgcn_raw_buffer_store_b128(rec[t >> 1][p][0], rs8, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(rec[t >> 1][p][1], rs8, o, P.HW * 16, 0);
                }
            } else {
                const int half = (ch0[t] >> 2) & 1;
#pragma unroll
                for (int p = 0; p < NPT; ++p) {
                    const int o = (tav && offS[p] != SOOB) ? offS[p] + so + 8 * half : SOOB;
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){rec[t >> 1][p][0][0], rec[t >> 1][p][0][1]}, rs8, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){rec[t >> 1][p][1][0], rec[t >> 1][p][1][1]}, rs8, o, P.HW * 16, 0);
                }
            }
        }
    }
    SSTAMP(19);
#ifdef OTP_CONVS_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 8192) otp_convs_stamps[blockIdx.x * 32 + 31] = __builtin_amdgcn_s_memrealtime();
#endif
}

// development override of the pixel tiles per wave (OTPOSE_S8_NPT = 2 / 4; default: by launch size)
int force_npt() {
    static const int v = [] {
        const char* e = getenv("OTPOSE_S8_NPT");
        return e ? atoi(e) : 0;
    }();
    return v;
}

int s8_ntw(int Cout) {
    const int c16 = (Cout + 15) / 16;
    return (c16 % 3 == 0) ? 3 : (c16 % 2 == 0 || c16 <= 2 ? 2 : 3);          // same rule as csrc/convx.hip (shared weight image)
}

bool convs_plan(const otp_conv_desc& d, SPlan& P) {
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1) return false;
    if (d.res_up > 1 || d.frame_split > 0 || d.in2_ctot > 0 || d.act == OTP_ACT_GELU) return false;
    if (d.Cin % 16 || d.Cout % 16 || ((d.H * d.W) & 3) || d.Ho != d.H || d.Wo != d.W) return false;
    P.N = d.N; P.C = d.Cin; P.H = d.H; P.W = d.W; P.HW = d.H * d.W; P.Cout = d.Cout; P.total = d.N * P.HW;
    P.out_ctot = d.out_ctot; P.out_coff = d.out_coff; P.act = d.act; P.f32_mode = S_F32_NONE;
    P.post = d.out_scale > 0.f ? d.out_scale : 1.f;
    P.pre = 1.f / P.post;
    P.res_s8 = d.res_layout == 1;
    P.NTW = s8_ntw(d.Cout);
    P.nN = ((d.Cout + 15) / 16 + P.NTW - 1) / P.NTW;
    P.NPT = 4;
    if ((long)((P.total + 255) / 256) * P.nN < 3 * 256 && force_npt() != 4) P.NPT = 2;
    if (force_npt() == 2) P.NPT = 2;
    const int bm = 64 * P.NPT;
    P.nTiles = (P.total + bm - 1) / bm;
    P.nChunks = d.Cin / 16;
    P.tpx = (P.nTiles + 7) / 8;
    P.VR = d.H + 1;
    P.W1 = d.W + 1;
    // window records of a tile: from the first pixel's tap (0, 0) = (row above, x0 - 1) to the last pixel's tap (2, 2), in the
    // row-major frame of W + 1 records per virtual row; the maximum over the launch's tiles sizes the LDS planes
    int NV = 0;
    for (int t = 0; t < P.nTiles; ++t) {
        const int a = t * bm, b = (a + bm < P.total ? a + bm : P.total) - 1;
        const int na = a / P.HW, ya = (a % P.HW) / d.W, xa = (a % P.HW) % d.W;
        const int nb = b / P.HW, yb = (b % P.HW) / d.W, xb = (b % P.HW) % d.W;
        const int rows = (nb * P.VR + yb + 1) - (na * P.VR + ya);   // virtual rows between the window's first and the last pixel's
        const int v = (rows + 1) * P.W1 + xb - xa + 3;
        if (v > NV) NV = v;
    }
    if (NV > 512) return false;
    P.NV = NV;
    P.pl = NV * 16;
    P.NIW = (NV + 63) / 64;
    P.mHW = smagic(P.HW); P.mW = smagic(d.W); P.mW1 = smagic(P.W1); P.mVR = smagic(P.VR);
    // exactness of the magic divisions (numerator * divisor < 2^32) and 31-bit byte offsets
    if ((long)(P.HW + bm) * P.HW >= (1l << 32) || (long)P.HW * d.W >= (1l << 32)) return false;
    if ((long)(d.N + 1) * P.VR * P.VR >= (1l << 32)) return false;
    {   // images a tile's window may touch (a 256-pixel tile of a small map spans many): (n - n0) * imgB stays below 2^31
        const long span = (long)(bm) / P.HW + 2;
        if (span * d.Cin * P.HW * 4 >= (1l << 31)) return false;
    }
    if (P.HW < 32) return false;
    // operands of the kernel's 24-bit index multiplies (smul)
    if (P.HW >= (1 << 24) || (long)(d.N + 8) * P.VR >= (1l << 24) || (long)(d.N + 8) * (d.Cout / 4 + d.out_ctot) + d.out_coff >= (1l << 24)) return false;
    if ((size_t)P.nN * P.nChunks * swch(P.NTW) * 1024 >= (1ull << 31)) return false;
    return true;
}

template <int NTW, bool NCHW, int NPT>
int convs_launch(const void* xs, const void* wpk, const float* shift, const float* res, float* outf, void* outs, const SPlan& P,
                 hipStream_t st) {
    auto kern = convs_kernel<NTW, NCHW, NPT>;
    size_t need = (size_t)4 * P.pl + swch(NTW) * 1024;
    if (NCHW && need < (size_t)NTW * 16 * (64 * NPT + 4) * 4) need = (size_t)NTW * 16 * (64 * NPT + 4) * 4;   // the output tile of the epilogue
    OTP_ALLOW_BIG_LDS(kern, need);
    hipLaunchKernelGGL(kern, dim3(8 * P.tpx * P.nN), dim3(256), need, st, static_cast<const unsigned char*>(xs),
                       static_cast<const u32x4*>(wpk), shift, res, outf, static_cast<u32x4*>(outs), P);
    return otp_launch_status();
}

}  // namespace

#ifdef OTP_CONVS_TIMING
extern "C" int otp_convs_read_stamps(void* host_out, size_t bytes) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(otp_convs_stamps), bytes) == hipSuccess ? OTP_OK : OTP_ERR_LAUNCH;
}
#endif

extern "C" size_t otp_s8_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || C % 8) return 0;
    return (size_t)N * C * H * W * 4;
}

extern "C" int otp_s8_pack(const void* in, void* out, void* out_c4, int N, int C, int H, int W, int in_ctot, int in_coff,
                           void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || in_ctot < in_coff + C || in_coff < 0) return OTP_ERR_BAD_ARG;
    if (C % 8 || ((H * W) & 3) ||
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(out_c4)) & 15))
        return OTP_ERR_UNSUPPORTED;
    const size_t items = (size_t)N * (C / 8) * (H * W);
    const int grid = (int)((items + 255) / 256 > 16384 ? 16384 : (items + 255) / 256);
    hipLaunchKernelGGL(s8_pack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float*>(in),
                       static_cast<u32x4*>(out), static_cast<float*>(out_c4), N, C, H * W, in_ctot, in_coff, otp_range_word());
    return otp_launch_status();
}

extern "C" int otp_s8_upsample_add(const void* const* lows, const int* factors, int nlow, const void* res, void* out_nchw,
                                   void* out_s8, void* out_c4, int N, int C, int Hh, int Wh, int relu, int res_ctot, int res_coff,
                                   int out_ctot, int out_coff, void* stream) {
    return otp_s8_upsample_add_ex(lows, factors, nlow, res, 0, out_nchw, out_s8, out_c4, N, C, Hh, Wh, relu, res_ctot, res_coff,
                                  out_ctot, out_coff, stream);
}

extern "C" int otp_s8_upsample_add_ex(const void* const* lows, const int* factors, int nlow, const void* res, int res_layout,
                                      void* out_nchw, void* out_s8, void* out_c4, int N, int C, int Hh, int Wh, int relu,
                                      int res_ctot, int res_coff, int out_ctot, int out_coff, void* stream) {
    if (!lows || !factors || !res || !out_s8 || nlow < 1 || nlow > 3 || N <= 0 || C <= 0 || Hh <= 0 || Wh <= 0)
        return OTP_ERR_BAD_ARG;                         // (out_c4 may be NULL: a consumer that reads its residual as S8 records)
    if (res_layout != 0 && res_layout != 1) return OTP_ERR_BAD_ARG;
    const int res_s8 = res_layout;                      // 1: `res` is the S8 image of the (N, C, Hh, Wh) residual
    if (Wh % 4 || C % 8 || (!res_s8 && res_ctot < res_coff + C) || (out_nchw && out_ctot < out_coff + C)) return OTP_ERR_UNSUPPORTED;
    S8Up U{};
    U.n = nlow;
    for (int k = 0; k < nlow; ++k) {
        const int f = factors[k];
        if (!lows[k]) return OTP_ERR_BAD_ARG;
        if (f < 2 || (f & (f - 1)) || Hh % f || Wh % f) return OTP_ERR_UNSUPPORTED;
        U.low[k] = static_cast<const float*>(lows[k]);
        U.f[k] = f;
    }
    if ((reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out_nchw) | reinterpret_cast<uintptr_t>(out_s8) |
         reinterpret_cast<uintptr_t>(out_c4)) & 15)
        return OTP_ERR_UNSUPPORTED;
    if (nlow >= 3) {
        const size_t items4 = (size_t)N * (C / 8) * (Hh * Wh / 4);
        const int grid4 = (int)((items4 + 255) / 256 > 8192 ? 8192 : (items4 + 255) / 256);
        hipLaunchKernelGGL(s8_upsample_add4_kernel, dim3(grid4), dim3(256), 0, static_cast<hipStream_t>(stream), U,
                           static_cast<const float*>(res), static_cast<float*>(out_nchw), static_cast<u32x4*>(out_s8),
                           static_cast<float*>(out_c4), N, C, Hh, Wh, relu, res_ctot, res_coff, out_ctot, out_coff, res_s8, otp_range_word());
        return otp_launch_status();
    }
    const size_t items = (size_t)N * (C / 8) * (Hh * Wh);
    const int grid = (int)((items + 255) / 256 > 16384 ? 16384 : (items + 255) / 256);
    hipLaunchKernelGGL(s8_upsample_add_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), U,
                       static_cast<const float*>(res), static_cast<float*>(out_nchw), static_cast<u32x4*>(out_s8),
                       static_cast<float*>(out_c4), N, C, Hh, Wh, relu, res_ctot, res_coff, out_ctot, out_coff, res_s8, otp_range_word());
    return otp_launch_status();
}

extern "C" int otp_s8_unpack(const void* in, void* out, int N, int C, int H, int W, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return OTP_ERR_BAD_ARG;
    if (C % 8) return OTP_ERR_UNSUPPORTED;
    const size_t items = (size_t)N * (C / 8) * H * W;
    const int grid = (int)((items + 255) / 256 > 8192 ? 8192 : (items + 255) / 256);
    hipLaunchKernelGGL(s8_unpack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const u32x4*>(in),
                       static_cast<float*>(out), N, C, H * W);
    return otp_launch_status();
}

extern "C" int otp_c4_unpack(const void* in, void* out, int N, int C, int H, int W, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return OTP_ERR_BAD_ARG;
    if (C % 4) return OTP_ERR_UNSUPPORTED;
    const size_t items = (size_t)N * (C / 4) * H * W;
    const int grid = (int)((items + 255) / 256 > 8192 ? 8192 : (items + 255) / 256);
    hipLaunchKernelGGL(c4_unpack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const f32x4*>(in),
                       static_cast<float*>(out), N, C, H * W);
    return otp_launch_status();
}

extern "C" int otp_conv3x3_s8_supported(const otp_conv_desc* desc) {
    if (!desc) return 0;
    SPlan P{};
    return convs_plan(*desc, P) ? 1 : 0;
}

extern "C" size_t otp_conv3x3_s8_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cin % 16) return 0;
    const int NTW = s8_ntw(Cout), nN = ((Cout + 15) / 16 + NTW - 1) / NTW;
    return (size_t)nN * (Cin / 16) * swch(NTW) * 1024;
}

extern "C" int otp_conv3x3_s8_pack_weight(const void* weight, const void* scale, void* wpacked, int Cout, int Cin, void* stream) {
    if (!weight || !wpacked || Cout <= 0 || Cin <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_conv3x3_s8_weight_bytes(Cout, Cin)) return OTP_ERR_UNSUPPORTED;
    const int NTW = s8_ntw(Cout), nN = ((Cout + 15) / 16 + NTW - 1) / NTW, nChunks = Cin / 16;
    const int total = nN * nChunks * SKS * NTW * 64;
    hipLaunchKernelGGL(s8_wpack_kernel, dim3(otp_ceil_div(total, 256) > 2048 ? 2048 : otp_ceil_div(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(weight), static_cast<const float*>(scale),
                       static_cast<u32x4*>(wpacked), Cout, Cin, NTW, nN, nChunks);
    return otp_launch_status();
}

extern "C" int otp_conv3x3_s8(const void* in_s8, const void* wpacked, const void* shift, const void* res_c4, void* out_f32,
                              int out_f32_layout, void* out_s8, const otp_conv_desc* desc, void* stream) {
    if (!in_s8 || !wpacked || !desc || (!out_f32 && !out_s8)) return OTP_ERR_BAD_ARG;
    const otp_conv_desc& d = *desc;
    if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0) return OTP_ERR_BAD_ARG;
    if (out_f32 && out_f32_layout != OTP_S8_F32_C4 && out_f32_layout != OTP_S8_F32_NCHW) return OTP_ERR_BAD_ARG;
    if (out_f32 && out_f32_layout == OTP_S8_F32_NCHW && d.out_ctot < d.out_coff + d.Cout) return OTP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in_s8) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out_f32) |
         reinterpret_cast<uintptr_t>(out_s8) | reinterpret_cast<uintptr_t>(res_c4) | reinterpret_cast<uintptr_t>(shift)) & 15)
        return OTP_ERR_UNSUPPORTED;
    SPlan P{};
    if (!convs_plan(d, P)) return OTP_ERR_UNSUPPORTED;
    P.rflag = otp_range_word();
    P.f32_mode = out_f32 ? (out_f32_layout == OTP_S8_F32_C4 ? S_F32_C4 : S_F32_NCHW) : S_F32_NONE;
    auto st = static_cast<hipStream_t>(stream);
    auto fs = static_cast<const float*>(shift);
    auto fr = static_cast<const float*>(res_c4);
    auto fo = static_cast<float*>(out_f32);
#define OTP_CONVS_GO(NTW_, NCHW_, NPT_) return convs_launch<NTW_, NCHW_, NPT_>(in_s8, wpacked, fs, fr, fo, out_s8, P, st)
    const bool nchw = P.f32_mode == S_F32_NCHW;
    if (P.NPT == 2) {
        if (P.NTW == 2) { if (nchw) OTP_CONVS_GO(2, true, 2); OTP_CONVS_GO(2, false, 2); }
        if (nchw) OTP_CONVS_GO(3, true, 2);
        OTP_CONVS_GO(3, false, 2);
    }
    if (P.NTW == 2) { if (nchw) OTP_CONVS_GO(2, true, 4); OTP_CONVS_GO(2, false, 4); }
    if (nchw) OTP_CONVS_GO(3, true, 4);
    OTP_CONVS_GO(3, false, 4);
#undef OTP_CONVS_GO
}
