// Pointwise (1x1) convolutions Cin -> Cout of HRNet's layer1 Bottlenecks (reference model/HRNet.py:551-571: conv1 256 -> 64,
// conv3 64 -> 256 + residual, the folded shortcut 128 -> 256; BatchNorm folded into scale / shift, ReLU) and of the fuse
// layers' up-sampling paths (:440-455: 96 -> 48, 192 -> 48, 192 -> 96; any Cin <= 256, padded with zero weights to 64 / 128 / 256) with split-bf16
// ("bf16x3") products - the same operator as csrc/convx.hip in its 1x1 mode, built like csrc/densex.hip instead: these
// layers move 0.7 GB each at cfg2 and do 9 MFLOP per pixel, so they are bound by their HBM streams, and the implicit-GEMM
// kernel (window through the LDS, one barrier per 32-channel chunk) ran them at a third of the HBM rate.
// Register-resident input: a wave owns 32 pixels and holds their Cin channels as split B-operand fragments (lane (pixel
// pair n, kq): channels 32 ks + 8 kq .. + 7); the weights stream through the LDS in blocks of 8 / KS sixteen-row output
// tiles (16 KB: LDS-DMA, double buffered, one barrier per block that waits for the DMA alone - the result stores stay in flight);
// a tile's 16 x 32 result is scaled, shifted, added to the residual, clamped and stored as soon as its 6 KS MFMAs are done.
// Tensors are fp32 NCHW channel slices: (ctot, coff) per operand, as everywhere in the engine.
#include "common.h"

namespace {

typedef otp_x3x8 h16x8;              // 8 operand pieces of the split products (common.h: IEEE half since round 4)
typedef otp_x3x2 h16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void px_split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 a = {v[2 * i], v[2 * i + 1]};
        const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, h16x2));
        const f32x2 af = otp_x3_widen(hb);
        h[i] = hb;
        l[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a - af, h16x2));
    }
    hi = __builtin_bit_cast(h16x8, (u32x4){h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(h16x8, (u32x4){l[0], l[1], l[2], l[3]});
}

constexpr int px_ks(int CIN) { return CIN / 32; }
constexpr int px_mpb(int CIN) { return 8 / px_ks(CIN); }                // 16-row output tiles per weight block
// a weight block: A fragments [tile][ks][hi, lo][1 KB] = 16 KB for every Cin (four whole passes of the 256 threads); scale and
// shift (Cout <= 256 floats each) follow the blocks in the packed image and stay in the LDS for the whole kernel
constexpr int px_block_bytes(int CIN) { return px_mpb(CIN) * px_ks(CIN) * 2048; }
// S8 output (records of 8 consecutive channels): tiles are processed in pairs, so at least two per block (32 KB at Cin = 256)
constexpr int px_mpb_s8(int CIN) { return px_mpb(CIN) < 2 ? 2 : px_mpb(CIN); }
constexpr int PX_MAX_COUT = 256;

__global__ void pointx_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                   unsigned char* __restrict__ packed, int Cin, int CinP, int Cout, int nblk, int s8) {
    // (CinP: Cin padded to 64 / 128 / 256; s8: the row order of the S8-output kernel - rows (4 kq + i) of the tile pair (2 p, 2 p + 1)
    // are channels 32 p + 8 kq + i and 32 p + 8 kq + 4 + i, so that a lane ends up with 8 consecutive channels of its pixels)
    const int KS = CinP / 32, MPB = s8 && 8 / KS < 2 ? 2 : 8 / KS, units = MPB * KS * 128;   // 16-byte units per block
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nblk * units) {
        const int blk = idx / units, u = idx - blk * units;
        const int frag = u >> 6, lane = u & 63, m = frag / (KS * 2), f2 = frag - m * KS * 2, ks = f2 >> 1;
        const int gt = blk * MPB + m, r16 = lane & 15, kq = lane >> 4;
        const int row = s8 ? 32 * (gt >> 1) + 8 * (r16 >> 2) + 4 * (gt & 1) + (r16 & 3) : 16 * gt + r16;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 32 * ks + 8 * kq + j;
            v[j] = (row < Cout && c < Cin) ? w[(size_t)row * Cin + c] : 0.f;
        }
        h16x8 hi, lo;
        px_split8(v, hi, lo);
        reinterpret_cast<u32x4*>(packed)[idx] = __builtin_bit_cast(u32x4, (f2 & 1) ? lo : hi);
    } else if (idx < nblk * units + 2 * PX_MAX_COUT / 4) {                  // [scale 256][shift 256] floats, by channel
        const int q = idx - nblk * units;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * q + i, c = e & (PX_MAX_COUT - 1);
            v[i] = c < Cout ? (e < PX_MAX_COUT ? (scale ? scale[c] : 1.f) : (shift ? shift[c] : 0.f)) : 0.f;
        }
        reinterpret_cast<u32x4*>(packed)[idx] = (u32x4){__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]),
                                                        __builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3])};
    }
}

// one weight block global -> LDS with the LDS-DMA (256 threads, four passes): unit u (16 bytes) lands at lds + 16 u
template <int BLKB>
__device__ __forceinline__ void px_stage(const unsigned char* __restrict__ src, unsigned char* lds) {
    constexpr int NST = BLKB / 16 / 256;
    static_assert(NST * 256 * 16 == BLKB, "whole passes");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int u0 = i * 256 + wave * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(u0 + lane) * 16),
                                         (__attribute__((address_space(3))) void*)(lds + u0 * 16), 16, 0, 0);
    }
}

struct PxArgs {
    const float* x;
    const unsigned char* packed;
    const float* res;
    float* out;
    int T, tiles_per_b, Cin, Cout, nblk, relu;
    int x_ctot, x_coff, r_ctot, r_coff, o_ctot, o_coff;
    unsigned* rflag;                                                         // range-guard word (common.h)
};

template <int CIN, bool RES>
__global__ __launch_bounds__(256, CIN <= 128 ? 3 : 2) void pointx_kernel(PxArgs A) {
    constexpr int KS = px_ks(CIN), MPB = px_mpb(CIN), BLKB = px_block_bytes(CIN);
    // (ONE LDS object: with scale / shift in an array of their own hipcc 7.2 drains vmcnt in front of the first fragment read
    // after every DMA issue)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BLKB + 2 * PX_MAX_COUT * 4];
    float* ss = reinterpret_cast<float*>(lds + 2 * BLKB);                    // scale[256], shift[256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int b = blockIdx.x / A.tiles_per_b, tile = blockIdx.x - b * A.tiles_per_b;
    const int T = A.T, tok = tile * 128 + wave * 32 + 2 * n;
    const bool valid = tok < T;
    px_stage<BLKB>(A.packed, lds);
    if (tid < 2 * PX_MAX_COUT / 4)
        reinterpret_cast<f32x4*>(ss)[tid] = reinterpret_cast<const f32x4*>(A.packed + (size_t)A.nblk * BLKB)[tid];
    const float* __restrict__ x = A.x + ((size_t)b * A.x_ctot + A.x_coff) * T + (valid ? tok : T - 2);
    h16x8 Xh[KS][2], Xl[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float v0[8], v1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 32 * ks + 8 * kq + j;                              // channels past Cin (Cin padded to the template's CIN):
            const bool live = c < A.Cin;                                     // zero weights, and nothing is read
            const f32x2 v = *reinterpret_cast<const f32x2*>(x + (size_t)(live ? c : 0) * T);
            v0[j] = live ? v.x : 0.f;
            v1[j] = live ? v.y : 0.f;
        }
        px_split8(v0, Xh[ks][0], Xl[ks][0]);
        px_split8(v1, Xh[ks][1], Xl[ks][1]);
    }
    const unsigned plane = (unsigned)((size_t)A.Cout * T * sizeof(float));
    const otp_rsrc ro = make_rsrc32(A.out + ((size_t)b * A.o_ctot + A.o_coff) * T, plane);
    const otp_rsrc rr = make_rsrc32(RES ? A.res + ((size_t)b * A.r_ctot + A.r_coff) * T : A.out, RES ? plane : 0u);
    const float lo_clamp = A.relu ? 0.f : -__builtin_inff();
    bool bad = false;                                  // range guard (common.h): results tested before the clamp swallows a NaN
    __syncthreads();                                   // weight block 0 and scale / shift landed
    // Every wave issues the SAME vector-memory instructions per block - residual loads, the DMA of the next block, 4 MPB stores,
    // lanes without a pixel or channel masked by an out-of-range offset - so the barrier can wait for the DMA alone
    // (`s_waitcnt vmcnt(4 MPB)`: all but the stores, which stay in flight under the next block).  The compiler waits for
    // EVERYTHING in front of the first use of a residual value (an LDS-DMA is pending), so all of a block's MFMAs come first.
#pragma unroll 1
    for (int blk = 0; blk < A.nblk; ++blk) {
        int voff[MPB][4];
        f32x2 r[MPB][4];
#pragma unroll
        for (int m = 0; m < MPB; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 16 * (blk * MPB + m) + 4 * kq + i;
                voff[m][i] = (valid && c < A.Cout) ? (c * T + tok) * 4 : -16;
                if (RES) r[m][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rr, voff[m][i], 0, 0));
            }
        asm volatile("" ::: "memory");
        // (the last trip re-stages block 0, which nobody reads: every wave issues the same instructions on every trip)
        px_stage<BLKB>(A.packed + (size_t)(blk + 1 < A.nblk ? blk + 1 : 0) * BLKB, lds + ((blk + 1) & 1) * BLKB);
        asm volatile("" ::: "memory");
        const unsigned char* P = lds + (blk & 1) * BLKB;
        f32x4 acc[MPB][2];
#pragma unroll
        for (int m = 0; m < MPB; ++m) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const h16x8 ah = *reinterpret_cast<const h16x8*>(P + ((m * KS + ks) * 2) * 1024 + lane * 16);
                const h16x8 al = *reinterpret_cast<const h16x8*>(P + ((m * KS + ks) * 2 + 1) * 1024 + lane * 16);
                acc0 = OTP_X3_MFMA(al, Xh[ks][0], acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(al, Xh[ks][1], acc1, 0, 0, 0);
                acc0 = OTP_X3_MFMA(ah, Xl[ks][0], acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(ah, Xl[ks][1], acc1, 0, 0, 0);
                acc0 = OTP_X3_MFMA(ah, Xh[ks][0], acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(ah, Xh[ks][1], acc1, 0, 0, 0);
            }
            acc[m][0] = acc0;
            acc[m][1] = acc1;
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m = 0; m < MPB; ++m) {
            const int c0 = (16 * (blk * MPB + m) + 4 * kq) & (PX_MAX_COUT - 1);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + c0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + PX_MAX_COUT + c0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 v = {acc[m][0][i] * sc[i] + sh[i], acc[m][1][i] * sc[i] + sh[i]};
                if (RES) v += r[m][i];
                bad |= otp_out_of_range(v.x);
                bad |= otp_out_of_range(v.y);
                v.x = fmaxf(v.x, lo_clamp);
                v.y = fmaxf(v.y, lo_clamp);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ro, voff[m][i], 0, 0);
            }
        }
        static_assert(4 * MPB <= 16, "stores after the DMA");
        if (MPB == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // this wave's part of the next block has landed
        else if (MPB == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (after the last trip: nothing may land in the LDS
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                 // once the wave has ended); the stores fly on
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    otp_range_report(A.rflag, bad, OTP_RANGE_POINTX);
}

// The same convolution writing the S8 image of its result ([B][Cout / 8][hi | lo][T] 16-byte records of 8 bf16: the operand
// records of csrc/convs.hip) instead of fp32 NCHW: a Bottleneck's conv1 feeding its 3x3 conv2 (model/HRNet.py:551-571).  Tiles
// are processed in pairs whose packed rows are permuted (pointx_pack_kernel, s8 = 1) so that lane (pixel pair n, kq) ends up with
// channels 32 p + 8 kq .. + 7 of its two pixels: one hi and one lo record per pixel, 16 lanes = 512 contiguous bytes per store.
// RES: + an fp32 NCHW residual before the activation (a Bottleneck's conv3, HRNet.py:566-571, whose consumers read S8 records:
// layer1's last block in front of transition1)
template <int CIN, bool RES = false>
__global__ __launch_bounds__(256, CIN <= 128 ? 3 : 2) void pointx_s8_kernel(PxArgs A) {
    constexpr int KS = px_ks(CIN), MPB = px_mpb_s8(CIN), BLKB = MPB * KS * 2048;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BLKB + 2 * PX_MAX_COUT * 4];
    float* ss = reinterpret_cast<float*>(lds + 2 * BLKB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int b = blockIdx.x / A.tiles_per_b, tile = blockIdx.x - b * A.tiles_per_b;
    const int T = A.T, tok = tile * 128 + wave * 32 + 2 * n;
    const bool valid = tok < T;
    px_stage<BLKB>(A.packed, lds);
    if (tid < 2 * PX_MAX_COUT / 4)
        reinterpret_cast<f32x4*>(ss)[tid] = reinterpret_cast<const f32x4*>(A.packed + (size_t)A.nblk * BLKB)[tid];
    const float* __restrict__ x = A.x + ((size_t)b * A.x_ctot + A.x_coff) * T + (valid ? tok : T - 2);
    h16x8 Xh[KS][2], Xl[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float v0[8], v1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(x + (size_t)(32 * ks + 8 * kq + j) * T);
            v0[j] = v.x;
            v1[j] = v.y;
        }
        px_split8(v0, Xh[ks][0], Xl[ks][0]);
        px_split8(v1, Xh[ks][1], Xl[ks][1]);
    }
    const unsigned plane = (unsigned)((size_t)A.Cout * T * sizeof(float));         // an S8 image is 4 bytes per element too
    const otp_rsrc ro = make_rsrc32(A.out + (size_t)b * A.Cout * T, plane);
    const otp_rsrc rr = make_rsrc32(RES ? A.res + ((size_t)b * A.r_ctot + A.r_coff) * T : A.out, RES ? plane : 0u);
    const float lo_clamp = A.relu ? 0.f : -__builtin_inff();
    bool bad = false;                                  // range guard (common.h)
    __syncthreads();
#pragma unroll 1
    for (int blk = 0; blk < A.nblk; ++blk) {
        px_stage<BLKB>(A.packed + (size_t)(blk + 1 < A.nblk ? blk + 1 : 0) * BLKB, lds + ((blk + 1) & 1) * BLKB);
        asm volatile("" ::: "memory");
        const unsigned char* P = lds + (blk & 1) * BLKB;
        f32x4 acc[MPB][2];
        // residual of the lane's two pixels, 8 channels per tile pair: in flight under the MFMAs of the block
        f32x2 rv[RES ? MPB / 2 : 1][8];
        if constexpr (RES) {
#pragma unroll
            for (int m = 0; m < MPB; m += 2) {
                const int pr = (blk * MPB + m) >> 1, c8 = 32 * pr + 8 * kq;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    rv[m >> 1][e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rr, (valid && c8 < A.Cout) ? ((c8 + e) * T + tok) * 4 : -16, 0, 0));
            }
        }
#pragma unroll
        for (int m = 0; m < MPB; ++m) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const h16x8 ah = *reinterpret_cast<const h16x8*>(P + ((m * KS + ks) * 2) * 1024 + lane * 16);
                const h16x8 al = *reinterpret_cast<const h16x8*>(P + ((m * KS + ks) * 2 + 1) * 1024 + lane * 16);
                acc0 = OTP_X3_MFMA(al, Xh[ks][0], acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(al, Xh[ks][1], acc1, 0, 0, 0);
                acc0 = OTP_X3_MFMA(ah, Xl[ks][0], acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(ah, Xl[ks][1], acc1, 0, 0, 0);
                acc0 = OTP_X3_MFMA(ah, Xh[ks][0], acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(ah, Xh[ks][1], acc1, 0, 0, 0);
            }
            acc[m][0] = acc0;
            acc[m][1] = acc1;
        }
#pragma unroll
        for (int m = 0; m < MPB; m += 2) {
            const int pr = (blk * MPB + m) >> 1, c8 = 32 * pr + 8 * kq, g = 4 * pr + kq;        // channel group of this lane
            const f32x4 sc0 = *reinterpret_cast<const f32x4*>(ss + (c8 & (PX_MAX_COUT - 1)));
            const f32x4 sc1 = *reinterpret_cast<const f32x4*>(ss + ((c8 + 4) & (PX_MAX_COUT - 1)));
            const f32x4 sh0 = *reinterpret_cast<const f32x4*>(ss + PX_MAX_COUT + (c8 & (PX_MAX_COUT - 1)));
            const f32x4 sh1 = *reinterpret_cast<const f32x4*>(ss + PX_MAX_COUT + ((c8 + 4) & (PX_MAX_COUT - 1)));
            const bool live = valid && c8 < A.Cout;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r0 = RES ? rv[RES ? m >> 1 : 0][i][h] : 0.f, r1 = RES ? rv[RES ? m >> 1 : 0][4 + i][h] : 0.f;
                    const float u0 = acc[m][h][i] * sc0[i] + sh0[i] + r0, u1 = acc[m + 1][h][i] * sc1[i] + sh1[i] + r1;
                    bad |= otp_out_of_range(u0);
                bad |= otp_out_of_range(u1);
                    v[i] = fmaxf(u0, lo_clamp);
                    v[4 + i] = fmaxf(u1, lo_clamp);
                }
                h16x8 hi, lo;
                px_split8(v, hi, lo);
                const int o = live ? ((g * 2) * T + tok + h) * 16 : -16;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), ro, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), ro, live ? o + T * 16 : -16, 0, 0);
            }
        }
        static_assert(2 * MPB == 4 || 2 * MPB == 8, "stores after the DMA");
        if (MPB == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    otp_range_report(A.rflag, bad, OTP_RANGE_POINTX);
}

bool px_cin_ok(int Cin) { return Cin >= 16 && Cin <= 256; }
int px_cin_pad(int Cin) { return Cin <= 64 ? 64 : (Cin <= 128 ? 128 : 256); }              // the kernel instantiation that holds it

}  // namespace

extern "C" int otp_pointwise_x3_supported(int Cin, int Cout, int T) {
    return (px_cin_ok(Cin) && Cout > 0 && Cout <= PX_MAX_COUT && T >= 2 && T % 2 == 0 && (size_t)Cout * T * 4 < (1ull << 31)) ? 1 : 0;
}

extern "C" int otp_pointwise_x3_s8_supported(int Cin, int Cout, int T) {
    return ((Cin == 64 || Cin == 256) && Cout > 0 && Cout <= PX_MAX_COUT && Cout % 32 == 0 && T >= 4 && T % 4 == 0 &&
            (size_t)Cout * T * 4 < (1ull << 31)) ? 1 : 0;
}

extern "C" size_t otp_pointwise_x3_s8_weight_bytes(int Cin, int Cout) {
    if (!(Cin == 64 || Cin == 256) || Cout <= 0 || Cout > PX_MAX_COUT || Cout % 32) return 0;
    const int KS = Cin / 32, MPB = 8 / KS < 2 ? 2 : 8 / KS, MT = Cout / 16, nblk = (MT + MPB - 1) / MPB;
    return (size_t)nblk * MPB * KS * 2048 + 2 * PX_MAX_COUT * sizeof(float);
}

extern "C" size_t otp_pointwise_x3_weight_bytes(int Cin, int Cout) {
    if (!px_cin_ok(Cin) || Cout <= 0 || Cout > PX_MAX_COUT) return 0;
    const int KS = px_cin_pad(Cin) / 32, MPB = 8 / KS, MT = (Cout + 15) / 16, nblk = (MT + MPB - 1) / MPB;
    return (size_t)nblk * 16384 + 2 * PX_MAX_COUT * sizeof(float);
}

namespace {
int px_pack(const void* w, const void* scale, const void* shift, void* packed, int Cin, int Cout, int s8, void* stream) {
    if (!w || !packed) return OTP_ERR_BAD_ARG;
    const size_t bytes = s8 ? otp_pointwise_x3_s8_weight_bytes(Cin, Cout) : otp_pointwise_x3_weight_bytes(Cin, Cout);
    if (!bytes) return OTP_ERR_UNSUPPORTED;
    const int CinP = px_cin_pad(Cin), KS = CinP / 32, MPB = s8 && 8 / KS < 2 ? 2 : 8 / KS, MT = (Cout + 15) / 16;
    const int nblk = (MT + MPB - 1) / MPB, total = (int)(bytes / 16);
    hipLaunchKernelGGL(pointx_pack_kernel, dim3(otp_ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w), static_cast<const float*>(scale), static_cast<const float*>(shift),
                       static_cast<unsigned char*>(packed), Cin, CinP, Cout, nblk, s8);
    return otp_launch_status();
}
}  // namespace

extern "C" int otp_pointwise_x3_pack(const void* w, const void* scale, const void* shift, void* packed, int Cin, int Cout,
                                     void* stream) {
    return px_pack(w, scale, shift, packed, Cin, Cout, 0, stream);
}

extern "C" int otp_pointwise_x3_s8_pack(const void* w, const void* scale, const void* shift, void* packed, int Cin, int Cout,
                                        void* stream) {
    return px_pack(w, scale, shift, packed, Cin, Cout, 1, stream);
}

extern "C" int otp_pointwise_x3(const void* x, const void* packed, const void* res, void* out, int B, int Cin, int Cout, int T,
                                int x_ctot, int x_coff, int res_ctot, int res_coff, int out_ctot, int out_coff, int relu,
                                void* stream) {
    if (!x || !packed || !out || B <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_pointwise_x3_supported(Cin, Cout, T)) return OTP_ERR_UNSUPPORTED;
    if (x_coff < 0 || x_coff + Cin > x_ctot || out_coff < 0 || out_coff + Cout > out_ctot ||
        (res && (res_coff < 0 || res_coff + Cout > res_ctot)))
        return OTP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out)) & 7 ||
        reinterpret_cast<uintptr_t>(packed) & 15)
        return OTP_ERR_BAD_ARG;
    const int CinP = px_cin_pad(Cin), KS = CinP / 32, MPB = 8 / KS, MT = (Cout + 15) / 16;
    PxArgs a;
    a.x = static_cast<const float*>(x);
    a.packed = static_cast<const unsigned char*>(packed);
    a.res = static_cast<const float*>(res);
    a.out = static_cast<float*>(out);
    a.T = T, a.tiles_per_b = otp_ceil_div(T, 128), a.Cin = Cin, a.Cout = Cout, a.nblk = (MT + MPB - 1) / MPB, a.relu = relu ? 1 : 0;
    a.x_ctot = x_ctot, a.x_coff = x_coff, a.r_ctot = res_ctot, a.r_coff = res_coff, a.o_ctot = out_ctot, a.o_coff = out_coff;
    a.rflag = otp_range_word();
    const dim3 grid((unsigned)(B * a.tiles_per_b));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define OTP_PX_GO(CIN_)                                                                                      \
    {                                                                                                       \
        if (res) hipLaunchKernelGGL((pointx_kernel<CIN_, true>), grid, dim3(256), 0, st, a);                \
        else hipLaunchKernelGGL((pointx_kernel<CIN_, false>), grid, dim3(256), 0, st, a);                   \
    }
    if (CinP == 64) OTP_PX_GO(64)
    else if (CinP == 128) OTP_PX_GO(128)
    else OTP_PX_GO(256)
#undef OTP_PX_GO
    return otp_launch_status();
}

extern "C" int otp_pointwise_x3_s8(const void* x, const void* packed, void* out_s8, int B, int Cin, int Cout, int T, int x_ctot,
                                   int x_coff, int relu, void* stream) {
    return otp_pointwise_x3_s8_res(x, packed, nullptr, out_s8, B, Cin, Cout, T, x_ctot, x_coff, 0, 0, relu, stream);
}

extern "C" int otp_pointwise_x3_s8_res(const void* x, const void* packed, const void* res, void* out_s8, int B, int Cin, int Cout,
                                       int T, int x_ctot, int x_coff, int r_ctot, int r_coff, int relu, void* stream) {
    if (!x || !packed || !out_s8 || B <= 0) return OTP_ERR_BAD_ARG;
    if (res && (r_coff < 0 || r_coff + Cout > r_ctot || (reinterpret_cast<uintptr_t>(res) & 7))) return OTP_ERR_BAD_ARG;
    if (!otp_pointwise_x3_s8_supported(Cin, Cout, T)) return OTP_ERR_UNSUPPORTED;
    if (x_coff < 0 || x_coff + Cin > x_ctot) return OTP_ERR_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(x) & 7 || (reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out_s8)) & 15)
        return OTP_ERR_BAD_ARG;
    const int KS = Cin / 32, MPB = 8 / KS < 2 ? 2 : 8 / KS, MT = Cout / 16;
    PxArgs a;
    a.x = static_cast<const float*>(x);
    a.packed = static_cast<const unsigned char*>(packed);
    a.res = static_cast<const float*>(res);
    a.out = static_cast<float*>(out_s8);
    a.T = T, a.tiles_per_b = otp_ceil_div(T, 128), a.Cin = Cin, a.Cout = Cout, a.nblk = (MT + MPB - 1) / MPB, a.relu = relu ? 1 : 0;
    a.x_ctot = x_ctot, a.x_coff = x_coff, a.r_ctot = r_ctot, a.r_coff = r_coff, a.o_ctot = Cout, a.o_coff = 0;
    a.rflag = otp_range_word();
    const dim3 grid((unsigned)(B * a.tiles_per_b));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (res) {
        if (Cin == 64) hipLaunchKernelGGL((pointx_s8_kernel<64, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pointx_s8_kernel<256, true>), grid, dim3(256), 0, st, a);
    } else {
        if (Cin == 64) hipLaunchKernelGGL((pointx_s8_kernel<64, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pointx_s8_kernel<256, false>), grid, dim3(256), 0, st, a);
    }
    return otp_launch_status();
}
