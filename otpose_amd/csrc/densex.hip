// The C -> C pointwise projections of MaskedMHCA (reference model/blocks.py:383-386 applied at :417-419 and :450) and
// its attention front end (:406-419 with the modules of :359-386) with split-bf16 ("bf16x3") products on the bf16 matrix
// cores - the same operators as csrc/dense.hip (otp_dense_cc, otp_qkv_front): fp32 storage, fp32 accumulation, depthwise
// conv / LayerNorm / scale / shift in fp32; every product of the projection is a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on
// v_mfma_f32_16x16x32_bf16 (csrc/convx.hip explains the split).  With the matrix work cut 4.9x both kernels are bound by
// their HBM streams (8-12 bytes / token / channel).
// Register-resident input as in csrc/mlpx.hip: a wave owns 32 tokens and holds their C channels as split B-operand
// fragments (lane (token pair n, kq): channels 32 ks + 8 kq .. + 7); the weights stream through the LDS one 16-row output
// block at a time (LDS-DMA, double buffered, one barrier per block); a block's 16 x 32 result is scaled, shifted, added to
// the residual and stored as soon as its 30 MFMAs are done.
//
// This file is compiled twice.  densex.o: IEEE-half pieces (csrc/common.h) for the forward's O(1) activations.  densex_grad.o
// (densex_grad.hip: -DOTP_X3_BF16, entry points suffixed _bf16p): bfloat16 pieces for operands whose magnitude is not known -
// the GRADIENTS the training backward sends through the same projection (dx = W^T dy, otpose_amd/train_ops.py): a half piece
// flushes 1e-8 to zero and holds 1e-5 to 8 bits, a bfloat16 pair keeps 16-17 bits at any magnitude.
#include "common.h"
#ifndef OTP_ENTRY
#define OTP_ENTRY(name) name
#endif

namespace {

typedef otp_x3x8 h16x8;              // 8 operand pieces of the split products (common.h: IEEE half since round 4)
typedef otp_x3x2 h16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dx_split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
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

__device__ __forceinline__ float dx_kslot_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

constexpr int dx_ks(int C) { return (C + 31) / 32; }
// bytes of one 16-row output block: A fragments [ks][hi, lo][1 KB], scale[16], shift[16]; whole 4 KB (256 x 16 B) passes
constexpr int dx_block_bytes(int C) { return (dx_ks(C) * 2 * 1024 + 128 + 4095) / 4096 * 4096; }

__global__ void densex_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                   unsigned char* __restrict__ packed, int C) {
    const int KS = dx_ks(C), MT = (C + 15) / 16, BLKB = dx_block_bytes(C), units = BLKB / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= MT * units) return;
    const int mt = idx / units, u = idx - mt * units;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (u < KS * 2 * 64) {
        const int frag = u >> 6, lane = u & 63, row = 16 * mt + (lane & 15), kq = lane >> 4, ks = frag >> 1;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 32 * ks + 8 * kq + j;
            v[j] = (row < C && c < C) ? w[(size_t)row * C + c] : 0.f;
        }
        h16x8 hi, lo;
        dx_split8(v, hi, lo);
        o = __builtin_bit_cast(u32x4, (frag & 1) ? lo : hi);
    } else if (u < KS * 2 * 64 + 8) {
        const int k = (u - KS * 2 * 64) * 4;                       // floats 0..15: scale, 16..31: shift
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = k + i, c = 16 * mt + (e & 15);
            v[i] = c < C ? (e < 16 ? (scale ? scale[c] : 1.f) : (shift ? shift[c] : 0.f)) : 0.f;
        }
        o = (u32x4){__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]), __builtin_bit_cast(uint32_t, v[2]),
                    __builtin_bit_cast(uint32_t, v[3])};
    }
    reinterpret_cast<u32x4*>(packed)[idx] = o;
}

// copy one weight block global -> LDS with the LDS-DMA (256 threads): unit u (16 bytes) lands at lds + 16 u
template <int BLKB>
__device__ __forceinline__ void dx_stage(const unsigned char* __restrict__ src, unsigned char* lds) {
    constexpr int NST = BLKB / 16 / 256;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int u0 = i * 256 + wave * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(u0 + lane) * 16),
                                         (__attribute__((address_space(3))) void*)(lds + u0 * 16), 16, 0, 0);
    }
}

// all MT output blocks of one problem for the wave's 32 tokens: X (split fragments) x streamed weight blocks -> out.
// Every wave issues the SAME vector-memory instructions per block - residual loads, the DMA of the next block, four result stores,
// lanes without a token or channel masked by an out-of-range buffer offset - so the barrier at the end of a block can wait for
// the DMA alone (`s_waitcnt vmcnt(4)`: all but the four stores, which stay in flight under the next block's MFMAs) instead of
// the `vmcnt(0)` of __syncthreads(), which made every block pay a store round trip.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// H1: the fp16 engine's arithmetic - the hi pieces only (operands rounded to half once), one MFMA per product
template <int C, bool RES, bool H1 = false>
__device__ __forceinline__ void dx_project(const h16x8 (&Xh)[dx_ks(C)][2], const h16x8 (&Xl)[dx_ks(C)][2],
                                           const unsigned char* __restrict__ packed, unsigned char* lds,
                                           const float* __restrict__ res, float* __restrict__ out, size_t base, int T, int tok,
                                           bool valid, unsigned* rflag) {
    constexpr int KS = dx_ks(C), MT = (C + 15) / 16, BLKB = dx_block_bytes(C);
    bool bad = false;                                 // range guard (common.h): a NaN sum = an operand beyond a half's range
    const int lane = threadIdx.x & 63, kq = lane >> 4;
    const unsigned plane = (unsigned)((size_t)C * T * sizeof(float));
    const otp_rsrc ro = make_rsrc32(out + base, plane);
    const otp_rsrc rr = make_rsrc32(RES && res ? res + base : out, RES && res ? plane : 0u);   // no residual: every load reads 0
#pragma unroll 1
    for (int mt = 0; mt < MT; ++mt) {
        int voff[4];
        f32x2 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 16 * mt + 4 * kq + i;
            voff[i] = (valid && c < C) ? (c * T + tok) * 4 : -16;                // masked lanes: past the descriptor
            if (RES) r[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rr, voff[i], 0, 0));
        }
        asm volatile("" ::: "memory");
        if (mt + 1 < MT) dx_stage<BLKB>(packed + (size_t)(mt + 1) * BLKB, lds + ((mt + 1) & 1) * BLKB);
        asm volatile("" ::: "memory");
        const unsigned char* P = lds + (mt & 1) * BLKB;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const h16x8 ah = *reinterpret_cast<const h16x8*>(P + (ks * 2) * 1024 + lane * 16);
            if constexpr (!H1) {
                const h16x8 al = *reinterpret_cast<const h16x8*>(P + (ks * 2 + 1) * 1024 + lane * 16);
                acc0 = OTP_X3_MFMA(al, Xh[ks][0], acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(al, Xh[ks][1], acc1, 0, 0, 0);
                acc0 = OTP_X3_MFMA(ah, Xl[ks][0], acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(ah, Xl[ks][1], acc1, 0, 0, 0);
            }
            acc0 = OTP_X3_MFMA(ah, Xh[ks][0], acc0, 0, 0, 0);
            acc1 = OTP_X3_MFMA(ah, Xh[ks][1], acc1, 0, 0, 0);
        }
        const f32x4 sc = *reinterpret_cast<const f32x4*>(P + KS * 2048 + 16 * kq);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(P + KS * 2048 + 64 + 16 * kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 v = {acc0[i] * sc[i] + sh[i], acc1[i] * sc[i] + sh[i]};
            if (RES) v += r[i];
            bad |= otp_out_of_range(v.x);
                bad |= otp_out_of_range(v.y);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ro, voff[i], 0, 0);
        }
        if (mt + 1 < MT) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // this wave's part of block mt + 1 has landed; the stores fly on
            __builtin_amdgcn_s_barrier();                         // block mt consumed by every wave, block mt + 1 landed for all
            asm volatile("" ::: "memory");
        }
    }
    otp_range_report(rflag, bad, OTP_RANGE_DENSEX);
}

struct DxArgs {
    const float* x[3];
    const unsigned char* packed[3];
    const float* res[3];
    float* out[3];
};

template <int C, bool H1 = false>
__global__ __launch_bounds__(256, 2) void densex_cc_kernel(DxArgs A, int T, int tiles_per_b, unsigned* rflag) {
    constexpr int KS = dx_ks(C), BLKB = dx_block_bytes(C);
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BLKB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    const int tok = tile * 128 + wave * 32 + 2 * n;
    const bool valid = tok < T;
    const size_t base = (size_t)b * C * T;
    const float* __restrict__ x = A.x[blockIdx.y];
    dx_stage<BLKB>(A.packed[blockIdx.y], lds);
    h16x8 Xh[KS][2], Xl[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float v0[8], v1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 32 * ks + 8 * kq + j;
            const bool live = 32 * ks + 24 + j < C || c < C;
            const f32x2 v = *reinterpret_cast<const f32x2*>(x + base + (size_t)(live ? c : 0) * T + (valid ? tok : T - 2));
            v0[j] = live ? v.x : 0.f;
            v1[j] = live ? v.y : 0.f;
        }
        dx_split8(v0, Xh[ks][0], Xl[ks][0]);
        dx_split8(v1, Xh[ks][1], Xl[ks][1]);
    }
    __syncthreads();
    dx_project<C, true, H1>(Xh, Xl, A.packed[blockIdx.y], lds, A.res[blockIdx.y], A.out[blockIdx.y], base, T, tok, valid, rflag);
}

struct QxArgs {
    const unsigned char* packed[3];
    float* out[3];
};

// q, k, v = W_p . LN_p(dwconv3_p(x)) + b_p in one launch (stride 1); table[p][c] = {dw0, dw1, dw2, gamma, beta, 0, 0, 0}
// (three waves per SIMD: 168 VGPRs with 21 spilled measured 119 us at cfg2 against 127 at two waves / 183 VGPRs and 133 at four)
template <int C, bool H1 = false>
__global__ __launch_bounds__(256, C <= 136 ? 3 : 1) void qkvx_front_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                            QxArgs A, int T, int tiles_per_b, float eps, unsigned* rflag) {
    constexpr int KS = dx_ks(C), BLKB = dx_block_bytes(C), TAB = 3 * C * 8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BLKB];
    __shared__ __attribute__((aligned(16))) float tab[TAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n = lane & 15;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
    const int tok = tile * 128 + wave * 32 + 2 * n;
    const bool valid = tok < T;
    const int tokc = valid ? tok : T - 2;
    const bool l_ok = tokc >= 1, r_ok = tokc + 2 < T;
    const size_t base = (size_t)b * C * T;
    const otp_rsrc rx = make_rsrc(x + base, (size_t)C * T * sizeof(float));
    constexpr float inv_c = 1.f / (float)C;
    const int vb0 = (8 * kq * T + tokc) * 4;
    for (int i = tid; i < TAB / 4; i += 256)
        reinterpret_cast<f32x4*>(tab)[i] = reinterpret_cast<const f32x4*>(table)[i];
    __syncthreads();

    // one problem (q, k or v) per workgroup: blockIdx.y (three times the workgroups, a third of the serial work each)
    for (int p = (int)blockIdx.y; p <= (int)blockIdx.y; ++p) {
        dx_stage<BLKB>(A.packed[p], lds);
        const float* tp = tab + p * C * 8;
        // depthwise k = 3 over the wave's token pair and its two neighbours (re-read per problem: L1 / L2 hits), per channel
        float X0[KS][8], X1[KS][8];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 32 * ks + 8 * kq + j;
                const bool live = 32 * ks + 24 + j < C || c < C;
                // lane part of the address in one VGPR (masked lanes point past the descriptor: zeros), the channel's uniform
                // part in the scalar offset - 120 distinct vector addresses would otherwise be hoisted out of the problem loop
                const int so = (32 * ks + j) * T * 4;
                const f32x2 m = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, live ? vb0 : -16, so, 0));
                const float la = bload(rx, live ? vb0 - 4 : -16, so), ld = bload(rx, live ? vb0 + 8 : -16, so);
                const float a = l_ok ? la : 0.f, d = r_ok ? ld : 0.f;
                const f32x4 w = *reinterpret_cast<const f32x4*>(tp + (live ? c : 0) * 8);
                X0[ks][j] = live ? w[0] * a + w[1] * m.x + w[2] * m.y : 0.f;
                X1[ks][j] = live ? w[0] * m.x + w[1] * m.y + w[2] * d : 0.f;
                s0 += X0[ks][j];
                s1 += X1[ks][j];
            }
        }
        const float m0 = dx_kslot_sum(s0) * inv_c, m1 = dx_kslot_sum(s1) * inv_c;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool live = 32 * ks + 24 + j < C || 32 * ks + 8 * kq + j < C;
                X0[ks][j] = live ? X0[ks][j] - m0 : 0.f;
                X1[ks][j] = live ? X1[ks][j] - m1 : 0.f;
                v0 += X0[ks][j] * X0[ks][j];
                v1 += X1[ks][j] * X1[ks][j];
            }
        const float r0 = 1.f / sqrtf(dx_kslot_sum(v0) * inv_c + eps), r1 = 1.f / sqrtf(dx_kslot_sum(v1) * inv_c + eps);
        h16x8 Xh[KS][2], Xl[KS][2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 32 * ks + 8 * kq + j;
                const bool live = 32 * ks + 24 + j < C || c < C;
                const float g = tp[(live ? c : 0) * 8 + 3], be = tp[(live ? c : 0) * 8 + 4];
                X0[ks][j] = live ? X0[ks][j] * r0 * g + be : 0.f;
                X1[ks][j] = live ? X1[ks][j] * r1 * g + be : 0.f;
            }
            dx_split8(X0[ks], Xh[ks][0], Xl[ks][0]);
            dx_split8(X1[ks], Xh[ks][1], Xl[ks][1]);
        }
        __syncthreads();                              // weight block 0 of this problem landed
        dx_project<C, false, H1>(Xh, Xl, A.packed[p], lds, nullptr, A.out[p], base, T, tok, valid, rflag);
    }
}

}  // namespace

// C = 136: 8 stacked maps x 17 joints (5-frame window); C = 204: 12 x 17 (the 7-frame window of BASELINE configs[4])
#ifndef OTP_X3_GRAD_COPY
extern "C" int otp_dense_x3_supported(int C, int T) { return ((C == 136 || C == 204) && T > 0 && T % 2 == 0) ? 1 : 0; }

extern "C" size_t otp_dense_x3_weight_bytes(int C) {
    if (C <= 0 || C % 4) return 0;
    return (size_t)((C + 15) / 16) * dx_block_bytes(C);
}
#endif

extern "C" int OTP_ENTRY(otp_dense_x3_pack)(const void* w, const void* scale, const void* shift, void* packed, int C, void* stream) {
    if (!w || !packed) return OTP_ERR_BAD_ARG;
    const size_t bytes = otp_dense_x3_weight_bytes(C);
    if (!bytes) return OTP_ERR_UNSUPPORTED;
    const int total = (int)(bytes / 16);
    hipLaunchKernelGGL(densex_pack_kernel, dim3(otp_ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w), static_cast<const float*>(scale), static_cast<const float*>(shift),
                       static_cast<unsigned char*>(packed), C);
    return otp_launch_status();
}

namespace {
int dx_dense_launch(const void* const* x, const void* const* packed, const void* const* res, void* const* out, int nprob, int B, int C,
                    int T, void* stream, bool h1) {
    if (!x || !packed || !out || nprob < 1 || nprob > 3 || B <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_dense_x3_supported(C, T)) return OTP_ERR_UNSUPPORTED;
    DxArgs a = {};
    for (int i = 0; i < nprob; ++i) {
        a.x[i] = static_cast<const float*>(x[i]);
        a.packed[i] = static_cast<const unsigned char*>(packed[i]);
        a.res[i] = res ? static_cast<const float*>(res[i]) : nullptr;
        a.out[i] = static_cast<float*>(out[i]);
        if (!a.x[i] || !a.packed[i] || !a.out[i]) return OTP_ERR_BAD_ARG;
        if ((reinterpret_cast<uintptr_t>(a.x[i]) | reinterpret_cast<uintptr_t>(a.res[i]) | reinterpret_cast<uintptr_t>(a.out[i])) & 7 ||
            reinterpret_cast<uintptr_t>(a.packed[i]) & 15)
            return OTP_ERR_BAD_ARG;
    }
    const int tiles = otp_ceil_div(T, 128);
    const dim3 grid((unsigned)(B * tiles), (unsigned)nprob);
    auto st = static_cast<hipStream_t>(stream);
    if (C == 204) {
        if (h1) hipLaunchKernelGGL((densex_cc_kernel<204, true>), grid, dim3(256), 0, st, a, T, tiles, otp_range_word());
        else hipLaunchKernelGGL((densex_cc_kernel<204, false>), grid, dim3(256), 0, st, a, T, tiles, otp_range_word());
    } else {
        if (h1) hipLaunchKernelGGL((densex_cc_kernel<136, true>), grid, dim3(256), 0, st, a, T, tiles, otp_range_word());
        else hipLaunchKernelGGL((densex_cc_kernel<136, false>), grid, dim3(256), 0, st, a, T, tiles, otp_range_word());
    }
    return otp_launch_status();
}
}  // namespace

extern "C" int OTP_ENTRY(otp_dense_x3)(const void* const* x, const void* const* packed, const void* const* res, void* const* out,
                            int nprob, int B, int C, int T, void* stream) {
    return dx_dense_launch(x, packed, res, out, nprob, B, C, T, stream, false);
}

#ifndef OTP_X3_GRAD_COPY
/* otp_dense_x3 with the fp16 engine's arithmetic: operands rounded to half once (the hi pieces of the same packed image), one MFMA
 * per product; fp32 tensors and accumulation */
extern "C" int otp_dense_h1(const void* const* x, const void* const* packed, const void* const* res, void* const* out, int nprob, int B,
                            int C, int T, void* stream) {
    return dx_dense_launch(x, packed, res, out, nprob, B, C, T, stream, true);
}
#endif

#ifndef OTP_X3_GRAD_COPY
namespace {
int dx_qkv_launch(const void* x, const void* table, const void* packed_q, const void* packed_k, const void* packed_v, void* q, void* k,
                  void* v, int B, int C, int T, float eps, void* stream, bool h1) {
    if (!x || !table || !packed_q || !packed_k || !packed_v || !q || !k || !v || B <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_dense_x3_supported(C, T)) return OTP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
         reinterpret_cast<uintptr_t>(v)) & 7 ||
        (reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(packed_q) | reinterpret_cast<uintptr_t>(packed_k) |
         reinterpret_cast<uintptr_t>(packed_v)) & 15)
        return OTP_ERR_BAD_ARG;
    if ((size_t)C * T * sizeof(float) >= (1ull << 31)) return OTP_ERR_UNSUPPORTED;
    QxArgs a;
    a.packed[0] = static_cast<const unsigned char*>(packed_q); a.packed[1] = static_cast<const unsigned char*>(packed_k);
    a.packed[2] = static_cast<const unsigned char*>(packed_v);
    a.out[0] = static_cast<float*>(q); a.out[1] = static_cast<float*>(k); a.out[2] = static_cast<float*>(v);
    const int tiles = otp_ceil_div(T, 128);
    const dim3 grid((unsigned)(B * tiles), 3u);
    auto st = static_cast<hipStream_t>(stream);
    auto xf = static_cast<const float*>(x), tf = static_cast<const float*>(table);
    if (C == 204) {
        if (h1) hipLaunchKernelGGL((qkvx_front_kernel<204, true>), grid, dim3(256), 0, st, xf, tf, a, T, tiles, eps, otp_range_word());
        else hipLaunchKernelGGL((qkvx_front_kernel<204, false>), grid, dim3(256), 0, st, xf, tf, a, T, tiles, eps, otp_range_word());
    } else {
        if (h1) hipLaunchKernelGGL((qkvx_front_kernel<136, true>), grid, dim3(256), 0, st, xf, tf, a, T, tiles, eps, otp_range_word());
        else hipLaunchKernelGGL((qkvx_front_kernel<136, false>), grid, dim3(256), 0, st, xf, tf, a, T, tiles, eps, otp_range_word());
    }
    return otp_launch_status();
}
}  // namespace

extern "C" int otp_qkv_front_x3(const void* x, const void* table, const void* packed_q, const void* packed_k,
                                const void* packed_v, void* q, void* k, void* v, int B, int C, int T, float eps, void* stream) {
    return dx_qkv_launch(x, table, packed_q, packed_k, packed_v, q, k, v, B, C, T, eps, stream, false);
}

/* otp_qkv_front_x3 with the fp16 engine's arithmetic (operands rounded to half once, one MFMA per product) */
extern "C" int otp_qkv_front_h1(const void* x, const void* table, const void* packed_q, const void* packed_k,
                                const void* packed_v, void* q, void* k, void* v, int B, int C, int T, float eps, void* stream) {
    return dx_qkv_launch(x, table, packed_q, packed_k, packed_v, q, k, v, B, C, T, eps, stream, true);
}
#endif  // OTP_X3_GRAD_COPY
