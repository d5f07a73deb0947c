// Shared helpers for the gfx950 kernels of libotpose_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "../../include/otpose_hip.h"

#define OTP_LDS_LIMIT (160 * 1024)

static inline int otp_launch_status() {
    return hipGetLastError() == hipSuccess ? OTP_OK : OTP_ERR_LAUNCH;
}

// Raise the dynamic-LDS limit of a kernel once per DEVICE and per kernel instantiation (the flags live in the
// enclosing template instantiation): hipFuncSetAttribute applies to the device that is current at the call, so a
// process driving several GPUs must set it on each.  Set-once, so launches stay graph-capturable; the flag is an
// atomic bit per device (two threads racing both set the attribute, which is idempotent).
#define OTP_MAX_DEVICES 64
#define OTP_ALLOW_BIG_LDS(kern, bytes)                                                                      \
    do {                                                                                                    \
        if ((bytes) > 64 * 1024) {                                                                          \
            static std::atomic<uint64_t> otp_done_{0};                                                      \
            int otp_dev_ = 0;                                                                               \
            (void)hipGetDevice(&otp_dev_);                                                                  \
            const uint64_t otp_bit_ = 1ull << (otp_dev_ & (OTP_MAX_DEVICES - 1));                           \
            if (!(otp_done_.load(std::memory_order_acquire) & otp_bit_)) {                                  \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, OTP_LDS_LIMIT);       \
                otp_done_.fetch_or(otp_bit_, std::memory_order_release);                                    \
            }                                                                                               \
        }                                                                                                   \
    } while (0)

static inline int otp_ceil_div(int a, int b) { return (a + b - 1) / b; }

// wave64 reductions (DPP-lowered __shfl_xor)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- buffer (SRSRC) addressing: one VGPR byte offset + one SGPR byte offset per access -------------
// gfx9-family dword3 (DST_SEL/format) constant for raw buffers
#define OTP_BUFFER_DWORD3 0x00020000
typedef __amdgpu_buffer_rsrc_t otp_rsrc;
__device__ __forceinline__ otp_rsrc make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0,
                                             (int)(bytes > 0xffffffffull ? 0xffffffffull : bytes), OTP_BUFFER_DWORD3);
}
// same with a 32-bit size known to fit (no 64-bit clamp arithmetic)
__device__ __forceinline__ otp_rsrc make_rsrc32(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, OTP_BUFFER_DWORD3);
}
__device__ __forceinline__ float bload(otp_rsrc r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));
}
typedef float otp_f32x4 __attribute__((ext_vector_type(4)));
typedef float otp_f32x2 __attribute__((ext_vector_type(2)));

// ---- the 16-bit operand type of the split ("x3") products ------------------------------------------------------------------
// Every fp32 operand a of the convolutions / projections / MLPs / attention products of the eval path is carried as two 16-bit
// pieces, hi = rne(a), lo = rne(a - hi), and a product is accumulated in fp32 as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on the
// 16x16x32 MFMA (csrc/convx.hip).  Rounds 2-3 used bfloat16 pieces: 8 significand bits each, a = hi + lo to 2^-17.  Round 4
// switched to IEEE half: 11 bits each, a = hi + lo to max(2^-22 |a|, 2^-25) - the f16 MFMA of gfx950 has the same rate as the
// bf16 one and takes SUBNORMAL f16 inputs exactly (tools/micro/mfma_f16_denorm.hip), which the small `lo` pieces need.  Range:
// operands above 65504 become infinities (BatchNorm-folded weights and heat-map activations are orders of magnitude below);
// -DOTP_X3_BF16 restores the bfloat16 pieces.
#ifdef OTP_X3_BF16
typedef __bf16 otp_x3_t;
#define OTP_X3_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#else
typedef _Float16 otp_x3_t;
#define OTP_X3_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#endif
typedef otp_x3_t otp_x3x2 __attribute__((ext_vector_type(2)));
typedef otp_x3_t otp_x3x8 __attribute__((ext_vector_type(8)));
// the two pieces of a packed pair back as fp32
__device__ __forceinline__ otp_f32x2 otp_x3_widen(uint32_t pair) {
    return __builtin_convertvector(__builtin_bit_cast(otp_x3x2, pair), otp_f32x2);
}
typedef unsigned int otp_u32x4 __attribute__((ext_vector_type(4)));

// ---- range guard of the 16-bit-operand kernels (csrc/range.hip) ----------------------------------------------------------------
// A half piece overflows at 65504: hi = rne(a) = inf, lo = rne(a - hi) = -inf, every product with it NaN - and a ReLU (v_max_f32
// drops a quiet NaN) or the DCN's open-interval test can swallow that NaN silently.  Every kernel that forms such pieces
// therefore tests ITS RESULTS before the activation - a NaN accumulator is an operand that overflowed, |v| >= 65504 is a value
// the next consumer could not split - and stores a code word through `otp_range_word()` (one word of pinned host memory, visible
// to every device of the process; written only on a violation, so the guard costs a compare per result and no memory traffic).
// include/otpose_hip.h: otp_range_flag_read / otp_range_poison.
#ifdef OTP_X3_BF16
#define OTP_RANGE_LIMIT 3.0e38f                                    /* bfloat16 pieces (the gradient builds): fp32's own range */
#else
#define OTP_RANGE_LIMIT 65504.f
#endif
enum {
    OTP_RANGE_CONVX = 1, OTP_RANGE_CONVS = 2, OTP_RANGE_CONVS2 = 3, OTP_RANGE_POINTX = 4, OTP_RANGE_STEM = 5, OTP_RANGE_S8PASS = 6,
    OTP_RANGE_MLPX = 7, OTP_RANGE_DENSEX = 8, OTP_RANGE_ATTN = 9, OTP_RANGE_DCNF = 10, OTP_RANGE_H16 = 11
};
unsigned* otp_range_word();                                        // host side: the word's address (NULL without a GPU)
#ifdef OTP_NO_RANGE_GUARD                                          /* development A/B only (tools/lib_variant.sh): what the guard costs */
__device__ __forceinline__ bool otp_out_of_range(float) { return false; }
#else
__device__ __forceinline__ bool otp_out_of_range(float v) { return !(__builtin_fabsf(v) < OTP_RANGE_LIMIT); }   // true for NaN too
#endif
__device__ __forceinline__ void otp_range_report(unsigned* word, bool bad, unsigned code) {
    if (bad && word) __hip_atomic_store(word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// max(x, 0) as ONE instruction.  fmaxf lowers to v_max_f32 x, x (the IEEE quieting of a signalling NaN) + v_max_f32 x, 0: two vector
// instructions per value in epilogues that are bound by vector issue (48 values per lane and tile in the conv kernels).
__device__ __forceinline__ float otp_relu(float x) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}
// 16-byte buffer load: offsets at or past the descriptor's size return zeros (hardware range check)
__device__ __forceinline__ otp_f32x4 bload4(otp_rsrc r, int voff_bytes) {
    return __builtin_bit_cast(otp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, 0, 0));
}
__device__ __forceinline__ void bstore(float v, otp_rsrc r, int voff_bytes, int soff_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff_bytes, soff_bytes, 0);
}
