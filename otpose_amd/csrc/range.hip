// Range guard of the 16-bit-operand kernels (common.h: otp_out_of_range / otp_range_report).
//
// The split-product ("f16x3") arithmetic of the eval path keeps fp32's significand but IEEE half's exponent range: an operand
// with |a| >= 65504 splits into hi = inf, lo = -inf and every product with it is NaN (csrc/common.h).  The reference computes
// these layers in fp32 (model/HRNet.py:500-530, model/blocks.py:248-254, 400-453) and has no such limit, so the limit must never
// be crossed silently: every kernel that forms half pieces tests its results before the activation (a ReLU would swallow the
// NaN) and reports through ONE word of pinned host memory.  The word is written only when a violation happens - the guard adds a
// compare per result and no memory traffic - and is sticky until the host reads it with reset.
//
//   otp_range_flag_read(reset)  0, or the OTP_RANGE_* code of (one of) the kernels that saw a violation since the last reset.  A
//                               definitive answer needs the launches in question to have completed (the caller synchronises).
//   otp_range_poison(out, n, s) last launch of a forward: fills out[0 .. n) with NaN when the word is set, so that a consumer that
//                               never asks still cannot read a finite-looking heat-map computed from an overflowed operand.
#include <cstring>
#include <mutex>

#include "common.h"

namespace {

unsigned* g_word = nullptr;
std::once_flag g_once;

// ONE workgroup, ONE read of the word: it lives in host memory, and every wave that asks pays a PCIe round trip (the first form - 256
// workgroups each reading it - took 130-170 us at the end of every forward; this one ~5 us).  The fill is the error path.
__global__ __launch_bounds__(1024) void range_poison_kernel(const unsigned* __restrict__ word, float* __restrict__ out, size_t n) {
    __shared__ unsigned flag;
    if (threadIdx.x == 0) flag = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (flag == 0u) return;
    const float nan = __builtin_nanf("");
    for (size_t i = threadIdx.x; i < n; i += 1024) out[i] = nan;
}

}  // namespace

unsigned* otp_range_word() {
    std::call_once(g_once, [] {
        void* p = nullptr;
        // pinned, device-mapped, coherent: the same address on the host and on every device of the process
        if (hipHostMalloc(&p, 64, hipHostMallocDefault) == hipSuccess && p) {
            std::memset(p, 0, 64);
            g_word = static_cast<unsigned*>(p);
        } else {
            (void)hipGetLastError();                               // no GPU (the CPU-side symbol tests): no guard word, no launches either
        }
    });
    return g_word;
}

extern "C" int otp_range_flag_read(int reset) {
    unsigned* w = otp_range_word();
    if (!w) return 0;
    const unsigned v = __atomic_load_n(w, __ATOMIC_RELAXED);
    if (reset && v) __atomic_store_n(w, 0u, __ATOMIC_RELAXED);
    return (int)v;
}

extern "C" int otp_range_poison(void* out, size_t n, void* stream) {
    if (!out || n == 0) return OTP_ERR_BAD_ARG;
    const unsigned* w = otp_range_word();
    if (!w) return OTP_ERR_LAUNCH;
    hipLaunchKernelGGL(range_poison_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), w, static_cast<float*>(out), n);
    return otp_launch_status();
}
