// HRNet's first convolution (reference model/HRNet.py:33-36 applied at :118-120): Conv2d(3, Cout, 3x3, stride 2, pad 1, no bias) +
// BatchNorm2d + ReLU on the frames of the clip tensor (model/OTPose.py:317: the (B, 5 * 3, H, W) clip read as 5 B frames of
// three channels, frame n = f B + b).  27 products per output value and 0.57 GB of output at cfg2: bound by its output stream.
// One k-step of the bf16 matrix cores holds the whole contraction (k = 3 tap + channel, 27 of 32 slots), so the layer is three
// split-bf16 MFMAs per 16 pixels x 16 channels: the A operand is gathered straight from the image (stride-2 taps, padding = loads
// past the buffer descriptor), the weights (BatchNorm scale folded in) sit in registers for the whole kernel, and a lane's
// accumulator registers are 4 consecutive pixels of one channel: one 16-byte store each.
#include "common.h"

namespace {

typedef otp_x3x8 h16x8;              // 8 operand pieces of the split products (common.h: IEEE half since round 4)
typedef otp_x3x2 h16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
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

constexpr int ST_CT = 4;          // 16-channel output tiles (Cout = 64)
// ST_LDS 1: a wave's 64 pixels x 64 channels pass through its LDS slab so that every store instruction covers 256 contiguous bytes
// per channel (64-byte runs straight from the accumulators: 311 us at cfg2, with the slab 202 us)
#ifndef ST_LDS
#define ST_LDS 1
#endif
constexpr int ST_ROW = 68;        // floats per channel row of a wave's LDS slab

// packed weights: [cout tile][hi | lo][64 lanes] 16-byte B fragments (lane (channel i16, kq): k slots 8 kq .. 8 kq + 7), then
// shift[64] * 2^k, then {2^-k, 0, 0, 0}.  The weights are stored times the power of two 2^k that puts the largest |w * scale| in
// [2^13, 2^14), so both half pieces of every weight are normal numbers (otp_conv_desc.out_scale in include/otpose_hip.h is the
// same device for the descriptor-driven kernels); every workgroup of the pack finds the maximum itself (27 Cout values).
__global__ void stem_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                 u32x4* __restrict__ packed, int Cout) {
    __shared__ float wmax[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < Cout * 27; i += blockDim.x) m = fmaxf(m, fabsf(w[i] * (scale ? scale[i / 27] : 1.f)));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    int e = 0;
    (void)frexpf(m, &e);                                                        // m = f 2^e, f in [0.5, 1)
    const int kx = (m > 0.f && m < 3e38f) ? min(40, max(-40, 14 - e)) : 0;
    const float pre = ldexpf(1.f, kx), post = ldexpf(1.f, -kx);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == ST_CT * 2 * 64 + 16) packed[idx] = (u32x4){__builtin_bit_cast(uint32_t, post), 0u, 0u, 0u};
    if (idx < ST_CT * 2 * 64) {
        const int t = idx >> 7, part = (idx >> 6) & 1, lane = idx & 63, co = 16 * t + (lane & 15), kq = lane >> 4;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kq + j, tap = k / 3, c = k - tap * 3;               // reference weight layout (Cout, 3, 3, 3): [co][c][dy][dx]
            v[j] = (k < 27 && co < Cout) ? w[(co * 3 + c) * 9 + tap] * (scale ? scale[co] : 1.f) * pre : 0.f;
        }
        h16x8 hi, lo;
        st_split8(v, hi, lo);
        packed[idx] = __builtin_bit_cast(u32x4, part ? lo : hi);
    } else if (idx < ST_CT * 2 * 64 + 16) {
        const int q = idx - ST_CT * 2 * 64;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (4 * q + i < Cout && shift) ? shift[4 * q + i] * pre : 0.f;
        packed[idx] = (u32x4){__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]), __builtin_bit_cast(uint32_t, v[2]),
                              __builtin_bit_cast(uint32_t, v[3])};
    }
}

struct StArgs {
    const float* in;
    const u32x4* packed;
    float* out;
    int B, F, H, W, Ho, Wo, HoWo, Cout, total4;      // total4: groups of 4 output pixels over all frames
    unsigned mWo4;                                    // magic divisor of Wo / 4
    unsigned* rflag;                                  // range-guard word (common.h)
};

__device__ __forceinline__ uint32_t st_div(uint32_t i, uint32_t magic) { return magic ? __umulhi(i, magic) : i; }
uint32_t st_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }

// a wave: 4 tiles of 16 consecutive output pixels; lane (pixel i16, kq) gathers k slots 8 kq .. + 7 of its pixel
template <int NPT>
__global__ __launch_bounds__(256) void stem_kernel(StArgs A) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
#if ST_LDS
    __shared__ __attribute__((aligned(16))) float slabs[4 * 64 * ST_ROW];
    float* slab = slabs + wave * 64 * ST_ROW;
#endif
    h16x8 Wh[ST_CT], Wl[ST_CT];
#pragma unroll
    for (int t = 0; t < ST_CT; ++t) {
        Wh[t] = __builtin_bit_cast(h16x8, A.packed[(t * 2) * 64 + lane]);
        Wl[t] = __builtin_bit_cast(h16x8, A.packed[(t * 2 + 1) * 64 + lane]);
    }
    const float shv[ST_CT] = {reinterpret_cast<const float*>(A.packed + ST_CT * 2 * 64)[i16],
                              reinterpret_cast<const float*>(A.packed + ST_CT * 2 * 64)[16 + i16],
                              reinterpret_cast<const float*>(A.packed + ST_CT * 2 * 64)[32 + i16],
                              reinterpret_cast<const float*>(A.packed + ST_CT * 2 * 64)[48 + i16]};
    const float post = reinterpret_cast<const float*>(A.packed + ST_CT * 2 * 64 + 16)[0];   // 2^-k of the packed weights
    const size_t clip = (size_t)3 * A.F * A.H * A.W;                           // floats of one clip
    const otp_rsrc rin = make_rsrc(A.in, (size_t)A.B * clip * sizeof(float));
    const long P0 = ((long)blockIdx.x * 4 + wave) * (16 * NPT);              // first output pixel of this wave (all frames, row-major)
    // frame and 4-pixel group of the wave's first pixel (one exact division per wave); its 16 NPT pixels span at most two frames
    const uint32_t H4 = (uint32_t)(A.HoWo >> 2), g0 = (uint32_t)(P0 >> 2), n0 = g0 / H4, rem0 = g0 - n0 * H4;
    bool bad = false;                                 // range guard (common.h): an image value beyond a half's range gives NaN sums
#pragma unroll 4
    for (int p = 0; p < NPT; ++p) {
        // ---- A fragment: the 8 (tap, channel) values of pixel P0 + 16 p + i16 --------------------------------------------------
        const long px = P0 + 16 * p + i16;
        const bool pv = px < 4l * A.total4;
        uint32_t r4 = rem0 + (uint32_t)((16 * p + i16) >> 2), n = n0;            // its 4-pixel group inside frame n
        if (r4 >= H4) r4 -= H4, ++n;
        const uint32_t yo = st_div(r4, A.mWo4), xo = 4 * (r4 - yo * (uint32_t)(A.Wo >> 2)) + (uint32_t)(px & 3);
        const int b = (int)(n % (uint32_t)A.B), f = (int)(n / (uint32_t)A.B);
        const int base = (b * 3 * A.F + 3 * f) * A.H * A.W;                     // channel 0 of frame f of clip b (fits 31 bits: checked on the host)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kq + j, tap = k / 3, c = k - tap * 3, dy = tap / 3, dx = tap - dy * 3;
            const int iy = 2 * (int)yo + dy - 1, ix = 2 * (int)xo + dx - 1;
            const bool ok = pv && k < 27 && iy >= 0 && iy < A.H && ix >= 0 && ix < A.W;
            v[j] = bload(rin, ok ? (base + (c * A.H + iy) * A.W + ix) * 4 : -16, 0);
        }
        h16x8 ah, al;
        st_split8(v, ah, al);
        // ---- 3 split products per channel tile; D row = pixel, column = channel: register r of lane (channel i16, kq) is pixel 4 kq + r
        const long q0 = P0 + 16 * p + 4 * kq;                                    // this lane's 4 consecutive output pixels
        const bool qv = q0 < 4l * A.total4;
        uint32_t r2 = rem0 + (uint32_t)(4 * p + kq), n2 = n0;
        if (r2 >= H4) r2 -= H4, ++n2;
        const uint32_t pi = 4 * r2;
#pragma unroll
        for (int t = 0; t < ST_CT; ++t) {
            f32x4 acc = {shv[t], shv[t], shv[t], shv[t]};
            acc = OTP_X3_MFMA(al, Wh[t], acc, 0, 0, 0);
            acc = OTP_X3_MFMA(ah, Wl[t], acc, 0, 0, 0);
            acc = OTP_X3_MFMA(ah, Wh[t], acc, 0, 0, 0);
            acc = acc * post;
#pragma unroll
            for (int r = 0; r < 4; ++r) bad |= otp_out_of_range(acc[r]);
            const int co = 16 * t + i16;
#if ST_LDS
            // through the wave's LDS slab [64 channels][64 + 4 pixels]: the stores below then cover 256 contiguous bytes per channel
            *reinterpret_cast<f32x4*>(slab + co * ST_ROW + 16 * (p & 3) + 4 * kq) =
                f32x4{fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f)};
            (void)qv; (void)n2; (void)pi;
#else
            if (qv && co < A.Cout) {
                f32x4 o = {fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f)};
                *reinterpret_cast<f32x4*>(A.out + ((size_t)n2 * A.Cout + co) * A.HoWo + pi) = o;
            }
#endif
        }
#if ST_LDS
        if ((p & 3) == 3) {                                                  // 64 pixels of this wave are in the slab: channel rows out
            const long r0 = P0 + 16 * (p - 3) + 4 * i16;                     // lane: pixels 4 i16 .. + 3 of the 64, channels kq + 4 j
            const bool rv = r0 < 4l * A.total4;
            uint32_t r3 = rem0 + (uint32_t)(4 * (p - 3) + i16), n3 = n0;
            if (r3 >= H4) r3 -= H4, ++n3;
            const uint32_t pj = 4 * r3;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = 4 * j + kq;
                const f32x4 o = *reinterpret_cast<const f32x4*>(slab + co * ST_ROW + 4 * i16);
                if (rv && co < A.Cout) *reinterpret_cast<f32x4*>(A.out + ((size_t)n3 * A.Cout + co) * A.HoWo + pj) = o;
            }
        }
#endif
    }
    otp_range_report(A.rflag, bad, OTP_RANGE_STEM);
}

}  // namespace

extern "C" int otp_stem_conv_x3_supported(int B, int F, int H, int W, int Cout) {
    if (B <= 0 || F <= 0 || H < 2 || W < 2 || Cout <= 0 || Cout > 16 * ST_CT) return 0;
    const int Wo = (W - 1) / 2 + 1;
    const int Ho = (H - 1) / 2 + 1;
    // a wave's 64 pixels span at most two frames; 31-bit pixel indices and input byte offsets; exact magic division by Wo / 4
    if (Wo % 4 || Ho * Wo < 64 || (size_t)B * 3 * F * H * W * 4 >= (1ull << 31) || (size_t)B * F * Ho * Wo >= (1ull << 31) ||
        (size_t)(Ho * Wo / 4) * (Wo / 4) >= (1ull << 32))
        return 0;
    return 1;
}

extern "C" size_t otp_stem_conv_x3_weight_bytes(int Cout) { return (Cout <= 0 || Cout > 16 * ST_CT) ? 0 : (ST_CT * 2 * 64 + 17) * 16; }

extern "C" int otp_stem_conv_x3_pack(const void* w, const void* scale, const void* shift, void* packed, int Cout, void* stream) {
    if (!w || !packed) return OTP_ERR_BAD_ARG;
    if (!otp_stem_conv_x3_weight_bytes(Cout)) return OTP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stem_pack_kernel, dim3(otp_ceil_div(ST_CT * 2 * 64 + 17, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w), static_cast<const float*>(scale), static_cast<const float*>(shift),
                       static_cast<u32x4*>(packed), Cout);
    return otp_launch_status();
}

/* out (F * B, Cout, Ho, Wo) = relu(conv3x3 s2 p1 of the frames of in (B, 3 F, H, W) * scale + shift), frame n = f B + b */
extern "C" int otp_stem_conv_x3(const void* in, const void* packed, void* out, int B, int F, int H, int W, int Cout, void* stream) {
    if (!in || !packed || !out) return OTP_ERR_BAD_ARG;
    if (!otp_stem_conv_x3_supported(B, F, H, W, Cout)) return OTP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(in) & 3) || ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) & 15))
        return OTP_ERR_BAD_ARG;
    StArgs a;
    a.in = static_cast<const float*>(in);
    a.packed = static_cast<const u32x4*>(packed);
    a.out = static_cast<float*>(out);
    a.B = B, a.F = F, a.H = H, a.W = W, a.Ho = (H - 1) / 2 + 1, a.Wo = (W - 1) / 2 + 1, a.HoWo = a.Ho * a.Wo, a.Cout = Cout;
    const long total = (long)B * F * a.HoWo;
    a.total4 = (int)(total / 4);
    a.mWo4 = st_magic((uint32_t)(a.Wo / 4));
    a.rflag = otp_range_word();
#ifndef ST_NPT
#define ST_NPT 4
#endif
    constexpr int NPT = ST_NPT;                                                // 16 NPT pixels per wave, 64 NPT per workgroup
    hipLaunchKernelGGL(stem_kernel<NPT>, dim3((unsigned)((total + 64 * NPT - 1) / (64 * NPT))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return otp_launch_status();
}
