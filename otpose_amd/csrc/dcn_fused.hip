// the offset / mask convolutions fused into the deformable-convolution gather, all dilations of
// the warping head in ONE launch.  Replaces, per dilation d of model/OTPose.py:381-392,
//     offsets = Conv2d(32 -> 18 J, 3x3, dilation d, no bias)(trans)            (model/OTPose.py:168-177, 382)
//     masks   = Conv2d(32 ->  9 J, 3x3, dilation d, no bias)(trans)            (:383)
//     warped  = ModulatedDeformConv(J -> J, 3x3, dilation d, deformable_groups J)(def_heatmaps, offsets, masks)   (:384)
// and the weighted sum over the dilations (:387-392): the 459 offset / mask channels per pixel and dilation (1.0 GB per
// forward at batch 16) are never written to HBM.
//
// A workgroup (8 waves) owns 128 consecutive pixels of one image, a wave 16 of them.  Per dilation:
//   * the wave's A operand - its 16 pixels x 9 dilated taps x 32 channels of `trans`, split into bf16 hi / lo pieces - is 18
//     16-byte loads per lane from an NHWC split copy of `trans` (a 14 MB pre-pass, L2 resident) and stays in registers for
//     all J deformable groups (zero padding = loads past the descriptor);
//   * per group g the 27 (padded to 32) offset / mask channels are one [16 pixels x 288] x [288 x 32] product on the bf16
//     matrix cores with split products (csrc/convx.hip): 54 MFMAs, the weight fragments streamed through the LDS by the
//     LDS-DMA (40 KB per group, double buffered);
//   * the 16 x 32 result goes through a per-wave LDS scratch so that four lanes per pixel share the nine taps: each reads
//     its taps' (dy, dx, mask), gathers the four bilinear corners of plane g of `def_heatmaps` from global memory (7.5 MB,
//     L2 resident; corners outside the image are loads past the descriptor = 0, the sample is dropped outside the open
//     interval (-1, H) x (-1, W) exactly like deform_conv_cuda_kernel.cu:403-432, 549-556) and accumulates
//     mask * sample * W_dcn[:, g, tap] into J output registers.
// The J outputs accumulate over groups AND dilations in registers; one store per pixel at the end.
#include "common.h"

namespace {

typedef otp_x3x8 h16x8;              // 8 operand pieces of the split products (common.h: IEEE half since round 4)
typedef otp_x3x2 h16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int FCIN = 32;                      // channels of `trans`
constexpr int FBLK = 40960;                   // bytes of one (dilation, group) weight block: 9 taps x 2 n-tiles x (hi, lo) x 1 KB, padded
constexpr int FPIX = 128;                     // pixels per workgroup (8 waves); the 9-wave form owns 144
constexpr int FMAXD = 8;                      // dilations per launch
#ifndef FSCR
#define FSCR 36
#endif
constexpr int FSCR_ = FSCR;                   // floats per pixel row of a wave's sampling scratch (32 channels + padding): with 32 the
                                              // 16 pixels of a read sat on two banks (50 % of the LDS-active cycles were conflicts,
                                              // profiles/r04_dcnf_pmc_fold.txt); 36 puts the 16 pixels 4 banks apart and the four
                                              // row groups of a write 16 banks apart

struct FusedPlan {
    int B, J, H, W, HW, ND, tilesPerImg;
    int dil[FMAXD];
    float alpha;
    unsigned* rflag;                                                // range-guard word (common.h)
};

__device__ __forceinline__ void f_split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 a = {v[2 * i], v[2 * i + 1]};
        const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, h16x2));
        const f32x2 af = otp_x3_widen(hb);
        h[i] = hb;
        l[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a - af, h16x2));
    }
    hi = (u32x4){h[0], h[1], h[2], h[3]};
    lo = (u32x4){l[0], l[1], l[2], l[3]};
}

// trans (B, 32, H, W) fp32 -> (B, H, W, [32 bf16 hi | 32 bf16 lo]): thread = (pixel, 8-channel group)
__global__ __launch_bounds__(256) void dcnf_split_kernel(const float* __restrict__ trans, u32x4* __restrict__ ws, int B, int HW) {
    const size_t total = (size_t)B * HW * 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3);
        const size_t px = i >> 2;
        const size_t n = px / HW, p = px - n * HW;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = trans[(n * FCIN + 8 * q + j) * HW + p];
        u32x4 hi, lo;
        f_split8(v, hi, lo);
        ws[px * 8 + q] = hi;
        ws[px * 8 + 4 + q] = lo;
    }
}

// def_heatmaps (B, J, H, W) -> planes with a one-pixel ZERO border, (H + 2) x (W + 2): a bilinear corner of a sample inside the
// open interval (-1, H) x (-1, W) is then always a valid element (row floor(h) + 1 in [0, H], + 1 in [1, H + 1]), so the gather
// needs no per-corner bounds logic - ~13 of the ~55 vector instructions per sample of a kernel that is bound by their issue
// (profiles/r04_dcnf_pmc_fold.txt: 140 M non-MFMA vector instructions per launch) - and the four corners share one address
// register (immediate offsets 4, Wp 4, Wp 4 + 4).  8.4 MB at cfg2, L2 resident like the planes it replaces.
__global__ __launch_bounds__(256) void dcnf_pad_kernel(const float* __restrict__ x, float* __restrict__ xp, int planes, int H, int W) {
    const int Hp = H + 2, Wp = W + 2;
    const size_t total = (size_t)planes * Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Wp);
        const size_t r = i / Wp;
        const int y = (int)(r % Hp);
        const size_t pl = r / Hp;
        const bool in = y >= 1 && y <= H && c >= 1 && c <= W;
        xp[i] = in ? x[(pl * H + (y - 1)) * W + (c - 1)] : 0.f;
    }
}

// The offset and the mask weights of a dilation are stored times a power of two each (2^k with max |w| 2^k in [2^13, 2^14):
// both half pieces of every weight are then normal numbers, 22 significand bits instead of ~17 - otp_conv_desc.out_scale in
// include/otpose_hip.h is the same device) and the kernel multiplies the sums by 2^-k on their way to the sampling scratch.
// One workgroup per (dilation, offsets | masks) finds the maximum and writes post = 2^-k into the image's tail.
__global__ __launch_bounds__(256) void dcnf_exp_kernel(const float* const* __restrict__ w_off, const float* const* __restrict__ w_mask,
                                                       float* __restrict__ post, int J) {
    __shared__ float wmax[4];
    const int di = blockIdx.x, kind = blockIdx.y;
    const float* w = kind ? w_mask[di] : w_off[di];
    const int count = (kind ? 9 : 18) * J * FCIN * 9;
    float m = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) m = fmaxf(m, fabsf(w[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        int e = 0;
        (void)frexpf(m, &e);
        const int k = (m > 0.f && m < 3e38f) ? min(40, max(-40, 14 - e)) : 0;
        post[di * 2 + kind] = ldexpf(1.f, -k);
    }
}

// packed image: [ND][J] weight blocks of FBLK bytes, then [ND][J][9][20] floats (W_dcn[o][g][k], o padded to 20), then the
// J bias sums (5 units), then post[ND][offsets | masks] (dcnf_exp_kernel, 4 units).  Weight block: [tap][n-tile][hi, lo][lane][8 bf16], lane = (channel 16 nt + (lane & 15) of the group's 32:
// 0..17 offsets 18 g + c, 18..26 masks 9 g + c - 18, 27..31 zero; kq = lane >> 4 -> input channels 8 kq .. + 7)
__global__ void dcnf_pack_kernel(const float* const* __restrict__ w_off, const float* const* __restrict__ w_mask,
                                 const float* const* __restrict__ w_dcn, const float* const* __restrict__ bias,
                                 unsigned char* __restrict__ packed, int ND, int J) {
    const int units = FBLK / 16;
    const size_t nW = (size_t)ND * J * units, nT = (size_t)ND * J * 9 * 5, total = nW + nT + 5;
    const float* post = reinterpret_cast<const float*>(packed + (total << 4));          // written by dcnf_exp_kernel before this launch
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        u32x4 o = {0u, 0u, 0u, 0u};
        if (idx < nW) {
            const int blk = (int)(idx / units), u = (int)(idx - (size_t)blk * units);
            const int di = blk / J, g = blk - di * J;
            if (u < 9 * 2 * 2 * 64) {
                const int frag = u >> 6, lane = u & 63, part = frag & 1, nt = (frag >> 1) & 1, tap = frag >> 2;
                const int ch = 16 * nt + (lane & 15), kq = lane >> 4;
                const float pre = 1.f / post[di * 2 + (ch < 18 ? 0 : 1)];
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ci = 8 * kq + j;
                    v[j] = pre * (ch < 18 ? w_off[di][((size_t)(18 * g + ch) * FCIN + ci) * 9 + tap]
                                          : (ch < 27 ? w_mask[di][((size_t)(9 * g + ch - 18) * FCIN + ci) * 9 + tap] : 0.f));
                }
                u32x4 hi, lo;
                f_split8(v, hi, lo);
                o = part ? lo : hi;
            }
        } else if (idx < nW + nT) {
            const size_t t = idx - nW;
            const int q = (int)(t % 5), k = (int)((t / 5) % 9), g = (int)((t / 45) % J), di = (int)(t / (45 * (size_t)J));
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int oc = 4 * q + i;
                v[i] = oc < J ? w_dcn[di][((size_t)oc * J + g) * 9 + k] : 0.f;
            }
            o = (u32x4){__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]), __builtin_bit_cast(uint32_t, v[2]),
                        __builtin_bit_cast(uint32_t, v[3])};
        } else {
            const int q = (int)(idx - nW - nT);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int oc = 4 * q + i;
                float s = 0.f;
                for (int di = 0; di < ND; ++di) s += (oc < J && bias[di]) ? bias[di][oc] : 0.f;
                v[i] = s;
            }
            o = (u32x4){__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]), __builtin_bit_cast(uint32_t, v[2]),
                        __builtin_bit_cast(uint32_t, v[3])};
        }
        reinterpret_cast<u32x4*>(packed)[idx] = o;
    }
}

template <int NW>
__device__ __forceinline__ void f_stage(const unsigned char* __restrict__ src, unsigned char* lds) {
    constexpr int UNITS = FBLK / 16, NST = (UNITS + NW * 64 - 1) / (NW * 64);   // 64-unit runs, one wave each
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int u0 = i * NW * 64 + wave * 64;
        if (UNITS % (NW * 64) == 0 || u0 < UNITS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(u0 + lane) * 16),
                                             (__attribute__((address_space(3))) void*)(lds + u0 * 16), 16, 0, 0);
    }
}

// NW waves per workgroup = 16 NW pixels.  8 is the general form; 9 exists for the maps it divides into a whole number of
// rounds: 16 x 96x72 is 864 workgroups of 128 pixels = 3.4 rounds of the 256 CUs (one workgroup per CU: the fourth round runs
// at 38 %), but 768 of 144 pixels = exactly 3.
template <int J, int NW>
__global__ __launch_bounds__(64 * NW) void dcn_fused_kernel(const unsigned char* __restrict__ ws, const float* __restrict__ x,
                                                            const unsigned char* __restrict__ packed, float* __restrict__ out,
                                                            const FusedPlan P) {
    constexpr int JP = (J + 3) & ~3;                                // outputs padded to float4s (20 for J = 17)
    static_assert(JP <= 20, "the W_dcn table rows hold 20 floats");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wbuf = smem;                                     // 2 x FBLK
    float* scratch = reinterpret_cast<float*>(smem + 2 * FBLK);     // [8 waves][16 pixels][32 channels]
    float* wd = scratch + NW * 16 * FSCR_;                          // [J][9][20]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int n = (int)blockIdx.x / P.tilesPerImg, p0 = ((int)blockIdx.x - n * P.tilesPerImg) * (16 * NW);
    const int p = p0 + wave * 16 + i16;                             // the lane's pixel (both as fragment row and as sampling pixel)
    const int y = p / P.W, xx0 = p - y * P.W;
    float* scr = scratch + wave * (16 * FSCR_);

    const otp_rsrc rws = make_rsrc32(ws + (size_t)n * P.HW * 128, (unsigned)P.HW * 128u);
    const int Wp = P.W + 2, HWp = (P.H + 2) * Wp;              // x: the zero-bordered planes of dcnf_pad_kernel
    const otp_rsrc rx = make_rsrc32(x + (size_t)n * J * HWp, (unsigned)(J * HWp) * 4u);
    const size_t table_off = (size_t)P.ND * J * FBLK;
    const float* postv = reinterpret_cast<const float*>(packed + table_off + ((size_t)P.ND * J * 45 + 5) * 16);

    float part[JP];
#pragma unroll
    for (int o = 0; o < JP; ++o) part[o] = 0.f;
    // range guard (common.h): a `trans` value beyond a half's range makes the offset / mask sums NaN, and the open-interval test
    // below would then DROP the sample silently (every comparison with a NaN is false) - so the sums are tested
    bool bad = false;
    // the lane's taps in the sampling phase: sub 0 -> taps 0, 1, 2; sub s > 0 -> taps 2 s + 1, 2 s + 2
    const int kbase = kq == 0 ? 0 : 2 * kq + 1, kcnt = kq == 0 ? 3 : 2;

    for (int di = 0; di < P.ND; ++di) {
        const int d = P.dil[di];
        // 2^-k of this dilation's packed offset / mask weights: accumulator 0 holds offset channels 0 .. 15, accumulator 1 the
        // offset channels 16, 17 and the masks
        const float post0 = postv[2 * di], post1 = i16 < 2 ? post0 : postv[2 * di + 1];
        __syncthreads();                                            // the previous dilation's table / weight buffers are free
        for (int i = tid; i < J * 9 * 5; i += 64 * NW)
            reinterpret_cast<u32x4*>(wd)[i] = reinterpret_cast<const u32x4*>(packed + table_off)[(size_t)di * J * 45 + i];
        // pixel fragments of all nine taps (rows outside the image / columns outside the row: offset past the descriptor = 0)
        u32x4 ah[9], al[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + (k / 3 - 1) * d, xx = xx0 + (k % 3 - 1) * d;
            const bool ok = yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
            const int off = ok ? (yy * P.W + xx) * 128 + kq * 16 : -128;
            ah[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, off, 0, 0));
            al[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, off, 64, 0));
        }
        f_stage<NW>(packed + (size_t)(di * J) * FBLK, wbuf);
        __syncthreads();                                            // block 0 landed, table visible

#pragma unroll 1
        for (int g = 0; g < J; ++g) {
            if (g + 1 < J) f_stage<NW>(packed + (size_t)(di * J + g + 1) * FBLK, wbuf + ((g + 1) & 1) * FBLK);
            const unsigned char* wb = wbuf + (g & 1) * FBLK + lane * 16;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const h16x8 b0h = *reinterpret_cast<const h16x8*>(wb + (k * 4 + 0) * 1024);
                const h16x8 b0l = *reinterpret_cast<const h16x8*>(wb + (k * 4 + 1) * 1024);
                const h16x8 b1h = *reinterpret_cast<const h16x8*>(wb + (k * 4 + 2) * 1024);
                const h16x8 b1l = *reinterpret_cast<const h16x8*>(wb + (k * 4 + 3) * 1024);
                const h16x8 a_h = __builtin_bit_cast(h16x8, ah[k]), a_l = __builtin_bit_cast(h16x8, al[k]);
                acc0 = OTP_X3_MFMA(a_l, b0h, acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(a_l, b1h, acc1, 0, 0, 0);
                acc0 = OTP_X3_MFMA(a_h, b0l, acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(a_h, b1l, acc1, 0, 0, 0);
                acc0 = OTP_X3_MFMA(a_h, b0h, acc0, 0, 0, 0);
                acc1 = OTP_X3_MFMA(a_h, b1h, acc1, 0, 0, 0);
            }
            // accumulator (channel i16 / 16 + i16, pixels 4 kq + r) -> scratch[pixel][channel]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float o0 = acc0[r] * post0, o1 = acc1[r] * post1;
                bad |= otp_out_of_range(o0);
                bad |= otp_out_of_range(o1);
                scr[(4 * kq + r) * FSCR_ + i16] = o0;
                scr[(4 * kq + r) * FSCR_ + 16 + i16] = o1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // sampling: lane = (pixel i16, sub kq)
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const bool live = it < kcnt;
                const int k = live ? kbase + it : kbase;
                const float oh = scr[i16 * FSCR_ + 2 * k], ow = scr[i16 * FSCR_ + 2 * k + 1], m = scr[i16 * FSCR_ + 18 + k];
                const int ky = k / 3, kx = k - 3 * ky;
                const float h = (float)(y + (ky - 1) * d) + oh, w = (float)(xx0 + (kx - 1) * d) + ow;
                const bool inside = h > -1.f && w > -1.f && h < (float)P.H && w < (float)P.W;
                const float hc = inside ? h : 0.f, wc = inside ? w : 0.f;
                const float hf = floorf(hc), wf = floorf(wc);
                const int hl = (int)hf, wl = (int)wf;
                const float lh = hc - hf, lw = wc - wf, hh = 1.f - lh, hw = 1.f - lw;
                // (hl, wl) in [-1, H - 1] x [-1, W - 1]: element (hl + 1, wl + 1) of the bordered plane and its three neighbours exist
                const int base = (g * HWp + (hl + 1) * Wp + wl + 1) * 4;
                const float v1 = bload(rx, base, 0);
                const float v2 = bload(rx, base + 4, 0);
                const float v3 = bload(rx, base + Wp * 4, 0);
                const float v4 = bload(rx, base + Wp * 4 + 4, 0);
                const float smp = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
                const float val = (live && inside) ? smp * m : 0.f;
                const float* wrow = wd + (g * 9 + k) * 20;
#pragma unroll
                for (int q = 0; q < JP / 4; ++q) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) part[4 * q + i] = fmaf(w4[i], val, part[4 * q + i]);
                }
            }
            __syncthreads();                                        // weight buffer swap; the wave's scratch is free again
        }
    }
    otp_range_report(P.rflag, bad, OTP_RANGE_DCNF);
    // the four subs of a pixel hold partial sums over their taps
#pragma unroll
    for (int o = 0; o < JP; ++o) {
        part[o] += __shfl_xor(part[o], 16, 64);
        part[o] += __shfl_xor(part[o], 32, 64);
    }
    if (kq == 0) {
        const float* bs = reinterpret_cast<const float*>(packed + table_off + (size_t)P.ND * J * 45 * 16);
#pragma unroll
        for (int o = 0; o < J; ++o) out[((size_t)n * J + o) * P.HW + p] = P.alpha * (part[o] + bs[o]);
    }
}

}  // namespace

namespace {
// 144-pixel workgroups: the map divides, and 8-wave workgroups would leave the last round of the 256 CUs under 70 % full
bool P_nine(int B, int HW) {
    if (HW % 144) return false;
    const long t8 = (long)B * (HW / FPIX), t9 = (long)B * (HW / 144);
    const double fill8 = (double)t8 / (((t8 + 255) / 256) * 256.0), fill9 = (double)t9 / (((t9 + 255) / 256) * 256.0);
    return t9 >= 256 && fill9 * 8.0 / 9.0 > fill8;          // a 9-wave workgroup takes 9/8 of the time (LDS-bound per CU)
}
}  // namespace

extern "C" int otp_dcn_fused_supported(int Cin, int J, int H, int W, int ND) {
    return (Cin == FCIN && J == 17 && H > 0 && W > 0 && (H * W) % FPIX == 0 && ND >= 1 && ND <= FMAXD &&
            (long)H * W * 128 < (1l << 31) && (long)J * (H + 2) * (W + 2) * 4 < (1l << 31))   // (the gather reads the bordered planes)
               ? 1 : 0;
}

extern "C" size_t otp_dcn_fused_weight_bytes(int ND, int J) {
    if (ND <= 0 || J <= 0 || J > 20) return 0;
    return (size_t)ND * J * FBLK + (size_t)ND * J * 45 * 16 + 5 * 16 + (FMAXD * 2 * 4);
}

// w_off[i] (18 J, 32, 3, 3), w_mask[i] (9 J, 32, 3, 3), w_dcn[i] (J, J, 3, 3), bias[i] (J) or NULL: device pointer arrays
// (device memory) of ND entries each
extern "C" int otp_dcn_fused_pack(const void* const* w_off, const void* const* w_mask, const void* const* w_dcn,
                                  const void* const* bias, void* packed, int ND, int J, void* stream) {
    if (!w_off || !w_mask || !w_dcn || !bias || !packed) return OTP_ERR_BAD_ARG;
    const size_t bytes = otp_dcn_fused_weight_bytes(ND, J);
    if (!bytes) return OTP_ERR_UNSUPPORTED;
    if (ND > FMAXD) return OTP_ERR_UNSUPPORTED;
    float* post = reinterpret_cast<float*>(static_cast<unsigned char*>(packed) + bytes - FMAXD * 2 * 4);
    hipLaunchKernelGGL(dcnf_exp_kernel, dim3(ND, 2), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float* const*>(w_off), reinterpret_cast<const float* const*>(w_mask), post, J);
    hipLaunchKernelGGL(dcnf_pack_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float* const*>(w_off), reinterpret_cast<const float* const*>(w_mask),
                       reinterpret_cast<const float* const*>(w_dcn), reinterpret_cast<const float* const*>(bias),
                       static_cast<unsigned char*>(packed), ND, J);
    return otp_launch_status();
}

// the split copy of `trans` ((B, H, W, [32 hi | 32 lo]) halves) + the zero-bordered copy of the J = 17 heat-map planes
static size_t dcnf_split_bytes(int B, int H, int W) { return (size_t)B * H * W * 128; }
extern "C" size_t otp_dcn_fused_workspace(int B, int H, int W) {
    constexpr int J = 17;                                           // the one instantiation (otp_dcn_fused_supported)
    return (B > 0 && H > 0 && W > 0) ? dcnf_split_bytes(B, H, W) + (size_t)B * J * (H + 2) * (W + 2) * sizeof(float) : 0;
}

extern "C" int otp_dcn_fused_forward(const void* trans, const void* x, const void* packed, void* out, void* workspace,
                                     size_t workspace_bytes, int B, int Cin, int J, int H, int W, const int* dilations, int ND,
                                     float alpha, void* stream) {
    if (!trans || !x || !packed || !out || !workspace || !dilations || B <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_dcn_fused_supported(Cin, J, H, W, ND)) return OTP_ERR_UNSUPPORTED;
    if (workspace_bytes < otp_dcn_fused_workspace(B, H, W)) return OTP_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(workspace)) & 15) return OTP_ERR_BAD_ARG;
    auto st = static_cast<hipStream_t>(stream);
    FusedPlan P{};
    // nine-wave workgroups when they tile the map and make the launch a whole number of rounds of the chip
    const bool nine = P_nine(B, H * W);
    const int px = nine ? 144 : FPIX;
    P.B = B; P.J = J; P.H = H; P.W = W; P.HW = H * W; P.ND = ND; P.tilesPerImg = P.HW / px; P.alpha = alpha;
    P.rflag = otp_range_word();
    for (int i = 0; i < ND; ++i) {
        if (dilations[i] <= 0) return OTP_ERR_BAD_ARG;
        P.dil[i] = dilations[i];
    }
    const size_t nsplit = (size_t)B * P.HW * 4;
    hipLaunchKernelGGL(dcnf_split_kernel, dim3((unsigned)((nsplit + 255) / 256 > 4096 ? 4096 : (nsplit + 255) / 256)), dim3(256), 0,
                       st, static_cast<const float*>(trans), static_cast<u32x4*>(workspace), B, P.HW);
    float* xp = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + dcnf_split_bytes(B, H, W));
    {
        const size_t npad = (size_t)B * J * (H + 2) * (W + 2);
        hipLaunchKernelGGL(dcnf_pad_kernel, dim3((unsigned)((npad + 255) / 256 > 4096 ? 4096 : (npad + 255) / 256)), dim3(256), 0, st,
                           static_cast<const float*>(x), xp, B * J, H, W);
    }
    const size_t lds = 2 * (size_t)FBLK + (size_t)(px / 16) * 16 * FSCR_ * 4 + (size_t)17 * 9 * 20 * 4 + 64;
    if (nine) {
        auto kern = dcn_fused_kernel<17, 9>;
        OTP_ALLOW_BIG_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * P.tilesPerImg)), dim3(576), lds, st, static_cast<const unsigned char*>(workspace),
                           static_cast<const float*>(xp), static_cast<const unsigned char*>(packed), static_cast<float*>(out), P);
    } else {
        auto kern = dcn_fused_kernel<17, 8>;
        OTP_ALLOW_BIG_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * P.tilesPerImg)), dim3(512), lds, st, static_cast<const unsigned char*>(workspace),
                           static_cast<const float*>(xp), static_cast<const unsigned char*>(packed), static_cast<float*>(out), P);
    }
    return otp_launch_status();
}
