// Dense convolution as an implicit GEMM on the gfx950 f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Covers every dense contraction of the OTPose forward: HRNet 3x3/1x1/stride-2 convs with folded
// BatchNorm (+ residual, + ReLU, + nearest-upsample-accumulate of the fuse layers), the RSB heads
// (bias + BN, pre-added inputs, channel-sliced split/concat), the dilated offset/mask convs, the
// final 1x1 layers and every Conv1d(k=1) of the ConvTransformers (q/k/v/proj with the residual-scale
// epilogue, MLP with exact GELU).  Reference call sites: model/HRNet.py:116-152,478-496,514-571;
// model/RSB.py:77-103; model/OTPose.py:372-383; model/blocks.py:248-254,418-420,450.
//
// GEMM view:  D[co, pixel] = sum_{tap, ci} Wp[tap][ci][co] * X[ci][pixel shifted by tap]
//   M = Cout (16-row MFMA blocks), N = output pixels (16-column blocks, lane = pixel so that stores
//   of one accumulator register are 64-byte runs along W), K = taps x input channels in steps of 4.
// A workgroup owns a CONTIGUOUS range of the flattened output pixels of one image (so any image
// width tiles without waste) times a slab of output channels; per chunk of CK input channels it
// stages the input rows it needs (zero padded, optionally the sum of two tensors) and the matching
// weight slab into LDS, then every wave runs MB x PB MFMAs per K-step from LDS fragments.
// f32-in/f32-accumulate MFMA is bit-identical to an fmaf chain, so results differ from a CPU conv
// only by summation order.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvPlan {
    otp_conv_desc d;
    int Cout16, KK;
    int WM, WP, Mtile, Ptile, tiles_per_img, HoWo;
    int CK;            // input channels per LDS chunk (multiple of 4)
    int flat;          // 1x1 / stride 1 / no padding: patch is the pixel range itself
    int NRmax, LW;     // staged rows per channel and row pitch (floats)
    int CS, MS;        // LDS channel stride / weight-row stride (floats), chosen against bank conflicts
    int nthreads;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

template <int MB, int PB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const float* __restrict__ in, const float* __restrict__ in2, const float* __restrict__ wp,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* res,
    float* out, ConvPlan P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const otp_conv_desc& d = P.d;
    float* inp = smem;                          // [CK][CS]
    float* wts = smem + P.CK * P.CS;            // [KK*CK][MS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = P.nthreads >> 6;
    const int n = blockIdx.x / P.tiles_per_img;
    const int q0 = (blockIdx.x - n * P.tiles_per_img) * P.Ptile;
    const int wm = wave / P.WP, wpi = wave - wm * P.WP;
    const int m_wg = blockIdx.y * P.Mtile;
    const int m_wave = wm * 16 * MB;            // within the workgroup slab
    const int pix_wave = q0 + wpi * 16 * PB;
    const int HW = d.H * d.W;

    // image n of the (possibly frame-split) input
    size_t in_base, in2_base = 0;
    if (d.frame_split > 0) {
        int b = n % d.frame_split, f = n / d.frame_split;
        in_base = ((size_t)b * d.in_ctot + d.in_coff + (size_t)f * d.Cin) * HW;
    } else {
        in_base = ((size_t)n * d.in_ctot + d.in_coff) * HW;
    }
    if (in2) in2_base = ((size_t)n * d.in2_ctot + d.in2_coff) * HW;

    // rows of the input this pixel range needs (2-D mode)
    const int y_first = q0 / d.Wo;
    const int q_last = min(q0 + P.Ptile, P.HoWo) - 1;
    const int rows_out = q_last / d.Wo - y_first + 1;
    const int NR = (rows_out - 1) * d.stride + (d.kh - 1) * d.dil + 1;
    const int r_in0 = y_first * d.stride - d.pad;
    const int LWused = d.W + 2 * d.pad;

    // per-lane LDS offsets of this wave's pixels
    int poff[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        int q = pix_wave + pb * 16 + (lane & 15);
        q = q < P.HoWo ? q : q0;                // padding lanes compute a valid pixel and are never stored
        if (P.flat) {
            poff[pb] = q - q0;
        } else {
            int y = q / d.Wo, x = q - y * d.Wo;
            poff[pb] = (y - y_first) * d.stride * P.LW + x * d.stride;
        }
    }
    const int kl = lane >> 4;                   // k index of this lane inside a K-step of 4

    f32x4 acc[MB][PB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < d.Cin; c0 += P.CK) {
        __syncthreads();
        // ---- stage the input patch ------------------------------------------------------------
        if (P.flat) {
            const int total = P.CK * P.Ptile;
            for (int idx = tid; idx < total; idx += P.nthreads) {
                int c = idx / P.Ptile, j = idx - c * P.Ptile;
                int q = q0 + j;
                float v = 0.f;
                if (c0 + c < d.Cin && q < HW) {
                    size_t o = (size_t)(c0 + c) * HW + q;
                    v = in[in_base + o];
                    if (in2) v += in2[in2_base + o];
                }
                inp[c * P.CS + j] = v;
            }
        } else {
            const int nrows = P.CK * NR;
            for (int rid = wave; rid < nrows; rid += nwaves) {
                int c = rid / NR, r = rid - c * NR;
                int y = r_in0 + r;
                bool row_ok = (c0 + c < d.Cin) && y >= 0 && y < d.H;
                size_t o = (size_t)(c0 + c) * HW + (size_t)(row_ok ? y : 0) * d.W;
                float* dst = inp + c * P.CS + r * P.LW;
                for (int col = lane; col < LWused; col += 64) {
                    int x = col - d.pad;
                    float v = 0.f;
                    if (row_ok && x >= 0 && x < d.W) {
                        v = in[in_base + o + x];
                        if (in2) v += in2[in2_base + o + x];
                    }
                    dst[col] = v;
                }
            }
        }
        // ---- stage the weight slab: wts[(tap*CK + c)][m] = Wp[tap][c0+c][m_wg+m] -----------------
        {
            const int rows = P.KK * P.CK;
            const int total = rows * P.Mtile;
            for (int idx = tid; idx < total; idx += P.nthreads) {
                int row = idx / P.Mtile, m = idx - row * P.Mtile;
                int tap = row / P.CK, c = row - tap * P.CK;
                float v = 0.f;
                if (c0 + c < d.Cin && m_wg + m < P.Cout16)
                    v = wp[((size_t)tap * d.Cin + c0 + c) * P.Cout16 + m_wg + m];
                wts[row * P.MS + m] = v;
            }
        }
        __syncthreads();
        // ---- MFMA over this chunk ------------------------------------------------------------------
        for (int tap = 0; tap < P.KK; ++tap) {
            int ti = tap / d.kw, tj = tap - ti * d.kw;
            const int tap_off = P.flat ? 0 : (ti * d.dil) * P.LW + tj * d.dil;
            const float* wrow = wts + (tap * P.CK + kl) * P.MS + m_wave + (lane & 15);
            const float* irow = inp + kl * P.CS + tap_off;
            for (int kc = 0; kc < P.CK; kc += 4) {
                float a[MB], b[PB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a[mb] = wrow[kc * P.MS + mb * 16];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) b[pb] = irow[kc * P.CS + poff[pb]];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[pb], acc[mb][pb], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: scale/shift (+res) (+act), optional nearest-upsample accumulate -------------------
    const int f = d.res_up > 1 ? d.res_up : 1;
    const int HWo_hi = P.HoWo * f * f, Wo_hi = d.Wo * f;
    int qhi[PB];                                 // index of the (dy=0, dx=0) target pixel on the output grid
    bool qok[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int q = pix_wave + pb * 16 + (lane & 15);
        qok[pb] = q < P.HoWo;
        if (f == 1) {
            qhi[pb] = q;
        } else {
            const int y = q / d.Wo, x = q - y * d.Wo;
            qhi[pb] = y * f * Wo_hi + x * f;
        }
    }
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
            const int sub = dy * Wo_hi + dx;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = m_wg + m_wave + mb * 16 + kl * 4 + r;
                    const bool co_ok = co < d.Cout;
                    const float sc = (co_ok && scale) ? scale[co] : 1.f;
                    const float sh = (co_ok && shift) ? shift[co] : 0.f;
                    const size_t obase = ((size_t)n * d.out_ctot + d.out_coff + co) * HWo_hi + sub;
                    const size_t rbase = ((size_t)n * d.res_ctot + d.res_coff + co) * HWo_hi + sub;
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        if (co_ok && qok[pb]) {
                            float v = fmaf(acc[mb][pb][r], sc, sh);
                            if (res) v += res[rbase + qhi[pb]];
                            if (d.act == OTP_ACT_RELU) v = fmaxf(v, 0.f);
                            else if (d.act == OTP_ACT_GELU) v = gelu_erf(v);
                            out[obase + qhi[pb]] = v;
                        }
                    }
                }
            }
        }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                   int KK, int Cout16) {
    const int total = KK * Cin * Cout16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int co = i % Cout16, r = i / Cout16;
        int ci = r % Cin, tap = r / Cin;
        wp[i] = co < Cout ? w[((size_t)co * Cin + ci) * KK + tap] : 0.f;
    }
}

// stride between LDS rows such that (4 consecutive k rows) x (16 consecutive floats) hit distinct banks
int pad_stride(int n, int want_mod32) {
    int r = ((n + 31) / 32) * 32 + want_mod32;
    while (r - 32 >= n) r -= 32;
    return r;
}

int g_force[4] = {0, 0, 0, 0};   // test hook: forced (MB, PB, WM, WP)

bool choose_plan(ConvPlan& P) {
    const otp_conv_desc& d = P.d;
    P.KK = d.kh * d.kw;
    P.Cout16 = (d.Cout + 15) & ~15;
    P.HoWo = d.Ho * d.Wo;
    P.flat = (d.kh == 1 && d.kw == 1 && d.stride == 1 && d.pad == 0) ? 1 : 0;
    const int mblk = P.Cout16 / 16, G = (P.HoWo + 15) / 16;
    int best[4] = {1, 7, 1, 1};
    double best_cost = 1e300;
    static const int wms[] = {1, 2, 4}, pbs[] = {7, 8, 9};
    for (int MB = 1; MB <= 4; ++MB)
        for (int PB : pbs)
            for (int WM : wms)
                for (int WP = 1; WM * WP <= 4; ++WP) {
                    if (g_force[0] && (MB != g_force[0] || PB != g_force[1] || WM != g_force[2] || WP != g_force[3]))
                        continue;
                    if (!g_force[0] && WM > 1 && MB * (WM - 1) >= mblk) continue;     // whole waves of padding
                    if (!g_force[0] && WP > 1 && PB * (WP - 1) >= G) continue;
                    long mt = (mblk + MB * WM - 1) / (MB * WM), pt = (G + PB * WP - 1) / (PB * WP);
                    long nwg = (long)d.N * mt * pt;
                    double padded = (double)nwg * WM * WP * MB * PB;          // MFMA tiles issued per K-step
                    double waves = (double)nwg * WM * WP;
                    // one wave per SIMD keeps the matrix pipe busy; model the tail of the last round
                    double rounds = ceil(waves / 1024.0);
                    double cost = rounds * MB * PB;                          // time ~ rounds x work per wave
                    cost = cost * 1.0 + padded / 1024.0 * 0.25;              // mild preference for less padding
                    cost *= (1.0 + 0.15 / MB + 0.05 / (WM * WP));            // operand reuse / staging amortisation
                    if (cost < best_cost) { best_cost = cost; best[0] = MB; best[1] = PB; best[2] = WM; best[3] = WP; }
                }
    if (best_cost == 1e300) return false;
    P.WM = best[2]; P.WP = best[3];
    P.Mtile = 16 * best[0] * P.WM;
    P.Ptile = 16 * best[1] * P.WP;
    P.tiles_per_img = (P.HoWo + P.Ptile - 1) / P.Ptile;
    P.nthreads = 64 * P.WM * P.WP;
    P.MS = pad_stride(P.Mtile, 16);
    if (P.flat) {
        P.NRmax = 1; P.LW = P.Ptile;
        P.CS = pad_stride(P.Ptile, 16);
    } else {
        int rows_out = (P.Ptile + d.Wo - 1) / d.Wo + 1;
        if (rows_out > d.Ho) rows_out = d.Ho;
        P.NRmax = (rows_out - 1) * d.stride + (d.kh - 1) * d.dil + 1;
        P.LW = d.W + 2 * d.pad;
        P.CS = pad_stride(P.NRmax * P.LW, d.stride == 1 ? 16 : 17);
    }
    // largest channel chunk that keeps two workgroups per CU (or at least fits)
    const size_t budget = 72 * 1024, hard = OTP_LDS_LIMIT;
    int ck = 32;
    auto lds_of = [&](int c) { return ((size_t)c * P.CS + (size_t)P.KK * c * P.MS) * sizeof(float); };
    while (ck > 4 && lds_of(ck) > budget) ck >>= 1;
    if (lds_of(ck) > hard) return false;
    int cin4 = (d.Cin + 3) & ~3;
    if (ck > cin4) ck = cin4;
    P.CK = ck;
    return true;
}

template <int MB, int PB>
int launch(const float* in, const float* in2, const float* wp, const float* scale, const float* shift,
           const float* res, float* out, const ConvPlan& P, hipStream_t st) {
    size_t lds = ((size_t)P.CK * P.CS + (size_t)P.KK * P.CK * P.MS) * sizeof(float);
    auto kern = conv_igemm_kernel<MB, PB>;
    OTP_ALLOW_BIG_LDS(kern, lds);
    dim3 grid(P.d.N * P.tiles_per_img, (P.Cout16 + P.Mtile - 1) / P.Mtile);
    hipLaunchKernelGGL(kern, grid, dim3(P.nthreads), lds, st, in, in2, wp, scale, shift, res, out, P);
    return otp_launch_status();
}

}  // namespace

extern "C" int otp_conv2d_set_tile(int MB, int PB, int WM, int WP) {
    g_force[0] = MB; g_force[1] = PB; g_force[2] = WM; g_force[3] = WP;
    return OTP_OK;
}

extern "C" int otp_conv2d_pack_weight(const void* weight, void* wpacked, int Cout, int Cin, int kh, int kw,
                                      void* stream) {
    if (!weight || !wpacked || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0) return OTP_ERR_BAD_ARG;
    int Cout16 = (Cout + 15) & ~15, total = kh * kw * Cin * Cout16;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(otp_ceil_div(total, 256) > 1024 ? 1024 : otp_ceil_div(total, 256)),
                       dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float*>(weight),
                       static_cast<float*>(wpacked), Cout, Cin, kh * kw, Cout16);
    return otp_launch_status();
}

extern "C" int otp_conv2d(const void* in, const void* in2, const void* wpacked, const void* scale,
                          const void* shift, const void* res, void* out, const otp_conv_desc* desc, void* stream) {
    if (!in || !wpacked || !out || !desc) return OTP_ERR_BAD_ARG;
    ConvPlan P;
    P.d = *desc;
    const otp_conv_desc& d = P.d;
    if (d.N <= 0 || d.Cin <= 0 || d.H <= 0 || d.W <= 0 || d.Cout <= 0 || d.kh <= 0 || d.kw <= 0 || d.stride <= 0 ||
        d.pad < 0 || d.dil <= 0)
        return OTP_ERR_BAD_ARG;
    int Ho = (d.H + 2 * d.pad - (d.dil * (d.kh - 1) + 1)) / d.stride + 1;
    int Wo = (d.W + 2 * d.pad - (d.dil * (d.kw - 1) + 1)) / d.stride + 1;
    if (Ho != d.Ho || Wo != d.Wo || Ho <= 0 || Wo <= 0) return OTP_ERR_BAD_ARG;
    if (d.res_up > 1 && d.act == OTP_ACT_GELU) return OTP_ERR_UNSUPPORTED;
    if (!choose_plan(P)) return OTP_ERR_UNSUPPORTED;
    auto st = static_cast<hipStream_t>(stream);
    auto a = static_cast<const float*>(in);
    auto b = static_cast<const float*>(in2);
    auto w = static_cast<const float*>(wpacked);
    auto sc = static_cast<const float*>(scale);
    auto sh = static_cast<const float*>(shift);
    auto r = static_cast<const float*>(res);
    auto o = static_cast<float*>(out);
    const int MB = P.Mtile / (16 * P.WM), PB = P.Ptile / (16 * P.WP);
#define OTP_CASE(M_, P_) if (MB == M_ && PB == P_) return launch<M_, P_>(a, b, w, sc, sh, r, o, P, st);
    OTP_CASE(1, 7) OTP_CASE(1, 8) OTP_CASE(1, 9)
    OTP_CASE(2, 7) OTP_CASE(2, 8) OTP_CASE(2, 9)
    OTP_CASE(3, 7) OTP_CASE(3, 8) OTP_CASE(3, 9)
    OTP_CASE(4, 7) OTP_CASE(4, 8) OTP_CASE(4, 9)
#undef OTP_CASE
    return OTP_ERR_UNSUPPORTED;
}
