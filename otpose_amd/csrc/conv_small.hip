// 3x3 / stride 1 / pad 1 convolutions with a handful of channels (<= 24 in, <= 24 out) and an optional pre-added second
// input: the staircase convs of the RSB blocks (reference model/RSB.py:80-92, `conv(spx[i] + out_prev)` with 6 / 13 / 20
// channels at 96x72).  288 useful MACs per (pixel, input channel) is far too little for a matrix-core tile, and the generic
// implicit-GEMM kernel they used to run on (conv_igemm_kernel, 37-83 us per launch, latency bound) pads them to 16 x 16:
// here a thread owns one output pixel and all output channels in registers, the (in + in2) tile with its one-pixel halo and the
// weights (pre-transposed to [ci][tap][co] by otp_conv3x3_small_pack) are read with uniform addresses through the scalar cache,
// so the inner loop is one LDS read of the input value per (input channel, tap) feeding Cout FMAs with scalar weight operands
// - exact fp32.  Measured at 16 x 96x72: 15 / 17 / 29 us for 6 / 13 / 20 channels (conv_igemm_kernel: 37 / 37 / 83 us);
// still latency bound (weight fetches per input channel), not by the 4320 FMAs per pixel.  (Splitting the output channels over several lighter workgroups was
// measured slower: the launch is bound by the tile staging, not by occupancy.)
// Epilogue as otp_conv2d: y = act(scale * acc + shift); channel-sliced views for in / in2 / out.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SmallPlan {
    int N, Cin, H, W, HW, Cout;
    int in_ctot, in_coff, in2_ctot, in2_coff, out_ctot, out_coff, act;
    int TW, TH, tilesX, tilesY;              // output tile of a workgroup (TW * TH <= 256), tiles per image
};

// BCP: output channels padded to a multiple of 4 (accumulators per thread).  SPLIT = 2: 512 threads, the second group of four
// waves owns the second half of the input channels of the same 256 pixels and hands its partial sums over through LDS - twice
// the waves to hide the weight fetches behind (13 / 20 channels: 26 -> 18 / 44 -> 28 us).
template <int BCP, int SPLIT>
__global__ __launch_bounds__(256 * SPLIT) void conv3x3_small_kernel(const float* __restrict__ in, const float* __restrict__ in2,
                                                                    const float* __restrict__ w, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, float* __restrict__ out,
                                                                    const SmallPlan P) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NT = 256 * SPLIT;
    const int LW = P.TW + 2, LH = P.TH + 2, plane = LW * LH;
    float* tile = sm;                               // [Cin][LH][LW]
    float* part = sm + P.Cin * plane;               // SPLIT == 2: [BCP][256] partial sums of the second channel half
    const int tid = threadIdx.x;
    const int n = (int)blockIdx.x / (P.tilesX * P.tilesY), t = (int)blockIdx.x - n * (P.tilesX * P.tilesY);
    const int ty0 = (t / P.tilesX) * P.TH, tx0 = (t - (t / P.tilesX) * P.tilesX) * P.TW;
    const int co0 = (int)blockIdx.y * BCP;          // output-channel block of this workgroup (one block today)

    // input tile with halo: in (+ in2), zeros outside the image.  A thread owns up to 2 / SPLIT tile positions (the tile has at
    // most 512 of them) and walks the channels: no integer division inside the channel loop
    const float* ib = in + ((size_t)n * P.in_ctot + P.in_coff) * P.HW;
    const float* ib2 = in2 ? in2 + ((size_t)n * P.in2_ctot + P.in2_coff) * P.HW : nullptr;
    constexpr int NP = 2 / SPLIT;
    int pos[NP], off[NP];
#pragma unroll
    for (int s2 = 0; s2 < NP; ++s2) {
        const int r = tid + NT * s2, py = r / LW, px = r - py * LW;
        const int y = ty0 + py - 1, x = tx0 + px - 1;
        pos[s2] = r < plane ? r : -1;
        off[s2] = (r < plane && y >= 0 && y < P.H && x >= 0 && x < P.W) ? y * P.W + x : -1;
    }
    // four channels per trip: their (up to 16) loads are issued together - one channel per trip was one HBM / L2 round trip per
    // channel (2.7 us per input channel of the launch, most of it this latency)
    for (int c0 = 0; c0 < P.Cin; c0 += 4) {
        float v[4][NP];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int s2 = 0; s2 < NP; ++s2) {
                const bool ok = c0 + u < P.Cin && off[s2] >= 0;
                const size_t a = (size_t)(c0 + u) * P.HW + (ok ? off[s2] : 0);
                v[u][s2] = ok ? ib[a] : 0.f;
                if (ib2) v[u][s2] += ok ? ib2[a] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int s2 = 0; s2 < NP; ++s2)
                if (c0 + u < P.Cin && pos[s2] >= 0) tile[(c0 + u) * plane + pos[s2]] = v[u][s2];
        }
    }
    __syncthreads();

    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);      // 0 / 1: which half of the input channels (uniform per wave)
    const int pt = tid & 255;
    const int ly = pt / P.TW, lx = pt - ly * P.TW;
    const bool live = ly < P.TH && ty0 + ly < P.H && tx0 + lx < P.W;
    float acc[BCP];
#pragma unroll
    for (int o = 0; o < BCP; ++o) acc[o] = 0.f;
    const float* tp = tile + (live ? ly * LW + lx : 0);
    const int cmid = SPLIT == 2 ? (P.Cin + 1) / 2 : P.Cin;
    const int cbeg = half ? cmid : 0, cend = half ? P.Cin : cmid;
    constexpr int CI_UNROLL = BCP <= 16 ? 2 : 1;      // two channels' weights in flight (24 outputs: SGPR spills instead)
#pragma unroll CI_UNROLL
    for (int ci = cbeg; ci < cend; ++ci) {
        const float* tc = tp + ci * plane;
        // the weights of (ci, tap) are the same for every thread: uniform addresses, so they arrive through the scalar cache
        // (s_load) and feed the FMAs as scalar operands - the LDS serves one read per tap only
        const float* wc = w + (size_t)ci * 9 * BCP;
        float xin[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) xin[tap] = tc[(tap / 3) * LW + (tap % 3)];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int o = 0; o < BCP; ++o) acc[o] = fmaf(wc[tap * BCP + o], xin[tap], acc[o]);
    }
    if constexpr (SPLIT == 2) {
        if (half) {
#pragma unroll
            for (int o = 0; o < BCP; ++o) part[o * 256 + pt] = acc[o];
        }
        __syncthreads();
        if (half) return;
#pragma unroll
        for (int o = 0; o < BCP; ++o) acc[o] += part[o * 256 + pt];
    }
    if (!live) return;
    float* ob = out + ((size_t)n * P.out_ctot + P.out_coff) * P.HW + (size_t)(ty0 + ly) * P.W + tx0 + lx;
#pragma unroll
    for (int o = 0; o < BCP; ++o) {
        if (co0 + o < P.Cout) {
            float y = acc[o] * (scale ? scale[co0 + o] : 1.f) + (shift ? shift[co0 + o] : 0.f);
            if (P.act == OTP_ACT_RELU) y = fmaxf(y, 0.f);
            ob[(size_t)(co0 + o) * P.HW] = y;
        }
    }
}

__global__ void conv3x3_small_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int BCP) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * 9 * BCP) return;
    const int co = i % BCP, r = i / BCP, tap = r % 9, ci = r / 9;
    wt[i] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
}

int small_bcp(int Cout) { return Cout <= 8 ? 8 : (Cout <= 16 ? 16 : 24); }
int small_split(int Cin) { return Cin >= 12 ? 2 : 1; }      // input-channel halves on two wave groups

bool small_plan(const otp_conv_desc& d, SmallPlan& P, size_t& lds, int& bcp) {
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1 || d.res_up > 1 || d.frame_split > 0) return false;
    if (d.Cin < 1 || d.Cin > 24 || d.Cout < 1 || d.Cout > 24 || d.res_ctot > 0) return false;
    if (d.act != OTP_ACT_NONE && d.act != OTP_ACT_RELU) return false;
    if (d.Ho != d.H || d.Wo != d.W) return false;
    P.N = d.N; P.Cin = d.Cin; P.H = d.H; P.W = d.W; P.HW = d.H * d.W; P.Cout = d.Cout;
    P.in_ctot = d.in_ctot; P.in_coff = d.in_coff; P.in2_ctot = d.in2_ctot; P.in2_coff = d.in2_coff;
    P.out_ctot = d.out_ctot; P.out_coff = d.out_coff; P.act = d.act;
    // tile width: the one of 8 / 16 / 24 / 32 / 64 that wastes the fewest threads on this map
    int best = 0;
    double bestu = -1.0;
    const int cands[5] = {8, 16, 24, 32, 64};
    for (int tw : cands) {
        const int th = 256 / tw;
        const int tx = (d.W + tw - 1) / tw, ty = (d.H + th - 1) / th;
        const double u = (double)d.H * d.W / ((double)tx * ty * 256.0);
        if (u > bestu + 1e-9) { bestu = u; best = tw; }
    }
    P.TW = best; P.TH = 256 / best;
    P.tilesX = (d.W + P.TW - 1) / P.TW; P.tilesY = (d.H + P.TH - 1) / P.TH;
    bcp = small_bcp(d.Cout);                        // output channels per workgroup (accumulators per thread)
    if ((P.TW + 2) * (P.TH + 2) > 512) return false;
    lds = ((size_t)d.Cin * (P.TW + 2) * (P.TH + 2) + (small_split(d.Cin) == 2 ? (size_t)bcp * 256 : 0)) * sizeof(float);
    return lds <= 64 * 1024 && (long)d.N * P.tilesX * P.tilesY < (1l << 31);
}

}  // namespace

extern "C" int otp_conv3x3_small_supported(const otp_conv_desc* desc) {
    if (!desc) return 0;
    SmallPlan P{};
    size_t lds = 0;
    int bcp = 0;
    return small_plan(*desc, P, lds, bcp) ? 1 : 0;
}

extern "C" size_t otp_conv3x3_small_weight_bytes(int Cout, int Cin) {
    if (Cout < 1 || Cout > 24 || Cin < 1 || Cin > 24) return 0;
    return (size_t)Cin * 9 * small_bcp(Cout) * sizeof(float);
}

extern "C" int otp_conv3x3_small_pack(const void* weight, void* wpacked, int Cout, int Cin, void* stream) {
    if (!weight || !wpacked) return OTP_ERR_BAD_ARG;
    if (!otp_conv3x3_small_weight_bytes(Cout, Cin)) return OTP_ERR_UNSUPPORTED;
    const int bcp = small_bcp(Cout), total = Cin * 9 * bcp;
    hipLaunchKernelGGL(conv3x3_small_pack_kernel, dim3(otp_ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(weight), static_cast<float*>(wpacked), Cout, Cin, bcp);
    return otp_launch_status();
}

extern "C" int otp_conv3x3_small(const void* in, const void* in2, const void* weight, const void* scale, const void* shift,
                                 void* out, const otp_conv_desc* desc, void* stream) {
    if (!in || !weight || !out || !desc) return OTP_ERR_BAD_ARG;
    const otp_conv_desc& d = *desc;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0) return OTP_ERR_BAD_ARG;
    if (d.in_ctot < d.in_coff + d.Cin || d.out_ctot < d.out_coff + d.Cout || (in2 && d.in2_ctot < d.in2_coff + d.Cin))
        return OTP_ERR_BAD_ARG;
    SmallPlan P{};
    size_t lds = 0;
    int bcp = 0;
    if (!small_plan(d, P, lds, bcp)) return OTP_ERR_UNSUPPORTED;
    auto st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(d.N * P.tilesX * P.tilesY), (unsigned)((d.Cout + bcp - 1) / bcp));
    auto fi = static_cast<const float*>(in), f2 = static_cast<const float*>(in2), fw = static_cast<const float*>(weight);
    auto fs = static_cast<const float*>(scale), fh = static_cast<const float*>(shift);
    auto fo = static_cast<float*>(out);
#define OTP_SMALL(B_, S_) hipLaunchKernelGGL((conv3x3_small_kernel<B_, S_>), grid, dim3(256 * S_), lds, st, fi, f2, fw, fs, fh, fo, P)
    if (small_split(d.Cin) == 2) {
        if (bcp == 8) OTP_SMALL(8, 2);
        else if (bcp == 16) OTP_SMALL(16, 2);
        else OTP_SMALL(24, 2);
    } else {
        if (bcp == 8) OTP_SMALL(8, 1);
        else if (bcp == 16) OTP_SMALL(16, 1);
        else OTP_SMALL(24, 1);
    }
#undef OTP_SMALL
    return otp_launch_status();
}
