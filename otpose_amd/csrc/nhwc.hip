// bf16 training path of the HRNet backbone (BASELINE configs[2]: "bf16 training step"; the reference step is
// script/Common.py:118-144 over model/HRNet.py:116-152): activations live in HBM as NHWC bfloat16 (channel stride
// rounded up to 8, padding channels zero), every contraction runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation,
// BatchNorm statistics / gradients and the weight gradients are fp32, master weights stay fp32 (cast while packing).
//
// Why NHWC here when the fp32 path is NCHW: a bf16 MFMA operand is 8 consecutive k-values per lane; with channels
// innermost both the activations (k = input channel) and the packed weights deliver a fragment as ONE 16-byte LDS
// read, taps shift whole pixels (16-byte aligned), and nothing is transposed on the way in or out.
//
//   nhwc_conv_kernel<MB,NB>   implicit GEMM  out[px][co] = sum_{tap,ci} W[co][tap][ci] * x[px+tap][ci]   (forward and,
//                             with flipped / transposed weights, the input gradient); the epilogue also leaves the
//                             per-tile sum / sum of squares of every output channel (BatchNorm batch statistics)
//   nhwc_wgrad_kernel         dW[co][tap][ci] = sum_px gy[px][co] * x[px+tap][ci]: both operands are read from their
//                             [pixel][channel] LDS images with ds_read_b64_tr_b16 (the hardware transpose read), so the
//                             contraction index (pixels) lands in the fragment's k slots without a transposed copy
//   bn_* / upsample_add_*     HBM-bound NHWC passes (16-byte accesses, fp32 arithmetic)
#include <stdlib.h>

#include "common.h"
#include "hb.h"
#include <cstring>

namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }

// 16-byte range-checked buffer load: staging loops mask a unit by pointing it past the descriptor (the hardware returns
// zeros) instead of branching around the load - hipcc waits for every outstanding load at each such branch, which turns a
// batch of independent loads into a chain of L2 round trips.
constexpr int OOB = -16;

#ifdef OTP_NHWC_TIMING
// development build only (tools/nhwc_timing.py): per-workgroup phase stamps of nhwc_conv_kernel, never in libotpose_hip.so
__device__ unsigned long long otp_nhwc_stamps[8192 * 8];
#define OTP_STAMP(slot)                                                                              \
    do {                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 8192) otp_nhwc_stamps[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define OTP_STAMP(slot)
#endif
__device__ __forceinline__ u32x4 bload16(otp_rsrc r, int voff_bytes) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, 0, 0));
}

// ---------------------------------------------------------------------------------------------------------------------
// plan: a pure function of the descriptor (host), shared by the packer and the launcher
// ---------------------------------------------------------------------------------------------------------------------
struct ConvPlan {
    int N, H, W, Cin, CinS, Ho, Wo, Cout, CoutS, kh, kw, stride, pad, dil, out_mode;
    int CK, CK8, CKp, nChunks, KGc, KS;      // channels per chunk, k-groups (8 channels of one tap) and k-steps per chunk
    int MB, NB, BM, nM, P, tilesPerImg;
    int RW, rowsMax;
    int ldsW, ldsX, ldsTab, lds;
    int pipe;                                 // the chunk pipeline of nhwc_conv_kernel applies (see there)
    size_t wbytes;
};

int round_up(int a, int b) { return (a + b - 1) / b * b; }

// sizes everything for a given (channels per chunk, N-blocks per wave); false when the LDS image does not fit
bool size_plan(const otp_nhwc_conv_desc* d, ConvPlan* p, int ck, int nb) {
    const int taps = d->kh * d->kw;
    p->H = d->H, p->W = d->W;
    p->Ho = (d->H + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
    p->Wo = (d->W + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    p->CK = ck, p->CK8 = ck / 8;
    p->CKp = (p->CK8 & 1) ? ck : ck + 8;        // pixel stride of the LDS window: an odd multiple of 16 bytes
    p->nChunks = (p->CinS + ck - 1) / ck;
    p->KGc = taps * p->CK8;
    p->KS = (p->KGc + 3) / 4;
    const int npx = p->Ho * p->Wo;
    p->NB = nb;
    p->P = 64 * nb;
    p->tilesPerImg = (npx + p->P - 1) / p->P;
    // a pointwise conv sees the image as rows of W' pixels, W' the largest divisor of H*W that is <= the tile size (and a
    // multiple of 8): a tile's window is then (nearly) the tile itself instead of the full-width image rows it touches -
    // (B, C, T) sequences enter as H = 1, W = T
    if (taps == 1 && d->stride == 1 && d->pad == 0) {
        int w1 = 0;
        for (int cand = p->P; cand >= 32; cand -= 8)
            if (npx % cand == 0) { w1 = cand; break; }
        if (w1) {
            p->W = p->Wo = w1;
            p->H = p->Ho = npx / w1;
        }
    }
    const int need = (p->Wo - 1) * d->stride + (d->kw - 1) * d->dil + 1;
    p->RW = p->W + 2 * d->pad > need ? p->W + 2 * d->pad : need;
    int rowsOut = p->Wo % p->P == 0 ? 1 : (p->P % p->Wo == 0 ? p->P / p->Wo : (p->P - 1 + p->Wo - 1) / p->Wo + 1);
    if (rowsOut > p->Ho) rowsOut = p->Ho;
    p->rowsMax = (rowsOut - 1) * d->stride + (d->kh - 1) * d->dil + 1;
    p->ldsW = p->KS * 4 * p->BM * 16;
    p->ldsX = round_up(p->rowsMax * p->RW * p->CKp * 2, 16);
    const int outTile = p->P * (p->BM + 8) * 2;                 // the epilogue's [pixel][channel] image reuses the operand space
    if (p->ldsW + p->ldsX < outTile) p->ldsX = outTile - p->ldsW;
    p->ldsTab = round_up(p->KS * 4 * 4, 16);
    p->lds = p->ldsW + p->ldsX + p->ldsTab + 4 * 2 * p->BM * 4;
    p->wbytes = (size_t)p->nChunks * p->nM * p->ldsW;
    {
        // chunk pipeline: 256 threads hold one chunk's window (one (column, channel group) unit x 8 rows per thread, 256 / units-per-
        // row row groups) and weight slab (PIPE_WU(MB) 16-byte units per thread) in registers
        const int rowUnits = p->RW * p->CK8;
        static const int min_chunks = getenv("OTP_NHWC_PIPE_MIN_CHUNKS") ? atoi(getenv("OTP_NHWC_PIPE_MIN_CHUNKS")) : 3;
        p->pipe = p->KS == 7 && p->nChunks >= min_chunks && rowUnits <= 256 && p->rowsMax <= 8 * (256 / rowUnits) &&
                  p->ldsW / 16 <= 256 * ((448 * (p->BM / 16) + 255) / 256);
    }
    return p->lds <= OTP_LDS_LIMIT;
}

bool make_plan(const otp_nhwc_conv_desc* d, ConvPlan* p) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->kh <= 0 || d->kw <= 0 ||
        d->stride <= 0 || d->dil <= 0 || d->pad < 0)
        return false;
    p->N = d->N, p->Cin = d->Cin, p->Cout = d->Cout;
    p->kh = d->kh, p->kw = d->kw, p->stride = d->stride, p->pad = d->pad, p->dil = d->dil, p->out_mode = d->out_mode;
    p->CinS = round_up(d->Cin, 8), p->CoutS = round_up(d->Cout, 8);
    const int ho = (d->H + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    if (ho <= 0 || wo <= 0) return false;
    // channels per chunk: 24 keeps the per-chunk weight slab small enough for 3-4 workgroups per CU
    const int taps = d->kh * d->kw;
    int ck = p->CinS <= 24 ? p->CinS : 24;
    if (taps == 1) {
        // pointwise: one chunk when the channels fit (136 = the temporal encoders' width), else equal chunks of 136 / 72 / 40
        if (p->CinS <= 144) ck = p->CinS;
        else if (p->CinS % 136 == 0) ck = 136;
        else if (p->CinS % 72 == 0) ck = 72;
        else ck = p->CinS % 40 == 0 ? 40 : (p->CinS % 24 == 0 ? 24 : 40);
    }
    const int c16 = round_up(d->Cout, 16);
    int bm = c16 <= 96 ? c16 : 96;
    if (c16 > 96 && c16 % 96 != 0 && c16 % 64 == 0) bm = 64;
    if (taps == 1 && c16 > 96) {                                // pointwise: up to 9 M-blocks, least padding first
        int best = 1 << 30;
        for (int cand = 144; cand >= 96; cand -= 16) {
            const int padded = (c16 + cand - 1) / cand * cand;
            if (padded < best) best = padded, bm = cand;
        }
    }
    p->BM = bm, p->MB = bm / 16, p->nM = (c16 + bm - 1) / bm;
    const int npx = ho * wo;
    // 256-pixel tiles (4 N-blocks per wave: 10 LDS fragment reads per 24 MFMAs at MB = 6 instead of 8 per 12) when that
    // still leaves >= 4 workgroups per CU to balance; 128-pixel tiles otherwise
    int nb = ((long)d->N * ((npx + 255) / 256) * p->nM >= 1024) ? 4 : 2;
    if (p->MB > 6) nb = 2;                                      // 7-9 M-blocks: 72 accumulator registers at NB = 2
    static const int nb_env = getenv("OTP_NHWC_NB") ? atoi(getenv("OTP_NHWC_NB")) : 0;      // tuning override (2 or 4)
    if (nb_env && p->MB <= 6) nb = nb_env;
    // candidates in order of preference; later ones shrink the LDS image (widely dilated taps on wide maps stage many rows)
    if (size_plan(d, p, ck, nb) && (taps > 1 || nb == 2 || p->lds <= 72 * 1024)) return true;
    if (nb == 4 && size_plan(d, p, ck, 2)) return true;
    if (ck > 8 && size_plan(d, p, 8, 2)) return true;
    return false;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight packing: fp32 (O, I, kh, kw) master weights -> bf16 [chunk][m-tile][k-group (KS*4)][BM][8]
// element (o, i, dy, dx) of the EFFECTIVE conv is w[base + o*so + i*si + dy*sdy + dx*sdx] (the input-gradient conv
// reads the same tensor with o <-> i swapped and the taps flipped)
// ---------------------------------------------------------------------------------------------------------------------
struct PackJob {                       // one weight re-layout: everything nhwc_pack_kernel takes by value
    const float* w;
    bf16* out;
    long so, si, sdy, sdx, base;
    size_t total;
    ConvPlan p;
    int hb, hbNTW, hbChunks;           // hb = 1: the operator of csrc/hb.hip's 3x3 kernel (layout: csrc/hb.h), p only carries Cin / Cout
    int hbKS;                          // hb = 2: of its 1x1 kernel, hbKS k-steps
};

// 3x3 / pad 1 / stride 1 or 2 convolutions with Cin % 16 == 0 run on csrc/hb.hip's kernel (the fp16 engine's conv for bfloat16
// NHWC tensors: whole-Cin window by LDS-DMA, weights streamed through registers, three workgroups per CU) where that is faster
// than nhwc_conv_kernel (otp_hb_pays: 96 / 192-channel branches, the stride-2 fuse convs).  OTPOSE_NHWC_HB=0 keeps every layer on nhwc_conv_kernel, =2 sends every
// shape the window kernel covers to it, whether it pays or not (A/B, tests).
bool use_hb(const otp_nhwc_conv_desc* d) {
    const char* e = getenv("OTPOSE_NHWC_HB");                      // (read per call: the tests switch it inside one process)
    const int mode = e ? atoi(e) : 1;
    return mode != 0 && d && (mode == 2 || otp_hb_pays(d)) && otp_hb_supported(d);
}

// 1x1 convolutions - the two projections of a TransformerBlock's MLP on (N, 1, T, C) sequences, HRNet's bottleneck / fuse 1x1s, and
// their input gradients - run on csrc/hb.hip's pointwise kernel (the input register-resident, the weights streamed through the LDS): nhwc_conv_kernel re-stages
// 46 KB of weights per 128-token tile and 136-channel chunk with one workgroup per CU (134 / 214 us per projection at cfg2).
// Image-shaped 1x1 layers (HRNet's bottleneck and fuse 1x1s, with BatchNorm statistics) too: forward / input gradient at 80 frames
// (tools/bf16_conv_bench.py, OTPOSE_NHWC_HB=2 against =0) 256 -> 64 @96x72 80.8 / 94.2 us against 174 / 141, 64 -> 256 106.7 / 81.7 against
// 156 / 176, 96 -> 48 @48x36 12.1 against 30.0, 384 -> 48 @12x9 8.9 against 31.7 - every shape measured.
bool hbpw_pays(const otp_nhwc_conv_desc*) { return true; }
bool use_hbpw(const otp_nhwc_conv_desc* d) {
    const char* e = getenv("OTPOSE_NHWC_HB");
    const char* e2 = getenv("OTPOSE_NHWC_HBPW");                   // (=0: only the 1x1 kernel off)
    const int mode = e ? atoi(e) : 1;
    if (mode == 0 || (e2 && atoi(e2) == 0) || !d) return false;
    return (mode == 2 || d->H == 1 || hbpw_pays(d)) && otp_hbpw_supported(d);
}

__device__ __forceinline__ float pack_hbpw_value(const float* __restrict__ w, const PackJob& jb, size_t i) {
    const int j = (int)(i & 7);
    const size_t r = i >> 3;
    const int units = otp_hbpw_blkb(jb.hbKS) / 16;
    const int blk = (int)(r / units), u = (int)(r % units);
    if (u >= 2 * jb.hbKS * 64) return 0.f;
    const int frag = u >> 6, lane = u & 63, m = frag / jb.hbKS, ks = frag - m * jb.hbKS, r16 = lane & 15, kq = lane >> 4;
    const int o = 32 * blk + 8 * (r16 >> 2) + 4 * m + (r16 & 3), ci = 32 * ks + 8 * kq + j;
    if (o >= jb.p.Cout || ci >= jb.p.Cin) return 0.f;
    return w[jb.base + o * jb.so + ci * jb.si];
}

// element i of the packed operator of csrc/hb.hip's kernel: [cout block][16-channel chunk][16-byte unit (csrc/hb.h)][8 channels]
__device__ __forceinline__ float pack_hb_value(const float* __restrict__ w, const PackJob& jb, size_t i) {
    const int j = (int)(i & 7);
    size_t r = i >> 3;
    const int WU = otp_hb_wb(jb.hbNTW) / 16;
    const int u = (int)(r % WU); r /= WU;
    const int chunk = (int)(r % jb.hbChunks), cb = (int)(r / jb.hbChunks);
    int s, t, lane;
    otp_hb_unit(u, jb.hbNTW, &s, &t, &lane);
    const int o = otp_hb_row2ch(cb * jb.hbNTW * 16, t, lane & 15, jb.hbNTW, jb.p.Cout);
    const int q = 4 * s + (lane >> 4), tap = q >> 1, ci = chunk * 16 + 8 * (q & 1) + j;
    if (tap > 8 || o >= jb.p.Cout || ci >= jb.p.Cin) return 0.f;
    return w[jb.base + o * jb.so + ci * jb.si + (tap / 3) * jb.sdy + (tap % 3) * jb.sdx];
}

__device__ __forceinline__ void pack_range(const float* __restrict__ w, bf16* __restrict__ out, const ConvPlan& p, long so, long si,
                                           long sdy, long sdx, long base, size_t total, size_t first, size_t step) {
    for (size_t i = first; i < total; i += step) {
        size_t r = i;
        const int j = r % 8; r /= 8;
        const int co = r % p.BM; r /= p.BM;
        const int kg = r % (p.KS * 4); r /= (p.KS * 4);
        const int mt = r % p.nM; r /= p.nM;
        const int ch = (int)r;
        float v = 0.f;
        if (kg < p.KGc) {
            const int tap = kg / p.CK8, cgi = kg % p.CK8;
            const int ci = ch * p.CK + cgi * 8 + j, o = mt * p.BM + co;
            if (ci < p.Cin && o < p.Cout) v = w[base + o * so + ci * si + (tap / p.kw) * sdy + (tap % p.kw) * sdx];
        }
        out[i] = (bf16)v;
    }
}

// One 16-byte unit (8 consecutive input channels of one (output channel, tap)) of a packed operator: the index arithmetic once per unit,
// the eight weight loads in flight together, one 16-byte store.  The element-wise forms above cost the batched pack 741 us at the head
// of every training step (a dependent scalar load per element, six integer divisions each; 160 iterations per thread on the 384 x 384
// layers).
__device__ __forceinline__ void pack_unit(const PackJob& jb, size_t unit) {
    const float* __restrict__ w = jb.w;
    long off = -1;                       // offset of the unit's first weight; consecutive input channels are jb.si apart
    int ci0 = 0;
    if (jb.hb == 1) {
        size_t r = unit;
        const int WU = otp_hb_wb(jb.hbNTW) / 16;
        const int u = (int)(r % WU); r /= WU;
        const int chunk = (int)(r % jb.hbChunks), cb = (int)(r / jb.hbChunks);
        int s_, t, lane;
        otp_hb_unit(u, jb.hbNTW, &s_, &t, &lane);
        const int o = otp_hb_row2ch(cb * jb.hbNTW * 16, t, lane & 15, jb.hbNTW, jb.p.Cout);
        const int q = 4 * s_ + (lane >> 4), tap = q >> 1;
        ci0 = chunk * 16 + 8 * (q & 1);
        if (tap <= 8 && o < jb.p.Cout) off = jb.base + o * jb.so + (tap / 3) * jb.sdy + (tap % 3) * jb.sdx;
    } else if (jb.hb == 2) {
        const int units = otp_hbpw_blkb(jb.hbKS) / 16;
        const int blk = (int)(unit / units), u = (int)(unit % units);
        if (u < 2 * jb.hbKS * 64) {
            const int frag = u >> 6, lane = u & 63, m = frag / jb.hbKS, ks = frag - m * jb.hbKS, r16 = lane & 15, kq = lane >> 4;
            const int o = 32 * blk + 8 * (r16 >> 2) + 4 * m + (r16 & 3);
            ci0 = 32 * ks + 8 * kq;
            if (o < jb.p.Cout) off = jb.base + o * jb.so;
        }
    } else {
        const ConvPlan& p = jb.p;
        size_t r = unit;
        const int co = r % p.BM; r /= p.BM;
        const int kg = r % (p.KS * 4); r /= (p.KS * 4);
        const int mt = r % p.nM; r /= p.nM;
        const int ch = (int)r;
        if (kg < p.KGc) {
            const int tap = kg / p.CK8, cgi = kg % p.CK8, o = mt * p.BM + co;
            ci0 = ch * p.CK + cgi * 8;
            if (o < p.Cout) off = jb.base + o * jb.so + (tap / p.kw) * jb.sdy + (tap % p.kw) * jb.sdx;
        }
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (off >= 0 && ci0 + j < jb.p.Cin) ? w[off + (long)(ci0 + j) * jb.si] : 0.f;
    bf16x8 o8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = (bf16)v[j];
    *reinterpret_cast<bf16x8*>(jb.out + unit * 8) = o8;
}

__global__ void nhwc_pack_kernel(const float* __restrict__ w, bf16* __restrict__ out, ConvPlan p, long so, long si, long sdy,
                                 long sdx, long base, size_t total) {
    pack_range(w, out, p, so, si, sdy, sdx, base, total, blockIdx.x * (size_t)blockDim.x + threadIdx.x,
               (size_t)gridDim.x * blockDim.x);
}

__global__ void nhwc_pack_hb_kernel(PackJob jb) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < jb.total; i += (size_t)gridDim.x * blockDim.x)
        jb.out[i] = (bf16)(jb.hb == 2 ? pack_hbpw_value(jb.w, jb, i) : pack_hb_value(jb.w, jb, i));
}

// every weight of a training step in one launch: blockIdx.y = job (the table lives in device memory, its fields arrive
// through the scalar cache), blockIdx.x strides over the job's elements
__global__ void nhwc_pack_batch_kernel(const PackJob* __restrict__ jobs) {
    const PackJob& jb = jobs[blockIdx.y];
    const size_t units = jb.total >> 3;                              // (every layout is whole 16-byte units)
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) pack_unit(jb, u);
}

// ---------------------------------------------------------------------------------------------------------------------
// implicit-GEMM convolution
// ---------------------------------------------------------------------------------------------------------------------
// bias, rounding, store; per-tile channel sums of the ROUNDED values (what BatchNorm will normalise).  Contains one
// workgroup barrier when statistics are requested.
// sum over the 16 lanes of a DPP row (one MFMA pixel column group) with four VALU adds; every lane ends with the total.
// (__shfl_xor with offsets 4 and 8 lowers to ds_bpermute on gfx950: 96 LDS round trips in the old epilogue)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);       // row_ror:4
    v += dpp_mov<0x128>(v);       // row_ror:8
    return v;
}

// bias, rounding, store; per-tile channel sums of the ROUNDED values (what BatchNorm will normalise).
// NHWC output: the tile is transposed through LDS (smem is free once every wave has left the MFMA loop) so that it leaves as
// 16-byte stores of whole pixel rows - the accumulator layout (4 channels x 1 pixel per lane) would otherwise issue MB*NB
// 8-byte stores per lane in 32-byte runs, which is store-issue bound (10-16k cycles per workgroup, measured with
// tools/nhwc_timing.py).  Contains workgroup barriers; every thread of the workgroup must call it.
template <int MB, int NB>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[MB][NB], const bool (&valid)[NB], const ConvPlan& p,
                                              const float* __restrict__ bias, const bf16* __restrict__ res,
                                              bf16* __restrict__ out, float* __restrict__ out_f32, float* __restrict__ stats,
                                              unsigned char* smem, float* sRed, int n, int tile, int mt, int p0, int npx) {
    constexpr int BM = MB * 16, P = 64 * NB, BMS = BM + 8;      // LDS row stride: 16-byte aligned, odd multiple of 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    if (p.out_mode == 1) {                                       // fp32 NCHW (hand-over to the fp32 NCHW kernels)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int co = mt * BM + m * 16 + lg * 4;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int pix = p0 + (wave * NB + nb) * 16 + l15;
                if (valid[nb]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.Cout)
                            out_f32[((size_t)n * p.Cout + co + r) * npx + pix] = acc[m][nb][r] + (bias ? bias[co + r] : 0.f);
                }
            }
        }
        return;
    }
    bf16* sOut = reinterpret_cast<bf16*>(smem);
    __syncthreads();                                             // every wave is done with the operand images
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int co = mt * BM + m * 16 + lg * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co + r < p.Cout) bv[r] = bias[co + r];
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = (bf16)(acc[m][nb][r] + bv[r]);
                const float f = valid[nb] ? bf2f(o[r]) : 0.f;
                s1[r] += f;
                s2[r] += f * f;
            }
            *reinterpret_cast<bf16x4*>(sOut + ((wave * NB + nb) * 16 + l15) * BMS + m * 16 + lg * 4) = o;
        }
        if (stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = row16_sum(s1[r]), b = row16_sum(s2[r]);
                if (l15 == 0) {
                    sRed[(wave * 2 + 0) * BM + m * 16 + lg * 4 + r] = a;
                    sRed[(wave * 2 + 1) * BM + m * 16 + lg * 4 + r] = b;
                }
            }
        }
    }
    __syncthreads();
    // rows of the tile -> global: (pixel, 8-channel group) units, consecutive threads = consecutive bytes of a pixel row
    int cvalid = p.CoutS - mt * BM;
    if (cvalid > BM) cvalid = BM;
    const int cu8 = cvalid / 8, units = P * cu8;
    const int pmax = min(P, npx - p0);
    for (int u = tid; u < units; u += 256) {
        const int px = u / cu8, cg = u - px * cu8;
        if (px < pmax) {
            const size_t o = ((size_t)n * npx + p0 + px) * p.CoutS + mt * BM + cg * 8;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(sOut + px * BMS + cg * 8);
            if (res) {                                           // out = bf16(conv) + res, rounded once more: what a separate
                const bf16x8 r = *reinterpret_cast<const bf16x8*>(res + o);       // bf16 add kernel would produce
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)(bf2f(v[e]) + bf2f(r[e]));
            }
            *reinterpret_cast<bf16x8*>(out + o) = v;
        }
    }
    if (stats) {
        for (int i = tid; i < 2 * BM; i += 256) {
            const int which = i / BM, c = i - which * BM, co = mt * BM + c;
            if (co < p.CoutS) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += sRed[(w * 2 + which) * BM + c];
                stats[((size_t)(n * p.tilesPerImg + tile) * 2 + which) * p.CoutS + co] = v;
            }
        }
    }
}

template <int MB, int NB, bool K7, bool PIPE>
__global__ __launch_bounds__(256) void nhwc_conv_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wpk,
                                                         const float* __restrict__ bias, const bf16* __restrict__ res,
                                                         bf16* __restrict__ out, float* __restrict__ out_f32,
                                                         float* __restrict__ stats, ConvPlan p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* sW = reinterpret_cast<bf16*>(smem);
    bf16* sX = reinterpret_cast<bf16*>(smem + p.ldsW);
    int* sTab = reinterpret_cast<int*>(smem + p.ldsW + p.ldsX);
    float* sRed = reinterpret_cast<float*>(smem + p.ldsW + p.ldsX + p.ldsTab);      // [4 waves][2][BM]
    constexpr int P = 64 * NB, BM = MB * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int mt = bid % p.nM;
    bid /= p.nM;
    const int tile = bid % p.tilesPerImg, n = bid / p.tilesPerImg;
    const int npx = p.Ho * p.Wo;
    const int p0 = tile * P, p1 = min(p0 + P, npx);
    const int oy0 = p0 / p.Wo, oy1 = (p1 - 1) / p.Wo;
    const int rowLo = oy0 * p.stride - p.pad;
    const int nrows = (oy1 - oy0) * p.stride + (p.kh - 1) * p.dil + 1;

    OTP_STAMP(0);
    // k-group -> element offset inside the window (tap shift + channel group); identical for every chunk
    for (int kg = tid; kg < p.KS * 4; kg += 256) {
        int v = 0;
        if (kg < p.KGc) {
            const int tap = kg / p.CK8, cgi = kg - tap * p.CK8;
            const int dy = tap / p.kw, dx = tap - dy * p.kw;
            v = ((dy * p.dil) * p.RW + dx * p.dil) * p.CKp + cgi * 8;
        }
        sTab[kg] = v;
    }
    int boff[NB];
    bool valid[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int pix = p0 + (wave * NB + nb) * 16 + l15;
        valid[nb] = pix < p1;
        const int pc = valid[nb] ? pix : p1 - 1;
        const int oy = pc / p.Wo, ox = pc - oy * p.Wo;
        boff[nb] = (((oy - oy0) * p.stride) * p.RW + ox * p.stride) * p.CKp;
    }
    f32x4 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wunits = p.ldsW / 16;
    const int rowUnits = p.RW * p.CK8;
    const otp_rsrc xres = make_rsrc(x, (size_t)p.N * p.H * p.W * p.CinS * 2);
    OTP_STAMP(1);
    if constexpr (K7 && PIPE) {
        {
            // ---- chunk pipeline (3x3 taps x 24-channel chunks, more than one chunk): the window and the weight slab of chunk c + 1
            // are loaded into REGISTERS while chunk c is multiplied, and go to the LDS between two barriers.  Without it a
            // workgroup's life at 48 -> 48 @96x72 is 29 k cycles for 6 k of MFMA work - load -> LDS store -> barrier -> multiply
            // with nothing in flight, hidden only by the other workgroups of the CU.  No branch around
            // the loads (hipcc would wait for every outstanding load at the join): a load that is not wanted points past its
            // descriptor and returns zeros.
            constexpr int WU = (448 * MB + 255) / 256;                 // 16-byte weight units per thread: 7 k-steps x 4 x BM x 16 B
            const int G = 256 / rowUnits, grp = tid / rowUnits, cu = tid - grp * rowUnits;
            const int col = cu / p.CK8, cgi = cu - col * p.CK8, ix = col - p.pad;
            const bool colIn = grp < G && ix >= 0 && ix < p.W;
            const int ldst = col * p.CKp + cgi * 8;
            const otp_rsrc wres = make_rsrc(wpk, p.wbytes);
            u32x4 xv[8], wv[WU];
            auto issue = [&](int ch) __attribute__((always_inline)) {
                const int c = ch * p.CK + cgi * 8;
                const bool colOK = colIn && ch < p.nChunks && c < p.CinS;
                const int gcol = (ix * p.CinS + c) * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = grp * 8 + j, iy = rowLo + r;
                    const bool ok = colOK && r < nrows && iy >= 0 && iy < p.H;
                    xv[j] = bload16(xres, ok ? (n * p.H + iy) * (p.W * p.CinS * 2) + gcol : OOB);
                }
                const int wbase = (ch * p.nM + mt) * wunits * 16;
#pragma unroll
                for (int i = 0; i < WU; ++i) {
                    const int u = tid + 256 * i;
                    wv[i] = bload16(wres, (u < wunits && ch < p.nChunks) ? wbase + u * 16 : OOB);
                }
            };
            issue(0);
            for (int ch = 0; ch < p.nChunks; ++ch) {
                if (ch) __syncthreads();                                  // every wave has left the previous chunk's images
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (grp < G && grp * 8 + j < nrows) *reinterpret_cast<u32x4*>(sX + (grp * 8 + j) * (p.RW * p.CKp) + ldst) = xv[j];
#pragma unroll
                for (int i = 0; i < WU; ++i)
                    if (tid + 256 * i < wunits) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(sW) + (tid + 256 * i) * 16) = wv[i];
                __syncthreads();
                if (ch == 0) OTP_STAMP(2);
                issue(ch + 1);                                            // in flight under this chunk's MFMAs
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    const int koff = sTab[ks * 4 + lg];
                    bf16x8 a[MB];
#pragma unroll
                    for (int m = 0; m < MB; ++m) a[m] = *reinterpret_cast<const bf16x8*>(sW + ((ks * 4 + lg) * BM + m * 16 + l15) * 8);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(sX + boff[nb] + koff);
#pragma unroll
                        for (int m = 0; m < MB; ++m) acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b, acc[m][nb], 0, 0, 0);
                    }
                }
                if (ch == 0) OTP_STAMP(3);
            }
            OTP_STAMP(4);
            conv_epilogue<MB, NB>(acc, valid, p, bias, res, out, out_f32, stats, smem, sRed, n, tile, mt, p0, npx);
            OTP_STAMP(5);
            return;
        }
    }
    for (int ch = 0; ch < p.nChunks; ++ch) {
        if (ch) __syncthreads();
        // One L2 round trip per chunk: the first 8 weight units AND the first 8 window rows of this thread are all in
        // flight before the first LDS store (weights: one contiguous slab per (chunk, m-tile); window: rows
        // [rowLo, rowLo + nrows) x columns [-pad, RW - pad) x CK channels, zeros outside the image; a thread owns one
        // (column, channel group) and walks the rows, so the column arithmetic is done once per chunk).
        const int wbase = (ch * p.nM + mt) * wunits * 16;
        const int c0 = ch * p.CK;
        u32x4 xv[8];
        // weight slab of this (chunk, m-tile): global -> LDS by the LDS-DMA (the packed slab IS the LDS image: unit i lands at
        // sW + 16 i), no staging registers, no ds_write pass
        for (int u0 = wave * 64; u0 < wunits; u0 += 256)
            if (u0 + lane < wunits)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(wpk) + wbase + (size_t)(u0 + lane) * 16),
                    (__attribute__((address_space(3))) void*)(reinterpret_cast<unsigned char*>(sW) + u0 * 16), 16, 0, 0);
        const int col0 = tid / p.CK8, cg0 = tid - col0 * p.CK8;
        const int ix0 = col0 - p.pad;
        const bool colOK0 = tid < rowUnits && ix0 >= 0 && ix0 < p.W && c0 + cg0 * 8 < p.CinS;
        const int gcol0 = (ix0 * p.CinS + c0 + cg0 * 8) * 2, ldst0 = col0 * p.CKp + cg0 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = rowLo + j;
            const bool ok = colOK0 && j < nrows && iy >= 0 && iy < p.H;
            xv[j] = bload16(xres, ok ? (n * p.H + iy) * (p.W * p.CinS * 2) + gcol0 : OOB);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (tid < rowUnits && j < nrows) *reinterpret_cast<u32x4*>(sX + j * (p.RW * p.CKp) + ldst0) = xv[j];
        // the rest (rows past 8, rows wider than 256 units)
        for (int sub = 0; sub * 256 < rowUnits; ++sub) {
            const int cu = sub * 256 + tid;
            const int col = cu / p.CK8, cgi = cu - col * p.CK8;
            const int ix = col - p.pad, c = c0 + cgi * 8;
            const bool colOK = cu < rowUnits && ix >= 0 && ix < p.W && c < p.CinS;
            const int gcol = (ix * p.CinS + c) * 2, ldst = col * p.CKp + cgi * 8;
            for (int r0 = sub ? 0 : 8; r0 < nrows; r0 += 8) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int iy = rowLo + r0 + j;
                    const bool ok = colOK && r0 + j < nrows && iy >= 0 && iy < p.H;
                    v[j] = bload16(xres, ok ? (n * p.H + iy) * (p.W * p.CinS * 2) + gcol : OOB);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (cu < rowUnits && r0 + j < nrows)
                        *reinterpret_cast<u32x4*>(sX + (r0 + j) * (p.RW * p.CKp) + ldst) = v[j];
            }
        }
        __syncthreads();
        if (ch == 0) OTP_STAMP(2);
        auto kstep = [&](int ks) {
            const int koff = sTab[ks * 4 + lg];
            bf16x8 a[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) a[m] = *reinterpret_cast<const bf16x8*>(sW + ((ks * 4 + lg) * BM + m * 16 + l15) * 8);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(sX + boff[nb] + koff);
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b, acc[m][nb], 0, 0, 0);
            }
        };
        if constexpr (K7) {               // 3x3 taps x 24 channels: fully unrolled so the LDS reads of step k+1 issue under step k
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) kstep(ks);
        } else {
            for (int ks = 0; ks < p.KS; ++ks) kstep(ks);
        }
        if (ch == 0) OTP_STAMP(3);
    }

    OTP_STAMP(4);
    conv_epilogue<MB, NB>(acc, valid, p, bias, res, out, out_f32, stats, smem, sRed, n, tile, mt, p0, npx);
    OTP_STAMP(5);
}

template <int MB, int NB, bool K7, bool PIPE>
int launch_conv_k(const ConvPlan& p, const void* x, const void* wpk, const void* bias, const void* res, void* out, void* stats,
                  hipStream_t st) {
    auto kern = nhwc_conv_kernel<MB, NB, K7, PIPE>;
    OTP_ALLOW_BIG_LDS(kern, p.lds);
    const int grid = p.N * p.tilesPerImg * p.nM;
    kern<<<grid, 256, p.lds, st>>>(static_cast<const bf16*>(x), static_cast<const bf16*>(wpk), static_cast<const float*>(bias),
                                   static_cast<const bf16*>(res), static_cast<bf16*>(out), static_cast<float*>(out),
                                   static_cast<float*>(stats), p);
    return otp_launch_status();
}

template <int MB, int NB>
int launch_conv(const ConvPlan& p, const void* x, const void* wpk, const void* bias, const void* res, void* out, void* stats,
                hipStream_t st) {
    if (p.KS == 7)
        return p.pipe ? launch_conv_k<MB, NB, true, true>(p, x, wpk, bias, res, out, stats, st)
                      : launch_conv_k<MB, NB, true, false>(p, x, wpk, bias, res, out, stats, st);
    return launch_conv_k<MB, NB, false, false>(p, x, wpk, bias, res, out, stats, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
// A workgroup owns (co block of up to 48) x (ci block of 16*cbw channels: 48 for 3x3 kernels, up to 192 for 1x1) x all taps
// and a contiguous range of pixel tiles of
// TPX output pixels (TPX / 32 k-steps per staged tile); wave w owns N-blocks {w, w+4, ...} of the (tap, ci16) list for all
// 3 co blocks.  The gy tile [TPX px][48 co] and the x window live in LDS as [pixel][channel]; a fragment is two
// ds_read_b64_tr_b16 (4 pixels x 16 channels each).
struct WgradPlan {
    int N, H, W, CinS, Ho, Wo, CoutS, Cin, Cout, kh, kw, stride, pad, dil;
    int nCo, nCi, splits, tilesPerImg, tilesTotal, tilesPerSplit, TPX;
    int RW, rowsMax, XC, ldsG, ldsX, lds;
    int tapmode;           // 1: the LDS image of x is [tap][tile pixel][XC] (each tap's shifted copy of the tile) instead of a window of rows
    int nbTot, cbw;        // N-blocks per workgroup = taps * cbw; cbw = 16-channel blocks of ci per workgroup (3 for 3x3 taps)
    int cg, spg;           // 1x1 kernels (nhwc_wgrad1x1_kernel): 48-channel co groups per workgroup, ci-block slots per wave and group; 0: not used
};

__device__ __forceinline__ bf16x4 tr_read(const bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)(const_cast<bf16*>(p)));
}

constexpr int WG_NBW = 7;      // N-blocks per wave (27 = 9 taps x 3 ci blocks over 4 waves)

constexpr int WG_GU = 3;       // gy units of 16 B per thread per tile (TPX * 6 / 256 at TPX = 128)
constexpr int WG_XU = 10;      // x window units per thread per tile that the prefetch path holds in registers

// PF = true: the loads of tile t+1 (gy tile + x window, at most WG_GU + WG_XU 16-byte units per thread) are issued right
// after tile t has been written to LDS and complete under its MFMAs; the unit -> (row, column, channel group)
// decomposition is computed once per thread.  PF = false: the general form (windows too large for the registers).
template <bool PF>
__global__ __launch_bounds__(256) void nhwc_wgrad_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gy,
                                                          float* __restrict__ part, WgradPlan p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* sG = reinterpret_cast<bf16*>(smem);                 // [TPX px][56] (48 channels + 8 pad: row stride 112 B)
    bf16* sX = reinterpret_cast<bf16*>(smem + p.ldsG);        // [rows][RW][XC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int split = bid % p.splits;
    bid /= p.splits;
    const int blk = bid;
    const int cib = bid % p.nCi, cob = bid / p.nCi;
    const int co0 = cob * 48, ci0 = cib * 16 * p.cbw;
    const int npx = p.Ho * p.Wo;
    const int TPX = p.TPX;

    // this wave's N-blocks: nb = wave + 4*i  ->  (tap, ci16 block)
    int xoff[WG_NBW];          // element offset inside the x window of (tap shift, ci block)
#pragma unroll
    for (int i = 0; i < WG_NBW; ++i) {
        const int nb = wave + 4 * i;
        const int nbc = nb < p.nbTot ? nb : 0;
        const int tap = nbc / p.cbw, cb = nbc - tap * p.cbw;
        const int dy = tap / p.kw, dx = tap - dy * p.kw;
        xoff[i] = (p.tapmode ? tap * p.TPX : (dy * p.dil) * p.RW + dx * p.dil) * p.XC + cb * 16;
    }
    f32x4 acc[3][WG_NBW];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < WG_NBW; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transpose-read addressing: within a 16-lane group, lane 4q+pp supplies row q (pixel), columns 4pp..4pp+3 (channels)
    const int q = l15 >> 2, pp = l15 & 3;
    const int xcu = p.XC / 8 - 1;                             // channel groups that carry data (the last 8 are zero padding)
    const int gunits = TPX * 6;
    const int rowUnits = p.RW * xcu;
    const otp_rsrc xres = make_rsrc(x, (size_t)p.N * p.H * p.W * p.CinS * 2);
    const otp_rsrc gres = make_rsrc(gy, (size_t)p.N * npx * p.CoutS * 2);

    // the padding channel group of every window pixel is zero for the whole kernel
    for (int i = tid; i < (p.tapmode ? p.kh * p.kw * p.TPX : p.rowsMax * p.RW); i += 256)
        *reinterpret_cast<u32x4*>(sX + (size_t)i * p.XC + xcu * 8) = u32x4{0u, 0u, 0u, 0u};

    struct Geom {
        int n, p0, p1, oy0, rowLo, nrows;
    };
    auto geom = [&](int t) {
        Geom g;
        g.n = t / p.tilesPerImg;
        const int tile = t - g.n * p.tilesPerImg;
        g.p0 = tile * TPX;
        g.p1 = min(g.p0 + TPX, npx);
        g.oy0 = g.p0 / p.Wo;
        const int oy1 = (g.p1 - 1) / p.Wo;
        g.rowLo = g.oy0 * p.stride - p.pad;
        g.nrows = (oy1 - g.oy0) * p.stride + (p.kh - 1) * p.dil + 1;
        return g;
    };

    // per-thread unit decomposition (tile independent)
    int gpx[WG_GU], gcg[WG_GU];
#pragma unroll
    for (int j = 0; j < WG_GU; ++j) {
        const int u = j * 256 + tid;
        gpx[j] = u / 6;
        gcg[j] = u - gpx[j] * 6;
    }
    int xu[PF ? WG_XU : 1];                                   // packed (row << 24 | column << 8 | channel group)
    if constexpr (PF) {
#pragma unroll
        for (int j = 0; j < WG_XU; ++j) {
            const int u = j * 256 + tid;
            const int r = u / rowUnits, ur = u - r * rowUnits;
            const int col = ur / xcu, cg = ur - col * xcu;
            xu[j] = (r << 24) | (col << 8) | cg;
        }
    }
    u32x4 gv[WG_GU], xv[PF ? WG_XU : 1];
    auto load_g = [&](const Geom& g) {
#pragma unroll
        for (int j = 0; j < WG_GU; ++j) {
            const bool ok = j * 256 + tid < gunits && g.p0 + gpx[j] < g.p1 && co0 + gcg[j] * 8 < p.CoutS;
            gv[j] = bload16(gres, ok ? ((g.n * npx + g.p0 + gpx[j]) * p.CoutS + co0 + gcg[j] * 8) * 2 : OOB);
        }
    };
    auto load_x = [&](const Geom& g) {
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < WG_XU; ++j) {
                const int r = xu[j] >> 24, col = (xu[j] >> 8) & 0xffff, cg = xu[j] & 255;
                const int iy = g.rowLo + r, ix = col - p.pad, c = ci0 + cg * 8;
                const bool ok = r < g.nrows && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.CinS;
                xv[j] = bload16(xres, ok ? (((g.n * p.H + iy) * p.W + ix) * p.CinS + c) * 2 : OOB);
            }
        }
    };

    OTP_STAMP(0);
    const int t0 = split * p.tilesPerSplit, t1 = min(t0 + p.tilesPerSplit, p.tilesTotal);
    Geom g = geom(t0);
    load_g(g);
    load_x(g);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();                                        // the previous tile's fragments have been read
        if (t == t0) OTP_STAMP(1);
#pragma unroll
        for (int j = 0; j < WG_GU; ++j)
            if (j * 256 + tid < gunits) *reinterpret_cast<u32x4*>(sG + gpx[j] * 56 + gcg[j] * 8) = gv[j];
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < WG_XU; ++j) {
                const int r = xu[j] >> 24, col = (xu[j] >> 8) & 0xffff, cg = xu[j] & 255;
                if (r < g.nrows) *reinterpret_cast<u32x4*>(sX + (r * p.RW + col) * p.XC + cg * 8) = xv[j];
            }
        } else {
            // general form: flat (row, column, channel group) units - or, in tap mode, (tap, tile pixel, channel group) units:
            // each tap's shifted copy of the tile (widely dilated or strided taps would otherwise drag many full-width rows
            // through LDS) - 8 loads in flight per thread
            if (p.tapmode) {
                const int tunits = p.kh * p.kw * TPX * xcu;
                for (int base = 0; base < tunits; base += 256 * 8) {
                    u32x4 v[8];
                    int dst[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int u = base + j * 256 + tid;
                        const int tp = u / xcu, cg = u - tp * xcu;
                        const int tap = tp / TPX, px = tp - tap * TPX;
                        const int dy = tap / p.kw, dx = tap - dy * p.kw;
                        const int pc = min(g.p0 + px, g.p1 - 1);
                        const int oy = pc / p.Wo, ox = pc - oy * p.Wo;
                        const int iy = oy * p.stride - p.pad + dy * p.dil, ix = ox * p.stride - p.pad + dx * p.dil, c = ci0 + cg * 8;
                        dst[j] = u < tunits ? tp * p.XC + cg * 8 : -1;
                        const bool ok = u < tunits && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.CinS;
                        v[j] = bload16(xres, ok ? (((g.n * p.H + iy) * p.W + ix) * p.CinS + c) * 2 : OOB);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (dst[j] >= 0) *reinterpret_cast<u32x4*>(sX + dst[j]) = v[j];
                }
            }
            const int xunits = p.tapmode ? 0 : g.nrows * rowUnits;
            for (int base = 0; base < xunits; base += 256 * 8) {
                u32x4 v[8];
                int dst[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int u = base + j * 256 + tid;
                    const int r = u / rowUnits, ur = u - r * rowUnits;
                    const int col = ur / xcu, cg = ur - col * xcu;
                    const int iy = g.rowLo + r, ix = col - p.pad, c = ci0 + cg * 8;
                    dst[j] = u < xunits ? (r * p.RW + col) * p.XC + cg * 8 : -1;
                    const bool ok = u < xunits && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.CinS;
                    v[j] = bload16(xres, ok ? (((g.n * p.H + iy) * p.W + ix) * p.CinS + c) * 2 : OOB);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (dst[j] >= 0) *reinterpret_cast<u32x4*>(sX + dst[j]) = v[j];
            }
        }
        __syncthreads();
        if (t == t0) OTP_STAMP(2);
        const Geom gc = g;
        if (t + 1 < t1) {                                       // next tile's loads fly under this tile's MFMAs
            g = geom(t + 1);
            load_g(g);
            load_x(g);
        }
        // k-steps of 32 pixels: lane group lg covers pixels 8*lg .. 8*lg+7 of the step (two transposed reads of 4 pixels)
        for (int k0 = 0; k0 < TPX; k0 += 32) {
            if (gc.p0 + k0 >= gc.p1) break;                      // uniform: the rest of the tile is past the image
            bf16x8 a[3];
            int xrow[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int px = k0 + 8 * lg + 4 * h + q;
                const int pc = min(gc.p0 + px, gc.p1 - 1);       // rows past the tile pair with zero gy rows
                const int oy = pc / p.Wo, ox = pc - oy * p.Wo;
                xrow[h] = (p.tapmode ? pc - gc.p0 : ((oy - gc.oy0) * p.stride) * p.RW + ox * p.stride) * p.XC + 4 * pp;
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const bf16x4 lo = tr_read(sG + (k0 + 8 * lg + q) * 56 + m * 16 + 4 * pp);
                const bf16x4 hi = tr_read(sG + (k0 + 8 * lg + 4 + q) * 56 + m * 16 + 4 * pp);
                a[m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < WG_NBW; ++i) {
                const bf16x4 lo = tr_read(sX + xrow[0] + xoff[i]);
                const bf16x4 hi = tr_read(sX + xrow[1] + xoff[i]);
                const bf16x8 b = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b, acc[m][i], 0, 0, 0);
            }
        }
        if (t == t0) OTP_STAMP(3);
    }
    OTP_STAMP(4);
    // partial sums in FRAGMENT order, one 16-byte store per lane and accumulator tile (1 KB per wave instruction):
    // part[split][block][wave][i][m][lane][4]; nhwc_wgrad_reduce_kernel maps them to (Cout, Cin, kh, kw)
    f32x4* dst = reinterpret_cast<f32x4*>(part) + ((size_t)(split * p.nCo * p.nCi + blk) * 4 + wave) * (WG_NBW * 3 * 64);
#pragma unroll
    for (int i = 0; i < WG_NBW; ++i)
#pragma unroll
        for (int m = 0; m < 3; ++m) dst[(i * 3 + m) * 64 + lane] = acc[m][i];
    OTP_STAMP(5);
}

// 1x1 / stride 1 kernels (the TransformerBlock MLP's projections on (N, 1, T, C) sequences, HRNet's bottleneck / fuse 1x1s).  The kernel
// above gives a 1x1 layer taps x cbw = 4 .. 12 of its 28 N-block slots and multiplies garbage in the rest (136 -> 544: 21 MFMAs per
// k-step and wave, 7 of them wanted), and re-stages the x tile once per 48 output channels.  Here the idle slots carry more OUTPUT
// channels: a workgroup owns CG groups of 48 co x all cbw ci blocks, slot i of a wave = (group i / SPG, ci block wave + 4 (i % SPG)) -
// (2, 3) for 9 - 12 ci blocks, (3, 2) for 5 - 8, (4, 1) up to 4 - one B fragment serves CG groups.  Tiles are TPX consecutive pixels of an
// image; both operands are [pixel][channel] LDS images read with the transposing ds_read_b64_tr_b16 as above (rows past the image
// hold zeros: their loads fall off the descriptor).  Partials leave in the same fragment order.
template <int CG, int SPG>
__global__ __launch_bounds__(256) void nhwc_wgrad1x1_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gy,
                                                             float* __restrict__ part, WgradPlan p) {
    constexpr int GC = 48 * CG + 8, NS = CG * SPG;
    static_assert(NS <= WG_NBW, "slots per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* sG = reinterpret_cast<bf16*>(smem);                 // [TPX px][GC]
    bf16* sX = reinterpret_cast<bf16*>(smem + p.ldsG);        // [TPX px][XC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int split = bid % p.splits;
    bid /= p.splits;
    const int blk = bid;
    const int cib = bid % p.nCi, cob = bid / p.nCi;
    const int co0 = cob * 48 * CG, ci0 = cib * 16 * p.cbw;
    const int npx = p.Ho * p.Wo, TPX = p.TPX;
    f32x4 acc[3][NS];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int i = 0; i < NS; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = l15 >> 2, pp = l15 & 3;
    const int xcu = p.XC / 8 - 1, gcu = 6 * CG;
    const int gunits = TPX * gcu, xunits = TPX * xcu;
    const otp_rsrc xres = make_rsrc(x, (size_t)p.N * npx * p.CinS * 2);
    const otp_rsrc gres = make_rsrc(gy, (size_t)p.N * npx * p.CoutS * 2);
    for (int i = tid; i < TPX; i += 256) {                     // padding channel groups: zero for the whole kernel
        *reinterpret_cast<u32x4*>(sX + (size_t)i * p.XC + xcu * 8) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(sG + (size_t)i * GC + 48 * CG) = u32x4{0u, 0u, 0u, 0u};
    }
    const int t0 = split * p.tilesPerSplit, t1 = min(t0 + p.tilesPerSplit, p.tilesTotal);
    for (int t = t0; t < t1; ++t) {
        const int n = t / p.tilesPerImg, p0 = (t - n * p.tilesPerImg) * TPX, p1 = min(p0 + TPX, npx);
        __syncthreads();                                        // the previous tile's fragments have been read
        for (int base = 0; base < gunits; base += 256 * 8) {
            u32x4 v[8];
            int dst[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u = base + j * 256 + tid;
                const int px = u / gcu, cgi = u - px * gcu;
                dst[j] = u < gunits ? px * GC + cgi * 8 : -1;
                const bool ok = u < gunits && p0 + px < p1 && co0 + cgi * 8 < p.CoutS;
                v[j] = bload16(gres, ok ? ((n * npx + p0 + px) * p.CoutS + co0 + cgi * 8) * 2 : OOB);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (dst[j] >= 0) *reinterpret_cast<u32x4*>(sG + dst[j]) = v[j];
        }
        for (int base = 0; base < xunits; base += 256 * 8) {
            u32x4 v[8];
            int dst[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u = base + j * 256 + tid;
                const int px = u / xcu, cgi = u - px * xcu;
                dst[j] = u < xunits ? px * p.XC + cgi * 8 : -1;
                const bool ok = u < xunits && p0 + px < p1 && ci0 + cgi * 8 < p.CinS;
                v[j] = bload16(xres, ok ? ((n * npx + p0 + px) * p.CinS + ci0 + cgi * 8) * 2 : OOB);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (dst[j] >= 0) *reinterpret_cast<u32x4*>(sX + dst[j]) = v[j];
        }
        __syncthreads();
        for (int k0 = 0; k0 < TPX; k0 += 32) {
            if (p0 + k0 >= p1) break;                           // uniform: the rest of the tile is past the image
            const int r0 = k0 + 8 * lg + q;                     // the lane's pixel row of the two transposed reads (r0, r0 + 4)
            bf16x8 a[CG][3];
#pragma unroll
            for (int g = 0; g < CG; ++g)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const bf16x4 lo = tr_read(sG + r0 * GC + g * 48 + m * 16 + 4 * pp);
                    const bf16x4 hi = tr_read(sG + (r0 + 4) * GC + g * 48 + m * 16 + 4 * pp);
                    a[g][m] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
            for (int j = 0; j < SPG; ++j) {
                const int xo = (wave + 4 * j) * 16 + 4 * pp;    // (a block past the layer's channels: finite neighbours, discarded)
                const bf16x4 lo = tr_read(sX + r0 * p.XC + xo);
                const bf16x4 hi = tr_read(sX + (r0 + 4) * p.XC + xo);
                const bf16x8 b = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int g = 0; g < CG; ++g)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        acc[m][g * SPG + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][m], b, acc[m][g * SPG + j], 0, 0, 0);
            }
        }
    }
    f32x4* dst = reinterpret_cast<f32x4*>(part) + ((size_t)(split * p.nCo * p.nCi + blk) * 4 + wave) * (WG_NBW * 3 * 64);
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int m = 0; m < 3; ++m) dst[(i * 3 + m) * 64 + lane] = acc[m][i];
}

// gw[co][ci][tap] = sum over splits of the fragment-ordered partials.  Threads walk the FRAGMENT order (coalesced reads of
// the splits x fragments slab, 4 split lanes per fragment element, 8 loads in flight) and scatter the few results.
__global__ __launch_bounds__(256) void nhwc_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, WgradPlan p,
                                                                 size_t frag_total) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t f = blockIdx.x * (size_t)64 + e;
    float s = 0.f;
    if (f < frag_total) {
        int k = sl;
        for (; k + 28 < p.splits; k += 32) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(k + 4 * j) * frag_total + f];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k < p.splits; k += 4) s += part[(size_t)k * frag_total + f];
    }
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && f < frag_total) {
        // f = (((blk*4 + wave)*WG_NBW + i)*3 + m)*256 + lane*4 + r
        const int r = (int)(f & 3), lane = (int)((f >> 2) & 63);
        size_t q = f >> 8;
        const int m = (int)(q % 3); q /= 3;
        const int i = (int)(q % WG_NBW); q /= WG_NBW;
        const int wave = (int)(q & 3);
        const int blk = (int)(q >> 2);
        const int cib = blk % p.nCi, cob = blk / p.nCi;
        const int nb = wave + 4 * i;
        if (p.cg > 0) {                                          // nhwc_wgrad1x1_kernel: slot i = (co group i / spg, ci block wave + 4 (i % spg))
            const int g = i / p.spg, cb = wave + 4 * (i - g * p.spg);
            const int co = (cob * p.cg + g) * 48 + m * 16 + (lane >> 4) * 4 + r, ci = cib * 16 * p.cbw + cb * 16 + (lane & 15);
            if (i < p.cg * p.spg && cb < p.cbw && co < p.Cout && ci < p.Cin)
                gw[(size_t)co * p.Cin + ci] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        } else if (nb < p.nbTot) {
            const int tap = nb / p.cbw, cb = nb - tap * p.cbw;
            const int co = cob * 48 + m * 16 + (lane >> 4) * 4 + r, ci = cib * 16 * p.cbw + cb * 16 + (lane & 15);
            if (co < p.Cout && ci < p.Cin)
                gw[((size_t)co * p.Cin + ci) * (p.kh * p.kw) + tap] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        }
    }
}

bool make_wgrad_plan(const otp_nhwc_conv_desc* d, WgradPlan* p) {
    ConvPlan c;
    if (!make_plan(d, &c)) return false;
    p->N = c.N, p->H = c.H, p->W = c.W, p->CinS = c.CinS, p->Ho = c.Ho, p->Wo = c.Wo, p->CoutS = c.CoutS;
    p->Cin = c.Cin, p->Cout = c.Cout, p->kh = c.kh, p->kw = c.kw, p->stride = c.stride, p->pad = c.pad, p->dil = c.dil;
    const int taps = c.kh * c.kw;
    if (taps * 1 > 4 * WG_NBW) return false;                   // up to 28 (tap, 16-channel) blocks per workgroup
    // ci blocks of 16 channels per workgroup: fill the 28 N-block slots (3 for 3x3 kernels, up to 12 = 192 channels for 1x1)
    int cbw = (4 * WG_NBW) / taps;
    if (cbw > 12) cbw = 12;
    if (cbw > (c.CinS + 15) / 16) cbw = (c.CinS + 15) / 16;
    p->cbw = cbw;
    p->nCo = (c.Cout + 47) / 48, p->nCi = (c.Cin + 16 * cbw - 1) / (16 * cbw);
    p->nbTot = taps * cbw;
    p->cg = p->spg = 0;
    {
        const char* e = getenv("OTPOSE_WGRAD1X1");                 // (=0: 1x1 layers on the general kernel; A/B, tests)
        if (taps == 1 && c.stride == 1 && c.pad == 0 && !(e && atoi(e) == 0)) {
            p->spg = (cbw + 3) / 4;                                  // 1 .. 3
            const int cgmax = p->spg == 1 ? 4 : (p->spg == 2 ? 3 : 2);
            p->cg = p->nCo < cgmax ? p->nCo : cgmax;
            if (p->cg < 2) p->cg = p->spg = 0;                       // <= 48 output channels: nothing to add to the general kernel
            else p->nCo = (c.Cout + 48 * p->cg - 1) / (48 * p->cg);
        }
    }
    // a pointwise conv sees the image as rows of 128 / 64 / 32 pixels (when that divides it): a tile's window is then the
    // tile itself instead of the full-width image rows it touches
    if (taps == 1 && c.stride == 1 && c.pad == 0) {
        const int npx1 = c.H * c.W;
        for (int w1 = 128; w1 >= 32; w1 >>= 1)
            if (npx1 % w1 == 0) {
                p->W = p->Wo = c.W = c.Wo = w1;
                p->H = p->Ho = c.H = c.Ho = npx1 / w1;
                break;
            }
    }
    const int npx = c.Ho * c.Wo;
    const int need = (c.Wo - 1) * c.stride + (c.kw - 1) * c.dil + 1;
    p->RW = c.W + 2 * c.pad > need ? c.W + 2 * c.pad : need;
    // window pixel stride: the ci block's channels + 8 zero channels (odd multiple of 16 bytes).  N-blocks past the real
    // channels read neighbouring pixels (finite, discarded): 128 bytes of slack.
    p->XC = 8 * (c.CinS / 8 < 2 * cbw ? c.CinS / 8 : 2 * cbw) + 8;
    // pixels per staged tile: the largest of 128 / 64 / 32 whose LDS image leaves room for two workgroups per CU, else
    // the largest that fits at all (full-width input rows make the window of wide strided layers large)
    bool found = false;
    p->tapmode = 0;
    for (int pass = 0; pass < 2 && !found; ++pass)
        for (int tpx = 128; tpx >= 32 && !found; tpx >>= 1) {
            // output rows a tile can touch (tiles start at multiples of tpx: aligned tiles never straddle a row)
            int rowsOut = c.Wo % tpx == 0 ? 1 : (tpx % c.Wo == 0 ? tpx / c.Wo : (tpx - 1 + c.Wo - 1) / c.Wo + 1);
            if (rowsOut > c.Ho) rowsOut = c.Ho;
            const int rows = (rowsOut - 1) * c.stride + (c.kh - 1) * c.dil + 1;
            const int ldsG = tpx * (p->cg > 0 ? 48 * p->cg + 8 : 56) * 2;
            const int winX = round_up(rows * p->RW * p->XC * 2, 16) + 128, tapX = taps * tpx * p->XC * 2 + 128;
            const int limit = pass == 0 ? 80 * 1024 : OTP_LDS_LIMIT;
            // the window of rows unless each tap's copy of the tile is smaller (wide dilations, strided wide maps)
            const bool tapm = taps > 1 && tapX < winX;
            const int ldsX = p->cg > 0 ? tpx * p->XC * 2 + 128 : (tapm ? tapX : winX);      // (1x1 kernel: the tile itself)
            if (ldsG + ldsX <= limit) {
                p->TPX = tpx, p->rowsMax = rows, p->ldsG = ldsG, p->ldsX = ldsX, p->lds = ldsG + ldsX, p->tapmode = tapm;
                found = true;
            }
        }
    if (!found) return false;
    p->tilesPerImg = (npx + p->TPX - 1) / p->TPX;
    p->tilesTotal = c.N * p->tilesPerImg;
    // workgroups aimed at = the 512 that are resident at once (two per CU: ~55 KB of LDS, ~200 registers): one full round instead of
    // 1.5 (768 until round 5) and a third fewer partial sums to write and fold - 96 -> 96 @48x36: 71.8 -> 61.3 us, 192 -> 192 @24x18:
    // 72.8 -> 57.3, 64 -> 64 @96x72: 182 -> 155, 64 -> 256 1x1: 129 -> 102 (OTPOSE_WGRAD_SPLITS: tuning override)
    static const int budget = getenv("OTPOSE_WGRAD_SPLITS") ? atoi(getenv("OTPOSE_WGRAD_SPLITS")) : 512;
    int splits = budget / (p->nCo * p->nCi);
    if (splits < 1) splits = 1;
    if (splits > p->tilesTotal) splits = p->tilesTotal;
    if (splits > 512) splits = 512;
    p->tilesPerSplit = (p->tilesTotal + splits - 1) / splits;
    p->splits = (p->tilesTotal + p->tilesPerSplit - 1) / p->tilesPerSplit;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm (batch statistics) on NHWC bf16, fp32 arithmetic
// ---------------------------------------------------------------------------------------------------------------------
// partial sums [rows][2][C] -> mean / rstd, scale = gamma*rstd, shift = beta - mean*scale, running statistics update
constexpr int BNF_RL = 128;                 // row lanes of the two finalize kernels (8 channels x BNF_RL rows = 1024 threads)
__global__ __launch_bounds__(8 * BNF_RL) void bn_finalize_kernel(const float* __restrict__ part, int rows, int C, int Ctrue, float count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                           float* __restrict__ scale_o, float* __restrict__ shift_o,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var, float eps,
                                                           float momentum) {
    // 8 channels x 128 row lanes per workgroup: the partial rows (up to N * tiles of the conv) are the long axis
    __shared__ double red[2][BNF_RL][8];
    const int cl = threadIdx.x & 7, rq = threadIdx.x >> 3, c = blockIdx.x * 8 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        int r = rq;
        for (; r + 3 * BNF_RL < rows; r += 4 * BNF_RL) {
            float a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = part[((size_t)(r + BNF_RL * j) * 2) * C + c];
                b[j] = part[((size_t)(r + BNF_RL * j) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s1 += (double)a[j], s2 += (double)b[j];
        }
        for (; r < rows; r += BNF_RL) {
            s1 += (double)part[((size_t)r * 2) * C + c];
            s2 += (double)part[((size_t)r * 2 + 1) * C + c];
        }
    }
    red[0][rq][cl] = s1, red[1][rq][cl] = s2;
    __syncthreads();
    if (rq < 16) {                                  // fixed-order fold: 128 -> 16 row lanes, then one thread per channel
        for (int k = rq + 16; k < BNF_RL; k += 16) s1 += red[0][k][cl], s2 += red[1][k][cl];
        red[0][rq][cl] = s1, red[1][rq][cl] = s2;
    }
    __syncthreads();
    if (rq == 0 && c < C) {
        for (int k = 1; k < 16; ++k) s1 += red[0][k][cl], s2 += red[1][k][cl];
        const double m = s1 / count;
        double var = s2 / count - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const bool live = c < Ctrue;
        const float g = live ? gamma[c] : 0.f, b = live ? beta[c] : 0.f;
        mean_o[c] = (float)m, rstd_o[c] = rstd;
        scale_o[c] = g * rstd;
        shift_o[c] = b - (float)m * g * rstd;
        if (live && run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
            const double unb = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
        }
    }
}

// y = act(x*scale[c] + shift[c] (+ res)); 8 channels per thread
template <bool HAS_RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const bf16* __restrict__ res,
                                                        bf16* __restrict__ y, unsigned char* __restrict__ mask, size_t units,
                                                        int C8, int relu) {
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(u % C8) * 8;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + u * 8);
        bf16x8 r8;
        if constexpr (HAS_RES) r8 = *reinterpret_cast<const bf16x8*>(res + u * 8);
        const f32x4 sa = *reinterpret_cast<const f32x4*>(scale + c), sb = *reinterpret_cast<const f32x4*>(scale + c + 4);
        const f32x4 ha = *reinterpret_cast<const f32x4*>(shift + c), hb = *reinterpret_cast<const f32x4*>(shift + c + 4);
        bf16x8 o;
        unsigned bits = 0;                                     // ReLU mask of the 8 channels (1 bit each): what backward needs of y
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf2f(v[j]) * (j < 4 ? sa[j] : sb[j - 4]) + (j < 4 ? ha[j] : hb[j - 4]);
            if constexpr (HAS_RES) f += bf2f(r8[j]);
            if (relu) {
                bits |= (f > 0.f ? 1u : 0u) << j;
                f = fmaxf(f, 0.f);
            }
            o[j] = (bf16)f;
        }
        *reinterpret_cast<bf16x8*>(y + u * 8) = o;
        if (mask) mask[u] = (unsigned char)bits;
    }
}

// backward pass 1: per-workgroup partial sums of g and g*xhat over a pixel range, g = gy * (y > 0 when relu)
// threads: (pixel lane pl, channel group cg); partial rows [wg][2][C]
template <int RELU>    // 0: no activation, 1: y tensor, 2: bit mask (compile time: a runtime branch around a load makes hipcc drain every outstanding load)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16* __restrict__ gy, const bf16* __restrict__ y,
                                                             const bf16* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* __restrict__ part,
                                                             size_t npix, int C8, int pixPerWg, int relu) {
    extern __shared__ float sred[];                       // [ppw][C8*16]
    const int ppw = 256 / C8;
    const int pl = threadIdx.x / C8, cg = threadIdx.x - pl * C8;
    const int C = C8 * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    if (pl < ppw) {
        float mu[8], rs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mu[j] = mean[cg * 8 + j], rs[j] = rstd[cg * 8 + j];
        const size_t pbeg = (size_t)blockIdx.x * pixPerWg;
        const size_t pend = pbeg + pixPerWg < npix ? pbeg + pixPerWg : npix;
        for (size_t px = pbeg + pl; px < pend; px += ppw) {
            const size_t o = px * C + cg * 8;
            const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gy + o);
            const bf16x8 x8 = *reinterpret_cast<const bf16x8*>(x + o);
            bf16x8 y8;
            unsigned bits = 0xffu;
            if constexpr (RELU == 1) y8 = *reinterpret_cast<const bf16x8*>(y + o);
            if constexpr (RELU == 2) bits = reinterpret_cast<const unsigned char*>(y)[px * C8 + cg];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = bf2f(g8[j]);
                if constexpr (RELU == 1) { if (!(bf2f(y8[j]) > 0.f)) g = 0.f; }
                if constexpr (RELU == 2) { if (!((bits >> j) & 1u)) g = 0.f; }
                s1[j] += g;
                s2[j] += g * (bf2f(x8[j]) - mu[j]) * rs[j];
            }
        }
    }
    if (pl < ppw) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sred[(pl * C8 + cg) * 16 + j] = s1[j];
            sred[(pl * C8 + cg) * 16 + 8 + j] = s2[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C8 * 16; i += 256) {
        float v = 0.f;
        for (int k = 0; k < ppw; ++k) v += sred[k * C8 * 16 + i];
        const int cgi = i / 16, j = i & 15;
        part[((size_t)blockIdx.x * 2 + (j >> 3)) * C + cgi * 8 + (j & 7)] = v;
    }
}

// partials -> dgamma, dbeta, and the three per-channel coefficients of pass 2
__global__ __launch_bounds__(8 * BNF_RL) void bn_bwd_finalize_kernel(const float* __restrict__ part, int rows, int C, int Ctrue,
                                                                      float count, const float* __restrict__ gamma,
                                                                      const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta, float* __restrict__ coef) {
    // 8 channels x 128 row lanes per workgroup: the partial rows (up to N * tiles of the conv, 2160 at 96x72 x 80) are the long
    // axis - with 32 row lanes this launch, between the two HBM passes of every layer's BatchNorm backward, was a 15 us chain
    __shared__ double red[2][BNF_RL][8];
    const int cl = threadIdx.x & 7, rq = threadIdx.x >> 3, c = blockIdx.x * 8 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        int r = rq;
        for (; r + 3 * BNF_RL < rows; r += 4 * BNF_RL) {
            float a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = part[((size_t)(r + BNF_RL * j) * 2) * C + c];
                b[j] = part[((size_t)(r + BNF_RL * j) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s1 += (double)a[j], s2 += (double)b[j];
        }
        for (; r < rows; r += BNF_RL) {
            s1 += (double)part[((size_t)r * 2) * C + c];
            s2 += (double)part[((size_t)r * 2 + 1) * C + c];
        }
    }
    red[0][rq][cl] = s1, red[1][rq][cl] = s2;
    __syncthreads();
    // fold the row lanes in a fixed order: 128 -> 16 on the first 16 row lanes, then one thread per channel
    if (rq < 16) {
        for (int k = rq + 16; k < BNF_RL; k += 16) s1 += red[0][k][cl], s2 += red[1][k][cl];
        red[0][rq][cl] = s1, red[1][rq][cl] = s2;
    }
    __syncthreads();
    if (rq == 0 && c < C) {
        for (int k = 1; k < 16; ++k) s1 += red[0][k][cl], s2 += red[1][k][cl];
        const bool live = c < Ctrue;
        if (live) dbeta[c] = (float)s1, dgamma[c] = (float)s2;
        const float k1 = live ? gamma[c] * rstd[c] : 0.f;
        coef[c] = k1;                                     // gx = k1 * (g - mg - xhat * mgx)
        coef[C + c] = (float)(s1 / count);
        coef[2 * C + c] = (float)(s2 / count);
    }
}

template <int RELU>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16* __restrict__ gy, const bf16* __restrict__ y,
                                                            const bf16* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ coef,
                                                            bf16* __restrict__ gx, bf16* __restrict__ gres, size_t units, int C8,
                                                            int relu) {
    const int C = C8 * 8;
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(u % C8) * 8;
        const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gy + u * 8);
        const bf16x8 x8 = *reinterpret_cast<const bf16x8*>(x + u * 8);
        bf16x8 y8;
        unsigned bits = 0xffu;
        if constexpr (RELU == 1) y8 = *reinterpret_cast<const bf16x8*>(y + u * 8);
        if constexpr (RELU == 2) bits = reinterpret_cast<const unsigned char*>(y)[u];
        bf16x8 o, gr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float g = bf2f(g8[j]);
            if constexpr (RELU == 1) { if (!(bf2f(y8[j]) > 0.f)) g = 0.f; }
            if constexpr (RELU == 2) { if (!((bits >> j) & 1u)) g = 0.f; }
            const float xh = (bf2f(x8[j]) - mean[c + j]) * rstd[c + j];
            o[j] = (bf16)(coef[c + j] * (g - coef[C + c + j] - xh * coef[2 * C + c + j]));
            gr[j] = (bf16)g;
        }
        *reinterpret_cast<bf16x8*>(gx + u * 8) = o;
        if (gres) *reinterpret_cast<bf16x8*>(gres + u * 8) = gr;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// nearest-upsample-accumulate (HRNet fuse rows, model/HRNet.py:426-439, 488-494) and layout converters
// ---------------------------------------------------------------------------------------------------------------------
// out[n, y, x, c] = act(res[n, y, x, c] + low[n, y/f, x/f, c])
__global__ __launch_bounds__(256) void upsample_add_nhwc_kernel(const bf16* __restrict__ low, const bf16* __restrict__ res,
                                                                 bf16* __restrict__ out, int N, int H, int W, int C8, int f,
                                                                 int relu) {
    const size_t units = (size_t)N * H * W * C8;
    const int Hl = H / f, Wl = W / f;
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        size_t r = u;
        const int cg = r % C8; r /= C8;
        const int xx = r % W; r /= W;
        const int yy = r % H;
        const int n = (int)(r / H);
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(res + u * 8);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(low + ((((size_t)n * Hl + yy / f) * Wl + xx / f) * C8 + cg) * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = bf2f(a[j]) + bf2f(b[j]);
            if (relu) v = fmaxf(v, 0.f);
            o[j] = (bf16)v;
        }
        *reinterpret_cast<bf16x8*>(out + u * 8) = o;
    }
}

// gres = gy * mask, glow[n, yl, xl, c] = sum over the f x f block of gy * mask   (mask = out > 0 when relu)
__global__ __launch_bounds__(256) void upsample_add_nhwc_bwd_kernel(const bf16* __restrict__ gy, const bf16* __restrict__ out,
                                                                     bf16* __restrict__ gres, bf16* __restrict__ glow, int N,
                                                                     int Hl, int Wl, int C8, int f, int relu) {
    const size_t units = (size_t)N * Hl * Wl * C8;
    const int H = Hl * f, W = Wl * f;
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        size_t r = u;
        const int cg = r % C8; r /= C8;
        const int xl = r % Wl; r /= Wl;
        const int yl = r % Hl;
        const int n = (int)(r / Hl);
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) {
                const size_t o = ((((size_t)n * H + yl * f + dy) * W + xl * f + dx) * C8 + cg) * 8;
                const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gy + o);
                bf16x8 y8, gr;
                if (relu) y8 = *reinterpret_cast<const bf16x8*>(out + o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float g = bf2f(g8[j]);
                    if (relu && !(bf2f(y8[j]) > 0.f)) g = 0.f;
                    s[j] += g;
                    gr[j] = (bf16)g;
                }
                *reinterpret_cast<bf16x8*>(gres + o) = gr;
            }
        bf16x8 o8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = (bf16)s[j];
        *reinterpret_cast<bf16x8*>(glow + u * 8) = o8;
    }
}

// (N, C, H, W) fp32 -> (N, H, W, CS) bf16, padding channels zero.  frame_split = B > 0 reads the (B, 5*C, H, W) clip tensor
// as (5B, C, H, W) with image n = f*B + b from channels [f*C, (f+1)*C) of sample b (model/OTPose.py:317).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, bf16* __restrict__ out, int N, int C,
                                                            int HW, int CS, int frame_split) {
    const size_t total = (size_t)N * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW), px = (int)(i - (size_t)n * HW);
        const float* src;
        if (frame_split > 0) {
            const int fr = n / frame_split, b = n - fr * frame_split;
            src = in + (((size_t)b * (N / frame_split) + fr) * C) * HW + px;
        } else {
            src = in + (size_t)n * C * HW + px;
        }
        for (int c0 = 0; c0 < CS; c0 += 8) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)(c0 + j < C ? src[(size_t)(c0 + j) * HW] : 0.f);
            *reinterpret_cast<bf16x8*>(out + i * CS + c0) = o;
        }
    }
}

// The same conversion for tensors with many channels (the (B, C, T) sequences of the temporal encoders: C = 136): a workgroup moves
// 64 pixels x all channels through the LDS - reads are 256-byte runs of one channel plane, writes 16-byte units of the tile's
// CONTIGUOUS 64 x CS x 2 bytes (the kernel above scatters a 16-byte piece per thread at the pixel stride: 46 us for 16 x 136 x 6912
// against 16 us for the opposite direction).  LDS rows of CS / 2 | 1 dwords (channel pairs): an odd stride, no bank conflicts.
__global__ __launch_bounds__(256) void nchw_to_nhwc_tile_kernel(const float* __restrict__ in, bf16* __restrict__ out, int C, int HW,
                                                                 int CS) {
    extern __shared__ unsigned int ttile[];
    const int tilesPerImg = (HW + 63) / 64;
    const int n = blockIdx.x / tilesPerImg, p0 = (blockIdx.x - n * tilesPerImg) * 64;
    const int RS = (CS / 2) | 1;
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
    const bool pv = p0 + px < HW;
    const float* src = in + (size_t)n * C * HW + p0 + (pv ? px : 0);
#pragma unroll 4
    for (int cp = g; cp < CS / 2; cp += 4) {
        const int c = 2 * cp;
        const float a = (pv && c < C) ? src[(size_t)c * HW] : 0.f, b = (pv && c + 1 < C) ? src[(size_t)(c + 1) * HW] : 0.f;
        typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        ttile[px * RS + cp] = __builtin_bit_cast(unsigned int, (bf16x2_t){(bf16)a, (bf16)b});
    }
    __syncthreads();
    const int U = CS / 8, npx = HW - p0 < 64 ? HW - p0 : 64;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    for (int u = threadIdx.x; u < npx * U; u += 256) {
        const int q = u / U, k = u - q * U;
        const unsigned int* r = ttile + q * RS + 4 * k;
        *reinterpret_cast<u32x4_t*>(out + ((size_t)n * HW + p0 + q) * CS + 8 * k) = (u32x4_t){r[0], r[1], r[2], r[3]};
    }
}

// (N, H, W, CS) bf16 -> (N, C, H, W) fp32: a workgroup moves 64 pixels x all channels through LDS, so that the reads are
// 16-byte units of whole pixel rows and the writes 256-byte runs of one channel plane
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const bf16* __restrict__ in, float* __restrict__ out, int N, int C,
                                                            int HW, int CS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* sT = reinterpret_cast<bf16*>(smem);                 // [64][CS + 8]
    const int CP = CS + 8, C8 = CS / 8;
    const int tilesPerImg = (HW + 63) / 64;
    const int n = blockIdx.x / tilesPerImg, p0 = (blockIdx.x - n * tilesPerImg) * 64;
    const int npx = min(64, HW - p0);
    for (int u = threadIdx.x; u < 64 * C8; u += 256) {
        const int px = u / C8, cg = u - px * C8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (px < npx) v = *reinterpret_cast<const u32x4*>(in + ((size_t)n * HW + p0 + px) * CS + cg * 8);
        *reinterpret_cast<u32x4*>(sT + px * CP + cg * 8) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < npx)
        for (int c = wave; c < C; c += 4) out[((size_t)n * C + c) * HW + p0 + lane] = bf2f(sT[lane * CP + c]);
}

// (N, C, H, W) fp32 gradient -> (N, H, W, CS) bf16 (same as nchw_to_nhwc without frame_split) is reused for grads.

// zero-insertion for the input gradient of a strided conv: out (N, H, W, C) = 0 except out[:, y*s, x*s] = in[:, y, x]
__global__ __launch_bounds__(256) void dilate_nhwc_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int N, int Hi,
                                                           int Wi, int s, int H, int W, int C8) {
    const size_t units = (size_t)N * H * W * C8;
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        size_t r = u;
        const int cg = r % C8; r /= C8;
        const int xx = r % W; r /= W;
        const int yy = r % H;
        const int n = (int)(r / H);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (yy % s == 0 && xx % s == 0 && yy / s < Hi && xx / s < Wi)
            v = *reinterpret_cast<const u32x4*>(in + ((((size_t)n * Hi + yy / s) * Wi + xx / s) * C8 + cg) * 8);
        *reinterpret_cast<u32x4*>(out + u * 8) = v;
    }
}

// per-channel sums over the pixels of an NHWC bf16 tensor (bias gradients): partial rows [wg][C] -> out[c]
__global__ __launch_bounds__(256) void channel_sum_nhwc_kernel(const bf16* __restrict__ g, float* __restrict__ part, size_t npix,
                                                                int C8, int pixPerWg) {
    extern __shared__ float sred[];                       // [ppw][C8*8]
    const int ppw = 256 / C8;
    const int pl = threadIdx.x / C8, cg = threadIdx.x - pl * C8;
    const int C = C8 * 8;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (pl < ppw) {
        const size_t pbeg = (size_t)blockIdx.x * pixPerWg;
        const size_t pend = pbeg + pixPerWg < npix ? pbeg + pixPerWg : npix;
        for (size_t px = pbeg + pl; px < pend; px += ppw) {
            const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(g + px * C + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += bf2f(g8[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sred[(pl * C8 + cg) * 8 + j] = s[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
        float v = 0.f;
        for (int k = 0; k < ppw; ++k) v += sred[k * C + i];
        part[(size_t)blockIdx.x * C + i] = v;
    }
}

__global__ __launch_bounds__(256) void channel_sum_nhwc_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                       int rows, int C, int Ctrue) {
    __shared__ double red[32][8];
    const int cl = threadIdx.x & 7, rq = threadIdx.x >> 3, c = blockIdx.x * 8 + cl;
    double s = 0.0;
    if (c < C)
        for (int r = rq; r < rows; r += 32) s += (double)part[(size_t)r * C + c];
    red[rq][cl] = s;
    __syncthreads();
    if (rq == 0 && c < Ctrue) {
        for (int k = 1; k < 32; ++k) s += red[k][cl];
        out[c] = (float)s;
    }
}

// exact-erf GELU on bf16 (nn.GELU of the TransformerBlock MLP, model/blocks.py:248-254), fp32 arithmetic
__global__ __launch_bounds__(256) void gelu_bf16_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t units) {
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + u * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            o[j] = (bf16)(0.5f * f * (1.f + erff(f * 0.70710678118654752440f)));
        }
        *reinterpret_cast<bf16x8*>(y + u * 8) = o;
    }
}
__global__ __launch_bounds__(256) void gelu_bf16_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                             bf16* __restrict__ dx, size_t units) {
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + u * 8), g = *reinterpret_cast<const bf16x8*>(dy + u * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            const float cdf = 0.5f * (1.f + erff(f * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * f * f);
            o[j] = (bf16)(bf2f(g[j]) * (cdf + f * pdf));
        }
        *reinterpret_cast<bf16x8*>(dx + u * 8) = o;
    }
}

// GELU followed by Dropout(p) (model/blocks.py:250-251) as one pass: y = keep ? gelu(x) / (1 - p) : 0, one rounding, plus the keep
// decisions as one bit per element (a byte per 8-element unit) for the backward - the separate launches cost a 240 MB pass forward
// (fused_dropout) and a 360 MB one backward (masked_scale) per MLP.  Draws: a counter-based hash of (element pair, seed), two 16-bit
// uniforms per 32-bit hash (csrc/hb.h: otp_drop_keep8), keep <=> u16 >= round(65536 p); the seed comes from PyTorch's CUDA generator
// (bf16_ops.gelu_dropout), so torch.manual_seed reproduces a step.
__global__ __launch_bounds__(256) void gelu_dropout_bf16_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                     unsigned char* __restrict__ keep, size_t units, uint32_t s0, uint32_t s1,
                                                                     uint32_t thr, float scale) {
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + u * 8);
        const unsigned bits = otp_drop_keep8(u, s0, s1, thr);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            const float g = 0.5f * f * (1.f + erff(f * 0.70710678118654752440f)) * scale;
            o[j] = (bf16)(((bits >> j) & 1u) ? g : 0.f);
        }
        *reinterpret_cast<bf16x8*>(y + u * 8) = o;
        keep[u] = (unsigned char)bits;
    }
}
__global__ __launch_bounds__(256) void gelu_dropout_bf16_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                     const unsigned char* __restrict__ keep, bf16* __restrict__ dx,
                                                                     size_t units, float scale) {
    for (size_t u = blockIdx.x * (size_t)blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + u * 8), g = *reinterpret_cast<const bf16x8*>(dy + u * 8);
        const unsigned bits = keep[u];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            const float cdf = 0.5f * (1.f + erff(f * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * f * f);
            o[j] = (bf16)(((bits >> j) & 1u) ? bf2f(g[j]) * scale * (cdf + f * pdf) : 0.f);
        }
        *reinterpret_cast<bf16x8*>(dx + u * 8) = o;
    }
}

int grid_for(size_t units) {
    size_t g = (units + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================
namespace {
size_t hb_weight_bytes(const otp_nhwc_conv_desc* d) {
    const int ntw = otp_hb_ntw(d->Cout), nN = ((d->Cout + 15) / 16 + ntw - 1) / ntw;
    return (size_t)nN * (d->Cin / 16) * otp_hb_wb(ntw);
}
size_t hbpw_weight_bytes(const otp_nhwc_conv_desc* d) { return (size_t)((d->Cout + 31) / 32) * otp_hbpw_blkb(otp_hbpw_ks(d->Cin)); }
}  // namespace

extern "C" size_t otp_nhwc_conv_weight_bytes(const otp_nhwc_conv_desc* d) {
    ConvPlan p;
    if (use_hb(d)) return hb_weight_bytes(d);
    if (use_hbpw(d)) return hbpw_weight_bytes(d);
    return make_plan(d, &p) ? p.wbytes : 0;
}

extern "C" int otp_nhwc_conv_stats_rows(const otp_nhwc_conv_desc* d) {
    ConvPlan p;
    if (use_hb(d)) return otp_hb_stats_rows(d);
    if (use_hbpw(d)) return otp_hbpw_stats_rows(d);
    return make_plan(d, &p) ? p.N * p.tilesPerImg : 0;
}

extern "C" int otp_nhwc_conv_plan(const otp_nhwc_conv_desc* d, int* out8) {
    ConvPlan p;
    if (!out8) return OTP_ERR_BAD_ARG;
    if (!make_plan(d, &p)) return OTP_ERR_UNSUPPORTED;
    out8[0] = p.MB, out8[1] = p.NB, out8[2] = p.CK, out8[3] = p.nChunks, out8[4] = p.nM, out8[5] = p.N * p.tilesPerImg * p.nM;
    out8[6] = p.lds, out8[7] = p.KS;
    return OTP_OK;
}

namespace {
bool make_pack_job(const void* weight, void* wpacked, const otp_nhwc_conv_desc* d, int dgrad, PackJob* jb) {
    if (!make_plan(d, &jb->p)) return false;
    const ConvPlan& p = jb->p;
    const long taps = (long)p.kh * p.kw;
    jb->hb = use_hb(d) ? 1 : (use_hbpw(d) ? 2 : 0);
    jb->hbNTW = otp_hb_ntw(d->Cout), jb->hbChunks = d->Cin / 16, jb->hbKS = otp_hbpw_ks(d->Cin);
    if (!dgrad) {
        jb->so = p.Cin * taps, jb->si = taps, jb->sdy = p.kw, jb->sdx = 1, jb->base = 0;
    } else {      // original weight is (O = d->Cin, I = d->Cout, kh, kw); this conv maps O -> I with flipped taps
        jb->so = taps, jb->si = (long)p.Cout * taps, jb->sdy = -p.kw, jb->sdx = -1, jb->base = taps - 1;
    }
    jb->total = (jb->hb == 1 ? hb_weight_bytes(d) : (jb->hb == 2 ? hbpw_weight_bytes(d) : p.wbytes)) / 2;
    jb->w = static_cast<const float*>(weight);
    jb->out = static_cast<bf16*>(wpacked);
    return true;
}
}  // namespace

extern "C" int otp_nhwc_conv_pack(const void* weight, void* wpacked, const otp_nhwc_conv_desc* d, int dgrad, void* stream) {
    PackJob jb;
    if (!weight || !wpacked) return OTP_ERR_BAD_ARG;
    if (!make_pack_job(weight, wpacked, d, dgrad, &jb)) return OTP_ERR_UNSUPPORTED;
    if (jb.hb) {
        nhwc_pack_hb_kernel<<<grid_for(jb.total), 256, 0, static_cast<hipStream_t>(stream)>>>(jb);
        return otp_launch_status();
    }
    nhwc_pack_kernel<<<grid_for(jb.total), 256, 0, static_cast<hipStream_t>(stream)>>>(jb.w, jb.out, jb.p, jb.so, jb.si, jb.sdy,
                                                                                      jb.sdx, jb.base, jb.total);
    return otp_launch_status();
}

extern "C" size_t otp_nhwc_conv_pack_job_bytes(void) { return sizeof(PackJob); }

extern "C" int otp_nhwc_conv_pack_job(const void* weight, void* wpacked, const otp_nhwc_conv_desc* d, int dgrad, void* job_host) {
    if (!weight || !wpacked || !job_host) return OTP_ERR_BAD_ARG;
    PackJob jb;
    memset(&jb, 0, sizeof(jb));
    if (!make_pack_job(weight, wpacked, d, dgrad, &jb)) return OTP_ERR_UNSUPPORTED;
    memcpy(job_host, &jb, sizeof(jb));
    return OTP_OK;
}

extern "C" int otp_nhwc_conv_pack_batch(const void* jobs_device, int n_jobs, void* stream) {
    if (!jobs_device || n_jobs < 0) return OTP_ERR_BAD_ARG;
    if (n_jobs == 0) return OTP_OK;
    // 32 workgroups per job: the largest weight of the path (384 x 384 x 9 -> 1.5 M packed elements) takes ~180 iterations
    // per thread, a 48 x 48 x 9 one finishes in the first
    nhwc_pack_batch_kernel<<<dim3(32, (unsigned)n_jobs), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const PackJob*>(jobs_device));
    return otp_launch_status();
}

extern "C" int otp_nhwc_conv_bf16_res(const void* x, const void* wpacked, const void* bias, const void* res, void* out,
                                      void* stats, const otp_nhwc_conv_desc* d, void* stream) {
    ConvPlan p;
    if (!x || !wpacked || !out) return OTP_ERR_BAD_ARG;
    if (!make_plan(d, &p)) return OTP_ERR_UNSUPPORTED;
    if (res && (p.out_mode != 0 || stats)) return OTP_ERR_BAD_ARG;   // the sum is an NHWC bf16 tensor; statistics are of conv(x)
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (use_hb(d)) return otp_hb_conv(x, wpacked, bias, res, out, stats, d, st);
    if (use_hbpw(d)) return otp_hbpw_conv(x, wpacked, bias, res, out, stats, d, st);
#define OTP_NHWC_CASE(mb, nb) \
    if (p.MB == mb && p.NB == nb) return launch_conv<mb, nb>(p, x, wpacked, bias, res, out, stats, st)
    OTP_NHWC_CASE(1, 2); OTP_NHWC_CASE(2, 2); OTP_NHWC_CASE(3, 2); OTP_NHWC_CASE(4, 2); OTP_NHWC_CASE(5, 2); OTP_NHWC_CASE(6, 2);
    OTP_NHWC_CASE(7, 2); OTP_NHWC_CASE(8, 2); OTP_NHWC_CASE(9, 2);
    OTP_NHWC_CASE(1, 4); OTP_NHWC_CASE(2, 4); OTP_NHWC_CASE(3, 4); OTP_NHWC_CASE(4, 4); OTP_NHWC_CASE(5, 4); OTP_NHWC_CASE(6, 4);
#undef OTP_NHWC_CASE
    return OTP_ERR_UNSUPPORTED;
}

extern "C" int otp_nhwc_conv_bf16(const void* x, const void* wpacked, const void* bias, void* out, void* stats,
                                  const otp_nhwc_conv_desc* d, void* stream) {
    return otp_nhwc_conv_bf16_res(x, wpacked, bias, nullptr, out, stats, d, stream);
}

extern "C" size_t otp_nhwc_wgrad_workspace(const otp_nhwc_conv_desc* d) {
    WgradPlan p;
    if (!make_wgrad_plan(d, &p)) return 0;
    return (size_t)p.splits * p.nCo * p.nCi * 4 * WG_NBW * 3 * 256 * sizeof(float);
}

extern "C" int otp_nhwc_wgrad_bf16(const void* x, const void* gy, void* grad_weight, void* workspace, size_t workspace_bytes,
                                   const otp_nhwc_conv_desc* d, void* stream) {
    WgradPlan p;
    if (!x || !gy || !grad_weight || !workspace) return OTP_ERR_BAD_ARG;
    if (!make_wgrad_plan(d, &p)) return OTP_ERR_UNSUPPORTED;
    if (workspace_bytes < otp_nhwc_wgrad_workspace(d)) return OTP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // register prefetch of the next tile when the window fits WG_XU units per thread
    const int xunits = p.rowsMax * p.RW * (p.XC / 8 - 1);
    const bool pf = !p.tapmode && xunits <= WG_XU * 256 && p.TPX * 6 <= WG_GU * 256;
    const int grid = p.nCo * p.nCi * p.splits;
    if (p.cg > 0) {
#define OTP_W1_CASE(CG_, SPG_)                                                                                                  \
    if (p.cg == CG_ && p.spg == SPG_) {                                                                                         \
        OTP_ALLOW_BIG_LDS((nhwc_wgrad1x1_kernel<CG_, SPG_>), p.lds);                                                            \
        nhwc_wgrad1x1_kernel<CG_, SPG_><<<grid, 256, p.lds, st>>>(static_cast<const bf16*>(x), static_cast<const bf16*>(gy),   \
                                                                  static_cast<float*>(workspace), p);                          \
    }
        OTP_W1_CASE(2, 1) OTP_W1_CASE(3, 1) OTP_W1_CASE(4, 1) OTP_W1_CASE(2, 2) OTP_W1_CASE(3, 2) OTP_W1_CASE(2, 3)
#undef OTP_W1_CASE
    } else if (pf) {
        OTP_ALLOW_BIG_LDS(nhwc_wgrad_kernel<true>, p.lds);
        nhwc_wgrad_kernel<true><<<grid, 256, p.lds, st>>>(static_cast<const bf16*>(x), static_cast<const bf16*>(gy),
                                                           static_cast<float*>(workspace), p);
    } else {
        OTP_ALLOW_BIG_LDS(nhwc_wgrad_kernel<false>, p.lds);
        nhwc_wgrad_kernel<false><<<grid, 256, p.lds, st>>>(static_cast<const bf16*>(x), static_cast<const bf16*>(gy),
                                                            static_cast<float*>(workspace), p);
    }
    if (otp_launch_status() != OTP_OK) return OTP_ERR_LAUNCH;
    const size_t frag_total = (size_t)p.nCo * p.nCi * 4 * WG_NBW * 3 * 256;
    nhwc_wgrad_reduce_kernel<<<(int)((frag_total + 63) / 64), 256, 0, st>>>(static_cast<const float*>(workspace),
                                                                             static_cast<float*>(grad_weight), p, frag_total);
    return otp_launch_status();
}

extern "C" int otp_nhwc_bn_finalize(const void* partials, int rows, int C, int CS, float count, const void* gamma,
                                    const void* beta, void* mean, void* rstd, void* scale, void* shift, void* running_mean,
                                    void* running_var, float eps, float momentum, void* stream) {
    if (!partials || !gamma || !beta || !mean || !rstd || !scale || !shift || rows <= 0 || C <= 0 || CS < C)
        return OTP_ERR_BAD_ARG;
    bn_finalize_kernel<<<(CS + 7) / 8, 8 * BNF_RL, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const float*>(partials), rows, CS, C, count, static_cast<const float*>(gamma), static_cast<const float*>(beta),
        static_cast<float*>(mean), static_cast<float*>(rstd), static_cast<float*>(scale), static_cast<float*>(shift),
        static_cast<float*>(running_mean), static_cast<float*>(running_var), eps, momentum);
    return otp_launch_status();
}

extern "C" int otp_nhwc_bn_apply(const void* x, const void* scale, const void* shift, const void* res, void* y, void* relu_mask,
                                 size_t pixels, int CS, int relu, void* stream) {
    if (!x || !scale || !shift || !y || CS <= 0 || CS % 8) return OTP_ERR_BAD_ARG;
    const size_t units = pixels * (CS / 8);
    if (res)
        bn_apply_kernel<true><<<grid_for(units), 256, 0, static_cast<hipStream_t>(stream)>>>(
            static_cast<const bf16*>(x), static_cast<const float*>(scale), static_cast<const float*>(shift),
            static_cast<const bf16*>(res), static_cast<bf16*>(y), static_cast<unsigned char*>(relu_mask), units, CS / 8, relu);
    else
        bn_apply_kernel<false><<<grid_for(units), 256, 0, static_cast<hipStream_t>(stream)>>>(
            static_cast<const bf16*>(x), static_cast<const float*>(scale), static_cast<const float*>(shift), nullptr,
            static_cast<bf16*>(y), static_cast<unsigned char*>(relu_mask), units, CS / 8, relu);
    return otp_launch_status();
}

static int bn_bwd_rows(size_t pixels, int* pixPerWg) {
    // a workgroup strides its pixel range with 256 / (C/8) pixel lanes: 64-pixel ranges still give every lane several
    // pixels at 384 channels, and the small maps (12x9 x 80 frames = 8640 pixels) still fill the chip
    int rows = (int)((pixels + 63) / 64);
    if (rows > 1024) rows = 1024;
    if (rows < 1) rows = 1;
    *pixPerWg = (int)((pixels + rows - 1) / rows);
    return (int)((pixels + *pixPerWg - 1) / *pixPerWg);
}

// conv + BatchNorm (batch statistics) + residual + ReLU as ONE call (three launches: otp_nhwc_conv_bf16 with statistics,
// otp_nhwc_bn_finalize, otp_nhwc_bn_apply): the training forward issues ~290 of these per step and is bound by the host once a loop
// synchronises every iteration (a ctypes call costs ~4 us of host time).  vec: 4 x CoutS floats (mean, rstd, scale, shift).
extern "C" int otp_nhwc_conv_bn_bf16(const void* x, const void* wpacked, const void* res, void* conv_out, void* stats, void* vec,
                                     const void* gamma, const void* beta, void* running_mean, void* running_var, float eps,
                                     float momentum, void* y, void* relu_mask, int relu, const otp_nhwc_conv_desc* d, void* stream) {
    if (!x || !wpacked || !conv_out || !stats || !vec || !gamma || !beta || !y || !d) return OTP_ERR_BAD_ARG;
    const int rows = otp_nhwc_conv_stats_rows(d);
    if (rows <= 0) return OTP_ERR_UNSUPPORTED;
    int rc = otp_nhwc_conv_bf16_res(x, wpacked, nullptr, nullptr, conv_out, stats, d, stream);
    if (rc != OTP_OK) return rc;
    const int CS = (d->Cout + 7) / 8 * 8;
    const int Ho = (d->H + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    const size_t pixels = (size_t)d->N * Ho * Wo;
    float* v = static_cast<float*>(vec);
    rc = otp_nhwc_bn_finalize(stats, rows, d->Cout, CS, (float)pixels, gamma, beta, v, v + CS, v + 2 * CS, v + 3 * CS, running_mean,
                              running_var, eps, momentum, stream);
    if (rc != OTP_OK) return rc;
    return otp_nhwc_bn_apply(conv_out, v + 2 * CS, v + 3 * CS, res, y, relu_mask, pixels, CS, relu, stream);
}

extern "C" size_t otp_nhwc_bn_backward_workspace(size_t pixels, int CS) {
    int ppw;
    const int rows = bn_bwd_rows(pixels, &ppw);
    return ((size_t)rows * 2 * CS + 3 * (size_t)CS) * sizeof(float);
}

// gx (and gres = gy * relu-mask when not NULL) from gy; dgamma / dbeta (C floats) overwritten.  y may be NULL when relu == 0.
extern "C" int otp_nhwc_bn_backward(const void* gy, const void* y, const void* x, const void* mean, const void* rstd,
                                    const void* gamma, void* gx, void* gres, void* dgamma, void* dbeta, void* workspace,
                                    size_t workspace_bytes, size_t pixels, int C, int CS, int relu, void* stream) {
    if (!gy || !x || !mean || !rstd || !gamma || !gx || !dgamma || !dbeta || !workspace || (relu && !y) || CS % 8 || CS > 2048)
        return OTP_ERR_BAD_ARG;
    if (workspace_bytes < otp_nhwc_bn_backward_workspace(pixels, CS)) return OTP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int ppw;
    const int rows = bn_bwd_rows(pixels, &ppw);
    float* part = static_cast<float*>(workspace);
    float* coef = part + (size_t)rows * 2 * CS;
    const int C8 = CS / 8;
    if (C8 > 256) return OTP_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(256 / C8) * C8 * 16 * sizeof(float);
#define OTP_BN_BWD(R)                                                                                               \
    {                                                                                                                \
        bn_bwd_reduce_kernel<R><<<rows, 256, lds, st>>>(static_cast<const bf16*>(gy), static_cast<const bf16*>(y),    \
                                                        static_cast<const bf16*>(x), static_cast<const float*>(mean), \
                                                        static_cast<const float*>(rstd), part, pixels, C8, ppw, relu); \
        bn_bwd_finalize_kernel<<<(CS + 7) / 8, 8 * BNF_RL, 0, st>>>(part, rows, CS, C, (float)pixels,                         \
                                                             static_cast<const float*>(gamma),                        \
                                                             static_cast<const float*>(rstd), static_cast<float*>(dgamma), \
                                                             static_cast<float*>(dbeta), coef);                       \
        bn_bwd_apply_kernel<R><<<grid_for(units), 256, 0, st>>>(                                                      \
            static_cast<const bf16*>(gy), static_cast<const bf16*>(y), static_cast<const bf16*>(x),                   \
            static_cast<const float*>(mean), static_cast<const float*>(rstd), coef, static_cast<bf16*>(gx),           \
            static_cast<bf16*>(gres), units, C8, relu);                                                               \
    }
    const size_t units = pixels * C8;
    if (relu == 0) OTP_BN_BWD(0)
    else if (relu == 1) OTP_BN_BWD(1)
    else OTP_BN_BWD(2)
#undef OTP_BN_BWD
    return otp_launch_status();
}

extern "C" int otp_nhwc_upsample_add(const void* low, const void* res, void* out, int N, int H, int W, int CS, int f, int relu,
                                     void* stream) {
    if (!low || !res || !out || f <= 0 || H % f || W % f || CS % 8) return OTP_ERR_BAD_ARG;
    upsample_add_nhwc_kernel<<<grid_for((size_t)N * H * W * (CS / 8)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const bf16*>(low), static_cast<const bf16*>(res), static_cast<bf16*>(out), N, H, W, CS / 8, f, relu);
    return otp_launch_status();
}

extern "C" int otp_nhwc_upsample_add_backward(const void* gy, const void* out, void* gres, void* glow, int N, int Hl, int Wl,
                                              int CS, int f, int relu, void* stream) {
    if (!gy || !gres || !glow || (relu && !out) || f <= 0 || CS % 8) return OTP_ERR_BAD_ARG;
    upsample_add_nhwc_bwd_kernel<<<grid_for((size_t)N * Hl * Wl * (CS / 8)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const bf16*>(gy), static_cast<const bf16*>(out), static_cast<bf16*>(gres), static_cast<bf16*>(glow), N, Hl, Wl,
        CS / 8, f, relu);
    return otp_launch_status();
}

extern "C" int otp_nchw_f32_to_nhwc_bf16(const void* in, void* out, int N, int C, int H, int W, int frame_split, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || (frame_split > 0 && N % frame_split)) return OTP_ERR_BAD_ARG;
    const int CS = (C + 7) / 8 * 8;
    if (frame_split <= 0 && CS >= 32 && CS <= 4096 && (size_t)N * ((H * W + 63) / 64) < (1ull << 31)) {
        nchw_to_nhwc_tile_kernel<<<N * ((H * W + 63) / 64), 256, (size_t)64 * ((CS / 2) | 1) * 4, static_cast<hipStream_t>(stream)>>>(
            static_cast<const float*>(in), static_cast<bf16*>(out), C, H * W, CS);
        return otp_launch_status();
    }
    nchw_to_nhwc_kernel<<<grid_for((size_t)N * H * W), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const float*>(in), static_cast<bf16*>(out), N, C, H * W, CS, frame_split);
    return otp_launch_status();
}

extern "C" int otp_nhwc_bf16_to_nchw_f32(const void* in, void* out, int N, int C, int H, int W, void* stream) {
    if (!in || !out || N <= 0 || C <= 0) return OTP_ERR_BAD_ARG;
    const int CS = (C + 7) / 8 * 8;
    if (CS > 2048) return OTP_ERR_UNSUPPORTED;
    const int tiles = N * ((H * W + 63) / 64);
    nhwc_to_nchw_kernel<<<tiles, 256, (size_t)64 * (CS + 8) * 2, static_cast<hipStream_t>(stream)>>>(
        static_cast<const bf16*>(in), static_cast<float*>(out), N, C, H * W, CS);
    return otp_launch_status();
}

extern "C" int otp_nhwc_dilate(const void* in, void* out, int N, int Hi, int Wi, int s, int H, int W, int CS, void* stream) {
    if (!in || !out || s <= 0 || CS % 8) return OTP_ERR_BAD_ARG;
    dilate_nhwc_kernel<<<grid_for((size_t)N * H * W * (CS / 8)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const bf16*>(in), static_cast<bf16*>(out), N, Hi, Wi, s, H, W, CS / 8);
    return otp_launch_status();
}

#ifdef OTP_NHWC_TIMING
extern "C" int otp_nhwc_read_stamps(void* host_out, size_t bytes) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(otp_nhwc_stamps), bytes) == hipSuccess ? OTP_OK : OTP_ERR_LAUNCH;
}
#endif

extern "C" size_t otp_nhwc_channel_sum_workspace(size_t pixels, int CS) {
    int ppw;
    const int rows = bn_bwd_rows(pixels, &ppw);
    return (size_t)rows * CS * sizeof(float);
}

// out[c] (C floats) = sum over pixels of g[pixel][c], g NHWC bf16 with channel stride CS (bias gradients)
extern "C" int otp_nhwc_channel_sum(const void* g, void* out, void* workspace, size_t workspace_bytes, size_t pixels, int C,
                                    int CS, void* stream) {
    if (!g || !out || !workspace || C <= 0 || CS < C || CS % 8 || CS > 2048) return OTP_ERR_BAD_ARG;
    if (workspace_bytes < otp_nhwc_channel_sum_workspace(pixels, CS)) return OTP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int ppw;
    const int rows = bn_bwd_rows(pixels, &ppw);
    const int C8 = CS / 8;
    channel_sum_nhwc_kernel<<<rows, 256, (size_t)(256 / C8) * CS * sizeof(float), st>>>(
        static_cast<const bf16*>(g), static_cast<float*>(workspace), pixels, C8, ppw);
    channel_sum_nhwc_finish_kernel<<<(CS + 7) / 8, 256, 0, st>>>(static_cast<const float*>(workspace), static_cast<float*>(out),
                                                                 rows, CS, C);
    return otp_launch_status();
}

extern "C" int otp_gelu_bf16_forward(const void* x, void* y, size_t n, void* stream) {
    if (!x || !y || n % 8) return OTP_ERR_BAD_ARG;
    gelu_bf16_fwd_kernel<<<grid_for(n / 8), 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const bf16*>(x),
                                                                                          static_cast<bf16*>(y), n / 8);
    return otp_launch_status();
}

// The TransformerBlock MLP interior with its element-wise passes folded into the projections' epilogues (csrc/hb.hip's pointwise kernel):
// up: pre = bf16(W1 x + b1) AND act = dropout(gelu(pre), p) (+ keep bits) in one launch; the down-projection's input gradient times
// gelu'(pre) and the dropout factor in one launch.  desc: the (N, 1, T, C) convolution of that launch (for the gradient: the input-gradient
// convolution, channels swapped).  OTP_ERR_UNSUPPORTED when the shape is not on the pointwise kernel - the caller keeps the separate launches.
extern "C" int otp_nhwc_mlp_fused_supported(const otp_nhwc_conv_desc* d) { return (d && d->out_mode == 0 && use_hbpw(d)) ? 1 : 0; }

extern "C" int otp_nhwc_mlp_up_bf16(const void* x, const void* wpacked, const void* bias, void* pre, void* act, void* keep_bits, float p,
                                    unsigned long long seed, const otp_nhwc_conv_desc* d, void* stream) {
    if (!x || !wpacked || !pre || !act || !keep_bits || !(p >= 0.f) || !(p < 1.f)) return OTP_ERR_BAD_ARG;
    if (!otp_nhwc_mlp_fused_supported(d)) return OTP_ERR_UNSUPPORTED;
    otp_hbpw_epi e{};
    e.mode = 1, e.out2 = act, e.keep = static_cast<unsigned char*>(keep_bits);
    e.s0 = (uint32_t)seed, e.s1 = (uint32_t)(seed >> 32), e.thr = (uint32_t)(p * 65536.f + 0.5f);
    e.scale = 65536.f / (float)(65536u - e.thr);
    return otp_hbpw_conv(x, wpacked, bias, nullptr, pre, nullptr, d, static_cast<hipStream_t>(stream), &e);
}

extern "C" int otp_nhwc_mlp_down_dgrad_bf16(const void* gy, const void* wpacked, const void* pre, const void* keep_bits, void* grad_pre,
                                            float p, const otp_nhwc_conv_desc* d, void* stream) {
    if (!gy || !wpacked || !pre || !keep_bits || !grad_pre || !(p >= 0.f) || !(p < 1.f)) return OTP_ERR_BAD_ARG;
    if (!otp_nhwc_mlp_fused_supported(d)) return OTP_ERR_UNSUPPORTED;
    otp_hbpw_epi e{};
    e.mode = 2, e.aux = pre, e.keep = const_cast<unsigned char*>(static_cast<const unsigned char*>(keep_bits));
    e.thr = (uint32_t)(p * 65536.f + 0.5f);
    e.scale = 65536.f / (float)(65536u - e.thr);
    return otp_hbpw_conv(gy, wpacked, nullptr, nullptr, grad_pre, nullptr, d, static_cast<hipStream_t>(stream), &e);
}

extern "C" int otp_gelu_dropout_bf16_forward(const void* x, void* y, void* keep_bits, size_t n, float p, unsigned long long seed,
                                             void* stream) {
    if (!x || !y || !keep_bits || n % 8 || !(p >= 0.f) || !(p < 1.f)) return OTP_ERR_BAD_ARG;
    const uint32_t thr = (uint32_t)(p * 65536.f + 0.5f);
    gelu_dropout_bf16_fwd_kernel<<<grid_for(n / 8), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const bf16*>(x), static_cast<bf16*>(y), static_cast<unsigned char*>(keep_bits), n / 8, (uint32_t)seed,
        (uint32_t)(seed >> 32), thr, 65536.f / (float)(65536u - thr));
    return otp_launch_status();
}

extern "C" int otp_gelu_dropout_bf16_backward(const void* x, const void* grad_y, const void* keep_bits, void* grad_x, size_t n, float p,
                                              void* stream) {
    if (!x || !grad_y || !keep_bits || !grad_x || n % 8 || !(p >= 0.f) || !(p < 1.f)) return OTP_ERR_BAD_ARG;
    const uint32_t thr = (uint32_t)(p * 65536.f + 0.5f);
    gelu_dropout_bf16_bwd_kernel<<<grid_for(n / 8), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const bf16*>(x), static_cast<const bf16*>(grad_y), static_cast<const unsigned char*>(keep_bits),
        static_cast<bf16*>(grad_x), n / 8, 65536.f / (float)(65536u - thr));
    return otp_launch_status();
}

extern "C" int otp_gelu_bf16_backward(const void* x, const void* grad_y, void* grad_x, size_t n, void* stream) {
    if (!x || !grad_y || !grad_x || n % 8) return OTP_ERR_BAD_ARG;
    gelu_bf16_bwd_kernel<<<grid_for(n / 8), 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const bf16*>(x), static_cast<const bf16*>(grad_y), static_cast<bf16*>(grad_x), n / 8);
    return otp_launch_status();
}
