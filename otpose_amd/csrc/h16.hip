// fp16-storage eval kernels of the HRNet backbone (BASELINE.json configs[4]: "fp16"; cfg.MODEL.DTYPE = "fp16").
//
// The reference's own native op dispatches half (thirdparty/deform_conv/src/deform_conv_cuda_kernel.cu:719,
// AT_DISPATCH_FLOATING_TYPES_AND_HALF) but its model code never runs below fp32 (no AMP anywhere), so an fp16 forward
// is an extension, validated against the fp32 HIP engine and the oracle with a stated tolerance (tests/test_gpu_h16.py).
// What these kernels compute is model/HRNet.py:116-152 with BatchNorm (eval) folded: conv3x3 (stride 1: BasicBlock :500-530,
// Bottleneck conv2 :551-571, transition :213-229; stride 2: fuse chains :442-470, transitions, stem conv2 :66-72), conv1x1
// (Bottleneck conv1 / conv3, fuse up-sampling paths :426-439, final_layer :108-114), the stem conv (:118-120) and the fuse rows'
// upsample-accumulate (:487-494).
//
// Arithmetic: operands are IEEE half - activations AS STORED, weights rounded once (times a per-layer power of two so small
// BatchNorm-folded weights stay normal numbers) - ONE v_mfma_f32_16x16x32_f16 per product where the fp32 engine spends three
// (csrc/convs.hip), fp32 accumulation, shift / residual / ReLU in fp32, one rounding to half per stored value.
//
// Storage: "H8" images.  A logical (N, C, H, W) tensor, C % 8 == 0, is [N][C / 8][H * W] records of 16 bytes = the 8 halves of
// channels 8 g .. 8 g + 7 at pixel p - 2 bytes per element.  A record IS one lane's B operand of a k-slot of the MFMA, a plane
// (n, g) is contiguous in p, so a conv stages its window with the LDS-DMA only (csrc/convs.hip's design with half the planes) and
// a 1x1 conv loads its operand fragments straight from global memory.  (gtot, goff): a tensor may be a range of channel groups
// of a wider H8 tensor - channel concatenations cost nothing.
#include "common.h"
#include "hb.h"
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {

// csrc/hb.hip compiles this file a second time with OTP_H16_BF16: the 3x3 kernel on bfloat16 NHWC tensors for the training step's
// forward / input-gradient convolutions (csrc/nhwc.hip dispatches to it) - same window, weight stream and MFMA schedule, the
// records' addresses generalised to (pixel stride, group stride), per-tile channel statistics instead of the folded BatchNorm.
#ifdef OTP_H16_BF16
typedef __bf16 h16;
#define H_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#else
typedef _Float16 h16;
#define H_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#endif
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef OTP_H16_TIMING
// development build only (tools/h16_timing.sh): per-workgroup phase stamps, never in libotpose_hip.so
__device__ unsigned long long otp_h16_stamps[8192 * 32];
#define HSTAMP(slot)                                                                                                       \
    do {                                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 8192) otp_h16_stamps[blockIdx.x * 32 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define HSTAMP(slot)
#endif

constexpr int HKS = OTP_HB_KS;            // k-steps per 16-channel chunk: 18 (tap, group) slots of 8 channels in 5 x 4 (2 empty)
constexpr int HOOB = -16;                 // buffer offset outside every descriptor: loads return / the LDS-DMA writes zeros
// packed weights of a (cout block, 16-channel chunk): 4 full k-steps x NTW tiles x 1 KB, then the half-filled fifth (k-slots 16, 17
// on lanes 0 .. 31: 512 bytes per tile)
__host__ __device__ constexpr int hwb(int ntw) { return otp_hb_wb(ntw); }
__host__ __device__ constexpr int hwp(int ntw) { return (hwb(ntw) + 1023) / 1024; }     // 1 KB pieces (the last may be half)
// bf16 (training) build: the four waves' partial channel sums behind the weight image
#ifdef OTP_H16_BF16
__host__ __device__ constexpr int hsred(int ntw) { return 4 * 2 * ntw * 16 * 4; }
#else
__host__ __device__ constexpr int hsred(int) { return 0; }
#endif

__device__ __forceinline__ uint32_t hdiv(uint32_t i, uint32_t magic) { return magic ? __umulhi(i, magic) : i; }
// a * b for per-lane index arithmetic whose operands stay below 2^24 (checked by the plan): v_mul_u32_u24 runs at full rate,
// v_mul_lo_u32 at a quarter - 40 of them sat in the set-up of every tile
__device__ __forceinline__ int hmul(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }
uint32_t hmagic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }   // exact while i * d < 2^32

__device__ __forceinline__ u32x4 hpack8(const float (&v)[8]) {
    uint32_t h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __builtin_bit_cast(uint32_t, (h16x2){(h16)v[2 * i], (h16)v[2 * i + 1]});
    return (u32x4){h[0], h[1], h[2], h[3]};
}
#ifdef OTP_H16_BF16
__device__ __forceinline__ f32x2 hwiden(uint32_t pair) {
    return (f32x2){__builtin_bit_cast(float, pair << 16), __builtin_bit_cast(float, pair & 0xffff0000u)};
}
#else
__device__ __forceinline__ f32x2 hwiden(uint32_t pair) { return __builtin_convertvector(__builtin_bit_cast(h16x2, pair), f32x2); }
#endif

// sum over the 16 lanes of a DPP row (one MFMA pixel column group), every lane ends with the total (csrc/nhwc.hip: row16_sum)
template <int CTRL>
__device__ __forceinline__ float hdpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float hrow16_sum(float v) {
    v += hdpp<0xB1>(v);           // quad_perm [1,0,3,2]
    v += hdpp<0x4E>(v);           // quad_perm [2,3,0,1]
    v += hdpp<0x124>(v);          // row_ror:4
    v += hdpp<0x128>(v);          // row_ror:8
    return v;
}

// Output-channel row of an MFMA tile <-> channel (the convention of csrc/convs.hip): cout tiles go in pairs (2 tp, 2 tp + 1) whose
// rows are permuted so that lane (pixel, kl) ends up with 8 CONSECUTIVE channels 32 tp + 8 kl .. + 7 of its pixel = one H8 record;
// a tile without a partner keeps the identity (4 consecutive channels per lane = half a record).
// (the definitions live in csrc/hb.h: the bf16 build's weight packer is csrc/nhwc.hip's)
__host__ __device__ inline bool hpaired(int co_blk, int t, int ntw, int Cout) { return otp_hb_paired(co_blk, t, ntw, Cout); }
__host__ __device__ inline int hrow2ch(int co_blk, int t, int row, int ntw, int Cout) { return otp_hb_row2ch(co_blk, t, row, ntw, Cout); }

// instruction order of one (k-step, pixel tile) block: NM MFMAs and NR LDS reads interleaved (csrc/convs.hip: sblock_sched)
template <int NM, int NR>
__device__ __forceinline__ void hblock_sched() {
    if constexpr (NR == 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    } else if constexpr (NR >= NM) {
        // [MFMA, read] pairs, the surplus reads behind the last MFMA's pair
#pragma unroll
        for (int g = 0; g < NM - 1; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR - (NM - 1), 0);
    } else {
#pragma unroll
        for (int g = 0; g < NR; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
    }
}

// ================================================================================================================================
// 3x3 / pad 1 convolution, stride 1 or 2, H8 -> H8 (+ H8 residual) (+ ReLU)
// ================================================================================================================================
struct HPlan {
    int N, C, H, W, HW, Ho, Wo, HWo, Cout, total;      // total = N * Ho * Wo output pixels
    int act;
    // where records live, in bytes: record (image n, channel group g, pixel p) of a tensor is at base + n imgB + g gS + p pS.
    // H8: pS = 16, gS = 16 HW, imgB = 16 gtot HW, base = 16 goff HW.  NHWC with CS channels: pS = 2 CS, gS = 16, imgB = 2 CS HW
    int in_pS, in_gS, in_imgB, out_pS, out_gS, out_imgB, res_pS, res_gS, res_imgB;
    size_t in_base, out_base, res_base;
    float* stats;                                      // bf16 build: per-tile channel sums [nTiles][2][Cout] of the stored values, or NULL
    float pre, post;                                   // weights carry 2^k = pre; the sum is multiplied by post = 2^-k
    int NTW, nN, nTiles, nChunks, tpx, NPT;
    int CK;                                            // input channels per WINDOW stage (a multiple of 16 that divides Cin)
    int VR, W1, NIW, NV, pl;                           // virtual rows per image (H + 1), records per virtual row (W + 1), 64-record
                                                       // pieces / records / bytes of a window plane
    uint32_t mHWo, mWo, mW1, mVR;
    unsigned* rflag;
};

// Workgroup = 64 NPT flattened output pixels x 16 NTW output channels, 4 waves; K = input channels x 9 taps.
// LDS: [CK / 8 window planes | one 16-channel weight chunk].  Window (stride 1, csrc/convs.hip): record index = (virtual row) *
// (W + 1) + 1 + x from the first record a tap of the tile reads, one zero record between rows, one zero row above every image;
// (stride 2, csrc/convs2.hip): a virtual row is [0 | odd columns | even columns] so that consecutive output pixels read consecutive
// records.  The window arrives by LDS-DMA; padding = lanes whose source offset is outside the descriptor.
// With ONE product per multiply a 16-channel chunk is ~1 k cycles of MFMA against a ~4 k cycle HBM round trip, and a ring of
// 16-channel stages measured no better than no ring at all (profiles/r05_h16_ring_sweep.txt: what hides latency is other
// workgroups, and every stage costs them LDS).  So the WINDOW of CK = 48 .. 96 channels is staged at once - one round trip per CK
// channels instead of one per 16 - and only the weights (L2-resident, 13.5 KB per 16 channels) stream: the next chunk's
// weights are loaded into registers under the current chunk's MFMAs and written to the LDS behind them.
#ifndef OTP_H16_MINWG
#define OTP_H16_MINWG 3                   /* development A/B (tools/lib_variant.sh): waves per SIMD the register budget is cut for */
#endif
// MODE (bf16 build; compile time - as run-time branches the statistics / residual forms cost the plain one registers): 0 plain,
// 1 per-tile channel statistics, 2 residual
template <int NTW, int NPT, int STRIDE, int MODE = 0>
__global__ __launch_bounds__(256, STRIDE == 1 ? OTP_H16_MINWG : 2) void h16_conv3x3_kernel(const unsigned char* __restrict__ xs,
                                                                                const unsigned char* __restrict__ wpk,
                                                                                const float* __restrict__ shift,
                                                                                const unsigned char* res, unsigned char* out,
                                                                                const HPlan P) {
    constexpr int BM = 64 * NPT, WB = hwb(NTW), WU = WB / 16, NWJ = (WU + 255) / 256, NBLK = HKS * NPT, MAXJ = STRIDE == 1 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PL = P.pl, NPL = P.CK >> 3, SUB = P.CK >> 4;          // bytes of a window plane, planes / weight chunks per stage
    unsigned char* const wl = smem + NPL * PL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kl = lane >> 4;
    const bool upper = kl >= 2;

    // workgroup -> (pixel tile, cout block): XCD x walks a contiguous tile range, the blocks of a tile back to back (shared L2 lines)
    const int xcd = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
    const int tl = jb / P.nN, cb = jb - tl * P.nN;
    const int tile = xcd * P.tpx + tl;
    if (tile >= P.nTiles) return;
    HSTAMP(0);
#ifdef OTP_H16_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 8192) otp_h16_stamps[blockIdx.x * 32 + 30] = __builtin_amdgcn_s_memrealtime();
#endif
    const int P0 = tile * BM;
    const int n0 = P0 / P.HWo, p0 = P0 - n0 * P.HWo;               // (uniform, once per workgroup)
    const int y0 = (int)hdiv((uint32_t)p0, P.mWo);
    const int x0 = STRIDE == 1 ? p0 - y0 * P.Wo : 0;               // stride 1: the window starts at record x0 of its first row
    const int Vf = n0 * P.VR + STRIDE * y0;                        // first virtual row of the window
    const int imgB = P.in_imgB;                                    // bytes of one image of the input tensor
    const int co_blk = cb * NTW * 16;

    // ---- window pieces of this wave: piece k = wave + 4 j covers window records 64 k .. 64 k + 63 of every plane ----------------
    // (arrays of a fixed bound: with a template-dependent bound captured by the lambda below hipcc 7.2's host pass emits no kernel stub)
    int voff[4];
    bool vlive[4];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int v = 64 * (wave + 4 * j) + lane;
        vlive[j] = v < P.NV;
        const int vv = v + x0;
        const int r = (int)hdiv((uint32_t)vv, P.mW1), i = vv - hmul(r, P.W1);
        const int V = Vf + r;
        const int n = (int)hdiv((uint32_t)V, P.mVR), yy = V - hmul(n, P.VR);
        const int col = STRIDE == 1 ? i - 1 : (i <= P.Wo ? 2 * (i - 1) + 1 : 2 * (i - P.Wo - 1));   // stride 2: odd columns, then even
        const bool ok = i >= 1 && yy >= 1 && n < P.N;
        voff[j] = ok ? (n - n0) * imgB + hmul(hmul(yy - 1, P.W) + col, P.in_pS) : HOOB;    // (imgB may pass 2^24: a full multiply)
    }
    const size_t left = (size_t)(P.N - n0) * imgB - P.in_base;
    const otp_rsrc rin = make_rsrc32(xs + (size_t)n0 * imgB + P.in_base, left > 0x7fffff00ull ? 0x7fffff00u : (unsigned)left);
    const otp_rsrc rw = make_rsrc32(wpk, (unsigned)((size_t)P.nN * P.nChunks * WB));

    // the window of window-stage sc: planes of channels CK sc .. CK sc + CK - 1
    auto stage_window = [&](int sc) __attribute__((always_inline)) {
        const int so0 = sc * NPL * P.in_gS, dso = P.in_gS;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int k = wave + 4 * j;
            // (piece outermost: its lane mask - a plane's last piece is partial, lanes past it write nothing - is set once for all the
            //  planes; inside, a plane costs two scalar adds and the DMA instruction)
            if (k < P.NIW && vlive[j]) {
                int so = so0, ld = k * 1024;
                for (int pl = 0; pl < NPL; ++pl, so += dso, ld += PL)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + ld), 16, voff[j], so, 0, 0);
            }
        }
    };
    // the weights of 16-channel chunk c: global -> registers (under the previous chunk's MFMAs) -> LDS; unit u = tid + 256 j
    u32x4 wr[4];                                                   // (fixed bound, NWJ <= 4: see voff above)
    static_assert(NWJ <= 4, "weight units per thread");
    auto wload = [&](int c) __attribute__((always_inline)) {
        const int wb = (cb * P.nChunks + c) * WB;
#pragma unroll
        for (int j = 0; j < NWJ; ++j)
            wr[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, tid + 256 * j < WU ? (tid + 256 * j) * 16 : HOOB, wb, 0));
    };
    auto wstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NWJ; ++j)
            if (tid + 256 * j < WU) *reinterpret_cast<u32x4*>(wl + (tid + 256 * j) * 16) = wr[j];
    };
    stage_window(0);
    wload(0);
    HSTAMP(1);

    // ---- per pixel tile: fragment address, lane offsets into the output / residual images --------------------------------------
    const unsigned obytes = (unsigned)((size_t)P.N * P.out_imgB - P.out_base);
    const otp_rsrc ro = make_rsrc32(out + P.out_base, obytes);
    const otp_rsrc rr = make_rsrc32(res ? res + P.res_base : xs, res ? (unsigned)((size_t)P.N * P.res_imgB - P.res_base) : 0u);
    const otp_rsrc rsh = make_rsrc32(shift ? shift : reinterpret_cast<const float*>(xs), shift ? (unsigned)(P.Cout * 4) : 0u);
    int pb[NPT], toff[HKS], offO[NPT], offR[NPT], ch0[NTW];
    f32x4 acc[NTW][NPT];
    {
#pragma unroll
        for (int s = 0; s < HKS; ++s) {
            const int q = 4 * s + kl;
            int tap = q >> 1;
            if (tap > 8) tap = 8;                                  // zero weights: any finite data
            const int dy = tap / 3, dx = tap - dy * 3;
            // record of the tap relative to the pixel's record of tap row dy = 0 (stride 2: parity de-interleaved virtual rows)
            const int rx = STRIDE == 1 ? dx - x0 : (dx == 0 ? 0 : (dx == 1 ? P.Wo + 1 : 1));
            toff[s] = (hmul(dy, P.W1) + rx) * 16 + (q & 1) * PL;
        }
        f32x4 sh[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            ch0[t] = hrow2ch(co_blk, t, 4 * kl, NTW, P.Cout);
            sh[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsh, co_blk + 16 * t < P.Cout ? ch0[t] * 4 : HOOB, 0, 0));
        }
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
            int m = (wave * NPT + p) * 16 + i16;
            const bool pv = P0 + m < P.total;
            if (!pv) m = P.total - 1 - P0;                         // tail tile: a finite address, the result is dropped
            const int q = p0 + m;
            const int dn = (int)hdiv((uint32_t)q, P.mHWo), pi = q - hmul(dn, P.HWo);
            const int y = (int)hdiv((uint32_t)pi, P.mWo), x = pi - hmul(y, P.Wo);
            pb[p] = (hmul(hmul(n0 + dn, P.VR) + STRIDE * y - Vf, P.W1) + x) * 16;
            offO[p] = pv ? (n0 + dn) * P.out_imgB + hmul(pi, P.out_pS) : HOOB;
            offR[p] = (pv && res) ? (n0 + dn) * P.res_imgB + hmul(pi, P.res_pS) : HOOB;
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t][p] = sh[t] * P.pre;
        }
    }

    // One 16-channel chunk: NBLK (k-step, pixel tile) blocks of NTW MFMAs; B fragments are read two blocks ahead (ring of three),
    // the weight fragments of the next k-step two blocks before it starts (csrc/convs.hip)
    auto mfma_phase = [&](int sub) __attribute__((always_inline)) {
        const unsigned char* win = smem + sub * 2 * PL;            // the two planes of this chunk
        h16x8 a[2][NTW], b[3];
        auto load_a = [&](int ab, int s) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (s < HKS - 1) {
                    a[ab][t] = *reinterpret_cast<const h16x8*>(wl + (s * NTW + t) * 1024 + lane * 16);
                } else {
                    // last k-step: k-slots 16, 17 (tap 8) on the lanes kl = 0, 1; kl = 2, 3 multiply zeros (not stored)
                    const h16x8 h = *reinterpret_cast<const h16x8*>(wl + (HKS - 1) * NTW * 1024 + t * 512 + (lane & 31) * 16);
                    const h16x8 z = __builtin_bit_cast(h16x8, (u32x4){0u, 0u, 0u, 0u});
                    a[ab][t] = upper ? z : h;
                }
            }
        };
        auto load_b = [&](int bb, int blk) __attribute__((always_inline)) {
            b[bb] = *reinterpret_cast<const h16x8*>(win + (pb[blk % NPT] + toff[blk / NPT]));
        };
        load_a(0, 0);
        load_b(0, 0);
        if (NBLK > 1) load_b(1, 1);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const int s = blk / NPT, p = blk % NPT, cur = blk % 3, sa = s & 1;
            const bool nb = blk + 2 < NBLK;
            const bool na = (NPT >= 2 ? p == NPT - 2 : true) && s + 1 < HKS;
            if (nb) load_b((blk + 2) % 3, blk + 2);
            if (na) load_a(sa ^ 1, s + 1);
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t][p] = H_MFMA(a[sa][t], b[cur], acc[t][p], 0, 0, 0);
            if (!nb && !na) hblock_sched<NTW, 0>();
            else if (nb && na) hblock_sched<NTW, 1 + NTW>();
            else if (na) hblock_sched<NTW, NTW>();
            else hblock_sched<NTW, 1>();
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    HSTAMP(2);
    const int nStages = P.C / P.CK;
    for (int sc = 0; sc < nStages; ++sc) {
        if (sc > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // every wave is done with the previous window (and weight chunk)
            asm volatile("" ::: "memory");
            stage_window(sc);
            wload(sc * SUB);
        }
        for (int sub = 0; sub < SUB; ++sub) {
            if (sub > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // every wave is done with the previous chunk's weight image
                asm volatile("" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the window (and its weight units) have landed
            }
            if (sc == 0 && sub < 3) HSTAMP(3 + 4 * sub);
            wstore();
            asm volatile("" ::: "memory");
            if (sub + 1 < SUB) wload(sc * SUB + sub + 1);          // the next chunk's weights fly under this chunk's MFMAs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // the weight image (and everybody's window pieces) are in the LDS
            asm volatile("" ::: "memory");
            if (sc == 0 && sub < 3) HSTAMP(4 + 4 * sub);
            mfma_phase(sub);
            if (sc == 0 && sub < 3) HSTAMP(5 + 4 * sub);
        }
    }
    HSTAMP(16);

    // ---- epilogue: post scale, residual, range guard, ReLU, one rounding to half, 16-byte record stores ------------------------
    // (the residual records are loaded here, not held across the chunk loop: 24 registers less is a fourth workgroup per CU, and
    //  another workgroup's MFMAs cover the round trip)
    u32x4 rres[(NTW + 1) / 2][NPT];
    constexpr bool load_res = MODE == 2;                           // (a template variant: the plain form carries no residual loads / adds)
#pragma unroll
    for (int t = 0; load_res && t < NTW; t += 2) {
        const bool tav = ch0[t] < P.Cout;                          // (per lane: Cout % 8 == 0, a lane's record exists or does not)
        const int go = hmul(ch0[t] >> 3, P.res_gS);
        if (hpaired(co_blk, t, NTW, P.Cout)) {
#pragma unroll
            for (int p = 0; p < NPT; ++p)
                rres[t >> 1][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rr, (tav && offR[p] != HOOB) ? offR[p] + go : HOOB, 0, 0));
        } else {
            const int half = (ch0[t] >> 2) & 1;
#pragma unroll
            for (int p = 0; p < NPT; ++p) {
                const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                    rr, (tav && offR[p] != HOOB) ? offR[p] + go + 8 * half : HOOB, 0, 0));
                rres[t >> 1][p] = (u32x4){h[0], h[1], 0u, 0u};
            }
        }
    }
    HSTAMP(17);
#ifdef OTP_H16_BF16
    // training form (csrc/nhwc.hip's contract): out = bf16(conv + bias); the per-tile channel sums / sums of squares are those of
    // the ROUNDED values (what BatchNorm will normalise); a residual is added to the rounded result and the sum rounded again -
    // bit for bit a separate bf16 add.  Sums: 16 lanes of a DPP row hold the 16 pixels of a tile -> row sum -> the four waves'
    // partials through the LDS behind the weight image, added in a fixed order.
    constexpr int CB = NTW * 16;
    float* sred = reinterpret_cast<float*>(wl + WB);               // [4 waves][2][CB]
#pragma unroll
    for (int t = 0; t < NTW; t += 2) {
        const bool paired = hpaired(co_blk, t, NTW, P.Cout);       // (uniform)
        const int t1 = t + 1 < NTW ? t + 1 : t;
        const bool tav = ch0[t] < P.Cout;                          // (per lane)
        const int go = hmul(ch0[t] >> 3, P.out_gS);
        float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
            float f[8] = {acc[t][p][0], acc[t][p][1], acc[t][p][2], acc[t][p][3], paired ? acc[t1][p][0] : 0.f,
                          paired ? acc[t1][p][1] : 0.f, paired ? acc[t1][p][2] : 0.f, paired ? acc[t1][p][3] : 0.f};
            u32x4 rec = hpack8(f);
            const bool pv = offO[p] != HOOB;
            if constexpr (MODE != 0) {
                const f32x2 w0 = hwiden(rec[0]), w1 = hwiden(rec[1]), w2 = hwiden(rec[2]), w3 = hwiden(rec[3]);
                float fr[8] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, w3.x, w3.y};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = pv ? fr[e] : 0.f;
                    s1[e] += v;
                    s2[e] += v * v;
                }
                if constexpr (MODE == 2) {
                    const u32x4 rq = rres[t >> 1][p];
                    const f32x2 r0 = hwiden(rq[0]), r1 = hwiden(rq[1]), r2 = hwiden(rq[2]), r3 = hwiden(rq[3]);
                    float g[8] = {fr[0] + r0.x, fr[1] + r0.y, fr[2] + r1.x, fr[3] + r1.y, fr[4] + r2.x, fr[5] + r2.y, fr[6] + r3.x, fr[7] + r3.y};
                    rec = hpack8(g);
                }
            }
            if (paired) {
                __builtin_amdgcn_raw_buffer_store_b128(rec, ro, (tav && pv) ? offO[p] + go : HOOB, 0, 0);
            } else {
                const int half = (ch0[t] >> 2) & 1;
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){rec[0], rec[1]}, ro, (tav && pv) ? offO[p] + go + 8 * half : HOOB, 0, 0);
            }
        }
        if constexpr (MODE == 1) {
            const int cl = ch0[t] - co_blk;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < 4 || paired) {
                    const float a = hrow16_sum(s1[e]), b = hrow16_sum(s2[e]);
                    if (i16 == 0) {
                        sred[(wave * 2 + 0) * CB + cl + e] = a;
                        sred[(wave * 2 + 1) * CB + cl + e] = b;
                    }
                }
            }
        }
    }
    if constexpr (MODE == 1) {
        __syncthreads();
        for (int i = tid; i < 2 * CB; i += 256) {
            const int which = i / CB, c = i - which * CB, co = co_blk + c;
            if (co < P.Cout) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += sred[(w * 2 + which) * CB + c];
                P.stats[((size_t)tile * 2 + which) * P.Cout + co] = v;
            }
        }
    }
#else
    bool bad = false;
#pragma unroll
    for (int t = 0; t < NTW; t += 2) {
        const bool paired = hpaired(co_blk, t, NTW, P.Cout);       // (uniform)
        const int t1 = t + 1 < NTW ? t + 1 : t;
        const bool tav = ch0[t] < P.Cout;                          // (per lane)
        const int go = hmul(ch0[t] >> 3, P.out_gS);
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
            const u32x4 rq = load_res ? rres[t >> 1][p] : (u32x4){0u, 0u, 0u, 0u};
            const f32x2 r0 = hwiden(rq[0]), r1 = hwiden(rq[1]), r2 = hwiden(rq[2]), r3 = hwiden(rq[3]);
            float f[8] = {acc[t][p][0] * P.post + r0.x, acc[t][p][1] * P.post + r0.y, acc[t][p][2] * P.post + r1.x,
                          acc[t][p][3] * P.post + r1.y,
                          paired ? acc[t1][p][0] * P.post + r2.x : 0.f, paired ? acc[t1][p][1] * P.post + r2.y : 0.f,
                          paired ? acc[t1][p][2] * P.post + r3.x : 0.f, paired ? acc[t1][p][3] * P.post + r3.y : 0.f};
#pragma unroll
            for (int e = 0; e < 8; ++e) bad |= otp_out_of_range(f[e]);
            if (P.act == OTP_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = otp_relu(f[e]);
            }
            const u32x4 rec = hpack8(f);
            if (paired) {
                __builtin_amdgcn_raw_buffer_store_b128(rec, ro, (tav && offO[p] != HOOB) ? offO[p] + go : HOOB, 0, 0);
            } else {
                const int half = (ch0[t] >> 2) & 1;
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){rec[0], rec[1]}, ro, (tav && offO[p] != HOOB) ? offO[p] + go + 8 * half : HOOB,
                                                      0, 0);
            }
        }
    }
    otp_range_report(P.rflag, bad, OTP_RANGE_H16);
#endif
    HSTAMP(19);
#ifdef OTP_H16_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 8192) otp_h16_stamps[blockIdx.x * 32 + 31] = __builtin_amdgcn_s_memrealtime();
#endif
}

int h16_ntw(int Cout) { return otp_hb_ntw(Cout); }

int h16_window_records(const HPlan& P, int bm, int stride) {
    int NV = 0;
    for (int t = 0; t < P.nTiles; ++t) {
        const int a = t * bm, b = (a + bm < P.total ? a + bm : P.total) - 1;
        const int na = a / P.HWo, ya = (a % P.HWo) / P.Wo, xa = (a % P.HWo) % P.Wo;
        const int nb = b / P.HWo, yb = (b % P.HWo) / P.Wo, xb = (b % P.HWo) % P.Wo;
        int v;
        if (stride == 1) {
            const int rows = (nb * P.VR + yb + 1) - (na * P.VR + ya);   // virtual rows between the window's first and the last pixel's
            v = (rows + 1) * P.W1 + xb - xa + 3;
        } else {
            const int rows = (nb * P.VR + 2 * yb + 2) - (na * P.VR + 2 * ya) + 1;
            v = rows * P.W1;
        }
        if (v > NV) NV = v;
    }
    return NV;
}

// nhwc: the tensors are NHWC with exactly Cin / Cout channels (csrc/hb.hip: HPlan's strides are those of 2-byte NHWC elements);
// otherwise H8 images, channel-group slices from the descriptor
bool h16_conv_plan(const otp_h16_conv_desc& d, HPlan& P, bool nhwc = false) {
    if (d.stride != 1 && d.stride != 2) return false;
    if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0) return false;
    if (d.Cin % 16 || d.Cout % 8 || (d.act != OTP_ACT_NONE && d.act != OTP_ACT_RELU)) return false;
    if (d.stride == 2 && ((d.H & 1) || (d.W & 1))) return false;
    const int S = d.stride, Ho = d.H / S, Wo = d.W / S;
    P.N = d.N; P.C = d.Cin; P.H = d.H; P.W = d.W; P.HW = d.H * d.W; P.Ho = Ho; P.Wo = Wo; P.HWo = Ho * Wo; P.Cout = d.Cout;
    P.total = d.N * P.HWo;
    const int in_gtot = d.in_gtot > 0 ? d.in_gtot : d.Cin / 8, in_goff = d.in_gtot > 0 ? d.in_goff : 0;
    const int out_gtot = d.out_gtot > 0 ? d.out_gtot : d.Cout / 8, out_goff = d.out_gtot > 0 ? d.out_goff : 0;
    const int res_gtot = d.res_gtot > 0 ? d.res_gtot : d.Cout / 8, res_goff = d.res_gtot > 0 ? d.res_goff : 0;
    if (in_goff < 0 || in_goff + d.Cin / 8 > in_gtot || out_goff < 0 || out_goff + d.Cout / 8 > out_gtot || res_goff < 0 ||
        res_goff + d.Cout / 8 > res_gtot)
        return false;
    if ((size_t)in_gtot * P.HW * 16 >= (1ull << 31) || (size_t)out_gtot * P.HWo * 16 >= (1ull << 31) || (size_t)res_gtot * P.HWo * 16 >= (1ull << 31))
        return false;
    if (nhwc) {
        if (d.in_gtot > 0 || d.out_gtot > 0 || d.res_gtot > 0) return false;
        P.in_pS = d.Cin * 2; P.in_gS = 16; P.in_imgB = P.HW * d.Cin * 2; P.in_base = 0;
        P.out_pS = P.res_pS = d.Cout * 2; P.out_gS = P.res_gS = 16; P.out_imgB = P.res_imgB = P.HWo * d.Cout * 2;
        P.out_base = P.res_base = 0;
    } else {
        P.in_pS = 16; P.in_gS = P.HW * 16; P.in_imgB = in_gtot * P.HW * 16; P.in_base = (size_t)in_goff * P.HW * 16;
        P.out_pS = 16; P.out_gS = P.HWo * 16; P.out_imgB = out_gtot * P.HWo * 16; P.out_base = (size_t)out_goff * P.HWo * 16;
        P.res_pS = 16; P.res_gS = P.HWo * 16; P.res_imgB = res_gtot * P.HWo * 16; P.res_base = (size_t)res_goff * P.HWo * 16;
    }
    P.stats = nullptr;
    P.act = d.act;
    P.post = d.out_scale > 0.f ? d.out_scale : 1.f;
    P.pre = 1.f / P.post;
    P.NTW = h16_ntw(d.Cout);
    P.nN = ((d.Cout + 15) / 16 + P.NTW - 1) / P.NTW;
    P.nChunks = d.Cin / 16;
    P.VR = d.H + 1;
    P.W1 = d.W + 1;
    const int maxrec = S == 1 ? 512 : 1024;                        // 2 / 4 pieces of 64 records per wave and plane
    // the pixel tile: 256 pixels, or 128 for launches that would leave the CUs with fewer than three workgroups each; smaller while
    // the window planes do not fit.
    bool found = false;
    int npt0 = 4;
    if (const char* e = getenv("OTPOSE_H16_NPT")) {                // development override of the largest pixel tile (1 / 2 / 4)
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) npt0 = v;
    }
    for (int npt = npt0; npt >= 1; npt >>= 1) {
        const int bm = 64 * npt;
        P.NPT = npt;
        P.nTiles = (P.total + bm - 1) / bm;
        const int NV = h16_window_records(P, bm, S);
        if (NV > maxrec) continue;
        const size_t st = (size_t)2 * NV * 16 + hwb(P.NTW) + hsred(P.NTW);   // the smallest stage: 16 channels
        if (st > 80 * 1024) continue;
        P.NV = NV;
        found = true;
        if ((long)P.nTiles * P.nN >= 3 * 256 || npt == 1) break;
    }
    if (!found) return false;
    {
        const int bm = 64 * P.NPT;
        P.nTiles = (P.total + bm - 1) / bm;
        P.NV = h16_window_records(P, bm, S);
        if (P.NV > maxrec) return false;
    }
    P.pl = P.NV * 16;
    // channels per window stage: the largest multiple of 16 dividing Cin whose planes + one weight chunk let three workgroups share a
    // CU's 160 KB (two when nothing else fits) - one HBM round trip per CK channels
    {
        const size_t wbytes = hwb(P.NTW) + hsred(P.NTW);
        int best = 0;
        for (size_t budget : {(size_t)(160 * 1024) / 3, (size_t)80 * 1024, (size_t)160 * 1024}) {
            for (int ck = 96; ck >= 16 && !best; ck -= 16)
                if (d.Cin % ck == 0 && (size_t)(ck / 8) * P.pl + wbytes <= budget) best = ck;
            if (best) break;
        }
        if (!best) return false;
        P.CK = best;
        if (const char* e = getenv("OTPOSE_H16_CK")) {             // development override
            const int v = atoi(e);
            if (v >= 16 && v <= 96 && v % 16 == 0 && d.Cin % v == 0 && (size_t)(v / 8) * P.pl + wbytes <= 160 * 1024) P.CK = v;
        }
    }
    P.tpx = (P.nTiles + 7) / 8;
    P.NIW = (P.NV + 63) / 64;
    P.mHWo = hmagic(P.HWo); P.mWo = hmagic(Wo); P.mW1 = hmagic(P.W1); P.mVR = hmagic(P.VR);
    // exactness of the magic divisions (numerator * divisor < 2^32) and 31-bit byte offsets
    if ((long)(P.HWo + 256) * P.HWo >= (1l << 32) || (long)P.HWo * Wo >= (1l << 32)) return false;
    if ((long)(d.N + 2) * P.VR * P.VR >= (1l << 32) || (long)(maxrec + d.W + 2) * P.W1 >= (1l << 32)) return false;
    if ((long)P.in_imgB * 20 >= (1l << 31)) return false;                 // a tile spans few images: per-lane offsets stay 31-bit
    if ((size_t)d.N * P.out_imgB >= (1ull << 31) || (size_t)d.N * P.res_imgB >= (1ull << 31)) return false;
    if ((size_t)P.nN * P.nChunks * hwb(P.NTW) >= (1ull << 31)) return false;
    if (P.HWo < 16) return false;
    // operands of the 24-bit multiplies of the kernel's index arithmetic (hmul)
    if (P.HW >= (1 << 24) || (long)(d.N + 2) * P.VR >= (1l << 24) || (long)(maxrec + 2 * d.W + 8) >= (1l << 24) ||
        (long)(d.N + 2) * P.VR * 2 + 2 * d.H >= (1l << 24) || P.in_pS >= (1 << 24) || P.out_pS >= (1 << 24) || P.in_gS >= (1 << 24) ||
        P.out_gS >= (1 << 24) || d.Cout / 8 >= (1 << 24))
        return false;
    // images a tile's window may touch: (n - n0) * imgB must stay below 2^31
    {
        const long span = (long)(64 * P.NPT) / P.HWo + 2;
        if (span * P.in_imgB >= (1l << 31)) return false;
    }
    return true;
}

template <int NTW, int NPT, int STRIDE, int MODE = 0>
int h16_conv_launch(const void* xs, const void* wpk, const float* shift, const void* res, void* out, const HPlan& P, hipStream_t st) {
    auto kern = h16_conv3x3_kernel<NTW, NPT, STRIDE, MODE>;
    const size_t need = (size_t)(P.CK / 8) * P.pl + hwb(NTW) + hsred(NTW);
    OTP_ALLOW_BIG_LDS(kern, need);
    hipLaunchKernelGGL(kern, dim3(8 * P.tpx * P.nN), dim3(256), need, st, static_cast<const unsigned char*>(xs),
                       static_cast<const unsigned char*>(wpk), shift, static_cast<const unsigned char*>(res),
                       static_cast<unsigned char*>(out), P);
    return otp_launch_status();
}

int h16_conv_dispatch(const void* xs, const void* wpk, const float* fs, const void* res, void* out, const HPlan& P, int stride,
                      hipStream_t st) {
#ifdef OTP_H16_BF16
#define OTP_H16_GO(NTW_, NPT_, S_)                                                                          \
    return P.stats ? h16_conv_launch<NTW_, NPT_, S_, 1>(xs, wpk, fs, res, out, P, st)                       \
                   : (res ? h16_conv_launch<NTW_, NPT_, S_, 2>(xs, wpk, fs, res, out, P, st)                \
                          : h16_conv_launch<NTW_, NPT_, S_, 0>(xs, wpk, fs, res, out, P, st))
#else
#define OTP_H16_GO(NTW_, NPT_, S_)                                                                          \
    return res ? h16_conv_launch<NTW_, NPT_, S_, 2>(xs, wpk, fs, res, out, P, st)                           \
               : h16_conv_launch<NTW_, NPT_, S_, 0>(xs, wpk, fs, res, out, P, st)
#endif
#define OTP_H16_NPT(NPT_, S_)                                  \
    if (P.NPT == NPT_) {                                       \
        if (P.NTW == 2) OTP_H16_GO(2, NPT_, S_);               \
        OTP_H16_GO(3, NPT_, S_);                               \
    }
    if (stride == 1) {
        OTP_H16_NPT(4, 1)
        OTP_H16_NPT(2, 1)
        OTP_H16_NPT(1, 1)
    } else {
        OTP_H16_NPT(4, 2)
        OTP_H16_NPT(2, 2)
        OTP_H16_NPT(1, 2)
    }
#undef OTP_H16_NPT
#undef OTP_H16_GO
    return OTP_ERR_UNSUPPORTED;
}

#ifndef OTP_H16_BF16
// packed weights: [cout block][chunk][k-step][cout tile][lane] 16-byte A fragments, lane (i16, kl): row i16 of the tile = channel
// hrow2ch(block, tile, i16), k-slot q = 4 s + kl -> tap q / 2, input channels 16 chunk + 8 (q % 2) .. + 7; the fifth k-step holds
// k-slots 16, 17 only (32 lanes per tile)
__global__ void h16_wpack_kernel(const float* __restrict__ w, const float* __restrict__ scale, u32x4* __restrict__ out, int Cout,
                                 int Cin, int NTW, int nN, int nChunks, float pre) {
    const int total = nN * nChunks * HKS * NTW * 64;
    const int WU = hwb(NTW) / 16;                                   // 16-byte units of one (cout block, chunk) image
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63;
        int r = idx >> 6;
        const int t = r % NTW; r /= NTW;
        const int s = r % HKS; r /= HKS;
        const int chunk = r % nChunks, cb = r / nChunks;
        const int cout = hrow2ch(cb * NTW * 16, t, lane & 15, NTW, Cout), kl = lane >> 4;
        const int q = 4 * s + kl, tap = q >> 1, ci0 = chunk * 16 + 8 * (q & 1);
        if (tap > 8) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = ci0 + j;
            v[j] = (cout < Cout && ci < Cin) ? w[((size_t)cout * Cin + ci) * 9 + tap] * (scale ? scale[cout] : 1.f) * pre : 0.f;
        }
        const size_t base = (size_t)(cb * nChunks + chunk) * WU;
        const size_t o = s < HKS - 1 ? base + (s * NTW + t) * 64 + lane : base + (HKS - 1) * NTW * 64 + t * 32 + lane;
        out[o] = hpack8(v);
    }
}

#endif  // !OTP_H16_BF16

// ================================================================================================================================
// 1x1 convolution on H8 records: out = act(W . x + shift (+ res)), H8 (+ H8 residual) -> H8, or -> a channel slice of fp32 NCHW
// ================================================================================================================================
// Register-resident input (csrc/pointx.hip's design, minus the split): a wave owns 32 flattened pixels and holds their Cin
// channels as B-operand fragments - lane (pixel i16 of tile h, kq) loads record (group 4 ks + kq, pixel) straight from global
// memory, 16 bytes, no conversion; the weights stream through the LDS in blocks of one cout-tile pair (32 output channels: 2 KS
// KB, LDS-DMA, double buffered); a pair's two 16 x 16 results per pixel tile are 8 consecutive channels per lane = one record.
// (csrc/hb.hip: the same kernel on bfloat16 NHWC tensors - the TransformerBlock MLP's two projections and their input gradients in
//  the training step, reached through csrc/nhwc.hip's otp_nhwc_conv_* - with the bias as a separate fp32 vector.)
struct HPw {
    const unsigned char* x;
    const unsigned char* packed;
    const unsigned char* res;
    unsigned char* out;
    const float* shift;                                           // Cout floats or NULL
    int total, HW, Cin, Cout, nblk, relu, f32out;
    // record (image n, channel group g, pixel p) of a tensor: base + n imgB + g gS + p pS bytes (see HPlan)
    size_t x_imgB, x_gS, x_pS, x_base, r_imgB, r_gS, r_pS, r_base, o_imgB, o_gS, o_pS, o_base;
    int o_tot, o_off;                                             // fp32 NCHW output: channels of the tensor / first channel
    float post;
    unsigned* rflag;
    float* stats;                                                 // bf16 build: per-tile channel sums [tiles of 128 pixels][2][Cout] of the stored values, or NULL
    otp_hbpw_epi epi;                                             // bf16 build: the MLP epilogues (csrc/hb.h); mode 0: none
};

template <int BLKB>
__device__ __forceinline__ void hpw_stage(const unsigned char* __restrict__ src, unsigned char* lds) {
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

constexpr int HPW_MAXC = 1024;            // shift table: output channels
// KS k-steps of 32 input channels (Cin padded with zero weights and masked loads); a weight block = one tile pair x KS x 1 KB each
// = 2 KS KB, rounded up to whole 4 KB passes of the 256 threads
__host__ __device__ constexpr int hpw_blkb(int KS) { return otp_hbpw_blkb(KS); }

// EPI (bf16 build; compile time, so that the plain form keeps its registers - as run-time branches the two epilogues below cost
// the MLP projections 19 registers and 41 -> 58 us): 0 plain, 1 per-tile channel statistics, 2 / 3 the MLP epilogues of csrc/hb.h
template <int KS, int EPI = 0>
__global__ __launch_bounds__(256, KS <= 4 ? 4 : (KS <= 8 ? 3 : 2)) void h16_pointwise_kernel(HPw A) {
    constexpr int BLKB = hpw_blkb(KS);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // 2 BLKB + HPW_MAXC * 4 (+ 4 waves x 2 x 32 floats: statistics)
    float* shl = reinterpret_cast<float*>(lds + 2 * BLKB);
#ifdef OTP_H16_BF16
    float* sred = shl + HPW_MAXC;                                            // [2 buffers][4 waves][2][32]
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, n16 = lane & 15;
    hpw_stage<BLKB>(A.packed, lds);
    for (int i = tid; i < HPW_MAXC; i += 256) shl[i] = (A.shift && i < A.Cout) ? A.shift[i] : 0.f;
    // the lane's two pixels (tile h: flattened pixel base + 16 h + n16), their image and in-image index
    int img[2], pix[2];
    bool pv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int P = (int)blockIdx.x * 128 + wave * 32 + 16 * h + n16;
        pv[h] = P < A.total;
        if (!pv[h]) P = A.total - 1;
        img[h] = P / A.HW;                                                     // (exact: P * HW may pass 2^32, twice per lane)
        pix[h] = P - img[h] * A.HW;
    }
    const int Gin = A.Cin >> 3;
    u32x4 X[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = 4 * ks + kq;
            const bool live = g < Gin;                                         // groups past Cin: zero operands (and zero weights)
            const u32x4* src = reinterpret_cast<const u32x4*>(A.x + A.x_base + img[h] * A.x_imgB + (live ? g : 0) * A.x_gS + pix[h] * A.x_pS);
            const u32x4 v = *src;
            X[ks][h] = live ? v : (u32x4){0u, 0u, 0u, 0u};
        }
    const float lo_clamp = A.relu ? 0.f : -__builtin_inff();
    bool bad = false;
    __syncthreads();                                   // weight block 0 and the shift table landed
#pragma unroll 1
    for (int blk = 0; blk < A.nblk; ++blk) {
        // (the last trip re-stages block 0, which nobody reads: every wave issues the same instructions on every trip)
        hpw_stage<BLKB>(A.packed + (size_t)(blk + 1 < A.nblk ? blk + 1 : 0) * BLKB, lds + ((blk + 1) & 1) * BLKB);
        asm volatile("" ::: "memory");
        const unsigned char* Pw = lds + (blk & 1) * BLKB;
        const int c8 = 32 * blk + 8 * kq;                                     // the lane's 8 output channels of this block
        const bool cl = c8 < A.Cout;
        u32x4 rq[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            rq[h] = (u32x4){0u, 0u, 0u, 0u};
            if (A.res && cl && pv[h])
                rq[h] = *reinterpret_cast<const u32x4*>(A.res + A.r_base + img[h] * A.r_imgB + (c8 >> 3) * A.r_gS + pix[h] * A.r_pS);
        }
        f32x4 acc[2][2];                                                       // [tile of the pair][pixel tile]
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const h16x8 aw = *reinterpret_cast<const h16x8*>(Pw + (m * KS + ks) * 1024 + lane * 16);
                a0 = H_MFMA(aw, __builtin_bit_cast(h16x8, X[ks][0]), a0, 0, 0, 0);
                a1 = H_MFMA(aw, __builtin_bit_cast(h16x8, X[ks][1]), a1, 0, 0, 0);
            }
            acc[m][0] = a0;
            acc[m][1] = a1;
        }
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(shl + (c8 & (HPW_MAXC - 1)));
        const f32x4 sh1 = *reinterpret_cast<const f32x4*>(shl + ((c8 + 4) & (HPW_MAXC - 1)));
#ifdef OTP_H16_BF16
        float st1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, st2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 r0 = hwiden(rq[h][0]), r1 = hwiden(rq[h][1]), r2 = hwiden(rq[h][2]), r3 = hwiden(rq[h][3]);
#ifdef OTP_H16_BF16
            float f[8] = {acc[0][h][0] + sh0[0], acc[0][h][1] + sh0[1], acc[0][h][2] + sh0[2], acc[0][h][3] + sh0[3],
                          acc[1][h][0] + sh1[0], acc[1][h][1] + sh1[1], acc[1][h][2] + sh1[2], acc[1][h][3] + sh1[3]};
#else
            float f[8] = {acc[0][h][0] * A.post + sh0[0] + r0.x, acc[0][h][1] * A.post + sh0[1] + r0.y,
                          acc[0][h][2] * A.post + sh0[2] + r1.x, acc[0][h][3] * A.post + sh0[3] + r1.y,
                          acc[1][h][0] * A.post + sh1[0] + r2.x, acc[1][h][1] * A.post + sh1[1] + r2.y,
                          acc[1][h][2] * A.post + sh1[2] + r3.x, acc[1][h][3] * A.post + sh1[3] + r3.y};
#endif
#ifdef OTP_H16_BF16
            if (A.res) {
                // csrc/nhwc.hip's contract: the residual is added to the ROUNDED result and the sum rounded again (a separate bf16 add)
                const u32x4 q = hpack8((const float(&)[8])f);
                const f32x2 w0 = hwiden(q[0]), w1 = hwiden(q[1]), w2 = hwiden(q[2]), w3 = hwiden(q[3]);
                const float g[8] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, w3.x, w3.y};
                const float rr[8] = {r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y};
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = g[e] + rr[e];
            }
#endif
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bad |= otp_out_of_range(f[e]);
                f[e] = fmaxf(f[e], lo_clamp);
            }
#ifdef OTP_H16_BF16
            if constexpr (EPI == 3) {
                // input gradient of the MLP's down-projection: times gelu'(pre-activation) and the dropout factor of the forward
                const size_t eo = A.o_base + img[h] * A.o_imgB + (c8 >> 3) * A.o_gS + pix[h] * A.o_pS;
                const bool lv = cl && pv[h];
                const u32x4 hq = lv ? *reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(A.epi.aux) + eo) : (u32x4){0u, 0u, 0u, 0u};
                const unsigned bits = lv ? A.epi.keep[eo >> 4] : 0u;
                const f32x2 h0 = hwiden(hq[0]), h1 = hwiden(hq[1]), h2 = hwiden(hq[2]), h3 = hwiden(hq[3]);
                const float hv[8] = {h0.x, h0.y, h1.x, h1.y, h2.x, h2.y, h3.x, h3.y};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = hv[e];
                    const float d = otp_phi_fast(x) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
                    f[e] = ((bits >> e) & 1u) ? f[e] * A.epi.scale * d : 0.f;
                }
            }
            if constexpr (EPI == 1) {
                // per-tile channel sums of the ROUNDED values (csrc/nhwc.hip's contract: what BatchNorm will normalise): the 16 lanes of
                // a DPP row hold 16 pixels of the lane's 8 channels
                const u32x4 qv = hpack8((const float(&)[8])f);
                const f32x2 w0 = hwiden(qv[0]), w1 = hwiden(qv[1]), w2 = hwiden(qv[2]), w3 = hwiden(qv[3]);
                const float g[8] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, w3.x, w3.y};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = pv[h] ? g[e] : 0.f;
                    st1[e] += v;
                    st2[e] += v * v;
                }
            }
#endif
            if (cl && pv[h]) {
                if (A.f32out) {
                    float* o = reinterpret_cast<float*>(A.out) + ((size_t)img[h] * A.o_tot + A.o_off + c8) * A.HW + pix[h];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (c8 + e < A.Cout) o[(size_t)e * A.HW] = f[e];
                } else {
                    const size_t eo = A.o_base + img[h] * A.o_imgB + (c8 >> 3) * A.o_gS + pix[h] * A.o_pS;
                    const u32x4 rec = hpack8(f);
                    *reinterpret_cast<u32x4*>(A.out + eo) = rec;
#ifdef OTP_H16_BF16
                    if constexpr (EPI == 2) {
                        // dropout(gelu(.)) of the ROUNDED result (what a separate pass over the stored tensor computes), one rounding
                        const f32x2 w0 = hwiden(rec[0]), w1 = hwiden(rec[1]), w2 = hwiden(rec[2]), w3 = hwiden(rec[3]);
                        const float hv[8] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, w3.x, w3.y};
                        const unsigned bits = otp_drop_keep8(eo >> 4, A.epi.s0, A.epi.s1, A.epi.thr);
                        float g[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[e] = ((bits >> e) & 1u) ? hv[e] * otp_phi_fast(hv[e]) * A.epi.scale : 0.f;
                        *reinterpret_cast<u32x4*>(static_cast<unsigned char*>(A.epi.out2) + eo) = hpack8(g);
                        A.epi.keep[eo >> 4] = (unsigned char)bits;
                    }
#endif
                }
            }
        }
#ifdef OTP_H16_BF16
        if constexpr (EPI == 1) {
            float* sr = sred + (blk & 1) * 256;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = hrow16_sum(st1[e]), b = hrow16_sum(st2[e]);
                if (n16 == 0) {
                    sr[(wave * 2 + 0) * 32 + 8 * kq + e] = a;
                    sr[(wave * 2 + 1) * 32 + 8 * kq + e] = b;
                }
            }
        }
#endif
#ifdef OTP_H16_BF16
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next block has landed, this block's stores and partial sums have left
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next block has landed (and this block's stores have left)
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef OTP_H16_BF16
        if (EPI == 1 && tid < 64) {
            // the four waves' partials in a fixed order; the buffer alternates, so the next block's writes cannot overtake these reads
            const float* sr = sred + (blk & 1) * 256;
            const int which = tid >> 5, c = tid & 31, co = 32 * blk + c;
            if (co < A.Cout)
                A.stats[((size_t)blockIdx.x * 2 + which) * A.Cout + co] =
                    sr[(0 * 2 + which) * 32 + c] + sr[(1 * 2 + which) * 32 + c] + sr[(2 * 2 + which) * 32 + c] + sr[(3 * 2 + which) * 32 + c];
        }
#endif
    }
#ifndef OTP_H16_BF16
    otp_range_report(A.rflag, bad, OTP_RANGE_H16);
#else
    (void)bad;
#endif
}

template <int KS, int EPI = 0>
int hpw_launch(const HPw& a, hipStream_t st) {
    auto kern = h16_pointwise_kernel<KS, EPI>;
#ifdef OTP_H16_BF16
    const size_t need = 2 * (size_t)hpw_blkb(KS) + HPW_MAXC * 4 + 2 * 256 * 4;
#else
    const size_t need = 2 * (size_t)hpw_blkb(KS) + HPW_MAXC * 4;
#endif
    OTP_ALLOW_BIG_LDS(kern, need);
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.total + 127) / 128)), dim3(256), need, st, a);
    return otp_launch_status();
}

#ifndef OTP_H16_BF16
// packed: nblk blocks of [tile of the pair][ks][lane] 16-byte A fragments (rows permuted: row r16 of tile m of pair p = channel
// 32 p + 8 (r16 >> 2) + 4 m + (r16 & 3)), padded to hpw_blkb(KS), then shift[HPW_MAXC] floats, then {post, 0, 0, 0}
__global__ void h16_pw_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                   unsigned char* __restrict__ packed, int Cin, int Cout, int KS, int nblk, int blkb, float pre) {
    const int units = blkb / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nblk * units) {
        const int blk = idx / units, u = idx - blk * units;
        u32x4 o = {0u, 0u, 0u, 0u};
        if (u < 2 * KS * 64) {
            const int frag = u >> 6, lane = u & 63, m = frag / KS, ks = frag - m * KS, r16 = lane & 15, kq = lane >> 4;
            const int row = 32 * blk + 8 * (r16 >> 2) + 4 * m + (r16 & 3);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 32 * ks + 8 * kq + j;
                v[j] = (row < Cout && c < Cin) ? w[(size_t)row * Cin + c] * (scale ? scale[row] : 1.f) * pre : 0.f;
            }
            o = hpack8(v);
        }
        reinterpret_cast<u32x4*>(packed)[idx] = o;
    } else if (idx < nblk * units + HPW_MAXC / 4 + 1) {
        const int q = idx - nblk * units;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (q < HPW_MAXC / 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (4 * q + i < Cout && shift) ? shift[4 * q + i] : 0.f;
        } else {
            v[0] = 1.f / pre;
        }
        reinterpret_cast<u32x4*>(packed)[idx] = (u32x4){__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]),
                                                        __builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3])};
    }
}

int hpw_ks(int Cin) {
    const int k = (Cin + 31) / 32;
    return k <= 2 ? 2 : (k <= 4 ? 4 : (k <= 8 ? 8 : (k <= 12 ? 12 : 0)));
}

// ================================================================================================================================
// stem: Conv2d(3, Cout <= 64, 3x3, stride 2, pad 1) + BN + ReLU on the frames of the fp32 clip tensor -> H8 (csrc/stem.hip)
// ================================================================================================================================
constexpr int HST_CT = 4;                 // 16-channel output tiles (Cout <= 64)
struct HSt {
    const float* in;
    const u32x4* packed;
    u32x4* out;
    int B, F, H, W, Ho, Wo, HoWo, Cout, total;
    uint32_t mWo;
    unsigned* rflag;
};

// packed: [cout tile][64 lanes] A fragments (lane (row i16, kq): k slots 8 kq .. 8 kq + 7, k = 3 tap + channel; rows of a tile pair
// permuted like everywhere in this file), then shift[64], then {post, 0, 0, 0}
__global__ void h16_stem_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                     u32x4* __restrict__ packed, int Cout) {
    __shared__ float wmax[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < Cout * 27; i += blockDim.x) m = fmaxf(m, fabsf(w[i] * (scale ? scale[i / 27] : 1.f)));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    int e = 0;
    (void)frexpf(m, &e);
    const int kx = (m > 0.f && m < 3e38f) ? min(40, max(-40, 14 - e)) : 0;
    const float pre = ldexpf(1.f, kx), post = ldexpf(1.f, -kx);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == HST_CT * 64 + 16) packed[idx] = (u32x4){__builtin_bit_cast(uint32_t, post), 0u, 0u, 0u};
    if (idx < HST_CT * 64) {
        const int t = idx >> 6, lane = idx & 63, r16 = lane & 15, kq = lane >> 4;
        const int co = 32 * (t >> 1) + 8 * (r16 >> 2) + 4 * (t & 1) + (r16 & 3);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kq + j, tap = k / 3, c = k - tap * 3;               // reference weight layout (Cout, 3, 3, 3): [co][c][dy][dx]
            v[j] = (k < 27 && co < Cout) ? w[(co * 3 + c) * 9 + tap] * (scale ? scale[co] : 1.f) * pre : 0.f;
        }
        packed[idx] = hpack8(v);
    } else if (idx < HST_CT * 64 + 16) {
        const int q = idx - HST_CT * 64;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (4 * q + i < Cout && shift) ? shift[4 * q + i] : 0.f;
        packed[idx] = (u32x4){__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]), __builtin_bit_cast(uint32_t, v[2]),
                              __builtin_bit_cast(uint32_t, v[3])};
    }
}

// a wave: NPT tiles of 16 consecutive output pixels (all frames, row-major); lane (pixel i16, kq) gathers k slots 8 kq .. + 7 of its
// pixel (the B operand); A = the weights, so a lane's accumulators of a tile pair are 8 consecutive channels of ITS pixel
template <int NPT>
__global__ __launch_bounds__(256) void h16_stem_kernel(HSt A) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    h16x8 Wf[HST_CT];
#pragma unroll
    for (int t = 0; t < HST_CT; ++t) Wf[t] = __builtin_bit_cast(h16x8, A.packed[t * 64 + lane]);
    const float* shp = reinterpret_cast<const float*>(A.packed + HST_CT * 64);
    f32x4 sh[HST_CT];                                               // shift of the lane's channels: pair tp, tile m: 32 tp + 8 kq + 4 m ..
#pragma unroll
    for (int t = 0; t < HST_CT; ++t) sh[t] = *reinterpret_cast<const f32x4*>(shp + 32 * (t >> 1) + 8 * kq + 4 * (t & 1));
    const float post = reinterpret_cast<const float*>(A.packed + HST_CT * 64 + 16)[0];
    const size_t clip = (size_t)3 * A.F * A.H * A.W;
    const otp_rsrc rin = make_rsrc(A.in, (size_t)A.B * clip * sizeof(float));
    const int G = (A.Cout + 7) >> 3;
    bool bad = false;
#pragma unroll 2
    for (int p = 0; p < NPT; ++p) {
        int px = ((int)blockIdx.x * 4 + wave) * (16 * NPT) + 16 * p + i16;
        const bool pv = px < A.total;
        if (!pv) px = A.total - 1;
        const int n = px / A.HoWo, pi = px - n * A.HoWo;                         // (exact division: px * HoWo passes 2^32 at cfg2)
        const int yo = (int)hdiv((uint32_t)pi, A.mWo), xo = pi - yo * A.Wo;
        const int b = n % A.B, f = n / A.B;                                   // frame n = f B + b (model/OTPose.py:317)
        const int base = (b * 3 * A.F + 3 * f) * A.H * A.W;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kq + j, tap = k / 3, c = k - tap * 3, dy = tap / 3, dx = tap - dy * 3;
            const int iy = 2 * yo + dy - 1, ix = 2 * xo + dx - 1;
            const bool ok = k < 27 && iy >= 0 && iy < A.H && ix >= 0 && ix < A.W;
            v[j] = bload(rin, ok ? (base + (c * A.H + iy) * A.W + ix) * 4 : -16, 0);
        }
        const h16x8 bx = __builtin_bit_cast(h16x8, hpack8(v));
#pragma unroll
        for (int tp = 0; tp < HST_CT / 2; ++tp) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            a0 = H_MFMA(Wf[2 * tp], bx, a0, 0, 0, 0);
            a1 = H_MFMA(Wf[2 * tp + 1], bx, a1, 0, 0, 0);
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = a0[r] * post + sh[2 * tp][r];
                o[4 + r] = a1[r] * post + sh[2 * tp + 1][r];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bad |= otp_out_of_range(o[e]);
                o[e] = fmaxf(o[e], 0.f);
            }
            const int g = 4 * tp + kq;
            if (pv && g < G) A.out[((size_t)n * G + g) * A.HoWo + pi] = hpack8(o);
        }
    }
    otp_range_report(A.rflag, bad, OTP_RANGE_H16);
}

// ================================================================================================================================
// element-wise passes on H8 images
// ================================================================================================================================
// fp32 NCHW channel slice -> H8 (one rounding); a thread owns one pixel of one 8-channel group (1 KB store runs per wave)
__global__ __launch_bounds__(256) void h8_pack_kernel(const float* __restrict__ in, u32x4* __restrict__ out, int N, int C, int HW,
                                                       int ctot, int coff, int gtot, int goff, unsigned* rflag) {
    const int G8 = C >> 3;
    const size_t items = (size_t)N * G8 * HW;
    bool bad = false;
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
        out[((size_t)n * gtot + goff + g) * HW + p] = hpack8(f);
    }
    otp_range_report(rflag, bad, OTP_RANGE_H16);
}

__global__ __launch_bounds__(256) void h8_unpack_kernel(const u32x4* __restrict__ in, float* __restrict__ out, int N, int C, int HW,
                                                         int gtot, int goff) {
    const int G8 = C >> 3;
    const size_t items = (size_t)N * G8 * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const size_t r = i / HW;
        const int g = (int)(r % G8), n = (int)(r / G8);
        const u32x4 v = in[((size_t)n * gtot + goff + g) * HW + p];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 a = hwiden(v[e]);
            out[((size_t)n * C + 8 * g + 2 * e) * HW + p] = a.x;
            out[((size_t)n * C + 8 * g + 2 * e + 1) * HW + p] = a.y;
        }
    }
}

// a fuse row's tail (model/HRNet.py:487-494): out = act(res + up_f0(low0) + up_f1(low1) + ...), nearest up-sampling, all H8; the
// terms are added in fp32 in that order and rounded once
struct H8Up {
    const u32x4* low[3];
    int f[3];
    int n;
};
__global__ __launch_bounds__(256) void h8_upsample_add_kernel(H8Up U, const u32x4* __restrict__ res, u32x4* __restrict__ out, int N,
                                                               int G8, int Hh, int Wh, int relu, unsigned* rflag) {
    const int HW = Hh * Wh;
    const size_t items = (size_t)N * G8 * HW;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const size_t r = i / HW;                                    // (n, g) plane
        const int y = p / Wh, x = p - y * Wh;
        const u32x4 rv = res[i];
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 a = hwiden(rv[e]);
            f[2 * e] = a.x;
            f[2 * e + 1] = a.y;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < U.n) {
                const int fk = U.f[k], Wl = Wh / fk, Hl = Hh / fk;
                const u32x4 lv = U.low[k][r * (size_t)(Hl * Wl) + (y / fk) * Wl + x / fk];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 a = hwiden(lv[e]);
                    f[2 * e] += a.x;
                    f[2 * e + 1] += a.y;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bad |= otp_out_of_range(f[e]);
            if (relu) f[e] = fmaxf(f[e], 0.f);
        }
        out[i] = hpack8(f);
    }
    otp_range_report(rflag, bad, OTP_RANGE_H16);
}

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------
#ifdef OTP_H16_TIMING
extern "C" int otp_h16_read_stamps(void* host_out, size_t bytes) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(otp_h16_stamps), bytes) == hipSuccess ? OTP_OK : OTP_ERR_LAUNCH;
}
#endif
extern "C" size_t otp_h8_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || C % 8) return 0;
    return (size_t)N * C * H * W * 2;
}

extern "C" int otp_h8_pack(const void* in, void* out, int N, int C, int H, int W, int in_ctot, int in_coff, int out_gtot,
                           int out_goff, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || in_coff < 0 || in_ctot < in_coff + C) return OTP_ERR_BAD_ARG;
    if (C % 8 || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(in) & 3)) return OTP_ERR_UNSUPPORTED;
    if (out_gtot <= 0) out_gtot = C / 8, out_goff = 0;
    if (out_goff < 0 || out_goff + C / 8 > out_gtot) return OTP_ERR_BAD_ARG;
    const size_t items = (size_t)N * (C / 8) * (H * W);
    const int grid = (int)((items + 255) / 256 > 16384 ? 16384 : (items + 255) / 256);
    hipLaunchKernelGGL(h8_pack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float*>(in),
                       static_cast<u32x4*>(out), N, C, H * W, in_ctot, in_coff, out_gtot, out_goff, otp_range_word());
    return otp_launch_status();
}

extern "C" int otp_h8_unpack(const void* in, void* out, int N, int C, int H, int W, int in_gtot, int in_goff, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return OTP_ERR_BAD_ARG;
    if (C % 8 || (reinterpret_cast<uintptr_t>(in) & 15)) return OTP_ERR_UNSUPPORTED;
    if (in_gtot <= 0) in_gtot = C / 8, in_goff = 0;
    if (in_goff < 0 || in_goff + C / 8 > in_gtot) return OTP_ERR_BAD_ARG;
    const size_t items = (size_t)N * (C / 8) * (H * W);
    const int grid = (int)((items + 255) / 256 > 16384 ? 16384 : (items + 255) / 256);
    hipLaunchKernelGGL(h8_unpack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const u32x4*>(in),
                       static_cast<float*>(out), N, C, H * W, in_gtot, in_goff);
    return otp_launch_status();
}

extern "C" int otp_h16_upsample_add(const void* const* lows, const int* factors, int nlow, const void* res, void* out, int N, int C,
                                    int Hh, int Wh, int relu, void* stream) {
    if (!lows || !factors || !res || !out || nlow < 1 || nlow > 3 || N <= 0 || C <= 0 || Hh <= 0 || Wh <= 0) return OTP_ERR_BAD_ARG;
    if (C % 8) return OTP_ERR_UNSUPPORTED;
    H8Up U{};
    U.n = nlow;
    for (int k = 0; k < nlow; ++k) {
        const int f = factors[k];
        if (!lows[k]) return OTP_ERR_BAD_ARG;
        if (f < 2 || (f & (f - 1)) || Hh % f || Wh % f) return OTP_ERR_UNSUPPORTED;
        if (reinterpret_cast<uintptr_t>(lows[k]) & 15) return OTP_ERR_UNSUPPORTED;
        U.low[k] = static_cast<const u32x4*>(lows[k]);
        U.f[k] = f;
    }
    if ((reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out)) & 15) return OTP_ERR_UNSUPPORTED;
    const size_t items = (size_t)N * (C / 8) * (Hh * Wh);
    const int grid = (int)((items + 255) / 256 > 16384 ? 16384 : (items + 255) / 256);
    hipLaunchKernelGGL(h8_upsample_add_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), U,
                       static_cast<const u32x4*>(res), static_cast<u32x4*>(out), N, C / 8, Hh, Wh, relu, otp_range_word());
    return otp_launch_status();
}

extern "C" int otp_h16_conv3x3_supported(const otp_h16_conv_desc* desc) {
    if (!desc) return 0;
    HPlan P{};
    return h16_conv_plan(*desc, P) ? 1 : 0;
}

extern "C" size_t otp_h16_conv3x3_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cin % 16) return 0;
    const int NTW = h16_ntw(Cout), nN = ((Cout + 15) / 16 + NTW - 1) / NTW;
    return (size_t)nN * (Cin / 16) * hwb(NTW);
}

extern "C" int otp_h16_conv3x3_pack_weight(const void* weight, const void* scale, void* wpacked, int Cout, int Cin, float pre,
                                           void* stream) {
    if (!weight || !wpacked || Cout <= 0 || Cin <= 0 || !(pre > 0.f)) return OTP_ERR_BAD_ARG;
    if (!otp_h16_conv3x3_weight_bytes(Cout, Cin)) return OTP_ERR_UNSUPPORTED;
    const int NTW = h16_ntw(Cout), nN = ((Cout + 15) / 16 + NTW - 1) / NTW, nChunks = Cin / 16;
    const int total = nN * nChunks * HKS * NTW * 64;
    hipLaunchKernelGGL(h16_wpack_kernel, dim3(otp_ceil_div(total, 256) > 2048 ? 2048 : otp_ceil_div(total, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float*>(weight), static_cast<const float*>(scale),
                       static_cast<u32x4*>(wpacked), Cout, Cin, NTW, nN, nChunks, pre);
    return otp_launch_status();
}

extern "C" int otp_h16_conv3x3(const void* in_h8, const void* wpacked, const void* shift, const void* res_h8, void* out_h8,
                               const otp_h16_conv_desc* desc, void* stream) {
    if (!in_h8 || !wpacked || !out_h8 || !desc) return OTP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in_h8) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out_h8) |
         reinterpret_cast<uintptr_t>(res_h8) | reinterpret_cast<uintptr_t>(shift)) & 15)
        return OTP_ERR_UNSUPPORTED;
    HPlan P{};
    if (!h16_conv_plan(*desc, P)) return OTP_ERR_UNSUPPORTED;
    P.rflag = otp_range_word();
    return h16_conv_dispatch(in_h8, wpacked, static_cast<const float*>(shift), res_h8, out_h8, P, desc->stride, static_cast<hipStream_t>(stream));
}

extern "C" int otp_h16_pointwise_supported(int Cin, int Cout) {
    return (Cin > 0 && Cin % 8 == 0 && hpw_ks(Cin) > 0 && Cout > 0 && Cout <= HPW_MAXC) ? 1 : 0;
}

extern "C" size_t otp_h16_pointwise_weight_bytes(int Cin, int Cout) {
    if (!otp_h16_pointwise_supported(Cin, Cout)) return 0;
    const int KS = hpw_ks(Cin), nblk = (Cout + 31) / 32;
    return (size_t)nblk * hpw_blkb(KS) + HPW_MAXC * 4 + 16;
}

extern "C" int otp_h16_pointwise_pack(const void* w, const void* scale, const void* shift, void* packed, int Cin, int Cout, float pre,
                                      void* stream) {
    if (!w || !packed || !(pre > 0.f)) return OTP_ERR_BAD_ARG;
    const size_t bytes = otp_h16_pointwise_weight_bytes(Cin, Cout);
    if (!bytes) return OTP_ERR_UNSUPPORTED;
    const int KS = hpw_ks(Cin), nblk = (Cout + 31) / 32, total = (int)(bytes / 16);
    hipLaunchKernelGGL(h16_pw_pack_kernel, dim3(otp_ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w), static_cast<const float*>(scale), static_cast<const float*>(shift),
                       static_cast<unsigned char*>(packed), Cin, Cout, KS, nblk, hpw_blkb(KS), pre);
    return otp_launch_status();
}

extern "C" int otp_h16_pointwise(const void* in_h8, const void* packed, const void* res_h8, void* out, int out_f32_nchw, int N, int Cin,
                                 int Cout, int HW, int in_gtot, int in_goff, int res_gtot, int res_goff, int out_tot, int out_off,
                                 int relu, float out_scale, void* stream) {
    if (!in_h8 || !packed || !out || N <= 0 || HW <= 0) return OTP_ERR_BAD_ARG;
    if (!otp_h16_pointwise_supported(Cin, Cout)) return OTP_ERR_UNSUPPORTED;
    if (!out_f32_nchw && Cout % 8) return OTP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(in_h8) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(res_h8)) & 15 ||
        reinterpret_cast<uintptr_t>(out) & (out_f32_nchw ? 3 : 15))
        return OTP_ERR_BAD_ARG;
    if (in_gtot <= 0) in_gtot = Cin / 8, in_goff = 0;
    if (res_gtot <= 0) res_gtot = Cout / 8, res_goff = 0;
    if (out_tot <= 0) out_tot = out_f32_nchw ? Cout : Cout / 8, out_off = 0;
    if (in_goff < 0 || in_goff + Cin / 8 > in_gtot || out_off < 0 || out_off + (out_f32_nchw ? Cout : Cout / 8) > out_tot ||
        (res_h8 && (Cout % 8 || res_goff < 0 || res_goff + Cout / 8 > res_gtot)))
        return OTP_ERR_BAD_ARG;
    if ((size_t)N * HW >= (1ull << 31)) return OTP_ERR_UNSUPPORTED;
    HPw a{};
    a.x = static_cast<const unsigned char*>(in_h8);
    a.packed = static_cast<const unsigned char*>(packed);
    a.res = static_cast<const unsigned char*>(res_h8);
    a.out = static_cast<unsigned char*>(out);
    a.total = N * HW, a.HW = HW, a.Cin = Cin, a.Cout = Cout, a.nblk = (Cout + 31) / 32, a.relu = relu ? 1 : 0, a.f32out = out_f32_nchw ? 1 : 0;
    const size_t hw16 = (size_t)HW * 16;
    a.x_imgB = in_gtot * hw16, a.x_gS = hw16, a.x_pS = 16, a.x_base = in_goff * hw16;
    a.r_imgB = res_gtot * hw16, a.r_gS = hw16, a.r_pS = 16, a.r_base = res_goff * hw16;
    a.o_imgB = out_tot * hw16, a.o_gS = hw16, a.o_pS = 16, a.o_base = out_off * hw16;
    a.o_tot = out_tot, a.o_off = out_off;
    a.post = out_scale > 0.f ? out_scale : 1.f;
    a.rflag = otp_range_word();
    const int KS = hpw_ks(Cin);
    a.shift = reinterpret_cast<const float*>(a.packed + (size_t)a.nblk * hpw_blkb(KS));
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (KS) {
        case 2: return hpw_launch<2>(a, st);
        case 4: return hpw_launch<4>(a, st);
        case 8: return hpw_launch<8>(a, st);
        default: return hpw_launch<12>(a, st);
    }
}

extern "C" int otp_h16_stem_supported(int B, int F, int H, int W, int Cout) {
    if (B <= 0 || F <= 0 || H < 2 || W < 2 || Cout <= 0 || Cout > 16 * HST_CT || Cout % 8) return 0;
    const int Wo = (W - 1) / 2 + 1, Ho = (H - 1) / 2 + 1;
    if ((size_t)B * 3 * F * H * W * 4 >= (1ull << 31) || (size_t)B * F * Ho * Wo >= (1ull << 31) ||
        (size_t)(Ho * Wo) * (size_t)(Ho * Wo) >= (1ull << 32))
        return 0;
    return 1;
}

extern "C" size_t otp_h16_stem_weight_bytes(int Cout) { return (Cout <= 0 || Cout > 16 * HST_CT) ? 0 : (HST_CT * 64 + 17) * 16; }

extern "C" int otp_h16_stem_pack(const void* w, const void* scale, const void* shift, void* packed, int Cout, void* stream) {
    if (!w || !packed) return OTP_ERR_BAD_ARG;
    if (!otp_h16_stem_weight_bytes(Cout)) return OTP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(h16_stem_pack_kernel, dim3(otp_ceil_div(HST_CT * 64 + 17, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(w), static_cast<const float*>(scale), static_cast<const float*>(shift),
                       static_cast<u32x4*>(packed), Cout);
    return otp_launch_status();
}

/* out_h8 (F * B, Cout, Ho, Wo) = relu(conv3x3 s2 p1 of the frames of in (B, 3 F, H, W) fp32 + shift), frame n = f B + b */
extern "C" int otp_h16_stem(const void* in, const void* packed, void* out_h8, int B, int F, int H, int W, int Cout, void* stream) {
    if (!in || !packed || !out_h8) return OTP_ERR_BAD_ARG;
    if (!otp_h16_stem_supported(B, F, H, W, Cout)) return OTP_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(in) & 3) || ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out_h8)) & 15))
        return OTP_ERR_BAD_ARG;
    HSt a{};
    a.in = static_cast<const float*>(in);
    a.packed = static_cast<const u32x4*>(packed);
    a.out = static_cast<u32x4*>(out_h8);
    a.B = B, a.F = F, a.H = H, a.W = W, a.Ho = (H - 1) / 2 + 1, a.Wo = (W - 1) / 2 + 1, a.HoWo = a.Ho * a.Wo, a.Cout = Cout;
    a.total = B * F * a.HoWo;
    a.mWo = hmagic((uint32_t)a.Wo);
    a.rflag = otp_range_word();
    constexpr int NPT = 4;
    hipLaunchKernelGGL(h16_stem_kernel<NPT>, dim3((unsigned)((a.total + 64 * NPT - 1) / (64 * NPT))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return otp_launch_status();
}

#else  // OTP_H16_BF16: csrc/hb.hip
}  // namespace

namespace {
// The plan walks every pixel tile of the launch (h16_window_records) - fine once per layer when a forward is captured into a graph, not
// per launch of an eager training step (~700 convolutions, each asking three or four times): plans are kept per shape.
struct HbKey {
    int v[6];
    bool operator<(const HbKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};
std::mutex g_hb_mutex;
std::map<HbKey, std::pair<bool, HPlan>> g_hb_plans;

bool hb_plan(const otp_nhwc_conv_desc* d, HPlan& P) {
    if (!d || d->kh != 3 || d->kw != 3 || d->pad != 1 || d->dil != 1 || (d->stride != 1 && d->stride != 2) || d->out_mode != 0) return false;
    const HbKey key{{d->N, d->Cin, d->H, d->W, d->Cout, d->stride}};
    std::lock_guard<std::mutex> lock(g_hb_mutex);
    auto it = g_hb_plans.find(key);
    if (it == g_hb_plans.end()) {
        otp_h16_conv_desc h{};
        h.N = d->N, h.Cin = d->Cin, h.H = d->H, h.W = d->W, h.Cout = d->Cout, h.stride = d->stride, h.act = OTP_ACT_NONE;
        h.out_scale = 1.f;
        HPlan Q{};
        const bool ok = h16_conv_plan(h, Q, true);
        if (g_hb_plans.size() > 4096) g_hb_plans.clear();
        it = g_hb_plans.emplace(key, std::make_pair(ok, Q)).first;
    }
    P = it->second.second;
    return it->second.first;
}
}  // namespace

// ---- 1x1 convolutions of (N, 1, T, C) sequences: the TransformerBlock MLP's projections and their input gradients ----------------
bool otp_hbpw_supported(const otp_nhwc_conv_desc* d) {
    if (!d || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->N <= 0 || d->H <= 0 || d->W <= 0) return false;
    if (d->Cin % 8 || d->Cout % 8 || d->Cout > HPW_MAXC || otp_hbpw_ks(d->Cin) == 0) return false;
    if (d->out_mode != 0 && d->out_mode != 1) return false;
    return (size_t)d->N * d->H * d->W < (1ull << 31);
}

int otp_hbpw_stats_rows(const otp_nhwc_conv_desc* d) { return otp_hbpw_supported(d) ? (d->N * d->H * d->W + 127) / 128 : 0; }

int otp_hbpw_conv(const void* x, const void* wpacked, const void* bias, const void* res, void* out, void* stats,
                  const otp_nhwc_conv_desc* d, hipStream_t stream, const otp_hbpw_epi* epi) {
    if (!otp_hbpw_supported(d)) return OTP_ERR_UNSUPPORTED;
    if (stats && d->out_mode != 0) return OTP_ERR_BAD_ARG;
    if (epi && epi->mode && (d->out_mode != 0 || stats || res || !epi->keep || (epi->mode == 1 ? !epi->out2 : !epi->aux) ||
                             ((reinterpret_cast<uintptr_t>(epi->out2) | reinterpret_cast<uintptr_t>(epi->aux)) & 15)))
        return OTP_ERR_BAD_ARG;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(res)) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & (d->out_mode ? 3 : 15)) || (reinterpret_cast<uintptr_t>(bias) & 3))
        return OTP_ERR_UNSUPPORTED;
    HPw a{};
    a.x = static_cast<const unsigned char*>(x);
    a.packed = static_cast<const unsigned char*>(wpacked);
    a.res = static_cast<const unsigned char*>(res);
    a.out = static_cast<unsigned char*>(out);
    a.shift = static_cast<const float*>(bias);
    const int HW = d->H * d->W;
    a.total = d->N * HW, a.HW = HW, a.Cin = d->Cin, a.Cout = d->Cout, a.nblk = (d->Cout + 31) / 32, a.relu = 0, a.f32out = d->out_mode ? 1 : 0;
    a.x_pS = (size_t)d->Cin * 2, a.x_gS = 16, a.x_imgB = HW * a.x_pS, a.x_base = 0;
    a.r_pS = a.o_pS = (size_t)d->Cout * 2, a.r_gS = a.o_gS = 16, a.r_imgB = a.o_imgB = HW * a.o_pS, a.r_base = a.o_base = 0;
    a.o_tot = d->Cout, a.o_off = 0;
    a.post = 1.f;
    a.rflag = nullptr;
    a.stats = static_cast<float*>(stats);
    if (epi) a.epi = *epi;
    const int ks = otp_hbpw_ks(d->Cin), mode = epi ? epi->mode : 0;
    if (mode) {                                                    // the MLP epilogues: the up-projection / the down-projection's input gradient (Cin = C)
#define OTP_HBPW_EPI(KS_)                                                     \
    if (ks == KS_) return mode == 1 ? hpw_launch<KS_, 2>(a, stream) : hpw_launch<KS_, 3>(a, stream);
        OTP_HBPW_EPI(2) OTP_HBPW_EPI(4) OTP_HBPW_EPI(5) OTP_HBPW_EPI(8)
#undef OTP_HBPW_EPI
        return OTP_ERR_UNSUPPORTED;
    }
#define OTP_HBPW_CASE(KS_) \
    if (ks == KS_) return stats ? hpw_launch<KS_, 1>(a, stream) : hpw_launch<KS_, 0>(a, stream);
    OTP_HBPW_CASE(2) OTP_HBPW_CASE(4) OTP_HBPW_CASE(5) OTP_HBPW_CASE(8) OTP_HBPW_CASE(12) OTP_HBPW_CASE(17)
#undef OTP_HBPW_CASE
    return OTP_ERR_UNSUPPORTED;
}

bool otp_hb_supported(const otp_nhwc_conv_desc* d) {
    HPlan P{};
    return hb_plan(d, P);
}

// Where the window kernel pays (MI355X, 80 frames, tools/bf16_conv_bench.py with OTPOSE_NHWC_HB=2 and =0; forward / input gradient us):
// 48 -> 48 @96x72 47.8 / 46.5 against 50.3 / 49.6, 96 -> 96 @48x36 42.1 / 43.9 against 52.4 / 49.7, 192 -> 192 @24x18 41.7 / 48.4 against
// 49.8 / 52.8, 64 -> 64 @96x72 70.7 / 70.8 against 78.2 / 75.3, 48 -> 96 stride 2 38.3 / 97.3 against 64.1 / 105.0; not at 384 channels
// (54.2 / 62.9 against 53.5 / 60.5) and not on the wide maps (64 -> 64 stride 2 @192x144: 241 / 469 against 234 / 421 - rows of more
// than 96 pixels leave 128- or 64-pixel tiles).  The 16-byte pieces of an NHWC pixel make the window's LDS-DMA gather up to six cache
// lines per instruction where the H8 engine reads one: the gain is a third of what the fp16 engine's layout gives the same kernel.
bool otp_hb_pays(const otp_nhwc_conv_desc* d) { return d && d->W <= 96 && d->Cin <= 192 && d->Cout <= 192; }

int otp_hb_stats_rows(const otp_nhwc_conv_desc* d) {
    HPlan P{};
    return hb_plan(d, P) ? P.nTiles : 0;
}

int otp_hb_conv(const void* x, const void* wpacked, const void* bias, const void* res, void* out, void* stats,
                const otp_nhwc_conv_desc* d, hipStream_t stream) {
    HPlan P{};
    if (!hb_plan(d, P)) return OTP_ERR_UNSUPPORTED;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out) |
          reinterpret_cast<uintptr_t>(res)) & 15) || (reinterpret_cast<uintptr_t>(bias) & 3))
        return OTP_ERR_UNSUPPORTED;
    P.stats = static_cast<float*>(stats);
    P.rflag = nullptr;
    return h16_conv_dispatch(x, wpacked, static_cast<const float*>(bias), res, out, P, d->stride, stream);
}
#endif
