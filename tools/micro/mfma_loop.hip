// Micro-benchmark (MI355X): cycles per v_mfma_f32_16x16x32_bf16 of the convs.hip inner loop shape, one or two waves per SIMD.
//   variant 0: MFMAs only (operands in registers)                          -> the 16-cycle floor
//   variant 1: + 2 ds_read_b128 per 9 MFMAs, next-block prefetch (the convs.hip block)
//   variant 2: as 1, 3 independent accumulators per block instead of 3 chains of 3
//   variant 3: as 1 with the operand pair shared by 18 MFMAs (two pixel tiles per B fragment read pair: half the reads)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_loop tools/micro/mfma_loop.hip && /tmp/mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u + i;
    __syncthreads();
    f32x4 acc[3][4];
    for (int t = 0; t < 3; ++t) for (int p = 0; p < 4; ++p) acc[t][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ah[3], al[3], bh[2], bl[2];
    for (int t = 0; t < 3; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(lds + t * 2048 + lane * 16);
        al[t] = *reinterpret_cast<const bf16x8*>(lds + t * 2048 + 1024 + lane * 16);
    }
    bh[0] = *reinterpret_cast<const bf16x8*>(lds + 8192 + lane * 16);
    bl[0] = *reinterpret_cast<const bf16x8*>(lds + 16384 + lane * 16);
    bh[1] = bh[0]; bl[1] = bl[0];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int cur = (s * 4 + p) & 1;
                if (V >= 1) {
                    const unsigned char* b = lds + 8192 + ((s * 4 + p + 1 + it) & 15) * 1040 + lane * 16;
                    bh[cur ^ 1] = *reinterpret_cast<const bf16x8*>(b);
                    bl[cur ^ 1] = *reinterpret_cast<const bf16x8*>(b + 8192 * 2);
                    if (p == 3) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            ah[t] = *reinterpret_cast<const bf16x8*>(lds + 32768 + ((s + it) & 3) * 6144 + t * 2048 + lane * 16);
                            al[t] = *reinterpret_cast<const bf16x8*>(lds + 32768 + ((s + it) & 3) * 6144 + t * 2048 + 1024 + lane * 16);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (V == 2) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bh[cur], acc[t][p], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t][(p + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bl[cur], acc[t][(p + 1) & 3], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t][(p + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh[cur], acc[t][(p + 2) & 3], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bh[cur], acc[t][p], 0, 0, 0);
                        acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bl[cur], acc[t][p], 0, 0, 0);
                        acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh[cur], acc[t][p], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 3; ++t) for (int p = 0; p < 4; ++p) s += acc[t][p][0] + acc[t][p][1] + acc[t][p][2] + acc[t][p][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// variant 3: B fragments prefetched TWO blocks ahead (ring of three), A fragments of the next k-step two blocks ahead
template <int MODE>
__global__ __launch_bounds__(512, 2) void k3(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u + i;
    __syncthreads();
    f32x4 acc[3][4];
    for (int t = 0; t < 3; ++t) for (int p = 0; p < 4; ++p) acc[t][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ah[2][3], al[2][3], bh[3], bl[3];
    for (int t = 0; t < 3; ++t) {
        ah[0][t] = ah[1][t] = *reinterpret_cast<const bf16x8*>(lds + t * 2048 + lane * 16);
        al[0][t] = al[1][t] = *reinterpret_cast<const bf16x8*>(lds + t * 2048 + 1024 + lane * 16);
    }
    for (int i = 0; i < 3; ++i) {
        bh[i] = *reinterpret_cast<const bf16x8*>(lds + 8192 + lane * 16);
        bl[i] = *reinterpret_cast<const bf16x8*>(lds + 16384 + lane * 16);
    }
    __syncthreads();
    int baddr[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) baddr[i] = 8192 + ((i * 7 + (int)cyc[0]) & 15) * 1040 + lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 3) {            // 3 x 20 blocks = 60: a multiple of the ring length
#pragma unroll
        for (int blk = 0; blk < 60; ++blk) {
            const int s = (blk / 4) % 5, p = blk & 3, cur = blk % 3, nxt = (blk + 2) % 3, sa = (blk / 4) & 1;
            const unsigned char* b = MODE >= 1 ? lds + baddr[(blk + 2) % 20] : lds + 8192 + ((blk + 2 + it) & 15) * 1040 + lane * 16;
            bh[nxt] = *reinterpret_cast<const bf16x8*>(b);
            bl[nxt] = *reinterpret_cast<const bf16x8*>(b + 8192 * 2);
            if (p == 2) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    ah[sa ^ 1][t] = *reinterpret_cast<const bf16x8*>(lds + 32768 + ((s + it) & 3) * 6144 + t * 2048 + lane * 16);
                    al[sa ^ 1][t] = *reinterpret_cast<const bf16x8*>(lds + 32768 + ((s + it) & 3) * 6144 + t * 2048 + 1024 + lane * 16);
                }
            }
            if (MODE < 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[sa][t], bh[cur], acc[t][p], 0, 0, 0);
                acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[sa][t], bl[cur], acc[t][p], 0, 0, 0);
                acc[t][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[sa][t], bh[cur], acc[t][p], 0, 0, 0);
            }
            if (MODE == 2) {
                // one LDS read, then MFMAs, repeated: the reads of a block spread between its MFMAs
                if (p == 2) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 3; ++t) for (int p = 0; p < 4; ++p) s += acc[t][p][0] + acc[t][p][1] + acc[t][p][2] + acc[t][p][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run3(int threads, const char* name) {
    const int blocks = 256, iters = 21;
    float* o; unsigned long long* c;
    hipMalloc(&o, blocks * 512 * 4); hipMalloc(&c, blocks * 8 * 8);
    hipMemset(c, 0, blocks * 8 * 8);
    hipFuncSetAttribute((const void*)k3<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k3<MODE>, dim3(blocks), dim3(threads), 128 * 1024, 0, o, c, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), c, blocks * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (auto x : h) if (x) v.push_back((double)x / (iters * 180.0));
    std::sort(v.begin(), v.end());
    printf("%-58s %d waves/SIMD: %.2f cycles per MFMA per wave (median), %.2f per SIMD\n",
           name, threads / 256, v[v.size() / 2], v[v.size() / 2] / (threads / 256));
    hipFree(o); hipFree(c);
}

template <int V>
void run(const char* name, int threads) {
    const int blocks = 256, iters = 20;
    float* o; unsigned long long* c;
    hipMalloc(&o, blocks * 512 * 4); hipMalloc(&c, blocks * 8 * 8);
    hipMemset(c, 0, blocks * 8 * 8);
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 128 * 1024, 0, o, c, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), c, blocks * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (auto x : h) if (x) v.push_back((double)x / (iters * 180.0));
    std::sort(v.begin(), v.end());
    printf("%-58s %d waves/SIMD: %.2f cycles per MFMA per wave (median), %.2f per SIMD\n", name, threads / 256, v[v.size() / 2],
           v[v.size() / 2] / (threads / 256));
    hipFree(o); hipFree(c);
}

int main() {
    for (int th : {256, 512}) {
        run<0>("MFMA only", th);
        run<1>("+ 2 ds_read_b128 / 9 MFMA, chains of 3 (convs.hip)", th);
        run<2>("+ 2 ds_read_b128 / 9 MFMA, independent accumulators", th);
        run3<0>(th, "prefetch TWO blocks ahead");
        run3<1>(th, "  + addresses precomputed (no VALU in the loop)");
        run3<2>(th, "  + reads interleaved between the MFMAs");
    }
    return 0;
}
