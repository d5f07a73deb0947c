// Micro-test (MI355X): are dependent chains of v_mfma_f32_16x16x32_bf16 exact when the issuing wave sits out wait states between
// them?  Two accumulators fed alternately (the shape of csrc/densex.hip's dx_project: acc0, acc1, gap, acc0, acc1, gap, ...),
// operands small integers (every product and sum exact in fp32), result compared with the closed form on the host.
//   gap variants: none, s_nop 1, s_nop 3, s_nop 7, 4 x s_nop 15;   waves per SIMD 1 / 2 / 4 (blocks of 256 / 512 / 1024 threads)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dep_gap tools/micro/mfma_dep_gap.hip && /tmp/mfma_dep_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int GAP>
__device__ __forceinline__ void gap() {
    __builtin_amdgcn_sched_barrier(0);
    if (GAP == 1) asm volatile("s_nop 1");
    if (GAP == 2) asm volatile("s_nop 3");
    if (GAP == 3) asm volatile("s_nop 7");
    if (GAP == 4) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    __builtin_amdgcn_sched_barrier(0);
}

// lane (r = lane % 16, kq = lane / 16): A row r holds the value (r % 5) + 1 in all 32 k; B column c holds (c % 3) + 1
// => every MFMA adds 32 * ((r % 5) + 1) * ((c % 3) + 1) to D[r][c]; D register i of lane (c, g) is row 4 g + i, column c
template <int GAP>
__global__ void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const float av = (float)((lane & 15) % 5 + 1), bv = (float)((lane & 15) % 3 + 1);
    bf16x8 a, b, a2;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)av;
        b[j] = (__bf16)bv;
        a2[j] = (__bf16)(av + 1.f);                                  // a second A operand: the chain alternates operands like hi / lo
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
        gap<GAP>();
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, acc1, 0, 0, 0);
        gap<GAP>();
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
        gap<GAP>();
    }
    float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    for (int i = 0; i < 4; ++i) {
        o[i] = acc0[i];
        o[4 + i] = acc1[i];
    }
}

template <int GAP>
long run(int threads, int blocks, int iters, float* d, std::vector<float>& h) {
    hipMemset(d, 0, h.size() * 4);
    hipLaunchKernelGGL(k<GAP>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, (size_t)blocks * threads * 8 * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    int rows[16] = {0};
    for (long t = 0; t < (long)blocks * threads; ++t) {
        const int lane = (int)(t & 63), c = lane & 15, g = lane >> 4;
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            const float ar = (float)(r % 5 + 1), bc = (float)(c % 3 + 1);
            const float want = (float)iters * 32.f * bc * (2.f * ar + (ar + 1.f));
            for (int w = 0; w < 2; ++w)
                if (h[t * 8 + 4 * w + i] != want) {
                    ++bad;
                    ++rows[r];
                }
        }
    }
    if (bad) {
        printf("      wrong rows:");
        for (int r = 0; r < 16; ++r)
            if (rows[r]) printf(" %d(%d)", r, rows[r]);
        printf("\n");
    }
    return bad;
}

int main() {
    const int blocks = 2048, iters = 40;
    float* d;
    std::vector<float> h((size_t)blocks * 1024 * 8);
    hipMalloc(&d, h.size() * 4);
    const char* names[5] = {"no gap", "s_nop 1", "s_nop 3", "s_nop 7", "4 x s_nop 15"};
    for (int threads : {256, 512, 1024})
        for (int gi = 0; gi < 5; ++gi) {
            long bad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                switch (gi) {
                    case 0: bad += run<0>(threads, blocks, iters, d, h); break;
                    case 1: bad += run<1>(threads, blocks, iters, d, h); break;
                    case 2: bad += run<2>(threads, blocks, iters, d, h); break;
                    case 3: bad += run<3>(threads, blocks, iters, d, h); break;
                    default: bad += run<4>(threads, blocks, iters, d, h); break;
                }
            }
            printf("%4d threads per block, %-12s: %ld wrong values in 3 launches of %d blocks\n", threads, names[gi], bad, blocks);
        }
    return 0;
}
