// Development probe (not part of the product): what v_mfma_f32_16x16x4_f32 sustains on this chip for the
// instruction mixes the conv kernel uses.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (operands in registers).  MODE 1: + MB+PB ds_read_b32 per step.  MODE 2: + cndmask per B operand.
template <int MB, int PB, int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int steps, int cs) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)(i & 15) * 0.001f;
    __syncthreads();
    f32x4 acc[MB][PB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int p = 0; p < PB; ++p) acc[m][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[MB], b[PB];
#pragma unroll
    for (int m = 0; m < MB; ++m) a[m] = (float)(lane + m) * 1e-3f;
#pragma unroll
    for (int p = 0; p < PB; ++p) b[p] = (float)(lane - p) * 1e-3f;
    const float* wrow = lds + (lane >> 4) * 48 + (lane & 15);
    const float* irow = lds + 1024 + (lane >> 4) * cs + (lane & 15);
    unsigned mask = 0x155 + lane;
    for (int s = 0; s < steps; ++s) {
        if (MODE >= 1) {
            const int kc = (s & 3) * 4;
#pragma unroll
            for (int m = 0; m < MB; ++m) a[m] = wrow[kc * 48 + m * 16];
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                float v = irow[kc * cs + p * 16];
                if (MODE >= 2) v = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe(mask, p, 1));
                b[p] = v;
            }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int p = 0; p < PB; ++p)
                acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[p], acc[m][p], 0, 0, 0);
    }
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int p = 0; p < PB; ++p) r += acc[m][p][0] + acc[m][p][1] + acc[m][p][2] + acc[m][p][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MB, int PB, int MODE>
void run(const char* name, int wgs, int threads, int steps, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto k = probe<MB, PB, MODE>;
    hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 40 * 1024, 0, out, steps, 624);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 40 * 1024, 0, out, steps, 624);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double flop = 2.0 * 16 * 16 * 4 * MB * PB * (double)steps * wgs * (threads / 64);
    printf("%-34s wgs %5d x %3d thr  %8.3f ms  %7.1f TFLOP/s\n", name, wgs, threads, ms, flop / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 24);
    const int steps = 4000;
    run<3, 9, 0>("3x9 mfma only, 1 WG/CU", 256, 256, steps, out);
    run<3, 9, 0>("3x9 mfma only, 2 WG/CU", 512, 256, steps, out);
    run<3, 9, 0>("3x9 mfma only, 4 rounds", 2048, 256, steps / 4, out);
    run<3, 9, 1>("3x9 + lds reads, 1 WG/CU", 256, 256, steps, out);
    run<3, 9, 1>("3x9 + lds reads, 2 WG/CU", 512, 256, steps, out);
    run<3, 9, 2>("3x9 + lds + mask, 1 WG/CU", 256, 256, steps, out);
    run<3, 9, 2>("3x9 + lds + mask, 2 WG/CU", 512, 256, steps, out);
    run<3, 7, 0>("3x7 mfma only, 2 WG/CU", 512, 256, steps, out);
    run<3, 7, 2>("3x7 + lds + mask, 2 WG/CU", 512, 256, steps, out);
    run<2, 7, 2>("2x7 + lds + mask, 2 WG/CU", 512, 256, steps, out);
    run<1, 7, 2>("1x7 + lds + mask, 2 WG/CU", 512, 256, steps, out);
    run<1, 7, 2>("1x7 + lds + mask, 4 WG/CU", 1024, 256, steps, out);
    run<4, 7, 2>("4x7 + lds + mask, 2 WG/CU", 512, 256, steps, out);
    hipFree(out);
    return 0;
}
