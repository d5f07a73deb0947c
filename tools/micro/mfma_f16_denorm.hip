// Does v_mfma_f32_16x16x32_f16 keep SUBNORMAL f16 inputs (|x| < 6.1e-5) or flush them?  (round 4: a two-piece fp16 split of an
// fp32 operand - hi = f16(a), lo = f16(a - hi) - carries 22 bits only if the small `lo` pieces survive the matrix unit.)
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float a, float b, float* out) {
    h8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (_Float16)0.f; B[j] = (_Float16)0.f; }
    // lane l: A[row l&15][k = 8 (l>>4) + j], B[k][col l&15]: put a at k = 0 of every row, b at k = 0 of every column
    if ((threadIdx.x >> 4) == 0) { A[0] = (_Float16)a; B[0] = (_Float16)b; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)(_Float16)a; out[2] = (float)(_Float16)b; }
}
int main() {
    float* d;
    hipMalloc(&d, 12);
    const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {1.0e-6f, 1.0f}, {6.0e-8f, 1.0f}, {3.0e-5f, 3.0e-5f}, {1.0f, 2.0e-7f},
                              {1.2e-4f, 0.3f}, {5.96e-8f, 4.0f}};
    for (auto& cs : cases) {
        k<<<1, 64>>>(cs[0], cs[1], d);
        float h[3];
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a %.4e (f16 %.6e)  b %.4e (f16 %.6e)  mfma %.6e  exact %.6e  %s\n", cs[0], h[1], cs[1], h[2], h[0],
               (double)h[1] * (double)h[2], std::fabs(h[0] - h[1] * h[2]) <= 1e-12 + 1e-6 * std::fabs(h[1] * h[2]) ? "kept" : "FLUSHED / changed");
    }
    return 0;
}
