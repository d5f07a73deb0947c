// Micro-test (MI355X): packed-fp32 VOP3P instructions whose LOW half reads the HIGH dword of a source pair (op_sel with a 1:
// what the SLP vectoriser emits for {a0 * s[1] + h[1], a1 * s[1] + h[1]}) in one wave while ANOTHER wave of the same SIMD
// issues v_mfma_f32_16x16x32_bf16.  csrc/densex.hip's epilogue lost the product term of exactly such a v_pk_fma_f32 in lanes
// 48-63 under that condition; this is the condition alone.
//   waves 0-3 of a 512-thread block (SIMD 0-3): the packed instruction in a loop, every result compared in-kernel with the
//     same arithmetic done by scalar-form instructions (all values small integers: exact)
//   waves 4-7 (the same SIMDs): MFMA pairs, optionally s_nop between pairs, or no MFMA at all (plain VALU loop)
// prints the number of wrong low / high halves per lane quarter and form
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sfma(float a, float b, float c) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// NEIGHBOUR: 0 MFMA pairs + s_nop 7, 1 MFMA back to back, 2 no MFMA (v_fma loop), 3 MFMA pairs + s_nop 7 but in waves of
// ANOTHER workgroup only (every wave of this one runs the packed instruction)
template <int FORM, int NEIGHBOUR>
__global__ __launch_bounds__(512) void k(unsigned* bad, float* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (1024 workgroups go round the 8 XCDs and the 32 CUs of each: CU c holds workgroups c, c + 256, c + 512, c + 768)
    const bool feeder = NEIGHBOUR == 3 ? ((blockIdx.x >> 8) & 1) : wave >= 4;
    if (feeder) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (__bf16)(float)(lane % 3);
            b[j] = (__bf16)(float)(lane % 5);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float f = (float)lane;
        for (int it = 0; it < iters; ++it) {
            if (NEIGHBOUR == 2) {
                f = sfma(f, 1.f, 1.f);
                f = sfma(f, 1.f, -1.f);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
                if (NEIGHBOUR != 1) asm volatile("s_nop 7");
            }
        }
        if (acc0[0] + acc1[1] + f == 12345.f) sink[threadIdx.x] = acc0[0];      // keep the chains alive
        return;
    }
    f32x2 a2 = {(float)(lane + 1), (float)(2 * lane + 3)};
    const f32x2 s2 = {7.f, 3.f}, h2 = {100.f, 200.f};
    unsigned nlo = 0, nhi = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 v;
        float wlo, whi;
        if (FORM == 0) {          // low: a.x * s.y + h.y
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]" : "=v"(v) : "v"(a2), "v"(s2), "v"(h2));
            wlo = sfma(a2.x, s2.y, h2.y), whi = sfma(a2.y, s2.y, h2.y);
        } else if (FORM == 1) {   // high half reads the LOW dwords (broadcast of element 0)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(v) : "v"(a2), "v"(s2), "v"(h2));
            wlo = sfma(a2.x, s2.x, h2.x), whi = sfma(a2.y, s2.x, h2.x);
        } else if (FORM == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(a2), "v"(s2), "v"(h2));
            wlo = sfma(a2.x, s2.x, h2.x), whi = sfma(a2.y, s2.y, h2.y);
        } else if (FORM == 3) {   // only src0's low half from the high dword
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(v) : "v"(a2), "v"(s2), "v"(h2));
            wlo = sfma(a2.y, s2.x, h2.x), whi = sfma(a2.y, s2.y, h2.y);
        } else if (FORM == 4) {   // only src1
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(v) : "v"(a2), "v"(s2), "v"(h2));
            wlo = sfma(a2.x, s2.y, h2.x), whi = sfma(a2.y, s2.y, h2.y);
        } else if (FORM == 5) {   // only src2
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(v) : "v"(a2), "v"(s2), "v"(h2));
            wlo = sfma(a2.x, s2.x, h2.y), whi = sfma(a2.y, s2.y, h2.y);
        } else if (FORM == 6) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(v) : "v"(a2), "v"(s2));
            wlo = sfma(a2.x, s2.y, 0.f), whi = sfma(a2.y, s2.y, 0.f);
        } else if (FORM == 7) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(v) : "v"(a2), "v"(s2));
            wlo = sfma(a2.x, 1.f, s2.y), whi = sfma(a2.y, 1.f, s2.y);
        } else if (FORM == 8) {   // swap the halves
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(v) : "v"(a2), "v"(a2));
            wlo = a2.y, whi = a2.x;
            v.y = whi;            // v_pk_mov's op_sel / op_sel_hi index the two SOURCES: compare the low half only
        } else {                  // both halves read the high dwords, all three sources
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1]" : "=v"(v) : "v"(a2), "v"(s2), "v"(h2));
            wlo = sfma(a2.y, s2.y, h2.y), whi = wlo;
        }
        nlo += v.x != wlo;
        nhi += v.y != whi;
        a2.x = sfma(a2.x, 1.f, 1.f);                                               // new operands every round
        a2.y = sfma(a2.y, 1.f, 2.f);
    }
    if (nlo) atomicAdd(&bad[(lane >> 4) * 2], nlo);
    if (nhi) atomicAdd(&bad[(lane >> 4) * 2 + 1], nhi);
}

template <int FORM, int NEIGHBOUR>
void run(const char* what, unsigned* d, float* sink) {
    unsigned h[8];
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL((k<FORM, NEIGHBOUR>), dim3(1024), dim3(512), 0, 0, d, sink, 4000);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-74s", what);
    for (int q = 0; q < 4; ++q) printf("  %u / %u", h[2 * q], h[2 * q + 1]);
    printf("\n");
}

int main() {
    unsigned* d;
    float* sink;
    hipMalloc(&d, 64);
    hipMalloc(&sink, 4096);
    printf("%-74s  wrong low / high halves in lanes 0-15, 16-31, 32-47, 48-63\n", "instruction, neighbour wave on the same SIMD");
    run<0, 0>("v_pk_fma_f32 op_sel:[0,1,1], MFMA pairs + s_nop 7", d, sink);
    run<0, 1>("v_pk_fma_f32 op_sel:[0,1,1], MFMA back to back", d, sink);
    run<0, 2>("v_pk_fma_f32 op_sel:[0,1,1], no MFMA (v_fma_f32 loop)", d, sink);
    run<0, 3>("v_pk_fma_f32 op_sel:[0,1,1], MFMA only in OTHER workgroups", d, sink);
    run<1, 0>("v_pk_fma_f32 op_sel_hi:[1,0,0], MFMA pairs + s_nop 7", d, sink);
    run<2, 0>("v_pk_fma_f32 (no op_sel), MFMA pairs + s_nop 7", d, sink);
    run<3, 0>("v_pk_fma_f32 op_sel:[1,0,0], MFMA pairs + s_nop 7", d, sink);
    run<4, 0>("v_pk_fma_f32 op_sel:[0,1,0], MFMA pairs + s_nop 7", d, sink);
    run<5, 0>("v_pk_fma_f32 op_sel:[0,0,1], MFMA pairs + s_nop 7", d, sink);
    run<9, 0>("v_pk_fma_f32 op_sel:[1,1,1], MFMA pairs + s_nop 7", d, sink);
    run<6, 0>("v_pk_mul_f32 op_sel:[0,1], MFMA pairs + s_nop 7", d, sink);
    run<7, 0>("v_pk_add_f32 op_sel:[0,1], MFMA pairs + s_nop 7", d, sink);
    run<8, 0>("v_pk_mov_b32 op_sel:[1,0] op_sel_hi:[0,1], MFMA pairs + s_nop 7", d, sink);
    return 0;
}
