// Micro-benchmark: cycles per ds_read_b128 wave-instruction for several lane -> address maps (one wave, one CU).
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_b128.hip -o /tmp/lds_b128 && /tmp/lds_b128
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const int* __restrict__ addr, unsigned long long* out, int reps, int write) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 16; i += blockDim.x) reinterpret_cast<u32x4*>(lds)[i] = (u32x4){1u, 2u, 3u, 4u};
    __syncthreads();
    const int a = addr[threadIdx.x & 63] + (threadIdx.x >> 6) * 1024 * 0;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (write) {
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(lds + ((a + j * 16384) & 0x1ffff)) = acc;
        } else {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4*>(lds + ((a + j * 16384) & 0x1ffff));
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc[0] == 0x12345678u) out[1] = acc[1];
}

int main() {
    int* d_addr; unsigned long long* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 16);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char* names[] = {"linear 16*l", "80*i16 + 16*kl (one record)", "80*i16 + 16*(kl&1) + 5920*(kl>>1) (row tap)",
                           "80*i16 + 16*(kl&1) + 80*(kl>>1) (column tap)", "80*i16 + 4096*kl (same banks per kl)",
                           "64*i16 + 16*kl", "320*l (write pattern, 4 pixels apart)", "80*l", "48*i16 + 16*(kl&1)+ 48*(kl>>1)",
                           "16*i16 + 256*kl", "16*(l&7) + 128*... 8-lane groups: 16*(l&7)+4096*(l>>3)", "32*l", "80*i16 + 16*(kl&1) + 160*(kl>>1)"};
    for (int pat = 0; pat < 13; ++pat) {
        int h[64];
        for (int l = 0; l < 64; ++l) {
            const int i16 = l & 15, kl = l >> 4;
            switch (pat) {
                case 0: h[l] = 16 * l; break;
                case 1: h[l] = 80 * i16 + 16 * kl; break;
                case 2: h[l] = 80 * i16 + 16 * (kl & 1) + 5920 * (kl >> 1); break;
                case 3: h[l] = 80 * i16 + 16 * (kl & 1) + 80 * (kl >> 1); break;
                case 4: h[l] = 80 * i16 + 4096 * kl; break;
                case 5: h[l] = 64 * i16 + 16 * kl; break;
                case 6: h[l] = 320 * l; break;
                case 7: h[l] = 80 * l; break;
                case 8: h[l] = 48 * i16 + 16 * (kl & 1) + 48 * (kl >> 1); break;
                case 9: h[l] = 16 * i16 + 256 * kl; break;
                case 10: h[l] = 16 * (l & 7) + 4096 * (l >> 3); break;
                case 11: h[l] = 32 * l; break;
                default: h[l] = 80 * i16 + 16 * (kl & 1) + 160 * (kl >> 1); break;
            }
        }
        hipMemcpy(d_addr, h, sizeof(h), hipMemcpyHostToDevice);
        for (int wr = 0; wr < 2; ++wr) {
            unsigned long long o[2] = {0, 0};
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024, 0, d_addr, d_out, 2000, wr);
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024, 0, d_addr, d_out, 2000, wr);
            hipMemcpy(o, d_out, 16, hipMemcpyDeviceToHost);
            printf("%-64s %s %6.2f cycles / wave-instr (4 waves issuing)\n", names[pat], wr ? "write" : "read ", (double)o[0] / (2000.0 * 8 * 4));
        }
    }
    return 0;
}
