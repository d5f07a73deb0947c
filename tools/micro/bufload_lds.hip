// Micro check (MI355X): `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor).
//  (1) lane l of a wave writes LDS[base + 16 l .. + 15]  (lane-linear destination, per-lane source offset)
//  (2) a source offset at or past the descriptor's size writes ZEROS to the lane's LDS slot (hardware range check)
//  (3) soffset (scalar) selects a plane without per-lane arithmetic
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/bufload_lds tools/micro/bufload_lds.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out, int plane_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)nbytes, 0x00020000);
    // wave w fills LDS [1024 w, 1024 w + 1024): even lanes read record (63 - lane) of plane w, odd lanes an out-of-range offset
    int voff = (lane & 1) ? (int)0x7ffffff0 : (63 - lane) * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 1024 * wave), 16, voff,
                                         wave * plane_bytes, 0, 0);
    
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = reinterpret_cast<unsigned*>(lds)[i];
}

int main() {
    const int planes = 4, plane_bytes = 1024;
    std::vector<unsigned> h(planes * plane_bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (unsigned)i;
    unsigned *d, *o;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 256>>>(d, (unsigned)(h.size() * 4), o, plane_bytes);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                unsigned got = r[w * 256 + l * 4 + e];
                unsigned want = (l & 1) ? 0u : 0x1000000u + (unsigned)(w * 256 + (63 - l) * 4 + e);
                if (got != want) { if (bad < 8) printf("wave %d lane %d e %d: got %08x want %08x\n", w, l, e, got, want); ++bad; }
            }
    printf("bufload_lds: %s (%d mismatches)\n", bad ? "FAIL" : "OK: lane-linear destination, OOB lanes write zeros, soffset selects the plane", bad);
    return bad != 0;
}
