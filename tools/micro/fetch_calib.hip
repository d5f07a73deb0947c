// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the kernels of this repository use
// (MI355X_MICROARCH.md, HBM section: 16 B per lane reads HALF; "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern").  The modulated-DCN operator (csrc/mdcn.hip) streams its offsets / masks as 27 coalesced
// DWORD streams per thread - this tool reads a known byte count with exactly that pattern (and with 8- and 16-byte lanes for
// comparison) so that `bytes / FETCH_SIZE` can be applied to the operator's own counters.
// Build + run on the GPU box (see tools/r05_dcn_traffic.sh):
//   hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -o t --output-format csv -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// `streams` arrays of n elements, element i of every stream read by thread i: the wave-instruction covers 64 consecutive elements
template <typename T>
__global__ __launch_bounds__(256) void calib_read(const T* __restrict__ in, T* __restrict__ out, size_t n, int streams) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T s = in[i];
    for (int k = 1; k < streams; ++k) s += in[(size_t)k * n + i];
    out[i] = s;
}

int main() {
    const size_t total = 27ull * 2 * 1024 * 1024 * 4;               // 216 MiB read per launch: the DCN launch's footprint at cfg2
    const int streams = 27;
    void *in, *out;
    hipMalloc(&in, total);
    hipMalloc(&out, total / streams);
    hipMemset(in, 0, total);
    for (int rep = 0; rep < 3; ++rep) {
        size_t n = total / streams / 4;
        hipLaunchKernelGGL(calib_read<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const float*)in, (float*)out, n, streams);
        n = total / streams / 8;
        hipLaunchKernelGGL(calib_read<f32x2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const f32x2*)in, (f32x2*)out, n, streams);
        n = total / streams / 16;
        hipLaunchKernelGGL(calib_read<f32x4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const f32x4*)in, (f32x4*)out, n, streams);
    }
    hipDeviceSynchronize();
    printf("fetch_calib: each launch reads %zu bytes (27 streams) and writes %zu\n", total, total / streams);
    return 0;
}
