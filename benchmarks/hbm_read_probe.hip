// hbm_read_probe.hip — what a read-only stream reaches on this device, with the scan's own access shape and
// none of its arithmetic: each wave walks 12-KiB tiles (12 x global_load_dwordx4 per lane in flight,
// non-temporal), adds the words up and writes one float.  The number to read the scan's 6.6-6.8 TB/s against.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_read_probe hbm_read_probe.hip && ./hbm_read_probe [GiB] [blocks_per_cu]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NLOADS, bool NT>
__global__ void __launch_bounds__(256) read_kernel(const f32x4* __restrict__ src, uint64_t ntiles, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    float acc = 0.0f;
    for (uint64_t t = gw; t < ntiles; t += nw) {
        const f32x4* p = src + t * (NLOADS * 64) + lane;
        f32x4 v[NLOADS];
#pragma unroll
        for (int i = 0; i < NLOADS; ++i) v[i] = NT ? __builtin_nontemporal_load(p + i * 64) : p[i * 64];
#pragma unroll
        for (int i = 0; i < NLOADS; ++i) acc += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    if (acc == 12345.678f) out[gw] = acc;  // keep the loads alive without a store per wave
    if (lane == 0 && gw == 0) out[0] = acc;
}

template <int NLOADS, bool NT>
static double run(const f32x4* d, size_t bytes, int blocks, float* d_out, int reps) {
    const uint64_t ntiles = bytes / (NLOADS * 64 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((read_kernel<NLOADS, NT>), dim3(blocks), dim3(256), 0, 0, d, ntiles, d_out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((read_kernel<NLOADS, NT>), dim3(blocks), dim3(256), 0, 0, d, ntiles, d_out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ntiles * NLOADS * 64 * 16 / (ms * 1e-3 / reps) / 1e12;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 14.3;  // 15.36 GB = the scan's corpus
    const size_t bytes = (size_t)(gib * (1ull << 30)) / (144 * 64 * 16) * (144 * 64 * 16);  // a multiple of every tile size tried
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    f32x4* d = nullptr;
    float* d_out = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&d_out, 1 << 22) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, bytes);
    for (int bpc : {1, 2, 4, 5, 8}) {
        const int blocks = cus * bpc;
        printf("%.2f GB, %d blocks/CU: 12 loads nt %.3f TB/s, 12 loads cached %.3f TB/s, 6 loads nt %.3f TB/s, 24 loads nt %.3f TB/s, "
               "36 loads nt %.3f TB/s, 48 loads nt %.3f TB/s\n",
               bytes / 1e9, bpc, run<12, true>(d, bytes, blocks, d_out, 10), run<12, false>(d, bytes, blocks, d_out, 10),
               run<6, true>(d, bytes, blocks, d_out, 10), run<24, true>(d, bytes, blocks, d_out, 10),
               run<36, true>(d, bytes, blocks, d_out, 10), run<48, true>(d, bytes, blocks, d_out, 10));
    }
    hipFree(d);
    hipFree(d_out);
    return 0;
}
