// hbm_write_probe.hip — what a store-only stream reaches on this device, to read the dense layers' epilogues
// against (the QKV layer writes 302 MB per call in split-f16 form).  Two shapes:
//   contiguous: every wave writes whole 1-KiB tiles (16 B per lane), grid-stride;
//   epilogue:   the wide GEMM's own pattern — a wave owns 64 rows x 384 B (three 128-B lines per row) of a
//               [rows][pitch] matrix, rows `pitch` bytes apart (4,608 B for N = 1,152), written as 16 B per lane,
//               24 lanes per row.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_write_probe hbm_write_probe.hip && ./hbm_write_probe [MB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) write_contiguous(f32x4* __restrict__ dst, uint64_t ntiles, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    const f32x4 v = {seed, seed + lane, seed, seed};
    for (uint64_t t = gw; t < ntiles; t += nw) {
        if (NT) __builtin_nontemporal_store(v, dst + t * 64 + lane);
        else dst[t * 64 + lane] = v;
    }
}

// rows x (pitch bytes); a wave takes a 64-row x 384-B patch; patches tile the matrix column-first
template <bool NT>
__global__ void __launch_bounds__(256) write_epilogue(char* __restrict__ dst, uint64_t rows, uint32_t pitch, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    const uint32_t patches_per_row = pitch / 384;
    const uint64_t npatches = (rows / 64) * patches_per_row;
    const f32x4 v = {seed, seed + lane, seed, seed};
    for (uint64_t p = gw; p < npatches; p += nw) {
        const uint64_t r0 = (p / patches_per_row) * 64;
        const uint32_t c0 = (uint32_t)(p % patches_per_row) * 384;
        // 64 rows x 24 lanes of 16 B = 1,536 stores of 16 B = 24 per lane
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const uint32_t flat = i * 64 + lane;      // 0 .. 1535
            const uint32_t r = flat / 24, c = (flat % 24) * 16;
            f32x4* q = reinterpret_cast<f32x4*>(dst + (r0 + r) * pitch + c0 + c);
            if (NT) __builtin_nontemporal_store(v, q);
            else *q = v;
        }
    }
}

// the same patch written as half lines: an instruction covers 16 rows x 64 B (4 lanes x 16 B per row), the hi half of
// a 128-B line, and the next instruction the lo half — what a register-to-register transpose (no LDS pass) could emit
template <bool NT>
__global__ void __launch_bounds__(256) write_half_lines(char* __restrict__ dst, uint64_t rows, uint32_t pitch, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    const uint32_t patches_per_row = pitch / 384;
    const uint64_t npatches = (rows / 64) * patches_per_row;
    const f32x4 v = {seed, seed + lane, seed, seed};
    const uint32_t r16 = lane & 15, g = lane >> 4;
    for (uint64_t p = gw; p < npatches; p += nw) {
        const uint64_t r0 = (p / patches_per_row) * 64;
        const uint32_t c0 = (uint32_t)(p % patches_per_row) * 384;
#pragma unroll
        for (int i = 0; i < 4; ++i)        // strips of 16 rows
#pragma unroll
            for (int l = 0; l < 3; ++l)    // the three lines of a row
#pragma unroll
                for (int h = 0; h < 2; ++h) {  // hi half, lo half
                    f32x4* q = reinterpret_cast<f32x4*>(dst + (r0 + 16 * i + r16) * pitch + c0 + 128 * l + 64 * h + 16 * g);
                    if (NT) __builtin_nontemporal_store(v, q);
                    else *q = v;
                }
    }
}

template <typename F>
static double timed(F launch, size_t bytes, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)bytes / (ms * 1e-3 / reps) / 1e12;
}

int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 302.0;  // QKV output of one 256 x 256 batch
    const uint32_t pitch = 4608;
    const uint64_t rows = (uint64_t)(mb * 1e6 / pitch) / 64 * 64;
    const size_t bytes = rows * pitch;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    char* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, bytes);
    for (int bpc : {1, 2, 4, 8}) {
        const int blocks = cus * bpc;
        const double a = timed([&] { hipLaunchKernelGGL(write_contiguous<false>, dim3(blocks), dim3(256), 0, 0, (f32x4*)d, bytes / 1024, 1.0f); }, bytes, 20);
        const double b = timed([&] { hipLaunchKernelGGL(write_contiguous<true>, dim3(blocks), dim3(256), 0, 0, (f32x4*)d, bytes / 1024, 1.0f); }, bytes, 20);
        const double c = timed([&] { hipLaunchKernelGGL(write_epilogue<false>, dim3(blocks), dim3(256), 0, 0, d, rows, pitch, 1.0f); }, bytes, 20);
        const double e = timed([&] { hipLaunchKernelGGL(write_epilogue<true>, dim3(blocks), dim3(256), 0, 0, d, rows, pitch, 1.0f); }, bytes, 20);
        const double f = timed([&] { hipLaunchKernelGGL(write_half_lines<false>, dim3(blocks), dim3(256), 0, 0, d, rows, pitch, 1.0f); }, bytes, 20);
        const double h = timed([&] { hipLaunchKernelGGL(write_half_lines<true>, dim3(blocks), dim3(256), 0, 0, d, rows, pitch, 1.0f); }, bytes, 20);
        printf("%.1f MB, %d blocks/CU: half-line instructions (16 rows x 64 B) %.3f TB/s (nt %.3f)\n", bytes / 1e6, bpc, f, h);
        printf("%.1f MB, %d blocks/CU: contiguous %.3f TB/s (nt %.3f), epilogue-shaped %.3f TB/s (nt %.3f)  -> %.1f us per %.0f MB at the best\n",
               bytes / 1e6, bpc, a, b, c, e, bytes / 1e6 / (a > b ? (a > c ? (a > e ? a : e) : (c > e ? c : e)) : (b > c ? (b > e ? b : e) : (c > e ? c : e))), bytes / 1e6);
    }
    hipFree(d);
    return 0;
}
