// (inline asm: the builtins let the allocator rotate the 4-register accumulators into one another and chain them)
// Issue rate of the MFMA forms the filter kernels use: one wave per SIMD (256 threads per block, one block per CU), each
// wave a chain-free stream of 8 independent accumulators.  Prints ops/s over the whole chip and cycles per instruction
// (s_memrealtime is 100 MHz; the core clock is read off the wall time of a known-cycle-count VALU loop is not needed:
// the table reports TOP/s, which is what bench.py prices against).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, int* out) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    int sink = 0;
    if (MODE == 0) {
        i32x16 c[8] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(c[j]) : "v"(a), "v"(b));
        for (int j = 0; j < 8; ++j) sink += c[j][0];
    } else if (MODE == 4) {  // as MODE 0, every MFMA on different pseudo-random operands (data toggling: power, clocks)
        i32x16 c[8] = {};
        i32x4 ra[8], rb[8];
        unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int j = 0; j < 8; ++j)
            for (int e = 0; e < 4; ++e) { x = x * 1664525u + 1013904223u; ra[j][e] = (int)x; x = x * 1664525u + 1013904223u; rb[j][e] = (int)x; }
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(c[j]) : "v"(ra[j]), "v"(rb[j]));
        for (int j = 0; j < 8; ++j) sink += c[j][0];
    } else if (MODE == 5) {  // f16 32x32x16 on pseudo-random finite operands
        f32x16 c[8] = {};
        f16x8 ra[8], rb[8];
        unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int j = 0; j < 8; ++j)
            for (int e = 0; e < 8; ++e) { x = x * 1664525u + 1013904223u; ra[j][e] = (_Float16)((int)(x >> 20) - 2048) * (_Float16)0.001f; x = x * 1664525u + 1013904223u; rb[j][e] = (_Float16)((int)(x >> 20) - 2048) * (_Float16)0.001f; }
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(ra[j]), "v"(rb[j]));
        for (int j = 0; j < 8; ++j) sink += (int)c[j][0];
    } else if (MODE == 6) {  // f16 16x16x32 on pseudo-random finite operands, 16 independent accumulators
        f32x4 c[16] = {};
        f16x8 ra[8], rb[8];
        unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int j = 0; j < 8; ++j)
            for (int e = 0; e < 8; ++e) { x = x * 1664525u + 1013904223u; ra[j][e] = (_Float16)((int)(x >> 20) - 2048) * (_Float16)0.001f; x = x * 1664525u + 1013904223u; rb[j][e] = (_Float16)((int)(x >> 20) - 2048) * (_Float16)0.001f; }
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(ra[j]), "v"(rb[j]));
        for (int j = 0; j < 8; ++j) sink += (int)c[j][0];
    } else if (MODE == 7) {  // i8 16x16x64 on pseudo-random operands
        i32x4 c[8] = {};
        i32x4 ra[8], rb[8];
        unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int j = 0; j < 8; ++j)
            for (int e = 0; e < 4; ++e) { x = x * 1664525u + 1013904223u; ra[j][e] = (int)x; x = x * 1664525u + 1013904223u; rb[j][e] = (int)x; }
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c[j]) : "v"(ra[j]), "v"(rb[j]));
        for (int j = 0; j < 8; ++j) sink += c[j][0];
    } else if (MODE == 1) {
        i32x4 c[8] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c[j]) : "v"(a), "v"(b));
        for (int j = 0; j < 8; ++j) sink += c[j][0];
    } else if (MODE == 2) {
        f32x16 c[8] = {};
        f16x8 fa = __builtin_bit_cast(f16x8, a), fb = __builtin_bit_cast(f16x8, b);
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(fa), "v"(fb));
        for (int j = 0; j < 8; ++j) sink += (int)c[j][0];
    } else {
        f32x4 c[8] = {};
        f16x8 fa = __builtin_bit_cast(f16x8, a), fb = __builtin_bit_cast(f16x8, b);
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(fa), "v"(fb));
        for (int j = 0; j < 8; ++j) sink += (int)c[j][0];
    }
    if (sink == 0x7fffffff) out[0] = sink;
}

template <int MODE>
void run(const char* name, double ops_per_inst) {
    int* d; hipMalloc(&d, 4);
    const int iters = 100000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)blocks * 4 * iters * 8;
        printf("%-28s %8.3f ms  %7.1f TOP/s  %6.1f ns per instruction per SIMD\n", name, ms, insts * ops_per_inst / (ms * 1e-3) / 1e12,
               ms * 1e6 / (iters * 8.0));
    }
    hipFree(d);
}
int main() {
    run<0>("v_mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32);
    run<1>("v_mfma_i32_16x16x64_i8", 2.0 * 16 * 16 * 64);
    run<2>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16);
    run<3>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32);
    run<4>("i32_32x32x32_i8 random data", 2.0 * 32 * 32 * 32);
    run<5>("f32_32x32x16_f16 random data", 2.0 * 32 * 32 * 16);
    run<6>("f32_16x16x32_f16 random data", 2.0 * 16 * 16 * 32);
    run<7>("i32_16x16x64_i8 random data", 2.0 * 16 * 16 * 64);
    run<0>("v_mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32);
    return 0;
}
