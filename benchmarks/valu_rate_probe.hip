// Issue rate of the f32 VALU forms the epilogues are made of: v_fma_f32 against v_pk_fma_f32 (two fmas per lane), as
// independent streams (8 accumulators per lane) and as ONE dependent chain, plus v_exp_f32 (transcendental) and
// ds_write_b8.  256 threads per block (one wave per SIMD) or 512 (two), one block per CU.  Prints cycles per instruction per
// wave at the clock implied by a known-length s_sleep-free loop: rate = instructions / (time x clock); the clock is taken
// from hipDeviceProp (the part runs these loops at its maximum clock: no matrix pipe, little data toggling).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, float* out) {
    extern __shared__ char lds[];
    float a = threadIdx.x * 1e-3f + 0.5f, b = 0.999f;
    float sink = 0.0f;
    if (MODE == 0) {  // 8 independent v_fma_f32 streams
        float c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) { const int j = jj & 7; asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b)); }
        for (int j = 0; j < 8; ++j) sink += c[j];
    } else if (MODE == 1) {  // 8 independent v_pk_fma_f32 streams
        f32x2 c[8], a2 = {a, a}, b2 = {b, b};
        for (int j = 0; j < 8; ++j) c[j] = f32x2{(float)j, (float)j + 0.5f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) { const int j = jj & 7; asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a2), "v"(b2)); }
        for (int j = 0; j < 8; ++j) sink += c[j][0] + c[j][1];
    } else if (MODE == 2) {  // one dependent chain of v_fma_f32
        float c = 1.0f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) { const int j = jj & 7; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(b), "v"(a)); }
        sink = c;
    } else if (MODE == 3) {  // one dependent chain of v_pk_fma_f32 (the assembler-required wait state between them)
        f32x2 c = {1.0f, 2.0f}, a2 = {a, a}, b2 = {b, b};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) { const int j = jj & 7; asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(c) : "v"(b2), "v"(a2)); }
        sink = c[0] + c[1];
    } else if (MODE == 4) {  // v_exp_f32, 8 independent
        float c[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) { const int j = jj & 7; asm volatile("v_exp_f32 %0, %0" : "+v"(c[j])); }
        for (int j = 0; j < 8; ++j) sink += c[j];
    } else if (MODE == 5) {  // ds_write_b8, lane l -> byte (l % 16) + 128 (l / 16): the re-quantising epilogue's store pattern
        const unsigned addr = (threadIdx.x & 15) + 128 * ((threadIdx.x >> 4) & 3) + 2048 * (threadIdx.x >> 6);
        int v = threadIdx.x;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) { const int j = jj & 7; asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(0) : "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sink = (float)lds[threadIdx.x];
    } else if (MODE == 6) {  // v_pk_mul_f32, 8 independent
        f32x2 c[8], b2 = {b, b};
        for (int j = 0; j < 8; ++j) c[j] = f32x2{(float)j + 1.0f, (float)j + 0.5f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) { const int j = jj & 7; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(c[j]) : "v"(b2)); }
        for (int j = 0; j < 8; ++j) sink += c[j][0] + c[j][1];
    }
    if (sink == 12345.678f) out[0] = sink;
}

template <int MODE>
static void run(const char* name, int threads, double insts_per_iter, double flops_per_inst) {
    int dev = 0;
    hipDeviceProp_t p;
    hipGetDevice(&dev);
    hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount, iters = 1 << 13;
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 16384, 0, 64, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 16384, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double clock = p.clockRate * 1e3;  // Hz
    const double per_wave = (double)iters * insts_per_iter;
    const double waves_per_simd = threads / 256.0;
    const double cyc = ms * 1e-3 * clock / (per_wave * waves_per_simd);
    const double tflops = per_wave * (threads / 64.0) * cus * 64.0 * flops_per_inst / (ms * 1e-3) / 1e12;
    printf("%-34s %d waves/SIMD  %7.3f ms  %.2f cycles per instruction per SIMD (at %.2f GHz)  %.1f TFLOP/s\n", name, threads / 256, ms, cyc,
           clock / 1e9, tflops);
    hipFree(out);
}

int main() {
    for (int t : {256, 512, 1024}) {
        run<0>("v_fma_f32 x8 independent", t, 64, 2);
        run<1>("v_pk_fma_f32 x8 independent", t, 64, 4);
        run<2>("v_fma_f32 dependent chain", t, 64, 2);
        run<3>("v_pk_fma_f32 dependent chain", t, 64, 4);
        run<6>("v_pk_mul_f32 x8 independent", t, 64, 2);
        run<4>("v_exp_f32 x8 independent", t, 64, 1);
        run<5>("ds_write_b8 (16 B x 4 rows per wave)", t, 64, 0);
    }
    return 0;
}
