// attn_probe.hip — diagnostic build of attention_split.hip with s_memtime stamps.  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I codesearch_amd/csrc benchmarks/attn_probe.hip -o benchmarks/attn_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long* g_stamps = nullptr;  // [blocks][4]
#define AT_STAMP(i)                                                                                       \
    do {                                                                                                  \
        if (threadIdx.x == 0 && g_stamps)                                                                 \
            g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (i)] = __builtin_readcyclecounter(); \
    } while (0)

#include "../codesearch_amd/csrc/attention_split.hip"

namespace cs {
std::string& last_error_ref() { static thread_local std::string m; return m; }
int32_t fail(int32_t code, const char* fmt, ...) { (void)fmt; return code; }
}  // namespace cs

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f);
    }
}
__global__ void ones_kernel(int* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1;
}

int main(int argc, char** argv) {
    const uint32_t B = argc > 1 ? atoi(argv[1]) : 256, L = argc > 2 ? atoi(argv[2]) : 256, H = 384, heads = 12;
    float* qkv; int* mask; _Float16* ctxs;
    hipMalloc(&qkv, (size_t)B * L * 3 * H * 4); hipMalloc(&mask, (size_t)B * L * 4); hipMalloc(&ctxs, (size_t)B * L * H * 4);
    fill_kernel<<<2048, 256>>>(qkv, (size_t)B * L * 3 * H, 7);
    ones_kernel<<<256, 256>>>(mask, (size_t)B * L);
    const size_t blocks = (size_t)heads * B;
    unsigned long long* d_st;
    hipMalloc(&d_st, blocks * 4 * 8); hipMemset(d_st, 0, blocks * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long* ptr = pass ? d_st : nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ptr, sizeof(ptr));
        for (int w = 0; w < 3; ++w) cs::launch_attention_sh(qkv, mask, nullptr, ctxs, nullptr, B, L, H, heads, nullptr);
        hipEventRecord(e0);
        const int iters = 20;
        for (int i = 0; i < iters; ++i) cs::launch_attention_sh(qkv, mask, nullptr, ctxs, nullptr, B, L, H, heads, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: B=%u L=%u  %.1f us/launch\n", pass ? "stamped" : "plain", B, L, ms * 1e3 / iters);
    }
    std::vector<unsigned long long> st(blocks * 4);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    double seg[2] = {0, 0}; size_t nb = 0;
    for (size_t b = 0; b < blocks; ++b) {
        const unsigned long long* s = &st[b * 4];
        if (!s[2]) continue;
        seg[0] += (double)(s[1] - s[0]); seg[1] += (double)(s[2] - s[1]); ++nb;
    }
    printf("blocks %zu: prologue %.0f  main %.0f cycles (avg per block)\n", nb, seg[0] / nb, seg[1] / nb);
    return 0;
}
