// attn_clock_probe.hip — the product attention kernel (attention_shx_kernel<1>, split-f16 qkv in, split ctx out) with an
// (s_memtime, s_memrealtime) pair at the start and the end of every block: in-kernel clock and block lifetime after
// > 2 s of back-to-back launches on random data (MI355X_MICROARCH.md, DVFS give-back item 6).  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I codesearch_amd/csrc benchmarks/attn_clock_probe.hip -o benchmarks/attn_clock_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__device__ unsigned long long* g_stamps = nullptr;  // [blocks][4]: clk0, real0, clk1, real1
#define AT_STAMP(i)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0 && g_stamps) {                                                                      \
            const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;           \
            g_stamps[blk * 4 + (i)] = __builtin_amdgcn_s_memtime();                                              \
            g_stamps[blk * 4 + (i) + 1] = __builtin_amdgcn_s_memrealtime();                                      \
        }                                                                                                        \
    } while (0)

#include "../codesearch_amd/csrc/attention_split.hip"

namespace cs {
std::string& last_error_ref() { static thread_local std::string m; return m; }
int32_t fail(int32_t code, const char* fmt, ...) { (void)fmt; return code; }
}  // namespace cs

__global__ void fill_f16_kernel(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f));  // hi and lo halves alike: full-range random
    }
}
__global__ void ones_kernel(int* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1;
}

int main(int argc, char** argv) {
    const uint32_t B = argc > 1 ? atoi(argv[1]) : 256, L = argc > 2 ? atoi(argv[2]) : 256, H = 384, heads = 12;
    const int iters = argc > 3 ? atoi(argv[3]) : 14000;
    _Float16 *qkv, *ctxs; int* mask; uint32_t* flag;
    hipMalloc(&qkv, (size_t)B * L * 3 * H * 4); hipMalloc(&mask, (size_t)B * L * 4); hipMalloc(&ctxs, (size_t)B * L * H * 4);
    hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
    fill_f16_kernel<<<2048, 256>>>(qkv, (size_t)B * L * 3 * H * 2, 7);
    ones_kernel<<<256, 256>>>(mask, (size_t)B * L);
    const size_t blocks = (size_t)heads * B * ((L + 127) / 128);
    unsigned long long* d_st;
    hipMalloc(&d_st, blocks * 4 * 8); hipMemset(d_st, 0, blocks * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long* ptr = pass ? d_st : nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ptr, sizeof(ptr));
        for (int w = 0; w < 3; ++w) cs::launch_attention_sh2(qkv, mask, ctxs, flag, B, L, H, heads, nullptr);
        hipEventRecord(e0);
        const int n = pass ? iters : 50;
        for (int i = 0; i < n; ++i) cs::launch_attention_sh2(qkv, mask, ctxs, flag, B, L, H, heads, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: B=%u L=%u  %.1f us/launch over %d launches\n", pass ? "stamped" : "plain", B, L, ms * 1e3 / n, n);
    }
    std::vector<unsigned long long> st(blocks * 4);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, life;
    for (size_t b = 0; b < blocks; ++b) {
        const unsigned long long* s = &st[b * 4];
        if (s[3] > s[1]) { ghz.push_back((double)(s[2] - s[0]) / (double)(s[3] - s[1]) * 0.1); life.push_back((double)(s[2] - s[0])); }
    }
    if (ghz.empty()) { printf("no stamps\n"); return 1; }
    std::sort(ghz.begin(), ghz.end()); std::sort(life.begin(), life.end());
    printf("in-kernel clock %.3f GHz (median of %zu blocks; p10 %.3f, p90 %.3f); block lifetime median %.0f cycles\n",
           ghz[ghz.size() / 2], ghz.size(), ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], life[life.size() / 2]);
    return 0;
}
