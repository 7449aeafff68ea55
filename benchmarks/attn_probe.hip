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
            g_stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (i)] = __builtin_readcyclecounter(); \
    } while (0)

#include "../codesearch_amd/csrc/attention_split.hip"

namespace cs {
std::string& last_error_ref() { static thread_local std::string m; return m; }
int32_t fail(int32_t code, const char* fmt, ...) { (void)fmt; return code; }
}  // namespace cs

__global__ void fill_kernel(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f));  // any f16 values time alike: [hi | lo] lines
    }
}
__global__ void ones_kernel(int* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1;
}

int main(int argc, char** argv) {
    const uint32_t B = argc > 1 ? atoi(argv[1]) : 256, L = argc > 2 ? atoi(argv[2]) : 256, H = 384, heads = 12;
    _Float16* qkv; int* mask; _Float16* ctxs;  // qkv: split-f16 [T][3H/32][64], same bytes as an f32 [T][3H]
    hipMalloc(&qkv, (size_t)B * L * 3 * H * 4); hipMalloc(&mask, (size_t)B * L * 4); hipMalloc(&ctxs, (size_t)B * L * H * 4);
    fill_kernel<<<2048, 256>>>(qkv, (size_t)B * L * 3 * H * 2, 7);
    ones_kernel<<<256, 256>>>(mask, (size_t)B * L);
    const size_t blocks = (size_t)heads * B * ((L + 127) / 128);  // upper bound over both kernels' grids
    unsigned long long* d_st;
    hipMalloc(&d_st, blocks * 4 * 8); hipMemset(d_st, 0, blocks * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long* ptr = pass ? d_st : nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ptr, sizeof(ptr));
        for (int w = 0; w < 3; ++w) cs::launch_attention_sh2(qkv, mask, ctxs, nullptr, B, L, H, heads, nullptr);
        hipEventRecord(e0);
        const int iters = 20;
        for (int i = 0; i < iters; ++i) cs::launch_attention_sh2(qkv, mask, ctxs, nullptr, B, L, H, heads, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: B=%u L=%u  %.1f us/launch\n", pass ? "stamped" : "plain", B, L, ms * 1e3 / iters);
    }
    std::vector<unsigned long long> st(blocks * 4);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    // stamps: 0 block start, 1 K/V resident (whole-sequence kernel, CS_ATTN_SHX1=0, only), 2 block end
    double pro = 0, life = 0; size_t nb = 0, npro = 0;
    for (size_t b = 0; b < blocks; ++b) {
        const unsigned long long* s = &st[b * 4];
        if (!s[2]) continue;
        if (s[1]) { pro += (double)(s[1] - s[0]); ++npro; }
        life += (double)(s[2] - s[0]); ++nb;
    }
    printf("blocks %zu: life %.0f cycles (avg per block)", nb, nb ? life / nb : 0.0);
    if (npro) printf(", prologue %.0f", pro / npro);
    printf("\n");
    return 0;
}
