// gemm_probe.hip — diagnostic build of the split-f16 GEMM (codesearch_amd/csrc/gemm_split.hip) with
// s_memtime stamps: where does a block spend its life?  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I codesearch_amd/csrc benchmarks/gemm_probe.hip -o benchmarks/gemm_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long* g_stamps = nullptr;  // [blocks][4]
__device__ unsigned int* g_where = nullptr;         // [blocks]: HW_ID | XCC_ID << 16 of the block's first wave
#define SH_STAMP(i)                                                                       \
    do {                                                                                  \
        if (threadIdx.x == 0 && g_stamps) {                                               \
            g_stamps[(size_t)blockIdx.x * 4 + (i)] = __builtin_readcyclecounter();        \
            if ((i) == 0 && g_where)                                                      \
                g_where[blockIdx.x] = (__builtin_amdgcn_s_getreg((4) | (0 << 6) | (15 << 11)) & 0xffffu) | \
                                      ((__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xfu) << 16);  \
        }                                                                                 \
    } while (0)

#include "../codesearch_amd/csrc/gemm_split.hip"

namespace cs {
std::string& last_error_ref() { static thread_local std::string m; return m; }
int32_t fail(int32_t code, const char* fmt, ...) { (void)fmt; return code; }
int32_t launch_gemm(int, const float*, const float*, const float*, const float*, float*, uint32_t, uint32_t, uint32_t, hipStream_t) { return 0; }
}  // namespace cs

__global__ void fill_kernel(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16)(((int)(x & 0xffff) - 32768) * (1.0f / 65536.0f));
    }
}

int main(int argc, char** argv) {
    const uint32_t M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 1152, K = argc > 3 ? atoi(argv[3]) : 384;
    const int epi = argc > 4 ? atoi(argv[4]) : 0;
    _Float16 *A, *W, *Cs;
    float *bias, *C, *resid;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&Cs, (size_t)M * N * 4);
    hipMalloc(&bias, N * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&resid, (size_t)M * N * 4);
    fill_kernel<<<2048, 256>>>(A, (size_t)M * K * 2, 1);
    fill_kernel<<<2048, 256>>>(W, (size_t)N * K * 2, 2);
    hipMemset(bias, 0, N * 4); hipMemset(resid, 0, (size_t)M * N * 4);
    const uint32_t bm = 128;
    const uint32_t blocks = cs::sh_grid_blocks((M + bm - 1) / bm, N / 128);
    unsigned long long* d_st;
    hipMalloc(&d_st, (size_t)blocks * 4 * 8);
    hipMemset(d_st, 0, (size_t)blocks * 4 * 8);
    unsigned int* d_wh;
    hipMalloc(&d_wh, (size_t)blocks * 4);
    hipMemset(d_wh, 0, (size_t)blocks * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_where), &d_wh, sizeof(d_wh));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long* ptr = pass ? d_st : nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ptr, sizeof(ptr));
        for (int w = 0; w < 3; ++w) cs::launch_gemm_split(epi, A, W, bias, resid, C, Cs, M, N, K, nullptr, nullptr);
        hipEventRecord(e0);
        const int iters = 20;
        for (int i = 0; i < iters; ++i) cs::launch_gemm_split(epi, A, W, bias, resid, C, Cs, M, N, K, nullptr, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: M=%u N=%u K=%u epi=%d  %.1f us/launch  %.0f TF executed f16\n", pass ? "stamped" : "plain", M, N, K, epi,
               ms * 1e3 / iters, 3.0 * 2.0 * M * N * K / (ms * 1e-3 / iters) / 1e12);
    }
    std::vector<unsigned long long> st((size_t)blocks * 4);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    double seg[3] = {0, 0, 0};
    unsigned long long tmin = ~0ull, tmax = 0;
    size_t nb = 0;
    std::vector<double> life;
    for (uint32_t b = 0; b < blocks; ++b) {
        const unsigned long long* s = &st[(size_t)b * 4];
        if (!s[3]) continue;
        for (int i = 0; i < 3; ++i) seg[i] += (double)(s[i + 1] - s[i]);
        tmin = std::min(tmin, s[0]); tmax = std::max(tmax, s[3]);
        life.push_back((double)(s[3] - s[0]));
        ++nb;
    }
    {   // per-CU timeline of the LAST stamped launch: how much of a CU's span is covered by resident blocks?
        std::vector<unsigned int> wh(blocks);
        hipMemcpy(wh.data(), d_wh, wh.size() * 4, hipMemcpyDeviceToHost);
        struct Ev { unsigned long long t0, t1; };
        std::map<unsigned int, std::vector<Ev>> cu;
        for (uint32_t b = 0; b < blocks; ++b) {
            const unsigned long long* s4 = &st[(size_t)b * 4];
            if (!s4[3]) continue;
            const unsigned int key = (wh[b] >> 16) << 16 | (wh[b] & 0xff00u);  // xcc | se, sh, cu
            cu[key].push_back({s4[0], s4[3]});
        }
        double busy = 0, span = 0, gap_sum = 0;
        size_t gaps = 0, maxb = 0;
        for (auto& kv : cu) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end(), [](const Ev& a, const Ev& b) { return a.t0 < b.t0; });
            unsigned long long lo = v.front().t0, hi = 0;
            for (auto& e : v) { busy += (double)(e.t1 - e.t0); hi = std::max(hi, e.t1); }
            span += (double)(hi - lo);
            maxb = std::max(maxb, v.size());
            // replacement latency: for each block end, the next block start on this CU at or after it
            for (auto& e : v) {
                unsigned long long best = ~0ull;
                for (auto& f : v) if (f.t0 >= e.t1 && f.t0 < best) best = f.t0;
                if (best != ~0ull) { gap_sum += (double)(best - e.t1); ++gaps; }
            }
        }
        printf("CUs seen %zu (max %zu blocks on one); sum(block life)/sum(CU span) = %.2f resident blocks per CU on average; "
               "mean CU span %.0f ticks; mean end->next-start on the same CU %.0f ticks\n",
               cu.size(), maxb, busy / span, span / cu.size(), gaps ? gap_sum / gaps : 0.0);
    }
    std::sort(life.begin(), life.end());
    printf("blocks %zu: mainloop %.0f  acc->lds %.0f  epilogue %.0f cycles (avg per block; s_memtime ticks)\n", nb, seg[0] / nb,
           seg[1] / nb, seg[2] / nb);
    printf("block life median %.0f p90 %.0f; kernel span %.0f ticks; sum(life)/span = %.1f blocks in flight (of %d slots)\n",
           life[life.size() / 2], life[life.size() * 9 / 10], (double)(tmax - tmin), (seg[0] + seg[1] + seg[2]) / (double)(tmax - tmin), 512);
    return 0;
}
