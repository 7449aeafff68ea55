// search_latency.cpp — latency of cs_index_search (host buffers: query in, results out, one synchronisation) measured from
// compiled code, at the corpus sizes the reference actually sees (its own benchmark indexes 592 chunks; its search
// path issues <= 9 query variants with limit up to 200, /root/reference/src/search/mod.rs:494-511).  The Python
// harness (search_latency.py) adds ~15-25 us of ctypes / numpy time per call to these figures.
//   g++ -O2 -std=c++17 benchmarks/search_latency.cpp -o benchmarks/search_latency_cpp -Lcodesearch_amd -lcsgpu -Wl,-rpath,$PWD/codesearch_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/codesearch_gpu.h"
#include "../include/cs_synth.h"

static double timed(cs_index* h, const std::vector<float>& q, uint32_t nq, uint32_t k, int reps, bool variants) {
    std::vector<float> cos((size_t)nq * k);
    std::vector<uint32_t> ids((size_t)nq * k), counts(nq);
    uint32_t cnt = 0;
    int32_t flag = 0;
    auto once = [&]() {
        if (variants) return cs_index_search_variants(h, q.data(), nq, 384, k, cos.data(), ids.data(), &cnt, &flag);
        return cs_index_search(h, q.data(), nq, 384, k, cos.data(), ids.data(), counts.data());
    };
    for (int i = 0; i < 400; ++i)
        if (once() != CS_OK) { std::printf("error: %s\n", cs_last_error()); return -1; }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) once();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main(int argc, char** argv) {
    if (cs_device_count() < 1) { std::printf("no HIP device\n"); return 77; }
    std::vector<uint64_t> sizes = {592ull, 10000ull, 100000ull, 1000000ull};
    if (argc > 1) { sizes.clear(); for (int i = 1; i < argc; ++i) sizes.push_back(strtoull(argv[i], nullptr, 10)); }
    for (uint64_t n : sizes) {
        cs_index* h = nullptr;
        if (cs_index_create(384, n, 0, 0, &h) != CS_OK || cs_index_add_synthetic(h, n, 1234, 0, nullptr) != CS_OK ||
            cs_index_build(h) != CS_OK) { std::printf("error: %s\n", cs_last_error()); return 1; }
        std::vector<float> q(9 * 384);
        for (size_t i = 0; i < q.size(); ++i) q[i] = cs_synth_value(99, i);
        std::printf("rows %llu: 1 query k=10 %.1f us; 9 queries k=10 %.1f us; 9 queries k=200 %.1f us; "
                    "9 variants k=200 merged on the device %.1f us\n", (unsigned long long)n, timed(h, q, 1, 10, 2000, false),
                    timed(h, q, 9, 10, 2000, false), timed(h, q, 9, 200, 1000, false), timed(h, q, 9, 200, 1000, true));
        std::fflush(stdout);
        cs_index_destroy(h);
    }
    return 0;
}
