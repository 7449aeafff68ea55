// block_select.hpp — top-k of up to R * T packed keys held in LDS without sorting them all.
//
// A block-wide bitonic sort of 1,024-2,048 64-bit keys on one CU is LDS-bandwidth-bound (15-25 us); the callers only
// need the best k.  The keys go to registers (R per thread), the k-th largest is bracketed by bisection on the key
// value — one ballot count and one barrier per bit, until at most next_pow2(k) keys remain at or above the bound — and
// only those are written back for the caller's (small) sort.  Keys are unique apart from 0 = empty, so the result is
// the same set a full sort would put in a[0 .. k).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace cs {

// a[0 .. n): keys (0 = empty), n <= R * T.  On return a[0 .. c) holds every key >= the bound found (c >= min(k, number
// of non-zero keys); usually c <= max(64, next_pow2(k)), twice that from k = 512), a[c .. ns) is zero, and ns (returned) is the power of two
// >= max(c, k, 64) the caller should sort.  slots: 66 words of LDS scratch.  All T threads must call it.
template <int T, int R>
__device__ __forceinline__ uint32_t block_select_topk(uint64_t* a, uint32_t n, uint32_t k, int tid, uint32_t* slots) {
    uint64_t key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)tid + (uint32_t)r * T;
        key[r] = i < n ? a[i] : 0ull;
    }
    for (int i = tid; i < 66; i += T) slots[i] = 0u;
    __syncthreads();
    auto count_ge = [&](uint64_t bound, int slot) -> uint32_t {  // block-uniform result; zero keys never count (bound >= 1)
        uint32_t w = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) w += (uint32_t)__popcll(__ballot(key[r] >= bound));
        if ((tid & 63) == 0 && w) atomicAdd(&slots[slot], w);
        __syncthreads();
        return slots[slot];
    };
    uint32_t want = 64;
    while (want < k) want <<= 1;
    if (k >= 512) want <<= 1;  // long lists: an exact bracket would cost all 64 bits; a 2k-key sort is cheaper than that
    uint32_t c = count_ge(1ull, 64);
    uint64_t lo = 1ull;
    if (c > want) {  // more live keys than the small sort takes: raise the bound bit by bit while >= k keys stay above it
        lo = 0ull;
#pragma unroll 1
        for (int bit = 63; bit >= 0; --bit) {
            const uint64_t cand = lo | (1ull << bit);
            const uint32_t cc = count_ge(cand, bit);
            if (cc >= k) {
                lo = cand;
                c = cc;
                if (c <= want) break;
            }
        }
        if (lo == 0ull) lo = 1ull;
    }
    // every thread holds its keys in registers: a[] can be rewritten
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (key[r] >= lo) a[atomicAdd(&slots[65], 1u)] = key[r];
    __syncthreads();
    const uint32_t kept = slots[65];
    uint32_t ns = 64;
    while (ns < kept || ns < k) ns <<= 1;
    for (uint32_t i = kept + (uint32_t)tid; i < ns; i += T) a[i] = 0ull;
    __syncthreads();
    return ns;
}

}  // namespace cs
