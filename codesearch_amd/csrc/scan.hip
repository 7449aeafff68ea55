// scan.hip — brute-force cosine scan with fused streaming top-k,
// the cross-block / cross-shard key merge (S2/S4) and the in-place corpus generator.
//
// Replaces, in /root/reference:
//   examples/benchmark_models.rs:155-165 + :323-328  (linear scan + cosine_similarity)
//   src/vectordb/store.rs:446-459                    (arroy nns().by_vector(): ANN there,
//                                                     exact here)
//
// HBM-bound: each corpus row is read exactly once per query tile as 16 B/lane
// coalesced loads (32 lanes x float4 = one 512 B row segment per half-wave), both
// dot(q,x) and |x|^2 are accumulated from the same registers, and selection happens in
// registers/LDS, so the only HBM traffic besides the matrix is k keys per block.
// Algorithmic bytes per row = dim*4 (1536 B at dim 384).
#include "scan.hpp"
#include "block_select.hpp"

#include "../../include/cs_synth.h"

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;   // 4 waves
constexpr int kWaves = kBlock / 64;
constexpr int kMergeBlock = 1024;
constexpr int kMergeCap = 4096;  // keys a merge block can sort (2048 used up to k = 512, see merge_group)

// ---- wave helpers ---------------------------------------------------------------------

// Sum over the 32 lanes of each half-wave; every lane of the half receives the total.
__device__ __forceinline__ float half_allreduce_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}
__device__ __forceinline__ float wave_allreduce_sum(float v) {
    v += __shfl_xor(v, 32, 64);
    return half_allreduce_sum(v);
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Per-wave candidate list: k live slots (of kpad) in LDS holding packed keys, 0 = empty.
// State kept by the caller: `thr` = cosine of the current worst slot (-inf while any slot
// is empty) and `wpos` = that slot's index.  Rows are streamed in ascending id, so a row
// that ties the worst cosine loses to it (id asc) and `c > thr` is the whole test — the
// `>` of benchmark_models.rs:160.  `floor` is what thr falls back to while a slot is empty:
// -inf, or the primed lower bound (see scan_topk_kernel's PRIME mode).
__device__ __forceinline__ void wave_list_insert(volatile uint64_t* list_generic, uint32_t k, int lane,
                                                 float c, uint32_t id, float& thr,
                                                 uint32_t& wpos,
                                                 float floor = -__builtin_huge_valf()) {
    // While the list still has empty slots they are filled in index order (the search below picks the lowest
    // empty slot), thr stays at the floor and nothing needs searching: wpos < kListFull counts the filled slots.
    // A k = 200 list over a small corpus never leaves this phase; the 64-lane search (~1,000 cycles) starts with
    // the insert that fills the last slot.
    constexpr uint32_t kListFull = 0x80000000u;
    const uint32_t slot = wpos & ~kListFull;
    // The list is LDS; say so.  Through the generic pointer these were flat_store / flat_load, which count on vmcnt
    // as well and return out of order with the corpus loads in flight: every `s_waitcnt vmcnt(N)` of the scan loop
    // after a possible insert degraded to vmcnt(0).
    typedef volatile uint64_t __attribute__((address_space(3))) lds_vu64;
    lds_vu64* const list = (lds_vu64*)list_generic;
    if (lane == 0) list[slot] = key_pack(c, id);
    if (!(wpos & kListFull) && slot + 1 < k) {
        wpos = slot + 1;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint64_t mk = ~0ull;
    uint32_t mp = 0xffffffffu;
    for (uint32_t i = lane; i < k; i += 64) {
        uint64_t v = list[i];
        if (v < mk) { mk = v; mp = i; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        uint64_t ok = shfl_xor_u64(mk, m);
        uint32_t op = __shfl_xor(mp, m, 64);
        if (ok < mk || (ok == mk && op < mp)) { mk = ok; mp = op; }
    }
    wpos = mp | kListFull;
    thr = (mk == 0ull) ? floor : key_cos(mk);
}

// Bitonic sort, descending, of a[0..n) (n a power of two) by all threads of the block.
// Pair t of a stage is handled by thread t % T, i.e. by wave (t / 64) % (T / 64), and for strides
// <= 64 it lies inside the 128-key segment [128 (t / 64), +128): such a stage reads only what the
// same wave wrote in the stage before, so it needs the wave's own LDS ordering, not a block
// barrier.  Only stages with stride >= 128, and the stage right after one, synchronise the block
// (20 of the 78 stages of a 4096-key sort).
template <int T>
__device__ __forceinline__ void block_bitonic_desc(uint64_t* a, uint32_t n, int tid) {
    uint32_t prev_stride = 128;  // whatever filled a[] was another wave
    for (uint32_t size = 2; size <= n; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 128 || prev_stride >= 128) {
                __syncthreads();
            } else {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            prev_stride = stride;
            for (uint32_t t = tid; t < (n >> 1); t += T) {
                uint32_t i = 2 * t - (t & (stride - 1));
                uint32_t j = i + stride;
                uint64_t x = a[i], y = a[j];
                bool desc = ((i & size) == 0);
                if ((x < y) == desc) { a[i] = y; a[j] = x; }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ bool row_is_dead(const uint32_t* dead, uint64_t row) {
    if (!dead) return false;
    return (dead[row >> 5] >> (row & 31)) & 1u;
}

// ---- S1+S2: the streaming scan ----------------------------------------------------------
//
// J  = float4 chunks per lane per row (dim = 128*J);  U = row pairs in flight per wave
// (a wave tile is 2U consecutive rows: half-wave h takes row 2u+h);  QT = queries scored
// per pass from registers;  NT = non-temporal corpus loads.
//
// Primed scans (large k).  A k=200 list costs ~450 cycles per insert (the 64-lane search for
// the new worst slot) and a wave that sees 2,000 rows makes ~650 of them: 10 % of the scan.
// Nearly all of those rows are nowhere near the global top-k.  PRIME = true is a cheap pass of
// the SAME arithmetic over a prefix of the corpus that keeps one number per wave and query, the
// best live cosine it saw; the waves' chunks are disjoint, so the k-th largest of those maxima
// is attained by k different rows and is a lower bound of the corpus' k-th best cosine.  The
// last block to finish selects it and writes floor_out[q] = the float just below it (rows
// EQUAL to the bound must still pass the `>`; a bound in the denormal range becomes -FLT_MIN so
// the test never depends on the denormal mode).  The full scan then starts every list with
// thr = floor instead of -inf and inserts ~10 rows per wave instead of ~650; results are
// bit-identical because only rows that cannot be among the best k are skipped.
template <int J, int U, int QT, bool NT, bool PRIME = false>
__global__ void __launch_bounds__(kBlock)
scan_topk_kernel(const float* __restrict__ corpus, uint64_t n_rows,
                 const float* __restrict__ queries, uint32_t nq, uint32_t k, uint32_t kpad,
                 const uint32_t* __restrict__ dead, RowIds id_base,
                 uint64_t* __restrict__ partial, const float* __restrict__ floor_in,
                 float* __restrict__ wave_max, uint32_t* __restrict__ done_ctr,
                 float* __restrict__ floor_out, const uint32_t* __restrict__ gate) {
    // gate != null: this launch is the exact rerun enqueued behind a batched (filter + refine) search on the
    // device API; it runs only if that search overflowed a candidate buffer (index.hip run_search)
    if (gate && *gate == 0u) return;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_keys[];  // [QT][kWaves][kpad]
    constexpr int DIM = 128 * J;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l32 = lane & 31;
    const uint32_t q0 = blockIdx.y * QT;

    for (uint32_t i = tid; i < QT * kWaves * kpad; i += kBlock) lds_keys[i] = 0ull;

    // query fragments + magnitudes (mag_a of benchmark_models.rs:325)
    f32x4 qf[QT][J];
    float qmag[QT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        const uint32_t q = (q0 + qi < nq) ? (q0 + qi) : (nq - 1);
        const f32x4* qp = reinterpret_cast<const f32x4*>(queries + (size_t)q * DIM) + l32;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            qf[qi][j] = qp[j * 32];
            s = fmaf(qf[qi][j].x, qf[qi][j].x, s);
            s = fmaf(qf[qi][j].y, qf[qi][j].y, s);
            s = fmaf(qf[qi][j].z, qf[qi][j].z, s);
            s = fmaf(qf[qi][j].w, qf[qi][j].w, s);
        }
        qmag[qi] = sqrtf(half_allreduce_sum(s));
    }
    float thr[QT], floor[QT];
    uint32_t wpos[QT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        floor[qi] = (!PRIME && floor_in) ? floor_in[(q0 + qi < nq) ? (q0 + qi) : (nq - 1)]
                                         : -__builtin_huge_valf();
        thr[qi] = floor[qi];  // PRIME: the lane's running maximum
        wpos[qi] = 0;
    }
    __syncthreads();

    const uint64_t gw = (uint64_t)blockIdx.x * kWaves + wave;
    const uint64_t nw = (uint64_t)gridDim.x * kWaves;
    const uint64_t ntiles = (n_rows + 2 * U - 1) / (2 * U);

    // One tile = 2U consecutive rows.  load_tile issues the tile's loads; score_tile consumes them.  (Issuing the
    // NEXT tile's loads before scoring the current one — two register sets, loop unrolled by two — was measured
    // and is slower at every depth: 10M x 384, one block per CU, U = 8: 2.146 -> 2.200 ms, U = 4: 2.33 ms.)
    auto load_tile = [&](f32x4 (&x)[U][J], uint64_t tile) {
        const uint64_t row0 = tile * (2 * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t r = row0 + 2 * u + half;
            r = r < n_rows ? r : n_rows - 1;  // tail rows re-read the last row, masked below
            const f32x4* p = reinterpret_cast<const f32x4*>(corpus + r * DIM) + l32;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if constexpr (NT) x[u][j] = __builtin_nontemporal_load(p + j * 32);
                else x[u][j] = p[j * 32];
            }
        }
    };
    auto score_tile = [&](const f32x4 (&x)[U][J], uint64_t tile) {
        const uint64_t row0 = tile * (2 * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float ss = 0.0f;
            float dot[QT];
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) dot[qi] = 0.0f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const f32x4 v = x[u][j];
                ss = fmaf(v.x, v.x, ss);
                ss = fmaf(v.y, v.y, ss);
                ss = fmaf(v.z, v.z, ss);
                ss = fmaf(v.w, v.w, ss);
#pragma unroll
                for (int qi = 0; qi < QT; ++qi) {
                    dot[qi] = fmaf(v.x, qf[qi][j].x, dot[qi]);
                    dot[qi] = fmaf(v.y, qf[qi][j].y, dot[qi]);
                    dot[qi] = fmaf(v.z, qf[qi][j].z, dot[qi]);
                    dot[qi] = fmaf(v.w, qf[qi][j].w, dot[qi]);
                }
            }
            const float xmag = sqrtf(half_allreduce_sum(ss));  // mag_b
            const uint64_t r = row0 + 2 * u + half;
            const bool valid = r < n_rows;
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) {
                const float d = half_allreduce_sum(dot[qi]);
                // batch.rs:320-323: zero magnitude -> 0.0, else dot / (mag_a * mag_b)
                const float c = (qmag[qi] == 0.0f || xmag == 0.0f) ? 0.0f : d / (qmag[qi] * xmag);
                if constexpr (PRIME) {
                    if (valid && c > thr[qi] && !row_is_dead(dead, r)) thr[qi] = c;
                    continue;
                }
                unsigned long long m = __ballot(valid && l32 == 0 && c > thr[qi]);
                if (m) {  // rare: wave-uniform slow path
                    volatile uint64_t* list = lds_keys + ((size_t)qi * kWaves + wave) * kpad;
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const float cc = __shfl(c, src, 64);
                        const uint64_t rr = row0 + 2 * u + (src >> 5);
                        if (cc > thr[qi] && !row_is_dead(dead, rr))
                            wave_list_insert(list, k, lane, cc, id_base.of(rr), thr[qi],
                                             wpos[qi], floor[qi]);
                    }
                }
            }
        }
    };
    for (uint64_t tile = gw; tile < ntiles; tile += nw) {
        f32x4 x[U][J];
        load_tile(x, tile);
        score_tile(x, tile);
    }
    if constexpr (PRIME) {
        // wave maxima -> HBM; the last block of this pass selects the k-th largest per query
        const uint32_t nwaves = gridDim.x * kWaves;
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            const float m = fmaxf(__shfl(thr[qi], 0, 64), __shfl(thr[qi], 32, 64));
            if (lane == 0 && q0 + qi < nq)
                __hip_atomic_store(wave_max + (size_t)(q0 + qi) * nwaves + gw, m, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        __shared__ uint32_t is_last;
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const uint32_t prev = __hip_atomic_fetch_add(done_ctr + blockIdx.y, 1u, __ATOMIC_ACQ_REL,
                                                         __HIP_MEMORY_SCOPE_AGENT);
            is_last = (prev == gridDim.x - 1);
        }
        __syncthreads();
        if (!is_last) return;
        __threadfence();
        uint32_t nsort = 64;
        while (nsort < nwaves) nsort <<= 1;  // host keeps nsort <= kWaves * kpad (the LDS size)
#pragma unroll 1
        for (int qi = 0; qi < QT; ++qi) {
            if (q0 + qi >= nq) break;
            __syncthreads();
            for (uint32_t i = tid; i < nsort; i += kBlock) {
                float m = -__builtin_huge_valf();
                if (i < nwaves)
                    m = __hip_atomic_load(wave_max + (size_t)(q0 + qi) * nwaves + i, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
                lds_keys[i] = (m == -__builtin_huge_valf()) ? 0ull : key_pack(m, 0u);
            }
            block_bitonic_desc<kBlock>(lds_keys, nsort, tid);
            if (tid == 0) {
                const uint64_t key = (k <= nsort) ? lds_keys[k - 1] : 0ull;
                float t = -__builtin_huge_valf();
                if (key) {
                    const uint32_t o = (uint32_t)(key >> 32) - 1u;  // next float below the bound
                    t = key_cos((uint64_t)o << 32);
                    if (fabsf(t) < 1.17549435e-38f) t = -1.17549435e-38f;
                }
                floor_out[q0 + qi] = t;
            }
        }
        if (tid == 0) __hip_atomic_store(done_ctr + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __syncthreads();

    // block merge: the 4 wave lists of each query -> best k, sorted, to HBM
    const uint32_t nsort = kWaves * kpad;
#pragma unroll 1
    for (int qi = 0; qi < QT; ++qi) {
        if (q0 + qi >= nq) break;
        uint64_t* a = lds_keys + (size_t)qi * nsort;
        block_bitonic_desc<kBlock>(a, nsort, tid);
        uint64_t* out = partial + ((size_t)(q0 + qi) * gridDim.x + blockIdx.x) * k;
        for (uint32_t i = tid; i < k; i += kBlock) out[i] = a[i];
    }
}

// Any-dim fallback (e.g. the reference's own 4-d unit test, store.rs:846-893): one wave
// per row, lanes stride over columns.  Not a tuned path.
__global__ void __launch_bounds__(kBlock)
scan_topk_generic_kernel(const float* __restrict__ corpus, uint64_t n_rows, uint32_t dim,
                         const float* __restrict__ queries, uint32_t nq, uint32_t k,
                         uint32_t kpad, const uint32_t* __restrict__ dead, RowIds id_base,
                         uint64_t* __restrict__ partial, const uint32_t* __restrict__ gate) {
    if (gate && *gate == 0u) return;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_keys[];  // [kWaves][kpad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t q = blockIdx.y;
    for (uint32_t i = tid; i < kWaves * kpad; i += kBlock) lds_keys[i] = 0ull;
    const float* qp = queries + (size_t)q * dim;
    float s = 0.0f;
    for (uint32_t c = lane; c < dim; c += 64) s = fmaf(qp[c], qp[c], s);
    const float qmag = sqrtf(wave_allreduce_sum(s));
    float thr = -__builtin_huge_valf();
    uint32_t wpos = 0;
    __syncthreads();
    volatile uint64_t* list = lds_keys + (size_t)wave * kpad;
    const uint64_t gw = (uint64_t)blockIdx.x * kWaves + wave;
    const uint64_t nw = (uint64_t)gridDim.x * kWaves;
    for (uint64_t r = gw; r < n_rows; r += nw) {
        const float* xp = corpus + r * dim;
        float ss = 0.0f, dot = 0.0f;
        for (uint32_t c = lane; c < dim; c += 64) {
            const float v = xp[c];
            ss = fmaf(v, v, ss);
            dot = fmaf(v, qp[c], dot);
        }
        const float xmag = sqrtf(wave_allreduce_sum(ss));
        const float d = wave_allreduce_sum(dot);
        const float c = (qmag == 0.0f || xmag == 0.0f) ? 0.0f : d / (qmag * xmag);
        if (c > thr && !row_is_dead(dead, r))  // wave-uniform
            wave_list_insert(list, k, lane, c, id_base.of(r), thr, wpos);
    }
    __syncthreads();
    const uint32_t nsort = kWaves * kpad;
    block_bitonic_desc<kBlock>(lds_keys, nsort, tid);
    uint64_t* out = partial + ((size_t)q * gridDim.x + blockIdx.x) * k;
    for (uint32_t i = tid; i < k; i += kBlock) out[i] = lds_keys[i];
}

// ---- S2/S4: key-list merge ----------------------------------------------------------------
// in: list l of query q starts at in + q*q_stride + l*l_stride ([nq][nlists][k] for the
// scan partials, [nlists][nq][k] for all-gathered shard results); block (g, q) sorts lists [g*G, min(nlists,(g+1)*G)) and writes its
// best k to out_keys[q][g][k].  When gridDim.x == 1 the result is final and is also
// decoded to cos / ids / counts.
__global__ void __launch_bounds__(kMergeBlock)
merge_topk_kernel(const uint64_t* __restrict__ in, uint32_t nlists, uint32_t k, uint32_t G,
                  uint64_t q_stride, uint64_t l_stride, uint64_t* __restrict__ out_keys, float* __restrict__ out_cos,
                  uint32_t* __restrict__ out_ids, uint32_t* __restrict__ out_counts,
                  const uint32_t* __restrict__ gate, uint32_t remap_stripe, uint32_t remap_shards) {
    if (gate && *gate == 0u) return;  // see scan_topk_kernel
    __shared__ __attribute__((aligned(16))) uint64_t a[kMergeCap];
    __shared__ uint32_t live;
    const int tid = threadIdx.x;
    const uint32_t g = blockIdx.x, q = blockIdx.y, ngroups = gridDim.x;
    const uint32_t lo = g * G;
    const uint32_t hi = (lo + G < nlists) ? lo + G : nlists;
    const uint32_t ncand = (hi - lo) * k;
    uint32_t nsort = 64;
    while (nsort < ncand) nsort <<= 1;
    const uint64_t* src = in + (size_t)q * q_stride + (size_t)lo * l_stride;
    for (uint32_t i = tid; i < nsort; i += kMergeBlock) {
        const uint32_t l = i / k, e = i - l * k;
        uint64_t key = (i < ncand) ? src[(size_t)l * l_stride + e] : 0ull;
        if (remap_stripe && key) {
            // striped shards (shards.hip): list lo + l comes from shard lo + l and carries that shard's LOCAL row
            // numbers; global id = ((row / stripe) * shards + shard) * stripe + row % stripe — monotone in the
            // row within a shard, so each list stays sorted under (cosine desc, id asc)
            const uint32_t row = key_id(key);
            const uint32_t gid = ((row / remap_stripe) * remap_shards + (lo + l)) * remap_stripe + row % remap_stripe;
            key = (key & 0xffffffff00000000ull) | (uint64_t)(~gid);
        }
        a[i] = key;
    }
    if (tid == 0) live = 0;
    if (nsort > 256) {  // thousands of keys, k wanted: bracket the k-th first and sort only what is above it
        __shared__ uint32_t sel_slots[66];
        __syncthreads();
        nsort = block_select_topk<kMergeBlock, kMergeCap / kMergeBlock>(a, ncand, k, tid, sel_slots);
    }
    block_bitonic_desc<kMergeBlock>(a, nsort, tid);
    const bool final_pass = (ngroups == 1);
    for (uint32_t i = tid; i < k; i += kMergeBlock) {
        const uint64_t key = (i < nsort) ? a[i] : 0ull;
        if (out_keys) out_keys[((size_t)q * ngroups + g) * k + i] = key;
        if (final_pass) {
            if (key) atomicAdd(&live, 1u);
            if (out_cos) out_cos[(size_t)q * k + i] = key ? key_cos(key) : 0.0f;
            if (out_ids) out_ids[(size_t)q * k + i] = key ? key_id(key) : 0xffffffffu;
        }
    }
    __syncthreads();
    if (final_pass && out_counts && tid == 0) out_counts[q] = live;
}

// ---- variant merge --------------------------------------------------------
// The step right after the per-variant searches in search::search (/root/reference/src/search/mod.rs:513-611):
// the <= 9 query variants' result lists are unioned, a chunk id found by several variants keeps its best score
// (mod.rs:547-566), the best `limit` survive, sorted best-first (mod.rs:570-590), and the search skips its
// text-search leg when the top five all have distance < 0.15 (mod.rs:595-611).  One block: keys [nv][k] ->
// sort by (id, cosine) to find duplicates -> zero all but the best key of every id -> sort by key -> top
// `limit`.  Scores are monotone in the cosine, so "best score" = largest key; equal scores fall back to
// (cosine desc, id asc) where the reference's HashMap order is unspecified.
// More than kVariantHashMax keys (the table would not fit in LDS): duplicates found by a sort on (id, cosine image).
__global__ void __launch_bounds__(kMergeBlock)
merge_variants_sort2_kernel(const uint64_t* __restrict__ keys, uint32_t nkeys, uint32_t nsort, uint32_t limit,
                      uint64_t* __restrict__ out_keys, float* __restrict__ out_cos, uint32_t* __restrict__ out_ids,
                      uint32_t* __restrict__ out_count, uint32_t* __restrict__ out_high_confidence,
                      float max_distance, uint32_t top_n) {
    extern __shared__ __attribute__((aligned(16))) uint64_t va[];  // [nsort]
    __shared__ uint32_t live, confident;
    const int tid = threadIdx.x;
    // (id, order-preserving cosine image): equal ids become neighbours, the best cosine of an id first
    for (uint32_t i = tid; i < nsort; i += kMergeBlock) {
        const uint64_t key = i < nkeys ? keys[i] : 0ull;
        va[i] = key ? (((uint64_t)key_id(key) << 32) | (key >> 32)) : 0ull;
    }
    if (tid == 0) { live = 0; confident = 0; }
    block_bitonic_desc<kMergeBlock>(va, nsort, tid);
    constexpr int PER = kMergeCap * 4 / kMergeBlock;  // up to 16384 keys: 16 per thread
    uint64_t mine[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t i = tid + j * kMergeBlock;
        uint64_t v = i < nsort ? va[i] : 0ull;
        if (v && i > 0 && (va[i - 1] >> 32) == (v >> 32)) v = 0ull;  // same id, better or equal cosine just before
        mine[j] = v ? ((v << 32) | (uint64_t)(~(uint32_t)(v >> 32))) : 0ull;  // back to (cosine image, ~id)
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t i = tid + j * kMergeBlock;
        if (i < nsort) va[i] = mine[j];
    }
    block_bitonic_desc<kMergeBlock>(va, nsort, tid);
    for (uint32_t i = tid; i < limit; i += kMergeBlock) {
        const uint64_t key = i < nsort ? va[i] : 0ull;
        if (key) atomicAdd(&live, 1u);
        if (out_keys) out_keys[i] = key;
        if (out_cos) out_cos[i] = key ? key_cos(key) : 0.0f;
        if (out_ids) out_ids[i] = key ? key_id(key) : 0xffffffffu;
        // mod.rs:601-611 on the reference's own scale: distance = (1 - cos) / 2 (cs_cos_to_distance)
        if (key && i < top_n && (1.0f - key_cos(key)) * 0.5f < max_distance) atomicAdd(&confident, 1u);
    }
    __syncthreads();
    if (tid == 0) {
        if (out_count) *out_count = live;
        const uint32_t want = live < top_n ? live : top_n;
        if (out_high_confidence) *out_high_confidence = (live > 0 && confident == want) ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(kMergeBlock)
merge_variants_kernel(const uint64_t* __restrict__ keys, uint32_t nkeys, uint32_t nsort, uint32_t limit,
                      uint64_t* __restrict__ out_keys, float* __restrict__ out_cos, uint32_t* __restrict__ out_ids,
                      uint32_t* __restrict__ out_count, uint32_t* __restrict__ out_high_confidence,
                      float max_distance, uint32_t top_n) {
    // Duplicates first, through an LDS hash table keyed by chunk id (open addressing, 2 * nsort slots, at most half
    // full): an entry is (id + 1) << 32 | cosine image, so entries of one id compare by cosine and a 64-bit LDS
    // atomic max keeps the best.  One sort of the survivors then orders them.  (The first version sorted twice — by
    // (id, cosine) to find duplicates, then by key: two 2,048-key sorts on one CU are LDS-bandwidth-bound, 38 us for
    // nine lists of 200.)
    extern __shared__ __attribute__((aligned(16))) uint64_t va[];  // [nsort] sort buffer, then [2 * nsort] table
    __shared__ uint32_t live, confident, nuniq;
    unsigned long long* table = reinterpret_cast<unsigned long long*>(va + nsort);
    const int tid = threadIdx.x;
    const uint32_t tmask = 2 * nsort - 1;
    for (uint32_t i = tid; i < 2 * nsort; i += kMergeBlock) table[i] = 0ull;
    if (tid == 0) { live = 0; confident = 0; nuniq = 0; }
    __syncthreads();
    for (uint32_t i = tid; i < nkeys; i += kMergeBlock) {
        const uint64_t key = keys[i];
        if (!key) continue;
        const uint32_t id = key_id(key);
        const unsigned long long mine = ((unsigned long long)(id + 1u) << 32) | (key >> 32);  // ids stop at 2^32 - 2
        uint32_t slot = (id * 2654435761u) & tmask;
        for (;;) {
            unsigned long long cur = table[slot];
            if (cur == 0ull) cur = atomicCAS(&table[slot], 0ull, mine);
            if (cur == 0ull) break;                                              // claimed an empty slot
            if ((uint32_t)(cur >> 32) == id + 1u) { atomicMax(&table[slot], mine); break; }  // same chunk: best cosine
            slot = (slot + 1) & tmask;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 2 * nsort; i += kMergeBlock) {
        const unsigned long long e = table[i];
        if (e) va[atomicAdd(&nuniq, 1u)] = (e << 32) | (uint64_t)(~((uint32_t)(e >> 32) - 1u));  // back to (image, ~id)
    }
    __syncthreads();
    uint32_t ns = 64;
    while (ns < nuniq) ns <<= 1;
    for (uint32_t i = nuniq + tid; i < ns; i += kMergeBlock) va[i] = 0ull;
    if (ns > 256 && limit < nuniq) {  // bracket the limit-th key, sort only what is above it (block_select.hpp)
        __shared__ uint32_t sel_slots[66];
        __syncthreads();
        ns = block_select_topk<kMergeBlock, 4>(va, nuniq, limit, tid, sel_slots);  // nuniq <= 4096 on this path
    }
    block_bitonic_desc<kMergeBlock>(va, ns, tid);
    for (uint32_t i = tid; i < limit; i += kMergeBlock) {
        const uint64_t key = i < ns ? va[i] : 0ull;
        if (key) atomicAdd(&live, 1u);
        if (out_keys) out_keys[i] = key;
        if (out_cos) out_cos[i] = key ? key_cos(key) : 0.0f;
        if (out_ids) out_ids[i] = key ? key_id(key) : 0xffffffffu;
        // mod.rs:601-611 on the reference's own scale: distance = (1 - cos) / 2 (cs_cos_to_distance)
        if (key && i < top_n && (1.0f - key_cos(key)) * 0.5f < max_distance) atomicAdd(&confident, 1u);
    }
    __syncthreads();
    if (tid == 0) {
        if (out_count) *out_count = live;
        const uint32_t want = live < top_n ? live : top_n;
        if (out_high_confidence) *out_high_confidence = (live > 0 && confident == want) ? 1u : 0u;
    }
}

int32_t launch_merge_variants(const uint64_t* d_keys, uint32_t nv, uint32_t k, uint32_t limit, uint64_t* d_out_keys,
                              float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_count,
                              uint32_t* d_out_high_confidence, hipStream_t stream) {
    const uint32_t nkeys = nv * k;
    if (nkeys == 0 || nkeys > (uint32_t)kMergeCap * 4)
        return fail(CS_ERR_BAD_ARG, "variant merge takes 1..%d keys, got %u", kMergeCap * 4, nkeys);
    uint32_t nsort = 64;
    while (nsort < nkeys) nsort <<= 1;
    constexpr uint32_t kVariantHashMax = 4096;  // 3 * 4096 * 8 B = 96 KiB of LDS
    static PerDeviceOnce attr_set;  // function attributes are per device
    CS_TRY(attr_set.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(merge_variants_sort2_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kMergeCap * 4 * sizeof(uint64_t)));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(merge_variants_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kVariantHashMax * 3 * sizeof(uint64_t)));
        return CS_OK;
    }));
    if (nsort <= kVariantHashMax)
        hipLaunchKernelGGL(merge_variants_kernel, dim3(1), dim3(kMergeBlock), (size_t)nsort * 3 * sizeof(uint64_t), stream,
                           d_keys, nkeys, nsort, limit, d_out_keys, d_out_cos, d_out_ids, d_out_count,
                           d_out_high_confidence, 0.15f, 5u);
    else
        hipLaunchKernelGGL(merge_variants_sort2_kernel, dim3(1), dim3(kMergeBlock), (size_t)nsort * sizeof(uint64_t), stream,
                           d_keys, nkeys, nsort, limit, d_out_keys, d_out_cos, d_out_ids, d_out_count,
                           d_out_high_confidence, 0.15f, 5u);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ---- synthetic corpus, generated in HBM ---------------------------------------------------
__global__ void synth_fill_kernel(float* __restrict__ out, uint64_t total, uint64_t seed,
                                  uint64_t first_flat) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += stride) {
        if (i + 4 <= total) {
            f32x4 v;
            v.x = cs_synth_value(seed, first_flat + i);
            v.y = cs_synth_value(seed, first_flat + i + 1);
            v.z = cs_synth_value(seed, first_flat + i + 2);
            v.w = cs_synth_value(seed, first_flat + i + 3);
            *reinterpret_cast<f32x4*>(out + i) = v;
        } else {
            for (uint64_t j = i; j < total; ++j) out[j] = cs_synth_value(seed, first_flat + j);
        }
    }
}

// ---- host side ----------------------------------------------------------------------------

static uint32_t kpad_for(uint32_t k) {
    uint32_t p = 64;
    while (p < k) p <<= 1;
    return p;
}

static bool fast_dim(uint32_t dim) { return dim == 384 || dim == 768 || dim == 1024; }

template <int J, int U, int QT>
static int occupancy_of(size_t lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, scan_topk_kernel<J, U, QT, true>, kBlock,
                                                     lds) != hipSuccess || nb < 1)
        nb = 2;
    return nb;
}

// Resident blocks per CU of the kernel instance a plan selects (the grid is sized to be
// fully resident: a grid-stride scan with a second wave of blocks would idle CUs).
static int blocks_per_cu(uint32_t dim, uint32_t qtile, uint32_t kpad, bool deep) {
    const size_t lds = (size_t)qtile * kWaves * kpad * sizeof(uint64_t);
    int nb;
    if (dim == 384) nb = qtile == 4 ? occupancy_of<3, 4, 4>(lds) : qtile == 2 ? occupancy_of<3, 4, 2>(lds) : occupancy_of<3, 4, 1>(lds);
    else if (dim == 768) nb = qtile == 4 ? occupancy_of<6, 2, 4>(lds) : qtile == 2 ? occupancy_of<6, 2, 2>(lds) : occupancy_of<6, 2, 1>(lds);
    else nb = qtile == 4 ? occupancy_of<8, 2, 4>(lds) : qtile == 2 ? occupancy_of<8, 2, 2>(lds) : occupancy_of<8, 2, 1>(lds);
    // One or two queries per pass: TWO resident blocks per CU (8 waves, 96 KiB of loads in flight), not the 5 the
    // occupancy allows.  A read-only stream of the scan's shape reaches 6.96 TB/s with 2 blocks per CU and 6.82
    // with 5 (benchmarks/hbm_read_probe.hip), and the scan follows: 10M x 384, same box, 2.344 -> 2.237 ms at
    // k = 10, 2.399 -> 2.305 at k = 200, two queries per pass 2.62 -> 2.29 ms; 1024-d 3.05 -> 2.97; 768-d within
    // 1 %; 3 and 4 blocks are no better than 5, 1 block is 6 % worse.  Four queries per pass are VALU-heavy and
    // keep the full occupancy (2.67 vs 3.22 ms).  CS_SCAN_BLOCKS_PER_CU overrides (A/B).
    static const int forced = [] {
        const char* e = cs_lab_env("CS_SCAN_BLOCKS_PER_CU");
        return e ? std::atoi(e) : 0;
    }();
    nb = nb > 8 ? 8 : nb;
    const int cap = forced > 0 ? forced : (deep ? 1 : qtile <= 2 ? 2 : 8);
    return nb > cap ? cap : nb;
}

// One query and a short list (k <= 64): ONE block per CU whose waves keep twice the rows in flight (24 x 16-B
// loads per lane: the same 96 KiB per CU from half the waves).  Fewer concurrent streams is what the memory
// system rewards: 10M x 384, same box, k = 10 2.236 -> 2.165 ms (88.7 % of HBM peak with the prime pass
// included), k = 64 2.267 -> 2.199, 1M rows 254 -> 246 us.  Longer lists lose (one wave per SIMD cannot hide the
// inserts: k = 200 2.305 -> 2.336 ms, k = 1024 2.45 -> 3.27), two queries per pass lose badly (2.23 -> 3.55).
// CS_SCAN_DEEP=0 disables, CS_SCAN_DEEP_MAX_K moves the limit.
// Between k = 65 and 128 it is still ahead over multi-million-row corpora (10M rows, k = 100: 2.228 vs 2.265 ms,
// k = 128: 2.250 vs 2.270) and behind at 1M rows (330 vs 320 us), hence the second limit.
static bool scan_deep(uint32_t dim, uint32_t qtile, uint32_t k, uint64_t n_rows) {
    static const int max_k = [] {
        const char* off = cs_lab_env("CS_SCAN_DEEP");
        if (off && off[0] == '0') return 0;
        const char* e = cs_lab_env("CS_SCAN_DEEP_MAX_K");
        return e ? std::atoi(e) : 64;
    }();
    if (!fast_dim(dim) || qtile != 1 || max_k == 0) return false;
    return (int)k <= max_k || ((int)k <= 2 * max_k && n_rows >= 4000000);
}

ScanPlan plan_scan(uint64_t n_rows, uint32_t dim, uint32_t nq, uint32_t k, int num_cus) {
    ScanPlan p{};
    p.kpad = kpad_for(k);
    if (fast_dim(dim)) {
        p.qtile = nq >= 4 ? 4 : (nq >= 2 ? 2 : 1);
        // LDS per block = qtile * 4 waves * kpad * 8 B; stay at >= 2 blocks per CU
        while (p.qtile > 1 && (size_t)p.qtile * kWaves * p.kpad * 8 > 64 * 1024) p.qtile >>= 1;
        p.deep = scan_deep(dim, p.qtile, k, n_rows);
        const uint32_t rows_per_tile = p.deep ? (dim == 384 ? 16 : dim == 768 ? 8 : 6) : (dim == 384) ? 8 : 4;
        const uint64_t ntiles = (n_rows + rows_per_tile - 1) / rows_per_tile;
        const uint64_t blocks = (ntiles + kWaves - 1) / kWaves;
        const uint64_t cap = (uint64_t)num_cus * blocks_per_cu(dim, p.qtile, p.kpad, p.deep);
        p.blocks = (uint32_t)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
    } else {
        p.qtile = 1;
        const uint64_t blocks = (n_rows + kWaves - 1) / kWaves;
        const uint64_t cap = (uint64_t)num_cus * 8;
        p.blocks = (uint32_t)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
    }
    p.passes = (nq + p.qtile - 1) / p.qtile;
    p.partial_keys = (size_t)nq * p.blocks * k;
    p.merge_keys = merge_tmp_keys(p.blocks, nq, k);
    return p;
}

bool scan_prime_supported(uint32_t dim) { return fast_dim(dim); }

// Prime pass geometry.  The bound is the k-th largest of W wave maxima, so W must exceed k by a
// good factor and every wave should see a few tiles: up to k = 256 one block per CU (W <= 1024
// waves); above, four per CU (W <= 4096, the most keys the selecting block's LDS holds) — with
// W = 1024 a k = 1024 bound is the smallest of all maxima, a third of the rows pass it and the scan
// takes 4.9 ms instead of 2.5.  Never more blocks than kpad (waves = 4 * blocks <= 4 * kpad).
static uint32_t prime_block_cap(uint32_t k, int num_cus) {
    const uint32_t kpad = kpad_for(k);
    uint32_t cap = (uint32_t)num_cus * (k > 256 ? 4u : 1u);
    if (cap > 1024) cap = 1024;
    return cap > kpad ? kpad : cap;
}

// Rows of the prime sample: the caller's default, raised to 32 rows per wave when k > 256.
uint64_t prime_sample_rows(uint64_t default_rows, uint32_t k, int num_cus) {
    if (k <= 256) return default_rows;
    const uint64_t want = (uint64_t)prime_block_cap(k, num_cus) * kWaves * 32;
    return want > default_rows ? want : default_rows;
}

ScanPlan plan_prime(uint64_t sample_rows, uint32_t dim, uint32_t nq, uint32_t k, int num_cus) {
    ScanPlan p = plan_scan(sample_rows, dim, nq, k, num_cus);
    const uint32_t cap = prime_block_cap(k, num_cus);
    if (p.blocks > cap) p.blocks = cap;
    p.partial_keys = 0;
    p.merge_keys = 0;
    return p;
}

template <int J, int U, int QT>
static void launch_fast(const ScanPlan& plan, const float* d_corpus, uint64_t n_rows,
                        const float* d_queries, uint32_t nq, uint32_t k, const uint32_t* d_dead,
                        RowIds id_base, uint64_t* d_partial, const ScanPrime* prime,
                        bool prime_pass, hipStream_t stream, const uint32_t* gate) {
    const size_t lds = (size_t)QT * kWaves * plan.kpad * sizeof(uint64_t);
    dim3 grid(plan.blocks, plan.passes);
    if (prime_pass)  // cached loads: the full scan re-reads these rows right after
        hipLaunchKernelGGL((scan_topk_kernel<J, U, QT, false, true>), grid, dim3(kBlock), lds, stream,
                           d_corpus, n_rows, d_queries, nq, k, plan.kpad, d_dead, id_base, nullptr,
                           nullptr, prime->d_wave_max, prime->d_done, prime->d_floor, nullptr);
    else
        hipLaunchKernelGGL((scan_topk_kernel<J, U, QT, true>), grid, dim3(kBlock), lds, stream,
                           d_corpus, n_rows, d_queries, nq, k, plan.kpad, d_dead, id_base, d_partial,
                           prime ? prime->d_floor : nullptr, nullptr, nullptr, nullptr, gate);
}

template <int J, int U>
static void launch_fast_q(const ScanPlan& plan, const float* d_corpus, uint64_t n_rows,
                          const float* d_queries, uint32_t nq, uint32_t k,
                          const uint32_t* d_dead, RowIds id_base, uint64_t* d_partial,
                          const ScanPrime* prime, bool prime_pass, hipStream_t stream, const uint32_t* gate) {
    switch (plan.qtile) {
        case 4: launch_fast<J, U, 4>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate); break;
        case 2: launch_fast<J, U, 2>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate); break;
        default: launch_fast<J, U, 1>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate); break;
    }
}

int32_t launch_scan(const ScanPlan& plan, const float* d_corpus, uint64_t n_rows, uint32_t dim,
                    const float* d_queries, uint32_t nq, uint32_t k, const uint32_t* d_dead,
                    RowIds id_base, uint64_t* d_partial, hipStream_t stream,
                    const ScanPrime* prime, bool prime_pass, const uint32_t* gate) {
    if (prime_pass && (!prime || !fast_dim(dim) || n_rows == 0))
        return fail(CS_ERR_BAD_ARG, "prime pass needs a 384/768/1024-d corpus prefix and its buffers");
    if (n_rows == 0) {  // nothing to score: all-empty partial lists
        if (gate) return CS_OK;  // a gated rerun follows a batched search, which needs rows
        CS_HIP(hipMemsetAsync(d_partial, 0, plan.partial_keys * sizeof(uint64_t), stream));
        return CS_OK;
    }
    if (plan.deep && plan.qtile == 1) {
        if (dim == 384) launch_fast<3, 8, 1>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate);
        else if (dim == 768) launch_fast<6, 4, 1>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate);
        else launch_fast<8, 3, 1>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate);
    } else if (dim == 384) launch_fast_q<3, 4>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate);
    else if (dim == 768) launch_fast_q<6, 2>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate);
    else if (dim == 1024) launch_fast_q<8, 2>(plan, d_corpus, n_rows, d_queries, nq, k, d_dead, id_base, d_partial, prime, prime_pass, stream, gate);
    else {
        const size_t lds = (size_t)kWaves * plan.kpad * sizeof(uint64_t);
        hipLaunchKernelGGL(scan_topk_generic_kernel, dim3(plan.blocks, nq), dim3(kBlock), lds,
                           stream, d_corpus, n_rows, dim, d_queries, nq, k, plan.kpad, d_dead,
                           id_base, d_partial, gate);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

static uint32_t merge_group(uint32_t k) {
    // lists merged per block: as many as kMergeCap (4,096) keys hold — the block brackets the k-th key by bisection
    // and sorts only the keys above it (block_select.hpp), so 256 block lists x k = 10 are ONE launch.  (When the block
    // sorted everything it loaded, short lists went through 512-key groups and a second level: 20 us for the two
    // launches against 36 us for one 4,096-key sort.)  Never fewer than 2.  CS_MERGE_GROUP_KEYS overrides (A/B).
    static const uint32_t cap = [] {
        const char* e = cs_lab_env("CS_MERGE_GROUP_KEYS");
        const int v = e ? std::atoi(e) : 0;
        return (v >= 64 && v <= kMergeCap) ? (uint32_t)v : (uint32_t)kMergeCap;
    }();
    const uint32_t g = cap / k;
    return g < 2 ? 2 : g;
}

size_t merge_tmp_keys(uint32_t nlists, uint32_t nq, uint32_t k) {
    const uint32_t G = merge_group(k);
    const uint32_t ngroups = (nlists + G - 1) / G;
    return ngroups > 1 ? (size_t)nq * ngroups * k : 0;
}

int32_t launch_merge(const uint64_t* d_lists, uint32_t nlists, uint32_t nq, uint32_t k,
                     bool list_major, uint64_t* d_tmp_a, uint64_t* d_tmp_b, uint64_t* d_out_keys, float* d_out_cos,
                     uint32_t* d_out_ids, uint32_t* d_out_counts, hipStream_t stream, const uint32_t* gate,
                     uint32_t remap_stripe, uint32_t remap_shards) {
    const uint32_t G = merge_group(k);
    const uint64_t* in = d_lists;
    uint64_t* bufs[2] = {d_tmp_a, d_tmp_b};
    int flip = 0;
    uint64_t q_stride = list_major ? k : (uint64_t)nlists * k;
    uint64_t l_stride = list_major ? (uint64_t)nq * k : k;
    for (;;) {
        const uint32_t ngroups = (nlists + G - 1) / G;
        const bool final_pass = ngroups == 1;
        uint64_t* out = final_pass ? d_out_keys : bufs[flip];
        if (!final_pass && !out) return fail(CS_ERR_BAD_ARG, "merge scratch missing");
        hipLaunchKernelGGL(merge_topk_kernel, dim3(ngroups, nq), dim3(kMergeBlock), 0, stream, in,
                           nlists, k, G, q_stride, l_stride, out, d_out_cos, d_out_ids, d_out_counts, gate,
                           remap_stripe, remap_shards);
        CS_HIP(hipGetLastError());
        remap_stripe = 0;  // ids are global after the first level
        if (final_pass) break;
        in = out;
        nlists = ngroups;
        q_stride = (uint64_t)nlists * k;
        l_stride = k;
        flip ^= 1;
    }
    return CS_OK;
}

int32_t launch_synth_fill(float* d_rows, uint64_t n, uint32_t dim, uint64_t seed,
                          uint64_t first_row, hipStream_t stream) {
    if (n == 0) return CS_OK;
    const uint64_t total = n * dim;
    uint64_t blocks = (total / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(synth_fill_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_rows,
                       total, seed, first_row * dim);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
