// scan.hpp — launch interface of the cosine scan + top-k kernels (scan.hip).
#pragma once

#include "common.hpp"

namespace cs {

struct ScanPlan {
    uint32_t blocks;     // scan grid.x
    uint32_t kpad;       // per-wave list capacity (power of two >= k, >= 64)
    uint32_t qtile;      // queries handled per scan pass
    uint32_t passes;     // ceil(nq / qtile)  (grid.y)
    bool deep;           // one block per CU with twice the rows in flight per wave (one query, k <= 64)
    size_t partial_keys; // u64 count needed for the scan's partial buffer
    size_t merge_keys;   // u64 count needed for the merge ping-pong buffer
};

// Geometry for a search of nq queries / top-k over n_rows rows of `dim` floats.
ScanPlan plan_scan(uint64_t n_rows, uint32_t dim, uint32_t nq, uint32_t k, int num_cus);

// Buffers of a primed scan (scan.hip, PRIME mode): a pass over a corpus prefix leaves in
// d_floor[q] a lower bound of query q's k-th best cosine, which the full scan starts from.
struct ScanPrime {
    float* d_wave_max = nullptr;  // [nq][4 * plan_prime().blocks]
    uint32_t* d_done = nullptr;   // [passes], zero between launches
    float* d_floor = nullptr;     // [nq]
};
bool scan_prime_supported(uint32_t dim);
uint64_t prime_sample_rows(uint64_t default_rows, uint32_t k, int num_cus);
ScanPlan plan_prime(uint64_t sample_rows, uint32_t dim, uint32_t nq, uint32_t k, int num_cus);

// Scores every live row of corpus[0..n_rows) against each query and leaves, per
// (query, block), the block's best k as packed keys in d_partial[q][block][k].
// prime + prime_pass: run the prime pass over these rows instead (plan from plan_prime, no
// partial lists written);  prime alone: start the lists from prime->d_floor.
int32_t launch_scan(const ScanPlan& plan, const float* d_corpus, uint64_t n_rows, uint32_t dim,
                    const float* d_queries, uint32_t nq, uint32_t k, const uint32_t* d_dead,
                    RowIds id_base, uint64_t* d_partial, hipStream_t stream,
                    const ScanPrime* prime = nullptr, bool prime_pass = false, const uint32_t* gate = nullptr);
// `gate` (launch_scan and launch_merge): device word; when non-null every block of the launch exits at
// once unless *gate != 0 — the exact rerun enqueued behind a batched search on the device API.

// Reduces nlists lists of k keys per query ([nq][nlists][k], or [nlists][nq][k] when
// list_major) to the best k per query,
// writing keys / decoded cosines / ids / counts (each optional).  d_tmp: ping-pong
// scratch of plan.merge_keys (may be null when nlists*k <= 2048).
int32_t launch_merge(const uint64_t* d_lists, uint32_t nlists, uint32_t nq, uint32_t k,
                     bool list_major, uint64_t* d_tmp_a, uint64_t* d_tmp_b, uint64_t* d_out_keys, float* d_out_cos,
                     uint32_t* d_out_ids, uint32_t* d_out_counts, hipStream_t stream,
                     const uint32_t* gate = nullptr, uint32_t remap_stripe = 0, uint32_t remap_shards = 0);
// remap_stripe != 0: list l holds shard l's local row numbers; they become global ids while the lists are
// read (striped row sharding, shards.hip).
size_t merge_tmp_keys(uint32_t nlists, uint32_t nq, uint32_t k);
// cs_merge_topk_device with the striped-shard id remap of launch_merge (index.hip; pooled scratch, asynchronous)
int32_t merge_topk_device_impl(int32_t device, const uint64_t* d_keys, uint32_t nlists, uint32_t nq, uint32_t k,
                               uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_counts,
                               hipStream_t stream, uint32_t remap_stripe, uint32_t remap_shards);

void merge_scratch_release(int device, hipStream_t stream);  // frees the pooled multi-level merge scratch of a stream
// shards.hip: make room for `rows` rows in all (the only step of an append that can run out of memory), and append
// rows resident on `src_device` with one asynchronous copy on `stream` (a stream of src_device).
int32_t index_reserve(cs_index* h, uint64_t rows);
int32_t index_append_from(cs_index* h, const float* d_rows, int src_device, uint64_t n, hipStream_t stream);

// Variant merge of search::search (src/search/mod.rs:513-611): keys [nv][k] -> the best `limit` distinct ids
// (a chunk keeps its best key), best-first, + count + the "top five all within distance 0.15" predicate.
int32_t launch_merge_variants(const uint64_t* d_keys, uint32_t nv, uint32_t k, uint32_t limit, uint64_t* d_out_keys,
                              float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_count,
                              uint32_t* d_out_high_confidence, hipStream_t stream);

// corpus[(first_out_row + r) * dim + c] = cs_synth_value(seed, (first_row + r) * dim + c)
int32_t launch_synth_fill(float* d_rows, uint64_t n, uint32_t dim, uint64_t seed,
                          uint64_t first_row, hipStream_t stream);

// ---- batched-query (MFMA) path, scan_mfma.hip ---------------------------------------------
// Candidate counters sit one per 128-byte line: appends are device-scope atomics, and ops on one
// line serialise at ~40 ns apiece whichever word they hit (measured: nine queries' counters in one
// line made the k=200 filter phases 50-190 us longer).
constexpr uint32_t kCntStride = 32;
// Device-API searches of up to this many queries carry a gated exact rerun behind the filter path (index.hip
// run_search); above it an overflowed candidate buffer is reported through the sticky word of BatchedState.
constexpr uint32_t kGatedMaxQ = 16;
// Phase 0 of the filter searches (scan_filter.hip): tau is still -inf, so the first rows are not filtered at all — every
// one of them is re-scored exactly and folded into the running best-k.  3,072 = a multiple of the filter kernels' 1,024
// row granule that, with the k <= 1,024 carried keys, still sorts as ONE 4,096-key chunk of select_candidates_kernel;
// round 3 used 1,024, which cost one more phase (filter + re-score + select: ~45 us of launches) on a 10M-row index.
constexpr uint32_t kFilterPhase0 = 3072;
struct BatchedState {
    uint64_t* d_cand = nullptr;   // [nq][cap] candidate keys
    uint32_t* d_cnt = nullptr;    // [nq][kCntStride], word 0 of each line used
    float* d_tau = nullptr;       // [nq]
    uint64_t* d_carry = nullptr;  // [nq][k]
    // [0] = a candidate buffer overflowed during the current search (reset by the search's first kernel);
    // [1] = sticky: set with [0], cleared only by cs_index_search_status; [2] = overflowed searches counted so far
    // (the first kernel of the NEXT search folds [0] into it)
    uint32_t* d_overflow = nullptr;
    // pinned host word the first kernel of every search copies [2] into: lets the host notice overflowed device-API
    // searches (which it never waits for) a search or two later, without a synchronisation (index.hip: int8 strikes)
    uint32_t* h_mirror = nullptr;
};
// One block per query folds st.d_cand[q][0..cnt[q]) (packed keys) into st.d_carry[q][k], sets
// tau[q] to the k-th best cosine and resets cnt[q]; with `last` it also writes the outputs.
int32_t launch_select_candidates(const BatchedState& st, uint32_t nq, uint32_t cap, uint32_t k, bool last,
                                 uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids,
                                 uint32_t* d_out_counts, hipStream_t stream);
bool batched_supported(uint32_t dim);
uint32_t batched_cap(uint32_t k);
int32_t launch_row_norms(const float* d_corpus, uint64_t first, uint64_t n, uint32_t dim,
                         float* d_norms, hipStream_t stream);
// Exact unless *st.d_overflow != 0 afterwards (then rerun on the list-based scan).
int32_t launch_scan_batched(const BatchedState& st, const float* d_corpus, const float* d_norms,
                            uint64_t n_rows, uint32_t dim, const float* d_queries, uint32_t nq, uint32_t k,
                            const uint32_t* d_dead, RowIds id_base, int num_cus, uint64_t* d_out_keys,
                            float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_counts,
                            hipStream_t stream);

// ---- batched-query filter-and-refine path (f16 unit-vector filter + exact refine), scan_filter.hip
struct SplitQueryWs {
    _Float16* d_qsplit = nullptr;  // [nq][dim] f16: q / |q|
    float* d_qmag = nullptr;       // [nq]
    // Host-buffer searches: the queries still sit in pinned host memory (device-addressable) and
    // d_queries is an empty device buffer — the prep kernel reads them from here and fills it,
    // which saves the H2D copy call (~6 us of a 45-us search over a small corpus).  Null otherwise.
    const float* q_pinned = nullptr;
    int8_t* d_q8q = nullptr;       // [nq][dim] int8: q / |q| on the query's own scale (int8 filter copy)
    float4* d_qmeta = nullptr;     // [4 nq]: [q] = {127 / max |q_i / |q||, 0.5001 sum |b_i|, q / |q| . mu, sqrt(sum b_i^2)},
                                   // [nq + q].x = sqrt(sum d_i^2), d = the query's rounding errors (q8_threshold);
                                   // [2 nq + q], [3 nq + q]: the same for the two-plane (128 times finer) quantisation
    int8_t* d_q8q_hi = nullptr;    // [nq][dim] x 2: the query as 128 hi + lo, both int8 (up to 64 queries)
    int8_t* d_q8q_lo = nullptr;
};
// int8 filter copy of the corpus (scan_filter.hip, "int8 filter copy"): complete 128-row tiles [0, rows / 128)
struct Q8View {
    const int8_t* d_q8 = nullptr;    // [tile][dim / 128][128 rows][128 B]
    const float4* d_tmeta = nullptr; // [tile] {127 / max |u - mu|, 0.5001 max row sum |a_i|, max row error norm, max row norm of a};
                                     // x = NaN: every row is a candidate
    const float* d_mu = nullptr;     // [dim] the mean unit row the copy is centred on (q.u = q.(u - mu) + q.mu)
    uint64_t rows = 0;               // multiple of 128
};
bool split_scan_supported(uint32_t dim);
// rows [first_tile * 128, (first_tile + ntiles) * 128) of the f32 corpus -> int8 tiles + their scales
int32_t launch_corpus_q8(const float* d_corpus, const float* d_norms, int8_t* d_q8, float4* d_tmeta, uint64_t first_tile,
                         uint64_t ntiles, uint32_t dim, const float* d_mu, hipStream_t stream);
// d_mu[dim] = mean over rows [0, n) of x / |x|
int32_t launch_unit_mean(const float* d_corpus, const float* d_norms, uint64_t n, uint32_t dim, float* d_mu,
                         hipStream_t stream);
// rows [first, first+n) of the f32 corpus, divided by their norms, as f16 -> the filter copy
// [rows][dim] (same row order; half the bytes of the f32 matrix)
int32_t launch_corpus_split(const float* d_corpus, const float* d_norms, _Float16* d_split, uint64_t first,
                            uint64_t n, uint32_t dim, hipStream_t stream);
// Exact (bit-identical to launch_scan + launch_merge) unless *st.d_overflow != 0 afterwards.
int32_t launch_scan_split(const BatchedState& st, const SplitQueryWs& qw, const float* d_corpus,
                          const _Float16* d_split, uint64_t n_rows, uint32_t dim, const float* d_queries, uint32_t nq, uint32_t k, const uint32_t* d_dead,
                          RowIds id_base, uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids,
                          uint32_t* d_out_counts, hipStream_t stream, float margin, const Q8View* q8 = nullptr);
// Proven bound of |filter cosine - exact cosine| for unit vectors of this width (scan_filter.hip header);
// `subnormals_exact` = the f16 MFMA consumes subnormal inputs exactly (sh_denorm_selftest).
float filter_margin(uint32_t dim, bool subnormals_exact);
int32_t sh_denorm_selftest(bool* ok, hipStream_t s);  // gemm_split.hip

}  // namespace cs
