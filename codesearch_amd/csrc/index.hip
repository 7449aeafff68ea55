// index.hip — cs_index_*: the vector half of the reference's VectorStore
// (/root/reference/src/vectordb/store.rs:94-750) as a device-resident row-major matrix
// searched by the exact scan of scan.hip.  Metadata (store.rs:19-85) stays with the caller.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include <cstdlib>

#include "scan.hpp"

namespace cs {

std::string& last_error_ref() {
    static thread_local std::string msg;
    return msg;
}

int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

struct EventTriple {
    hipEvent_t e0, e1, e2;
};

// Per-call scratch.  Host-API calls borrow one from the pool (own stream); device-API
// calls use the one bound to the caller's stream (stream order makes reuse safe).
struct Workspace {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t* d_partial = nullptr; size_t partial_cap = 0;
    uint64_t* d_tmp_a = nullptr; uint64_t* d_tmp_b = nullptr; size_t tmp_cap = 0;
    ScanPrime prime; size_t prime_max_cap = 0, prime_nq_cap = 0;
    float* d_queries = nullptr; size_t q_cap = 0;
    uint64_t* d_keys = nullptr; float* d_cos = nullptr; uint32_t* d_ids = nullptr; size_t out_cap = 0;
    uint32_t* d_counts = nullptr; size_t cnt_cap = 0;
    // pinned staging of the host-buffer API: queries in, packed keys out (ONE copy each way; the host
    // unpacks cosine / id / count from the keys — three more small D2H copies cost ~8 us apiece)
    float* h_queries = nullptr; size_t h_q_cap = 0;
    uint64_t* h_keys = nullptr; size_t h_out_cap = 0;
    uint32_t* h_variant_meta = nullptr;  // pinned: [0] count, [1] high-confidence flag (cs_index_search_variants)
    std::vector<EventTriple> free_events;
    BatchedState bs;
    size_t bs_nq = 0, bs_cand = 0, bs_carry = 0;
    SplitQueryWs qw;
    size_t qw_elems = 0, qw_nq = 0, q8_elems = 0;
    uint32_t* h_overflow = nullptr;
    uint32_t mirror_seen = 0;  // value of h_overflow[3] already accounted for
    bool last_via_q8 = false;  // the previous filter search on this workspace read the int8 copy (strike bookkeeping)

    // want_q8: the int8 copy is this search's filter operand — its three query planes are allocated only then, and
    // *q8_ok = false (never an error) when they do not fit: the caller filters on the f16 copy instead
    int32_t reserve_split_queries(uint32_t nq, uint32_t dim, bool want_q8, bool* q8_ok) {
        const size_t elems = (size_t)nq * dim;
        *q8_ok = false;
        if (elems > qw_elems) {
            if (qw.d_qsplit) (void)hipFree(qw.d_qsplit);
            qw.d_qsplit = nullptr; qw_elems = 0;
            CS_HIP(hipMalloc(&qw.d_qsplit, elems * sizeof(_Float16)));
            qw_elems = elems;
        }
        if (want_q8 && elems > q8_elems) {
            if (qw.d_q8q) (void)hipFree(qw.d_q8q);
            if (qw.d_q8q_hi) (void)hipFree(qw.d_q8q_hi);
            qw.d_q8q = qw.d_q8q_hi = qw.d_q8q_lo = nullptr; q8_elems = 0;
            if (hipMalloc(&qw.d_q8q, elems) != hipSuccess || hipMalloc(&qw.d_q8q_hi, 2 * elems) != hipSuccess) {
                (void)hipGetLastError();
                if (qw.d_q8q) (void)hipFree(qw.d_q8q);
                qw.d_q8q = nullptr;
            } else {
                qw.d_q8q_lo = qw.d_q8q_hi + elems;
                q8_elems = elems;
            }
        }
        *q8_ok = want_q8 && elems <= q8_elems;
        if (nq > qw_nq) {
            if (qw.d_qmag) (void)hipFree(qw.d_qmag);
            if (qw.d_qmeta) (void)hipFree(qw.d_qmeta);
            qw.d_qmag = nullptr; qw.d_qmeta = nullptr; qw_nq = 0;
            CS_HIP(hipMalloc(&qw.d_qmag, nq * sizeof(float)));
            CS_HIP(hipMalloc(&qw.d_qmeta, 4 * (size_t)nq * sizeof(float4)));
            qw_nq = nq;
        }
        return CS_OK;
    }

    int32_t reserve_prime(const ScanPlan& pp, uint32_t nq) {
        const size_t nmax = (size_t)nq * pp.blocks * 4;
        if (nmax > prime_max_cap) {
            if (prime.d_wave_max) (void)hipFree(prime.d_wave_max);
            prime.d_wave_max = nullptr; prime_max_cap = 0;
            CS_HIP(hipMalloc(&prime.d_wave_max, nmax * sizeof(float)));
            prime_max_cap = nmax;
        }
        if (nq > prime_nq_cap) {  // passes <= nq
            if (prime.d_done) (void)hipFree(prime.d_done);
            if (prime.d_floor) (void)hipFree(prime.d_floor);
            prime.d_done = nullptr; prime.d_floor = nullptr; prime_nq_cap = 0;
            CS_HIP(hipMalloc(&prime.d_done, nq * sizeof(uint32_t)));
            CS_HIP(hipMalloc(&prime.d_floor, nq * sizeof(float)));
            CS_HIP(hipMemset(prime.d_done, 0, nq * sizeof(uint32_t)));
            prime_nq_cap = nq;
        }
        return CS_OK;
    }

    int32_t reserve_batched(uint32_t nq, uint32_t k) {
        const size_t cand = (size_t)nq * batched_cap(k), carry = (size_t)nq * k;
        if (!h_overflow) {
            CS_HIP(hipHostMalloc(&h_overflow, 4 * sizeof(uint32_t)));
            h_overflow[3] = 0;
            bs.h_mirror = h_overflow + 3;  // written by the device (scan.hpp BatchedState::h_mirror)
        }
        if (!bs.d_overflow) {  // [0] this search, [1] sticky, [2] overflowed searches so far (scan.hpp BatchedState)
            CS_HIP(hipMalloc(&bs.d_overflow, 4 * sizeof(uint32_t)));
            CS_HIP(hipMemset(bs.d_overflow, 0, 4 * sizeof(uint32_t)));
        }
        if (nq > bs_nq) {
            if (bs.d_cnt) (void)hipFree(bs.d_cnt);
            if (bs.d_tau) (void)hipFree(bs.d_tau);
            bs.d_cnt = nullptr; bs.d_tau = nullptr; bs_nq = 0;
            CS_HIP(hipMalloc(&bs.d_cnt, (size_t)nq * kCntStride * sizeof(uint32_t)));
            CS_HIP(hipMalloc(&bs.d_tau, nq * sizeof(float)));
            bs_nq = nq;
        }
        if (cand > bs_cand) {
            if (bs.d_cand) (void)hipFree(bs.d_cand);
            bs.d_cand = nullptr; bs_cand = 0;
            CS_HIP(hipMalloc(&bs.d_cand, cand * sizeof(uint64_t)));
            bs_cand = cand;
        }
        if (carry > bs_carry) {
            if (bs.d_carry) (void)hipFree(bs.d_carry);
            bs.d_carry = nullptr; bs_carry = 0;
            CS_HIP(hipMalloc(&bs.d_carry, carry * sizeof(uint64_t)));
            bs_carry = carry;
        }
        return CS_OK;
    }

    int32_t reserve(const ScanPlan& p, uint32_t nq, uint32_t dim, uint32_t k, bool host_io) {
        if (p.partial_keys > partial_cap) {
            if (d_partial) (void)hipFree(d_partial);
            d_partial = nullptr; partial_cap = 0;
            CS_HIP(hipMalloc(&d_partial, p.partial_keys * sizeof(uint64_t)));
            partial_cap = p.partial_keys;
        }
        if (p.merge_keys > tmp_cap) {
            if (d_tmp_a) (void)hipFree(d_tmp_a);
            if (d_tmp_b) (void)hipFree(d_tmp_b);
            d_tmp_a = d_tmp_b = nullptr; tmp_cap = 0;
            CS_HIP(hipMalloc(&d_tmp_a, p.merge_keys * sizeof(uint64_t)));
            CS_HIP(hipMalloc(&d_tmp_b, p.merge_keys * sizeof(uint64_t)));
            tmp_cap = p.merge_keys;
        }
        if (!host_io) return CS_OK;
        const size_t qn = (size_t)nq * dim, on = (size_t)nq * k;
        if (qn > q_cap) {
            if (d_queries) (void)hipFree(d_queries);
            d_queries = nullptr; q_cap = 0;
            CS_HIP(hipMalloc(&d_queries, qn * sizeof(float)));
            q_cap = qn;
        }
        if (on > out_cap) {
            if (d_keys) (void)hipFree(d_keys);
            if (d_cos) (void)hipFree(d_cos);
            if (d_ids) (void)hipFree(d_ids);
            d_keys = nullptr; d_cos = nullptr; d_ids = nullptr; out_cap = 0;
            CS_HIP(hipMalloc(&d_keys, on * sizeof(uint64_t)));
            CS_HIP(hipMalloc(&d_cos, on * sizeof(float)));
            CS_HIP(hipMalloc(&d_ids, on * sizeof(uint32_t)));
            out_cap = on;
        }
        if (nq > cnt_cap) {
            if (d_counts) (void)hipFree(d_counts);
            d_counts = nullptr; cnt_cap = 0;
            CS_HIP(hipMalloc(&d_counts, nq * sizeof(uint32_t)));
            cnt_cap = nq;
        }
        if (on > h_out_cap) {
            if (h_keys) (void)hipHostFree(h_keys);
            h_keys = nullptr; h_out_cap = 0;
            CS_HIP(hipHostMalloc(&h_keys, on * sizeof(uint64_t)));
            h_out_cap = on;
        }
        if ((size_t)nq * dim > h_q_cap) {
            if (h_queries) (void)hipHostFree(h_queries);
            h_queries = nullptr; h_q_cap = 0;
            CS_HIP(hipHostMalloc(&h_queries, (size_t)nq * dim * sizeof(float)));
            h_q_cap = (size_t)nq * dim;
        }
        return CS_OK;
    }

    void release_all() {
        if (d_partial) (void)hipFree(d_partial);
        if (prime.d_wave_max) (void)hipFree(prime.d_wave_max);
        if (prime.d_done) (void)hipFree(prime.d_done);
        if (prime.d_floor) (void)hipFree(prime.d_floor);
        if (d_tmp_a) (void)hipFree(d_tmp_a);
        if (d_tmp_b) (void)hipFree(d_tmp_b);
        if (d_queries) (void)hipFree(d_queries);
        if (d_keys) (void)hipFree(d_keys);
        if (d_cos) (void)hipFree(d_cos);
        if (d_ids) (void)hipFree(d_ids);
        if (d_counts) (void)hipFree(d_counts);
        if (h_keys) (void)hipHostFree(h_keys);
        if (h_queries) (void)hipHostFree(h_queries);
        if (h_overflow) (void)hipHostFree(h_overflow);
        if (h_variant_meta) (void)hipHostFree(h_variant_meta);
        if (bs.d_cand) (void)hipFree(bs.d_cand);
        if (bs.d_cnt) (void)hipFree(bs.d_cnt);
        if (bs.d_tau) (void)hipFree(bs.d_tau);
        if (bs.d_carry) (void)hipFree(bs.d_carry);
        if (bs.d_overflow) (void)hipFree(bs.d_overflow);
        if (qw.d_qsplit) (void)hipFree(qw.d_qsplit);
        if (qw.d_qmag) (void)hipFree(qw.d_qmag);
        if (qw.d_q8q) (void)hipFree(qw.d_q8q);
        if (qw.d_q8q_hi) (void)hipFree(qw.d_q8q_hi);
        if (qw.d_qmeta) (void)hipFree(qw.d_qmeta);
        for (auto& t : free_events) {
            (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); (void)hipEventDestroy(t.e2);
        }
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace cs

using namespace cs;

struct cs_index {
    int device = 0;
    int num_cus = 256;
    uint32_t dim = 0;
    uint32_t id_base = 0;
    uint64_t capacity = 0;   // rows allocated
    uint64_t n_rows = 0;     // rows in storage
    uint64_t n_ids = 0;      // ids issued (next_id - id_base): n_rows until cs_index_build first reclaims deleted rows
    uint64_t n_removed = 0;  // tombstoned rows still in storage
    // Reclaiming deleted rows (store.rs:548-610: arroy drops deleted items at the next build; the incremental `index` deletes a
    // changed file's chunks and re-inserts them, src/index/mod.rs:525,544 — a store re-indexed daily would otherwise only grow):
    // when at least compact_dead_pct % of the stored rows are tombstones, cs_index_build rewrites corpus (norms and the filter
    // copies are rebuilt from it) without them.  Ids stay what they were: h_ids / d_ids = the id of each stored row,
    // ascending; empty = never compacted, id = id_base + row.  (CS_INDEX_COMPACT_DEAD_PCT, default 10; 0 = never.)
    std::vector<uint32_t> h_ids;
    uint32_t* d_ids = nullptr;
    uint64_t ids_cap = 0, ids_uploaded = 0;
    uint32_t compact_dead_pct = 10;
    uint64_t compactions = 0;
    RowIds row_ids() const { return RowIds(id_base, h_ids.empty() ? nullptr : d_ids); }
    float* d_corpus = nullptr;
    uint32_t* d_dead = nullptr;  // bitmap over rows, sized for `capacity`
    float* d_norms = nullptr;    // |row| for rows [0, normed_rows) (batched-query path)
    uint64_t normed_rows = 0;
    // unit rows [0, split_rows) as f16 [row][dim]: the filter operand of the batched path WHEN the int8 copy does not
    // serve (no int8 copy, retired by its spread or by strikes, <= 1024 rows).  Built on demand (ensure_f16): an index
    // whose int8 copy serves holds 5 bytes per element (f32 + int8), not 7, and its builds skip the conversion.
    _Float16* d_split = nullptr;
    uint64_t split_rows = 0, split_cap = 0;
    bool use_split = false;
    bool f16_eager = false;    // CS_FILTER_F16_EAGER=1 (or an A/B knob that needs both copies): build it at every cs_index_build
    bool f16_failed = false;   // no room for it: searches stay on the int8 copy / the exact paths (reset by clear())
    std::mutex filter_mu;      // guards the on-demand build (searches are re-entrant)
    // int8 filter copy (scan_filter.hip): complete 128-row tiles [0, q8_rows / 128), a quarter of the f32 bytes
    int8_t* d_q8 = nullptr;
    float4* d_tmeta = nullptr;
    float* d_mu = nullptr;   // [dim] mean unit row of the first build: the copy holds u - mu (fixed until clear())
    uint64_t q8_rows = 0;
    bool use_q8 = false;
    // The int8 copy stops being the filter operand (the f16 copy takes over, results unchanged) when the data defeat its
    // error band: at build, when the tiles' scales say so (outlier coordinates: q8_spread), and at run time after two
    // searches through it overflowed a candidate buffer.  clear() resets both.
    std::atomic<bool> q8_active{true};
    std::atomic<uint32_t> q8_strikes{0}, q8_searches{0};
    // a strike; the copy is retired once there are two and they are more than one in sixteen of its searches
    void q8_strike() {
        const uint32_t s = q8_strikes.fetch_add(1) + 1;
        if (s >= 2 && (uint64_t)s * 16 >= q8_searches.load()) q8_active.store(false);
    }
    float q8_spread = 0.0f;      // median over tiles of max |u - mu| * sqrt(dim) at the last build (isotropic rows: ~4.4)
    float q8_max_spread = 7.0f;  // CS_FILTER_INT8_MAX_SPREAD
    float filter_margin = 0.0f;  // scan_filter.hip: bound of the f16 filter's error for this dim
    int filter_min_q = 2;  // query count from which the f16 filter + exact refine path is used
    uint32_t single_filter_min_k = 100;  // ... and one query too from this k on, over >= 2M rows (0 = never)
    // One query: CS_ROUTE_COST (default) takes the filter over >= single_int8_min_rows rows whenever the int8 copy serves
    // (same bits, 0.66 vs 2.16 ms over 10M x 384 at k = 10: the filter streams a quarter of the bytes), and from
    // single_filter_min_k on over >= single_filter_min_rows rows with the f16 copy; CS_ROUTE_STREAM always runs the f32 streaming scan (the north-star
    // kernel: bench.py selects it for `value`); CS_ROUTE_FILTER takes the filter whenever a copy can serve.
    int single_route = CS_ROUTE_COST;
    uint64_t single_filter_min_rows = 2000000;  // ... with the f16 copy (and k >= single_filter_min_k)
    // ... with the int8 copy: the measured crossover of the two routes, which depends on the list length because the
    // filter's round plan does (scan_filter.hip: growth up to 24 - one round up to 60 x 3,072 rows - below k = 48, 5.5 from
    // there on).  profiles/r04_route_crossover_by_k.log, us per search, stream / filter: k = 10: 20k rows 54 / 57, 35k 63 / 58,
    // 100k 83 / 65, 184k 105 / 72; k = 25: 35k 71 / 63, 100k 104 / 86; k = 40: 200k 158 / 102 — k = 50: 150k 97 / 114, 300k 129 / 127,
    // 400k 150 / 131; k = 75: 300k 141 / 135; k = 99: 300k 142 / 141.  (Round 4's first figure, 150,000 rows for every k, was
    // taken before the phase plan and the one-round phase 0.)
    uint64_t single_int8_min_rows = 32768;        // k < 48 (CS_FILTER_SINGLE_MIN_ROWS)
    uint64_t single_int8_min_rows_long = 300000;  // k >= 48 (CS_FILTER_SINGLE_MIN_ROWS_LONG)
    uint64_t few_queries_min_rows = 40000;        // two or three queries: rows from which they take the filter (CS_FILTER_FEW_MIN_ROWS) ...
    uint64_t few_queries_min_rows_short = 16384;  // ... with k <= 16 (both follow CS_FILTER_FEW_MIN_ROWS when it is set)
    uint64_t single_batched_max_rows = 1024;  // ... and one query over at most this many rows (0 = never; CS_SINGLE_BATCHED_MAX_ROWS)
    // primed streaming scan (scan.hip PRIME mode): from this k and this many rows on, a pass over
    // the first prime_rows rows bounds the list inserts of the full scan
    // (measured, 1 query x 384-d: 10M rows k=10 2.37 -> 2.31 ms, k=200 2.62 -> 2.41 ms; 1M rows
    // k=200 382 -> 279 us).  prime_rows 0 = n_rows / 256 clamped to [4096, 16384]; prime_min_rows 0 =
    // 500,000 rows below k = 48 and 100,000 from there on.
    uint32_t prime_min_k = 1;
    uint64_t prime_min_rows = 0, prime_rows = 0;
    uint64_t batched_searches = 0, batched_fallbacks = 0, q8_reruns = 0;
    std::vector<uint32_t> h_dead;
    bool built = false;
    // streams of OTHER devices that carry unfinished appends into this corpus (index_append_from: an encoder replica on
    // another GPU writing its rows over xGMI); hipDeviceSynchronize on this device does not wait for them
    // peer appends in flight: ONE event per (source device, stream), re-recorded by every append on that stream (a later
    // record covers the stream's earlier copies), so an ingest of millions of rows in mini-batches keeps a handful of events.
    // CONTRACT (include/codesearch_gpu.h, cs_shards_add_device / cs_embedders_index_*): a stream that carried an append must
    // stay alive until the next build / search / read of this index has drained it — a destroyed stream whose handle value is
    // handed out again would have its slot's event re-recorded on the NEW stream and the old copies would go untracked
    struct ForeignAppend { int device; hipStream_t stream; hipEvent_t done; };
    std::vector<ForeignAppend> foreign_appends;

    std::mutex mu;  // guards the pools below (search is re-entrant)
    std::vector<Workspace*> pool;
    // device-API scratch: one set per (stream, calling thread) — stream order makes reuse safe within a
    // thread, and two threads sharing a stream (e.g. both on the null stream) never share a set
    std::map<std::pair<hipStream_t, std::thread::id>, Workspace*> by_stream;
    bool profile = false;
    std::vector<EventTriple> pending;
    double scan_ms = 0.0, merge_ms = 0.0;
    uint64_t scan_launches = 0;
};

namespace {

// Appends through cs_index_add_device / index_append_from may still be in flight on caller streams that do not
// order against the null stream (hipStreamNonBlocking, torch side streams) or that belong to another device.
int32_t drain_appends(cs_index* h) {
    int32_t st = CS_OK;
    for (const auto& fe : h->foreign_appends) {
        DeviceGuard g(fe.device);
        if (hipEventSynchronize(fe.done) != hipSuccess && st == CS_OK)
            st = fail(CS_ERR_HIP, "a peer append into the index did not complete: %s", hipGetErrorString(hipGetLastError()));
        (void)hipEventDestroy(fe.done);
    }
    h->foreign_appends.clear();
    // (the index's own device is drained whether or not a peer reported a failure)
    if (hipDeviceSynchronize() != hipSuccess && st == CS_OK)
        st = fail(CS_ERR_HIP, "hipDeviceSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
    return st;
}

// New buffers of a grow(), freed on every path that does not commit them (a failing copy between the
// allocations and the pointer swap used to return with nc / nd / nn / n8 / nm still allocated).
struct GrowBuffers {
    float* nc = nullptr;      // corpus
    uint32_t* nd = nullptr;   // tombstone bitmap
    float* nn = nullptr;      // row norms
    _Float16* ns = nullptr;   // f16 filter copy
    int8_t* n8 = nullptr;     // int8 filter copy
    float4* nm = nullptr;     // its tile metadata
    ~GrowBuffers() {
        for (void* p : {(void*)nc, (void*)nd, (void*)nn, (void*)ns, (void*)n8, (void*)nm})
            if (p) (void)hipFree(p);
    }
};

// CS_FAULT_GROW_COPY=<stage> (tests): the copy of that stage reports a failure — 1 f16 copy, 2 int8 copy, 3 bitmap clear,
// 4 norms, 5 corpus, 6 bitmap upload
static bool grow_fault(int stage, const cs_index* h) {
    const char* e = cs_lab_env("CS_FAULT_GROW_COPY");  // (read per call: a grow is rare, and tests set it mid-process)
    return e && std::atoi(e) == stage && h->capacity != 0;
}
#define CS_GROW_COPY(stage, call)                                                                            \
    do {                                                                                                     \
        const hipError_t _e = grow_fault(stage, h) ? hipErrorUnknown : (call);                               \
        if (_e != hipSuccess)                                                                                \
            return fail(CS_ERR_HIP, "growing the index: copy stage %d failed: %s", stage, hipGetErrorString(_e)); \
    } while (0)

// Transactional: every new buffer is allocated and filled first; the handle's pointers are swapped (and the old buffers
// freed) only after the last copy succeeded.  A failure leaves the index exactly as it was and frees what was allocated.
// Two degradations are not failures: without room for a larger f16 / int8 filter copy that copy is dropped (searches
// rebuild the f16 copy on demand or stay on the exact paths).
int32_t grow(cs_index* h, uint64_t need_rows) {
    if (need_rows <= h->capacity) return CS_OK;
    // drain the device before the old buffers are copied and freed, or rows of an unfinished append would be lost
    if (h->n_rows) CS_TRY(drain_appends(h));
    uint64_t cap = h->capacity ? h->capacity * 2 : 1024;
    if (cap < need_rows) cap = need_rows;
    GrowBuffers nb;
    const size_t words = (size_t)((cap + 31) / 32);
    hipError_t e = hipMalloc(&nb.nc, (size_t)cap * h->dim * sizeof(float));
    if (e != hipSuccess) return fail(CS_ERR_OOM, "hipMalloc(corpus, %llu rows) failed: %s", (unsigned long long)cap, hipGetErrorString(e));
    e = hipMalloc(&nb.nd, words * sizeof(uint32_t));
    if (e != hipSuccess) return fail(CS_ERR_OOM, "hipMalloc(dead bitmap) failed: %s", hipGetErrorString(e));
    e = hipMalloc(&nb.nn, (size_t)cap * sizeof(float));
    if (e != hipSuccess) return fail(CS_ERR_OOM, "hipMalloc(row norms) failed: %s", hipGetErrorString(e));
    bool drop_split = false, drop_q8 = false;
    size_t cap256 = 0;
    if (h->d_split) {
        // the f16 copy exists (the int8 copy does not serve): it grows with the corpus.  Without room for it, it is
        // dropped — searches then build it again on demand or stay on the exact paths; the int8 copy below does not
        // depend on it (whole 128-row tiles, an even number of them: 256-row blocks)
        cap256 = ((size_t)cap + 255) / 256 * 256;
        if (hipMalloc(&nb.ns, cap256 * h->dim * sizeof(_Float16)) != hipSuccess) {
            (void)hipGetLastError();
            nb.ns = nullptr;
            drop_split = true;
        } else if (h->split_rows) {
            CS_GROW_COPY(1, hipMemcpy(nb.ns, h->d_split, ((size_t)h->split_rows + 127) / 128 * 128 * h->dim * sizeof(_Float16),
                                      hipMemcpyDeviceToDevice));
        }
    }
    if (h->use_q8) {
        const size_t tiles = ((size_t)cap + 255) / 256 * 2;  // an even number: the 256-row tile kernel reads whole pairs
        const bool fault = cs_lab_env("CS_FAULT_INT8_ALLOC") != nullptr && h->capacity != 0;  // tests: the failure path
        if (fault || hipMalloc(&nb.n8, tiles * 128 * h->dim) != hipSuccess || hipMalloc(&nb.nm, tiles * sizeof(float4)) != hipSuccess) {
            (void)hipGetLastError();  // no room: no int8 copy from here on (the f16 copy, or the exact paths, serve)
            if (nb.n8) (void)hipFree(nb.n8);
            nb.n8 = nullptr; nb.nm = nullptr;
            drop_q8 = true;
        } else if (h->q8_rows) {
            CS_GROW_COPY(2, hipMemcpy(nb.n8, h->d_q8, (size_t)h->q8_rows * h->dim, hipMemcpyDeviceToDevice));
            CS_GROW_COPY(2, hipMemcpy(nb.nm, h->d_tmeta, (size_t)(h->q8_rows / 128) * sizeof(float4), hipMemcpyDeviceToDevice));
        }
    }
    CS_GROW_COPY(3, hipMemset(nb.nd, 0, words * sizeof(uint32_t)));
    if (h->normed_rows)
        CS_GROW_COPY(4, hipMemcpy(nb.nn, h->d_norms, (size_t)h->normed_rows * sizeof(float), hipMemcpyDeviceToDevice));
    if (h->n_rows) {
        CS_GROW_COPY(5, hipMemcpy(nb.nc, h->d_corpus, (size_t)h->n_rows * h->dim * sizeof(float), hipMemcpyDeviceToDevice));
        CS_GROW_COPY(6, hipMemcpy(nb.nd, h->h_dead.data(), h->h_dead.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // ---- commit: nothing below can fail ----
    if (h->d_split) {
        (void)hipFree(h->d_split);
        h->d_split = nb.ns;
        nb.ns = nullptr;
        if (drop_split) { h->split_rows = 0; h->split_cap = 0; }
        else h->split_cap = cap256;
    }
    if (h->use_q8) {
        // the old, smaller buffers never survive a grow: a build that found them would convert tiles past their end
        if (h->d_q8) (void)hipFree(h->d_q8);
        if (h->d_tmeta) (void)hipFree(h->d_tmeta);
        h->d_q8 = nb.n8;
        h->d_tmeta = nb.nm;
        nb.n8 = nullptr; nb.nm = nullptr;
        if (drop_q8) { h->use_q8 = false; h->q8_rows = 0; }
    }
    if (h->d_norms) (void)hipFree(h->d_norms);
    h->d_norms = nb.nn;
    if (h->d_corpus) (void)hipFree(h->d_corpus);
    if (h->d_dead) (void)hipFree(h->d_dead);
    h->d_corpus = nb.nc;
    h->d_dead = nb.nd;
    nb.nn = nullptr; nb.nc = nullptr; nb.nd = nullptr;
    h->capacity = cap;
    return CS_OK;
}
#undef CS_GROW_COPY

// the int8 copy is the filter's operand: it exists, was not retired, and reaches past phase 0's rows
bool q8_serves(const cs_index* h) {
    return h->use_q8 && h->d_q8 && h->d_tmeta && h->q8_active.load() && h->q8_rows > kFilterPhase0;
}

// The f16 copy, complete for the rows of the last build — built now if it is not there (allocation + one conversion pass
// over the missing rows on the null stream, waited for: 7.68 GB and ~4.7 ms per 10M x 384, paid once, by the first
// search that needs it).  false: no room (remembered until clear()).
bool ensure_f16(cs_index* h) {
    if (!h->use_split) return false;
    std::lock_guard<std::mutex> lk(h->filter_mu);
    if (h->d_split && h->split_rows >= h->n_rows) return true;
    if (h->f16_failed || h->normed_rows < h->n_rows) return false;
    const size_t cap256 = ((size_t)h->capacity + 255) / 256 * 256;
    if (!h->d_split || h->split_cap < cap256) {
        if (h->d_split) (void)hipFree(h->d_split);
        h->d_split = nullptr; h->split_rows = 0; h->split_cap = 0;
        if (cs_lab_env("CS_FAULT_F16_ALLOC") != nullptr || hipMalloc(&h->d_split, cap256 * h->dim * sizeof(_Float16)) != hipSuccess) {
            (void)hipGetLastError();
            h->d_split = nullptr;
            h->f16_failed = true;
            return false;
        }
        h->split_cap = cap256;
    }
    if (launch_corpus_split(h->d_corpus, h->d_norms, h->d_split, h->split_rows, h->n_rows - h->split_rows, h->dim, nullptr) != CS_OK ||
        hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipGetLastError();
        h->f16_failed = true;
        return false;
    }
    h->split_rows = h->n_rows;
    return true;
}

// rows idx[0 .. n) of src (relative to it) -> dst rows 0 .. n, `dim` floats each
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n, uint32_t dim, float* __restrict__ dst) {
    const uint64_t total = n * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / dim;
        dst[i] = src[(size_t)idx[r] * dim + (i - r * dim)];
    }
}

// cs_index_build, when enough of the stored rows are tombstones: the live rows move to the front of the corpus in their order
// (chunk by chunk: a chunk's survivors are gathered straight into place when their destination lies wholly before the chunk,
// through a bounded staging buffer otherwise — never a second corpus), the row -> id table keeps their ids, and everything
// derived from the rows (norms, mean unit row, int8 / f16 filter copies) is rebuilt by the build that called this.  A search
// afterwards streams only live rows, and returns what it returned before: the same ids, the same cosines, bit for bit.
int32_t compact(cs_index* h) {
    const uint64_t live = h->n_rows - h->n_removed;
    constexpr uint64_t CH = 1u << 18;  // rows per chunk (staging: 403 MB at dim 384)
    const uint64_t chunk = std::min<uint64_t>(CH, h->n_rows);
    struct Tmp {
        float* rows = nullptr; uint32_t* idx = nullptr;
        ~Tmp() { if (rows) (void)hipFree(rows); if (idx) (void)hipFree(idx); }
    } t;
    CS_HIP(hipMalloc(&t.rows, (size_t)chunk * h->dim * sizeof(float)));
    CS_HIP(hipMalloc(&t.idx, (size_t)chunk * sizeof(uint32_t)));
    std::vector<uint32_t> nid;
    nid.reserve((size_t)live);
    std::vector<uint32_t> idx;
    idx.reserve((size_t)chunk);
    const bool ident = h->h_ids.empty();
    uint64_t dst = 0;
    for (uint64_t c0 = 0; c0 < h->n_rows; c0 += chunk) {
        const uint64_t c1 = std::min(h->n_rows, c0 + chunk);
        idx.clear();
        for (uint64_t r = c0; r < c1; ++r)
            if (!((h->h_dead[(size_t)(r >> 5)] >> (r & 31)) & 1u)) {
                idx.push_back((uint32_t)(r - c0));
                nid.push_back(ident ? h->id_base + (uint32_t)r : h->h_ids[(size_t)r]);
            }
        const uint64_t cnt = idx.size();
        if (cnt == 0) continue;
        if (dst == c0 && cnt == c1 - c0) { dst += cnt; continue; }  // nothing deleted up to here: the rows are in place
        CS_HIP(hipMemcpyAsync(t.idx, idx.data(), (size_t)cnt * sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((cnt * h->dim + 255) / 256, (uint64_t)h->num_cus * 16);
        const bool direct = dst + cnt <= c0;  // the destination does not reach into the chunk being read
        hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, nullptr, h->d_corpus + (size_t)c0 * h->dim, t.idx, cnt, h->dim,
                           direct ? h->d_corpus + (size_t)dst * h->dim : t.rows);
        CS_HIP(hipGetLastError());
        if (!direct)
            CS_HIP(hipMemcpyAsync(h->d_corpus + (size_t)dst * h->dim, t.rows, (size_t)cnt * h->dim * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
        CS_HIP(hipStreamSynchronize(nullptr));  // (idx is reused by the next chunk)
        dst += cnt;
    }
    if (dst != live) return fail(CS_ERR_HIP, "compaction moved %llu rows, expected %llu", (unsigned long long)dst, (unsigned long long)live);
    h->n_rows = live;
    h->n_removed = 0;
    h->h_dead.assign((size_t)((live + 31) / 32), 0u);
    if (h->d_dead && h->capacity) CS_HIP(hipMemset(h->d_dead, 0, (size_t)((h->capacity + 31) / 32) * sizeof(uint32_t)));
    h->h_ids.swap(nid);
    h->ids_uploaded = 0;
    // everything derived from the rows is rebuilt over the new storage order by the build that follows
    h->normed_rows = 0;
    h->q8_rows = 0;
    h->split_rows = 0;
    h->compactions += 1;
    return CS_OK;
}

int32_t check_append(cs_index* h, uint64_t n, uint32_t dim) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    if (dim != h->dim)  // store.rs:667-671
        return fail(CS_ERR_DIM_MISMATCH, "Embedding dimension mismatch: expected %u, got %u",
                    h->dim, dim);
    if ((uint64_t)h->id_base + h->n_ids + n > 0xffffffffull)
        return fail(CS_ERR_BAD_ARG, "id space exhausted: ids are u32 (store.rs:97)");
    return CS_OK;
}

void finish_append(cs_index* h, uint64_t n, uint32_t* out_ids) {
    const uint32_t start = h->id_base + (uint32_t)h->n_ids;   // ids are never reused (store.rs:101)
    if (out_ids)
        for (uint64_t i = 0; i < n; ++i) out_ids[i] = start + (uint32_t)i;  // store.rs:684
    if (!h->h_ids.empty())  // a compacted index: the new rows' ids join the row -> id table (uploaded by the next build)
        for (uint64_t i = 0; i < n; ++i) h->h_ids.push_back(start + (uint32_t)i);
    h->n_ids += n;
    h->n_rows += n;
    h->h_dead.resize((size_t)((h->n_rows + 31) / 32), 0u);
    if (n) h->built = false;  // store.rs:682
}

Workspace* acquire_pooled(cs_index* h) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->pool.empty()) {
        Workspace* w = h->pool.back();
        h->pool.pop_back();
        return w;
    }
    Workspace* w = new Workspace();
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) {
        delete w;
        return nullptr;
    }
    w->own_stream = true;
    return w;
}

void release_pooled(cs_index* h, Workspace* w) {
    std::lock_guard<std::mutex> lk(h->mu);
    h->pool.push_back(w);
}

Workspace* for_stream(cs_index* h, hipStream_t s) {
    std::lock_guard<std::mutex> lk(h->mu);
    const auto key = std::make_pair(s, std::this_thread::get_id());
    auto it = h->by_stream.find(key);
    if (it != h->by_stream.end()) return it->second;
    Workspace* w = new Workspace();
    w->stream = s;
    h->by_stream[key] = w;
    return w;
}

bool take_events(cs_index* h, Workspace* w, EventTriple* t) {
    if (!h->profile) return false;
    if (!w->free_events.empty()) {
        *t = w->free_events.back();
        w->free_events.pop_back();
        return true;
    }
    if (hipEventCreate(&t->e0) != hipSuccess) return false;
    if (hipEventCreate(&t->e1) != hipSuccess) return false;
    if (hipEventCreate(&t->e2) != hipSuccess) return false;
    return true;
}

// scan + merge on `stream`; outputs are device pointers (any may be null).
// h_queries_pinned != null: the queries are still in that pinned host buffer and d_queries is empty;
// the filter path lets its prep kernel bring them over, every other path copies them first.
int32_t run_search(cs_index* h, Workspace* w, const ScanPlan& plan, const float* d_queries,
                   uint32_t nq, uint32_t k, uint64_t* d_keys, float* d_cos, uint32_t* d_ids,
                   uint32_t* d_counts, hipStream_t stream, const float* h_queries_pinned = nullptr,
                   bool may_sync = true) {
    EventTriple ev{};
    const bool timed = take_events(h, w, &ev);
    if (timed) CS_HIP(hipEventRecord(ev.e0, stream));
    // >= 5 queries: MFMA scoring + phased candidate selection (scan_mfma.hip)
    // One query normally stays on the exact f32 streaming scan (the north-star kernel).  With a long list over a
    // multi-million-row index — the reference's own retrieval_limit (100 or 200) when a search has no query variants —
    // the filter + refine path is taken instead: same bits, 1.40 vs 2.36 ms at k = 200 over 10M x 384, because
    // the scan's list inserts need a second block per CU there and the filter reads half the bytes.
    // ... and over a corpus of the reference's own size (hundreds of chunks) the batched path — prep, direct scoring of
    // every row, select: three small launches, no filter involved below a candidate buffer's worth of rows — answers one
    // query faster than the streaming scan's per-wave lists do (592 rows: 30 vs 41 us; 1,000: 33 vs 46; from 2,000
    // rows on the scan is ahead: 43 vs 48 us).
    const bool single_filter =
        nq == 1 && h->single_route != CS_ROUTE_STREAM &&
        (h->single_route == CS_ROUTE_FILTER ||
         (q8_serves(h) && h->n_rows >= (k < 48 ? h->single_int8_min_rows : h->single_int8_min_rows_long)) ||
         (h->n_rows >= h->single_filter_min_rows && h->single_filter_min_k && k >= h->single_filter_min_k));
    // (First measurement, round 4:) two to four queries over a corpus between one phase 0 and ~50,000 rows: the streaming scan (one pass per query
    // tile) is ahead of the filter's fixed rounds (profiles/r04_batched_route_by_size.log, us per search at nq = 2, k = 25,
    // filter / stream: 2,000 rows 37 / 45; 5,000 63 / 45; 20,000 71 / 60; 100,000 97 / 117); from five queries on the filter
    // wins at every size (9 x 200: 41 ... 277 us against 81 ... 600 on the exact-f32 MFMA path).
    // Re-measured behind the one-round phase 0 and the one-round plan of small corpora (profiles/r04_few_queries_crossover.log,
    // us per search, stream / filter): FOUR queries are ahead on the filter from 5,000 rows on (k = 10: 5k 63 / 54, 20k 78 / 61,
    // 50k 119 / 67; k = 25: 5k 64 / 58, 50k 95 / 76); TWO stream up to ~16,000 rows with a short list (k = 10: 10k 47 / 56,
    // 20k 66 / 61) and up to ~40,000 rows above (k = 25: 20k 60 / 65, 35k 66 / 69, 50k 82 / 73), and THREE cost the streaming
    // scan what two do (one pass: 5k rows 44 / 55, 10k 48 / 57, 20k at k = 25 60 / 66; four take a second pass: 63), so they
    // follow two.
    const uint64_t few_min = k <= 16 ? h->few_queries_min_rows_short : h->few_queries_min_rows;
    const bool few_small = nq >= 2 && nq <= 3 && h->filter_min_q == 2 && h->n_rows > kFilterPhase0 && h->n_rows < few_min;
    const bool wants_filter = ((int)nq >= h->filter_min_q && !few_small) || single_filter ||
                              (nq == 1 && h->n_rows <= h->single_batched_max_rows);
    const bool normed = h->n_rows > 0 && h->normed_rows >= h->n_rows;
    // Which copy filters: the int8 one when it serves; else the f16 one, built here, once, if it is not there yet; with
    // neither (no room) the search takes the exact paths below.
    bool via_q8 = false, use_filter = false;
    if (h->use_split && wants_filter && normed) {
        CS_TRY(w->reserve_batched(nq, k));
        // overflowed searches the device has reported since this workspace last looked: a strike against the int8 copy
        // only when the search that overflowed read it (the exact-f32 batched path and the f16 filter share the word).
        // A strike can retire the copy: the choice is made after it.
        if (w->bs.h_mirror) {
            const uint32_t seen = *reinterpret_cast<volatile uint32_t*>(w->bs.h_mirror);
            if (seen != w->mirror_seen) {
                w->mirror_seen = seen;
                if (w->last_via_q8 && h->q8_active.load()) h->q8_strike();
            }
        }
        via_q8 = q8_serves(h);
        bool planes = false;
        CS_TRY(w->reserve_split_queries(nq, h->dim, via_q8, &planes));
        if (via_q8 && !planes) via_q8 = false;  // no room for the int8 query planes
        use_filter = via_q8 || ensure_f16(h);
    }
    const bool filter_path = use_filter;
    w->qw.q_pinned = filter_path ? h_queries_pinned : nullptr;
    // A streaming scan of a few blocks (a corpus of the reference's own size: hundreds to thousands of chunks) reads
    // the queries straight from the pinned buffer too: a copy launch costs more than <= 64 blocks' reads over the link.
    const bool streaming = !filter_path && !(normed && nq >= 5 && batched_supported(h->dim));
    if (h_queries_pinned && streaming && scan_prime_supported(h->dim) /* queries go to registers once */ &&
        (uint64_t)plan.blocks * plan.passes <= 64)
        d_queries = h_queries_pinned;
    else if (h_queries_pinned && !filter_path)
        CS_HIP(hipMemcpyAsync(const_cast<float*>(d_queries), h_queries_pinned, (size_t)nq * h->dim * sizeof(float),
                              hipMemcpyHostToDevice, stream));
    // Two or more queries: the filter reads a quarter (int8) or half (f16) of the bytes of the f32 scan once for up to
    // 128 queries and the refine step keeps the result bit-identical.  One query stays on the streaming f32 scan
    // (the north-star kernel).  Without a filter copy, >= 5 queries use the exact-f32 MFMA path.
    if (normed && (use_filter || (nq >= 5 && batched_supported(h->dim)))) {
        CS_TRY(w->reserve_batched(nq, k));
        if (use_filter) {
            Q8View q8;
            w->last_via_q8 = via_q8;
            if (via_q8) {
                q8.d_q8 = h->d_q8; q8.d_tmeta = h->d_tmeta; q8.d_mu = h->d_mu; q8.rows = h->q8_rows;
                h->q8_searches.fetch_add(1);
            }
            CS_TRY(launch_scan_split(w->bs, w->qw, h->d_corpus, h->d_split, h->n_rows, h->dim,
                                     d_queries, nq, k, h->n_removed ? h->d_dead : nullptr, h->row_ids(), d_keys,
                                     d_cos, d_ids, d_counts, stream, h->filter_margin, &q8));
        } else {
            w->last_via_q8 = false;
            CS_TRY(launch_scan_batched(w->bs, h->d_corpus, h->d_norms, h->n_rows, h->dim, d_queries, nq, k,
                                       h->n_removed ? h->d_dead : nullptr, h->row_ids(), h->num_cus, d_keys, d_cos,
                                       d_ids, d_counts, stream));
        }
        if (timed) CS_HIP(hipEventRecord(ev.e1, stream));
        // A candidate buffer holds batched_cap(k) entries and a phase appends at most one per row: it can
        // only overflow over more rows than that (adversarial row order; thousands of near-duplicate rows).
        const bool can_overflow = h->n_rows > batched_cap(k);
        if (!may_sync) {
            // Device API: never wait for the device.  Up to kGatedMaxQ queries, the exact list-based scan and
            // its merge are enqueued right behind the search with the overflow word as their gate: every
            // block exits at once unless the search overflowed (then they overwrite its outputs), so what
            // the caller's stream delivers is exact either way; cost when not taken ~3 near-empty launches.
            // Above that (hundreds of query passes would be enqueued) the sticky word is left for
            // cs_index_search_status().
            if (can_overflow && nq <= kGatedMaxQ) {
                const uint32_t* gate = w->bs.d_overflow;
                CS_TRY(launch_scan(plan, h->d_corpus, h->n_rows, h->dim, d_queries, nq, k,
                                   h->n_removed ? h->d_dead : nullptr, h->row_ids(), w->d_partial, stream, nullptr, false,
                                   gate));
                CS_TRY(launch_merge(w->d_partial, plan.blocks, nq, k, false, w->d_tmp_a, w->d_tmp_b, d_keys, d_cos, d_ids,
                                    d_counts, stream, gate));
            }
            std::lock_guard<std::mutex> lk(h->mu);
            h->batched_searches++;
            if (timed) {
                CS_HIP(hipEventRecord(ev.e2, stream));
                h->pending.push_back(ev);
            }
            return CS_OK;
        }
        bool overflow = false;
        if (can_overflow) {  // host-buffer API: it synchronises for its results anyway
            CS_HIP(hipMemcpyAsync(w->h_overflow, w->bs.d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            CS_HIP(hipStreamSynchronize(stream));
            overflow = *w->h_overflow != 0;
            if (overflow && via_q8) h->q8_strike();  // the int8 copy's band let too many rows through
            if (overflow && via_q8 && ensure_f16(h)) {
                // ... and the f16 copy (band 0.001; built now if this is the first time it is needed) answers this search
                // before the exact list-based scan is asked to
                w->last_via_q8 = false;
                CS_TRY(launch_scan_split(w->bs, w->qw, h->d_corpus, h->d_split, h->n_rows, h->dim, d_queries, nq, k,
                                         h->n_removed ? h->d_dead : nullptr, h->row_ids(), d_keys, d_cos, d_ids, d_counts,
                                         stream, h->filter_margin, nullptr));
                CS_HIP(hipMemcpyAsync(w->h_overflow, w->bs.d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                CS_HIP(hipStreamSynchronize(stream));
                overflow = *w->h_overflow != 0;
                // the rerun's first kernel has mirrored the count of overflowed searches, this one included: seen
                w->mirror_seen = *reinterpret_cast<volatile uint32_t*>(w->bs.h_mirror);
                std::lock_guard<std::mutex> lk(h->mu);
                h->q8_reruns++;
            }
        }
        {
            std::lock_guard<std::mutex> lk(h->mu);
            h->batched_searches++;
            if (overflow) h->batched_fallbacks++;
            if (timed) {
                CS_HIP(hipEventRecord(ev.e2, stream));
                h->pending.push_back(ev);
            }
        }
        if (!overflow) return CS_OK;
        // candidate buffer overflowed: exact list-based rerun below
        EventTriple none{};
        ev = none;
        return [&]() -> int32_t {
            CS_TRY(launch_scan(plan, h->d_corpus, h->n_rows, h->dim, d_queries, nq, k,
                               h->n_removed ? h->d_dead : nullptr, h->row_ids(), w->d_partial, stream));
            return launch_merge(w->d_partial, plan.blocks, nq, k, false, w->d_tmp_a, w->d_tmp_b, d_keys, d_cos,
                                d_ids, d_counts, stream);
        }();
    }
    const uint32_t* d_dead = h->n_removed ? h->d_dead : nullptr;
    const ScanPrime* prime = nullptr;
    uint64_t prime_rows = h->prime_rows;
    if (!prime_rows) {
        prime_rows = (h->n_rows / 256) & ~(uint64_t)63;
        // short lists need fewer wave maxima for a useful bound: 8,192 rows (512 waves of 16) up to k = 16 — over 10M
        // rows the pass costs 18 instead of 26 us and the scan the same (k = 10: 2,122 -> 2,114 us; k = 64 and 99 lose
        // 5 and 17 us with the smaller sample and keep 16,384)
        const uint64_t prime_cap = k <= 16 ? 8192 : 16384;
        prime_rows = prime_rows < 4096 ? 4096 : (prime_rows > prime_cap ? prime_cap : prime_rows);
        const uint64_t big = prime_sample_rows(prime_rows, k, h->num_cus);  // k > 256: more waves, 32 rows each
        if (h->n_rows >= 4 * big) prime_rows = big;
    }
    const uint64_t prime_min_rows = h->prime_min_rows ? h->prime_min_rows : (k >= 48 ? 100000 : 500000);
    if (h->prime_min_k && k >= h->prime_min_k && h->n_rows >= prime_min_rows &&
        h->n_rows >= 4 * prime_rows && scan_prime_supported(h->dim)) {
        const ScanPlan pp = plan_prime(prime_rows, h->dim, nq, k, h->num_cus);
        CS_TRY(w->reserve_prime(pp, nq));
        CS_TRY(launch_scan(pp, h->d_corpus, prime_rows, h->dim, d_queries, nq, k, d_dead, h->row_ids(),
                           nullptr, stream, &w->prime, true));
        prime = &w->prime;
    }
    CS_TRY(launch_scan(plan, h->d_corpus, h->n_rows, h->dim, d_queries, nq, k, d_dead, h->row_ids(),
                       w->d_partial, stream, prime));
    if (timed) CS_HIP(hipEventRecord(ev.e1, stream));
    CS_TRY(launch_merge(w->d_partial, plan.blocks, nq, k, false, w->d_tmp_a, w->d_tmp_b, d_keys, d_cos,
                        d_ids, d_counts, stream));
    if (timed) {
        CS_HIP(hipEventRecord(ev.e2, stream));
        std::lock_guard<std::mutex> lk(h->mu);
        h->pending.push_back(ev);
    }
    return CS_OK;
}

int32_t check_search(const cs_index* h, uint32_t nq, uint32_t dim, uint32_t k) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    if (dim != h->dim)  // store.rs:432-438
        return fail(CS_ERR_DIM_MISMATCH,
                    "Query embedding dimension mismatch: expected %u, got %u", h->dim, dim);
    if (!h->built)  // store.rs:440-444
        return fail(CS_ERR_NOT_BUILT,
                    "Index not built. Call build_index() after inserting chunks.");
    if (nq == 0 || nq > CS_MAX_QUERIES)
        return fail(CS_ERR_BAD_ARG, "nq must be in 1..%u, got %u", CS_MAX_QUERIES, nq);
    if (k == 0 || k > CS_MAX_K)
        return fail(CS_ERR_BAD_ARG, "k must be in 1..%u, got %u", CS_MAX_K, k);
    return CS_OK;
}

}  // namespace

extern "C" {

const char* cs_last_error(void) { return last_error_ref().c_str(); }
uint32_t cs_abi_version(void) { return 6; }  // 5: cs_bert_config gained arch, rotary_base; 6: rotary_base_local, local_window, global_every

int32_t cs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t cs_index_create(uint32_t dim, uint64_t capacity_rows, int32_t device, uint32_t id_base,
                        cs_index** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (dim == 0 || dim > 8192) return fail(CS_ERR_BAD_ARG, "dim must be in 1..8192, got %u", dim);
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(CS_ERR_HIP, "HIP device %d not available (%d visible); there is no CPU fallback",
                    device, ndev);
    DeviceGuard g(device);
    hipDeviceProp_t prop;
    CS_HIP(hipGetDeviceProperties(&prop, device));
    cs_index* h = new cs_index();
    h->device = device;
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->dim = dim;
    h->id_base = id_base;
    {
        const char* env = std::getenv("CS_INDEX_SPLIT");  // "0": keep the batched path on the exact-f32 MFMA
        h->use_split = split_scan_supported(dim) && !(env && env[0] == '0');
        const char* e8 = std::getenv("CS_FILTER_INT8");  // "0": filter on the f16 copy only
        h->use_q8 = h->use_split && !(e8 && e8[0] == '0');
        const char* ee = std::getenv("CS_FILTER_F16_EAGER");  // "1": keep the f16 copy beside a serving int8 copy
        h->f16_eager = (ee && ee[0] == '1') || cs_lab_env("CS_FILTER_INT8_MAX_Q") != nullptr;
        if (const char* e = std::getenv("CS_FILTER_INT8_MAX_SPREAD")) h->q8_max_spread = (float)std::atof(e);
        if (const char* e = std::getenv("CS_FILTER_MIN_Q")) {
            h->filter_min_q = std::atoi(e);
            if (h->filter_min_q < 1) h->filter_min_q = 1;
        }
        if (const char* e = std::getenv("CS_FILTER_SINGLE_MIN_K")) {  // "0": one query never takes the filter (= CS_ROUTE_STREAM)
            h->single_filter_min_k = (uint32_t)std::atol(e);
            if (h->single_filter_min_k == 0) h->single_route = CS_ROUTE_STREAM;
        }
        if (const char* e = std::getenv("CS_INDEX_COMPACT_DEAD_PCT")) h->compact_dead_pct = (uint32_t)std::max(0, std::min(100, std::atoi(e)));
        if (const char* e = std::getenv("CS_FILTER_SINGLE_MIN_ROWS")) h->single_int8_min_rows = (uint64_t)std::atoll(e);
        if (const char* e = std::getenv("CS_FILTER_SINGLE_MIN_ROWS_LONG")) h->single_int8_min_rows_long = (uint64_t)std::atoll(e);
        if (const char* e = cs_lab_env("CS_FILTER_FEW_MIN_ROWS")) h->few_queries_min_rows = h->few_queries_min_rows_short = (uint64_t)std::atoll(e);
        if (const char* e = std::getenv("CS_SINGLE_BATCHED_MAX_ROWS")) h->single_batched_max_rows = (uint64_t)std::atoll(e);
        if (const char* e = std::getenv("CS_SCAN_PRIME_MIN_K")) h->prime_min_k = (uint32_t)std::atol(e);  // 0 = off
        if (const char* e = std::getenv("CS_SCAN_PRIME_MIN_ROWS")) h->prime_min_rows = (uint64_t)std::atoll(e);
        if (const char* e = std::getenv("CS_SCAN_PRIME_ROWS")) h->prime_rows = (uint64_t)std::atoll(e);
    }
    if (h->use_split) {
        // does the f16 MFMA take subnormal inputs exactly?  One one-wave launch per device per process.
        static std::mutex mu;
        static std::map<int, bool> known;
        std::lock_guard<std::mutex> lk(mu);
        auto it = known.find(device);
        if (it == known.end()) {
            bool ok = false;
            int32_t s = sh_denorm_selftest(&ok, nullptr);
            if (s != CS_OK) { delete h; return s; }
            it = known.emplace(device, ok).first;
        }
        h->filter_margin = filter_margin(dim, it->second);
    }
    if (capacity_rows) {
        int32_t s = grow(h, capacity_rows);
        if (s != CS_OK) { delete h; return s; }
    }
    *out = h;
    return CS_OK;
}

void cs_index_destroy(cs_index* h) {
    if (!h) return;
    DeviceGuard g(h->device);
    (void)drain_appends(h);
    for (auto* w : h->pool) { w->release_all(); delete w; }
    for (auto& kv : h->by_stream) { kv.second->release_all(); delete kv.second; }
    for (auto& t : h->pending) {
        (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); (void)hipEventDestroy(t.e2);
    }
    if (h->d_corpus) (void)hipFree(h->d_corpus);
    if (h->d_dead) (void)hipFree(h->d_dead);
    if (h->d_norms) (void)hipFree(h->d_norms);
    if (h->d_split) (void)hipFree(h->d_split);
    if (h->d_q8) (void)hipFree(h->d_q8);
    if (h->d_tmeta) (void)hipFree(h->d_tmeta);
    if (h->d_mu) (void)hipFree(h->d_mu);
    if (h->d_ids) (void)hipFree(h->d_ids);
    delete h;
}

int32_t cs_index_add(cs_index* h, const float* rows, uint64_t n, uint32_t dim, uint32_t* out_ids) {
    CS_TRY(check_append(h, n, dim));
    if (n == 0) return CS_OK;  // store.rs:655-657
    if (!rows) return fail(CS_ERR_BAD_ARG, "rows is null");
    DeviceGuard g(h->device);
    CS_TRY(grow(h, h->n_rows + n));
    CS_HIP(hipMemcpy(h->d_corpus + (size_t)h->n_rows * h->dim, rows,
                     (size_t)n * h->dim * sizeof(float), hipMemcpyHostToDevice));
    finish_append(h, n, out_ids);
    return CS_OK;
}

int32_t cs_index_add_device(cs_index* h, const float* d_rows, uint64_t n, uint32_t dim,
                            uint32_t* out_ids, void* stream) {
    CS_TRY(check_append(h, n, dim));
    if (n == 0) return CS_OK;
    if (!d_rows) return fail(CS_ERR_BAD_ARG, "d_rows is null");
    DeviceGuard g(h->device);
    CS_TRY(grow(h, h->n_rows + n));
    CS_HIP(hipMemcpyAsync(h->d_corpus + (size_t)h->n_rows * h->dim, d_rows,
                          (size_t)n * h->dim * sizeof(float), hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
    finish_append(h, n, out_ids);
    return CS_OK;
}

// E8 in place ("optionally write straight into corpus matrix row"): room for n more rows is made, *d_rows is where
// they go — the caller has them WRITTEN there (cs_embedder_embed_*_device with this address as its output: the pooling kernel's
// own stores land in the corpus) — and cs_index_commit_rows makes them rows of the index with the next n ids.  No staging buffer,
// no device-to-device copy.  Between the two calls the handle must see no other mutating call (&mut self, as insert_chunks).
int32_t cs_index_reserve_rows(cs_index* h, uint64_t n, uint32_t dim, float** d_rows) {
    CS_TRY(check_append(h, n, dim));
    if (!d_rows) return fail(CS_ERR_BAD_ARG, "d_rows is null");
    DeviceGuard g(h->device);
    CS_TRY(grow(h, h->n_rows + n));
    *d_rows = h->d_corpus + (size_t)h->n_rows * h->dim;
    return CS_OK;
}

int32_t cs_index_commit_rows(cs_index* h, uint64_t n, uint32_t* out_ids) {
    CS_TRY(check_append(h, n, h ? h->dim : 0));
    if (h->n_rows + n > h->capacity)
        return fail(CS_ERR_BAD_ARG, "cs_index_commit_rows: %llu rows were not reserved (capacity %llu, stored %llu)", (unsigned long long)n,
                    (unsigned long long)h->capacity, (unsigned long long)h->n_rows);
    finish_append(h, n, out_ids);
    return CS_OK;
}

int32_t cs_index_add_synthetic(cs_index* h, uint64_t n, uint64_t seed, uint64_t first_row,
                               uint32_t* out_first_id) {
    CS_TRY(check_append(h, n, h ? h->dim : 0));
    DeviceGuard g(h->device);
    CS_TRY(grow(h, h->n_rows + n));
    CS_TRY(launch_synth_fill(h->d_corpus + (size_t)h->n_rows * h->dim, n, h->dim, seed, first_row,
                             nullptr));
    CS_HIP(hipStreamSynchronize(nullptr));
    if (out_first_id) *out_first_id = h->id_base + (uint32_t)h->n_ids;
    finish_append(h, n, nullptr);
    return CS_OK;
}

int32_t cs_index_remove(cs_index* h, const uint32_t* ids, uint64_t n, uint64_t* removed) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    if (removed) *removed = 0;
    if (n == 0) return CS_OK;  // store.rs:585-587
    if (!ids) return fail(CS_ERR_BAD_ARG, "ids is null");
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (ids[i] < h->id_base) continue;
        uint64_t row = (uint64_t)ids[i] - h->id_base;
        if (!h->h_ids.empty()) {  // compacted: the id's row by bisection of the ascending row -> id table
            const auto it = std::lower_bound(h->h_ids.begin(), h->h_ids.end(), ids[i]);
            if (it == h->h_ids.end() || *it != ids[i]) continue;  // never issued, or deleted and reclaimed: not counted
            row = (uint64_t)(it - h->h_ids.begin());
        }
        if (row >= h->n_rows) continue;  // del_item fails -> not counted (store.rs:594)
        uint32_t& w = h->h_dead[(size_t)(row >> 5)];
        const uint32_t bit = 1u << (row & 31);
        if (w & bit) continue;
        w |= bit;
        ++cnt;
    }
    if (cnt) {
        DeviceGuard g(h->device);
        CS_HIP(hipMemcpy(h->d_dead, h->h_dead.data(), h->h_dead.size() * sizeof(uint32_t),
                         hipMemcpyHostToDevice));
        h->n_removed += cnt;
        h->built = false;  // store.rs:604-606
    }
    if (removed) *removed = cnt;
    return CS_OK;
}

int32_t cs_index_build(cs_index* h) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    DeviceGuard g(h->device);
    CS_TRY(drain_appends(h));  // appended rows (incl. async device appends) are now visible
    if (h->compact_dead_pct && h->n_removed && h->n_removed * 100 >= (uint64_t)h->compact_dead_pct * h->n_rows) CS_TRY(compact(h));
    if (!h->h_ids.empty() && h->ids_uploaded < h->n_rows) {  // the row -> id table of a compacted index, for the rows that are new
        if (h->ids_cap < h->n_rows) {
            uint32_t* nt = nullptr;
            const uint64_t cap = std::max<uint64_t>(h->capacity, h->n_rows);
            CS_HIP(hipMalloc(&nt, (size_t)cap * sizeof(uint32_t)));
            if (h->d_ids) (void)hipFree(h->d_ids);
            h->d_ids = nt;
            h->ids_cap = cap;
            h->ids_uploaded = 0;
        }
        CS_HIP(hipMemcpy(h->d_ids + h->ids_uploaded, h->h_ids.data() + h->ids_uploaded,
                         (size_t)(h->n_rows - h->ids_uploaded) * sizeof(uint32_t), hipMemcpyHostToDevice));
        h->ids_uploaded = h->n_rows;
    }
    if ((batched_supported(h->dim) || h->use_split) && h->normed_rows < h->n_rows) {
        CS_TRY(launch_row_norms(h->d_corpus, h->normed_rows, h->n_rows - h->normed_rows, h->dim,
                                h->d_norms, nullptr));
        CS_HIP(hipDeviceSynchronize());
        h->normed_rows = h->n_rows;
    }
    if (h->use_q8 && h->d_q8 && h->q8_rows / 128 < h->n_rows / 128) {  // tiles that became complete
        if (!h->d_mu) CS_HIP(hipMalloc(&h->d_mu, h->dim * sizeof(float)));
        if (h->q8_rows == 0)  // first tiles of this copy: centre it on the mean unit row of what is there now
            CS_TRY(launch_unit_mean(h->d_corpus, h->d_norms, std::min<uint64_t>(h->n_rows, 1u << 20), h->dim, h->d_mu, nullptr));
        CS_TRY(launch_corpus_q8(h->d_corpus, h->d_norms, h->d_q8, h->d_tmeta, h->q8_rows / 128,
                                h->n_rows / 128 - h->q8_rows / 128, h->dim, h->d_mu, nullptr));
        CS_HIP(hipDeviceSynchronize());
        h->q8_rows = h->n_rows / 128 * 128;
        // outlier coordinates: the tile scale is the tile's largest |u - mu|; where the typical tile's is far above what
        // evenly spread coordinates give (4.3 / sqrt(dim) for Gaussian rows), the band (it grows with the square) lets
        // through more rows than the candidate buffers hold — the f16 copy serves the filter then
        {
            const uint64_t nt = std::min<uint64_t>(h->q8_rows / 128, 4096);
            std::vector<float4> tm((size_t)nt);
            CS_HIP(hipMemcpy(tm.data(), h->d_tmeta, (size_t)nt * sizeof(float4), hipMemcpyDeviceToHost));
            std::vector<float> sp;
            sp.reserve((size_t)nt);
            for (const float4& t : tm)
                if (t.x == t.x && t.x > 0.0f) sp.push_back(127.0f / t.x * std::sqrt((float)h->dim));
            if (!sp.empty()) {
                std::nth_element(sp.begin(), sp.begin() + sp.size() / 2, sp.end());
                h->q8_spread = sp[sp.size() / 2];
                if (h->q8_spread > h->q8_max_spread) h->q8_active.store(false);
            }
        }
    }
    if (h->use_split) {
        // The f16 copy is kept only where the int8 copy does not serve (none, retired, <= 1024 rows) or on request: an
        // index the int8 copy serves holds f32 + int8 = 5 bytes per element and its builds skip the conversion pass.
        // Should the int8 copy be retired later (two overflowed searches), the first search after that builds it.
        if (h->f16_eager || !q8_serves(h)) {
            (void)ensure_f16(h);  // no room: not an error — searches take the exact paths
        } else if (h->d_split) {
            std::lock_guard<std::mutex> lk(h->filter_mu);
            (void)hipFree(h->d_split);
            h->d_split = nullptr;
            h->split_rows = 0;
            h->split_cap = 0;
        }
    }
    h->built = true;                 // store.rs:428
    return CS_OK;
}

int32_t cs_index_clear(cs_index* h) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    DeviceGuard g(h->device);
    CS_TRY(drain_appends(h));
    if (h->d_dead && h->capacity)
        CS_HIP(hipMemset(h->d_dead, 0, (size_t)((h->capacity + 31) / 32) * sizeof(uint32_t)));
    h->n_rows = 0;  // store.rs:701 next_id = 0
    h->n_ids = 0;
    h->h_ids.clear();
    h->ids_uploaded = 0;
    h->normed_rows = 0;
    h->split_rows = 0;
    h->f16_failed = false;
    h->q8_rows = 0;
    h->q8_active.store(true);
    h->q8_strikes.store(0);
    h->q8_searches.store(0);
    h->n_removed = 0;
    h->h_dead.clear();
    h->built = false;  // store.rs:702
    return CS_OK;
}

int32_t cs_index_is_built(const cs_index* h) { return h && h->built ? 1 : 0; }
uint64_t cs_index_len(const cs_index* h) { return h ? h->n_rows - h->n_removed : 0; }
uint64_t cs_index_stored_rows(const cs_index* h) { return h ? h->n_rows : 0; }
uint32_t cs_index_next_id(const cs_index* h) { return h ? h->id_base + (uint32_t)h->n_ids : 0; }
uint32_t cs_index_dim(const cs_index* h) { return h ? h->dim : 0; }
int32_t cs_index_device(const cs_index* h) { return h ? h->device : -1; }

int32_t cs_index_search(cs_index* h, const float* queries, uint32_t nq, uint32_t dim, uint32_t k,
                        float* out_cos, uint32_t* out_ids, uint32_t* out_counts) {
    CS_TRY(check_search(h, nq, dim, k));
    if (!queries || !out_cos || !out_ids || !out_counts)
        return fail(CS_ERR_BAD_ARG, "null buffer");
    DeviceGuard g(h->device);
    const ScanPlan plan = plan_scan(h->n_rows, h->dim, nq, k, h->num_cus);
    Workspace* w = acquire_pooled(h);
    if (!w) return fail(CS_ERR_HIP, "could not create a HIP stream");
    int32_t s = w->reserve(plan, nq, h->dim, k, true);
    if (s == CS_OK) {
        s = [&]() -> int32_t {
            memcpy(w->h_queries, queries, (size_t)nq * h->dim * sizeof(float));
            // the last kernel of the search writes the packed keys straight into the pinned host buffer
            // (device-addressable, coherent): no D2H copy call, one stream sync
            CS_TRY(run_search(h, w, plan, w->d_queries, nq, k, w->h_keys, nullptr, nullptr, nullptr, w->stream,
                              w->h_queries));
            CS_HIP(hipStreamSynchronize(w->stream));
            for (uint32_t q = 0; q < nq; ++q) {  // keys are best-first, 0 = empty slot
                uint32_t c = 0;
                for (uint32_t j = 0; j < k; ++j) {
                    const uint64_t key = w->h_keys[(size_t)q * k + j];
                    if (key) ++c;
                    out_cos[(size_t)q * k + j] = key ? key_cos(key) : 0.0f;
                    out_ids[(size_t)q * k + j] = key ? key_id(key) : 0xFFFFFFFFu;
                }
                out_counts[q] = c;
            }
            return CS_OK;
        }();
    }
    release_pooled(h, w);
    return s;
}

int32_t cs_index_search_device(cs_index* h, const float* d_queries, uint32_t nq, uint32_t dim,
                               uint32_t k, uint64_t* d_out_keys, float* d_out_cos,
                               uint32_t* d_out_ids, uint32_t* d_out_counts, void* stream) {
    CS_TRY(check_search(h, nq, dim, k));
    if (!d_queries) return fail(CS_ERR_BAD_ARG, "d_queries is null");
    DeviceGuard g(h->device);
    const ScanPlan plan = plan_scan(h->n_rows, h->dim, nq, k, h->num_cus);
    Workspace* w = for_stream(h, (hipStream_t)stream);
    CS_TRY(w->reserve(plan, nq, h->dim, k, false));
    return run_search(h, w, plan, d_queries, nq, k, d_out_keys, d_out_cos, d_out_ids, d_out_counts,
                      (hipStream_t)stream, nullptr, /*may_sync=*/false);
}

int32_t cs_index_search_variants(cs_index* h, const float* queries, uint32_t nq, uint32_t dim, uint32_t k,
                                 float* out_cos, uint32_t* out_ids, uint32_t* out_count,
                                 int32_t* out_high_confidence) {
    CS_TRY(check_search(h, nq, dim, k));
    if (nq > CS_MAX_VARIANTS)
        return fail(CS_ERR_BAD_ARG, "at most %u query variants per call, got %u", CS_MAX_VARIANTS, nq);
    if (!queries || !out_cos || !out_ids || !out_count) return fail(CS_ERR_BAD_ARG, "null buffer");
    DeviceGuard g(h->device);
    const ScanPlan plan = plan_scan(h->n_rows, h->dim, nq, k, h->num_cus);
    Workspace* w = acquire_pooled(h);
    if (!w) return fail(CS_ERR_HIP, "could not create a HIP stream");
    int32_t s = w->reserve(plan, nq, h->dim, k, true);
    if (s == CS_OK) {
        s = [&]() -> int32_t {
            if (!w->h_variant_meta) CS_HIP(hipHostMalloc(&w->h_variant_meta, 2 * sizeof(uint32_t)));
            memcpy(w->h_queries, queries, (size_t)nq * h->dim * sizeof(float));
            // per-variant lists stay in HBM (exact without a host round trip: <= 16 queries carry the gated rerun),
            // the merge kernel writes the <= k survivors and the two scalars straight into pinned host memory
            CS_TRY(run_search(h, w, plan, w->d_queries, nq, k, w->d_keys, nullptr, nullptr, nullptr, w->stream,
                              w->h_queries, /*may_sync=*/false));
            CS_TRY(launch_merge_variants(w->d_keys, nq, k, k, w->h_keys, nullptr, nullptr, w->h_variant_meta,
                                         w->h_variant_meta + 1, w->stream));
            CS_HIP(hipStreamSynchronize(w->stream));
            for (uint32_t j = 0; j < k; ++j) {
                const uint64_t key = w->h_keys[j];
                out_cos[j] = key ? key_cos(key) : 0.0f;
                out_ids[j] = key ? key_id(key) : 0xFFFFFFFFu;
            }
            *out_count = w->h_variant_meta[0];
            if (out_high_confidence) *out_high_confidence = (int32_t)w->h_variant_meta[1];
            return CS_OK;
        }();
    }
    release_pooled(h, w);
    return s;
}

int32_t cs_merge_variants_device(int32_t device, const uint64_t* d_keys, uint32_t nv, uint32_t k, uint32_t limit,
                                 uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_count,
                                 uint32_t* d_out_high_confidence, void* stream) {
    if (!d_keys || nv == 0 || nv > CS_MAX_VARIANTS || k == 0 || k > CS_MAX_K || limit == 0 || limit > CS_MAX_K)
        return fail(CS_ERR_BAD_ARG, "bad variant-merge arguments");
    DeviceGuard g(device);
    return launch_merge_variants(d_keys, nv, k, limit, d_out_keys, d_out_cos, d_out_ids, d_out_count,
                                 d_out_high_confidence, (hipStream_t)stream);
}

int32_t cs_index_search_status(cs_index* h, void* stream, uint32_t* overflowed) {
    if (!h || !overflowed) return fail(CS_ERR_BAD_ARG, "null argument");
    *overflowed = 0;
    DeviceGuard g(h->device);
    Workspace* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        auto it = h->by_stream.find(std::make_pair((hipStream_t)stream, std::this_thread::get_id()));
        if (it != h->by_stream.end()) w = it->second;
    }
    if (!w || !w->bs.d_overflow) return CS_OK;  // no batched search was issued here
    CS_HIP(hipMemcpyAsync(w->h_overflow, w->bs.d_overflow, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                          (hipStream_t)stream));
    CS_HIP(hipMemsetAsync(w->bs.d_overflow + 1, 0, sizeof(uint32_t), (hipStream_t)stream));
    CS_HIP(hipStreamSynchronize((hipStream_t)stream));
    *overflowed = w->h_overflow[1];
    return CS_OK;
}

int32_t cs_index_release_stream(cs_index* h, void* stream) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    DeviceGuard g(h->device);
    std::vector<Workspace*> gone;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        for (auto it = h->by_stream.begin(); it != h->by_stream.end();) {
            if (it->first.first == (hipStream_t)stream) { gone.push_back(it->second); it = h->by_stream.erase(it); }
            else ++it;
        }
    }
    if (gone.empty()) return CS_OK;
    CS_HIP(hipStreamSynchronize((hipStream_t)stream));  // searches still using the scratch
    uint64_t fallbacks = 0;
    for (Workspace* w : gone) {
        if (w->bs.d_overflow) {  // overflows counted on the device so far stay in the handle's totals
            uint32_t v[3] = {0, 0, 0};
            CS_HIP(hipMemcpy(v, w->bs.d_overflow, sizeof v, hipMemcpyDeviceToHost));
            fallbacks += (uint64_t)v[2] + (v[0] ? 1 : 0);
        }
        w->release_all();
        delete w;
    }
    cs::merge_scratch_release(h->device, (hipStream_t)stream);
    std::lock_guard<std::mutex> lk(h->mu);
    h->batched_fallbacks += fallbacks;
    return CS_OK;
}

int32_t cs_merge_topk_device(int32_t device, const uint64_t* d_keys, uint32_t nlists, uint32_t nq,
                             uint32_t k, uint64_t* d_out_keys, float* d_out_cos,
                             uint32_t* d_out_ids, uint32_t* d_out_counts, void* stream) {
    return cs::merge_topk_device_impl(device, d_keys, nlists, nq, k, d_out_keys, d_out_cos, d_out_ids, d_out_counts,
                                      (hipStream_t)stream, 0, 0);
}

}  // extern "C"

namespace {
struct MergeScratch { uint64_t* a = nullptr; uint64_t* b = nullptr; size_t cap = 0; };
std::mutex g_merge_mu;
std::map<std::tuple<int, hipStream_t, std::thread::id>, MergeScratch> g_merge_pool;
}  // namespace

// the caller has synchronised `stream`
void cs::merge_scratch_release(int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_merge_mu);
    for (auto it = g_merge_pool.begin(); it != g_merge_pool.end();) {
        if (std::get<0>(it->first) == device && std::get<1>(it->first) == stream) {
            if (it->second.a) { (void)hipFree(it->second.a); (void)hipFree(it->second.b); }
            it = g_merge_pool.erase(it);
        } else ++it;
    }
}

int32_t cs::index_reserve(cs_index* h, uint64_t rows) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    DeviceGuard g(h->device);
    return grow(h, rows);
}

int32_t cs::index_append_from(cs_index* h, const float* d_rows, int src_device, uint64_t n, hipStream_t stream) {
    CS_TRY(check_append(h, n, h ? h->dim : 0));
    if (n == 0) return CS_OK;
    {
        DeviceGuard g(h->device);
        CS_TRY(grow(h, h->n_rows + n));
    }
    float* dst = h->d_corpus + (size_t)h->n_rows * h->dim;
    const size_t bytes = (size_t)n * h->dim * sizeof(float);
    DeviceGuard g(src_device);  // `stream` belongs to the source device
    if (src_device == h->device) CS_HIP(hipMemcpyAsync(dst, d_rows, bytes, hipMemcpyDeviceToDevice, stream));
    else {
        CS_HIP(hipMemcpyPeerAsync(dst, h->device, d_rows, src_device, bytes, stream));
        cs_index::ForeignAppend* slot = nullptr;
        for (auto& fe : h->foreign_appends)
            if (fe.device == src_device && fe.stream == stream) { slot = &fe; break; }
        if (slot) {  // the same stream again: re-recording its event covers this copy and every earlier one
            if (hipEventRecord(slot->done, stream) != hipSuccess) return fail(CS_ERR_HIP, "hipEventRecord on the peer stream failed");
        } else {
            hipEvent_t done = nullptr;
            CS_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            if (hipEventRecord(done, stream) != hipSuccess) {
                (void)hipEventDestroy(done);
                return fail(CS_ERR_HIP, "hipEventRecord on the peer stream failed");
            }
            h->foreign_appends.push_back({src_device, stream, done});
        }
    }
    finish_append(h, n, nullptr);
    return CS_OK;
}

int32_t cs::merge_topk_device_impl(int32_t device, const uint64_t* d_keys, uint32_t nlists, uint32_t nq, uint32_t k,
                                   uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_counts,
                                   hipStream_t stream, uint32_t remap_stripe, uint32_t remap_shards) {
    if (!d_keys || nlists == 0 || nq == 0 || k == 0 || k > CS_MAX_K)
        return fail(CS_ERR_BAD_ARG, "bad merge arguments");
    DeviceGuard g(device);
    const size_t tmp = merge_tmp_keys(nlists, nq, k);
    uint64_t *ta = nullptr, *tb = nullptr;
    if (tmp) {
        // More than one merge level (over 2048 keys per query; 8 shards x k <= 256 fit one): ping-pong scratch
        // kept per (device, stream, calling thread) and grown on demand — never freed or synchronised per call, so
        // the merge stays asynchronous between the all-gather and whatever the caller enqueues next.
        std::lock_guard<std::mutex> lk(g_merge_mu);
        MergeScratch& sc = g_merge_pool[std::make_tuple(device, (hipStream_t)stream, std::this_thread::get_id())];
        if (tmp > sc.cap) {
            // the old pair may still be in use by merges already enqueued on this stream: let them finish
            if (sc.a) { CS_HIP(hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(sc.a); (void)hipFree(sc.b); }
            sc = MergeScratch();
            CS_HIP(hipMalloc(&sc.a, tmp * sizeof(uint64_t)));
            CS_HIP(hipMalloc(&sc.b, tmp * sizeof(uint64_t)));
            sc.cap = tmp;
        }
        ta = sc.a;
        tb = sc.b;
    }
    return launch_merge(d_keys, nlists, nq, k, true, ta, tb, d_out_keys, d_out_cos, d_out_ids, d_out_counts,
                        (hipStream_t)stream, nullptr, remap_stripe, remap_shards);
}

extern "C" {

int32_t cs_index_read_rows(cs_index* h, uint64_t first_row, uint64_t n, float* out_rows) {
    if (!h || !out_rows) return fail(CS_ERR_BAD_ARG, "null argument");
    if (first_row + n > h->n_ids)
        return fail(CS_ERR_BAD_ARG, "rows [%llu, %llu) out of range (have %llu)",
                    (unsigned long long)first_row, (unsigned long long)(first_row + n),
                    (unsigned long long)h->n_ids);
    if (n == 0) return CS_OK;
    DeviceGuard g(h->device);
    CS_TRY(drain_appends(h));
    if (!h->h_ids.empty()) {  // compacted: rows are named by their ids (first_row = id - id_base); runs of neighbours in one copy
        uint64_t i = 0;
        while (i < n) {
            const uint32_t id = h->id_base + (uint32_t)(first_row + i);
            const auto it = std::lower_bound(h->h_ids.begin(), h->h_ids.end(), id);
            if (it == h->h_ids.end() || *it != id)
                return fail(CS_ERR_BAD_ARG, "row of id %u was deleted and reclaimed by cs_index_build", id);
            const uint64_t row = (uint64_t)(it - h->h_ids.begin());
            uint64_t run = 1;
            while (i + run < n && row + run < h->n_rows && h->h_ids[(size_t)(row + run)] == id + run) ++run;
            CS_HIP(hipMemcpy(out_rows + (size_t)i * h->dim, h->d_corpus + (size_t)row * h->dim, (size_t)run * h->dim * sizeof(float),
                             hipMemcpyDeviceToHost));
            i += run;
        }
        return CS_OK;
    }
    CS_HIP(hipMemcpy(out_rows, h->d_corpus + (size_t)first_row * h->dim,
                     (size_t)n * h->dim * sizeof(float), hipMemcpyDeviceToHost));
    return CS_OK;
}

int32_t cs_index_set_filter_min_queries(cs_index* h, uint32_t min_queries) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    h->filter_min_q = min_queries < 1 ? 1 : (int)min_queries;
    return CS_OK;
}

int32_t cs_index_set_single_query_route(cs_index* h, int32_t route) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    if (route != CS_ROUTE_COST && route != CS_ROUTE_STREAM && route != CS_ROUTE_FILTER)
        return fail(CS_ERR_BAD_ARG, "route must be CS_ROUTE_COST, CS_ROUTE_STREAM or CS_ROUTE_FILTER, got %d", route);
    h->single_route = route;
    return CS_OK;
}

int32_t cs_index_debug_counters(cs_index* h, uint64_t* batched_searches, uint64_t* batched_fallbacks) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    DeviceGuard g(h->device);
    // device-API searches count their overflows on the device (no host round trip per search): drain and read
    uint64_t dev_fallbacks = 0;
    std::vector<Workspace*> ws;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        for (auto& kv : h->by_stream) ws.push_back(kv.second);
    }
    for (Workspace* w : ws) {
        if (!w->bs.d_overflow) continue;
        uint32_t v[3] = {0, 0, 0};
        // a caller's stream may have been destroyed since (cs_index_release_stream is the orderly way): the blocking
        // copy below orders against everything still running on the device either way
        if (hipStreamSynchronize(w->stream) != hipSuccess) (void)hipGetLastError();
        CS_HIP(hipMemcpy(v, w->bs.d_overflow, sizeof v, hipMemcpyDeviceToHost));
        dev_fallbacks += (uint64_t)v[2] + (v[0] ? 1 : 0);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    if (batched_searches) *batched_searches = h->batched_searches;
    if (batched_fallbacks) *batched_fallbacks = h->batched_fallbacks + dev_fallbacks;
    return CS_OK;
}

int32_t cs_index_filter_state(cs_index* h, int32_t* copy, float* spread, uint64_t* int8_reruns) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    std::lock_guard<std::mutex> lk(h->mu);
    const bool i8 = q8_serves(h);
    if (copy) *copy = i8 ? 2 : h->use_split ? 1 : 0;
    if (spread) *spread = h->q8_spread;
    if (int8_reruns) *int8_reruns = h->q8_reruns;
    return CS_OK;
}

int32_t cs_index_filter_copies(cs_index* h, int32_t* has_int8, int32_t* has_f16, uint64_t* filter_bytes) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    std::lock_guard<std::mutex> lk(h->filter_mu);
    const bool i8 = h->d_q8 && h->q8_rows > 0, f16 = h->d_split && h->split_rows > 0;
    if (has_int8) *has_int8 = i8 ? 1 : 0;
    if (has_f16) *has_f16 = f16 ? 1 : 0;
    if (filter_bytes)
        *filter_bytes = (h->d_q8 ? ((uint64_t)h->capacity + 255) / 256 * 256 * h->dim : 0) +
                        (h->d_split ? (uint64_t)h->split_cap * h->dim * 2 : 0);
    return CS_OK;
}

int32_t cs_index_profile(cs_index* h, int32_t enable) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    std::lock_guard<std::mutex> lk(h->mu);
    h->profile = enable != 0;
    return CS_OK;
}

int32_t cs_index_profile_read(cs_index* h, double* scan_ms, uint64_t* scan_launches,
                              double* merge_ms, int32_t reset) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null index handle");
    DeviceGuard g(h->device);
    std::vector<EventTriple> pend;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        pend.swap(h->pending);
    }
    double s_ms = 0.0, m_ms = 0.0;
    for (auto& t : pend) {
        CS_HIP(hipEventSynchronize(t.e2));
        float a = 0.f, b = 0.f;
        CS_HIP(hipEventElapsedTime(&a, t.e0, t.e1));
        CS_HIP(hipEventElapsedTime(&b, t.e1, t.e2));
        s_ms += a;
        m_ms += b;
        (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); (void)hipEventDestroy(t.e2);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    h->scan_ms += s_ms;
    h->merge_ms += m_ms;
    h->scan_launches += pend.size();
    if (scan_ms) *scan_ms = h->scan_ms;
    if (merge_ms) *merge_ms = h->merge_ms;
    if (scan_launches) *scan_launches = h->scan_launches;
    if (reset) { h->scan_ms = 0.0; h->merge_ms = 0.0; h->scan_launches = 0; }
    return CS_OK;
}

}  // extern "C"
