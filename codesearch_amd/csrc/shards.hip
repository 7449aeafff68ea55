// shards.hip — cs_shards_*: the vector half of the reference's VectorStore
// (/root/reference/src/vectordb/store.rs:94-750) over SEVERAL GPUs of one node, owned by ONE
// process — what a Rust `VectorStore` behind `codesearch search` can call: the reference calls
// `store.search` from a single process (src/search/mod.rs:508-511).  SURVEY.md §8e.
//
// Layout: the corpus is row-sharded in stripes of `stripe` consecutive ids, dealt round-robin:
//     stripe t = id / stripe,  shard = t % N,  local row = (t / N) * stripe + id % stripe.
// With stripe = rows per GPU and a corpus of up to N * stripe rows this is the contiguous-range layout of
// BASELINE.json config 5 (shard g holds ids [g * stripe, (g + 1) * stripe)); with a smaller stripe a corpus
// of any size is spread over all GPUs while ids stay contiguous from next_id as in the reference
// (store.rs:659-685).  Every shard is a plain cs_index with id_base 0.
//
// One search = ONE exchange step over xGMI:
//   1. the queries ([nq, dim] f32, pinned host memory) go to every shard's device on that shard's stream
//      (the broadcast: nq * dim * 4 bytes per GPU);
//   2. every shard runs cs_index_search_device on its own stream — the single-query streaming scan or the
//      batched filter + refine path, unchanged — and leaves its [nq, k] packed keys in a local buffer that ONE
//      hipMemcpyPeerAsync moves into its slot of the root GPU's gather buffer (nq * k * 8 bytes per shard over
//      xGMI: a one-step direct gather, no ring).  With CS_SHARDS_DIRECT=1 and peer access the last
//      kernel of the search writes the keys straight into that slot instead; opt-in until a run on two distinct
//      devices has compared both (tests/test_gpu_shards.py::test_shards_over_distinct_devices, bench.py --gpus N);
//   3. the root stream waits for one event per shard and runs the key merge (scan.hip merge_topk_kernel),
//      which turns local row numbers into global ids while it reads the lists; the merged keys land in
//      pinned host memory.
// top-k of a union = top-k of the per-shard top-ks and a shard's local order is its global order, so the
// result equals the single-index search of the whole corpus bit for bit.
// The multi-process variant (one rank per GPU, RCCL all-gather) is codesearch_amd/sharded.py.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "scan.hpp"

using namespace cs;

namespace {

struct ShardCtx {               // per shard, inside one search context
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    float* d_queries = nullptr; size_t q_cap = 0;     // on the shard's device
    const float* q_src = nullptr;                     // where the current search's queries are read from on that device
    uint64_t* d_keys = nullptr; size_t key_cap = 0;   // local result buffer (copy path only)
};

struct SearchCtx {              // one in-flight search; pooled (search is `&self`: concurrent callers)
    std::vector<ShardCtx> sh;
    hipStream_t root_stream = nullptr;
    uint64_t* d_gathered = nullptr; size_t gathered_cap = 0;  // [N][nq][k] on the root device
    float* h_queries = nullptr; size_t h_q_cap = 0;           // pinned, visible to every device
    uint64_t* h_keys = nullptr; size_t h_key_cap = 0;         // pinned: merged [nq][k]
    uint32_t* h_meta = nullptr;                               // pinned: variant merge count + high-confidence flag
    // device-pointer searches (cs_shards_search_device): `ready` is recorded on the caller's stream when a search is
    // enqueued (queries valid, earlier work of that stream done), `merged` behind its merge; the shard streams of the
    // NEXT search through this context wait for both, so the gather buffer is never rewritten under a merge
    hipEvent_t ready = nullptr, merged = nullptr;
    bool merged_pending = false;
};

}  // namespace

struct cs_shards {
    uint32_t dim = 0, n = 0;
    uint64_t stripe = 0;
    std::vector<int> devices;
    std::vector<cs_index*> idx;
    int root = 0;          // device of shard 0: gathers and merges
    bool direct = false;   // shards write their keys straight into the root's gather buffer (CS_SHARDS_DIRECT=1)
    bool force_copy = false;  // CS_SHARDS_DIRECT=0: even a shard on the root device goes through its local buffer + copy
    uint64_t next = 0;     // rows appended so far == next_id
    bool poisoned = false; // a piece of an append failed after others went in: the stripe map no longer matches the rows
    bool built = false;
    std::mutex mu;
    std::vector<SearchCtx*> pool;
    std::vector<SearchCtx*> all;  // every context ever created (cs_shards_search_status visits their shard streams)
};

namespace {

struct Piece { uint32_t shard; uint64_t local_row, count, offset; };

// ids [first, first + n) as runs that stay inside one stripe
std::vector<Piece> pieces_of(const cs_shards* h, uint64_t first, uint64_t n) {
    std::vector<Piece> out;
    uint64_t id = first, off = 0;
    while (off < n) {
        const uint64_t t = id / h->stripe, in = id % h->stripe;
        const uint64_t cnt = std::min<uint64_t>(h->stripe - in, n - off);
        out.push_back(Piece{(uint32_t)(t % h->n), (t / h->n) * h->stripe + in, cnt, off});
        id += cnt;
        off += cnt;
    }
    return out;
}

void free_ctx(cs_shards* h, SearchCtx* c) {
    for (uint32_t s = 0; s < c->sh.size(); ++s) {
        DeviceGuard g(h->devices[s]);
        ShardCtx& x = c->sh[s];
        if (x.stream) (void)hipStreamSynchronize(x.stream);
        if (x.d_queries) (void)hipFree(x.d_queries);
        if (x.d_keys) (void)hipFree(x.d_keys);
        if (x.done) (void)hipEventDestroy(x.done);
        if (x.stream) (void)hipStreamDestroy(x.stream);
    }
    DeviceGuard g(h->root);
    if (c->root_stream) { (void)hipStreamSynchronize(c->root_stream); (void)hipStreamDestroy(c->root_stream); }
    if (c->d_gathered) (void)hipFree(c->d_gathered);
    if (c->h_queries) (void)hipHostFree(c->h_queries);
    if (c->h_keys) (void)hipHostFree(c->h_keys);
    if (c->h_meta) (void)hipHostFree(c->h_meta);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->merged) (void)hipEventDestroy(c->merged);
    delete c;
}

int32_t new_ctx(cs_shards* h, SearchCtx** out) {
    SearchCtx* c = new SearchCtx();
    c->sh.resize(h->n);
    auto bail = [&](int32_t s) { free_ctx(h, c); return s; };
    for (uint32_t s = 0; s < h->n; ++s) {
        DeviceGuard g(h->devices[s]);
        if (hipStreamCreateWithFlags(&c->sh[s].stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->sh[s].done, hipEventDisableTiming) != hipSuccess)
            return bail(fail(CS_ERR_HIP, "could not create a stream/event on device %d", h->devices[s]));
    }
    DeviceGuard g(h->root);
    if (hipStreamCreateWithFlags(&c->root_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->merged, hipEventDisableTiming) != hipSuccess)
        return bail(fail(CS_ERR_HIP, "could not create the root stream"));
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->all.push_back(c);
    }
    *out = c;
    return CS_OK;
}

int32_t reserve_ctx(cs_shards* h, SearchCtx* c, uint32_t nq, uint32_t k, bool host_io) {
    const size_t qn = (size_t)nq * h->dim, kn = (size_t)nq * k;
    if (host_io && qn > c->h_q_cap) {
        if (c->h_queries) (void)hipHostFree(c->h_queries);
        c->h_queries = nullptr; c->h_q_cap = 0;
        CS_HIP(hipHostMalloc(&c->h_queries, qn * sizeof(float), hipHostMallocPortable | hipHostMallocMapped));
        c->h_q_cap = qn;
    }
    if (host_io && kn > c->h_key_cap) {
        if (c->h_keys) (void)hipHostFree(c->h_keys);
        c->h_keys = nullptr; c->h_key_cap = 0;
        CS_HIP(hipHostMalloc(&c->h_keys, kn * sizeof(uint64_t), hipHostMallocPortable | hipHostMallocMapped));
        c->h_key_cap = kn;
    }
    {
        DeviceGuard g(h->root);
        if (kn * (h->n + 1) > c->gathered_cap) {  // [N][nq][k] shard slots + one merged [nq][k] block (variant searches)
            if (c->d_gathered) { CS_HIP(hipStreamSynchronize(c->root_stream)); (void)hipFree(c->d_gathered); }
            c->d_gathered = nullptr; c->gathered_cap = 0;
            CS_HIP(hipMalloc(&c->d_gathered, kn * (h->n + 1) * sizeof(uint64_t)));
            c->gathered_cap = kn * (h->n + 1);
        }
        if (!c->h_meta) CS_HIP(hipHostMalloc(&c->h_meta, 2 * sizeof(uint32_t), hipHostMallocPortable | hipHostMallocMapped));
    }
    for (uint32_t s = 0; s < h->n; ++s) {
        DeviceGuard g(h->devices[s]);
        ShardCtx& x = c->sh[s];
        if (qn > x.q_cap) {
            if (x.d_queries) (void)hipFree(x.d_queries);
            x.d_queries = nullptr; x.q_cap = 0;
            CS_HIP(hipMalloc(&x.d_queries, qn * sizeof(float)));
            x.q_cap = qn;
        }
        if (!h->direct && (h->force_copy || h->devices[s] != h->root) && kn > x.key_cap) {
            if (x.d_keys) (void)hipFree(x.d_keys);
            x.d_keys = nullptr; x.key_cap = 0;
            CS_HIP(hipMalloc(&x.d_keys, kn * sizeof(uint64_t)));
            x.key_cap = kn;
        }
    }
    return CS_OK;
}

// Enqueue the search of queries [q0, q0 + qn) on shard s; its keys go to gathered[s][q0 ..][k]
// (row stride of the gather buffer is the FULL nq).  src != null: the [nq, dim] queries are first brought to the
// shard's device — from pinned host memory (src_device < 0) or from HBM of device src_device (the root of a
// device-pointer search; a shard living on that device reads them in place).
int32_t enqueue_shard(cs_shards* h, SearchCtx* c, uint32_t s, uint32_t q0, uint32_t qn, uint32_t nq, uint32_t k,
                      const float* src, int src_device) {
    DeviceGuard g(h->devices[s]);
    ShardCtx& x = c->sh[s];
    const size_t qbytes = (size_t)nq * h->dim * sizeof(float);
    const float* q = src ? x.d_queries : x.q_src;  // src == null: a rerun of slices of the search enqueued before
    if (src && src_device < 0)
        CS_HIP(hipMemcpyAsync(x.d_queries, src, qbytes, hipMemcpyHostToDevice, x.stream));
    else if (src && src_device == h->devices[s])
        q = src;
    else if (src)
        CS_HIP(hipMemcpyPeerAsync(x.d_queries, h->devices[s], src, src_device, qbytes, x.stream));
    x.q_src = q;
    uint64_t* slot = c->d_gathered + ((size_t)s * nq + q0) * k;
    // default: a shard on another device fills a local buffer that one peer copy moves; one on the root device writes
    // its slot itself
    const bool local = !h->direct && (h->force_copy || h->devices[s] != h->root);
    uint64_t* dst = local ? x.d_keys + (size_t)q0 * k : slot;
    CS_TRY(cs_index_search_device(h->idx[s], q + (size_t)q0 * h->dim, qn, h->dim, k, dst, nullptr, nullptr,
                                  nullptr, x.stream));
    if (local)
        CS_HIP(hipMemcpyPeerAsync(slot, h->root, dst, h->devices[s], (size_t)qn * k * sizeof(uint64_t), x.stream));
    return CS_OK;
}

// Before a context's shard streams take new work: the previous device-pointer search through it may still be merging
// out of the gather buffer on a caller's stream.
int32_t order_behind_previous(cs_shards* h, SearchCtx* c) {
    if (!c->merged_pending) return CS_OK;
    for (uint32_t s = 0; s < h->n; ++s) {
        DeviceGuard g(h->devices[s]);
        CS_HIP(hipStreamWaitEvent(c->sh[s].stream, c->merged, 0));
    }
    return CS_OK;
}

SearchCtx* take_ctx(cs_shards* h) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->pool.empty()) return nullptr;
    SearchCtx* c = h->pool.back();
    h->pool.pop_back();
    return c;
}

}  // namespace

namespace {

// Slices of <= kGatedMaxQ queries carry their own exact rerun (index.hip); a shard whose larger search overflowed a
// candidate buffer is redone that way.  Returns in *again whether anything was re-enqueued (then merge again).
int32_t rerun_overflowed(cs_shards* h, SearchCtx* c, uint32_t nq, uint32_t k, bool* again) {
    *again = false;
    if (nq <= kGatedMaxQ) return CS_OK;
    for (uint32_t s = 0; s < h->n; ++s) {
        uint32_t ov = 0;
        CS_TRY(cs_index_search_status(h->idx[s], c->sh[s].stream, &ov));
        if (!ov) continue;
        *again = true;
        for (uint32_t q0 = 0; q0 < nq; q0 += kGatedMaxQ)
            CS_TRY(enqueue_shard(h, c, s, q0, std::min<uint32_t>(kGatedMaxQ, nq - q0), nq, k, nullptr, -1));
        DeviceGuard g(h->devices[s]);
        CS_HIP(hipEventRecord(c->sh[s].done, c->sh[s].stream));
    }
    return CS_OK;
}

void drain_ctx(cs_shards* h, SearchCtx* c) {  // leave nothing in flight behind a failed call
    for (uint32_t s = 0; s < h->n; ++s) { DeviceGuard g(h->devices[s]); (void)hipStreamSynchronize(c->sh[s].stream); }
    DeviceGuard g(h->root);
    (void)hipStreamSynchronize(c->root_stream);
}

int32_t check_shards_search(const cs_shards* h, uint32_t nq, uint32_t dim, uint32_t k) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null shards handle");
    if (dim != h->dim)  // store.rs:432-438
        return fail(CS_ERR_DIM_MISMATCH, "Query embedding dimension mismatch: expected %u, got %u", h->dim, dim);
    if (!h->built)  // store.rs:440-444
        return fail(CS_ERR_NOT_BUILT, "Index not built. Call build_index() after inserting chunks.");
    if (nq == 0 || nq > CS_MAX_QUERIES) return fail(CS_ERR_BAD_ARG, "nq must be in 1..%u, got %u", CS_MAX_QUERIES, nq);
    if (k == 0 || k > CS_MAX_K) return fail(CS_ERR_BAD_ARG, "k must be in 1..%u, got %u", CS_MAX_K, k);
    return CS_OK;
}

// variants != null: after the shard merge the nq lists are merged as query variants (scan.hip merge_variants_kernel) and
// out_cos / out_ids hold ONE list of k; variants[0] = count, variants[1] = high-confidence flag.
int32_t shards_search_impl(cs_shards* h, const float* queries, uint32_t nq, uint32_t dim, uint32_t k, float* out_cos,
                           uint32_t* out_ids, uint32_t* out_counts, uint32_t* variants) {
    CS_TRY(check_shards_search(h, nq, dim, k));
    if (!queries || !out_cos || !out_ids || !out_counts) return fail(CS_ERR_BAD_ARG, "null buffer");
    SearchCtx* c = take_ctx(h);
    if (!c) CS_TRY(new_ctx(h, &c));
    const int32_t st = [&]() -> int32_t {
        CS_TRY(reserve_ctx(h, c, nq, k, true));
        CS_TRY(order_behind_previous(h, c));
        memcpy(c->h_queries, queries, (size_t)nq * h->dim * sizeof(float));
        for (uint32_t s = 0; s < h->n; ++s) {
            CS_TRY(enqueue_shard(h, c, s, 0, nq, nq, k, c->h_queries, -1));
            DeviceGuard g(h->devices[s]);
            CS_HIP(hipEventRecord(c->sh[s].done, c->sh[s].stream));
        }
        auto merge = [&]() -> int32_t {
            DeviceGuard g(h->root);
            for (uint32_t s = 0; s < h->n; ++s) CS_HIP(hipStreamWaitEvent(c->root_stream, c->sh[s].done, 0));
            if (variants) {
                // shard merge into the spare block behind the shard slots, then the variant merge into pinned memory
                uint64_t* d_merged = c->d_gathered + (size_t)h->n * nq * k;
                CS_TRY(merge_topk_device_impl(h->root, c->d_gathered, h->n, nq, k, d_merged, nullptr, nullptr, nullptr,
                                              c->root_stream, (uint32_t)h->stripe, h->n));
                CS_TRY(launch_merge_variants(d_merged, nq, k, k, c->h_keys, nullptr, nullptr, c->h_meta, c->h_meta + 1,
                                             c->root_stream));
            } else {
                CS_TRY(merge_topk_device_impl(h->root, c->d_gathered, h->n, nq, k, c->h_keys, nullptr, nullptr, nullptr,
                                              c->root_stream, (uint32_t)h->stripe, h->n));
            }
            CS_HIP(hipStreamSynchronize(c->root_stream));
            return CS_OK;
        };
        CS_TRY(merge());
        c->merged_pending = false;  // every stream of the context is idle now
        bool again = false;
        CS_TRY(rerun_overflowed(h, c, nq, k, &again));
        if (again) CS_TRY(merge());
        if (variants) {
            for (uint32_t j = 0; j < k; ++j) {
                const uint64_t key = c->h_keys[j];
                out_cos[j] = key ? key_cos(key) : 0.0f;
                out_ids[j] = key ? key_id(key) : 0xFFFFFFFFu;
            }
            variants[0] = c->h_meta[0];
            variants[1] = c->h_meta[1];
            return CS_OK;
        }
        for (uint32_t q = 0; q < nq; ++q) {  // keys are best-first, 0 = empty slot
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < k; ++j) {
                const uint64_t key = c->h_keys[(size_t)q * k + j];
                if (key) ++cnt;
                out_cos[(size_t)q * k + j] = key ? key_cos(key) : 0.0f;
                out_ids[(size_t)q * k + j] = key ? key_id(key) : 0xFFFFFFFFu;
            }
            out_counts[q] = cnt;
        }
        return CS_OK;
    }();
    if (st != CS_OK) drain_ctx(h, c);
    std::lock_guard<std::mutex> lk(h->mu);
    h->pool.push_back(c);
    return st;
}

// cs_shards_search_device: queries and outputs in HBM of the root device, everything enqueued, nothing waited for.
//   caller's stream:  --ready--------------------------------(wait done[0..N))--merge--merged-->
//   shard s stream :    (wait ready, wait previous merged) copy queries, search, gather copy --done[s]
int32_t shards_search_device_impl(cs_shards* h, const float* d_queries, uint32_t nq, uint32_t dim, uint32_t k,
                                  uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_counts,
                                  hipStream_t stream) {
    CS_TRY(check_shards_search(h, nq, dim, k));
    if (!d_queries) return fail(CS_ERR_BAD_ARG, "d_queries is null");
    SearchCtx* c = take_ctx(h);
    if (!c) CS_TRY(new_ctx(h, &c));
    const int32_t st = [&]() -> int32_t {
        CS_TRY(reserve_ctx(h, c, nq, k, false));
        CS_TRY(order_behind_previous(h, c));
        {
            DeviceGuard g(h->root);
            CS_HIP(hipEventRecord(c->ready, stream));
        }
        for (uint32_t s = 0; s < h->n; ++s) {
            {
                DeviceGuard g(h->devices[s]);
                CS_HIP(hipStreamWaitEvent(c->sh[s].stream, c->ready, 0));
            }
            CS_TRY(enqueue_shard(h, c, s, 0, nq, nq, k, d_queries, h->root));
            DeviceGuard g(h->devices[s]);
            CS_HIP(hipEventRecord(c->sh[s].done, c->sh[s].stream));
        }
        DeviceGuard g(h->root);
        for (uint32_t s = 0; s < h->n; ++s) CS_HIP(hipStreamWaitEvent(stream, c->sh[s].done, 0));
        CS_TRY(merge_topk_device_impl(h->root, c->d_gathered, h->n, nq, k, d_out_keys, d_out_cos, d_out_ids, d_out_counts,
                                      stream, (uint32_t)h->stripe, h->n));
        CS_HIP(hipEventRecord(c->merged, stream));
        c->merged_pending = true;
        return CS_OK;
    }();
    if (st != CS_OK) {
        drain_ctx(h, c);
        DeviceGuard g(h->root);
        (void)hipStreamSynchronize(stream);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    h->pool.push_back(c);
    return st;
}

}  // namespace

extern "C" {

int32_t cs_shards_create(uint32_t dim, uint32_t nshards, const int32_t* devices, uint64_t rows_per_stripe,
                         uint64_t capacity_rows, cs_shards** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (nshards == 0 || nshards > 64 || !devices) return fail(CS_ERR_BAD_ARG, "nshards must be in 1..64 with a device list");
    if (rows_per_stripe == 0) return fail(CS_ERR_BAD_ARG, "rows_per_stripe must be > 0");
    if (rows_per_stripe > 0xffffffffull) return fail(CS_ERR_BAD_ARG, "rows_per_stripe must fit u32 (ids are u32, store.rs:97)");
    cs_shards* h = new cs_shards();
    h->dim = dim;
    h->n = nshards;
    h->stripe = rows_per_stripe;
    h->devices.assign(devices, devices + nshards);
    h->root = devices[0];
    const uint64_t per_shard = (capacity_rows + nshards - 1) / nshards;
    for (uint32_t s = 0; s < nshards; ++s) {
        cs_index* ix = nullptr;
        // whole stripes, so a shard never grows in the middle of a bulk append that was sized up front
        const uint64_t cap = per_shard ? (per_shard + rows_per_stripe - 1) / rows_per_stripe * rows_per_stripe : 0;
        const int32_t st = cs_index_create(dim, cap, devices[s], 0, &ix);
        if (st != CS_OK) { cs_shards_destroy(h); return st; }
        h->idx.push_back(ix);
    }
    // direct gather (opt-in, CS_SHARDS_DIRECT=1): every shard's device must be able to write the root's memory.
    // Default: the keys travel by one hipMemcpyPeerAsync per shard (needs no peer mapping).
    bool direct = false;
    if (const char* e = std::getenv("CS_SHARDS_DIRECT")) { direct = (e[0] == '1'); h->force_copy = (e[0] == '0'); }
    for (uint32_t s = 0; s < nshards && direct; ++s) {
        if (devices[s] == h->root) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[s], h->root) != hipSuccess || !can) { direct = false; break; }
        DeviceGuard g(devices[s]);
        const hipError_t e = hipDeviceEnablePeerAccess(h->root, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { direct = false; }
        (void)hipGetLastError();
    }
    h->direct = direct;
    *out = h;
    return CS_OK;
}

void cs_shards_destroy(cs_shards* h) {
    if (!h) return;
    for (SearchCtx* c : h->all) free_ctx(h, c);
    for (cs_index* ix : h->idx) cs_index_destroy(ix);
    delete h;
}

// Appends are all-or-nothing across the shards (a half-applied append would leave `next` behind the rows some
// shards already hold and shift every later row): capacity is reserved on every touched shard first — the only step
// that can fail for lack of memory — and `next` advances only after every piece went in.
int32_t reserve_pieces(cs_shards* h, const std::vector<Piece>& pieces) {
    std::vector<uint64_t> need(h->n, 0);
    for (const Piece& p : pieces) need[p.shard] = std::max<uint64_t>(need[p.shard], p.local_row + p.count);
    for (uint32_t s = 0; s < h->n; ++s)
        if (need[s]) CS_TRY(index_reserve(h->idx[s], need[s]));
    return CS_OK;
}

int32_t check_shards_append(cs_shards* h, uint64_t n, uint32_t dim) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null shards handle");
    if (h->poisoned)
        return fail(CS_ERR_BAD_ARG, "sharded store is inconsistent after a failed append; clear() or reopen it");
    if (dim != h->dim)  // store.rs:667-671
        return fail(CS_ERR_DIM_MISMATCH, "Embedding dimension mismatch: expected %u, got %u", h->dim, dim);
    if (h->next + n > 0xffffffffull) return fail(CS_ERR_BAD_ARG, "id space exhausted: ids are u32 (store.rs:97)");
    return CS_OK;
}

void finish_shards_append(cs_shards* h, uint64_t n, uint32_t* out_ids) {
    if (out_ids)
        for (uint64_t i = 0; i < n; ++i) out_ids[i] = (uint32_t)(h->next + i);  // store.rs:684
    h->next += n;
    if (n) h->built = false;  // store.rs:682
}

int32_t cs_shards_add(cs_shards* h, const float* rows, uint64_t n, uint32_t dim, uint32_t* out_ids) {
    CS_TRY(check_shards_append(h, n, dim));
    if (n == 0) return CS_OK;
    if (!rows) return fail(CS_ERR_BAD_ARG, "rows is null");
    const std::vector<Piece> pieces = pieces_of(h, h->next, n);
    CS_TRY(reserve_pieces(h, pieces));
    for (const Piece& p : pieces) {
        const int32_t st = cs_index_add(h->idx[p.shard], rows + (size_t)p.offset * dim, p.count, dim, nullptr);
        if (st != CS_OK) { h->poisoned = true; return st; }  // a HIP failure after the reservation: not recoverable here
    }
    finish_shards_append(h, n, out_ids);
    return CS_OK;
}

// Rows already in HBM of `src_device` (an encoder replica's pooled + normalised output): every piece goes to its
// shard with one asynchronous copy on `stream` (a stream of src_device) — device-to-device in place when the shard
// lives on src_device, over xGMI otherwise.  d_rows must stay valid until `stream` has passed this point.
int32_t cs_shards_add_device(cs_shards* h, const float* d_rows, int32_t src_device, uint64_t n, uint32_t dim,
                             uint32_t* out_ids, void* stream) {
    CS_TRY(check_shards_append(h, n, dim));
    if (n == 0) return CS_OK;
    if (!d_rows) return fail(CS_ERR_BAD_ARG, "d_rows is null");
    const std::vector<Piece> pieces = pieces_of(h, h->next, n);
    CS_TRY(reserve_pieces(h, pieces));
    for (const Piece& p : pieces) {
        const int32_t st = index_append_from(h->idx[p.shard], d_rows + (size_t)p.offset * dim, src_device, p.count,
                                             (hipStream_t)stream);
        if (st != CS_OK) { h->poisoned = true; return st; }
    }
    finish_shards_append(h, n, out_ids);
    return CS_OK;
}

// Where the next n rows will live: run i = rows [first[i], first[i] + count[i]) of the append, all on shard[i]; runs
// ascend and cover [0, n).  *n_runs = the number of runs; the arrays are filled when max_runs holds them all.
int32_t cs_shards_plan_append(const cs_shards* h, uint64_t n, uint32_t max_runs, uint32_t* shard, uint64_t* first,
                              uint64_t* count, uint32_t* n_runs) {
    if (!h || !n_runs) return fail(CS_ERR_BAD_ARG, "null argument");
    const std::vector<Piece> pieces = pieces_of(h, h->next, n);
    *n_runs = (uint32_t)pieces.size();
    if (max_runs < pieces.size() || !shard || !first || !count) return CS_OK;
    for (size_t i = 0; i < pieces.size(); ++i) { shard[i] = pieces[i].shard; first[i] = pieces[i].offset; count[i] = pieces[i].count; }
    return CS_OK;
}

// cs_shards_add_device for rows that sit in several buffers: part i = the next counts[i] rows, in HBM of
// src_devices[i] at d_rows[i] (encoder replicas on several GPUs, each holding the rows of its own shards).  The parts
// must be the runs cs_shards_plan_append reports for their total, in that order.  Copies are asynchronous on the null
// stream of each source device.
int32_t cs_shards_add_device_parts(cs_shards* h, uint32_t nparts, const float* const* d_rows, const int32_t* src_devices,
                                   const uint64_t* counts, uint32_t dim, uint32_t* out_ids) {
    if (nparts && (!d_rows || !src_devices || !counts)) return fail(CS_ERR_BAD_ARG, "null argument");
    uint64_t n = 0;
    for (uint32_t i = 0; i < nparts; ++i) n += counts[i];
    CS_TRY(check_shards_append(h, n, dim));
    if (n == 0) return CS_OK;
    const std::vector<Piece> pieces = pieces_of(h, h->next, n);
    if (pieces.size() != nparts) return fail(CS_ERR_BAD_ARG, "parts do not match the append plan (%zu runs, got %u)", pieces.size(), nparts);
    for (uint32_t i = 0; i < nparts; ++i)
        if (pieces[i].count != counts[i] || !d_rows[i])
            return fail(CS_ERR_BAD_ARG, "part %u does not match the append plan", i);
    CS_TRY(reserve_pieces(h, pieces));
    for (uint32_t i = 0; i < nparts; ++i) {
        const int32_t st = index_append_from(h->idx[pieces[i].shard], d_rows[i], src_devices[i], counts[i], nullptr);
        if (st != CS_OK) { h->poisoned = true; return st; }
    }
    finish_shards_append(h, n, out_ids);
    return CS_OK;
}

int32_t cs_shards_add_synthetic(cs_shards* h, uint64_t n, uint64_t seed, uint64_t first_row, uint32_t* out_first_id) {
    CS_TRY(check_shards_append(h, n, h ? h->dim : 0));
    const std::vector<Piece> pieces = pieces_of(h, h->next, n);
    CS_TRY(reserve_pieces(h, pieces));
    for (const Piece& p : pieces) {
        const int32_t st = cs_index_add_synthetic(h->idx[p.shard], p.count, seed, first_row + p.offset, nullptr);
        if (st != CS_OK) { h->poisoned = true; return st; }
    }
    if (out_first_id) *out_first_id = (uint32_t)h->next;
    finish_shards_append(h, n, nullptr);
    return CS_OK;
}

int32_t cs_shards_remove(cs_shards* h, const uint32_t* ids, uint64_t n, uint64_t* removed) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null shards handle");
    if (removed) *removed = 0;
    if (n == 0) return CS_OK;
    if (!ids) return fail(CS_ERR_BAD_ARG, "ids is null");
    std::vector<std::vector<uint32_t>> local(h->n);
    for (uint64_t i = 0; i < n; ++i) {
        if (ids[i] >= h->next) continue;  // unknown id: ignored (store.rs:594)
        const uint64_t t = ids[i] / h->stripe;
        local[t % h->n].push_back((uint32_t)((t / h->n) * h->stripe + ids[i] % h->stripe));
    }
    uint64_t total = 0;
    for (uint32_t s = 0; s < h->n; ++s) {
        if (local[s].empty()) continue;
        uint64_t r = 0;
        CS_TRY(cs_index_remove(h->idx[s], local[s].data(), local[s].size(), &r));
        total += r;
    }
    if (total) h->built = false;  // store.rs:604-606
    if (removed) *removed = total;
    return CS_OK;
}

int32_t cs_shards_build(cs_shards* h) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null shards handle");
    for (cs_index* ix : h->idx) CS_TRY(cs_index_build(ix));
    h->built = true;  // store.rs:428
    return CS_OK;
}

int32_t cs_shards_clear(cs_shards* h) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null shards handle");
    for (cs_index* ix : h->idx) CS_TRY(cs_index_clear(ix));
    h->next = 0;
    h->built = false;
    h->poisoned = false;
    return CS_OK;
}

int32_t cs_shards_is_built(const cs_shards* h) { return h && h->built ? 1 : 0; }
uint64_t cs_shards_len(const cs_shards* h) {
    uint64_t n = 0;
    if (h) for (cs_index* ix : h->idx) n += cs_index_len(ix);
    return n;
}
uint32_t cs_shards_next_id(const cs_shards* h) { return h ? (uint32_t)h->next : 0; }
uint32_t cs_shards_dim(const cs_shards* h) { return h ? h->dim : 0; }
uint32_t cs_shards_count(const cs_shards* h) { return h ? h->n : 0; }
int32_t cs_shards_direct_gather(const cs_shards* h) { return h && h->direct ? 1 : 0; }
uint64_t cs_shards_stored_rows(const cs_shards* h) {
    uint64_t n = 0;
    if (h) for (cs_index* ix : h->idx) n += cs_index_stored_rows(ix);
    return n;
}
uint64_t cs_shards_shard_len(const cs_shards* h, uint32_t shard) {
    return h && shard < h->n ? cs_index_len(h->idx[shard]) : 0;
}

int32_t cs_shards_read_rows(cs_shards* h, uint64_t first_id, uint64_t n, float* out_rows) {
    if (!h || !out_rows) return fail(CS_ERR_BAD_ARG, "null argument");
    if (first_id + n > h->next)
        return fail(CS_ERR_BAD_ARG, "rows [%llu, %llu) out of range (have %llu)", (unsigned long long)first_id,
                    (unsigned long long)(first_id + n), (unsigned long long)h->next);
    for (const Piece& p : pieces_of(h, first_id, n))
        CS_TRY(cs_index_read_rows(h->idx[p.shard], p.local_row, p.count, out_rows + (size_t)p.offset * h->dim));
    return CS_OK;
}

int32_t cs_shards_search(cs_shards* h, const float* queries, uint32_t nq, uint32_t dim, uint32_t k, float* out_cos,
                         uint32_t* out_ids, uint32_t* out_counts) {
    return shards_search_impl(h, queries, nq, dim, k, out_cos, out_ids, out_counts, nullptr);
}

int32_t cs_shards_search_device(cs_shards* h, const float* d_queries, uint32_t nq, uint32_t dim, uint32_t k,
                                uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_counts,
                                void* stream) {
    return shards_search_device_impl(h, d_queries, nq, dim, k, d_out_keys, d_out_cos, d_out_ids, d_out_counts,
                                     (hipStream_t)stream);
}

int32_t cs_shards_search_status(cs_shards* h, void* stream, uint32_t* overflowed) {
    if (!h || !overflowed) return fail(CS_ERR_BAD_ARG, "null argument");
    *overflowed = 0;
    {
        DeviceGuard g(h->root);
        CS_HIP(hipStreamSynchronize((hipStream_t)stream));  // the merges behind every shard search issued on it are done
    }
    std::vector<SearchCtx*> all;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        all = h->all;
    }
    for (SearchCtx* c : all)
        for (uint32_t s = 0; s < h->n; ++s) {
            uint32_t ov = 0;
            CS_TRY(cs_index_search_status(h->idx[s], c->sh[s].stream, &ov));
            *overflowed |= ov;
        }
    return CS_OK;
}

int32_t cs_shards_root_device(const cs_shards* h) { return h ? h->root : -1; }
cs_index* cs_shards_shard_index(cs_shards* h, uint32_t shard) { return h && shard < h->n ? h->idx[shard] : nullptr; }
int32_t cs_shards_shard_device(const cs_shards* h, uint32_t shard) { return h && shard < h->n ? h->devices[shard] : -1; }

// search::search's vector leg over the sharded store (cs_index_search_variants' counterpart): per-variant searches on
// every shard, shard merge, then the variant merge (dedup by id keeping the best key, top k, early-termination
// predicate) on the first device.  out_cos / out_ids: [k]; *out_count valid entries.
int32_t cs_shards_search_variants(cs_shards* h, const float* queries, uint32_t nq, uint32_t dim, uint32_t k, float* out_cos,
                                  uint32_t* out_ids, uint32_t* out_count, int32_t* out_high_confidence) {
    if (nq > CS_MAX_VARIANTS) return fail(CS_ERR_BAD_ARG, "at most %u query variants per call, got %u", CS_MAX_VARIANTS, nq);
    if (!out_count) return fail(CS_ERR_BAD_ARG, "null buffer");
    uint32_t meta[2] = {0, 0};
    CS_TRY(shards_search_impl(h, queries, nq, dim, k, out_cos, out_ids, out_count, meta));
    *out_count = meta[0];
    if (out_high_confidence) *out_high_confidence = (int32_t)meta[1];
    return CS_OK;
}

}  // extern "C"
