// embedders.hip — cs_embedders_*: one encoder replica per GPU inside ONE process, and the index loop of the
// reference (/root/reference/src/index/mod.rs:626-762: embed_chunks :692 -> insert_chunks_with_ids :723) over a
// row-sharded store.  SURVEY.md §8e: "each GPU embeds the chunks destined for its own shard and E8 writes rows in
// place, so indexing needs no collective" — replicas only, weights replicated, no exchange step.
//
// Built on the public C ABI alone (cs_embedder_*, cs_shards_*): a replica is a plain cs_embedder.
//   cs_embedders_embed_*  : the n inputs are cut into one contiguous range per replica, each embedded by its own
//                           host thread on its own device, rows written straight into the caller's buffer: row i is
//                           input i (embed_batch's contract, embedder.rs:249-263).
//   cs_embedders_index_*  : ids stay contiguous from next_id (store.rs:659-685), so the shard every input will live
//                           on is known before anything is embedded (cs_shards_plan_append).  Each replica embeds
//                           the inputs destined for the shards it serves — the shards on its own device, shared out
//                           among that device's replicas; a shard whose device has no replica goes to replica
//                           shard % count and its rows cross xGMI once — into a buffer in its own HBM; when every
//                           replica has finished, the runs are appended in id order (cs_shards_add_device_parts:
//                           capacity reserved on every shard first, then one asynchronous copy per run).  A failure
//                           or a shutdown request before that point leaves the store untouched.
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

using namespace cs;

struct cs_embedders {
    std::vector<cs_embedder*> rep;
    std::vector<int> devices;
    uint32_t dim = 0;
};

namespace {

struct ThreadResult { int32_t status = CS_OK; std::string text; };

// Runs job(r) for every replica with work on a thread of its own; the first failure (lowest replica) is returned
// with its message re-installed as the calling thread's cs_last_error().
template <class Job>
int32_t run_replicas(uint32_t n, const std::vector<bool>& has_work, Job job) {
    std::vector<ThreadResult> res(n);
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < n; ++r) {
        if (!has_work[r]) continue;
        th.emplace_back([&, r] {
            res[r].status = job(r);
            if (res[r].status != CS_OK) res[r].text = last_error_ref();
        });
    }
    for (auto& t : th) t.join();
    for (uint32_t r = 0; r < n; ++r)
        if (res[r].status != CS_OK) return fail(res[r].status, "%s", res[r].text.c_str());
    return CS_OK;
}

// [lo, hi) of input i's range for replica r of R: contiguous, balanced, whole mini-batches where possible
void replica_range(uint64_t n, uint32_t R, uint32_t batch, uint32_t r, uint64_t* lo, uint64_t* hi) {
    const uint64_t nb = (n + batch - 1) / batch;  // mini-batches in all
    const uint64_t b0 = nb * r / R, b1 = nb * (r + 1) / R;
    *lo = std::min<uint64_t>(n, b0 * batch);
    *hi = std::min<uint64_t>(n, b1 * batch);
}

uint32_t policy_batch(const cs_embedders* e, uint32_t batch) {
    if (batch) return batch;
    return e->dim <= 384 ? 256 : (e->dim <= 768 ? 128 : 64);  // embedder.rs:251-261 (CODESEARCH_BATCH_SIZE is read by the replica)
}

struct Run { uint32_t shard; uint64_t first, count; };  // inputs [first, first + count) of the call live on `shard`

int32_t plan_runs(cs_shards* store, uint64_t n, std::vector<Run>* runs) {
    uint32_t cap = 0;
    CS_TRY(cs_shards_plan_append(store, n, 0, nullptr, nullptr, nullptr, &cap));
    std::vector<uint32_t> sh(cap);
    std::vector<uint64_t> first(cap), count(cap);
    uint32_t got = 0;
    CS_TRY(cs_shards_plan_append(store, n, cap, sh.data(), first.data(), count.data(), &got));
    runs->clear();
    for (uint32_t i = 0; i < got; ++i) runs->push_back(Run{sh[i], first[i], count[i]});
    return CS_OK;
}

// which replica embeds the rows of shard s
uint32_t replica_of_shard(const cs_embedders* e, cs_shards* store, uint32_t s) {
    const int dev = cs_shards_shard_device(store, s);
    std::vector<uint32_t> local;
    for (uint32_t r = 0; r < e->rep.size(); ++r)
        if (e->devices[r] == dev) local.push_back(r);
    if (!local.empty()) return local[s % local.size()];
    return s % (uint32_t)e->rep.size();
}

// The index loop.  gather(r, input index list) -> embeds those inputs on replica r into d_out (device buffer).
template <class Embed>
int32_t index_impl(cs_embedders* e, cs_shards* store, uint64_t n, uint32_t* out_ids, Embed embed) {
    if (!e || !store) return fail(CS_ERR_BAD_ARG, "null handle");
    if (cs_shards_dim(store) != e->dim)  // store.rs:667-671
        return fail(CS_ERR_DIM_MISMATCH, "Embedding dimension mismatch: expected %u, got %u", cs_shards_dim(store), e->dim);
    if (n == 0) return CS_OK;
    const uint32_t R = (uint32_t)e->rep.size();
    std::vector<Run> runs;
    CS_TRY(plan_runs(store, n, &runs));
    // per replica: the inputs it embeds, run after run (so a run is a contiguous slice of the replica's buffer)
    std::vector<std::vector<uint64_t>> inputs(R);
    std::vector<uint32_t> run_rep(runs.size());
    std::vector<uint64_t> run_off(runs.size());
    for (size_t i = 0; i < runs.size(); ++i) {
        const uint32_t r = replica_of_shard(e, store, runs[i].shard);
        run_rep[i] = r;
        run_off[i] = inputs[r].size();
        for (uint64_t j = 0; j < runs[i].count; ++j) inputs[r].push_back(runs[i].first + j);
    }
    std::vector<float*> d_buf(R, nullptr);
    std::vector<bool> has_work(R);
    for (uint32_t r = 0; r < R; ++r) has_work[r] = !inputs[r].empty();
    auto release = [&] {
        for (uint32_t r = 0; r < R; ++r)
            if (d_buf[r]) { DeviceGuard g(e->devices[r]); (void)hipFree(d_buf[r]); }
    };
    int32_t st = run_replicas(R, has_work, [&](uint32_t r) -> int32_t {
        DeviceGuard g(e->devices[r]);
        CS_HIP(hipMalloc(&d_buf[r], inputs[r].size() * e->dim * sizeof(float)));
        return embed(r, inputs[r], d_buf[r]);
    });
    if (st == CS_OK) {
        std::vector<const float*> ptr(runs.size());
        std::vector<int32_t> dev(runs.size());
        std::vector<uint64_t> cnt(runs.size());
        for (size_t i = 0; i < runs.size(); ++i) {
            ptr[i] = d_buf[run_rep[i]] + run_off[i] * e->dim;
            dev[i] = e->devices[run_rep[i]];
            cnt[i] = runs[i].count;
        }
        st = cs_shards_add_device_parts(store, (uint32_t)runs.size(), ptr.data(), dev.data(), cnt.data(), e->dim, out_ids);
        // the copies run on each source device's null stream: wait for them before the buffers go
        for (uint32_t r = 0; r < R; ++r)
            if (d_buf[r]) { DeviceGuard g(e->devices[r]); (void)hipDeviceSynchronize(); }
    }
    release();
    return st;
}

}  // namespace

extern "C" {

int32_t cs_embedders_create(const cs_bert_config* cfg, const float* params, uint64_t seed, const int32_t* devices,
                            uint32_t n, cs_embedders** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!cfg || !devices || n == 0 || n > 64) return fail(CS_ERR_BAD_ARG, "need a config and 1..64 devices");
    cs_embedders* e = new cs_embedders();
    e->dim = cfg->hidden;
    for (uint32_t i = 0; i < n; ++i) {
        cs_embedder* h = nullptr;
        const int32_t st = cs_embedder_create(cfg, params, seed, devices[i], &h);
        if (st != CS_OK) { cs_embedders_destroy(e); return st; }
        e->rep.push_back(h);
        e->devices.push_back(devices[i]);
    }
    *out = e;
    return CS_OK;
}

int32_t cs_embedders_create_from_dir(const char* model_dir, int32_t pooling, const int32_t* devices, uint32_t n,
                                     cs_embedders** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!model_dir || !devices || n == 0 || n > 64) return fail(CS_ERR_BAD_ARG, "need a model directory and 1..64 devices");
    cs_embedders* e = new cs_embedders();
    for (uint32_t i = 0; i < n; ++i) {
        cs_embedder* h = nullptr;
        const int32_t st = cs_embedder_create_from_dir(model_dir, pooling, devices[i], &h);
        if (st != CS_OK) { cs_embedders_destroy(e); return st; }
        e->rep.push_back(h);
        e->devices.push_back(devices[i]);
        e->dim = cs_embedder_dim(h);
    }
    *out = e;
    return CS_OK;
}

void cs_embedders_destroy(cs_embedders* e) {
    if (!e) return;
    for (cs_embedder* h : e->rep) cs_embedder_destroy(h);
    delete e;
}

uint32_t cs_embedders_count(const cs_embedders* e) { return e ? (uint32_t)e->rep.size() : 0; }
uint32_t cs_embedders_dim(const cs_embedders* e) { return e ? e->dim : 0; }
cs_embedder* cs_embedders_replica(cs_embedders* e, uint32_t i) { return e && i < e->rep.size() ? e->rep[i] : nullptr; }
int32_t cs_embedders_device(const cs_embedders* e, uint32_t i) { return e && i < e->rep.size() ? e->devices[i] : -1; }

int32_t cs_embedders_embed_texts(cs_embedders* e, const cs_tokenizer* t, const char* utf8, const uint64_t* offsets,
                                 uint64_t n, uint32_t batch, float* out, const volatile int32_t* cancel) {
    if (!e) return fail(CS_ERR_BAD_ARG, "null embedders handle");
    if (n == 0) return CS_OK;
    if (!utf8 || !offsets || !out) return fail(CS_ERR_BAD_ARG, "null buffer");
    const uint32_t R = (uint32_t)e->rep.size(), b = policy_batch(e, batch);
    std::vector<bool> has_work(R);
    for (uint32_t r = 0; r < R; ++r) { uint64_t lo, hi; replica_range(n, R, b, r, &lo, &hi); has_work[r] = hi > lo; }
    return run_replicas(R, has_work, [&](uint32_t r) -> int32_t {
        uint64_t lo, hi;
        replica_range(n, R, b, r, &lo, &hi);
        // offsets index into the one blob: a range is the same blob with the offsets advanced
        return cs_embedder_embed_texts(e->rep[r], t, utf8, offsets + lo, hi - lo, batch, out + lo * e->dim, cancel);
    });
}

int32_t cs_embedders_embed_ids(cs_embedders* e, const int32_t* ids, const int32_t* mask, uint64_t n, uint32_t seq_len,
                               uint32_t batch, float* out, const volatile int32_t* cancel) {
    if (!e) return fail(CS_ERR_BAD_ARG, "null embedders handle");
    if (n == 0) return CS_OK;
    if (!ids || !mask || !out) return fail(CS_ERR_BAD_ARG, "null buffer");
    const uint32_t R = (uint32_t)e->rep.size(), b = policy_batch(e, batch);
    std::vector<bool> has_work(R);
    for (uint32_t r = 0; r < R; ++r) { uint64_t lo, hi; replica_range(n, R, b, r, &lo, &hi); has_work[r] = hi > lo; }
    return run_replicas(R, has_work, [&](uint32_t r) -> int32_t {
        uint64_t lo, hi;
        replica_range(n, R, b, r, &lo, &hi);
        return cs_embedder_embed_ids(e->rep[r], ids + lo * seq_len, mask + lo * seq_len, hi - lo, seq_len, batch,
                                     out + lo * e->dim, cancel);
    });
}

int32_t cs_embedders_index_ids(cs_embedders* e, cs_shards* store, const int32_t* ids, const int32_t* mask, uint64_t n,
                               uint32_t seq_len, uint32_t batch, uint32_t* out_ids, const volatile int32_t* cancel) {
    if (n && (!ids || !mask)) return fail(CS_ERR_BAD_ARG, "null buffer");
    return index_impl(e, store, n, out_ids, [&](uint32_t r, const std::vector<uint64_t>& in, float* d_out) -> int32_t {
        std::vector<int32_t> bi(in.size() * seq_len), bm(in.size() * seq_len);
        for (size_t j = 0; j < in.size(); ++j) {
            std::memcpy(bi.data() + j * seq_len, ids + in[j] * seq_len, seq_len * sizeof(int32_t));
            std::memcpy(bm.data() + j * seq_len, mask + in[j] * seq_len, seq_len * sizeof(int32_t));
        }
        return cs_embedder_embed_ids_device(e->rep[r], bi.data(), bm.data(), in.size(), seq_len, batch, d_out, cancel);
    });
}

int32_t cs_embedders_index_texts(cs_embedders* e, const cs_tokenizer* t, cs_shards* store, const char* utf8,
                                 const uint64_t* offsets, uint64_t n, uint32_t batch, uint32_t* out_ids,
                                 const volatile int32_t* cancel) {
    if (n && (!utf8 || !offsets)) return fail(CS_ERR_BAD_ARG, "null buffer");
    for (uint64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(CS_ERR_BAD_ARG, "text offsets must be non-decreasing");
    return index_impl(e, store, n, out_ids, [&](uint32_t r, const std::vector<uint64_t>& in, float* d_out) -> int32_t {
        // the replica's texts, re-packed as one blob of their own
        std::vector<uint64_t> off(in.size() + 1, 0);
        for (size_t j = 0; j < in.size(); ++j) off[j + 1] = off[j] + (offsets[in[j] + 1] - offsets[in[j]]);
        std::string blob(off.back(), '\0');
        for (size_t j = 0; j < in.size(); ++j)
            std::memcpy(&blob[off[j]], utf8 + offsets[in[j]], offsets[in[j] + 1] - offsets[in[j]]);
        return cs_embedder_embed_texts_device(e->rep[r], t, blob.data(), off.data(), in.size(), batch, d_out, cancel);
    });
}

}  // extern "C"
