// embedder_queue.hip — the submission queue (cs_embedder_submit_* / cs_embedder_wait*): the reference's 32-chunk calls coalesced into full device batches.
// (one of the translation units behind cs_embedder_*: see embedder_state.hpp)
#include "embedder_state.hpp"

using namespace cs;

namespace cs {
namespace emb {

// ---- submission queue ---------------------------------------------------------------------------------------------
// The reference feeds its embedder 32 chunks per call, one file at a time, under a mutex
// (/root/reference/src/embed/batch.rs:70,84-115; src/embed/mod.rs:41): at that shape a device batch is an eighth of
// what fills the chip.  submit() only queues token rows; the first wait() that needs an unfinished ticket embeds
// EVERYTHING queued so far as length-grouped mini-batches of the embed_batch size (256 for 384-d models), so eight
// slices of 32 run as one 256-row forward; rows come back per ticket, in submission order.

int32_t queue_push(cs_embedder* h, std::shared_ptr<QueueEntry> e, uint64_t* ticket) {
    std::lock_guard<std::mutex> lk(h->qmu);
    e->ticket = h->next_ticket++;
    h->queue[e->ticket] = e;
    *ticket = e->ticket;
    return CS_OK;
}

// Embeds every QUEUED entry.  Caller holds h->cmu.
int32_t flush_queue(cs_embedder* h, const volatile int32_t* cancel) {
    std::vector<std::shared_ptr<QueueEntry>> todo;
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        for (auto& kv : h->queue)
            if (kv.second->state == QueueEntry::QUEUED) { kv.second->state = QueueEntry::COMPUTING; todo.push_back(kv.second); }
    }
    if (todo.empty()) return CS_OK;
    const uint32_t H = h->cfg.hidden, batch = default_batch(h);
    std::vector<SeqView> seqs;
    for (auto& e : todo)
        for (size_t r = 0; r < e->ids.size(); ++r)
            seqs.push_back(SeqView{e->ids[r].data(), e->mask.empty() || e->mask[r].empty() ? nullptr : e->mask[r].data(),
                                   (uint32_t)e->ids[r].size()});
    auto fl = std::make_shared<QueueFlush>();
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        if (!h->qpool) { h->qpool = std::make_shared<QueuePool>(); h->qpool->device = h->device; }
        fl->pool = h->qpool;
    }
    const int32_t st = [&]() -> int32_t {
        DeviceGuard g(h->device);
        const size_t need = seqs.size() * H;
        fl->used = need;
        {   // smallest pooled buffer that fits, else a new one
            std::lock_guard<std::mutex> lk(fl->pool->mu);
            auto& fb = fl->pool->free_bufs;
            size_t best = fb.size();
            for (size_t i = 0; i < fb.size(); ++i)
                if (fb[i].second >= need && (best == fb.size() || fb[i].second < fb[best].second)) best = i;
            if (best < fb.size()) { fl->d_rows = fb[best].first; fl->cap = fb[best].second; fb.erase(fb.begin() + best); }
        }
        if (!fl->d_rows) {
            const size_t cap = std::max<size_t>(need, (size_t)default_batch(h) * H);
            CS_HIP(hipMalloc(&fl->d_rows, cap * sizeof(float)));
            fl->cap = cap;
        }
        const size_t window = (size_t)batch * 16;
        std::vector<uint32_t> order;
        std::vector<int32_t> ids, mask;
        if (h->gemm_mode == CS_GEMM_Q8_DYNAMIC) {
            // A quantised model's activations are quantised per CALL tensor (embedder.rs:286-289 hands ORT one submission
            // at a time, fastembed cuts it into `batch` consecutive rows padded to their longest): those tensors stay the
            // quantisation UNITS, but several of them share a device batch — each row carries its unit's range slot, and a
            // unit's rows beyond its own padded length are kept out of its range (UnitSpec, gemm_q8.hpp).  A unit is
            // never split over two device batches.
            struct Unit { size_t first, rows; uint32_t len; };
            std::vector<Unit> us;
            size_t lo = 0;
            for (auto& e : todo) {
                const size_t n = e->ids.size();
                for (size_t b0 = 0; b0 < n; b0 += batch) {
                    Unit u{lo + b0, std::min<size_t>(batch, n - b0), 1};
                    for (size_t r = 0; r < u.rows; ++r) u.len = std::max(u.len, seqs[u.first + r].len);
                    us.push_back(u);
                }
                lo += n;
            }
            const uint64_t budget = (uint64_t)batch * std::min<uint32_t>(256, h->cfg.max_position);  // as run_window's
            std::vector<uint32_t> seq_unit, unit_len;
            for (size_t u0 = 0; u0 < us.size();) {
                if (cancel && *cancel) return fail(CS_ERR_CANCELLED, "Embedding interrupted by shutdown request");
                size_t u1 = u0 + 1, rows = us[u0].rows;
                uint32_t L = us[u0].len;
                while (u1 < us.size() && us[u1].first == us[u1 - 1].first + us[u1 - 1].rows && rows + us[u1].rows <= batch &&
                       (uint64_t)(rows + us[u1].rows) * std::max(L, us[u1].len) <= budget) {
                    rows += us[u1].rows;
                    L = std::max(L, us[u1].len);
                    ++u1;
                }
                ids.assign(rows * L, 0);
                mask.assign(rows * L, 0);
                seq_unit.resize(rows);
                unit_len.resize(u1 - u0);
                size_t r = 0;
                for (size_t u = u0; u < u1; ++u) {
                    unit_len[u - u0] = us[u].len;
                    for (size_t i = 0; i < us[u].rows; ++i, ++r) {
                        const SeqView& v = seqs[us[u].first + i];
                        std::copy(v.ids, v.ids + v.len, ids.begin() + r * L);
                        if (v.mask) std::copy(v.mask, v.mask + v.len, mask.begin() + r * L);
                        else std::fill(mask.begin() + r * L, mask.begin() + r * L + v.len, 1);
                        seq_unit[r] = (uint32_t)(u - u0);
                    }
                }
                {
                    DeviceGuard g2(h->device);
                    CS_TRY(reserve(h, std::max<size_t>(rows, h->cap_seqs), std::max<size_t>(rows * L, h->cap_tokens)));
                }
                UnitSpec spec{seq_unit.data(), unit_len.data(), (uint32_t)(u1 - u0)};
                CS_TRY(embed_impl(h, ids.data(), mask.data(), rows, L, (uint32_t)rows, fl->d_rows + us[u0].first * H, true, nullptr,
                                  nullptr, &spec));
                u0 = u1;
            }
            return CS_OK;
        }
        for (size_t lo = 0; lo < seqs.size(); lo += window) {
            const std::vector<SeqView> win(seqs.begin() + lo, seqs.begin() + std::min(seqs.size(), lo + window));
            CS_TRY(run_window(h, win, batch, 0, fl->d_rows + lo * H, true, cancel, order, ids, mask));
        }
        return CS_OK;
    }();
    std::lock_guard<std::mutex> lk(h->qmu);
    uint64_t row = 0;
    for (auto& e : todo) {
        if (st == CS_OK) { e->state = QueueEntry::DONE; e->flush = fl; e->first_row = row; }
        else if (st == CS_ERR_CANCELLED) e->state = QueueEntry::QUEUED;  // embedder.rs:280-282: nothing is lost, a later wait retries
        else { e->state = QueueEntry::FAILED; e->error = st; e->error_text = last_error_ref(); }
        row += e->ids.size();
    }
    return st;
}

int32_t queue_wait(cs_embedder* h, uint64_t ticket, float* out, bool out_on_device, const volatile int32_t* cancel) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (!out) return fail(CS_ERR_BAD_ARG, "null buffer");
    std::shared_ptr<QueueEntry> e;
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        auto it = h->queue.find(ticket);
        if (it == h->queue.end()) return fail(CS_ERR_BAD_ARG, "unknown or already collected ticket %llu", (unsigned long long)ticket);
        e = it->second;
    }
    int32_t flush_st = CS_OK;
    {
        std::lock_guard<std::mutex> lk(h->cmu);  // waits for a flush another caller is running (it may cover this ticket)
        bool queued;
        {
            std::lock_guard<std::mutex> q(h->qmu);
            queued = e->state == QueueEntry::QUEUED;
        }
        if (queued) flush_st = flush_queue(h, cancel);
    }
    {
        // the entry leaves the queue under the lock; the copy below runs WITHOUT it (a blocking copy under qmu stalled
        // every submit / wait of other threads for its duration: ADVICE r3)
        std::lock_guard<std::mutex> lk(h->qmu);
        if (e->state == QueueEntry::QUEUED) return flush_st != CS_OK ? flush_st : fail(CS_ERR_HIP, "ticket was not embedded");
        h->queue.erase(ticket);
        if (e->state == QueueEntry::FAILED) return fail(e->error, "%s", e->error_text.c_str());
    }
    const size_t n = e->ids.size(), H = h->cfg.hidden;
    if (n == 0) return CS_OK;
    DeviceGuard g(h->device);
    QueueFlush& fl = *e->flush;
    if (!out_on_device) {
        std::lock_guard<std::mutex> lk(fl.hmu);
        if (!fl.host_ready) {  // the first host wait of this flush: the whole buffer, once
            const size_t need = fl.used;
            {
                std::lock_guard<std::mutex> pk(fl.pool->mu);
                auto& fh = fl.pool->free_host;
                size_t best = fh.size();
                for (size_t i = 0; i < fh.size(); ++i)
                    if (fh[i].second >= need && (best == fh.size() || fh[i].second < fh[best].second)) best = i;
                if (best < fh.size()) { fl.h_rows = fh[best].first; fl.h_cap = fh[best].second; fh.erase(fh.begin() + best); }
            }
            if (!fl.h_rows) {
                const size_t cap = std::max<size_t>(need, (size_t)default_batch(h) * H);
                CS_HIP(hipHostMalloc(reinterpret_cast<void**>(&fl.h_rows), cap * sizeof(float), hipHostMallocDefault));
                fl.h_cap = cap;
            }
            CS_HIP(hipMemcpyAsync(fl.h_rows, fl.d_rows, need * sizeof(float), hipMemcpyDeviceToHost, h->stream));
            CS_HIP(hipStreamSynchronize(h->stream));
            fl.host_ready = true;
        }
        std::memcpy(out, fl.h_rows + e->first_row * H, n * H * sizeof(float));
        return CS_OK;
    }
    // On the embedder's own stream, and waited for: the flush buffer goes back to the pool when `e` drops its reference at
    // return, and the next flush writes it on this (non-blocking) stream — a null-stream device-to-device copy is neither
    // ordered against that stream nor waited for by the host.
    CS_HIP(hipMemcpyAsync(out, fl.d_rows + e->first_row * H, n * H * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    CS_HIP(hipStreamSynchronize(h->stream));
    return CS_OK;
}
}  // namespace emb
}  // namespace cs

using namespace cs::emb;

extern "C" {

int32_t cs_embedder_submit_texts(cs_embedder* h, const cs_tokenizer* t, const char* utf8, const uint64_t* offsets,
                                 uint64_t n, uint64_t* ticket) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (!ticket) return fail(CS_ERR_BAD_ARG, "ticket is null");
    if (!t) return fail(CS_ERR_BAD_ARG, "Failed to generate embeddings: no tokenizer attached");
    if (n && (!utf8 || !offsets)) return fail(CS_ERR_BAD_ARG, "null buffer");
    if (n > 0xffffffffull) return fail(CS_ERR_BAD_ARG, "too many texts in one submission");
    for (uint64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(CS_ERR_BAD_ARG, "text offsets must be non-decreasing");
    auto e = std::make_shared<QueueEntry>();
    if (n) cs::tokenize_texts(t, utf8, offsets, (uint32_t)n, h->cfg.max_position, e->ids);  // on the caller's thread
    for (const auto& row : e->ids)
        for (int32_t id : row)
            if (id < 0 || (uint32_t)id >= h->cfg.vocab_size)
                return fail(CS_ERR_BAD_ARG, "Failed to generate embeddings: token id %d outside vocabulary of %u", id,
                            h->cfg.vocab_size);
    return queue_push(h, e, ticket);
}

int32_t cs_embedder_submit_ids(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n, uint32_t seq_len,
                               uint64_t* ticket) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (!ticket) return fail(CS_ERR_BAD_ARG, "ticket is null");
    if (n && (!ids || !mask)) return fail(CS_ERR_BAD_ARG, "null buffer");
    if (n && (seq_len == 0 || seq_len > h->cfg.max_position))
        return fail(CS_ERR_BAD_ARG, "seq_len %u outside 1..%u (max_position_embeddings)", seq_len, h->cfg.max_position);
    auto e = std::make_shared<QueueEntry>();
    e->ids.resize(n);
    e->mask.resize(n);
    for (uint64_t r = 0; r < n; ++r) {
        const int32_t* m = mask + r * seq_len;
        const int32_t* v = ids + r * seq_len;
        uint32_t len = seq_len;
        while (len > 1 && m[len - 1] == 0) --len;  // a row's length = the position after its last mask bit
        bool prefix = true;
        for (uint32_t i = 0; i < len; ++i) {
            if (v[i] < 0 || (uint32_t)v[i] >= h->cfg.vocab_size)
                return fail(CS_ERR_BAD_ARG, "Failed to generate embeddings: token id %d outside vocabulary of %u", v[i],
                            h->cfg.vocab_size);
            prefix = prefix && m[i] != 0;
        }
        e->ids[r].assign(v, v + len);
        if (!prefix) e->mask[r].assign(m, m + len);
    }
    return queue_push(h, e, ticket);
}

int32_t cs_embedder_wait(cs_embedder* h, uint64_t ticket, float* out, const volatile int32_t* cancel) {
    return queue_wait(h, ticket, out, false, cancel);
}

int32_t cs_embedder_wait_device(cs_embedder* h, uint64_t ticket, float* d_out, const volatile int32_t* cancel) {
    return queue_wait(h, ticket, d_out, true, cancel);
}

int32_t cs_embedder_discard(cs_embedder* h, uint64_t ticket) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    std::lock_guard<std::mutex> c(h->cmu);  // not while a flush holds pointers into the entry
    std::lock_guard<std::mutex> lk(h->qmu);
    if (!h->queue.erase(ticket)) return fail(CS_ERR_BAD_ARG, "unknown or already collected ticket %llu", (unsigned long long)ticket);
    return CS_OK;
}

uint64_t cs_embedder_queued_rows(cs_embedder* h) {
    if (!h) return 0;
    std::lock_guard<std::mutex> lk(h->qmu);
    uint64_t n = 0;
    for (auto& kv : h->queue)
        if (kv.second->state == QueueEntry::QUEUED) n += kv.second->ids.size();
    return n;
}

}  // extern "C"
