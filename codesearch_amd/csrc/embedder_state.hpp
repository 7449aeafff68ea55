// embedder_state.hpp — the state behind a cs_embedder handle and the functions its translation units share.  The embedder
// is split by concern: embedder.hip (create / destroy / the C entry points), embedder_forward.hip (one
// mini-batch through the encoder kernels), embedder_embed.hip (mini-batching from ids and from strings),
// embedder_queue.hip (the submission queue).  Everything here is library-internal.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "encoder.hpp"
#include "small_forward.hpp"
#include "small_path.hpp"
#include "split_f16.hpp"
#include "gemm_q8.hpp"

using namespace cs;

// ---- submission queue (cs_embedder_submit_* / cs_embedder_wait*) -------------------------------------------------
// One flush embeds everything queued; its rows stay in one device buffer until every ticket of the flush has been
// collected.
// Result buffers are recycled through a small grow-only pool owned by the embedder (hipFree waits for the whole device:
// freeing one per flush would stall every stream of the process between files).
struct QueuePool {
    std::mutex mu;
    std::vector<std::pair<float*, size_t>> free_bufs;  // (pointer, capacity in floats)
    std::vector<std::pair<float*, size_t>> free_host;  // pinned host mirrors of such buffers
    int device = 0;
    ~QueuePool() {
        cs::DeviceGuard g(device);
        for (auto& b : free_bufs) (void)hipFree(b.first);
        for (auto& b : free_host) (void)hipHostFree(b.first);
    }
};
// A flush's rows reach host callers through ONE device-to-host copy of the whole buffer into a pinned mirror, made by the
// first host wait; every wait is then a memcpy.  (A copy + stream synchronisation per ticket was 26 us apiece — 1.6 ms for
// the 64 small calls of a directory of small files, against 2.5 ms of device time.)
struct QueueFlush {
    std::shared_ptr<QueuePool> pool;
    float* d_rows = nullptr;
    size_t cap = 0;
    size_t used = 0;          // floats written by the flush
    std::mutex hmu;
    float* h_rows = nullptr;  // pinned, `h_cap` floats; valid once host_ready
    size_t h_cap = 0;
    bool host_ready = false;
    ~QueueFlush() {
        std::lock_guard<std::mutex> lk(pool->mu);
        if (d_rows) pool->free_bufs.emplace_back(d_rows, cap);
        if (h_rows) pool->free_host.emplace_back(h_rows, h_cap);
    }
};
struct QueueEntry {
    uint64_t ticket = 0;
    std::vector<std::vector<int32_t>> ids;   // per row: token ids up to its length
    std::vector<std::vector<int32_t>> mask;  // per row, only for submit_ids rows whose mask has holes; else empty
    enum { QUEUED, COMPUTING, DONE, FAILED } state = QUEUED;
    std::shared_ptr<QueueFlush> flush;       // DONE: rows [first_row, first_row + ids.size()) of flush->d_rows
    uint64_t first_row = 0;
    int32_t error = 0;
    std::string error_text;
};

struct cs_embedder {
    std::mutex qmu;                 // the queue below
    std::mutex cmu;                 // one flush at a time (and excludes nothing else: embed_* keep `&mut self` rules)
    std::map<uint64_t, std::shared_ptr<QueueEntry>> queue;  // by ticket = submission order
    std::shared_ptr<QueuePool> qpool;
    uint64_t next_ticket = 1;
    int device = 0;
    cs_bert_config cfg{};
    cs_bert_offsets off{};
    float* d_params = nullptr;
    float* d_wqkv = nullptr;  // [layers][3H][H]  (query | key | value rows)
    float* d_bqkv = nullptr;  // [layers][3H]
    // CS_ARCH_NOMIC: the up projection's bias as one [2I] vector per layer (fc11's and fc12's entries interleaved in groups
    // of 16, like the rows of the packed weight) and the rotary table [max_position][d_h / 2] (cos, sin)
    float* d_bup = nullptr;
    float2* d_rope = nullptr;
    // CS_ARCH_MODERN: the rotary table of the local-attention layers (d_rope: of the global ones) and a row of H zeros (the
    // token-type row the embedding kernel adds: this family has none)
    float2* d_rope_local = nullptr;
    float* d_zero_row = nullptr;
    // CS_ARCH_JINA*: the ALiBi head slopes, [2][heads]: as they are | times log2 e (attention_split.hip adds in the exp2 domain)
    float* d_alibi = nullptr;
    _Float16* d_wsplit = nullptr;  // per layer: wqkv | attention-out | ffn-up | ffn-down, split-f16 rows
    uint32_t* d_flag = nullptr;    // split-f16 range flag
    // the one-launch forward of short queries (small_forward.hip): the layers' pointers on the device, its barrier words,
    // whether this mini-batch ran it (embed_impl then reads the give-up word), how often it ran / gave up
    SfLayer* d_sf_layers = nullptr;
    uint32_t* d_sf_sync = nullptr;
    uint64_t* d_sf_dbg = nullptr;   // CS_SMALL_FORWARD_DEBUG: per-block tick sums of the last launch (printed to stderr)
    // small_path.hip / small_forward.hip workspace: [4][SP_MAX_ROWS][H] FFN-down K-slice slabs | [SP_MAX_ROWS][H] the residual
    // stream behind a layer's last LayerNorm (d_x holds it behind the attention block's)
    float* d_sp_ws = nullptr;
    bool sf_ran = false, sf_off = false;
    uint64_t sf_forwards = 0, sf_fallbacks = 0;
    // dynamically quantised models (gemm_q8.hip): s8 weights per layer (q8_layer), their column metadata, the running
    // range slot of every quantised tensor of a forward ([layers][4][q8_units][Q8_RANGE_WORDS]) and the rows' metadata
    bool quantized = false;
    int8_t* d_wq8 = nullptr;
    Q8ColMeta* d_cmeta = nullptr;
    int8_t* d_wq8_stages = nullptr;     // [layers][H * H + H * I]: out-proj and FFN-down once more in the order gemm_q8_ln_kernel streams them (384-wide models)
    uint32_t* d_cmeta_tiles = nullptr;  // the same columns as 2-KiB structure-of-arrays tiles (gemm_q8_slab.hip), 16 B per column
    uint32_t* d_range = nullptr;
    uint32_t q8_units = 1;
    Q8RowMeta* d_rmeta = nullptr;  // [cap_tokens] (workspace): rows of the tensor being multiplied
    Q8RowMeta* d_rmeta2 = nullptr; // [cap_tokens]: rows of the re-quantised FFN intermediate
    float* d_range_pairs = nullptr; // (lo, hi) per block / wave of the kernel that produced the tensor quantised next
    size_t cap_range_pairs = 0, cap_range_pairs2 = 0;
    // several quantisation units (calls of the reference) in one device batch: per sequence its unit, per unit its own
    // padded length, per row its range slot (gemm_q8.hpp); cur_units = units of the mini-batch being run (1: none of this)
    uint32_t* d_seq_unit = nullptr;
    uint32_t* d_unit_len = nullptr;
    uint32_t* d_row_slot = nullptr;
    uint32_t cur_units = 1;
    uint32_t q8_x_pairs = 0;        // (lo, hi) pairs the kernel that last wrote the residual stream left in d_range_pairs
    int gemm_mode = CS_GEMM_SPLIT_F16;
    bool split_unavailable = false;  // device flushes f16 subnormals in the MFMA: exact-f32 kernels only
    bool wide_ok = false;            // every |w| < 31.98: the one-accumulator 128 x 384 kernels may run (gemm_wide.hip)
    int streams_in_flight = 1;       // slices of the current mini-batch running side by side (forward())
    uint64_t split_forwards = 0, f32_forwards = 0, range_fallbacks = 0, q8_forwards = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;     // second half of a mini-batch runs here (see forward())
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t xstreams[2] = {nullptr, nullptr};  // CS_ENCODER_STREAMS=3|4: further slices of the mini-batch
    hipEvent_t xjoin[2] = {nullptr, nullptr};
    int n_streams = 2;
    bool streams_forced = false;       // CS_ENCODER_STREAMS given: forward() does not second-guess it
    size_t cap_tokens = 0, cap_seqs = 0;
    int32_t* d_ids = nullptr;
    int32_t* d_mask = nullptr;
    float* d_x = nullptr;       // [T, H]
    float* d_xs = nullptr;      // [T, H/32, 64] f16: x in split form (same bytes as f32)
    float* d_qkv = nullptr;     // [T, 3H]
    float* d_ctx = nullptr;     // [T, H]   (f32, or split form: same bytes)
    float* d_mid = nullptr;     // [T, I]   (f32, or split form: same bytes); CS_ARCH_NOMIC: [T, 3I] per slice (mid_width)
    float* d_pooled = nullptr;  // [B, H]
    float* pooled_dst = nullptr;  // set around a forward whose pooled rows go straight to the caller's device buffer (a corpus region: E8 in place)
    uint32_t* d_perm = nullptr; // [B] destination row of each pooled row (length-sorted text mini-batches)
    std::vector<float> h_pooled; // host staging of a mini-batch's rows when they are scattered
    char* h_pin = nullptr;       // 128 KiB of pinned host memory: a small mini-batch's ids | mask going in, its range flag + rows coming out
    uint32_t last_B = 0, last_L = 0;
    bool last_hidden_partial = false;  // the last forward ran the CLS tail: d_x holds the previous layer outside the CLS rows
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double forward_ms = 0.0;
    uint64_t forwards = 0;
    // cs_embedder_profile_stages: one HIP event after every kernel of a forward (single stream), durations
    // summed per kernel class
    bool stage_profile = false;
    std::vector<hipEvent_t> stage_ev;      // pool; stage_ev[0] precedes the first kernel
    std::vector<int> stage_tag;            // tag of the kernel that ends at stage_ev[i + 1]
    double stage_us[CS_ENCODER_STAGES] = {};
    uint64_t stage_forwards = 0;
};

namespace cs {
namespace emb {

// Floats per token row of the feed-forward workspace: [I]; gated feed-forwards: [2I] (value | gate) + [I] (their product)
size_t mid_width(const cs_bert_config& c);
void free_workspace(cs_embedder* h);
int32_t reserve(cs_embedder* h, size_t seqs, size_t tokens);
// Offsets (in f16 elements) of one layer's split weights inside d_wsplit.  Gated feed-forwards: `up` holds value | gate, [2I][H].
struct SplitLayer { size_t qkv, ao, up, down, total; };
SplitLayer split_layer(const cs_bert_config& c);
// One mini-batch already on the device (d_ids / d_mask) -> d_pooled [B, H]  (embedder_forward.hip)
int32_t forward(cs_embedder* h, uint32_t B, uint32_t L, int mode);
uint32_t default_batch(const cs_embedder* h);

// Several quantisation units in ONE mini-batch (dynamic-quantisation mode: calls of the reference embedded together, each
// still quantised as the tensor it would have been on its own): the unit of every sequence, each unit's own padded length.
struct UnitSpec {
    const uint32_t* seq_unit = nullptr;  // [n]
    const uint32_t* unit_len = nullptr;  // [units]
    uint32_t units = 1;
};
// (embedder_embed.hip)
int32_t embed_impl(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n, uint32_t seq_len, uint32_t batch,
                   float* out, bool out_on_device, const volatile int32_t* cancel, const uint32_t* perm = nullptr,
                   const UnitSpec* units = nullptr);
int32_t embed_texts_impl(cs_embedder* h, const cs_tokenizer* t, const char* utf8, const uint64_t* offsets,
                         uint64_t n, uint32_t batch, float* out, bool out_on_device, const volatile int32_t* cancel);
int32_t embed_ids_entry(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n, uint32_t seq_len,
                        uint32_t batch, float* out, bool out_on_device, const volatile int32_t* cancel);
struct SeqView { const int32_t* ids; const int32_t* mask; uint32_t len; };
int32_t run_window(cs_embedder* h, const std::vector<SeqView>& seqs, uint32_t batch, int32_t pad, float* out,
                   bool out_on_device, const volatile int32_t* cancel, std::vector<uint32_t>& order,
                   std::vector<int32_t>& ids, std::vector<int32_t>& mask);

}  // namespace emb
}  // namespace cs
