// embedder.hip — cs_embedder_*: the device half of the reference's FastEmbedder
// (/root/reference/src/embed/embedder.rs:201-322).  Starts from token ids (tokenisation is a
// host concern); runs the encoder kernels of encoder.hip; returns pooled, L2-normalised
// embeddings.  Mini-batching and the shutdown poll follow embed_batch_chunked
// (embedder.rs:266-295).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "encoder.hpp"
#include "scan.hpp"  // launch_synth_fill (cs_debug_gemm_time)
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
    uint32_t* d_perm = nullptr; // [B] destination row of each pooled row (length-sorted text mini-batches)
    std::vector<float> h_pooled; // host staging of a mini-batch's rows when they are scattered
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

namespace {

// dst[perm[r]] = src[r] for r < rows: one float4 per thread (H % 4 == 0)
__global__ void __launch_bounds__(256)
scatter_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ perm, float* __restrict__ dst,
                    uint32_t rows, uint32_t h4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * h4) return;
    const uint32_t r = i / h4, c = i % h4;
    reinterpret_cast<float4*>(dst)[(size_t)perm[r] * h4 + c] = reinterpret_cast<const float4*>(src)[(size_t)r * h4 + c];
}

// Floats per token row of the feed-forward workspace: [I]; CS_ARCH_NOMIC: [2I] (value | gate) + [I] (their product)
size_t mid_width(const cs_bert_config& c) { return (size_t)c.intermediate * (cs_arch_gated(c.arch) ? 3 : 1); }

void free_workspace(cs_embedder* h) {
    if (h->d_ids) (void)hipFree(h->d_ids);
    if (h->d_mask) (void)hipFree(h->d_mask);
    if (h->d_x) (void)hipFree(h->d_x);
    if (h->d_xs) (void)hipFree(h->d_xs);
    if (h->d_qkv) (void)hipFree(h->d_qkv);
    if (h->d_ctx) (void)hipFree(h->d_ctx);
    if (h->d_mid) (void)hipFree(h->d_mid);
    if (h->d_pooled) (void)hipFree(h->d_pooled);
    if (h->d_perm) (void)hipFree(h->d_perm);
    if (h->d_rmeta) (void)hipFree(h->d_rmeta);
    if (h->d_rmeta2) (void)hipFree(h->d_rmeta2);
    if (h->d_range_pairs) (void)hipFree(h->d_range_pairs);
    if (h->d_seq_unit) (void)hipFree(h->d_seq_unit);
    if (h->d_unit_len) (void)hipFree(h->d_unit_len);
    if (h->d_row_slot) (void)hipFree(h->d_row_slot);
    if (h->d_range) (void)hipFree(h->d_range);
    h->d_rmeta = h->d_rmeta2 = nullptr;
    h->d_range_pairs = nullptr;
    h->d_seq_unit = h->d_unit_len = h->d_row_slot = h->d_range = nullptr;
    h->d_perm = nullptr;
    h->d_ids = h->d_mask = nullptr;
    h->d_x = h->d_xs = h->d_qkv = h->d_ctx = h->d_mid = h->d_pooled = nullptr;
    h->cap_tokens = h->cap_seqs = 0;
}

int32_t reserve(cs_embedder* h, size_t seqs, size_t tokens) {
    if (tokens <= h->cap_tokens && seqs <= h->cap_seqs) return CS_OK;
    free_workspace(h);
    const size_t H = h->cfg.hidden, I = h->cfg.intermediate;
    CS_HIP(hipMalloc(&h->d_ids, tokens * sizeof(int32_t)));
    CS_HIP(hipMalloc(&h->d_mask, tokens * sizeof(int32_t)));
    CS_HIP(hipMalloc(&h->d_x, tokens * H * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_xs, tokens * H * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_qkv, tokens * 3 * H * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_ctx, tokens * H * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_mid, tokens * mid_width(h->cfg) * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_pooled, seqs * H * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_perm, seqs * sizeof(uint32_t)));
    if (h->quantized) {
        CS_HIP(hipMalloc(&h->d_rmeta, tokens * sizeof(Q8RowMeta)));
        CS_HIP(hipMalloc(&h->d_rmeta2, tokens * sizeof(Q8RowMeta)));
        // LayerNorm: a pair per four rows (per row with several units in the batch); attention: four per (head group,
        // sequence, 128 queries)
        h->cap_range_pairs = std::max<size_t>(tokens + 1, (size_t)h->cfg.heads * 4 * (tokens / 128 + seqs));
        // (+ a second set for the few-rows path: FFN-up leaves a pair per 16 x 16 output tile while it reads the first set)
        h->cap_range_pairs2 = (size_t)(I / 16) * (tokens / 16 + 1);
        CS_HIP(hipMalloc(&h->d_range_pairs, (h->cap_range_pairs + h->cap_range_pairs2) * 2 * sizeof(float)));
        CS_HIP(hipMalloc(&h->d_seq_unit, seqs * sizeof(uint32_t)));
        CS_HIP(hipMalloc(&h->d_unit_len, seqs * sizeof(uint32_t)));
        CS_HIP(hipMalloc(&h->d_row_slot, tokens * sizeof(uint32_t)));
        // a range slot per (layer, quantised tensor, unit): at most one unit per sequence
        h->q8_units = (uint32_t)seqs;
        CS_HIP(hipMalloc(&h->d_range, (size_t)h->cfg.layers * 4 * Q8_RANGE_WORDS * h->q8_units * sizeof(uint32_t)));
    }
    h->cap_tokens = tokens;
    h->cap_seqs = seqs;
    return CS_OK;
}

// Offsets (in f16 elements) of one layer's split weights inside d_wsplit.  CS_ARCH_NOMIC: `up` holds fc11 | fc12, [2I][H].
struct SplitLayer { size_t qkv, ao, up, down, total; };
SplitLayer split_layer(const cs_bert_config& c) {
    const size_t H = c.hidden, I = c.intermediate;
    SplitLayer o;
    o.qkv = 0;
    o.ao = o.qkv + 3 * H * H * 2;
    o.up = o.ao + H * H * 2;
    o.down = o.up + (cs_arch_gated(c.arch) ? 2 : 1) * I * H * 2;
    o.total = o.down + H * I * 2;
    return o;
}

// Sequences [b0, b0 + nb) of the mini-batch on stream s.  Every kernel but attention is local to
// a token row and attention is local to a sequence, so a range of sequences is an independent job
// on the same buffers at a token offset.
int32_t forward_range(cs_embedder* h, hipStream_t s, uint32_t b0, uint32_t nb, uint32_t L, int mode) {
    const cs_bert_config& c = h->cfg;
    const uint32_t H = c.hidden, I = c.intermediate, T = nb * L;
    const size_t t0 = (size_t)b0 * L;
    const float* P = h->d_params;
    const bool q8 = mode == CS_GEMM_Q8_DYNAMIC;
    const bool split = mode == CS_GEMM_SPLIT_F16 || q8;  // q8: attention and the buffers as in split mode
    float* x = h->d_x + t0 * H;
    float* qkv = h->d_qkv + t0 * 3 * H;
    float* ctx = h->d_ctx + t0 * H;
    // `nomic`: every family with a gated feed-forward and no position table (NomicBert, JinaBert); `rotary` / `jina` what
    // only one of them does (rotary map on Q / K | ALiBi on the scores, GELU gate, optional LayerNorm on Q / K rows)
    const bool nomic = cs_arch_gated(c.arch), rotary = c.arch == CS_ARCH_NOMIC, jina = cs_arch_alibi(c.arch);
    const bool qknorm = c.arch == CS_ARCH_JINA_QKNORM;
    const float* alibi = jina ? h->d_alibi : nullptr;
    float* mid = h->d_mid + t0 * mid_width(c);
    const int32_t* mask = h->d_mask + t0;
    EncoderLaunch a;
    a.ids = h->d_ids + t0; a.mask = mask;
    a.word = P + h->off.word; a.pos = nomic ? nullptr : P + h->off.pos; a.type0 = P + h->off.type;
    a.g = P + h->off.emb_ln_g; a.b = P + h->off.emb_ln_b;
    a.eps = c.layer_norm_eps; a.T = T; a.L = L; a.B = nb; a.vocab = c.vocab_size;
    a.pooling = c.pooling; a.x = x; a.out = h->d_pooled + (size_t)b0 * H;
    a.xs = (split && !q8) ? (void*)(h->d_xs + t0 * H) : nullptr;  // q8: the xs buffer holds the quantised rows instead
    a.flag = h->d_flag;
    if (q8) a.range_out = h->d_range_pairs;  // LayerNorm leaves its blocks' ranges for the quantising pass that follows
    const uint32_t ln_pairs = (T + 3) / 4;
    // several quantisation units in a batch the row-block kernels take: every product quantises its own rows with their
    // unit's parameters, the producers' pairs are reduced per unit (LayerNorm: a pair per row)
    static const bool q8_mu_on = [] { const char* e = std::getenv("CS_Q8_ROWS_UNITS"); return !(e && e[0] == '0'); }();
    const bool q8_mu = q8 && q8_mu_on && h->cur_units > 1 && q8_rows_from_source(T, H) && T <= h->cap_range_pairs;
    a.range_rows = q8_mu;
    _Float16* xs = reinterpret_cast<_Float16*>(h->d_xs + t0 * H);
    _Float16* ctxs = reinterpret_cast<_Float16*>(ctx);
    _Float16* mids = reinterpret_cast<_Float16*>(mid);
    const SplitLayer sl = split_layer(c);
    static const uint32_t split_k_min = [] { const char* e = std::getenv("CS_GEMM_SPLITK_MIN_M"); return e ? (uint32_t)std::atoi(e) : 1100u; }();
    static const uint32_t split_k_max = [] { const char* e = std::getenv("CS_GEMM_SPLITK_MAX_M"); return e ? (uint32_t)std::atoi(e) : 6144u; }();
    // device us per forward, fused / FFN-down in 3 K slices / out-proj too: 1,280 rows 1320 / 1020 / 971, 2,048
    // 1331 / 1052 / 1021, 4,096 1538 / 1311 / 1328, 6,144 1841 / 1619 / 1654, 8,192 2210 / 2264 / -
    // two slices up to 10,240 rows: 7,168 rows 2048 -> 1891 us, 8,192 2203 -> 2060, 10,240 2443 -> 2369, 12,288 3034 -> 3167
    static const uint32_t split_k_max2 = [] { const char* e = std::getenv("CS_GEMM_SPLITK_MAX2_M"); return e ? (uint32_t)std::atoi(e) : 10240u; }();
    static const uint32_t split_k_ao_max = [] { const char* e = std::getenv("CS_GEMM_SPLITK_AO_MAX_M"); return e ? (uint32_t)std::atoi(e) : 2560u; }();
    // stage profile: an event after each kernel (only on the one-stream path, see forward())
    auto mark = [&](int tag) -> int32_t {
        if (!h->stage_profile) return CS_OK;
        const size_t i = h->stage_tag.size() + 1;
        while (h->stage_ev.size() <= i) {
            hipEvent_t e;
            CS_HIP(hipEventCreate(&e));
            h->stage_ev.push_back(e);
        }
        if (tag < 0) { CS_HIP(hipEventRecord(h->stage_ev[0], s)); return CS_OK; }
        CS_HIP(hipEventRecord(h->stage_ev[i], s));
        h->stage_tag.push_back(tag);
        return CS_OK;
    };
    // dense layer: the persistent 128 x 384 one-accumulator kernel from wide_min_m token rows on (gemm_wide.hip),
    // else the 128 x 128 / skinny kernels of gemm_split.hip
    // A persistent block owns whole 128 x 384 tiles, so a launch needs about one tile per CU to fill the chip: the wide
    // kernel takes a layer when its tiles cover >= 85 % of the CUs, or from wide_min_m rows when the other half-batch
    // runs beside it on the second stream (measured, device ms per forward, wide / 128 x 128: 32 x 256 tokens 2.67 /
    // 2.08, 64 x 256 4.01 / 3.57 — one stream, N = 384 layers leave half the chip idle — 128 x 256 5.95 / 6.40,
    // 256 x 256 11.4 / 12.5).
    static const uint32_t wide_min_m = [] { const char* e = std::getenv("CS_GEMM_WIDE_MIN_M"); return e ? (uint32_t)std::atoll(e) : 12288u; }();
    auto takes_wide = [&](uint32_t Mr, uint32_t Nn, uint32_t Kk) {
        if (!h->wide_ok || !wide_min_m || !gemm_wide_supported(Nn, Kk) || Nn % 384) return false;
        const uint32_t tiles = ((Mr + 127) / 128) * (Nn / 384);
        return tiles >= 218 || (h->streams_in_flight >= 2 && Mr >= wide_min_m);
    };
    // Mid-size launches (the reference's 32-chunk calls: 8,192 token rows): the 128 x 128 grid is 1.1 rounds of
    // blocks for QKV (576 tiles on 512 slots); 128 x 192 tiles at two blocks per CU make it ONE round (384 tiles for
    // QKV, 512 for FFN-up).  Taken when that single round is at least 70 % full.
    static const bool mid192 = [] { const char* e = std::getenv("CS_GEMM_WIDE_MID"); return !(e && e[0] == '0'); }();
    auto takes_192 = [&](uint32_t Mr, uint32_t Nn, uint32_t Kk) {
        if (!mid192 || !h->wide_ok || !gemm_wide_supported(Nn, Kk) || h->streams_in_flight >= 2) return false;
        const uint32_t tiles = ((Mr + 127) / 128) * (Nn / 192);
        return tiles >= 358 && tiles <= 512;
    };
    auto dense = [&](int epi, const _Float16* Ain, const _Float16* Wt, const float* bias, const float* resid, float* Cf,
                     _Float16* Csp, uint32_t Mr, uint32_t Nn, uint32_t Kk) -> int32_t {
        if (takes_wide(Mr, Nn, Kk)) return launch_gemm_wide(epi, Ain, Wt, bias, resid, Cf, Csp, Mr, Nn, Kk, h->d_flag, s);
        if (takes_192(Mr, Nn, Kk)) return launch_gemm_wide(epi, Ain, Wt, bias, resid, Cf, Csp, Mr, Nn, Kk, h->d_flag, s, 192);
        return launch_gemm_split(epi, Ain, Wt, bias, resid, Cf, Csp, Mr, Nn, Kk, h->d_flag, s);
    };
    // ---- a few short sequences (under 200 token rows: the query side) ----
    // small_path.hip: LayerNorm as the prologue of the dense layer that reads it, FFN-down as four K slices summed by the
    // LayerNorm that follows: 62 launches per 12-layer forward instead of 86, none of them pulling 196 KB through one CU
    // (CS_SMALL_PATH=0: the general small-batch kernels below).  CS_SMALL_FORWARD=1: the same arithmetic as ONE launch
    // (small_forward.hip) — bit-identical, measured slower than the launches: opt-in.
    h->sf_ran = false;
    const char* e0 = std::getenv("CS_SMALL_PATH");  // (read per forward: tests flip it mid-process)
    const bool sp_on = !(e0 && e0[0] == '0');
    if (mode == CS_GEMM_SPLIT_F16 && sp_on && !nomic && b0 == 0 && T < 200 && small_path_supported(H, I, T)) {
        if (!h->d_sp_ws) CS_HIP(hipMalloc(&h->d_sp_ws, (size_t)5 * SP_MAX_ROWS * H * sizeof(float)));
        float* parts = h->d_sp_ws;                                   // [4][T][H]
        float* xa = h->d_sp_ws + (size_t)4 * SP_MAX_ROWS * H;        // [T][H]
        float* y = h->d_xs + t0 * H;                                  // [T][H] (the split copy of x is not used on this path)
        _Float16* qkvs = reinterpret_cast<_Float16*>(qkv);
        const char* e1 = std::getenv("CS_SMALL_FORWARD");  // (read per forward: tests and A/B runs flip it mid-process)
        if (e1 && e1[0] == '1' && h->d_sf_layers && !h->sf_off && !h->stage_profile && small_forward_supported(H, I, c.heads, T, L)) {
            uint32_t hb = L <= 32 ? 4u : (L <= 64 ? 2u : 1u);  // heads per attention block, as launch_attention_sh2 packs them
            if (const char* ph = std::getenv("CS_ATTN_PACK_HEADS")) if (ph[0] == '0') hb = 1;
            while (c.heads % hb) hb >>= 1;
            SfArgs sa{};
            sa.ids = a.ids; sa.mask = mask; sa.word = a.word; sa.pos = a.pos; sa.type0 = a.type0; sa.emb_g = a.g; sa.emb_b = a.b;
            sa.layers = h->d_sf_layers; sa.n_layers = c.layers; sa.eps = c.layer_norm_eps;
            sa.T = T; sa.L = L; sa.B = nb; sa.vocab = c.vocab_size; sa.heads = c.heads; sa.hb = hb;
            sa.X = x; sa.XA = xa; sa.Y = y; sa.PARTS = parts; sa.QKVS = qkvs; sa.CTXS = ctxs;
            sa.MIDS = reinterpret_cast<_Float16*>(mid); sa.flag = h->d_flag; sa.sync = h->d_sf_sync;
            sa.dbg = h->d_sf_dbg;
            CS_HIP(hipMemsetAsync(h->d_sf_sync, 0, 16, s));
            CS_TRY(launch_small_forward(sa, s));
            h->sf_ran = true;
            h->last_hidden_partial = false;
            CS_TRY(launch_row_kernel(2, a, H, s));  // E7 + E8
            return CS_OK;
        }
        _Float16* ctxs2 = ctxs;
        _Float16* mids2 = reinterpret_cast<_Float16*>(mid);
        const SplitLayer sl2 = split_layer(c);
        CS_TRY(mark(-1));
        for (uint32_t l = 0; l < c.layers; ++l) {
            cs_bert_layer_offsets lo, lp;
            cs_bert_layer_layout(&c, &h->off, l, &lo);
            if (l) cs_bert_layer_layout(&c, &h->off, l - 1, &lp);
            const _Float16* ws = h->d_wsplit + (size_t)l * sl2.total;
            SpLnGemmArgs g1{};
            g1.Y = y; g1.parts = parts; g1.parts_bias = l ? P + lp.down_b : nullptr; g1.X = x;
            g1.ids = a.ids; g1.word = a.word; g1.pos = a.pos; g1.type0 = a.type0; g1.L = L; g1.vocab = c.vocab_size;
            g1.ln_g = l ? P + lp.out_ln_g : a.g; g1.ln_b = l ? P + lp.out_ln_b : a.b; g1.eps = c.layer_norm_eps;
            g1.Xout = xa; g1.W = ws + sl2.qkv; g1.bias = h->d_bqkv + (size_t)l * 3 * H; g1.Cs = qkvs; g1.T = T; g1.N = 3 * H; g1.flag = h->d_flag;
            CS_TRY(launch_sp_ln_gemm(SH_OUT_SPLIT, l ? 1 : 2, g1, H, s));                                        // (E1 | LN) + E2
            CS_TRY(mark(CS_STAGE_QKV));
            CS_TRY(launch_attention_sh2(qkvs, mask, ctxs2, h->d_flag, nb, L, H, c.heads, s));                      // E3
            CS_TRY(mark(CS_STAGE_ATTENTION));
            CS_TRY(launch_gemm_split(SH_OUT_F32_RESID, ctxs2, ws + sl2.ao, P + lo.ao_b, xa, y, nullptr, T, H, H, h->d_flag, s));  // E4 -> y
            CS_TRY(mark(CS_STAGE_OUT_PROJ));
            SpLnGemmArgs g4 = g1;
            g4.ln_g = P + lo.ao_ln_g; g4.ln_b = P + lo.ao_ln_b; g4.Xout = x; g4.W = ws + sl2.up; g4.bias = P + lo.up_b; g4.Cs = mids2; g4.N = I;
            CS_TRY(launch_sp_ln_gemm(SH_OUT_SPLIT_GELU, 0, g4, H, s));                                           // LN + E5
            CS_TRY(mark(CS_STAGE_FFN_UP));
            CS_TRY(launch_sp_partial(mids2, ws + sl2.down, parts, T, H, H, s));                                  // E6, four K slices
            CS_TRY(mark(CS_STAGE_FFN_DOWN));
        }
        cs_bert_layer_offsets ll;
        cs_bert_layer_layout(&c, &h->off, c.layers - 1, &ll);
        a.parts = parts; a.nparts = 4; a.bias = P + ll.down_b; a.g = P + ll.out_ln_g; a.b = P + ll.out_ln_b;
        a.xs = nullptr;
        CS_TRY(launch_row_kernel(3, a, H, s));  // the last LayerNorm: (slabs + bias) + x -> x
        CS_TRY(mark(CS_STAGE_LN_FFN));
        h->last_hidden_partial = false;
        CS_TRY(launch_row_kernel(2, a, H, s));  // E7 + E8
        CS_TRY(mark(CS_STAGE_POOL));
        return CS_OK;
    }
    CS_TRY(mark(-1));
    CS_TRY(launch_row_kernel(0, a, H, s));  // E1
    CS_TRY(mark(CS_STAGE_EMBED_LN));
    for (uint32_t l = 0; l < c.layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(&c, &h->off, l, &lo);
        const float* bqkv = h->d_bqkv + (size_t)l * 3 * H;
        if (q8) {
            // Every Linear as the quantised file's graph runs it: DynamicQuantizeLinear of its input (one range per
            // call tensor), MatMulInteger on the int8 MFMA, * (x_scale * W_scale), + bias (gemm_q8.hip)
            const Q8Layer ql = q8_layer(H, I);
            const int8_t* wq = h->d_wq8 + (size_t)l * ql.total;
            const Q8ColMeta* cm = h->d_cmeta + (size_t)l * (5 * (size_t)H + I);
            const uint32_t U = h->cur_units;
            uint32_t* rg = h->d_range + (size_t)l * 4 * Q8_RANGE_WORDS * U;
            const size_t rstep = (size_t)Q8_RANGE_WORDS * U;
            // several units in the batch: every row carries its unit's slot, ranges come from passes over the tensors
            // (the producers' per-block ranges and the two-pass FFN-up assume one unit)
            const uint32_t* rs = U > 1 ? h->d_row_slot : nullptr;
            int8_t* xq = reinterpret_cast<int8_t*>(h->d_xs + t0 * H);  // [T][<= 4H] bytes
            Q8RowMeta* rm = h->d_rmeta + t0;
            _Float16* qkvs = reinterpret_cast<_Float16*>(qkv);
            float* rp = h->d_range_pairs;
            if (!rs && T <= q8_skinny_max_m() && I <= 3072 && (uint64_t)(I / 16) * ((T + 15) / 16) <= h->cap_range_pairs2) {
                // a few token rows (queries): one launch per Linear — range reduction and quantisation inside the product
                float* rp2 = rp + 2 * h->cap_range_pairs;
                CS_TRY(launch_gemm_q8_skinny(SH_OUT_SPLIT, Q8_SRC_F32, x, rp, ln_pairs, wq + ql.qkv, cm, nullptr, nullptr, qkvs, T, 3 * H, H, h->d_flag,
                                             nullptr, nullptr, s));  // E2
                CS_TRY(mark(CS_STAGE_QKV));
                uint32_t att_pairs = 0, up_pairs = 0;
                CS_TRY(launch_attention_sh2(qkvs, mask, ctxs, h->d_flag, nb, L, H, c.heads, s, rp, &att_pairs));  // E3
                CS_TRY(mark(CS_STAGE_ATTENTION));
                if (!att_pairs) return fail(CS_ERR_UNSUPPORTED, "attention kernel without range pairs in the few-rows quantised path");
                CS_TRY(launch_gemm_q8_skinny(SH_OUT_F32_RESID, Q8_SRC_SPLIT, ctxs, rp, att_pairs, wq + ql.ao, cm + 3 * H, x, x, nullptr, T, H, H,
                                             h->d_flag, nullptr, nullptr, s));  // E4
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
                a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
                CS_TRY(launch_row_kernel(1, a, H, s));
                CS_TRY(mark(CS_STAGE_LN_ATTN));
                CS_TRY(launch_gemm_q8_skinny(SH_OUT_SPLIT_GELU, Q8_SRC_F32, x, rp, ln_pairs, wq + ql.up, cm + 4 * H, nullptr, nullptr, mids, T, I, H,
                                             h->d_flag, rp2, &up_pairs, s));  // E5
                CS_TRY(mark(CS_STAGE_FFN_UP));
                CS_TRY(launch_gemm_q8_skinny(SH_OUT_F32_RESID, Q8_SRC_SPLIT, mids, rp2, up_pairs, wq + ql.down, cm + 4 * H + I, x, x, nullptr, T, H, I,
                                             h->d_flag, nullptr, nullptr, s));  // E6
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
                a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
                CS_TRY(launch_row_kernel(1, a, H, s));
                CS_TRY(mark(CS_STAGE_LN_FFN));
                if (l + 1 == c.layers) h->last_hidden_partial = false;
                continue;
            }
            if (!rs && q8_rows_from_source(T, H)) {
                // one unit, K = 384, a row block per CU: the products quantise their own rows on the way in — per tensor only
                // its range is needed first (a reduction of the pairs its producer left)
                CS_TRY(launch_q8_range(Q8_SRC_F32, x, T, H, rg, s, rp, ln_pairs));
                CS_TRY(launch_gemm_q8_from_source(SH_OUT_SPLIT, Q8_SRC_F32, x, rg, wq + ql.qkv, cm, bqkv, nullptr, nullptr, qkvs, T, 3 * H, H, h->d_flag, s));  // E2
                CS_TRY(mark(CS_STAGE_QKV));
                uint32_t att_pairs = 0;
                CS_TRY(launch_attention_sh2(qkvs, mask, ctxs, h->d_flag, nb, L, H, c.heads, s, rp, &att_pairs));  // E3
                CS_TRY(mark(CS_STAGE_ATTENTION));
                CS_TRY(launch_q8_range(Q8_SRC_SPLIT, ctxs, T, H, rg + rstep, s, rp, att_pairs));
                CS_TRY(launch_gemm_q8_from_source(SH_OUT_F32_RESID, Q8_SRC_SPLIT, ctxs, rg + rstep, wq + ql.ao, cm + 3 * H, P + lo.ao_b, x, x, nullptr, T, H, H,
                                                  h->d_flag, s));  // E4
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
                a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
                CS_TRY(launch_row_kernel(1, a, H, s));
                CS_TRY(mark(CS_STAGE_LN_ATTN));
                CS_TRY(launch_q8_range(Q8_SRC_F32, x, T, H, rg + 2 * rstep, s, rp, ln_pairs));
                int8_t* midq = reinterpret_cast<int8_t*>(mid);
                Q8RowMeta* rm2 = h->d_rmeta2 + t0;
                CS_TRY(launch_gemm_q8_gelu_requant_from_source(x, rg + 2 * rstep, wq + ql.up, cm + 4 * H, P + lo.up_b, T, I, H, rg + 3 * rstep, midq, rm2, s));  // E5
                CS_TRY(mark(CS_STAGE_FFN_UP));
                CS_TRY(launch_gemm_q8(SH_OUT_F32_RESID, midq, rm2, wq + ql.down, cm + 4 * H + I, P + lo.down_b, x, x, nullptr, T, H, I, h->d_flag, s));  // E6
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
                a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
                CS_TRY(launch_row_kernel(1, a, H, s));
                CS_TRY(mark(CS_STAGE_LN_FFN));
                if (l + 1 == c.layers) h->last_hidden_partial = false;
                continue;
            }
            if (rs && l == 0) CS_TRY(launch_q8_row_slots(h->d_seq_unit, h->d_unit_len, T, L, h->d_row_slot, s));
            if (q8_mu) {
                // the one-unit path above with every range kept per unit
                const uint32_t* su = h->d_seq_unit + b0;
                CS_TRY(launch_q8_range_units(rp, L, true, su, h->d_unit_len, nb, U, rg, s));
                CS_TRY(launch_gemm_q8_from_source(SH_OUT_SPLIT, Q8_SRC_F32, x, rg, wq + ql.qkv, cm, bqkv, nullptr, nullptr, qkvs, T, 3 * H, H, h->d_flag, s, rs));  // E2
                CS_TRY(mark(CS_STAGE_QKV));
                uint32_t att_pairs = 0;
                CS_TRY(launch_attention_sh2(qkvs, mask, ctxs, h->d_flag, nb, L, H, c.heads, s, rp, &att_pairs, su, h->d_unit_len));  // E3
                CS_TRY(mark(CS_STAGE_ATTENTION));
                if (!att_pairs || att_pairs > h->cap_range_pairs)
                    return fail(CS_ERR_HIP, "attention range pairs (%u) do not fit the pair buffer (%zu)", att_pairs, h->cap_range_pairs);
                CS_TRY(launch_q8_range_units(rp, att_pairs / nb, false, su, h->d_unit_len, nb, U, rg + rstep, s));
                CS_TRY(launch_gemm_q8_from_source(SH_OUT_F32_RESID, Q8_SRC_SPLIT, ctxs, rg + rstep, wq + ql.ao, cm + 3 * H, P + lo.ao_b, x, x, nullptr, T, H, H,
                                                  h->d_flag, s, rs));  // E4
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
                a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
                CS_TRY(launch_row_kernel(1, a, H, s));
                CS_TRY(mark(CS_STAGE_LN_ATTN));
                CS_TRY(launch_q8_range_units(rp, L, true, su, h->d_unit_len, nb, U, rg + 2 * rstep, s));
                int8_t* midq = reinterpret_cast<int8_t*>(mid);
                Q8RowMeta* rm2 = h->d_rmeta2 + t0;
                CS_TRY(launch_gemm_q8_gelu_requant_from_source(x, rg + 2 * rstep, wq + ql.up, cm + 4 * H, P + lo.up_b, T, I, H, rg + 3 * rstep, midq, rm2, s, rs));  // E5
                CS_TRY(mark(CS_STAGE_FFN_UP));
                CS_TRY(launch_gemm_q8(SH_OUT_F32_RESID, midq, rm2, wq + ql.down, cm + 4 * H + I, P + lo.down_b, x, x, nullptr, T, H, I, h->d_flag, s));  // E6
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
                a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
                CS_TRY(launch_row_kernel(1, a, H, s));
                CS_TRY(mark(CS_STAGE_LN_FFN));
                if (l + 1 == c.layers) h->last_hidden_partial = false;
                continue;
            }
            CS_TRY(launch_q8_quantize(Q8_SRC_F32, x, T, H, rg, rs, xq, rm, s, rp, ln_pairs));
            CS_TRY(launch_gemm_q8(SH_OUT_SPLIT, xq, rm, wq + ql.qkv, cm, bqkv, nullptr, nullptr, qkvs, T, 3 * H, H, h->d_flag, s));  // E2
            CS_TRY(mark(CS_STAGE_QKV));
            uint32_t att_pairs = 0;
            CS_TRY(launch_attention_sh2(qkvs, mask, ctxs, h->d_flag, nb, L, H, c.heads, s, rp, &att_pairs));  // E3
            if (att_pairs > h->cap_range_pairs) return fail(CS_ERR_HIP, "range pair buffer too small (%u > %zu)", att_pairs, h->cap_range_pairs);
            CS_TRY(mark(CS_STAGE_ATTENTION));
            CS_TRY(launch_q8_quantize(Q8_SRC_SPLIT, ctxs, T, H, rg + rstep, rs, xq, rm, s, rp, att_pairs));
            CS_TRY(launch_gemm_q8(SH_OUT_F32_RESID, xq, rm, wq + ql.ao, cm + 3 * H, P + lo.ao_b, x, x, nullptr, T, H, H, h->d_flag, s));  // E4
            CS_TRY(mark(CS_STAGE_OUT_PROJ));
            a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
            CS_TRY(launch_row_kernel(1, a, H, s));
            CS_TRY(mark(CS_STAGE_LN_ATTN));
            CS_TRY(launch_q8_quantize(Q8_SRC_F32, x, T, H, rg + 2 * rstep, rs, xq, rm, s, rp, ln_pairs));
            // E5: GELU(x W1^T + b1) leaves already re-quantised for E6 (two passes over the int8 product instead of 1.2 GB of
            // f32-class hand-over at 65,536 rows: launch_gemm_q8_gelu_requant)
            int8_t* midq = reinterpret_cast<int8_t*>(mid);
            Q8RowMeta* rm2 = h->d_rmeta2 + t0;
            if (rs) {  // several units: GELU output in split form, then its own range + quantising passes (into the x_q buffer)
                CS_TRY(launch_gemm_q8(SH_OUT_SPLIT_GELU, xq, rm, wq + ql.up, cm + 4 * H, P + lo.up_b, nullptr, nullptr, mids, T, I, H, h->d_flag, s));
                CS_TRY(mark(CS_STAGE_FFN_UP));
                CS_TRY(launch_q8_quantize(Q8_SRC_SPLIT, mids, T, I, rg + 3 * rstep, rs, xq, rm, s));
                midq = xq;
                rm2 = rm;
            } else {
                CS_TRY(launch_gemm_q8_gelu_requant(xq, rm, wq + ql.up, cm + 4 * H, P + lo.up_b, T, I, H, rg + 3 * rstep, midq, rm2, s));
                CS_TRY(mark(CS_STAGE_FFN_UP));
            }
            CS_TRY(launch_gemm_q8(SH_OUT_F32_RESID, midq, rm2, wq + ql.down, cm + 4 * H + I, P + lo.down_b, x, x, nullptr, T, H, I, h->d_flag, s));  // E6
            CS_TRY(mark(CS_STAGE_FFN_DOWN));
            a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
            CS_TRY(launch_row_kernel(1, a, H, s));
            CS_TRY(mark(CS_STAGE_LN_FFN));
            if (l + 1 == c.layers) h->last_hidden_partial = false;
        } else if (split) {
            const _Float16* ws = h->d_wsplit + (size_t)l * sl.total;
            {
                _Float16* qkvs = reinterpret_cast<_Float16*>(qkv);  // [T][3H/32][64] f16: same bytes as the f32 qkv
                // CLS pooling reads ONE row per sequence of the last layer: its attention needs every key and value but
                // only the CLS query, and everything behind it runs on nb rows instead of nb * L (cls_tail.hip).  Same
                // embedding, 1/12 less work at 12 layers.  Compact rows live in the (idle) intermediate buffer of the slice.
                static const bool cls_tail_on = [] { const char* e = std::getenv("CS_ENCODER_CLS_TAIL"); return !(e && e[0] == '0'); }();
                static const uint32_t cls_tail_min = [] { const char* e = std::getenv("CS_ENCODER_CLS_TAIL_MIN_TOKENS"); return e ? (uint32_t)std::atoll(e) : 4096u; }();
                // ... where the tail's kernels and scratch fit (else the full layer, never an error): attention_cls_kernel
                // takes <= 512 keys and head_dim 32 | 64; the compact rows (4 nb H + nb I floats) live in the slice's
                // [T, I] intermediate buffer
                const uint32_t dh_tail = c.heads ? H / c.heads : 0;
                const bool cls_tail_fits = L <= 512 && (dh_tail == 32 || dh_tail == 64) && H % c.heads == 0 &&
                                           (uint64_t)(L - 1) * I >= (uint64_t)4 * H;
                if (l + 1 == c.layers) h->last_hidden_partial = false;
                if (cls_tail_on && cls_tail_fits && !nomic && c.pooling == CS_POOL_CLS && l + 1 == c.layers && T >= cls_tail_min && L >= 16) {
                    h->last_hidden_partial = true;
                    float* x_cls = mid;                                            // [nb, H] f32
                    _Float16* xs_cls = reinterpret_cast<_Float16*>(mid + (size_t)nb * H);       // [nb][H/32][64]
                    _Float16* ctxs_cls = reinterpret_cast<_Float16*>(mid + (size_t)2 * nb * H);
                    _Float16* q_cls = reinterpret_cast<_Float16*>(mid + (size_t)3 * nb * H);
                    _Float16* mids_cls = reinterpret_cast<_Float16*>(mid + (size_t)4 * nb * H);  // [nb][I/32][64]
                    // E2: K and V for every token (the packed weight's rows H .. 3H: [T][2H/32][64]), Q for the CLS rows only
                    CS_TRY(dense(SH_OUT_SPLIT, xs, ws + sl.qkv + (size_t)H * H * 2, bqkv + H, nullptr, nullptr, qkvs, T, 2 * H, H));
                    CS_TRY(launch_gather_cls(xs, x_cls, xs_cls, nb, L, H, s));
                    CS_TRY(launch_gemm_split(SH_OUT_SPLIT, xs_cls, ws + sl.qkv, bqkv, nullptr, nullptr, q_cls, nb, H, H, h->d_flag, s));
                    CS_TRY(mark(CS_STAGE_QKV));
                    CS_TRY(launch_attention_cls(q_cls, qkvs, mask, ctxs_cls, h->d_flag, nb, L, H, c.heads, s));   // E3, one query per sequence
                    CS_TRY(mark(CS_STAGE_ATTENTION));
                    EncoderLaunch t = a;
                    t.x = x_cls; t.xs = xs_cls; t.T = nb; t.L = 1; t.B = nb;
                    CS_TRY(launch_gemm_split(SH_OUT_F32_RESID, ctxs_cls, ws + sl.ao, P + lo.ao_b, x_cls, x_cls, nullptr, nb, H, H, h->d_flag, s));  // E4
                    CS_TRY(mark(CS_STAGE_OUT_PROJ));
                    t.g = P + lo.ao_ln_g; t.b = P + lo.ao_ln_b;
                    CS_TRY(launch_row_kernel(1, t, H, s));
                    CS_TRY(mark(CS_STAGE_LN_ATTN));
                    CS_TRY(launch_gemm_split(SH_OUT_SPLIT_GELU, xs_cls, ws + sl.up, P + lo.up_b, nullptr, nullptr, mids_cls, nb, I, H, h->d_flag, s));  // E5
                    CS_TRY(mark(CS_STAGE_FFN_UP));
                    CS_TRY(launch_gemm_split(SH_OUT_F32_RESID, mids_cls, ws + sl.down, P + lo.down_b, x_cls, x_cls, nullptr, nb, H, I, h->d_flag, s));  // E6
                    CS_TRY(mark(CS_STAGE_FFN_DOWN));
                    t.g = P + lo.out_ln_g; t.b = P + lo.out_ln_b;
                    CS_TRY(launch_row_kernel(1, t, H, s));
                    CS_TRY(mark(CS_STAGE_LN_FFN));
                    CS_TRY(launch_row_kernel(2, t, H, s));  // E7 + E8 on the compact rows (L = 1: row b IS the CLS row)
                    CS_TRY(mark(CS_STAGE_POOL));
                    return CS_OK;
                }
                CS_TRY(dense(SH_OUT_SPLIT, xs, ws + sl.qkv, bqkv, nullptr, nullptr, qkvs, T, 3 * H, H));  // E2
                if (rotary) CS_TRY(launch_rope_split(qkvs, h->d_rope, T, L, H, c.heads, h->d_flag, s));  // rotary map on Q and K (nomic.hip)
                if (qknorm) CS_TRY(launch_qk_layernorm_split(qkvs, P + lo.qln_g, c.layer_norm_eps, T, H, h->d_flag, s));  // JinaBert qk-post-norm
                CS_TRY(mark(CS_STAGE_QKV));
                CS_TRY(launch_attention_sh2(qkvs, mask, ctxs, h->d_flag, nb, L, H, c.heads, s, nullptr, nullptr, nullptr, nullptr, alibi));  // E3
                CS_TRY(mark(CS_STAGE_ATTENTION));
            }
            a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
            // N = 384 at indexing batch sizes: dense layer + residual + LayerNorm in one kernel (gemm_wide.hip)
            static const bool ln_fuse_on = [] { const char* e = std::getenv("CS_GEMM_WIDE_LN"); return !(e && e[0] == '0'); }();
            const bool fuse_ln = ln_fuse_on && H == 384 && takes_wide(T, H, H);
            static const bool split_resid_on = [] { const char* e = std::getenv("CS_GEMM_WIDE_LN_SPLIT_RESID"); return !(e && e[0] == '0'); }();
            const bool split_resid = fuse_ln && split_resid_on;  // every N = 384 layer of this forward is fused or none is
            if (fuse_ln) {
                // the residual stream is carried in split form alone between the fused layers (read from xs, no f32
                // copy written: 100 MB less per layer and 65,536 rows); the last layer writes x for the pooling
                CS_TRY(launch_gemm_wide_ln(ctxs, ws + sl.ao, P + lo.ao_b, x, a.g, a.b, c.layer_norm_eps,
                                           split_resid ? nullptr : x, xs, T, H, h->d_flag, s, split_resid ? xs : nullptr));  // E4
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
            } else if (T > split_k_min && T <= split_k_max && T <= split_k_ao_max) {
                CS_TRY(launch_gemm_split_partial(ctxs, ws + sl.ao, qkv, T, H, H, 3, s));  // E4, K slices as for E6 below
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
                a.parts = qkv; a.nparts = 3; a.bias = P + lo.ao_b;
                CS_TRY(launch_row_kernel(3, a, H, s));
            } else {
                CS_TRY(dense(SH_OUT_F32_RESID, ctxs, ws + sl.ao, P + lo.ao_b, x, x, nullptr, T, H, H));  // E4
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
                CS_TRY(launch_row_kernel(1, a, H, s));
            }
            CS_TRY(mark(CS_STAGE_LN_ATTN));
            a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
            const _Float16* ffn_in = mids;  // E6's operand
            if (nomic) {
                // E5 of the gated feed-forward: ONE product over fc11's and fc12's rows ([2I, H], interleaved in groups of 16)
                // into the first 2I columns of the workspace, then value * silu(gate) into its last I columns — E6's operand
                _Float16* gated = reinterpret_cast<_Float16*>(mid + (size_t)T * 2 * I);
                const float* bup = h->d_bup + (size_t)l * 2 * I;
                static const bool gate_fused = [] { const char* e = std::getenv("CS_NOMIC_GATE_FUSED"); return !(e && e[0] == '0'); }();
                const bool w384 = takes_wide(T, 2 * I, H), w192 = !w384 && takes_192(T, 2 * I, H);
                if (gate_fused && (w384 || w192)) {  // the gate as the product's epilogue: the raw [T, 2I] tensor never exists
                    CS_TRY(launch_gemm_wide(jina ? GW_OUT_GEGLU : GW_OUT_SWIGLU, xs, ws + sl.up, bup, nullptr, nullptr, gated, T, 2 * I, H, h->d_flag, s, w192 ? 192 : 0));
                } else {
                    CS_TRY(dense(SH_OUT_SPLIT, xs, ws + sl.up, bup, nullptr, nullptr, mids, T, 2 * I, H));
                    CS_TRY(launch_swiglu_split(mids, gated, T, I, h->d_flag, s, jina));
                }
                ffn_in = gated;
            } else {
                CS_TRY(dense(SH_OUT_SPLIT_GELU, xs, ws + sl.up, P + lo.up_b, nullptr, nullptr, mids, T, I, H));    // E5
            }
            CS_TRY(mark(CS_STAGE_FFN_UP));
            if (fuse_ln) {
                CS_TRY(launch_gemm_wide_ln(ffn_in, ws + sl.down, P + lo.down_b, x, a.g, a.b, c.layer_norm_eps,
                                           (split_resid && l + 1 < c.layers) ? nullptr : x, xs, T, I, h->d_flag, s,
                                           split_resid ? xs : nullptr));  // E6
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
            } else if (T > split_k_min && T <= split_k_max2) {
                // a few thousand token rows: FFN-down is 3 x T / 128 blocks walking 48 K stages one exposed
                // latency each; three K slices per tile (two from 6,144 rows: still one round of blocks), partial
                // slabs in the qkv buffer (free by now), summed with bias and residual by the LayerNorm that follows
                const uint32_t ks = T <= split_k_max ? 3 : 2;
                CS_TRY(launch_gemm_split_partial(ffn_in, ws + sl.down, qkv, T, H, I, ks, s));  // E6
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
                a.parts = qkv; a.nparts = ks; a.bias = P + lo.down_b;
                CS_TRY(launch_row_kernel(3, a, H, s));
            } else {
                CS_TRY(dense(SH_OUT_F32_RESID, ffn_in, ws + sl.down, P + lo.down_b, x, x, nullptr, T, H, I)); // E6
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
                CS_TRY(launch_row_kernel(1, a, H, s));
            }
            CS_TRY(mark(CS_STAGE_LN_FFN));
        } else {
            const float* wqkv = h->d_wqkv + (size_t)l * 3 * H * H;
            CS_TRY(launch_gemm(GEMM_BIAS, x, wqkv, bqkv, nullptr, qkv, T, 3 * H, H, s));        // E2
            if (rotary) CS_TRY(launch_rope_f32(qkv, h->d_rope, T, L, H, c.heads, s));
            if (qknorm) CS_TRY(launch_qk_layernorm_f32(qkv, P + lo.qln_g, c.layer_norm_eps, T, H, s));
            CS_TRY(mark(CS_STAGE_QKV));
            CS_TRY(launch_attention(qkv, mask, ctx, nb, L, H, c.heads, s, alibi));              // E3
            CS_TRY(mark(CS_STAGE_ATTENTION));
            CS_TRY(launch_gemm(GEMM_RESID, ctx, P + lo.ao_w, P + lo.ao_b, x, x, T, H, H, s));   // E4
            CS_TRY(mark(CS_STAGE_OUT_PROJ));
            a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
            CS_TRY(launch_row_kernel(1, a, H, s));
            CS_TRY(mark(CS_STAGE_LN_ATTN));
            if (nomic) {  // value and gate as two products, value *= silu(gate)
                float* gate = mid + (size_t)T * I;
                CS_TRY(launch_gemm(GEMM_BIAS, x, P + lo.up_w, P + lo.up_b, nullptr, mid, T, I, H, s));
                CS_TRY(launch_gemm(GEMM_BIAS, x, P + lo.gate_w, P + lo.gate_b, nullptr, gate, T, I, H, s));
                CS_TRY(launch_swiglu_f32(mid, gate, T, I, s, jina));
            } else {
                CS_TRY(launch_gemm(GEMM_GELU, x, P + lo.up_w, P + lo.up_b, nullptr, mid, T, I, H, s)); // E5
            }
            CS_TRY(mark(CS_STAGE_FFN_UP));
            CS_TRY(launch_gemm(GEMM_RESID, mid, P + lo.down_w, P + lo.down_b, x, x, T, H, I, s));  // E6
            CS_TRY(mark(CS_STAGE_FFN_DOWN));
            a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
            CS_TRY(launch_row_kernel(1, a, H, s));
            CS_TRY(mark(CS_STAGE_LN_FFN));
        }
    }
    CS_TRY(launch_row_kernel(2, a, H, s));  // E7 + E8
    CS_TRY(mark(CS_STAGE_POOL));
    return CS_OK;
}

// One mini-batch already on the device (d_ids/d_mask) -> d_pooled [B, H].  The batch is cut into
// two halves on two streams: each kernel alternates an MFMA-bound main loop with an HBM-bound
// epilogue (and attention / LayerNorm are memory-heavy throughout), so blocks of two different
// kernels sharing a CU keep both the matrix pipe and the memory system busy.
int32_t forward(cs_embedder* h, uint32_t B, uint32_t L, int mode) {
    hipStream_t s = h->stream;
    CS_HIP(hipEventRecord(h->ev0, s));
    if (mode != CS_GEMM_F32) CS_HIP(hipMemsetAsync(h->d_flag, 0, sizeof(uint32_t), s));
    if (mode == CS_GEMM_Q8_DYNAMIC)  // every range starts from (+0, +0)
        CS_HIP(hipMemsetAsync(h->d_range, 0, (size_t)h->cfg.layers * 4 * Q8_RANGE_WORDS * h->cur_units * sizeof(uint32_t), s));
    // Slicing pays from ~20,000 tokens (device us per forward, one stream / two: 16,384 tokens 3505 / 3542,
    // 24,576 5267 / 4916, 32,768 6517 / 6275, 49,152 9568 / 9437); below that it only multiplies launches
    // of kernels that already leave the chip part-empty.
    static const uint64_t stream_min_tokens = [] {
        const char* e = std::getenv("CS_ENCODER_STREAM_MIN_TOKENS");
        return e ? (uint64_t)std::atoll(e) : (uint64_t)20000;
    }();
    h->stage_tag.clear();
    // The persistent wide kernels give every CU a whole number of tiles when the tile counts of the three layer shapes
    // (T/128 x {1, 3, 4}) are multiples of the CU count; then one stream is as good or better (256 x 256 tokens: 11.05
    // vs 11.20 ms) and the second stream only helps where a last round of tiles would leave CUs idle (160 x 256: 8.08
    // one stream, 7.07 two).  CS_ENCODER_STREAMS forces the count either way.
    bool whole_rounds = false;
    if (mode == CS_GEMM_SPLIT_F16 && h->wide_ok && !h->streams_forced) {
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        const uint64_t mt = ((uint64_t)B * L + 127) / 128;
        auto eff = [&](uint64_t tiles) { return cus > 0 ? (double)tiles / (double)(((tiles + cus - 1) / cus) * cus) : 0.0; };
        whole_rounds = h->cfg.hidden == 384 && mt >= 218 && eff(mt) >= 0.96 && eff(3 * mt) >= 0.96 && eff(4 * mt) >= 0.96;
    }
    // (a quantised tensor is the WHOLE mini-batch: slices on several streams would each see their own range)
    if (!h->stage_profile && !whole_rounds && mode != CS_GEMM_Q8_DYNAMIC && h->n_streams >= 2 && B >= (uint32_t)h->n_streams &&
        (uint64_t)B * L >= stream_min_tokens) {
        const uint32_t ns = (uint32_t)h->n_streams;
        h->streams_in_flight = (int)ns;
        hipStream_t st[4] = {s, h->stream2, h->xstreams[0], h->xstreams[1]};
        hipEvent_t jn[4] = {nullptr, h->ev_join, h->xjoin[0], h->xjoin[1]};
        CS_HIP(hipEventRecord(h->ev_fork, s));
        for (uint32_t i = ns; i-- > 0;) {  // slice 0 last, on the caller-visible stream
            const uint32_t lo = (uint32_t)((uint64_t)B * i / ns), hi = (uint32_t)((uint64_t)B * (i + 1) / ns);
            if (i) CS_HIP(hipStreamWaitEvent(st[i], h->ev_fork, 0));
            CS_TRY(forward_range(h, st[i], lo, hi - lo, L, mode));
            if (i) CS_HIP(hipEventRecord(jn[i], st[i]));
        }
        for (uint32_t i = 1; i < ns; ++i) CS_HIP(hipStreamWaitEvent(s, jn[i], 0));
    } else {
        h->streams_in_flight = 1;
        CS_TRY(forward_range(h, s, 0, B, L, mode));
    }
    CS_HIP(hipEventRecord(h->ev1, s));
    h->last_B = B;
    h->last_L = L;
    return CS_OK;
}

uint32_t default_batch(const cs_embedder* h) {
    // embedder.rs:251-261: CODESEARCH_BATCH_SIZE (unparsable -> 256), else 256/128/64 by dims
    if (const char* env = std::getenv("CODESEARCH_BATCH_SIZE")) {
        char* end = nullptr;
        const long v = std::strtol(env, &end, 10);
        if (end != env && *end == '\0' && v > 0) return (uint32_t)v;
        return 256;
    }
    const uint32_t d = h->cfg.hidden;
    return d <= 384 ? 256 : (d <= 768 ? 128 : 64);
}

// Several quantisation units in ONE mini-batch (dynamic-quantisation mode: calls of the reference embedded together, each
// still quantised as the tensor it would have been on its own): the unit of every sequence, each unit's own padded length.
struct UnitSpec {
    const uint32_t* seq_unit = nullptr;  // [n]
    const uint32_t* unit_len = nullptr;  // [units]
    uint32_t units = 1;
};

// perm (optional, only with n <= batch): pooled row r of the mini-batch goes to out row perm[r].
// units (optional, only with n <= batch and CS_GEMM_Q8_DYNAMIC): see UnitSpec.
int32_t embed_impl(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n,
                   uint32_t seq_len, uint32_t batch, float* out, bool out_on_device,
                   const volatile int32_t* cancel, const uint32_t* perm = nullptr, const UnitSpec* units = nullptr) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (n == 0) return CS_OK;  // embedder.rs:271-273
    if (!ids || !mask || !out) return fail(CS_ERR_BAD_ARG, "null buffer");
    if (seq_len == 0 || seq_len > h->cfg.max_position)
        return fail(CS_ERR_BAD_ARG, "seq_len %u outside 1..%u (max_position_embeddings)", seq_len,
                    h->cfg.max_position);
    if (batch == 0) batch = default_batch(h);
    DeviceGuard g(h->device);
    const uint32_t H = h->cfg.hidden;
    const size_t bmax = n < batch ? (size_t)n : batch;
    CS_TRY(reserve(h, bmax, bmax * seq_len));
    for (uint64_t done = 0; done < n; done += batch) {
        if (cancel && *cancel)  // embedder.rs:280-282
            return fail(CS_ERR_CANCELLED, "Embedding interrupted by shutdown request");
        const uint32_t B = (uint32_t)((n - done) < batch ? (n - done) : batch);
        const size_t tok = (size_t)B * seq_len;
        const int32_t* bi = ids + done * seq_len;
        for (size_t i = 0; i < tok; ++i)
            if (bi[i] < 0 || (uint32_t)bi[i] >= h->cfg.vocab_size)
                return fail(CS_ERR_BAD_ARG, "Failed to generate embeddings: token id %d outside vocabulary of %u",
                            bi[i], h->cfg.vocab_size);
        CS_HIP(hipMemcpyAsync(h->d_ids, bi, tok * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        CS_HIP(hipMemcpyAsync(h->d_mask, mask + done * seq_len, tok * sizeof(int32_t),
                              hipMemcpyHostToDevice, h->stream));
        int mode = h->gemm_mode;
        h->cur_units = 1;
        if (units && units->units > 1 && mode == CS_GEMM_Q8_DYNAMIC && n <= batch) {
            CS_HIP(hipMemcpyAsync(h->d_seq_unit, units->seq_unit, B * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            CS_HIP(hipMemcpyAsync(h->d_unit_len, units->unit_len, units->units * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            h->cur_units = units->units;
        }
        CS_TRY(forward(h, B, seq_len, mode));
        if (mode == CS_GEMM_Q8_DYNAMIC) {
            uint32_t flag = 0;
            CS_HIP(hipMemcpyAsync(&flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
            CS_HIP(hipStreamSynchronize(h->stream));
            h->q8_forwards += 1;
            if (flag)  // Q / K / V or a GELU output beyond 65504: the f32 kernels would run a different graph — refuse
                return fail(CS_ERR_UNSUPPORTED, "Failed to generate embeddings: an activation of the quantised model left the "
                            "f16 range of the attention / GELU hand-over (|x| > 65504)");
        } else if (mode == CS_GEMM_SPLIT_F16) {
            uint32_t flag = 0;
            CS_HIP(hipMemcpyAsync(&flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
            if (h->sf_ran) {  // the one-launch forward: did it reach its end?
                uint32_t sync[4] = {0, 0, 0, 0};
                CS_HIP(hipMemcpyAsync(sync, h->d_sf_sync, sizeof sync, hipMemcpyDeviceToHost, h->stream));
                CS_HIP(hipStreamSynchronize(h->stream));
                h->sf_forwards += 1;
                if (h->d_sf_dbg && !sync[1]) {  // diagnostics: where the blocks' time went (medians over the 96 blocks, us)
                    std::vector<uint64_t> d(96 * 3 + 8);
                    CS_HIP(hipMemcpy(d.data(), h->d_sf_dbg, d.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
                    double med[3];
                    for (int k = 0; k < 3; ++k) {
                        std::vector<uint64_t> v;
                        for (int b = 0; b < 96; ++b) v.push_back(d[3 * b + k]);
                        std::sort(v.begin(), v.end());
                        med[k] = v[48] * 0.01;
                    }
                    fprintf(stderr, "small_forward B=%u L=%u: per block (median) compute %.1f us, store drain %.1f us, grid barriers %.1f us; "
                                    "block 0: %.1f / %.1f / %.1f; block 0's compute by phase kind: QKV %.1f attention %.1f out-proj %.1f FFN-up %.1f FFN-down %.1f\n",
                            B, seq_len, med[0], med[1], med[2], d[0] * 0.01, d[1] * 0.01, d[2] * 0.01, d[288] * 0.01, d[289] * 0.01,
                            d[290] * 0.01, d[291] * 0.01, d[292] * 0.01);
                }
                if (sync[1]) {  // a grid barrier gave up (blocks not co-resident): this mini-batch again, kernel by kernel
                    h->sf_fallbacks += 1;
                    h->sf_off = true;
                    const int32_t st = forward(h, B, seq_len, mode);
                    h->sf_off = false;
                    CS_TRY(st);
                    CS_HIP(hipMemcpyAsync(&flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
                }
            }
            CS_HIP(hipStreamSynchronize(h->stream));
            h->split_forwards += 1;
            if (flag) {  // an activation left the f16 range: redo this mini-batch on the exact-f32 MFMA
                h->range_fallbacks += 1;
                mode = CS_GEMM_F32;
                CS_TRY(forward(h, B, seq_len, mode));
            }
        }
        if (mode == CS_GEMM_F32) h->f32_forwards += 1;
        if (!perm) {
            CS_HIP(hipMemcpyAsync(out + done * H, h->d_pooled, (size_t)B * H * sizeof(float),
                                  out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
            CS_HIP(hipStreamSynchronize(h->stream));
        } else if (out_on_device) {
            CS_HIP(hipMemcpyAsync(h->d_perm, perm, B * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            const uint32_t h4 = H / 4;
            hipLaunchKernelGGL(scatter_rows_kernel, dim3((B * h4 + 255) / 256), dim3(256), 0, h->stream, h->d_pooled,
                               h->d_perm, out, B, h4);
            CS_HIP(hipGetLastError());
            CS_HIP(hipStreamSynchronize(h->stream));
        } else {
            h->h_pooled.resize((size_t)B * H);
            CS_HIP(hipMemcpyAsync(h->h_pooled.data(), h->d_pooled, (size_t)B * H * sizeof(float),
                                  hipMemcpyDeviceToHost, h->stream));
            CS_HIP(hipStreamSynchronize(h->stream));
            for (uint32_t r = 0; r < B; ++r)
                std::memcpy(out + (size_t)perm[r] * H, h->h_pooled.data() + (size_t)r * H, H * sizeof(float));
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) {
            h->forward_ms += ms;
            h->forwards += 1;
        }
        if (h->stage_profile && !h->stage_tag.empty()) {  // the stream is idle here (synchronised above)
            for (size_t i = 0; i < h->stage_tag.size(); ++i) {
                float us = 0.f;
                if (hipEventElapsedTime(&us, h->stage_ev[i], h->stage_ev[i + 1]) == hipSuccess)
                    h->stage_us[h->stage_tag[i]] += (double)us * 1e3;
            }
            h->stage_forwards += 1;
        }
    }
    return CS_OK;
}


// embed_batch_chunked from strings (embedder.rs:266-295).  Texts are taken in WINDOWS of 16 mini-batches:
// window w+1 is tokenised on host threads while the device runs window w, and inside a window the
// texts are grouped into mini-batches BY TOKEN COUNT (stable sort), each padded to its own longest
// sequence.  fastembed pads every mini-batch of consecutive texts to its longest member; padding is
// masked out of attention and pooling, so an embedding does not depend on what it was batched with
// beyond f32 rounding (asserted in tests/test_gpu_encoder.py), and on code chunks of mixed length the
// grouping removes ~1/3 of the padded tokens the device would otherwise compute.
// CS_EMBED_LENGTH_SORT=0 keeps the caller's order (mini-batches of consecutive texts, as fastembed).
struct TokenWindow {
    std::vector<std::vector<int32_t>> enc;
};

void tokenize_window(const cs_tokenizer* t, const char* utf8, const uint64_t* offsets, uint32_t n,
                     uint32_t max_length, TokenWindow* out) {
    cs::tokenize_texts(t, utf8, offsets, n, max_length, out->enc);
}

bool length_sort_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("CS_EMBED_LENGTH_SORT");
        return !(e && e[0] == '0');
    }();
    return on;
}

// One window of sequences, each a (ids, mask, length) view with every position >= length padding:
// group them into mini-batches by length, pad each mini-batch to ITS longest member, run it, and put
// row r of the result at out[order[r]].  mask == nullptr means "ones up to length".
struct SeqView { const int32_t* ids; const int32_t* mask; uint32_t len; };

int32_t run_window(cs_embedder* h, const std::vector<SeqView>& seqs, uint32_t batch, int32_t pad, float* out,
                   bool out_on_device, const volatile int32_t* cancel, std::vector<uint32_t>& order,
                   std::vector<int32_t>& ids, std::vector<int32_t>& mask) {
    const uint32_t wn = (uint32_t)seqs.size();
    // Length-grouped mini-batches are cut by TOKENS, not by rows: a mini-batch of `batch` short sequences is a fraction
    // of the token rows the dense layers are tuned on (256 x 256 = 65,536 for the 384-d models: whole tile rounds on
    // 256 CUs), so short sequences fill the same budget with more rows (up to 8 x batch).  Sorted ascending, the row
    // that would join next is also the new longest.  CS_EMBED_TOKEN_BATCH=0: `batch` rows whatever their length.
    static const bool token_batches = [] {
        const char* e = std::getenv("CS_EMBED_TOKEN_BATCH");
        return !(e && e[0] == '0');
    }();
    // (a quantised model's tensors are the reference's call units: `batch` consecutive texts, padded to their longest)
    const bool sorted = length_sort_enabled() && wn > batch && h->gemm_mode != CS_GEMM_Q8_DYNAMIC;
    const uint64_t budget = (uint64_t)batch * std::min<uint32_t>(256, h->cfg.max_position);
    const uint32_t max_rows = sorted && token_batches ? batch * 8 : batch;
    {   // workspace for the window's longest sequence once, not once per (growing) mini-batch
        size_t longest = 1;
        for (const SeqView& v : seqs) longest = std::max<size_t>(longest, v.len);
        const size_t bmax = std::min<size_t>(max_rows, wn);
        const size_t tokens = std::max<size_t>(std::min<size_t>(batch, wn) * longest, max_rows > batch ? (size_t)budget : 0);
        DeviceGuard g(h->device);
        CS_TRY(reserve(h, std::max(bmax, h->cap_seqs), std::max(tokens, h->cap_tokens)));
    }
    order.resize(wn);
    for (uint32_t i = 0; i < wn; ++i) order[i] = i;
    if (sorted)
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return seqs[a].len < seqs[b].len; });
    uint32_t B = 0;
    for (uint32_t b0 = 0; b0 < wn; b0 += B) {
        if (cancel && *cancel)  // embedder.rs:280-282
            return fail(CS_ERR_CANCELLED, "Embedding interrupted by shutdown request");
        B = std::min<uint32_t>(batch, wn - b0);
        while (b0 + B < wn && B < max_rows && (uint64_t)(B + 1) * seqs[order[b0 + B]].len <= budget) ++B;
        uint32_t L = 1;
        for (uint32_t r = 0; r < B; ++r) L = std::max(L, seqs[order[b0 + r]].len);
        ids.assign((size_t)B * L, pad);
        mask.assign((size_t)B * L, 0);
        for (uint32_t r = 0; r < B; ++r) {
            const SeqView& v = seqs[order[b0 + r]];
            std::copy(v.ids, v.ids + v.len, ids.begin() + (size_t)r * L);
            if (v.mask) std::copy(v.mask, v.mask + v.len, mask.begin() + (size_t)r * L);
            else std::fill(mask.begin() + (size_t)r * L, mask.begin() + (size_t)r * L + v.len, 1);
        }
        CS_TRY(embed_impl(h, ids.data(), mask.data(), B, L, B, out, out_on_device, nullptr, order.data() + b0));
    }
    return CS_OK;
}

int32_t embed_texts_impl(cs_embedder* h, const cs_tokenizer* t, const char* utf8, const uint64_t* offsets,
                         uint64_t n, uint32_t batch, float* out, bool out_on_device,
                         const volatile int32_t* cancel) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (!t) return fail(CS_ERR_BAD_ARG, "Failed to generate embeddings: no tokenizer attached");
    if (n == 0) return CS_OK;  // embedder.rs:271-273
    if (!utf8 || !offsets || !out) return fail(CS_ERR_BAD_ARG, "null buffer");
    for (uint64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(CS_ERR_BAD_ARG, "text offsets must be non-decreasing");
    if (batch == 0) batch = default_batch(h);
    const uint32_t max_length = h->cfg.max_position;
    const int32_t pad = cs_tokenizer_pad_id(t);  // [PAD], or <pad> of a unigram tokenizer.json
    const uint32_t H = h->cfg.hidden;
    const uint64_t window = (uint64_t)batch * 16;
    auto span = [&](uint64_t lo) { return (uint32_t)std::min<uint64_t>(window, n - lo); };
    TokenWindow cur, nxt;
    tokenize_window(t, utf8, offsets, span(0), max_length, &cur);
    std::vector<uint32_t> order;
    std::vector<int32_t> ids, mask;
    std::vector<SeqView> seqs;
    for (uint64_t lo = 0; lo < n; lo += window) {
        std::thread ahead;
        if (lo + window < n)
            ahead = std::thread(tokenize_window, t, utf8, offsets + lo + window, span(lo + window), max_length, &nxt);
        struct Joiner {
            std::thread& th;
            ~Joiner() { if (th.joinable()) th.join(); }
        } joiner{ahead};
        seqs.clear();
        for (const auto& e : cur.enc) seqs.push_back(SeqView{e.data(), nullptr, (uint32_t)e.size()});
        CS_TRY(run_window(h, seqs, batch, pad, out + lo * H, out_on_device, cancel, order, ids, mask));
        if (ahead.joinable()) ahead.join();
        std::swap(cur, nxt);
    }
    return CS_OK;
}

// cs_embedder_embed_ids with more than one mini-batch: the same windows over the caller's padded rows.
// A row's length is the position after its last mask bit; mini-batches are cut to their longest member
// (the columns dropped hold padding in every row of the mini-batch) and grouped by length.
int32_t embed_ids_windowed(cs_embedder* h, const int32_t* ids_in, const int32_t* mask_in, uint64_t n,
                           uint32_t seq_len, uint32_t batch, float* out, bool out_on_device,
                           const volatile int32_t* cancel) {
    const uint32_t H = h->cfg.hidden;
    const uint64_t window = (uint64_t)batch * 16;
    std::vector<uint32_t> order;
    std::vector<int32_t> ids, mask;
    std::vector<SeqView> seqs;
    for (uint64_t lo = 0; lo < n; lo += window) {
        const uint32_t wn = (uint32_t)std::min<uint64_t>(window, n - lo);
        seqs.clear();
        for (uint32_t i = 0; i < wn; ++i) {
            const int32_t* m = mask_in + (lo + i) * seq_len;
            uint32_t len = seq_len;
            while (len > 1 && m[len - 1] == 0) --len;
            seqs.push_back(SeqView{ids_in + (lo + i) * seq_len, m, len});
        }
        CS_TRY(run_window(h, seqs, batch, 0, out + lo * H, out_on_device, cancel, order, ids, mask));
    }
    return CS_OK;
}

int32_t embed_ids_entry(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n, uint32_t seq_len,
                        uint32_t batch, float* out, bool out_on_device, const volatile int32_t* cancel) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    const uint32_t b = batch ? batch : default_batch(h);
    // a single mini-batch runs exactly as given (cs_embedder_last_hidden then has the caller's [n, seq_len] layout)
    if (n <= b || !ids || !mask || !out || seq_len == 0 || seq_len > h->cfg.max_position || !length_sort_enabled() ||
        h->gemm_mode == CS_GEMM_Q8_DYNAMIC)
        return embed_impl(h, ids, mask, n, seq_len, batch, out, out_on_device, cancel);
    return embed_ids_windowed(h, ids, mask, n, seq_len, b, out, out_on_device, cancel);
}

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

}  // namespace

extern "C" {

void cs_bert_config_bge_small(cs_bert_config* cfg) {
    if (!cfg) return;
    cfg->vocab_size = 30522; cfg->hidden = 384; cfg->layers = 12; cfg->heads = 12;
    cfg->intermediate = 1536; cfg->max_position = 512; cfg->type_vocab_size = 2;
    cfg->layer_norm_eps = 1e-12f; cfg->pooling = CS_POOL_CLS;
    cfg->arch = CS_ARCH_BERT; cfg->rotary_base = 0.0f;
}

uint64_t cs_bert_param_count(const cs_bert_config* cfg) {
    if (!cfg) return 0;
    cs_bert_offsets off;
    cs_bert_layout(cfg, &off);
    return off.total;
}

static int32_t create_impl(const cs_bert_config* cfg, const float* params, uint64_t seed, const float* wscale,
                           int32_t device, cs_embedder** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!cfg) return fail(CS_ERR_BAD_ARG, "cfg is null");
    if (cfg->hidden == 0 || cfg->heads == 0 || cfg->hidden % cfg->heads || cfg->layers == 0 ||
        cfg->vocab_size == 0 || cfg->max_position == 0 || cfg->type_vocab_size == 0)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: inconsistent config");
    if (cfg->hidden != 384 && cfg->hidden != 768 && cfg->hidden != 1024)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: hidden size %u not supported", cfg->hidden);
    // head_dim 64: BGE-base / BGE-large / mxbai-large
    if (cfg->hidden / cfg->heads != 32 && cfg->hidden / cfg->heads != 64)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: head_dim %u not supported (32 or 64)",
                    cfg->hidden / cfg->heads);
    if (cfg->intermediate % 128 || cfg->hidden % 128)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: sizes must be multiples of 128");
    if (cfg->pooling != CS_POOL_CLS && cfg->pooling != CS_POOL_MEAN)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: unknown pooling %d", cfg->pooling);
    if (cfg->arch != CS_ARCH_BERT && !cs_arch_gated(cfg->arch))
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: unknown encoder family %u", cfg->arch);
    if (cs_arch_alibi(cfg->arch) && wscale)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: the dynamic-quantisation mode is not built for "
                    "the JinaBert encoder (create it from the dequantised weights: cs_embedder_create)");
    if (cfg->arch == CS_ARCH_NOMIC) {
        if (!(cfg->rotary_base > 1.0f) || !(cfg->rotary_base < 1.0e9f))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: rotary base %g", (double)cfg->rotary_base);
        // the quantised export's graph quantises the rotated and gated tensors in places of its own: not restated
        if (wscale) return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: the dynamic-quantisation mode is not built for "
                                "the Nomic encoder (create it from the dequantised weights: cs_embedder_create)");
    }
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(CS_ERR_HIP, "HIP device %d not available (%d visible); there is no CPU fallback", device, ndev);
    DeviceGuard g(device);
    cs_embedder* h = new cs_embedder();
    h->device = device;
    h->cfg = *cfg;
    cs_bert_layout(cfg, &h->off);
    const size_t H = cfg->hidden;
    auto cleanup = [&](int32_t s) { cs_embedder_destroy(h); return s; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess)
        return cleanup(fail(CS_ERR_HIP, "could not create stream/events"));
    if (hipMalloc(&h->d_params, h->off.total * sizeof(float)) != hipSuccess ||
        hipMalloc(&h->d_wqkv, (size_t)cfg->layers * 3 * H * H * sizeof(float)) != hipSuccess ||
        hipMalloc(&h->d_bqkv, (size_t)cfg->layers * 3 * H * sizeof(float)) != hipSuccess)
        return cleanup(fail(CS_ERR_OOM, "hipMalloc(parameters) failed"));
    int32_t s = CS_OK;
    if (params) {
        if (hipMemcpy(h->d_params, params, h->off.total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(CS_ERR_HIP, "parameter upload failed"));
    } else {
        s = launch_synth_params(h->d_params, *cfg, seed, h->stream);
        if (s != CS_OK) return cleanup(s);
    }
    // pack query|key|value into one [3H, H] weight and [3H] bias per layer (E2 is one GEMM)
    for (uint32_t l = 0; l < cfg->layers && s == CS_OK; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(cfg, &h->off, l, &lo);
        const uint64_t w[3] = {lo.q_w, lo.k_w, lo.v_w}, b[3] = {lo.q_b, lo.k_b, lo.v_b};
        for (int i = 0; i < 3; ++i) {
            if (hipMemcpyAsync(h->d_wqkv + ((size_t)l * 3 + i) * H * H, h->d_params + w[i], H * H * sizeof(float),
                               hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                hipMemcpyAsync(h->d_bqkv + ((size_t)l * 3 + i) * H, h->d_params + b[i], H * sizeof(float),
                               hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                s = fail(CS_ERR_HIP, "QKV packing failed");
        }
    }
    if (s == CS_OK && cs_arch_alibi(cfg->arch)) {
        // JinaBert's `_get_alibi_head_slopes`: the geometric sequence from 2^(-8 / n) for the closest power of two n below the
        // head count, then every second slope of the doubled set — formed in double as the module forms them in Python floats
        const uint32_t nh = cfg->heads;
        std::vector<float> sl(2 * (size_t)nh);
        uint32_t closest = 1;
        while (closest * 2 <= nh) closest *= 2;
        auto slope = [](uint32_t n, uint32_t i) { const double start = std::exp2(-8.0 / (double)n); return start * std::pow(start, (double)i); };
        for (uint32_t i = 0; i < nh; ++i) sl[i] = (float)(i < closest ? slope(closest, i) : slope(2 * closest, 2 * (i - closest)));
        for (uint32_t i = 0; i < nh; ++i) sl[nh + i] = sl[i] * 1.4426950408889634f;
        if (hipMalloc(&h->d_alibi, sl.size() * sizeof(float)) != hipSuccess) return cleanup(fail(CS_ERR_OOM, "hipMalloc(parameters) failed"));
        if (hipMemcpy(h->d_alibi, sl.data(), sl.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            s = fail(CS_ERR_HIP, "ALiBi slope upload failed");
    }
    if (s == CS_OK && cs_arch_gated(cfg->arch)) {
        const size_t I = cfg->intermediate, half = H / cfg->heads / 2;
        if (hipMalloc(&h->d_bup, (size_t)cfg->layers * 2 * I * sizeof(float)) != hipSuccess ||
            (cfg->arch == CS_ARCH_NOMIC && hipMalloc(&h->d_rope, (size_t)cfg->max_position * half * sizeof(float2)) != hipSuccess))
            return cleanup(fail(CS_ERR_OOM, "hipMalloc(parameters) failed"));
        for (uint32_t l = 0; l < cfg->layers && s == CS_OK; ++l) {
            cs_bert_layer_offsets lo;
            cs_bert_layer_layout(cfg, &h->off, l, &lo);
            // value and gate interleaved in groups of 16 columns, the order of the packed weight (GW_OUT_SWIGLU, encoder.hpp)
            float* bl = h->d_bup + (size_t)l * 2 * I;
            if (hipMemcpy2DAsync(bl, 32 * sizeof(float), h->d_params + lo.up_b, 16 * sizeof(float), 16 * sizeof(float), I / 16,
                                 hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                hipMemcpy2DAsync(bl + 16, 32 * sizeof(float), h->d_params + lo.gate_b, 16 * sizeof(float), 16 * sizeof(float), I / 16,
                                 hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                s = fail(CS_ERR_HIP, "feed-forward bias packing failed");
        }
        // the module's cos / sin cache, formed as it forms it: inv_freq_i = 1 / base^(2i / d_h) and pos * inv_freq_i in f32
        std::vector<float2> rope(cfg->arch == CS_ARCH_NOMIC ? (size_t)cfg->max_position * half : 0);
        const float dh = (float)(2 * half);
        for (size_t i = 0; i < half && !rope.empty(); ++i) {
            const float inv_freq = 1.0f / powf(cfg->rotary_base, (float)(2 * i) / dh);
            for (size_t p = 0; p < cfg->max_position; ++p) {
                const float ang = (float)p * inv_freq;
                rope[p * half + i] = make_float2(cosf(ang), sinf(ang));
            }
        }
        if (s == CS_OK && !rope.empty() && hipMemcpy(h->d_rope, rope.data(), rope.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
            s = fail(CS_ERR_HIP, "rotary table upload failed");
    }
    // split-f16 copies of the four dense weights of every layer (split_f16.hpp)
    if (s == CS_OK) {
        const SplitLayer sl = split_layer(*cfg);
        const size_t I = cfg->intermediate;
        if (hipMalloc(&h->d_wsplit, (size_t)cfg->layers * sl.total * sizeof(_Float16)) != hipSuccess ||
            hipMalloc(&h->d_flag, sizeof(uint32_t)) != hipSuccess)
            return cleanup(fail(CS_ERR_OOM, "hipMalloc(split weights) failed"));
        if (hipMemsetAsync(h->d_flag, 0, sizeof(uint32_t), h->stream) != hipSuccess) s = fail(CS_ERR_HIP, "memset failed");
        float* d_updown = nullptr;  // CS_ARCH_NOMIC: one layer's fc11 / fc12 rows interleaved in groups of 16, [2I][H]
        if (cs_arch_gated(cfg->arch) && hipMalloc(&d_updown, 2 * I * H * sizeof(float)) != hipSuccess)
            return cleanup(fail(CS_ERR_OOM, "hipMalloc(split weights) failed"));
        for (uint32_t l = 0; l < cfg->layers && s == CS_OK; ++l) {
            cs_bert_layer_offsets lo;
            cs_bert_layer_layout(cfg, &h->off, l, &lo);
            _Float16* ws = h->d_wsplit + (size_t)l * sl.total;
            s = launch_split_rows(h->d_wqkv + (size_t)l * 3 * H * H, ws + sl.qkv, 3 * H, (uint32_t)H, h->d_flag, h->stream);
            if (s == CS_OK) s = launch_split_rows(h->d_params + lo.ao_w, ws + sl.ao, H, (uint32_t)H, h->d_flag, h->stream);
            if (s == CS_OK && cs_arch_gated(cfg->arch)) {
                // one [2I, H] weight: raw output columns 32 u .. 32 u + 15 = fc11's rows 16 u .., the next sixteen fc12's, so that
                // a value and its gate meet in one lane of the product's epilogue (GW_OUT_SWIGLU) and in one line of its output
                const size_t grp = 16 * H * sizeof(float);
                if (hipMemcpy2DAsync(d_updown, 2 * grp, h->d_params + lo.up_w, grp, grp, I / 16, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                    hipMemcpy2DAsync(d_updown + 16 * H, 2 * grp, h->d_params + lo.gate_w, grp, grp, I / 16, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                    s = fail(CS_ERR_HIP, "feed-forward weight packing failed");
                if (s == CS_OK) s = launch_split_rows(d_updown, ws + sl.up, 2 * I, (uint32_t)H, h->d_flag, h->stream);
            } else if (s == CS_OK) {
                s = launch_split_rows(h->d_params + lo.up_w, ws + sl.up, I, (uint32_t)H, h->d_flag, h->stream);
            }
            if (s == CS_OK) s = launch_split_rows(h->d_params + lo.down_w, ws + sl.down, H, (uint32_t)I, h->d_flag, h->stream);
        }
        if (d_updown) {
            (void)hipStreamSynchronize(h->stream);
            (void)hipFree(d_updown);
        }
        uint32_t wflag = 0;
        if (s == CS_OK && (hipMemcpyAsync(&wflag, h->d_flag, sizeof(wflag), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                           hipStreamSynchronize(h->stream) != hipSuccess))
            s = fail(CS_ERR_HIP, "parameter setup failed");
        if (const char* env = std::getenv("CS_ENCODER_GEMM"))
            h->gemm_mode = (std::strcmp(env, "f32") == 0) ? CS_GEMM_F32 : CS_GEMM_SPLIT_F16;
        if (wflag) h->gemm_mode = CS_GEMM_F32;  // a weight outside the f16 range: exact path only
        if (s == CS_OK && !wflag) {  // may the one-accumulator kernels scale w_hi by 2^11 in f16?
            bool fit = false;
            s = sh_weights_fit_wide(h->d_wsplit, (uint64_t)cfg->layers * sl.total, h->d_flag, &fit, h->stream);
            const char* e = std::getenv("CS_GEMM_WIDE");  // "0": never
            h->wide_ok = fit && !(e && e[0] == '0');
        }
        bool denorm_ok = false;  // the split format relies on exact f16-subnormal MFMA inputs
        if (s == CS_OK) s = sh_denorm_selftest(&denorm_ok, h->stream);
        if (s == CS_OK && !denorm_ok) { h->gemm_mode = CS_GEMM_F32; h->split_unavailable = true; }
        // the one-launch forward of short queries (small_forward.hip) reads the layers' pointers from a device table
        if (s == CS_OK && !cs_arch_gated(cfg->arch) && small_forward_supported((uint32_t)H, (uint32_t)I, cfg->heads, 1, 1)) {
            std::vector<SfLayer> tab(cfg->layers);
            for (uint32_t l = 0; l < cfg->layers; ++l) {
                cs_bert_layer_offsets lo;
                cs_bert_layer_layout(cfg, &h->off, l, &lo);
                const _Float16* wl = h->d_wsplit + (size_t)l * sl.total;
                const float* P = h->d_params;
                tab[l] = SfLayer{wl + sl.qkv, wl + sl.ao, wl + sl.up, wl + sl.down, h->d_bqkv + (size_t)l * 3 * H, P + lo.ao_b, P + lo.up_b,
                                 P + lo.down_b, P + lo.ao_ln_g, P + lo.ao_ln_b, P + lo.out_ln_g, P + lo.out_ln_b};
            }
            if (hipMalloc(&h->d_sf_layers, tab.size() * sizeof(SfLayer)) != hipSuccess || hipMalloc(&h->d_sf_sync, 16) != hipSuccess ||
                hipMemcpyAsync(h->d_sf_layers, tab.data(), tab.size() * sizeof(SfLayer), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess)
                s = fail(CS_ERR_OOM, "the one-launch forward's layer table could not be set up");
            if (s == CS_OK && std::getenv("CS_SMALL_FORWARD_DEBUG") && hipMalloc(&h->d_sf_dbg, (96 * 3 + 8) * sizeof(uint64_t)) != hipSuccess)
                h->d_sf_dbg = nullptr;
        }
        if (const char* env = std::getenv("CS_ENCODER_STREAMS")) {
            h->streams_forced = true;
            const int v = std::atoi(env);
            h->n_streams = v >= 4 ? 4 : (v >= 1 ? v : 1);
        }
        for (int i = 0; i + 2 < h->n_streams; ++i)
            if (hipStreamCreateWithFlags(&h->xstreams[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&h->xjoin[i], hipEventDisableTiming) != hipSuccess) {
                h->n_streams = 2;
                break;
            }
    }
    // dynamically quantised model: the s8 form of every Linear weight and its column metadata (gemm_q8.hip)
    if (s == CS_OK && wscale) {
        const size_t I = cfg->intermediate, cols = 5 * H + I;
        const Q8Layer ql = q8_layer((uint32_t)H, (uint32_t)I);
        float* d_ws = nullptr;
        uint32_t* d_bad = nullptr;
        if (I > 4 * H || I > 4096) s = fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: intermediate size above 4 x hidden or 4,096");
        if (s == CS_OK && (hipMalloc(&h->d_wq8, (size_t)cfg->layers * ql.total) != hipSuccess ||
                           hipMalloc(&h->d_cmeta, (size_t)cfg->layers * cols * sizeof(Q8ColMeta)) != hipSuccess ||
                           hipMalloc(&d_ws, (size_t)cfg->layers * cols * sizeof(float)) != hipSuccess ||
                           hipMalloc(&d_bad, sizeof(uint32_t)) != hipSuccess))
            s = fail(CS_ERR_OOM, "hipMalloc(quantised weights) failed");
        if (s == CS_OK && (hipMemcpyAsync(d_ws, wscale, (size_t)cfg->layers * cols * sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                           hipMemsetAsync(d_bad, 0, sizeof(uint32_t), h->stream) != hipSuccess))
            s = fail(CS_ERR_HIP, "quantised weight setup failed");
        for (uint32_t l = 0; l < cfg->layers && s == CS_OK; ++l) {
            cs_bert_layer_offsets lo;
            cs_bert_layer_layout(cfg, &h->off, l, &lo);
            int8_t* wq = h->d_wq8 + (size_t)l * ql.total;
            Q8ColMeta* cm = h->d_cmeta + (size_t)l * cols;
            const float* sc = d_ws + (size_t)l * cols;
            s = launch_q8_pack_weight(h->d_wqkv + (size_t)l * 3 * H * H, sc, h->d_bqkv + (size_t)l * 3 * H, (uint32_t)(3 * H), (uint32_t)H, wq + ql.qkv, cm, d_bad, h->stream);
            if (s == CS_OK) s = launch_q8_pack_weight(h->d_params + lo.ao_w, sc + 3 * H, h->d_params + lo.ao_b, (uint32_t)H, (uint32_t)H, wq + ql.ao, cm + 3 * H, d_bad, h->stream);
            if (s == CS_OK) s = launch_q8_pack_weight(h->d_params + lo.up_w, sc + 4 * H, h->d_params + lo.up_b, (uint32_t)I, (uint32_t)H, wq + ql.up, cm + 4 * H, d_bad, h->stream);
            if (s == CS_OK) s = launch_q8_pack_weight(h->d_params + lo.down_w, sc + 4 * H + I, h->d_params + lo.down_b, (uint32_t)H, (uint32_t)I, wq + ql.down, cm + 4 * H + I, d_bad, h->stream);
        }
        uint32_t bad = 0;
        if (s == CS_OK && (hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                           hipStreamSynchronize(h->stream) != hipSuccess))
            s = fail(CS_ERR_HIP, "quantised weight setup failed");
        if (d_ws) (void)hipFree(d_ws);
        if (d_bad) (void)hipFree(d_bad);
        if (s == CS_OK && bad)
            s = fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s", (bad & 2)
                         ? "a Linear weight is not an integer multiple of its column scale (not a quantised block)"
                         : "the integers of a weight column span more than 8 bits");
        if (s == CS_OK) {
            h->quantized = true;
            const char* env = std::getenv("CS_ENCODER_QUANT");  // "0": the f32 graph of the quantised weights
            // an explicit CS_ENCODER_GEMM is honoured for quantised models too: "f32" / "split" select the f32
            // graph of the dequantised weights on that arithmetic, exactly as CS_ENCODER_QUANT=0 does
            const bool explicit_gemm = std::getenv("CS_ENCODER_GEMM") != nullptr;
            if (!h->split_unavailable && !explicit_gemm && !(env && env[0] == '0')) h->gemm_mode = CS_GEMM_Q8_DYNAMIC;
        }
    }
    if (s == CS_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = fail(CS_ERR_HIP, "parameter setup failed");
    if (s != CS_OK) return cleanup(s);
    *out = h;
    return CS_OK;
}

int32_t cs_embedder_create(const cs_bert_config* cfg, const float* params, uint64_t seed,
                           int32_t device, cs_embedder** out) {
    return create_impl(cfg, params, seed, nullptr, device, out);
}

uint64_t cs_bert_quant_columns(const cs_bert_config* cfg) {
    return cfg ? 5 * (uint64_t)cfg->hidden + cfg->intermediate : 0;
}

int32_t cs_embedder_create_quantized(const cs_bert_config* cfg, const float* params, const float* wscale,
                                     uint64_t n_wscale, int32_t device, cs_embedder** out) {
    if (out) *out = nullptr;
    if (!cfg || !params || !wscale) return fail(CS_ERR_BAD_ARG, "cs_embedder_create_quantized: null argument");
    if (n_wscale != (uint64_t)cfg->layers * cs_bert_quant_columns(cfg))
        return fail(CS_ERR_BAD_ARG, "cs_embedder_create_quantized: %llu column scales given, %llu expected (layers x (5 hidden + intermediate))",
                    (unsigned long long)n_wscale, (unsigned long long)((uint64_t)cfg->layers * cs_bert_quant_columns(cfg)));
    return create_impl(cfg, params, 0, wscale, device, out);
}

void cs_embedder_destroy(cs_embedder* h) {
    if (!h) return;
    DeviceGuard g(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    for (int i = 0; i < 2; ++i) {
        if (h->xstreams[i]) { (void)hipStreamSynchronize(h->xstreams[i]); (void)hipStreamDestroy(h->xstreams[i]); }
        if (h->xjoin[i]) (void)hipEventDestroy(h->xjoin[i]);
    }
    free_workspace(h);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_wqkv) (void)hipFree(h->d_wqkv);
    if (h->d_bqkv) (void)hipFree(h->d_bqkv);
    if (h->d_bup) (void)hipFree(h->d_bup);
    if (h->d_rope) (void)hipFree(h->d_rope);
    if (h->d_alibi) (void)hipFree(h->d_alibi);
    if (h->d_wsplit) (void)hipFree(h->d_wsplit);
    if (h->d_flag) (void)hipFree(h->d_flag);
    if (h->d_sf_layers) (void)hipFree(h->d_sf_layers);
    if (h->d_sf_sync) (void)hipFree(h->d_sf_sync);
    if (h->d_sf_dbg) (void)hipFree(h->d_sf_dbg);
    if (h->d_sp_ws) (void)hipFree(h->d_sp_ws);
    if (h->d_wq8) (void)hipFree(h->d_wq8);
    if (h->d_cmeta) (void)hipFree(h->d_cmeta);
    for (hipEvent_t e : h->stage_ev) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

uint32_t cs_embedder_dim(const cs_embedder* h) { return h ? h->cfg.hidden : 0; }

int32_t cs_embedder_embed_ids(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n,
                              uint32_t seq_len, uint32_t batch, float* out,
                              const volatile int32_t* cancel) {
    return embed_ids_entry(h, ids, mask, n, seq_len, batch, out, false, cancel);
}

int32_t cs_embedder_embed_ids_device(cs_embedder* h, const int32_t* ids, const int32_t* mask,
                                     uint64_t n, uint32_t seq_len, uint32_t batch, float* d_out,
                                     const volatile int32_t* cancel) {
    return embed_ids_entry(h, ids, mask, n, seq_len, batch, d_out, true, cancel);
}

int32_t cs_embedder_embed_texts(cs_embedder* h, const cs_tokenizer* t, const char* utf8,
                                const uint64_t* offsets, uint64_t n, uint32_t batch, float* out,
                                const volatile int32_t* cancel) {
    return embed_texts_impl(h, t, utf8, offsets, n, batch, out, false, cancel);
}

int32_t cs_embedder_embed_texts_device(cs_embedder* h, const cs_tokenizer* t, const char* utf8,
                                       const uint64_t* offsets, uint64_t n, uint32_t batch, float* d_out,
                                       const volatile int32_t* cancel) {
    return embed_texts_impl(h, t, utf8, offsets, n, batch, d_out, true, cancel);
}

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

int32_t cs_embedder_last_hidden(cs_embedder* h, float* out, uint64_t n_tokens) {
    if (!h || !out) return fail(CS_ERR_BAD_ARG, "null argument");
    if (n_tokens > (uint64_t)h->last_B * h->last_L)
        return fail(CS_ERR_BAD_ARG, "only %u tokens in the last mini-batch", h->last_B * h->last_L);
    if (h->last_hidden_partial)
        return fail(CS_ERR_UNSUPPORTED, "the last forward computed its final layer for the CLS rows only (cls_tail.hip): "
                                        "set CS_ENCODER_CLS_TAIL=0 to read every token's last hidden state");
    DeviceGuard g(h->device);
    CS_HIP(hipStreamSynchronize(h->stream));
    CS_HIP(hipMemcpy(out, h->d_x, n_tokens * h->cfg.hidden * sizeof(float), hipMemcpyDeviceToHost));
    return CS_OK;
}

int32_t cs_embedder_profile_read(cs_embedder* h, double* forward_ms, uint64_t* forwards, int32_t reset) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (forward_ms) *forward_ms = h->forward_ms;
    if (forwards) *forwards = h->forwards;
    if (reset) { h->forward_ms = 0.0; h->forwards = 0; }
    return CS_OK;
}

int32_t cs_embedder_profile_stages(cs_embedder* h, int32_t enable) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    h->stage_profile = enable != 0;
    return CS_OK;
}

int32_t cs_embedder_profile_stages_read(cs_embedder* h, double* us_per_stage, uint64_t* forwards, int32_t reset) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (us_per_stage)
        for (int i = 0; i < CS_ENCODER_STAGES; ++i) us_per_stage[i] = h->stage_us[i];
    if (forwards) *forwards = h->stage_forwards;
    if (reset) {
        for (int i = 0; i < CS_ENCODER_STAGES; ++i) h->stage_us[i] = 0.0;
        h->stage_forwards = 0;
    }
    return CS_OK;
}

int32_t cs_embedder_set_gemm_mode(cs_embedder* h, int32_t mode) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (mode != CS_GEMM_F32 && mode != CS_GEMM_SPLIT_F16 && mode != CS_GEMM_Q8_DYNAMIC) return fail(CS_ERR_BAD_ARG, "unknown gemm mode %d", mode);
    if (mode == CS_GEMM_Q8_DYNAMIC && !h->quantized)
        return fail(CS_ERR_UNSUPPORTED, "dynamic-quantisation mode needs a quantised model (cs_embedder_create_quantized / a *Q model directory)");
    if (mode != CS_GEMM_F32 && h->split_unavailable)
        return fail(CS_ERR_UNSUPPORTED, "split-f16 mode needs exact f16-subnormal MFMA inputs, which this device/mode lacks");
    h->gemm_mode = mode;
    return CS_OK;
}

int32_t cs_embedder_gemm_mode(const cs_embedder* h) { return h ? h->gemm_mode : -1; }

int32_t cs_embedder_debug_counters(cs_embedder* h, uint64_t* split_forwards, uint64_t* f32_forwards,
                                   uint64_t* range_fallbacks) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (split_forwards) *split_forwards = h->split_forwards;
    if (f32_forwards) *f32_forwards = h->f32_forwards;
    if (range_fallbacks) *range_fallbacks = h->range_fallbacks;
    return CS_OK;
}

int32_t cs_embedder_small_forward_counters(cs_embedder* h, uint64_t* forwards, uint64_t* fallbacks) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (forwards) *forwards = h->sf_forwards;
    if (fallbacks) *fallbacks = h->sf_fallbacks;
    return CS_OK;
}

int32_t cs_debug_gemm(int32_t device, int32_t mode, int32_t epilogue, const float* A, const float* W,
                      const float* bias, const float* resid, float* C, uint32_t M, uint32_t N, uint32_t K,
                      uint32_t* range_flag) {
    if (!A || !W || !bias || !C || ((epilogue == 2 || epilogue == 3) && !resid)) return fail(CS_ERR_BAD_ARG, "null buffer");
    const bool wide = mode == 2;  // diagnostics only: the 128 x 384 one-accumulator kernel whatever M is
    if (wide) mode = CS_GEMM_SPLIT_F16;
    // epilogue 3 (wide only, N = 384): + resid, LayerNorm with gamma = bias + 1, beta = -bias, eps 1e-12; C receives
    // the f32 output re-assembled from the SPLIT output (hi + lo / 2048), so both stores are exercised
    if (epilogue == 3 && !(wide && N == 384)) return fail(CS_ERR_UNSUPPORTED, "epilogue 3 needs mode 2 and N = 384");
    if (epilogue < 0 || epilogue > 4 || (mode != CS_GEMM_F32 && mode != CS_GEMM_SPLIT_F16))
        return fail(CS_ERR_BAD_ARG, "unknown epilogue/mode");
    if (wide && !gemm_wide_supported(N, K)) return fail(CS_ERR_UNSUPPORTED, "wide kernel needs N %% 384 == 0");
    if (M == 0 || N % 128 || K % 32 || K == 0) return fail(CS_ERR_UNSUPPORTED, "cs_debug_gemm needs M > 0, N %% 128 == 0, K %% 32 == 0");
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CS_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    const size_t a_n = (size_t)M * K, w_n = (size_t)N * K, c_n = (size_t)M * N;
    float *dA = nullptr, *dW = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
    _Float16 *sA = nullptr, *sW = nullptr, *sC = nullptr;
    uint32_t* dF = nullptr;
    int32_t st = CS_OK;
    auto run = [&]() -> int32_t {
        CS_HIP(hipMalloc(&dA, a_n * 4)); CS_HIP(hipMalloc(&dW, w_n * 4)); CS_HIP(hipMalloc(&dB, (size_t)N * 4));
        CS_HIP(hipMalloc(&dC, c_n * 4)); CS_HIP(hipMalloc(&dF, 4));
        CS_HIP(hipMemcpy(dA, A, a_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dW, W, w_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemset(dF, 0, 4));
        if (epilogue >= 2) {
            CS_HIP(hipMalloc(&dR, c_n * 4));
            CS_HIP(hipMemcpy(dR, resid, c_n * 4, hipMemcpyHostToDevice));
        }
        if (mode == CS_GEMM_F32) {
            CS_TRY(launch_gemm(epilogue, dA, dW, dB, dR, dC, M, N, K, nullptr));
        } else {
            CS_HIP(hipMalloc(&sA, a_n * 4)); CS_HIP(hipMalloc(&sW, w_n * 4));
            CS_TRY(launch_split_rows(dA, sA, M, K, dF, nullptr));
            CS_TRY(launch_split_rows(dW, sW, N, K, dF, nullptr));
            auto run_gemm = [&](int e, const _Float16* a_, const _Float16* w_, const float* b_, const float* r_, float* c_, _Float16* cs_,
                                uint32_t m_, uint32_t n_, uint32_t k_, uint32_t* f_, hipStream_t st_) {
                return wide ? launch_gemm_wide(e, a_, w_, b_, r_, c_, cs_, m_, n_, k_, f_, st_, 0) : launch_gemm_split(e, a_, w_, b_, r_, c_, cs_, m_, n_, k_, f_, st_);
            };
            if (epilogue == 4) {  // LayerNorm epilogue, residual given (and overwritten) in split form, no f32 output
                std::vector<float> gam(N), bet(N);
                for (uint32_t n = 0; n < N; ++n) { gam[n] = bias[n] + 1.0f; bet[n] = -bias[n]; }
                float *dG = nullptr, *dBe = nullptr;
                CS_HIP(hipMalloc(&dG, (size_t)N * 4)); CS_HIP(hipMalloc(&dBe, (size_t)N * 4));
                CS_HIP(hipMemcpy(dG, gam.data(), (size_t)N * 4, hipMemcpyHostToDevice));
                CS_HIP(hipMemcpy(dBe, bet.data(), (size_t)N * 4, hipMemcpyHostToDevice));
                CS_HIP(hipMalloc(&sC, c_n * 4));
                CS_TRY(launch_split_rows(dR, sC, M, N, dF, nullptr));
                const int32_t st4 = launch_gemm_wide_ln(sA, sW, dB, nullptr, dG, dBe, 1e-12f, nullptr, sC, M, K, dF, nullptr, sC);
                CS_HIP(hipDeviceSynchronize());
                (void)hipFree(dG); (void)hipFree(dBe);
                CS_TRY(st4);
            } else if (epilogue == 3) {
                std::vector<float> gam(N), bet(N);
                for (uint32_t n = 0; n < N; ++n) { gam[n] = bias[n] + 1.0f; bet[n] = -bias[n]; }
                float *dG = nullptr, *dBe = nullptr;
                CS_HIP(hipMalloc(&dG, (size_t)N * 4)); CS_HIP(hipMalloc(&dBe, (size_t)N * 4));
                CS_HIP(hipMemcpy(dG, gam.data(), (size_t)N * 4, hipMemcpyHostToDevice));
                CS_HIP(hipMemcpy(dBe, bet.data(), (size_t)N * 4, hipMemcpyHostToDevice));
                CS_HIP(hipMalloc(&sC, c_n * 4));
                const int32_t st3 = launch_gemm_wide_ln(sA, sW, dB, dR, dG, dBe, 1e-12f, dR, sC, M, K, dF, nullptr);  // in place over resid
                CS_HIP(hipDeviceSynchronize());
                std::vector<float> f32out(c_n);
                CS_HIP(hipMemcpy(f32out.data(), dR, c_n * 4, hipMemcpyDeviceToHost));
                (void)hipFree(dG); (void)hipFree(dBe);
                CS_TRY(st3);
                // the two outputs must describe the same values: checked here, the split one is what C receives below
                std::vector<_Float16> hs(c_n * 2);
                CS_HIP(hipMemcpy(hs.data(), sC, c_n * 4, hipMemcpyDeviceToHost));
                for (size_t m = 0; m < M; ++m)
                    for (size_t n = 0; n < N; ++n) {
                        const _Float16* line = hs.data() + (m * (N / 32) + n / 32) * 64;
                        const float v = (float)line[n % 32] + (float)line[32 + n % 32] * (1.0f / 2048.0f);
                        if (!(fabsf(v - f32out[m * N + n]) <= 1e-6f * fmaxf(1.0f, fabsf(v))))
                            return fail(CS_ERR_HIP, "LayerNorm epilogue: f32 and split outputs disagree at (%zu, %zu): %g vs %g", m, n,
                                        (double)f32out[m * N + n], (double)v);
                    }
            } else if (epilogue == 1) {  // the GELU epilogue writes split form: read it back through hi + lo / 2048
                CS_HIP(hipMalloc(&sC, c_n * 4));
                CS_TRY(run_gemm(SH_OUT_SPLIT_GELU, sA, sW, dB, nullptr, nullptr, sC, M, N, K, dF, nullptr));
            } else {
                CS_TRY(run_gemm(epilogue == 2 ? SH_OUT_F32_RESID : SH_OUT_F32, sA, sW, dB, dR, dC, nullptr, M, N, K, dF, nullptr));
            }
        }
        CS_HIP(hipDeviceSynchronize());
        if (sC) {
            std::vector<_Float16> hs(c_n * 2);
            CS_HIP(hipMemcpy(hs.data(), sC, c_n * 4, hipMemcpyDeviceToHost));
            const size_t nch = N / 32;
            for (size_t m = 0; m < M; ++m)
                for (size_t n = 0; n < N; ++n) {
                    const _Float16* line = hs.data() + (m * nch + n / 32) * 64;
                    C[m * N + n] = (float)line[n % 32] + (float)line[32 + n % 32] * (1.0f / 2048.0f);
                }
        } else {
            CS_HIP(hipMemcpy(C, dC, c_n * 4, hipMemcpyDeviceToHost));
        }
        if (range_flag) CS_HIP(hipMemcpy(range_flag, dF, 4, hipMemcpyDeviceToHost));
        return CS_OK;
    };
    st = run();
    for (void* p : {(void*)dA, (void*)dW, (void*)dB, (void*)dR, (void*)dC, (void*)sA, (void*)sW, (void*)sC, (void*)dF})
        if (p) (void)hipFree(p);
    return st;
}

int32_t cs_debug_gemm_q8(int32_t device, int32_t epilogue, int32_t a_split, const float* A, const float* W,
                         const float* wscale, const float* bias, const float* resid, float* C, uint32_t M, uint32_t N,
                         uint32_t K, uint8_t* xq_out, float* xparams, int32_t* acc_out) {
    if (!A || !W || !wscale || !bias || !C || (epilogue == 2 && !resid)) return fail(CS_ERR_BAD_ARG, "null buffer");
    if (epilogue != 0 && epilogue != 1 && epilogue != 2 && epilogue != 4 && epilogue != 5) return fail(CS_ERR_BAD_ARG, "unknown epilogue %d", epilogue);
    if (M == 0 || N % 128 || K % 128 || K == 0) return fail(CS_ERR_UNSUPPORTED, "cs_debug_gemm_q8 needs M > 0, N %% 128 == 0, K %% 128 == 0");
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CS_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    const size_t a_n = (size_t)M * K, w_n = (size_t)N * K, c_n = (size_t)M * N;
    float *dA = nullptr, *dW = nullptr, *dS = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
    _Float16 *sA = nullptr, *sC = nullptr;
    int8_t *dXq = nullptr, *dWq = nullptr;
    Q8RowMeta* dRm = nullptr;
    Q8ColMeta* dCm = nullptr;
    uint32_t *dF = nullptr, *dRange = nullptr;
    int32_t* dAcc = nullptr;
    auto run = [&]() -> int32_t {
        CS_HIP(hipMalloc(&dA, a_n * 4)); CS_HIP(hipMalloc(&dW, w_n * 4)); CS_HIP(hipMalloc(&dS, (size_t)N * 4));
        CS_HIP(hipMalloc(&dB, (size_t)N * 4)); CS_HIP(hipMalloc(&dC, c_n * 4)); CS_HIP(hipMalloc(&dF, 16));
        CS_HIP(hipMalloc(&dXq, a_n)); CS_HIP(hipMalloc(&dWq, w_n)); CS_HIP(hipMalloc(&dRm, (size_t)M * sizeof(Q8RowMeta)));
        CS_HIP(hipMalloc(&dCm, (size_t)N * sizeof(Q8ColMeta))); CS_HIP(hipMalloc(&dRange, Q8_RANGE_WORDS * 4));
        CS_HIP(hipMemcpy(dA, A, a_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dW, W, w_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dS, wscale, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemset(dF, 0, 16));
        CS_HIP(hipMemset(dRange, 0, Q8_RANGE_WORDS * 4));
        if (epilogue == 2) {
            CS_HIP(hipMalloc(&dR, c_n * 4));
            CS_HIP(hipMemcpy(dR, resid, c_n * 4, hipMemcpyHostToDevice));
        }
        if (acc_out && epilogue != 5) CS_HIP(hipMalloc(&dAcc, c_n * 4));
        CS_TRY(launch_q8_pack_weight(dW, dS, dB, N, K, dWq, dCm, dF + 1, nullptr));
        if (a_split & 1) {
            CS_HIP(hipMalloc(&sA, a_n * 4));
            CS_TRY(launch_split_rows(dA, sA, M, K, dF, nullptr));
            CS_TRY(launch_q8_quantize(Q8_SRC_SPLIT, sA, M, K, dRange, nullptr, dXq, dRm, nullptr));
        } else {
            CS_TRY(launch_q8_quantize(Q8_SRC_F32, dA, M, K, dRange, nullptr, dXq, dRm, nullptr));
        }
        if (epilogue == 5) {  // GELU -> re-quantised (the two-pass FFN-up): C = the uint8 output, xparams[2..3] = its scale / zero point
            int8_t* dOut = nullptr;
            Q8RowMeta* dRm2 = nullptr;
            uint32_t* dRange2 = nullptr;
            CS_HIP(hipMalloc(&dOut, c_n)); CS_HIP(hipMalloc(&dRm2, (size_t)M * sizeof(Q8RowMeta))); CS_HIP(hipMalloc(&dRange2, Q8_RANGE_WORDS * 4));
            CS_HIP(hipMemset(dRange2, 0, Q8_RANGE_WORDS * 4));
            int32_t st5 = (a_split & 8) ? launch_gemm_q8_gelu_requant_from_source(dA, dRange, dWq, dCm, dB, M, N, K, dRange2, dOut, dRm2, nullptr)
                                        : launch_gemm_q8_gelu_requant(dXq, dRm, dWq, dCm, dB, M, N, K, dRange2, dOut, dRm2, nullptr);
            if (st5 == CS_OK && hipDeviceSynchronize() != hipSuccess) st5 = fail(CS_ERR_HIP, "requant GEMM failed");
            std::vector<int8_t> ho(c_n);
            std::vector<Q8RowMeta> hr(M);
            if (st5 == CS_OK && (hipMemcpy(ho.data(), dOut, c_n, hipMemcpyDeviceToHost) != hipSuccess ||
                                 hipMemcpy(hr.data(), dRm2, (size_t)M * sizeof(Q8RowMeta), hipMemcpyDeviceToHost) != hipSuccess))
                st5 = fail(CS_ERR_HIP, "requant GEMM read-back failed");
            (void)hipFree(dOut); (void)hipFree(dRm2); (void)hipFree(dRange2);
            CS_TRY(st5);
            for (size_t i = 0; i < c_n; ++i) C[i] = (float)((int)ho[i] + 128);
            if (acc_out) for (size_t m = 0; m < M; ++m) acc_out[m] = hr[m].rowsum + 128 * (int32_t)N;  // row sums of the uint8 output
            if (xparams) { xparams[2] = hr[0].xs; xparams[3] = (float)(hr[0].za + 128); }
            uint32_t flags5[2] = {0, 0};
            CS_HIP(hipMemcpy(flags5, dF, 8, hipMemcpyDeviceToHost));
            if (flags5[1]) return fail(CS_ERR_BAD_ARG, "cs_debug_gemm_q8: W is not a quantised matrix for these column scales (flag %u)", flags5[1]);
            if (xq_out || xparams) {
                std::vector<int8_t> hq(a_n);
                Q8RowMeta rm0;
                CS_HIP(hipMemcpy(hq.data(), dXq, a_n, hipMemcpyDeviceToHost));
                CS_HIP(hipMemcpy(&rm0, dRm, sizeof(rm0), hipMemcpyDeviceToHost));
                if (xq_out) for (size_t i = 0; i < a_n; ++i) xq_out[i] = (uint8_t)((int)hq[i] + 128);
                if (xparams) { xparams[0] = rm0.xs; xparams[1] = (float)(rm0.za + 128); }
            }
            return CS_OK;
        }
        const int epi = epilogue == 0 ? SH_OUT_F32 : epilogue == 1 ? SH_OUT_SPLIT_GELU : epilogue == 2 ? SH_OUT_F32_RESID : SH_OUT_SPLIT;
        if (epi == SH_OUT_SPLIT_GELU || epi == SH_OUT_SPLIT) CS_HIP(hipMalloc(&sC, c_n * 4));
        if (a_split & 4) {  // the few-rows kernel: its "pairs" are the one (lo, hi) in the slot (the words are the floats' bits)
            if (dAcc) CS_HIP(hipMemset(dAcc, 0, c_n * 4));
            CS_TRY(launch_gemm_q8_skinny(epi, (a_split & 1) ? Q8_SRC_SPLIT : Q8_SRC_F32, (a_split & 1) ? (const void*)sA : (const void*)dA,
                                         reinterpret_cast<const float*>(dRange), 1, dWq, dCm, dR, dC, sC, M, N, K, dF, nullptr, nullptr, nullptr));
        } else if (a_split & 8) {  // the products that quantise their own rows on the way in (row-block kernel; acc is not reported)
            if (dAcc) CS_HIP(hipMemset(dAcc, 0, c_n * 4));
            CS_TRY(launch_gemm_q8_from_source(epi, (a_split & 1) ? Q8_SRC_SPLIT : Q8_SRC_F32, (a_split & 1) ? (const void*)sA : (const void*)dA, dRange,
                                              dWq, dCm, dB, dR, dC, sC, M, N, K, dF, nullptr));
        } else
        CS_TRY(launch_gemm_q8(epi, dXq, dRm, dWq, dCm, dB, dR, dC, sC, M, N, K, dF, nullptr, dAcc));
        CS_HIP(hipDeviceSynchronize());
        uint32_t flags[2] = {0, 0};
        CS_HIP(hipMemcpy(flags, dF, 8, hipMemcpyDeviceToHost));
        if (flags[1]) return fail(CS_ERR_BAD_ARG, "cs_debug_gemm_q8: W is not a quantised matrix for these column scales (flag %u)", flags[1]);
        if (sC) {
            std::vector<_Float16> hs(c_n * 2);
            CS_HIP(hipMemcpy(hs.data(), sC, c_n * 4, hipMemcpyDeviceToHost));
            const size_t nch = N / 32;
            for (size_t m = 0; m < M; ++m)
                for (size_t n = 0; n < N; ++n) {
                    const _Float16* line = hs.data() + (m * nch + n / 32) * 64;
                    C[m * N + n] = (float)line[n % 32] + (float)line[32 + n % 32] * (1.0f / 2048.0f);
                }
        } else {
            CS_HIP(hipMemcpy(C, dC, c_n * 4, hipMemcpyDeviceToHost));
        }
        if (acc_out) CS_HIP(hipMemcpy(acc_out, dAcc, c_n * 4, hipMemcpyDeviceToHost));
        if (xq_out || xparams) {
            std::vector<int8_t> hq(a_n);
            Q8RowMeta rm0;
            CS_HIP(hipMemcpy(hq.data(), dXq, a_n, hipMemcpyDeviceToHost));
            CS_HIP(hipMemcpy(&rm0, dRm, sizeof(rm0), hipMemcpyDeviceToHost));
            if (xq_out) for (size_t i = 0; i < a_n; ++i) xq_out[i] = (uint8_t)((int)hq[i] + 128);
            if (xparams) { xparams[0] = rm0.xs; xparams[1] = (float)(rm0.za + 128); }
        }
        return CS_OK;
    };
    const int32_t st = run();
    for (void* p : {(void*)dA, (void*)dW, (void*)dS, (void*)dB, (void*)dR, (void*)dC, (void*)sA, (void*)sC, (void*)dXq, (void*)dWq,
                    (void*)dRm, (void*)dCm, (void*)dF, (void*)dRange, (void*)dAcc})
        if (p) (void)hipFree(p);
    return st;
}

// Diagnostics: the row-block products over a tensor of SEVERAL quantisation units (queued calls in one device batch):
// row_slot [M] as launch_q8_quantize takes it.  epilogue 4 (f32 source -> split store), 2 (split source, + residual) or
// 5 (FFN-up: GELU, quantised again per unit).  row_params [M][4] = per row (x_scale, x_zero_point, out_scale,
// out_zero_point) — the last two only for epilogue 5, where rowsums [M] receives each output row's sum of uint8 values.
int32_t cs_debug_gemm_q8_units(int32_t device, int32_t epilogue, const float* A, const float* W, const float* wscale,
                               const float* bias, const float* resid, float* C, uint32_t M, uint32_t N, uint32_t K,
                               const uint32_t* row_slot, uint32_t units, float* row_params, int32_t* rowsums) {
    if (!A || !W || !wscale || !bias || !C || !row_slot || (epilogue == 2 && !resid)) return fail(CS_ERR_BAD_ARG, "null buffer");
    if (epilogue != 2 && epilogue != 4 && epilogue != 5) return fail(CS_ERR_BAD_ARG, "unknown epilogue %d", epilogue);
    if (M == 0 || units == 0 || N % 128 || !q8_rows_from_source(M, K))
        return fail(CS_ERR_UNSUPPORTED, "cs_debug_gemm_q8_units: M=%u N=%u K=%u is not a row-block product", M, N, K);
    for (uint32_t m = 0; m < M; ++m)
        if ((row_slot[m] & 0x7fffffffu) >= units || (m && (row_slot[m] & 0x7fffffffu) < (row_slot[m - 1] & 0x7fffffffu)))
            return fail(CS_ERR_BAD_ARG, "row_slot[%u]: units must be consecutive runs of rows, in order", m);
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CS_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    const size_t a_n = (size_t)M * K, w_n = (size_t)N * K, c_n = (size_t)M * N;
    float *dA = nullptr, *dW = nullptr, *dS = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
    _Float16 *sA = nullptr, *sC = nullptr;
    int8_t *dXq = nullptr, *dWq = nullptr, *dOut = nullptr;
    Q8RowMeta *dRm = nullptr, *dRm2 = nullptr;
    Q8ColMeta* dCm = nullptr;
    uint32_t *dF = nullptr, *dRange = nullptr, *dRange2 = nullptr, *dSlot = nullptr;
    auto run = [&]() -> int32_t {
        const size_t rbytes = (size_t)units * Q8_RANGE_WORDS * 4;
        CS_HIP(hipMalloc(&dA, a_n * 4)); CS_HIP(hipMalloc(&dW, w_n * 4)); CS_HIP(hipMalloc(&dS, (size_t)N * 4));
        CS_HIP(hipMalloc(&dB, (size_t)N * 4)); CS_HIP(hipMalloc(&dC, c_n * 4)); CS_HIP(hipMalloc(&dF, 16));
        CS_HIP(hipMalloc(&dXq, a_n)); CS_HIP(hipMalloc(&dWq, w_n)); CS_HIP(hipMalloc(&dRm, (size_t)M * sizeof(Q8RowMeta)));
        CS_HIP(hipMalloc(&dCm, (size_t)N * sizeof(Q8ColMeta))); CS_HIP(hipMalloc(&dRange, rbytes)); CS_HIP(hipMalloc(&dRange2, rbytes));
        CS_HIP(hipMalloc(&dSlot, (size_t)M * 4));
        CS_HIP(hipMemcpy(dA, A, a_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dW, W, w_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dS, wscale, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dSlot, row_slot, (size_t)M * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemset(dF, 0, 16));
        CS_HIP(hipMemset(dRange, 0, rbytes));
        CS_HIP(hipMemset(dRange2, 0, rbytes));
        CS_TRY(launch_q8_pack_weight(dW, dS, dB, N, K, dWq, dCm, dF + 1, nullptr));
        // the units' ranges by a pass over the tensor (the quantised rows this also writes only serve row_params)
        if (epilogue == 2) {
            CS_HIP(hipMalloc(&sA, a_n * 4));
            CS_HIP(hipMalloc(&dR, c_n * 4));
            CS_HIP(hipMemcpy(dR, resid, c_n * 4, hipMemcpyHostToDevice));
            CS_TRY(launch_split_rows(dA, sA, M, K, dF, nullptr));
            CS_TRY(launch_q8_quantize(Q8_SRC_SPLIT, sA, M, K, dRange, dSlot, dXq, dRm, nullptr));
            CS_TRY(launch_gemm_q8_from_source(SH_OUT_F32_RESID, Q8_SRC_SPLIT, sA, dRange, dWq, dCm, dB, dR, dC, nullptr, M, N, K, dF, nullptr, dSlot));
        } else {
            CS_TRY(launch_q8_quantize(Q8_SRC_F32, dA, M, K, dRange, dSlot, dXq, dRm, nullptr));
            if (epilogue == 4) {
                CS_HIP(hipMalloc(&sC, c_n * 4));
                CS_TRY(launch_gemm_q8_from_source(SH_OUT_SPLIT, Q8_SRC_F32, dA, dRange, dWq, dCm, dB, nullptr, nullptr, sC, M, N, K, dF, nullptr, dSlot));
            } else {
                CS_HIP(hipMalloc(&dOut, c_n));
                CS_HIP(hipMalloc(&dRm2, (size_t)M * sizeof(Q8RowMeta)));
                CS_TRY(launch_gemm_q8_gelu_requant_from_source(dA, dRange, dWq, dCm, dB, M, N, K, dRange2, dOut, dRm2, nullptr, dSlot));
            }
        }
        CS_HIP(hipDeviceSynchronize());
        uint32_t flags[2] = {0, 0};
        CS_HIP(hipMemcpy(flags, dF, 8, hipMemcpyDeviceToHost));
        if (flags[1]) return fail(CS_ERR_BAD_ARG, "cs_debug_gemm_q8_units: W is not a quantised matrix for these column scales (flag %u)", flags[1]);
        std::vector<Q8RowMeta> hr(M), hr2;
        CS_HIP(hipMemcpy(hr.data(), dRm, (size_t)M * sizeof(Q8RowMeta), hipMemcpyDeviceToHost));
        if (epilogue == 5) {
            std::vector<int8_t> ho(c_n);
            hr2.resize(M);
            CS_HIP(hipMemcpy(ho.data(), dOut, c_n, hipMemcpyDeviceToHost));
            CS_HIP(hipMemcpy(hr2.data(), dRm2, (size_t)M * sizeof(Q8RowMeta), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < c_n; ++i) C[i] = (float)((int)ho[i] + 128);
            if (rowsums) for (size_t m = 0; m < M; ++m) rowsums[m] = hr2[m].rowsum + 128 * (int32_t)N;
        } else if (sC) {
            std::vector<_Float16> hs(c_n * 2);
            CS_HIP(hipMemcpy(hs.data(), sC, c_n * 4, hipMemcpyDeviceToHost));
            const size_t nch = N / 32;
            for (size_t m = 0; m < M; ++m)
                for (size_t n = 0; n < N; ++n) {
                    const _Float16* line = hs.data() + (m * nch + n / 32) * 64;
                    C[m * N + n] = (float)line[n % 32] + (float)line[32 + n % 32] * (1.0f / 2048.0f);
                }
        } else {
            CS_HIP(hipMemcpy(C, dC, c_n * 4, hipMemcpyDeviceToHost));
        }
        if (row_params)
            for (size_t m = 0; m < M; ++m) {
                row_params[4 * m] = hr[m].xs;
                row_params[4 * m + 1] = (float)(hr[m].za + 128);
                row_params[4 * m + 2] = epilogue == 5 ? hr2[m].xs : 0.0f;
                row_params[4 * m + 3] = epilogue == 5 ? (float)(hr2[m].za + 128) : 0.0f;
            }
        return CS_OK;
    };
    const int32_t st = run();
    for (void* p : {(void*)dA, (void*)dW, (void*)dS, (void*)dB, (void*)dR, (void*)dC, (void*)sA, (void*)sC, (void*)dXq, (void*)dWq,
                    (void*)dOut, (void*)dRm, (void*)dRm2, (void*)dCm, (void*)dF, (void*)dRange, (void*)dRange2, (void*)dSlot})
        if (p) (void)hipFree(p);
    return st;
}

// Diagnostics: device time of one dense layer on synthetic operands already in HBM (no PCIe, no allocation inside the
// timed region).  mode as cs_debug_gemm (0 f32 MFMA, 1 split-f16 128 x 128 / skinny kernels, 2 split-f16 wide kernel);
// epilogue 0 f32 store, 1 GELU -> split store, 2 + residual, 3 LayerNorm-fused (mode 2, N = 384), 4 bias -> split store
// (the QKV projection).  `ablation` (mode 2, epilogue 4 only): 1 no LDS-DMA, 2 no MFMA, 3 DMAs issued at the step start.
int32_t cs_debug_gemm_time(int32_t device, int32_t mode, int32_t epilogue, uint32_t M, uint32_t N, uint32_t K,
                           uint32_t iters, int32_t ablation, double* ms_per_launch) {
    if (!ms_per_launch || iters == 0 || M == 0 || N % 128 || K % 32 || K == 0) return fail(CS_ERR_BAD_ARG, "bad arguments");
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CS_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    const size_t a_n = (size_t)M * K, w_n = (size_t)N * K, c_n = (size_t)M * N;
    float *dA = nullptr, *dW = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
    _Float16 *sA = nullptr, *sW = nullptr, *sC = nullptr;
    uint32_t* dF = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto run = [&]() -> int32_t {
        CS_HIP(hipMalloc(&dA, a_n * 4)); CS_HIP(hipMalloc(&dW, w_n * 4)); CS_HIP(hipMalloc(&dB, (size_t)N * 4));
        CS_HIP(hipMalloc(&dC, c_n * 4)); CS_HIP(hipMalloc(&dR, c_n * 4)); CS_HIP(hipMalloc(&dF, 4));
        CS_HIP(hipMalloc(&sA, a_n * 4)); CS_HIP(hipMalloc(&sW, w_n * 4)); CS_HIP(hipMalloc(&sC, c_n * 4));
        // operands from the counter-based generator: unit-scale activations, weights / 20
        CS_TRY(launch_synth_fill(dA, M, K, 11, 0, nullptr));
        CS_TRY(launch_synth_fill(dW, N, K, 12, 0, nullptr));
        CS_TRY(launch_synth_fill(dR, M, N, 13, 0, nullptr));
        if (std::getenv("CS_DEBUG_GEMM_ZERO")) {  // all-zero operands: what the clock (DVFS), not the schedule, is worth
            CS_HIP(hipMemset(dA, 0, a_n * 4)); CS_HIP(hipMemset(dW, 0, w_n * 4)); CS_HIP(hipMemset(dR, 0, c_n * 4));
        }
        CS_HIP(hipMemset(dB, 0, (size_t)N * 4));
        CS_HIP(hipMemset(dF, 0, 4));
        CS_TRY(launch_split_rows(dA, sA, M, K, dF, nullptr));
        CS_TRY(launch_split_rows(dW, sW, N, K, dF, nullptr));
        CS_HIP(hipEventCreate(&e0)); CS_HIP(hipEventCreate(&e1));
        auto once = [&]() -> int32_t {
            if (mode == CS_GEMM_F32)
                return launch_gemm(epilogue == 1 ? GEMM_GELU : epilogue == 2 ? GEMM_RESID : GEMM_BIAS, dA, dW, dB, dR, dC, M, N, K, nullptr);
            if (epilogue == 3) return launch_gemm_wide_ln(sA, sW, dB, dR, dB, dB, 1e-12f, dR, sC, M, K, dF, nullptr);
            const int epi = epilogue == 0 ? SH_OUT_F32 : epilogue == 1 ? SH_OUT_SPLIT_GELU : epilogue == 2 ? SH_OUT_F32_RESID : SH_OUT_SPLIT;
            if (mode == 2) return launch_gemm_wide(epi, sA, sW, dB, dR, dC, sC, M, N, K, dF, nullptr, 0);
            return launch_gemm_split(epi, sA, sW, dB, dR, dC, sC, M, N, K, dF, nullptr);
        };
        // ablation >= 100: DMA schedule ablation - 100 of the product kernel (gemm_wide.hip gw_dma_slot), any epilogue
        // ablation 192 / 384: that block shape of the product kernel, any epilogue
        // ablation 3192 / 3384: that block shape with the main loop on the 32 x 32 x 16 MFMA (gemm_wide32.hip); 1192 / 1384:
        // the 16 x 16 x 32 form whatever CS_GEMM_WIDE_MFMA says
        if (mode == 2 && (ablation == 3192 || ablation == 3384 || ablation == 1192 || ablation == 1384)) {
            cs::g_gemm_wide_mfma = ablation >= 3000 ? 32 : 16;
            ablation %= 1000;
        }
        cs::g_gemm_wide_shape = (mode == 2 && (ablation == 192 || ablation == 384)) ? ablation : 0;
        cs::g_gemm_wide_ablation = (mode == 2 && epilogue == 4 && ablation < 100) ? ablation : 0;
        for (int i = 0; i < 3; ++i) CS_TRY(once());
        CS_HIP(hipEventRecord(e0, nullptr));
        for (uint32_t i = 0; i < iters; ++i) CS_TRY(once());
        CS_HIP(hipEventRecord(e1, nullptr));
        CS_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        CS_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_per_launch = (double)ms / iters;
        if (cs::g_gemm_wide_ablation == 7)  // stamped build: the clock the blocks of the LAST launch ran at
        {
            double mc = 0.0, ec = 0.0;
            const double ghz = cs::gemm_wide_read_clock_ghz(&mc, &ec);
            fprintf(stderr, "gemm_wide in-kernel clock: %.3f GHz (median over blocks, last of %u launches, %.1f us each); per tile: "
                            "k loop %.0f cycles, epilogue %.0f cycles\n", ghz, iters, (double)ms / iters * 1e3, mc, ec);
        }
        return CS_OK;
    };
    const int32_t st = run();
    cs::g_gemm_wide_ablation = 0;
    cs::g_gemm_wide_shape = 0;
    cs::g_gemm_wide_mfma = 0;
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)dA, (void*)dW, (void*)dB, (void*)dR, (void*)dC, (void*)sA, (void*)sW, (void*)sC, (void*)dF})
        if (p) (void)hipFree(p);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return st;
}

}  // extern "C"
