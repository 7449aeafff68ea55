// embedder_embed.hip — embed_batch_chunked (embedder.rs:266-295) from token ids and from strings: mini-batches, the shutdown poll, length-grouped windows.
// (one of the translation units behind cs_embedder_*: see embedder_state.hpp)
#include "embedder_state.hpp"

using namespace cs;

namespace cs {
namespace emb {

// dst[perm[r]] = src[r] for r < rows: one float4 per thread (H % 4 == 0)
__global__ void __launch_bounds__(256)
scatter_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ perm, float* __restrict__ dst,
                    uint32_t rows, uint32_t h4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * h4) return;
    const uint32_t r = i / h4, c = i % h4;
    reinterpret_cast<float4*>(dst)[(size_t)perm[r] * h4 + c] = reinterpret_cast<const float4*>(src)[(size_t)r * h4 + c];
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

// perm (optional, only with n <= batch): pooled row r of the mini-batch goes to out row perm[r].
// units (optional, only with n <= batch and CS_GEMM_Q8_DYNAMIC): see UnitSpec.
int32_t embed_impl(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n,
                   uint32_t seq_len, uint32_t batch, float* out, bool out_on_device,
                   const volatile int32_t* cancel, const uint32_t* perm, const UnitSpec* units) {
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
        // a small mini-batch (a query and its variants) goes through 128 KiB of pinned memory: ids | mask in, the range flag and the
        // rows out behind ONE wait — from pageable memory every transfer is staged by the runtime and waits on its own
        constexpr size_t kPinHalf = 64 << 10;
        const bool pin_in = h->h_pin && tok * 2 * sizeof(int32_t) <= kPinHalf;
        if (pin_in) {  // (the stream is idle: the previous mini-batch ended with a wait)
            std::memcpy(h->h_pin, bi, tok * sizeof(int32_t));
            std::memcpy(h->h_pin + tok * sizeof(int32_t), mask + done * seq_len, tok * sizeof(int32_t));
            CS_HIP(hipMemcpyAsync(h->d_ids, h->h_pin, tok * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            CS_HIP(hipMemcpyAsync(h->d_mask, h->h_pin + tok * sizeof(int32_t), tok * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        } else {
            CS_HIP(hipMemcpyAsync(h->d_ids, bi, tok * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            CS_HIP(hipMemcpyAsync(h->d_mask, mask + done * seq_len, tok * sizeof(int32_t),
                                  hipMemcpyHostToDevice, h->stream));
        }
        const bool pin_out = h->h_pin && !perm && !out_on_device && (size_t)B * H * sizeof(float) + 64 <= kPinHalf;
        uint32_t* pin_flag = reinterpret_cast<uint32_t*>(h->h_pin + kPinHalf);
        float* pin_rows = reinterpret_cast<float*>(h->h_pin + kPinHalf + 64);
        bool fetched = false;  // the rows are in pin_rows already
        int mode = h->gemm_mode;
        h->cur_units = 1;
        // results wanted in device memory in input order: the pooling kernel stores them there itself (E8 in place when `out`
        // is a corpus region, cs_index_reserve_rows) — no [B, H] copy behind the forward
        struct DstGuard { cs_embedder* e; ~DstGuard() { e->pooled_dst = nullptr; } } dst_guard{h};
        h->pooled_dst = (out_on_device && !perm) ? out + done * H : nullptr;
        if (units && units->units > 1 && mode == CS_GEMM_Q8_DYNAMIC && n <= batch) {
            CS_HIP(hipMemcpyAsync(h->d_seq_unit, units->seq_unit, B * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            CS_HIP(hipMemcpyAsync(h->d_unit_len, units->unit_len, units->units * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            h->cur_units = units->units;
        }
        CS_TRY(forward(h, B, seq_len, mode));
        if (mode == CS_GEMM_Q8_DYNAMIC) {
            uint32_t flag = 0;
            if (pin_out) {
                CS_HIP(hipMemcpyAsync(pin_flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
                CS_HIP(hipMemcpyAsync(pin_rows, h->d_pooled, (size_t)B * H * sizeof(float), hipMemcpyDeviceToHost, h->stream));
                CS_HIP(hipStreamSynchronize(h->stream));
                flag = *pin_flag;
                fetched = !flag;
            } else {
                CS_HIP(hipMemcpyAsync(&flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
                CS_HIP(hipStreamSynchronize(h->stream));
            }
            h->q8_forwards += 1;
            if (flag) {
                // Q / K / V, an attention output or a GELU output beyond 65504 does not fit the split-f16 hand-over.  onnxruntime has
                // no such limit (embedder.rs:286-289 returns an embedding whatever the activations), so neither has this mode: the
                // mini-batch is run again as the f32 graph of the dequantised weights on the exact-f32 kernels — the same model
                // without the 8-bit rounding of the activations, i.e. inside the quantised graph's own noise band — and counted.
                h->range_fallbacks += 1;
                mode = CS_GEMM_F32;
                h->cur_units = 1;
                CS_TRY(forward(h, B, seq_len, mode));
            }
        } else if (mode == CS_GEMM_SPLIT_F16) {
            uint32_t flag = 0;
            bool pinned_here = pin_out;
#ifdef CS_DIAGNOSTICS
            if (h->sf_ran) pinned_here = false;  // (the one-launch forward's own hand-shake below)
#endif
            if (pinned_here) {
                CS_HIP(hipMemcpyAsync(pin_flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
                CS_HIP(hipMemcpyAsync(pin_rows, h->d_pooled, (size_t)B * H * sizeof(float), hipMemcpyDeviceToHost, h->stream));
            } else {
                CS_HIP(hipMemcpyAsync(&flag, h->d_flag, sizeof(flag), hipMemcpyDeviceToHost, h->stream));
            }
#ifdef CS_DIAGNOSTICS
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
#endif
            CS_HIP(hipStreamSynchronize(h->stream));
            if (pinned_here) { flag = *pin_flag; fetched = !flag; }
            h->split_forwards += 1;
            if (flag) {  // an activation left the f16 range: redo this mini-batch on the exact-f32 MFMA
                h->range_fallbacks += 1;
                mode = CS_GEMM_F32;
                CS_TRY(forward(h, B, seq_len, mode));
            }
        }
        if (mode == CS_GEMM_F32) h->f32_forwards += 1;
        if (fetched) {
            std::memcpy(out + done * H, pin_rows, (size_t)B * H * sizeof(float));
        } else if (!perm) {
            if (!h->pooled_dst)
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
        // (the wide GEMM addresses an operand with 32-bit byte offsets: a mini-batch's largest split-f16 tensor — token rows x
        // max(intermediate, 3 hidden) x 4 B — stays under 4 GiB; only a CODESEARCH_BATCH_SIZE far above the reference's 256 gets here.
        // Not for a quantised model's call tensors: they are the reference's units, and their tensors are a quarter the size)
        if (h->gemm_mode != CS_GEMM_Q8_DYNAMIC) {
            const uint64_t tok_cap = ((1ull << 32) - 1) / (4ull * std::max<uint64_t>(h->cfg.intermediate, 3ull * h->cfg.hidden));
            while (B > 1 && (uint64_t)B * L > tok_cap) {
                --B;
                L = 1;
                for (uint32_t r = 0; r < B; ++r) L = std::max(L, seqs[order[b0 + r]].len);
            }
        }
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
}  // namespace emb
}  // namespace cs
