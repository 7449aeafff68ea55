// embedder.hip — cs_embedder_*: the device half of the reference's FastEmbedder
// (/root/reference/src/embed/embedder.rs:201-322).  Starts from token ids (tokenisation is a
// host concern); runs the encoder kernels of encoder.hip; returns pooled, L2-normalised
// embeddings.  Mini-batching and the shutdown poll follow embed_batch_chunked
// (embedder.rs:266-295).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "encoder.hpp"

using namespace cs;

struct cs_embedder {
    int device = 0;
    cs_bert_config cfg{};
    cs_bert_offsets off{};
    float* d_params = nullptr;
    float* d_wqkv = nullptr;  // [layers][3H][H]  (query | key | value rows)
    float* d_bqkv = nullptr;  // [layers][3H]
    hipStream_t stream = nullptr;
    size_t cap_tokens = 0, cap_seqs = 0;
    int32_t* d_ids = nullptr;
    int32_t* d_mask = nullptr;
    float* d_x = nullptr;       // [T, H]
    float* d_qkv = nullptr;     // [T, 3H]
    float* d_ctx = nullptr;     // [T, H]
    float* d_mid = nullptr;     // [T, I]
    float* d_pooled = nullptr;  // [B, H]
    uint32_t last_B = 0, last_L = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double forward_ms = 0.0;
    uint64_t forwards = 0;
};

namespace {

void free_workspace(cs_embedder* h) {
    if (h->d_ids) (void)hipFree(h->d_ids);
    if (h->d_mask) (void)hipFree(h->d_mask);
    if (h->d_x) (void)hipFree(h->d_x);
    if (h->d_qkv) (void)hipFree(h->d_qkv);
    if (h->d_ctx) (void)hipFree(h->d_ctx);
    if (h->d_mid) (void)hipFree(h->d_mid);
    if (h->d_pooled) (void)hipFree(h->d_pooled);
    h->d_ids = h->d_mask = nullptr;
    h->d_x = h->d_qkv = h->d_ctx = h->d_mid = h->d_pooled = nullptr;
    h->cap_tokens = h->cap_seqs = 0;
}

int32_t reserve(cs_embedder* h, size_t seqs, size_t tokens) {
    if (tokens <= h->cap_tokens && seqs <= h->cap_seqs) return CS_OK;
    free_workspace(h);
    const size_t H = h->cfg.hidden, I = h->cfg.intermediate;
    CS_HIP(hipMalloc(&h->d_ids, tokens * sizeof(int32_t)));
    CS_HIP(hipMalloc(&h->d_mask, tokens * sizeof(int32_t)));
    CS_HIP(hipMalloc(&h->d_x, tokens * H * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_qkv, tokens * 3 * H * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_ctx, tokens * H * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_mid, tokens * I * sizeof(float)));
    CS_HIP(hipMalloc(&h->d_pooled, seqs * H * sizeof(float)));
    h->cap_tokens = tokens;
    h->cap_seqs = seqs;
    return CS_OK;
}

// One mini-batch already on the device (d_ids/d_mask) -> d_pooled [B, H].
int32_t forward(cs_embedder* h, uint32_t B, uint32_t L) {
    const cs_bert_config& c = h->cfg;
    const uint32_t H = c.hidden, I = c.intermediate, T = B * L;
    const float* P = h->d_params;
    hipStream_t s = h->stream;
    CS_HIP(hipEventRecord(h->ev0, s));
    EncoderLaunch a;
    a.ids = h->d_ids; a.mask = h->d_mask;
    a.word = P + h->off.word; a.pos = P + h->off.pos; a.type0 = P + h->off.type;
    a.g = P + h->off.emb_ln_g; a.b = P + h->off.emb_ln_b;
    a.eps = c.layer_norm_eps; a.T = T; a.L = L; a.B = B; a.vocab = c.vocab_size;
    a.pooling = c.pooling; a.x = h->d_x; a.out = h->d_pooled;
    CS_TRY(launch_row_kernel(0, a, H, s));  // E1
    for (uint32_t l = 0; l < c.layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(&c, &h->off, l, &lo);
        const float* wqkv = h->d_wqkv + (size_t)l * 3 * H * H;
        const float* bqkv = h->d_bqkv + (size_t)l * 3 * H;
        CS_TRY(launch_gemm(GEMM_BIAS, h->d_x, wqkv, bqkv, nullptr, h->d_qkv, T, 3 * H, H, s));        // E2
        CS_TRY(launch_attention(h->d_qkv, h->d_mask, h->d_ctx, B, L, H, c.heads, s));                 // E3
        CS_TRY(launch_gemm(GEMM_RESID, h->d_ctx, P + lo.ao_w, P + lo.ao_b, h->d_x, h->d_x, T, H, H, s)); // E4
        a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
        CS_TRY(launch_row_kernel(1, a, H, s));
        CS_TRY(launch_gemm(GEMM_GELU, h->d_x, P + lo.up_w, P + lo.up_b, nullptr, h->d_mid, T, I, H, s)); // E5
        CS_TRY(launch_gemm(GEMM_RESID, h->d_mid, P + lo.down_w, P + lo.down_b, h->d_x, h->d_x, T, H, I, s)); // E6
        a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
        CS_TRY(launch_row_kernel(1, a, H, s));
    }
    CS_TRY(launch_row_kernel(2, a, H, s));  // E7 + E8
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

int32_t embed_impl(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n,
                   uint32_t seq_len, uint32_t batch, float* out, bool out_on_device,
                   const volatile int32_t* cancel) {
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
        CS_TRY(forward(h, B, seq_len));
        CS_HIP(hipMemcpyAsync(out + done * H, h->d_pooled, (size_t)B * H * sizeof(float),
                              out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        CS_HIP(hipStreamSynchronize(h->stream));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) {
            h->forward_ms += ms;
            h->forwards += 1;
        }
    }
    return CS_OK;
}

}  // namespace

extern "C" {

void cs_bert_config_bge_small(cs_bert_config* cfg) {
    if (!cfg) return;
    cfg->vocab_size = 30522; cfg->hidden = 384; cfg->layers = 12; cfg->heads = 12;
    cfg->intermediate = 1536; cfg->max_position = 512; cfg->type_vocab_size = 2;
    cfg->layer_norm_eps = 1e-12f; cfg->pooling = CS_POOL_CLS;
}

uint64_t cs_bert_param_count(const cs_bert_config* cfg) {
    if (!cfg) return 0;
    cs_bert_offsets off;
    cs_bert_layout(cfg, &off);
    return off.total;
}

int32_t cs_embedder_create(const cs_bert_config* cfg, const float* params, uint64_t seed,
                           int32_t device, cs_embedder** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!cfg) return fail(CS_ERR_BAD_ARG, "cfg is null");
    if (cfg->hidden == 0 || cfg->heads == 0 || cfg->hidden % cfg->heads || cfg->layers == 0 ||
        cfg->vocab_size == 0 || cfg->max_position == 0 || cfg->type_vocab_size == 0)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: inconsistent config");
    if (cfg->hidden != 384 && cfg->hidden != 768 && cfg->hidden != 1024)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: hidden size %u not supported", cfg->hidden);
    if (cfg->hidden / cfg->heads != 32)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: head_dim %u not supported (32 only)",
                    cfg->hidden / cfg->heads);
    if (cfg->intermediate % 128 || cfg->hidden % 128)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: sizes must be multiples of 128");
    if (cfg->pooling != CS_POOL_CLS && cfg->pooling != CS_POOL_MEAN)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: unknown pooling %d", cfg->pooling);
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
    if (s == CS_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = fail(CS_ERR_HIP, "parameter setup failed");
    if (s != CS_OK) return cleanup(s);
    *out = h;
    return CS_OK;
}

void cs_embedder_destroy(cs_embedder* h) {
    if (!h) return;
    DeviceGuard g(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_workspace(h);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_wqkv) (void)hipFree(h->d_wqkv);
    if (h->d_bqkv) (void)hipFree(h->d_bqkv);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

uint32_t cs_embedder_dim(const cs_embedder* h) { return h ? h->cfg.hidden : 0; }

int32_t cs_embedder_embed_ids(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n,
                              uint32_t seq_len, uint32_t batch, float* out,
                              const volatile int32_t* cancel) {
    return embed_impl(h, ids, mask, n, seq_len, batch, out, false, cancel);
}

int32_t cs_embedder_embed_ids_device(cs_embedder* h, const int32_t* ids, const int32_t* mask,
                                     uint64_t n, uint32_t seq_len, uint32_t batch, float* d_out,
                                     const volatile int32_t* cancel) {
    return embed_impl(h, ids, mask, n, seq_len, batch, d_out, true, cancel);
}

int32_t cs_embedder_last_hidden(cs_embedder* h, float* out, uint64_t n_tokens) {
    if (!h || !out) return fail(CS_ERR_BAD_ARG, "null argument");
    if (n_tokens > (uint64_t)h->last_B * h->last_L)
        return fail(CS_ERR_BAD_ARG, "only %u tokens in the last mini-batch", h->last_B * h->last_L);
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

}  // extern "C"
