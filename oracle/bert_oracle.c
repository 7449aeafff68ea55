/*
 * bert_oracle.c — CPU restatement of the embedding forward pass.
 * TEST INFRASTRUCTURE ONLY (see cs_oracle.h).
 *
 * The reference computes embeddings inside un-vendored third-party code: fastembed 5.8.1
 * -> ort 2.0.0-rc.11 (ONNX Runtime) running the exported BAAI/bge-small-en-v1.5 graph
 * (call site /root/reference/src/embed/embedder.rs:286-289, `self.model.embed(refs, None)`).
 * This file restates the published algorithm of that graph = HF `BertModel` (no pooler):
 *   embeddings = word + position + token_type(0) -> LayerNorm(eps)
 *   12 x { Q,K,V = x W^T + b ; per head softmax(QK^T/sqrt(d_h) + mask) V ;
 *          x = LayerNorm(x + attn W_o^T + b_o) ;
 *          x = LayerNorm(x + gelu_erf(x W_1^T + b_1) W_2^T + b_2) }
 * followed by fastembed's pooling (CLS for the BGE family, mean for MiniLM/E5/...) and
 * L2 normalisation v / (|v| + 1e-12).
 * PARITY UNPINNED against the reference itself (it ships no embedding vectors,
 * SURVEY.md §4, §8c); pinned against HF transformers BertModel (float64) by
 * tests/golden/make_encoder_golden.py -> tests/golden/encoder_golden.npz.
 *
 * fp32 throughout, one rounding per multiply and per add (-ffp-contract=off), sums in
 * ascending k.  Parameters: flat block, layout of include/cs_bert_params.h.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/cs_bert_params.h"
#include "cs_oracle.h"

/* y[T,N] = x[T,K] W[N,K]^T + b.  wt = W transposed to [K,N] so the inner loop runs over
 * outputs (vectorisable) while each output still accumulates in ascending k. */
static void linear(const float* x, const float* w, const float* b, float* y, size_t T, size_t K,
                   size_t N) {
    float* wt = (float*)malloc(sizeof(float) * K * N);
    for (size_t n = 0; n < N; ++n)
        for (size_t k = 0; k < K; ++k) wt[k * N + n] = w[n * K + k];
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t t = 0; t < (int64_t)T; ++t) {
        float* yt = y + (size_t)t * N;
        const float* xt = x + (size_t)t * K;
        for (size_t n = 0; n < N; ++n) yt[n] = 0.0f;
        for (size_t k = 0; k < K; ++k) {
            const float xv = xt[k];
            const float* wr = wt + k * N;
            for (size_t n = 0; n < N; ++n) yt[n] += xv * wr[n];
        }
        for (size_t n = 0; n < N; ++n) yt[n] += b[n];
    }
    free(wt);
}

static void layer_norm_rows(float* x, const float* g, const float* b, size_t T, size_t H, float eps) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t t = 0; t < (int64_t)T; ++t) {
        float* r = x + (size_t)t * H;
        float mean = 0.0f;
        for (size_t i = 0; i < H; ++i) mean += r[i];
        mean /= (float)H;
        float var = 0.0f;
        for (size_t i = 0; i < H; ++i) { float d = r[i] - mean; var += d * d; }
        var /= (float)H;
        const float inv = 1.0f / sqrtf(var + eps);
        for (size_t i = 0; i < H; ++i) r[i] = (r[i] - mean) * inv * g[i] + b[i];
    }
}

static inline float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

/* ids/mask: [B, L] int32.  hidden_out: optional [B*L*H] last_hidden_state.  pooled_out:
 * [B, H] pooled + L2-normalised.  layer_hidden_out: optional [layers+1][B*L*H] (embedding
 * output then every layer's output) for per-layer parity checks. */
void cs_oracle_bert_forward(const cs_bert_config* cfg, const float* params, const int32_t* ids,
                            const int32_t* mask, uint32_t B, uint32_t L, float* hidden_out,
                            float* pooled_out, float* layer_hidden_out) {
    const size_t H = cfg->hidden, I = cfg->intermediate, NH = cfg->heads, DH = H / NH;
    const size_t T = (size_t)B * L;
    cs_bert_offsets off;
    cs_bert_layout(cfg, &off);
    float* x = (float*)malloc(sizeof(float) * T * H);
    float* q = (float*)malloc(sizeof(float) * T * H);
    float* k = (float*)malloc(sizeof(float) * T * H);
    float* v = (float*)malloc(sizeof(float) * T * H);
    float* ctx = (float*)malloc(sizeof(float) * T * H);
    float* tmp = (float*)malloc(sizeof(float) * T * H);
    float* mid = (float*)malloc(sizeof(float) * T * I);

    /* embeddings: word + token_type(0) + position, then LayerNorm (HF BertEmbeddings order:
     * inputs_embeds + token_type_embeddings, then + position_embeddings) */
    for (size_t t = 0; t < T; ++t) {
        const size_t pos = t % L;
        const float* we = params + off.word + (size_t)ids[t] * H;
        const float* pe = params + off.pos + pos * H;
        const float* te = params + off.type; /* token_type_ids = 0 */
        for (size_t i = 0; i < H; ++i) x[t * H + i] = (we[i] + te[i]) + pe[i];
    }
    layer_norm_rows(x, params + off.emb_ln_g, params + off.emb_ln_b, T, H, cfg->layer_norm_eps);
    if (layer_hidden_out) memcpy(layer_hidden_out, x, sizeof(float) * T * H);

    const float scale = 1.0f / sqrtf((float)DH);
    for (uint32_t l = 0; l < cfg->layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(cfg, &off, l, &lo);
        linear(x, params + lo.q_w, params + lo.q_b, q, T, H, H);
        linear(x, params + lo.k_w, params + lo.k_b, k, T, H, H);
        linear(x, params + lo.v_w, params + lo.v_b, v, T, H, H);
        /* attention per (batch, head, query row) */
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static)
#endif
        for (int64_t b = 0; b < (int64_t)B; ++b) {
            for (int64_t h = 0; h < (int64_t)NH; ++h) {
                float* s = (float*)malloc(sizeof(float) * L);
                for (size_t i = 0; i < L; ++i) {
                    const float* qi = q + ((size_t)b * L + i) * H + (size_t)h * DH;
                    float mx = -INFINITY;
                    for (size_t j = 0; j < L; ++j) {
                        const float* kj = k + ((size_t)b * L + j) * H + (size_t)h * DH;
                        float d = 0.0f;
                        for (size_t e = 0; e < DH; ++e) d += qi[e] * kj[e];
                        d *= scale;
                        /* additive mask: (1 - m) * finfo(f32).min, as HF get_extended_attention_mask */
                        if (!mask[(size_t)b * L + j]) d += -3.4028234663852886e38f;
                        s[j] = d;
                        if (d > mx) mx = d;
                    }
                    float sum = 0.0f;
                    for (size_t j = 0; j < L; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
                    float* o = ctx + ((size_t)b * L + i) * H + (size_t)h * DH;
                    for (size_t e = 0; e < DH; ++e) o[e] = 0.0f;
                    for (size_t j = 0; j < L; ++j) {
                        const float p = s[j] / sum;
                        const float* vj = v + ((size_t)b * L + j) * H + (size_t)h * DH;
                        for (size_t e = 0; e < DH; ++e) o[e] += p * vj[e];
                    }
                }
                free(s);
            }
        }
        linear(ctx, params + lo.ao_w, params + lo.ao_b, tmp, T, H, H);
        for (size_t i = 0; i < T * H; ++i) x[i] = tmp[i] + x[i]; /* BertSelfOutput: dense + residual */
        layer_norm_rows(x, params + lo.ao_ln_g, params + lo.ao_ln_b, T, H, cfg->layer_norm_eps);
        linear(x, params + lo.up_w, params + lo.up_b, mid, T, H, I);
        for (size_t i = 0; i < T * I; ++i) mid[i] = gelu_erf(mid[i]);
        linear(mid, params + lo.down_w, params + lo.down_b, tmp, T, I, H);
        for (size_t i = 0; i < T * H; ++i) x[i] = tmp[i] + x[i];
        layer_norm_rows(x, params + lo.out_ln_g, params + lo.out_ln_b, T, H, cfg->layer_norm_eps);
        if (layer_hidden_out) memcpy(layer_hidden_out + (size_t)(l + 1) * T * H, x, sizeof(float) * T * H);
    }
    if (hidden_out) memcpy(hidden_out, x, sizeof(float) * T * H);

    if (pooled_out) {
        for (size_t b = 0; b < B; ++b) {
            float* p = pooled_out + b * H;
            if (cfg->pooling == CS_POOL_CLS) {
                memcpy(p, x + b * L * H, sizeof(float) * H);
            } else { /* mean over unmasked tokens: sum(m*h) / max(sum m, 1e-9) */
                float cnt = 0.0f;
                for (size_t i = 0; i < H; ++i) p[i] = 0.0f;
                for (size_t t = 0; t < L; ++t) {
                    if (!mask[b * L + t]) continue;
                    cnt += 1.0f;
                    const float* r = x + (b * L + t) * H;
                    for (size_t i = 0; i < H; ++i) p[i] += r[i];
                }
                if (cnt < 1e-9f) cnt = 1e-9f;
                for (size_t i = 0; i < H; ++i) p[i] /= cnt;
            }
            float ss = 0.0f;
            for (size_t i = 0; i < H; ++i) ss += p[i] * p[i];
            const float den = sqrtf(ss) + 1e-12f;
            for (size_t i = 0; i < H; ++i) p[i] /= den;
        }
    }
    free(x); free(q); free(k); free(v); free(ctx); free(tmp); free(mid);
}

void cs_oracle_bert_synth_params(const cs_bert_config* cfg, uint64_t seed, float* out) {
    cs_bert_offsets off;
    cs_bert_layout(cfg, &off);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t e = 0; e < (int64_t)off.total; ++e) out[e] = cs_bert_synth_param(cfg, &off, seed, (uint64_t)e);
}

uint64_t cs_oracle_bert_param_count(const cs_bert_config* cfg) {
    cs_bert_offsets off;
    cs_bert_layout(cfg, &off);
    return off.total;
}
