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
 * CS_ARCH_NOMIC (cs_bert_config.arch; the registry's three Nomic entries, /root/reference/src/embed/embedder.rs:30-35,
 * :64-66 -> fastembed's NomicEmbedTextV1 / V15 / V15Q) restates the NomicBert encoder those files export
 * (nomic-ai/nomic-embed-text-v1*, `modeling_hf_nomic_bert.py` of the model repository; third-party code, absent here —
 * restated from its published definition): no position table; Q and K of every head rotated by the non-interleaved
 * rotary map (x1, x2) -> (x1 cos - x2 sin, x2 cos + x1 sin) over the head's two halves, angle = pos * base^(-2i/d_h)
 * (f32 throughout, as the module computes its cos / sin cache); the feed-forward is  fc2( fc11(x) * silu(fc12(x)) );
 * post-LayerNorm, masking, mean pooling and normalisation as above.  Pinned against a float64 torch statement that takes
 * its rotary map from transformers' own `rotate_half` / `apply_rotary_pos_emb` (tests/golden/make_nomic_golden.py).
 * CS_ARCH_JINA / CS_ARCH_JINA_QKNORM (the registry's JinaEmbeddingsV2BaseCode, /root/reference/src/embed/embedder.rs:40-41,
 * :69, :92, :112 -> fastembed's JinaEmbeddingsV2BaseCode) restate JinaBert (jinaai/jina-bert-implementation and
 * jinaai/jina-bert-v2-qk-post-norm `modeling_bert.py`; third-party code, absent here — restated from its published
 * definition): embeddings = word + token_type -> LayerNorm (no position table); scores
 *   softmax(Q K^T / sqrt(d_h) + mask + alibi),  alibi[h][i][j] = -slope_h |i - j|  (the symmetric, encoder form),
 * slopes the geometric sequence of the ALiBi paper (closest power of two, then every second slope of the doubled set);
 * _QKNORM: LayerNorm over the WHOLE query row and the whole key row (H columns, own gamma / beta each) before the cut into
 * heads; the feed-forward is  down( value * gelu_erf(gate) )  over the two halves of one bias-free [2I, H] projection;
 * post-LayerNorm, mean pooling, normalisation as above.  Pinned against a float64 torch statement written from the same
 * definition (tests/golden/make_jina_golden.py) — PARITY UNPINNED against the model itself.
 * CS_ARCH_MODERN (the registry's ModernBertEmbedLarge, /root/reference/src/embed/embedder.rs:47, :72 -> fastembed's
 * ModernBertEmbedLarge = lightonai/modernbert-embed-large) restates HF transformers' `ModernBertModel` (modeling_modernbert.py,
 * installed here as a library and read as its published definition): tok_embeddings -> LayerNorm; per layer
 *   x = x + Wo( attention( rope( Wqkv LN_attn(x) ) ) )      (layer 0: no LN_attn; rope base per layer type; local layers mask
 *   x = x + Wo_mlp( gelu_erf(Wi_a LN_mlp(x)) * Wi_b LN_mlp(x) )                         |i - j| > local_window)
 * then final_norm, mean pooling, L2 normalisation.  Pinned against HF's own ModernBertModel in float64
 * (tests/golden/make_modern_golden.py).
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

/* ---- dynamic quantisation (the registry's *Q models; the reference's DEFAULT model is one: ModelType::AllMiniLML6V2Q,
 * /root/reference/src/embed/embedder.rs:12-13,367-372) -------------------------------------------------------------
 * Their ONNX files come out of onnxruntime's quantize_dynamic: every Linear is
 *   x -> DynamicQuantizeLinear -> MatMulInteger(x_q, W_q, x_zp, W_zp) -> Cast(f32) -> Mul(x_scale * W_scale) -> Add(bias)
 * The ops are ONNX standard ops (opset 11 / 10); this restates their published definitions:
 *   DynamicQuantizeLinear (per tensor, uint8):  lo = min(0, min x), hi = max(0, max x);
 *       x_scale = (hi - lo) / 255   (1 when hi == lo);   x_zp = sat_u8(round_half_even(0 - lo / x_scale));
 *       x_q = sat_u8(round_half_even(x / x_scale) + x_zp)
 *   MatMulInteger:  acc[t][n] = sum_k (x_q[t][k] - x_zp) * (W_q[k][n] - W_zp[n])   in int32 (exact)
 * `w` is the f32 block's dequantised weight (W_q - W_zp) * W_scale, [N][K]; the integers W_q - W_zp are recovered
 * exactly as round(w / W_scale) (at most 255 in magnitude, one f32 rounding in w; a zero scale means a zero column), so
 * the zero point itself is not needed.  wscale: one per output column n (a per-tensor file repeats its value).  PARITY UNPINNED against onnxruntime itself (not installed here; no model file offline). */
static void linear_q8(const float* x, const float* w, const float* wscale, const float* b,
                      float* y, size_t T, size_t K, size_t N) {
    float lo = 0.0f, hi = 0.0f;
    for (size_t i = 0; i < T * K; ++i) {
        if (x[i] < lo) lo = x[i];
        if (x[i] > hi) hi = x[i];
    }
    const float xs = hi == lo ? 1.0f : (hi - lo) / 255.0f;
    float z = 0.0f - lo / xs;
    if (z < 0.0f) z = 0.0f;
    if (z > 255.0f) z = 255.0f;
    const int32_t xz = (int32_t)nearbyintf(z); /* FE_TONEAREST: ties to even */
    uint8_t* xq = (uint8_t*)malloc(T * K);
    for (size_t i = 0; i < T * K; ++i) {
        float v = nearbyintf(x[i] / xs) + (float)xz;
        if (v < 0.0f) v = 0.0f;
        if (v > 255.0f) v = 255.0f;
        xq[i] = (uint8_t)v;
    }
    int32_t* wq = (int32_t*)malloc(sizeof(int32_t) * K * N); /* [K][N], already minus its zero point */
    for (size_t n = 0; n < N; ++n)
        for (size_t k = 0; k < K; ++k)
            wq[k * N + n] = wscale[n] != 0.0f ? (int32_t)nearbyintf(w[n * K + k] / wscale[n]) : 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t t = 0; t < (int64_t)T; ++t) {
        int32_t* acc = (int32_t*)calloc(N, sizeof(int32_t));
        const uint8_t* xt = xq + (size_t)t * K;
        for (size_t k = 0; k < K; ++k) {
            const int32_t xv = (int32_t)xt[k] - xz;
            const int32_t* wr = wq + k * N;
            for (size_t n = 0; n < N; ++n) acc[n] += xv * wr[n];
        }
        float* yt = y + (size_t)t * N;
        for (size_t n = 0; n < N; ++n) yt[n] = (float)acc[n] * (xs * wscale[n]) + b[n];
        free(acc);
    }
    free(wq);
    free(xq);
}

/* One dynamically quantised Linear on its own (tests pin it against a numpy statement of the ONNX operators). */
void cs_oracle_linear_q8(const float* x, const float* w, const float* wscale, const float* b, float* y, uint64_t T,
                         uint64_t K, uint64_t N) {
    linear_q8(x, w, wscale, b, y, (size_t)T, (size_t)K, (size_t)N);
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
static inline float silu(float v) { return v / (1.0f + expf(-v)); }

/* NomicBert: rotary position map on the [T, H] tensor of queries (or keys), head by head, in place.  The cos / sin cache
 * as the module builds it: inv_freq_i = 1 / base^(2i / d_h), angle = pos * inv_freq_i, all in f32. */
static void rotary_rows(float* qk, size_t T, size_t L, size_t NH, size_t DH, float base) {
    const size_t half = DH / 2;
    float* inv_freq = (float*)malloc(sizeof(float) * half);
    for (size_t i = 0; i < half; ++i) inv_freq[i] = 1.0f / powf(base, (float)(2 * i) / (float)DH);
    for (size_t t = 0; t < T; ++t) {
        const float pos = (float)(t % L);
        for (size_t h = 0; h < NH; ++h) {
            float* r = qk + t * NH * DH + h * DH;
            for (size_t i = 0; i < half; ++i) {
                const float ang = pos * inv_freq[i];
                const float c = cosf(ang), s = sinf(ang);
                const float x1 = r[i], x2 = r[i + half];
                r[i] = x1 * c - x2 * s;
                r[i + half] = x2 * c + x1 * s;
            }
        }
    }
    free(inv_freq);
}

/* ALiBi head slopes as JinaBert's `_get_alibi_head_slopes` forms them (Python floats = doubles, then an f32 tensor). */
static void alibi_pow2(size_t n, double* out) {
    const double start = pow(2.0, -pow(2.0, -(log2((double)n) - 3.0)));
    for (size_t i = 0; i < n; ++i) out[i] = start * pow(start, (double)i);
}
static void alibi_slopes(size_t n, float* out) {
    double* tmp = (double*)malloc(sizeof(double) * 2 * n + 16);
    size_t closest = 1;
    while (closest * 2 <= n) closest *= 2;
    if (closest == n) {
        alibi_pow2(n, tmp);
        for (size_t i = 0; i < n; ++i) out[i] = (float)tmp[i];
    } else {
        alibi_pow2(closest, tmp);
        for (size_t i = 0; i < closest; ++i) out[i] = (float)tmp[i];
        alibi_pow2(2 * closest, tmp); /* (2 * closest is a power of two: the recursion ends here) */
        for (size_t i = 0; i < n - closest; ++i) out[closest + i] = (float)tmp[2 * i];
    }
    free(tmp);
}

/* ids/mask: [B, L] int32.  hidden_out: optional [B*L*H] last_hidden_state.  pooled_out:
 * [B, H] pooled + L2-normalised.  layer_hidden_out: optional [layers+1][B*L*H] (embedding
 * output then every layer's output) for per-layer parity checks. */
/* ModernBERT (CS_ARCH_MODERN): pre-norm layers, see the header comment. */
static void modern_forward_impl(const cs_bert_config* cfg, const float* params, const int32_t* ids, const int32_t* mask,
                                uint32_t B, uint32_t L, float* hidden_out, float* pooled_out, float* layer_hidden_out) {
    const size_t H = cfg->hidden, I = cfg->intermediate, NH = cfg->heads, DH = H / NH;
    const size_t T = (size_t)B * L;
    cs_bert_offsets off;
    cs_bert_layout(cfg, &off);
    float* x = (float*)malloc(sizeof(float) * T * H);
    float* n = (float*)malloc(sizeof(float) * T * H);
    float* q = (float*)malloc(sizeof(float) * T * H);
    float* k = (float*)malloc(sizeof(float) * T * H);
    float* v = (float*)malloc(sizeof(float) * T * H);
    float* ctx = (float*)malloc(sizeof(float) * T * H);
    float* tmp = (float*)malloc(sizeof(float) * T * H);
    float* mid = (float*)malloc(sizeof(float) * T * I);
    float* gate = (float*)malloc(sizeof(float) * T * I);
    for (size_t t = 0; t < T; ++t) memcpy(x + t * H, params + off.word + (size_t)ids[t] * H, sizeof(float) * H);
    layer_norm_rows(x, params + off.emb_ln_g, params + off.emb_ln_b, T, H, cfg->layer_norm_eps);
    if (layer_hidden_out) memcpy(layer_hidden_out, x, sizeof(float) * T * H);
    const float scale = 1.0f / sqrtf((float)DH);  /* head_dim ** -0.5 */
    for (uint32_t l = 0; l < cfg->layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(cfg, &off, l, &lo);
        const int global = cfg->global_every ? (l % cfg->global_every == 0) : 1;
        memcpy(n, x, sizeof(float) * T * H);
        if (l) layer_norm_rows(n, params + lo.ao_ln_g, params + lo.ao_ln_b, T, H, cfg->layer_norm_eps); /* attn_norm; Identity in layer 0 */
        linear(n, params + lo.q_w, params + lo.q_b, q, T, H, H);
        linear(n, params + lo.k_w, params + lo.k_b, k, T, H, H);
        linear(n, params + lo.v_w, params + lo.v_b, v, T, H, H);
        rotary_rows(q, T, L, NH, DH, global ? cfg->rotary_base : cfg->rotary_base_local);
        rotary_rows(k, T, L, NH, DH, global ? cfg->rotary_base : cfg->rotary_base_local);
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
                        const size_t dist = i > j ? i - j : j - i;
                        /* additive mask, finfo(f32).min: padded keys, and keys outside the local window */
                        if (!mask[(size_t)b * L + j] || (!global && dist > cfg->local_window)) d += -3.4028234663852886e38f;
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
        for (size_t i = 0; i < T * H; ++i) x[i] = x[i] + tmp[i];
        memcpy(n, x, sizeof(float) * T * H);
        layer_norm_rows(n, params + lo.out_ln_g, params + lo.out_ln_b, T, H, cfg->layer_norm_eps); /* mlp_norm */
        linear(n, params + lo.gate_w, params + lo.gate_b, gate, T, H, I); /* Wi rows [0, I): through the activation */
        linear(n, params + lo.up_w, params + lo.up_b, mid, T, H, I);      /* Wi rows [I, 2I) */
        for (size_t i = 0; i < T * I; ++i) mid[i] = gelu_erf(gate[i]) * mid[i];
        linear(mid, params + lo.down_w, params + lo.down_b, tmp, T, I, H);
        for (size_t i = 0; i < T * H; ++i) x[i] = x[i] + tmp[i];
        if (layer_hidden_out) memcpy(layer_hidden_out + (size_t)(l + 1) * T * H, x, sizeof(float) * T * H);
    }
    layer_norm_rows(x, params + off.final_ln_g, params + off.final_ln_b, T, H, cfg->layer_norm_eps);
    if (hidden_out) memcpy(hidden_out, x, sizeof(float) * T * H);
    if (pooled_out) {
        for (size_t b = 0; b < B; ++b) {
            float* p = pooled_out + b * H;
            if (cfg->pooling == CS_POOL_CLS) {
                memcpy(p, x + b * L * H, sizeof(float) * H);
            } else {
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
    free(x); free(n); free(q); free(k); free(v); free(ctx); free(tmp); free(mid); free(gate);
}

static void bert_forward_impl(const cs_bert_config* cfg, const float* params, const float* qscale,
                              const int32_t* ids,
                              const int32_t* mask, uint32_t B, uint32_t L, float* hidden_out,
                              float* pooled_out, float* layer_hidden_out) {
    if (cfg->arch == CS_ARCH_MODERN) {
        modern_forward_impl(cfg, params, ids, mask, B, L, hidden_out, pooled_out, layer_hidden_out);
        return;
    }
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
    const int nomic = cfg->arch == CS_ARCH_NOMIC;
    const int gated = cs_arch_gated(cfg->arch), alibi = cs_arch_alibi(cfg->arch);
    float* gate = gated ? (float*)malloc(sizeof(float) * T * I) : NULL;
    float* slopes = (float*)malloc(sizeof(float) * NH);
    if (alibi) alibi_slopes(NH, slopes);

    /* embeddings: word + token_type(0) + position, then LayerNorm (HF BertEmbeddings order:
     * inputs_embeds + token_type_embeddings, then + position_embeddings) */
    for (size_t t = 0; t < T; ++t) {
        const size_t pos = t % L;
        const float* we = params + off.word + (size_t)ids[t] * H;
        const float* pe = params + off.pos + pos * H;
        const float* te = params + off.type; /* token_type_ids = 0 */
        if (gated) { /* NomicBertEmbeddings / JinaBertEmbeddings (alibi): word + token_type, no position table */
            for (size_t i = 0; i < H; ++i) x[t * H + i] = we[i] + te[i];
            continue;
        }
        for (size_t i = 0; i < H; ++i) x[t * H + i] = (we[i] + te[i]) + pe[i];
    }
    layer_norm_rows(x, params + off.emb_ln_g, params + off.emb_ln_b, T, H, cfg->layer_norm_eps);
    if (layer_hidden_out) memcpy(layer_hidden_out, x, sizeof(float) * T * H);

    const float scale = 1.0f / sqrtf((float)DH);
    for (uint32_t l = 0; l < cfg->layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(cfg, &off, l, &lo);
        /* quantised files: column scales of the layer, query | key | value | attention.output |
         * intermediate | output (cs_bert_quant_columns, include/cs_bert_params.h) */
        const size_t qcols = 5 * H + I;
        const float* qs = qscale ? qscale + (size_t)l * qcols : NULL;
#define DENSE(xin, W, Bv, yout, Kk, Nn, col0)                                                     \
    do {                                                                                          \
        if (qs) linear_q8(xin, params + (W), qs + (col0), params + (Bv), yout, T, Kk, Nn);           \
        else linear(xin, params + (W), params + (Bv), yout, T, Kk, Nn);                           \
    } while (0)
        /* (the three projections share one DynamicQuantizeLinear of x in the graph: same x, same parameters) */
        DENSE(x, lo.q_w, lo.q_b, q, H, H, 0);
        DENSE(x, lo.k_w, lo.k_b, k, H, H, H);
        DENSE(x, lo.v_w, lo.v_b, v, H, H, 2 * H);
        if (nomic) {
            rotary_rows(q, T, L, NH, DH, cfg->rotary_base);
            rotary_rows(k, T, L, NH, DH, cfg->rotary_base);
        }
        if (cfg->arch == CS_ARCH_JINA_QKNORM) { /* layer_norm_q / layer_norm_k over the whole rows, before the heads */
            layer_norm_rows(q, params + lo.qln_g, params + lo.qln_b, T, H, cfg->layer_norm_eps);
            layer_norm_rows(k, params + lo.kln_g, params + lo.kln_b, T, H, cfg->layer_norm_eps);
        }
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
                        /* JinaBert: softmax(scores + bias), bias = slopes * -|i - j| (an f32 tensor product) */
                        if (alibi) d += slopes[h] * -(float)(i > j ? i - j : j - i);
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
        DENSE(ctx, lo.ao_w, lo.ao_b, tmp, H, H, 3 * H);
        for (size_t i = 0; i < T * H; ++i) x[i] = tmp[i] + x[i]; /* BertSelfOutput: dense + residual */
        layer_norm_rows(x, params + lo.ao_ln_g, params + lo.ao_ln_b, T, H, cfg->layer_norm_eps);
        DENSE(x, lo.up_w, lo.up_b, mid, H, I, 4 * H);
        if (nomic) { /* NomicBertGatedMLP, activation swiglu: y = fc11(x) * silu(fc12(x)) */
            linear(x, params + lo.gate_w, params + lo.gate_b, gate, T, H, I);
            for (size_t i = 0; i < T * I; ++i) mid[i] = mid[i] * silu(gate[i]);
        } else if (gated) { /* JinaBertGLUMLP (geglu): y = value * gelu(gate) over the halves of one [2I, H] projection */
            linear(x, params + lo.gate_w, params + lo.gate_b, gate, T, H, I);
            for (size_t i = 0; i < T * I; ++i) mid[i] = mid[i] * gelu_erf(gate[i]);
        } else {
            for (size_t i = 0; i < T * I; ++i) mid[i] = gelu_erf(mid[i]);
        }
        DENSE(mid, lo.down_w, lo.down_b, tmp, I, H, 4 * H + I);
#undef DENSE
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
    free(x); free(q); free(k); free(v); free(ctx); free(tmp); free(mid); free(gate); free(slopes);
}

void cs_oracle_bert_forward(const cs_bert_config* cfg, const float* params, const int32_t* ids,
                            const int32_t* mask, uint32_t B, uint32_t L, float* hidden_out,
                            float* pooled_out, float* layer_hidden_out) {
    bert_forward_impl(cfg, params, NULL, ids, mask, B, L, hidden_out, pooled_out, layer_hidden_out);
}

/* The same forward with every Linear run as onnxruntime's dynamic quantiser rewrites it (linear_q8 above): ONE
 * quantisation of each activation tensor [B*L, K] per call — padding rows included, as the graph sees them. */
void cs_oracle_bert_forward_q8(const cs_bert_config* cfg, const float* params, const float* wscale,
                               const int32_t* ids, const int32_t* mask, uint32_t B,
                               uint32_t L, float* hidden_out, float* pooled_out, float* layer_hidden_out) {
    bert_forward_impl(cfg, params, wscale, ids, mask, B, L, hidden_out, pooled_out, layer_hidden_out);
}

void cs_oracle_alibi_slopes(uint32_t n, float* out) { alibi_slopes((size_t)n, out); }

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
