/*
 * cs_bert_params.h — flat f32 parameter layout of the encoder and the synthetic-weight
 * rule.  Shared by libcsgpu (cs_embedder_create), the CPU oracle and the Python mirror
 * (codesearch_amd/bert_params.py).
 *
 * Layout = HF `BertModel(add_pooling_layer=False).state_dict()` order, every tensor
 * row-major, Linear weights [out_features, in_features] (y = x W^T + b):
 *
 *   embeddings.word_embeddings.weight            [V, H]
 *   embeddings.position_embeddings.weight        [P, H]
 *   embeddings.token_type_embeddings.weight      [T, H]
 *   embeddings.LayerNorm.weight / .bias          [H] [H]
 *   for each layer:
 *     attention.self.query.weight / .bias        [H, H] [H]
 *     attention.self.key.weight / .bias          [H, H] [H]
 *     attention.self.value.weight / .bias        [H, H] [H]
 *     attention.output.dense.weight / .bias      [H, H] [H]
 *     attention.output.LayerNorm.weight / .bias  [H] [H]
 *     intermediate.dense.weight / .bias          [I, H] [I]
 *     output.dense.weight / .bias                [H, I] [H]
 *     output.LayerNorm.weight / .bias            [H] [H]
 *
 * CS_ARCH_NOMIC (NomicBert: the registry's nomic-embed-text entries) keeps this order with two differences: there is NO
 * position table (`pos` == `type`: zero floats), and every layer carries a second up-projection behind the first —
 *     intermediate.dense.weight / .bias          [I, H] [I]     = mlp.fc11 (the value branch)
 *     intermediate.gate.weight / .bias           [I, H] [I]     = mlp.fc12 (the gate: y = fc11(x) * silu(fc12(x)))
 * — so that the two are ONE [2I, H] weight for the up-projection GEMM.  The published checkpoints have no Linear biases
 * (their slots are zero); the encoder applies whatever the slots hold.
 *
 * CS_ARCH_MODERN (ModernBERT: the registry's ModernBertEmbedLarge) keeps the gated order too — no position AND no token-type
 * table; per layer query | key | value (the thirds of the fused Wqkv) with their biases, attention.output.dense (= attn.Wo),
 * attention.output.LayerNorm (= the layer's attn_norm, applied BEFORE attention; layer 0's slot is never read),
 * intermediate.dense (= the half of mlp.Wi that is NOT activated: rows [I, 2I)), intermediate.gate (= rows [0, I), through
 * GELU), output.dense (= mlp.Wo), output.LayerNorm (= mlp_norm, applied BEFORE the feed-forward) — and behind the last layer
 *     final_norm.weight / .bias                  [H] [H]
 * The published checkpoints have no biases at all (attention_bias, mlp_bias, norm_bias false: zero slots).
 *
 * Synthetic weights (no checkpoint is reachable from the build/GPU boxes): element at flat
 * offset e of a tensor of kind K is  base(K) + cs_synth_weight(seed, e, shift(K))  — see
 * cs_bert_synth_rule().  Shifts keep activations O(1) and attention non-degenerate.
 */
#ifndef CS_BERT_PARAMS_H
#define CS_BERT_PARAMS_H

#include <stdint.h>

#include "codesearch_gpu.h"
#include "cs_synth.h"

typedef enum cs_bert_tensor_kind {
    CS_T_WORD_EMB = 0, CS_T_POS_EMB, CS_T_TYPE_EMB, CS_T_LN_GAMMA, CS_T_LN_BETA,
    CS_T_QK_W, CS_T_V_W, CS_T_ATTN_OUT_W, CS_T_FFN_UP_W, CS_T_FFN_DOWN_W, CS_T_BIAS
} cs_bert_tensor_kind;

/* shift, base for a tensor kind */
CS_SYNTH_FN void cs_bert_synth_rule(int kind, int* shift, float* base) {
    *base = 0.0f;
    switch (kind) {
        case CS_T_WORD_EMB: case CS_T_POS_EMB: case CS_T_TYPE_EMB: *shift = 2; break;
        case CS_T_LN_GAMMA: *shift = 3; *base = 1.0f; break;
        case CS_T_LN_BETA: *shift = 3; break;
        case CS_T_QK_W: *shift = 3; break;
        case CS_T_V_W: case CS_T_ATTN_OUT_W: case CS_T_FFN_UP_W: *shift = 4; break;
        case CS_T_FFN_DOWN_W: *shift = 5; break;
        default: *shift = 4; break; /* biases */
    }
}

typedef struct cs_bert_layer_offsets {
    uint64_t q_w, q_b, k_w, k_b, v_w, v_b, ao_w, ao_b, ao_ln_g, ao_ln_b;
    uint64_t up_w, up_b, down_w, down_b, out_ln_g, out_ln_b;
    uint64_t gate_w, gate_b; /* gated feed-forwards only (else == down_w) */
    uint64_t qln_g, qln_b, kln_g, kln_b; /* CS_ARCH_JINA_QKNORM only (else == ao_w) */
} cs_bert_layer_offsets;

typedef struct cs_bert_offsets {
    uint64_t word, pos, type, emb_ln_g, emb_ln_b;
    uint64_t layer0;       /* offset of layer 0 */
    uint64_t layer_stride; /* floats per layer */
    uint64_t final_ln_g, final_ln_b; /* CS_ARCH_MODERN only (else == total) */
    uint64_t total;
} cs_bert_offsets;

/* families without a position table and with a gated feed-forward (a second [I, H] up-projection per layer) */
CS_SYNTH_FN int cs_arch_gated(uint32_t arch) {
    return arch == CS_ARCH_NOMIC || arch == CS_ARCH_JINA || arch == CS_ARCH_JINA_QKNORM || arch == CS_ARCH_MODERN;
}
CS_SYNTH_FN int cs_arch_alibi(uint32_t arch) { return arch == CS_ARCH_JINA || arch == CS_ARCH_JINA_QKNORM; }

CS_SYNTH_FN void cs_bert_layout(const cs_bert_config* c, cs_bert_offsets* o) {
    const uint64_t H = c->hidden, I = c->intermediate;
    uint64_t p = 0;
    o->word = p; p += (uint64_t)c->vocab_size * H;
    o->pos = p; if (!cs_arch_gated(c->arch)) p += (uint64_t)c->max_position * H;
    o->type = p; if (c->arch != CS_ARCH_MODERN) p += (uint64_t)c->type_vocab_size * H;
    o->emb_ln_g = p; p += H;
    o->emb_ln_b = p; p += H;
    o->layer0 = p;
    o->layer_stride = 4 * (H * H + H) + 2 * H + (I * H + I) + (H * I + H) + 2 * H;
    if (cs_arch_gated(c->arch)) o->layer_stride += I * H + I;
    if (c->arch == CS_ARCH_JINA_QKNORM) o->layer_stride += 4 * H;
    o->total = p + (uint64_t)c->layers * o->layer_stride;
    o->final_ln_g = o->final_ln_b = o->total;
    if (c->arch == CS_ARCH_MODERN) { o->final_ln_g = o->total; o->final_ln_b = o->total + H; o->total += 2 * H; }
}

CS_SYNTH_FN void cs_bert_layer_layout(const cs_bert_config* c, const cs_bert_offsets* o,
                                      uint32_t layer, cs_bert_layer_offsets* l) {
    const uint64_t H = c->hidden, I = c->intermediate;
    uint64_t p = o->layer0 + (uint64_t)layer * o->layer_stride;
    l->q_w = p; p += H * H; l->q_b = p; p += H;
    l->k_w = p; p += H * H; l->k_b = p; p += H;
    l->v_w = p; p += H * H; l->v_b = p; p += H;
    l->qln_g = l->qln_b = l->kln_g = l->kln_b = p;
    if (c->arch == CS_ARCH_JINA_QKNORM) { l->qln_g = p; p += H; l->qln_b = p; p += H; l->kln_g = p; p += H; l->kln_b = p; p += H; }
    l->ao_w = p; p += H * H; l->ao_b = p; p += H;
    l->ao_ln_g = p; p += H; l->ao_ln_b = p; p += H;
    l->up_w = p; p += I * H; l->up_b = p; p += I;
    l->gate_w = p; l->gate_b = p;
    if (cs_arch_gated(c->arch)) { p += I * H; l->gate_b = p; p += I; }
    l->down_w = p; p += H * I; l->down_b = p; p += H;
    l->out_ln_g = p; p += H; l->out_ln_b = p; p += H;
}

/* Kind of the tensor that flat offset e falls in (used by the in-place generators). */
CS_SYNTH_FN int cs_bert_kind_at(const cs_bert_config* c, const cs_bert_offsets* o, uint64_t e) {
    const uint64_t H = c->hidden, I = c->intermediate;
    if (e < o->pos) return CS_T_WORD_EMB;
    if (e < o->type) return CS_T_POS_EMB;
    if (e < o->emb_ln_g) return CS_T_TYPE_EMB;
    if (e < o->emb_ln_b) return CS_T_LN_GAMMA;
    if (e < o->layer0) return CS_T_LN_BETA;
    if (e >= o->final_ln_g && o->final_ln_g != o->total) return e < o->final_ln_b ? CS_T_LN_GAMMA : CS_T_LN_BETA;
    uint64_t r = (e - o->layer0) % o->layer_stride;
    const uint64_t lin = H * H + H;
    if (r < 2 * lin) return (r % lin) < H * H ? CS_T_QK_W : CS_T_BIAS;          /* q, k */
    if (r < 3 * lin) return (r - 2 * lin) < H * H ? CS_T_V_W : CS_T_BIAS;       /* v */
    if (c->arch == CS_ARCH_JINA_QKNORM) { /* layer_norm_q, layer_norm_k: gamma, beta, gamma, beta */
        if (r < 3 * lin + 4 * H) return (((r - 3 * lin) / H) & 1) ? CS_T_LN_BETA : CS_T_LN_GAMMA;
        r -= 4 * H;
    }
    if (r < 4 * lin) return (r - 3 * lin) < H * H ? CS_T_ATTN_OUT_W : CS_T_BIAS;
    r -= 4 * lin;
    if (r < H) return CS_T_LN_GAMMA;
    if (r < 2 * H) return CS_T_LN_BETA;
    r -= 2 * H;
    if (r < I * H) return CS_T_FFN_UP_W;
    if (r < I * H + I) return CS_T_BIAS;
    r -= I * H + I;
    if (cs_arch_gated(c->arch)) {
        if (r < I * H) return CS_T_FFN_UP_W;
        if (r < I * H + I) return CS_T_BIAS;
        r -= I * H + I;
    }
    if (r < H * I) return CS_T_FFN_DOWN_W;
    if (r < H * I + H) return CS_T_BIAS;
    r -= H * I + H;
    return r < H ? CS_T_LN_GAMMA : CS_T_LN_BETA;
}

CS_SYNTH_FN float cs_bert_synth_param(const cs_bert_config* c, const cs_bert_offsets* o,
                                      uint64_t seed, uint64_t e) {
    int shift; float base;
    cs_bert_synth_rule(cs_bert_kind_at(c, o, e), &shift, &base);
    return base + cs_synth_weight(seed, e, shift);
}

#endif /* CS_BERT_PARAMS_H */
