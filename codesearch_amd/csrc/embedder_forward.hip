// embedder_forward.hip — one mini-batch through the encoder kernels: workspace, the per-layer launch sequence of every GEMM mode and encoder family (forward_range), stream slicing (forward).
// (one of the translation units behind cs_embedder_*: see embedder_state.hpp)
#include "embedder_state.hpp"

using namespace cs;

namespace cs {
namespace emb {

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
    a.pooling = c.pooling; a.x = x; a.out = (h->pooled_dst ? h->pooled_dst : h->d_pooled) + (size_t)b0 * H;
    a.xs = (split && !q8) ? (void*)(h->d_xs + t0 * H) : nullptr;  // q8: the xs buffer holds the quantised rows instead
    a.flag = h->d_flag;
    if (q8) a.range_out = h->d_range_pairs;  // LayerNorm leaves its blocks' ranges for the quantising pass that follows
    const uint32_t ln_pairs = (T + 3) / 4;
    // several quantisation units in a batch the row-block kernels take: every product quantises its own rows with their
    // unit's parameters, the producers' pairs are reduced per unit (LayerNorm: a pair per row)
    static const bool q8_mu_on = [] { const char* e = cs_lab_env("CS_Q8_ROWS_UNITS"); return !(e && e[0] == '0'); }();
    const bool q8_mu = q8 && q8_mu_on && h->cur_units > 1 && q8_rows_from_source(T, H) && T <= h->cap_range_pairs;
    a.range_rows = q8_mu;
    _Float16* xs = reinterpret_cast<_Float16*>(h->d_xs + t0 * H);
    _Float16* ctxs = reinterpret_cast<_Float16*>(ctx);
    _Float16* mids = reinterpret_cast<_Float16*>(mid);
    const SplitLayer sl = split_layer(c);
    static const uint32_t split_k_min = [] { const char* e = cs_lab_env("CS_GEMM_SPLITK_MIN_M"); return e ? (uint32_t)std::atoi(e) : 1100u; }();
    static const uint32_t split_k_max = [] { const char* e = cs_lab_env("CS_GEMM_SPLITK_MAX_M"); return e ? (uint32_t)std::atoi(e) : 6144u; }();
    // device us per forward, fused / FFN-down in 3 K slices / out-proj too: 1,280 rows 1320 / 1020 / 971, 2,048
    // 1331 / 1052 / 1021, 4,096 1538 / 1311 / 1328, 6,144 1841 / 1619 / 1654, 8,192 2210 / 2264 / -
    // two slices up to 10,240 rows: 7,168 rows 2048 -> 1891 us, 8,192 2203 -> 2060, 10,240 2443 -> 2369, 12,288 3034 -> 3167
    static const uint32_t split_k_max2 = [] { const char* e = cs_lab_env("CS_GEMM_SPLITK_MAX2_M"); return e ? (uint32_t)std::atoi(e) : 10240u; }();
    static const uint32_t split_k_ao_max = [] { const char* e = cs_lab_env("CS_GEMM_SPLITK_AO_MAX_M"); return e ? (uint32_t)std::atoi(e) : 2560u; }();
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
    static const uint32_t wide_min_m = [] { const char* e = cs_lab_env("CS_GEMM_WIDE_MIN_M"); return e ? (uint32_t)std::atoll(e) : 12288u; }();
    auto takes_wide = [&](uint32_t Mr, uint32_t Nn, uint32_t Kk) {
        if (!h->wide_ok || !wide_min_m || !gemm_wide_supported(Nn, Kk) || Nn % 384) return false;
        const uint32_t tiles = ((Mr + 127) / 128) * (Nn / 384);
        return tiles >= 218 || (h->streams_in_flight >= 2 && Mr >= wide_min_m);
    };
    // Mid-size launches (the reference's 32-chunk calls: 8,192 token rows): the 128 x 128 grid is 1.1 rounds of
    // blocks for QKV (576 tiles on 512 slots); 128 x 192 tiles at two blocks per CU make it ONE round (384 tiles for
    // QKV, 512 for FFN-up).  Taken when that single round is at least 70 % full.
    static const bool mid192 = [] { const char* e = cs_lab_env("CS_GEMM_WIDE_MID"); return !(e && e[0] == '0'); }();
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
    // ---- ModernBERT (CS_ARCH_MODERN): pre-norm layers -----------------------------------------------------------------------------
    // x is the residual stream and is only ever added to: x += Wo attention(rope(Wqkv LN_attn(x))) (layer 0 takes the embedding
    // LayerNorm's output as it is); x += Wo_mlp(gelu(Wi_a LN_mlp(x)) * Wi_b LN_mlp(x)); a final LayerNorm in front of the
    // pooling.  The LayerNorm outputs go to the context buffer (f32, free at both points) and to xs in split form; the rotary
    // table and the attention window follow the layer's type (global every `global_every`-th layer, local otherwise); the
    // gate is the up projection's epilogue at indexing sizes (GW_OUT_GEGLU, as for JinaBert: value = the half of Wi that is
    // not activated).  The same kernels as every other family; exact-f32 mode included.
    if (c.arch == CS_ARCH_MODERN) {
        if (q8) return fail(CS_ERR_UNSUPPORTED, "the dynamic-quantisation mode is not built for the ModernBERT encoder");
        a.pos = nullptr; a.type0 = h->d_zero_row;
        if (!split) a.xs = nullptr;
        CS_TRY(mark(-1));
        CS_TRY(launch_row_kernel(0, a, H, s));  // E1 (its split copy is layer 0's operand: attn_norm is the identity there)
        CS_TRY(mark(CS_STAGE_EMBED_LN));
        _Float16* qkvs = reinterpret_cast<_Float16*>(qkv);
        EncoderLaunch n = a;   // LayerNorm of the residual stream into the context buffer (+ xs)
        n.src = x; n.x = ctx;
        for (uint32_t l = 0; l < c.layers; ++l) {
            cs_bert_layer_offsets lo;
            cs_bert_layer_layout(&c, &h->off, l, &lo);
            const bool global = c.global_every == 0 || l % c.global_every == 0;
            const float2* rope = global ? h->d_rope : h->d_rope_local;
            const uint32_t window = global ? 0u : c.local_window;
            const float* bqkv = h->d_bqkv + (size_t)l * 3 * H;
            if (l) {
                n.g = P + lo.ao_ln_g; n.b = P + lo.ao_ln_b;
                CS_TRY(launch_row_kernel(4, n, H, s));  // attn_norm
            }
            if (split) {
                const _Float16* ws = h->d_wsplit + (size_t)l * sl.total;
                CS_TRY(dense(SH_OUT_SPLIT, xs, ws + sl.qkv, bqkv, nullptr, nullptr, qkvs, T, 3 * H, H));  // E2
                CS_TRY(launch_rope_split(qkvs, rope, T, L, H, c.heads, h->d_flag, s));
                CS_TRY(mark(CS_STAGE_QKV));
                CS_TRY(launch_attention_sh2(qkvs, mask, ctxs, h->d_flag, nb, L, H, c.heads, s, nullptr, nullptr, nullptr, nullptr, nullptr, window));  // E3
                CS_TRY(mark(CS_STAGE_ATTENTION));
                CS_TRY(dense(SH_OUT_F32_RESID, ctxs, ws + sl.ao, P + lo.ao_b, x, x, nullptr, T, H, H));  // E4: x += Wo ctx
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
                n.g = P + lo.out_ln_g; n.b = P + lo.out_ln_b;
                CS_TRY(launch_row_kernel(4, n, H, s));  // mlp_norm
                CS_TRY(mark(CS_STAGE_LN_ATTN));
                _Float16* gated = reinterpret_cast<_Float16*>(mid + (size_t)T * 2 * I);
                const float* bup = h->d_bup + (size_t)l * 2 * I;
                const bool w384 = takes_wide(T, 2 * I, H), w192 = !w384 && takes_192(T, 2 * I, H);
                if (w384 || w192) {
                    CS_TRY(launch_gemm_wide(GW_OUT_GEGLU, xs, ws + sl.up, bup, nullptr, nullptr, gated, T, 2 * I, H, h->d_flag, s, w192 ? 192 : 0));
                } else {
                    CS_TRY(dense(SH_OUT_SPLIT, xs, ws + sl.up, bup, nullptr, nullptr, mids, T, 2 * I, H));
                    CS_TRY(launch_swiglu_split(mids, gated, T, I, h->d_flag, s, true));
                }
                CS_TRY(mark(CS_STAGE_FFN_UP));
                CS_TRY(dense(SH_OUT_F32_RESID, gated, ws + sl.down, P + lo.down_b, x, x, nullptr, T, H, I));  // E6: x += Wo_mlp(...)
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
            } else {
                const float* nin = l ? ctx : x;  // layer 0: the embedding LayerNorm's output itself
                const float* wqkv = h->d_wqkv + (size_t)l * 3 * H * H;
                CS_TRY(launch_gemm(GEMM_BIAS, nin, wqkv, bqkv, nullptr, qkv, T, 3 * H, H, s));
                CS_TRY(launch_rope_f32(qkv, rope, T, L, H, c.heads, s));
                CS_TRY(mark(CS_STAGE_QKV));
                CS_TRY(launch_attention(qkv, mask, ctx, nb, L, H, c.heads, s, nullptr, window));
                CS_TRY(mark(CS_STAGE_ATTENTION));
                CS_TRY(launch_gemm(GEMM_RESID, ctx, P + lo.ao_w, P + lo.ao_b, x, x, T, H, H, s));
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
                n.g = P + lo.out_ln_g; n.b = P + lo.out_ln_b;
                CS_TRY(launch_row_kernel(4, n, H, s));
                CS_TRY(mark(CS_STAGE_LN_ATTN));
                float* gate = mid + (size_t)T * I;
                CS_TRY(launch_gemm(GEMM_BIAS, ctx, P + lo.up_w, P + lo.up_b, nullptr, mid, T, I, H, s));
                CS_TRY(launch_gemm(GEMM_BIAS, ctx, P + lo.gate_w, P + lo.gate_b, nullptr, gate, T, I, H, s));
                CS_TRY(launch_swiglu_f32(mid, gate, T, I, s, true));
                CS_TRY(mark(CS_STAGE_FFN_UP));
                CS_TRY(launch_gemm(GEMM_RESID, mid, P + lo.down_w, P + lo.down_b, x, x, T, H, I, s));
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
            }
        }
        a.g = P + h->off.final_ln_g; a.b = P + h->off.final_ln_b; a.xs = nullptr;
        CS_TRY(launch_row_kernel(1, a, H, s));  // final_norm, in place
        CS_TRY(mark(CS_STAGE_LN_FFN));
        h->last_hidden_partial = false;
        CS_TRY(launch_row_kernel(2, a, H, s));  // E7 + E8
        CS_TRY(mark(CS_STAGE_POOL));
        return CS_OK;
    }
    // ---- a few short sequences (under 200 token rows: the query side) ----
    // small_path.hip: LayerNorm as the prologue of the dense layer that reads it, FFN-down as four K slices summed by the
    // LayerNorm that follows: 62 launches per 12-layer forward instead of 86, none of them pulling 196 KB through one CU
    // (CS_SMALL_PATH=0: the general small-batch kernels below).  Diagnostic library, CS_SMALL_FORWARD=1: the same arithmetic as ONE
    // launch (small_forward.hip) — bit-identical, measured slower than the launches.
    h->sf_ran = false;
    const char* e0 = std::getenv("CS_SMALL_PATH");  // (read per forward: tests flip it mid-process)
    const bool sp_on = !(e0 && e0[0] == '0');
    if (mode == CS_GEMM_SPLIT_F16 && sp_on && !nomic && b0 == 0 && T < 200 && small_path_supported(H, I, T)) {
        if (!h->d_sp_ws) CS_HIP(hipMalloc(&h->d_sp_ws, (size_t)5 * SP_MAX_ROWS * H * sizeof(float)));
        float* parts = h->d_sp_ws;                                   // [4][T][H]
        float* xa = h->d_sp_ws + (size_t)4 * SP_MAX_ROWS * H;        // [T][H]
        float* y = h->d_xs + t0 * H;                                  // [T][H] (the split copy of x is not used on this path)
        _Float16* qkvs = reinterpret_cast<_Float16*>(qkv);
#ifdef CS_DIAGNOSTICS
        const char* e1 = cs_lab_env("CS_SMALL_FORWARD");  // (read per forward: tests and A/B runs flip it mid-process)
        if (e1 && e1[0] == '1' && h->d_sf_layers && !h->sf_off && !h->stage_profile && small_forward_supported(H, I, c.heads, T, L)) {
            uint32_t hb = L <= 32 ? 4u : (L <= 64 ? 2u : 1u);  // heads per attention block, as launch_attention_sh2 packs them
            if (const char* ph = cs_lab_env("CS_ATTN_PACK_HEADS")) if (ph[0] == '0') hb = 1;
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
#endif
        _Float16* ctxs2 = ctxs;
        _Float16* mids2 = reinterpret_cast<_Float16*>(mid);
        // sequences of up to 32 tokens (a query and its variants) of a 384-wide model: attention inside the out-projection's blocks, 50
        // instead of 62 launches per 12-layer forward.  CS_SMALL_FUSE=0 (read per forward: the tests compare the two): the two launches.
        const char* e2 = std::getenv("CS_SMALL_FUSE");
        const bool sp_fused = !(e2 && e2[0] == '0') && sp_attn_proj_supported(H, c.heads, T, L);
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
            if (sp_fused) {  // E3 + E4 -> y in one launch: every out-projection block computes its rows' attention itself (small_path.hip)
                CS_TRY(mark(CS_STAGE_ATTENTION));
                CS_TRY(launch_sp_attn_proj(qkvs, mask, ws + sl2.ao, P + lo.ao_b, xa, y, T, L, H, c.heads, h->d_flag, s));
            } else {
                CS_TRY(launch_attention_sh2(qkvs, mask, ctxs2, h->d_flag, nb, L, H, c.heads, s));                      // E3
                CS_TRY(mark(CS_STAGE_ATTENTION));
                CS_TRY(launch_gemm_split(SH_OUT_F32_RESID, ctxs2, ws + sl2.ao, P + lo.ao_b, xa, y, nullptr, T, H, H, h->d_flag, s));  // E4 -> y
            }
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
            const int8_t* wst = h->d_wq8_stages ? h->d_wq8_stages + (size_t)l * ((size_t)H * H + (size_t)H * I) : nullptr;  // (out-proj | FFN-down, stage-major)
            const uint32_t* cmt = (H % 128 == 0 && I % 128 == 0) ? h->d_cmeta_tiles + (size_t)l * (5 * (size_t)H + I) * 4 : nullptr;  // (slab kernel)
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
                // Up to 16 rows of a 384-wide model (one short query): the two LayerNorms of a layer are the prologues of the products
                // that read them (Q8_SRC_LN: the block's 16 rows are the whole tensor, so it knows the range) — five launches per layer
                // instead of seven.  The products behind attention and GELU then write the PRE-norm rows to ybuf and add the
                // normalised ones (x, written by the prologue's column-tile-0 blocks) as their residual.  Same arithmetic, same bits.
                // CS_Q8_SKINNY_LN=0 (read per forward: the tests compare the two): the LayerNorm launches.
                const char* e3 = std::getenv("CS_Q8_SKINNY_LN");
                const bool fold = !(e3 && e3[0] == '0') && T <= 16 && H == 384;
                float* ybuf = h->d_xs + t0 * H;              // [T][H] f32 (the split copy of x is not used on this path)
                const bool last = l + 1 == c.layers;
                if (fold && l) {
                    cs_bert_layer_offsets lp;
                    cs_bert_layer_layout(&c, &h->off, l - 1, &lp);
                    CS_TRY(launch_gemm_q8_skinny_ln(SH_OUT_SPLIT, ybuf, P + lp.out_ln_g, P + lp.out_ln_b, c.layer_norm_eps, x, wq + ql.qkv, cm, qkvs, T, 3 * H,
                                                    h->d_flag, nullptr, nullptr, s));  // LN (layer l - 1's second) + E2
                } else {
                    CS_TRY(launch_gemm_q8_skinny(SH_OUT_SPLIT, Q8_SRC_F32, x, rp, ln_pairs, wq + ql.qkv, cm, nullptr, nullptr, qkvs, T, 3 * H, H, h->d_flag,
                                                 nullptr, nullptr, s));  // E2
                }
                CS_TRY(mark(CS_STAGE_QKV));
                uint32_t att_pairs = 0, up_pairs = 0;
                CS_TRY(launch_attention_sh2(qkvs, mask, ctxs, h->d_flag, nb, L, H, c.heads, s, rp, &att_pairs));  // E3
                CS_TRY(mark(CS_STAGE_ATTENTION));
                if (!att_pairs) return fail(CS_ERR_UNSUPPORTED, "attention kernel without range pairs in the few-rows quantised path");
                CS_TRY(launch_gemm_q8_skinny(SH_OUT_F32_RESID, Q8_SRC_SPLIT, ctxs, rp, att_pairs, wq + ql.ao, cm + 3 * H, x, fold ? ybuf : x, nullptr, T, H, H,
                                             h->d_flag, nullptr, nullptr, s));  // E4
                CS_TRY(mark(CS_STAGE_OUT_PROJ));
                if (fold) {
                    CS_TRY(mark(CS_STAGE_LN_ATTN));
                    CS_TRY(launch_gemm_q8_skinny_ln(SH_OUT_SPLIT_GELU, ybuf, P + lo.ao_ln_g, P + lo.ao_ln_b, c.layer_norm_eps, x, wq + ql.up, cm + 4 * H, mids, T, I,
                                                    h->d_flag, rp2, &up_pairs, s));  // LN + E5
                } else {
                    a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
                    CS_TRY(launch_row_kernel(1, a, H, s));
                    CS_TRY(mark(CS_STAGE_LN_ATTN));
                    CS_TRY(launch_gemm_q8_skinny(SH_OUT_SPLIT_GELU, Q8_SRC_F32, x, rp, ln_pairs, wq + ql.up, cm + 4 * H, nullptr, nullptr, mids, T, I, H,
                                                 h->d_flag, rp2, &up_pairs, s));  // E5
                }
                CS_TRY(mark(CS_STAGE_FFN_UP));
                CS_TRY(launch_gemm_q8_skinny(SH_OUT_F32_RESID, Q8_SRC_SPLIT, mids, rp2, up_pairs, wq + ql.down, cm + 4 * H + I, x, fold && !last ? ybuf : x, nullptr,
                                             T, H, I, h->d_flag, nullptr, nullptr, s));  // E6
                CS_TRY(mark(CS_STAGE_FFN_DOWN));
                if (!fold || last) {  // (folded: the next layer's first product normalises ybuf)
                    a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
                    CS_TRY(launch_row_kernel(1, a, H, s));
                }
                CS_TRY(mark(CS_STAGE_LN_FFN));
                if (l + 1 == c.layers) h->last_hidden_partial = false;
                continue;
            }
            if (!rs && q8_rows_from_source(T, H)) {
                // one unit, K = 384, a row block per CU: the products quantise their own rows on the way in — per tensor only
                // its range is needed first (a reduction of the pairs its producer left).  x_pairs: how many pairs the
                // kernel that wrote x left (LayerNorm: one per four rows; the LayerNorm-fused products: one per sixteen)
                // (q8_x_pairs == 0: the LayerNorm-fused product that wrote x widened this tensor's slot itself — CS_Q8_LN_SLOT=1; measured:
                // what the consumers save on the reduction launch, 4 us each, the producers pay for the block's meeting and its
                // agent-scope update, profiles/r05_q8_ln_epilogue_ab.log: opt-in.  Default: pairs + a reduction launch)
                static const bool ln_slot = [] { const char* e = cs_lab_env("CS_Q8_LN_SLOT"); return e && e[0] == '1'; }();
                if (l == 0) h->q8_x_pairs = ln_pairs;
                if (h->q8_x_pairs) CS_TRY(launch_q8_range(Q8_SRC_F32, x, T, H, rg, s, rp, h->q8_x_pairs));
                CS_TRY(launch_gemm_q8_from_source(SH_OUT_SPLIT, Q8_SRC_F32, x, rg, wq + ql.qkv, cm, bqkv, nullptr, nullptr, qkvs, T, 3 * H, H, h->d_flag, s, nullptr, cmt,
                                                  xq));  // E2 (xq: scratch for the quantised rows of a call of few slabs)
                CS_TRY(mark(CS_STAGE_QKV));
                uint32_t att_pairs = 0;
                CS_TRY(launch_attention_sh2(qkvs, mask, ctxs, h->d_flag, nb, L, H, c.heads, s, rp, &att_pairs));  // E3
                CS_TRY(mark(CS_STAGE_ATTENTION));
                CS_TRY(launch_q8_range(Q8_SRC_SPLIT, ctxs, T, H, rg + rstep, s, rp, att_pairs));
                if (q8_ln_fused_takes(T, H, H)) {  // E4 with its residual add and LayerNorm in one kernel (gemm_q8_ln_kernel)
                    CS_TRY(launch_gemm_q8_ln(Q8_SRC_SPLIT, ctxs, nullptr, rg + rstep, wst ? wst : wq + ql.ao, cm + 3 * H, x, P + lo.ao_ln_g, P + lo.ao_ln_b,
                                             c.layer_norm_eps, T, H, rp, &h->q8_x_pairs, s, ln_slot ? rg + 2 * rstep : nullptr, wst != nullptr));
                    CS_TRY(mark(CS_STAGE_OUT_PROJ));
                } else {
                    CS_TRY(launch_gemm_q8_from_source(SH_OUT_F32_RESID, Q8_SRC_SPLIT, ctxs, rg + rstep, wq + ql.ao, cm + 3 * H, P + lo.ao_b, x, x, nullptr, T, H, H,
                                                      h->d_flag, s));  // E4
                    CS_TRY(mark(CS_STAGE_OUT_PROJ));
                    a.g = P + lo.ao_ln_g; a.b = P + lo.ao_ln_b;
                    CS_TRY(launch_row_kernel(1, a, H, s));
                    h->q8_x_pairs = ln_pairs;
                }
                CS_TRY(mark(CS_STAGE_LN_ATTN));
                if (h->q8_x_pairs) CS_TRY(launch_q8_range(Q8_SRC_F32, x, T, H, rg + 2 * rstep, s, rp, h->q8_x_pairs));
                int8_t* midq = reinterpret_cast<int8_t*>(mid);
                Q8RowMeta* rm2 = h->d_rmeta2 + t0;
                CS_TRY(launch_gemm_q8_gelu_requant_from_source(x, rg + 2 * rstep, wq + ql.up, cm + 4 * H, P + lo.up_b, T, I, H, rg + 3 * rstep, midq, rm2, s, nullptr,
                                                               cmt ? cmt + 4 * 4 * H : nullptr, xq));  // E5 (xq: the range pass's quantised rows for the store pass)
                CS_TRY(mark(CS_STAGE_FFN_UP));
                if (q8_ln_fused_takes(T, H, I)) {  // E6 likewise
                    // (the next layer's first slot; the last layer's output is not quantised again: pairs nobody reads)
                    CS_TRY(launch_gemm_q8_ln(Q8_SRC_PREQUANT, midq, rm2, nullptr, wst ? wst + (size_t)H * H : wq + ql.down, cm + 4 * H + I, x, P + lo.out_ln_g, P + lo.out_ln_b,
                                             c.layer_norm_eps, T, I, rp, &h->q8_x_pairs, s, ln_slot && l + 1 < c.layers ? rg + 4 * rstep : nullptr, wst != nullptr));
                    CS_TRY(mark(CS_STAGE_FFN_DOWN));
                } else {
                    CS_TRY(launch_gemm_q8(SH_OUT_F32_RESID, midq, rm2, wq + ql.down, cm + 4 * H + I, P + lo.down_b, x, x, nullptr, T, H, I, h->d_flag, s));  // E6
                    CS_TRY(mark(CS_STAGE_FFN_DOWN));
                    a.g = P + lo.out_ln_g; a.b = P + lo.out_ln_b;
                    CS_TRY(launch_row_kernel(1, a, H, s));
                    h->q8_x_pairs = ln_pairs;
                }
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
            static const bool ln_fuse_on = [] { const char* e = cs_lab_env("CS_GEMM_WIDE_LN"); return !(e && e[0] == '0'); }();
            const bool fuse_ln = ln_fuse_on && H == 384 && takes_wide(T, H, H);
            static const bool split_resid_on = [] { const char* e = cs_lab_env("CS_GEMM_WIDE_LN_SPLIT_RESID"); return !(e && e[0] == '0'); }();
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
                static const bool gate_fused = [] { const char* e = cs_lab_env("CS_NOMIC_GATE_FUSED"); return !(e && e[0] == '0'); }();
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
        const char* e = cs_lab_env("CS_ENCODER_STREAM_MIN_TOKENS");
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
}  // namespace emb
}  // namespace cs
