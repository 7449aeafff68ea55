/*
 * cs_oracle.h — CPU restatement of the reference's embedding + similarity arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so, and only as the checker / the reported CPU baseline.  libcsgpu.so
 * neither links nor calls it.
 *
 * What it restates (paths relative to /root/reference, Rust, not compilable here:
 * no cargo/rustc in the image, and the arithmetic bottoms out in un-vendored crates —
 * fastembed 5.8.1 / ort 2.0.0-rc.11 / arroy 0.5.0, Cargo.lock):
 *   - cosine_similarity            examples/benchmark_models.rs:323-328
 *                                  src/embed/batch.rs:316-324 (zero guard)
 *   - linear best-match scan       examples/benchmark_models.rs:155-165
 *     generalised from top-1 to top-k with the total order (cosine desc, id asc)
 *   - result/score mapping         src/vectordb/store.rs:464-483
 *   - BERT encoder + pooling       HF BertModel semantics for BAAI/bge-small-en-v1.5 as
 *                                  run by fastembed (call site src/embed/embedder.rs:286-289)
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   scan/top-k : pinned on the reference's own known answers (store.rs:846-893 4-d
 *                case, batch.rs:326-340 3-d cases) and on an independent float64
 *                exhaustive check; the reference publishes no larger vectors.
 *   encoder    : PARITY UNPINNED against the reference (it holds no embedding vectors,
 *                SURVEY.md §4); pinned instead against HF transformers BertModel run in
 *                the build container (tests/golden/make_encoder_golden.py).
 *   quantised encoder (cs_oracle_bert_forward_q8): PARITY UNPINNED against onnxruntime (not installed here); its Linear is
 *                pinned bit for bit to a numpy statement of ONNX DynamicQuantizeLinear / MatMulInteger
 *                (tests/test_oracle_encoder.py).
 */
#ifndef CS_ORACLE_H
#define CS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scan / top-k ------------------------------------------------------------- */

/* f32, sequential left-to-right sums, one sqrt per magnitude, dot / (mag_a * mag_b);
 * returns 0.0 when either magnitude is 0 (batch.rs:320-322). */
float cs_oracle_cosine(const float* a, const float* b, size_t dim);
/* Same in float64 (independent checker). */
double cs_oracle_cosine_f64(const float* a, const float* b, size_t dim);

/* Literal scan: for each row in id order compute cs_oracle_cosine(q, row) and keep the
 * best k under (cosine desc, id asc).  `dead` (may be NULL) is a bitmap over rows,
 * bit set = tombstoned.  NaN cosines are never selected (a NaN fails the reference's
 * `score > best_score`, benchmark_models.rs:160).  Returns the number of results
 * written (<= k).  ids written are id_base + row.  Single thread, scalar. */
uint32_t cs_oracle_scan_topk(const float* corpus, uint64_t n, uint32_t dim, const float* q,
                             uint32_t k, const uint32_t* dead, uint32_t id_base,
                             float* out_cos, uint32_t* out_ids);
/* float64 accumulation version; out_cos receives the f64 cosine. */
uint32_t cs_oracle_scan_topk_f64(const float* corpus, uint64_t n, uint32_t dim,
                                 const float* q, uint32_t k, const uint32_t* dead,
                                 uint32_t id_base, double* out_cos, uint32_t* out_ids);
/* Tuned CPU port: OpenMP over `threads` (0 = all), contiguous matrix, vectorisable
 * 8-lane partial sums, per-thread top-k + merge.  Same selection rule; cosines may
 * differ from the literal scan in the last bits (different summation order). */
uint32_t cs_oracle_scan_topk_omp(const float* corpus, uint64_t n, uint32_t dim,
                                 const float* q, uint32_t k, const uint32_t* dead,
                                 uint32_t id_base, int threads, float* out_cos,
                                 uint32_t* out_ids);
/* Merge `nlists` lists of (cos,id) of length k each (count[i] valid) into the best k. */
uint32_t cs_oracle_merge_topk(const float* cos, const uint32_t* ids, const uint32_t* counts,
                              uint32_t nlists, uint32_t k, float* out_cos,
                              uint32_t* out_ids);
int cs_oracle_num_threads(void);

/* ---- encoder (bert_oracle.c) ------------------------------------------------------- */
struct cs_bert_config; /* include/codesearch_gpu.h */
/* fp32 BertModel forward + pooling + L2 normalise.  ids/mask [B,L] i32; hidden_out
 * (optional) [B*L*H]; pooled_out (optional) [B,H]; layer_hidden_out (optional)
 * [(layers+1)*B*L*H]: embedding output followed by each layer's output. */
void cs_oracle_bert_forward(const struct cs_bert_config* cfg, const float* params,
                            const int32_t* ids, const int32_t* mask, uint32_t B, uint32_t L,
                            float* hidden_out, float* pooled_out, float* layer_hidden_out);
/* Dynamic-quantised Linear layers (the registry's *Q models, onnxruntime quantize_dynamic files): `params` holds the
 * dequantised weights (W_q - W_zp) * W_scale; wscale [layers][5H + I] gives each output column's scale in the order
 * query | key | value | attention.output | intermediate | output (include/cs_bert_params.h).  Every Linear
 * = DynamicQuantizeLinear(x over the whole [B*L, K] tensor) -> MatMulInteger -> * (x_scale * w_scale) -> + bias. */
void cs_oracle_bert_forward_q8(const struct cs_bert_config* cfg, const float* params, const float* wscale,
                               const int32_t* ids, const int32_t* mask, uint32_t B, uint32_t L,
                               float* hidden_out, float* pooled_out, float* layer_hidden_out);
/* y[T,N] = one such Linear: x [T,K] f32, w [N,K] the dequantised weight, wscale [N], b [N]. */
void cs_oracle_linear_q8(const float* x, const float* w, const float* wscale, const float* b, float* y, uint64_t T,
                         uint64_t K, uint64_t N);
/* Flat parameter block from the synthetic rule of include/cs_bert_params.h. */
/* JinaBert's ALiBi head slopes (CS_ARCH_JINA*), n floats. */
void cs_oracle_alibi_slopes(uint32_t n, float* out);
void cs_oracle_bert_synth_params(const struct cs_bert_config* cfg, uint64_t seed, float* out);
uint64_t cs_oracle_bert_param_count(const struct cs_bert_config* cfg);

/* ---- synthetic data (include/cs_synth.h) ------------------------------------------ */
void cs_oracle_synth_rows(uint64_t seed, uint64_t first_row, uint64_t n, uint32_t dim,
                          float* out);
void cs_oracle_synth_planted(uint64_t seed_c, uint64_t seed_q, const uint64_t* rows,
                             uint64_t nq, uint32_t dim, float* out);

#ifdef __cplusplus
}
#endif
#endif
