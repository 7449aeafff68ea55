// encoder.hpp — launch interface of the encoder kernels (encoder.hip).
#pragma once

#include "common.hpp"

#include "../../include/cs_bert_params.h"

namespace cs {

// Arguments of the row-wise kernels (which: 0 = embeddings+LN, 1 = LN in place,
// 2 = pool + L2-normalise).
struct EncoderLaunch {
    const int32_t* ids = nullptr;
    const int32_t* mask = nullptr;
    const float* word = nullptr;
    const float* pos = nullptr;
    const float* type0 = nullptr;
    const float* g = nullptr;
    const float* b = nullptr;
    float eps = 1e-12f;
    uint32_t T = 0, L = 0, B = 0, vocab = 0;
    int pooling = CS_POOL_CLS;
    float* x = nullptr;
    float* out = nullptr;
    void* xs = nullptr;         // optional: the row also in split-f16 form (split_f16.hpp), [T][H/32][64] f16
    const float* parts = nullptr;  // which == 3: [nparts][T][H] partial sums of a split-K GEMM
    const float* bias = nullptr;   //             the layer's bias ([H])
    uint32_t nparts = 0;
    const float* src = nullptr;    // which == 4: LayerNorm of src -> x (and xs), src left as it is (pre-norm families)
    uint32_t* flag = nullptr;   // split-f16 overflow flag (device)
    float* range_out = nullptr; // which 0 / 1, optional: [ceil(T / 4)][2] — each block's (lo, hi) of the rows it wrote
    bool range_rows = false;    //   ... or [T][2], a pair per token row (several quantisation units in the batch)
};

enum { GEMM_BIAS = 0, GEMM_GELU = 1, GEMM_RESID = 2 };

int32_t launch_row_kernel(int which, const EncoderLaunch& a, uint32_t H, hipStream_t s);
// C[M,N] = A[M,K] W[N,K]^T + bias, then epilogue (GEMM_*); resid is [M,N].
int32_t launch_gemm(int epi, const float* A, const float* W, const float* bias, const float* resid,
                    float* C, uint32_t M, uint32_t N, uint32_t K, hipStream_t s);
// qkv [B*L, 3H] (Q | K | V), mask [B, L] -> ctx [B*L, H]
// alibi (optional, device, [heads]): JinaBert's head slopes — the score of (query i, key j) gets -slope_h |i - j|
// window (0 = none): ModernBERT's local layers — keys with |i - j| > window are masked like padding
int32_t launch_attention(const float* qkv, const int32_t* mask, float* ctx, uint32_t B, uint32_t L,
                         uint32_t H, uint32_t heads, hipStream_t s, const float* alibi = nullptr, uint32_t window = 0);
size_t attention_lds_bytes(uint32_t L);
// attention_split.hip: the same attention on the f16 MFMA with split-f16 operands:
// attention on a split-f16 qkv [T][3H/32][64] (the QKV GEMM's SH_OUT_SPLIT output): K/V go to LDS by
// LDS-DMA with no conversion, V is consumed through the transposing LDS read; writes ctx in split form.
// range_out (optional, with range_pairs): every wave's (lo, hi) of the values it stored — *range_pairs pairs are written,
// sequence by sequence (*range_pairs / B each; 0 when the kernel that ran does not report them: the caller then takes its
// own range pass).  seq_unit / unit_len (optional, device): sequence b belongs to quantisation unit seq_unit[b], whose own
// padded length is unit_len[unit] — query rows at or beyond it are kept out of the pairs.
int32_t launch_attention_sh2(const _Float16* qkv_split, const int32_t* mask, void* ctx_split, uint32_t* flag,
                             uint32_t B, uint32_t L, uint32_t H, uint32_t heads, hipStream_t s, float* range_out = nullptr,
                             uint32_t* range_pairs = nullptr, const uint32_t* seq_unit = nullptr,
                             const uint32_t* unit_len = nullptr, const float* alibi = nullptr, uint32_t window = 0);

// Split-f16 GEMM (gemm_split.hip): A [M][K/32][64] f16, W [N][K/32][64] f16 (split_f16.hpp).
enum { SH_OUT_F32 = 0, SH_OUT_F32_RESID = 1, SH_OUT_SPLIT_GELU = 2, SH_OUT_SPLIT = 3,
       SH_OUT_PARTIAL = 4 /* raw f32 partial sums of a K slice, no bias: slab blockIdx-slice of C */ };
int32_t launch_gemm_split_partial(const _Float16* A, const _Float16* W, float* Cpart, uint32_t M, uint32_t N,
                                  uint32_t K, uint32_t ksplit, hipStream_t s);
int32_t launch_gemm_split(int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid,
                          float* C, _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag,
                          hipStream_t s);
// gemm_wide.hip: the same product as a persistent kernel over 128 x 384 tiles with one accumulator per output
// (w_hi scaled by 2^11 in registers).  N % 384 == 0; weights must pass sh_weights_fit_wide (|w| < 31.98).
bool gemm_wide_supported(uint32_t N, uint32_t K);
// epilogue of launch_gemm_wide only: W = value and gate rows interleaved in groups of 16 ([N][K], N = 2 x gated width);
// Cs [M][N/64][64] = value * silu(gate) in split form (gemm_wide.hip; nomic.hip holds the stand-alone form)
constexpr int GW_OUT_SWIGLU = 17;
constexpr int GW_OUT_GEGLU = 18;   // the same with value * gelu_erf(gate): JinaBert's feed-forward
// shape: 0 = CS_GEMM_WIDE_SHAPE / default (128 x 384 where N allows), 192 = the 128 x 192 two-blocks-per-CU shape, 384
int32_t launch_gemm_wide(int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid, float* C,
                         _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s, int shape = 0);
// N = 384 only: dense layer + bias + residual + LayerNorm in one kernel; X (f32, may alias resid; null = not written)
// and Xs (split form).  resid_split (optional): the residual in split form instead of `resid` (may be Xs).
int32_t launch_gemm_wide_ln(const _Float16* A, const _Float16* W, const float* bias, const float* resid, const float* gamma,
                            const float* beta, float eps, float* X, _Float16* Xs, uint32_t M, uint32_t K, uint32_t* d_flag,
                            hipStream_t s, const _Float16* resid_split = nullptr);
// cls_tail.hip: the last layer of a CLS-pooled model restricted to the B rows the embedding reads — attention of the CLS
// query q_cls [B][H/32][64] against every key of kv_split [T][2H/32][64] (plain f32 on the split operands) -> ctxs_cls, and the gather of the residual
// stream's CLS rows -> x_cls [B, H] f32 + xs_cls split.
int32_t launch_attention_cls(const _Float16* q_cls, const _Float16* kv_split, const int32_t* mask, _Float16* ctxs_cls,
                             uint32_t* flag, uint32_t B, uint32_t L, uint32_t H, uint32_t heads, hipStream_t s);
int32_t launch_gather_cls(const _Float16* xs, float* x_cls, _Float16* xs_cls, uint32_t B, uint32_t L, uint32_t H, hipStream_t s);
// nomic.hip (CS_ARCH_NOMIC): the rotary position map on the Q and K columns of a QKV tensor, in place — split form
// [T][3H/32][64] or f32 [T][3H]; rope [L_max][d_h / 2] (cos, sin) — and the feed-forward gate value * silu(gate): up2
// [T][2I/32][64] (every 128-byte line: 16 values | 16 gates, the interleaved row order of GW_OUT_SWIGLU's weight) -> out
// [T][I/32][64], or value [T][I] *= silu(gate [T][I]).
int32_t launch_rope_split(_Float16* qkvs, const float2* rope, uint32_t T, uint32_t L, uint32_t H, uint32_t heads,
                          uint32_t* flag, hipStream_t s);
int32_t launch_rope_f32(float* qkv, const float2* rope, uint32_t T, uint32_t L, uint32_t H, uint32_t heads, hipStream_t s);
// gelu_gate: value * gelu_erf(gate) instead (CS_ARCH_JINA*)
int32_t launch_swiglu_split(const _Float16* up2, _Float16* out, uint32_t T, uint32_t I, uint32_t* flag, hipStream_t s,
                            bool gelu_gate = false);
int32_t launch_swiglu_f32(float* value, const float* gate, uint32_t T, uint32_t I, hipStream_t s, bool gelu_gate = false);
// CS_ARCH_JINA_QKNORM: LayerNorm over the whole query row and the whole key row of a QKV tensor, in place (split form
// [T][3H/32][64] or f32 [T][3H]); ln = gamma_q | beta_q | gamma_k | beta_k, [4][H]
int32_t launch_qk_layernorm_split(_Float16* qkvs, const float* ln, float eps, uint32_t T, uint32_t H, uint32_t* flag, hipStream_t s);
int32_t launch_qk_layernorm_f32(float* qkv, const float* ln, float eps, uint32_t T, uint32_t H, hipStream_t s);
#ifdef CS_DIAGNOSTICS  // libcsgpu_diag.so only (diagnostics.hip: cs_debug_gemm_time)
extern int g_gemm_wide_ablation;  // ablation instantiation of the wide kernel to launch, 0 = the product kernel
extern int g_gemm_wide_shape;      // block shape override (192 | 384), 0 = default
extern int g_gemm_wide_mfma;       // MFMA shape of the wide kernel's main loop (16 | 32), 0 = default
double gemm_wide_read_clock_ghz(double* main_cycles, double* epi_cycles);  // after an ablation-7 launch: median in-kernel clock
#else
constexpr int g_gemm_wide_ablation = 0, g_gemm_wide_shape = 0, g_gemm_wide_mfma = 0;
#endif
int32_t sh_weights_fit_wide(const _Float16* d_wsplit, uint64_t n_f16, uint32_t* d_scratch_flag, bool* ok, hipStream_t s);
int32_t launch_synth_params(float* d_out, const cs_bert_config& cfg, uint64_t seed, hipStream_t s);

}  // namespace cs
