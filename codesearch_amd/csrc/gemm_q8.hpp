// gemm_q8.hpp — launch interface of the dynamic-quantised dense layers (gemm_q8.hip).
#pragma once
#include "common.hpp"

namespace cs {

// What the GEMM needs to know about a row of activations / a column of weights (16 B each):
//   activation row m:  x[m][k] = (a[m][k] - za) * xs      a = stored s8 (the graph's uint8 minus 128), za = x_zero_point - 128
//   weight column n:   w[n][k] = (b[n][k] - zw) * ws      b = stored s8 (W_q - W_zp re-centred into [-128, 127])
// rowsum / colsum are the plain sums of the stored s8 values over k.
struct Q8RowMeta { float xs; int32_t za; int32_t rowsum; uint32_t pad; };
struct Q8ColMeta { float ws; int32_t zw; int32_t colsum; float bias; };  // bias: the layer's bias for this column (may be 0)

enum { Q8_SRC_F32 = 0, Q8_SRC_SPLIT = 1, Q8_SRC_LN = 2 };  // (Q8_SRC_LN: launch_gemm_q8_skinny_ln only)

// A quantisation unit's range slot: words 0, 1 = the bits of (lo, hi); 2..4 = scratch of the FFN-up range pass (gemm_q8.hip);
// all zero before the unit's first kernel of a forward.
constexpr uint32_t Q8_RANGE_WORDS = 8;

// One layer's quantised weights, bytes from the start of the layer's block: wqkv [3H][H] | ao [H][H] | up [I][H] | down [H][I]
struct Q8Layer { size_t qkv, ao, up, down, total; };
inline Q8Layer q8_layer(uint32_t H, uint32_t I) {
    Q8Layer o;
    o.qkv = 0;
    o.ao = o.qkv + (size_t)3 * H * H;
    o.up = o.ao + (size_t)H * H;
    o.down = o.up + (size_t)I * H;
    o.total = o.down + (size_t)H * I;
    return o;
}

// W [N][K] f32 = (integer) * scale[n]  ->  wq [N][K] s8 + cmeta [N].  *d_bad (device u32, zeroed by the caller) is OR-ed
// with 1 when a row's integers span more than 8 bits, with 2 when a weight is not a multiple of its scale.
int32_t launch_q8_pack_weight(const float* d_W, const float* d_scale, const float* d_bias, uint32_t N, uint32_t K, int8_t* d_wq,
                              Q8ColMeta* d_cmeta, uint32_t* d_bad, hipStream_t s);

// d_row_slot [T] for a device batch of sequences of L positions each that holds several quantisation units (calls):
// d_seq_unit [T / L] = the unit of each sequence, d_unit_len [units] = each unit's own padded length.
int32_t launch_q8_row_slots(const uint32_t* d_seq_unit, const uint32_t* d_unit_len, uint32_t T, uint32_t L, uint32_t* d_row_slot,
                            hipStream_t s);

// DynamicQuantizeLinear of [T][K] activations (f32 rows, or split-f16 lines [T][K/32][64]):
//   d_range [slots][Q8_RANGE_WORDS] u32: running (lo, hi) of each quantisation unit, all zero before the first call of a forward slot
//   d_row_slot (may be null: one unit, slot 0): per row, the unit it belongs to; bit 31 set = the row lies outside its unit's
//              own padded length (not part of the tensor the reference quantises): quantised, but kept out of the range
// -> d_xq [T][K] s8, d_rmeta [T].
// d_range_pairs / n_pairs (optional): (lo, hi) float pairs the tensor's producer left per block or wave (LayerNorm,
// attention): reduced instead of reading the tensor for its range.
int32_t launch_q8_quantize(int src_kind, const void* d_src, uint32_t T, uint32_t K, uint32_t* d_range, const uint32_t* d_row_slot,
                           int8_t* d_xq, Q8RowMeta* d_rmeta, hipStream_t s, const float* d_range_pairs = nullptr,
                           uint32_t n_pairs = 0);

// C = MatMulInteger(xq, wq) * (xs * ws) + bias, then the epilogue `epi` (SH_OUT_*, encoder.hpp).  N % 128 == 0, K % 128 == 0.
int32_t launch_gemm_q8(int epi, const int8_t* d_xq, const Q8RowMeta* d_rmeta, const int8_t* d_wq, const Q8ColMeta* d_cmeta,
                       const float* bias, const float* resid, float* C, _Float16* Cs, uint32_t M, uint32_t N, uint32_t K,
                       uint32_t* d_flag, hipStream_t s, int32_t* d_acc_dbg = nullptr);

// From 4,096 rows a K = 384 layer can take the f32-class tensor itself (q8_rows_from_source): the product kernel's
// blocks quantise their own rows on the way in, only the tensor's range (launch_q8_range: a reduction of the producer's
// pairs, or a pass over the tensor) is needed first.  One quantisation unit only.  CS_Q8_ROWS_SRC=0: never.
bool q8_rows_from_source(uint32_t M, uint32_t K);
int32_t launch_q8_range(int src_kind, const void* d_src, uint32_t T, uint32_t K, uint32_t* d_range, hipStream_t s,
                        const float* d_range_pairs = nullptr, uint32_t n_pairs = 0);
// d_row_slot (optional): SEVERAL units in the tensor — d_in_range / d_range_out are then the units' slot arrays and every
// row is quantised with its own unit's parameters (row_slot as for launch_q8_quantize).
int32_t launch_gemm_q8_from_source(int epi, int src_kind, const void* d_src, const uint32_t* d_in_range, const int8_t* d_wq,
                                   const Q8ColMeta* d_cmeta, const float* bias, const float* resid, float* C, _Float16* Cs,
                                   uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s,
                                   const uint32_t* d_row_slot = nullptr, const uint32_t* d_cmeta_tiles = nullptr, int8_t* d_xq_scratch = nullptr);
int32_t launch_gemm_q8_gelu_requant_from_source(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const Q8ColMeta* d_cmeta,
                                                const float* bias, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_range_out,
                                                int8_t* d_out, Q8RowMeta* d_rmeta_out, hipStream_t s,
                                                const uint32_t* d_row_slot = nullptr, const uint32_t* d_cmeta_tiles = nullptr,
                                                int8_t* d_xq_scratch = nullptr);
// The same two products at indexing batch sizes (gemm_q8_slab.hip: a block owns 256 rows, W the MFMA's first operand, the tile
// leaves from registers): taken by the two calls above when d_cmeta_tiles — the weight's column metadata as 2-KiB
// structure-of-arrays tiles (ws | -zw | colsum | bias per 128 columns; launch_q8_cmeta_tiles, N * 16 bytes) — is given, the
// tensor is one unit of f32 rows and q8_slab_takes(M, N, K).  Same bits.  CS_Q8_SLAB_MIN_M (default 8192; 0 = never).
bool q8_slab_takes(uint32_t M, uint32_t N, uint32_t K);
uint32_t q8_gelu_table_on();  // CS_Q8_GELU_TABLE=0: the store pass computes the re-quantised GELU byte directly (read per launch)
int32_t launch_q8_cmeta_tiles(const Q8ColMeta* d_cmeta, uint32_t N, uint32_t* d_tiles, hipStream_t s);
int32_t launch_gemm_q8_slab_split(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const uint32_t* d_cmeta_tiles,
                                  _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s, int8_t* d_xq_scratch = nullptr);
int32_t launch_gemm_q8_slab_gelu_requant(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const uint32_t* d_cmeta_tiles,
                                         uint32_t M, uint32_t N, uint32_t K, uint32_t* d_range_out, int8_t* d_out, Q8RowMeta* d_rmeta_out,
                                         uint32_t use_table, hipStream_t s, int8_t* d_xq_scratch = nullptr);
// N = 384 layers of a one-unit batch from 4,096 rows: product + bias + residual + LayerNorm in ONE kernel (gemm_q8_ln_kernel; a
// wave owns 16 whole rows).  src_kind: Q8_SRC_SPLIT (d_src split-f16 [M][K/32][64], d_in_range its range slot; K = 384) or
// Q8_SRC_PREQUANT (d_src s8 [M][K], d_rmeta its rows; K = 384 | 1536).  X [M][384]: the residual on entry, the normalised rows on
// return; d_range_pairs receives *out_pairs (lo, hi) pairs (one per 16 rows) for the quantisation that follows.
constexpr int Q8_SRC_PREQUANT = -1;
bool q8_ln_fused_takes(uint32_t M, uint32_t N, uint32_t K);
int32_t launch_gemm_q8_ln(int src_kind, const void* d_src, const Q8RowMeta* d_rmeta, const uint32_t* d_in_range, const int8_t* d_wq,
                          const Q8ColMeta* d_cmeta, float* X, const float* ln_g, const float* ln_b, float eps, uint32_t M, uint32_t K,
                          float* d_range_pairs, uint32_t* out_pairs, hipStream_t s, uint32_t* d_out_slot = nullptr, bool w_stage_major = false);
// d_out [K / 64][N][64] from d_wq [N][K]: the order launch_gemm_q8_ln streams a weight in (w_stage_major: every request whole
// 128-byte lines; row-major it takes 64 B out of each line it touches)
int32_t launch_q8_stage_major(const int8_t* d_wq, uint32_t N, uint32_t K, int8_t* d_out, hipStream_t s);
// The units' ranges from what the tensor's producer left: pairs_per_seq (lo, hi) pairs per sequence, sequence by sequence
// (attention: its waves' pairs; LayerNorm with EncoderLaunch::range_rows: one pair per token row — pairs_are_rows, and
// only positions below the unit's own padded length count).  Writes words 0, 1 of every unit's slot.
int32_t launch_q8_range_units(const float* d_range_pairs, uint32_t pairs_per_seq, bool pairs_are_rows, const uint32_t* d_seq_unit,
                              const uint32_t* d_unit_len, uint32_t B, uint32_t units, uint32_t* d_range, hipStream_t s);

// A few token rows (up to q8_skinny_max_m; CS_Q8_SKINNY_MAX_M, 0 = never): one launch per Linear — each block reduces
// the (lo, hi) pairs its input's producer left, quantises its 16 rows into LDS and multiplies one 16 x 16 tile with K
// split over its waves.  With d_range_out every block leaves the (lo, hi) of what it stored (*out_pairs of them).
uint32_t q8_skinny_max_m();
int32_t launch_gemm_q8_skinny(int epi, int src_kind, const void* d_src, const float* d_range_pairs, uint32_t n_pairs,
                              const int8_t* d_wq, const Q8ColMeta* d_cmeta, const float* resid, float* C, _Float16* Cs, uint32_t M,
                              uint32_t N, uint32_t K, uint32_t* d_flag, float* d_range_out, uint32_t* out_pairs, hipStream_t s);

// The same for up to 16 token rows of a 384-wide tensor that still wants its LayerNorm: d_y [M][384] f32 is normalised by every
// block itself (the block's 16 rows ARE the tensor: its range is theirs), quantised and multiplied; the blocks of column tile 0
// write the normalised rows to d_xout (the residual the next product adds; never d_y).  epi: SH_OUT_SPLIT | SH_OUT_SPLIT_GELU.
int32_t launch_gemm_q8_skinny_ln(int epi, const float* d_y, const float* ln_g, const float* ln_b, float eps, float* d_xout,
                                 const int8_t* d_wq, const Q8ColMeta* d_cmeta, _Float16* Cs, uint32_t M, uint32_t N, uint32_t* d_flag,
                                 float* d_range_out, uint32_t* out_pairs, hipStream_t s);

// FFN-up of a quantised model in two passes over the same product: GELU(x W^T + b) re-quantised for FFN-down without the
// f32-class tensor ever reaching HBM.  d_range_out (one slot, zero before the call) collects the output tensor's range;
// d_out [M][N] s8 and d_rmeta_out [M] are the next launch_gemm_q8's operands.
int32_t launch_gemm_q8_gelu_requant(const int8_t* d_xq, const Q8RowMeta* d_rmeta, const int8_t* d_wq, const Q8ColMeta* d_cmeta,
                                    const float* bias, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_range_out, int8_t* d_out,
                                    Q8RowMeta* d_rmeta_out, hipStream_t s);

}  // namespace cs
