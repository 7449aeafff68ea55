/*
 * codesearch_gpu_diag.h — operator-level diagnostics of libcsgpu, exported ONLY by codesearch_amd/libcsgpu_diag.so (the
 * product library built once more with -DCS_DIAGNOSTICS: it additionally holds csrc/diagnostics.hip, the ablation
 * instantiations of the wide GEMM kernel and its in-kernel clock stamps).  Used by tests/test_gpu_gemm_split.py,
 * tests/test_gpu_quantized.py (operator parity) and benchmarks/ (A/B timing); nothing in INTEGRATION.md binds it and a
 * deployment does not ship it.  The diagnostic library exports every symbol of include/codesearch_gpu.h as well, so a
 * process may use it in place of libcsgpu.so.
 */
#ifndef CODESEARCH_GPU_DIAG_H
#define CODESEARCH_GPU_DIAG_H

#include "codesearch_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics: one dense layer of the encoder on host buffers, for unit parity tests of the
 * GEMM kernels (E2/E4/E5/E6).  C[M,N] = A[M,K] W[N,K]^T + bias, epilogue 0 = none,
 * 1 = erf-GELU, 2 = + resid[M,N]; mode = cs_gemm_mode (| 2: the persistent wide kernel).  N % 128 == 0, K % 32 == 0.
 * Wide kernel, N = 384 only: 3 = + resid, then LayerNorm (gamma = bias + 1, beta = -bias) in the same kernel, f32 and
 * split outputs; 4 = the same with the residual handed over in split form in the output buffer and no f32 output (how
 * the encoder runs it).  *range_flag (optional) is set when a split-f16 operand left the f16 range. */
CS_API int32_t cs_debug_gemm(int32_t device, int32_t mode, int32_t epilogue, const float* A,
                      const float* W, const float* bias, const float* resid, float* C,
                      uint32_t M, uint32_t N, uint32_t K, uint32_t* range_flag);

/* Diagnostics: one dynamically quantised dense layer on host buffers (unit parity of csrc/gemm_q8.hip against the ONNX
 * definitions of DynamicQuantizeLinear / MatMulInteger).  A [M,K] f32 activations (a_split & 1: staged through the
 * split-f16 form first, as attention and GELU hand them over; a_split & 4: the few-rows kernel (one launch: range from
 * pairs, quantisation and product; epilogues 0 / 1 / 2 / 4 as built for the encoder's chain, acc not reported); a_split & 8: the row-block products that quantise their own
 * rows on the way in — K = 384, M >= 4,096, epilogues 4 (f32 source), 2 (split source) and 5; acc then not reported; a_split & 16:
 * the slab kernel (csrc/gemm_q8_slab.hip: f32 rows, K = 384, epilogues 4 and 5, any M; | 32: its FFN-up store pass takes the s8
 * rows its range pass left instead of quantising again)); W [N,K] f32 = integer multiples of wscale[n]; epilogue as
 * cs_debug_gemm 0 / 1 / 2, 4 = bias -> split store.  Optional outputs: xq [M,K] the uint8 activations, xparams[2] =
 * (x_scale, x_zero_point), acc [M,N] the int32 MatMulInteger result.  N % 128 == 0, K % 128 == 0.
 * epilogue 5 = the FFN-up form (GELU, then quantised again for the next Linear, two passes over the product): C receives
 * the uint8 output as floats, xparams (then FOUR floats) also its (scale, zero_point), the first M entries of acc each
 * output row's sum of uint8 values. */
CS_API int32_t cs_debug_gemm_q8(int32_t device, int32_t epilogue, int32_t a_split, const float* A, const float* W,
                         const float* wscale, const float* bias, const float* resid, float* C, uint32_t M, uint32_t N,
                         uint32_t K, uint8_t* xq, float* xparams, int32_t* acc);

/* Diagnostics: the same products over a tensor that holds SEVERAL quantisation units (queued calls sharing a device
 * batch, cs_embedder_submit_*): row_slot [M] = each row's unit (consecutive runs of rows, in order), bit 31 set where the
 * row lies beyond its call's own padded length (quantised with the unit's parameters, never part of a range).  Row-block
 * products only (K = 384, M >= 4,096); epilogue 4 (f32 source -> split store), 2 (split source, + residual), 5 (FFN-up:
 * GELU, quantised again per unit; C = the uint8 output as floats).  row_params [M][4] = per row (x_scale, x_zero_point,
 * out_scale, out_zero_point) (the last two: epilogue 5), rowsums [M] (epilogue 5) each output row's sum of uint8 values. */
CS_API int32_t cs_debug_gemm_q8_units(int32_t device, int32_t epilogue, const float* A, const float* W, const float* wscale,
                               const float* bias, const float* resid, float* C, uint32_t M, uint32_t N, uint32_t K,
                               const uint32_t* row_slot, uint32_t units, float* row_params, int32_t* rowsums);

/* Diagnostics: the one-launch forward of short queries (csrc/small_forward.hip; embed_one / embed_queries_batch,
 * src/embed/mod.rs:164-226) — mini-batches of a 384-d BERT model with at most 192 token rows in split-f16 mode as ONE kernel,
 * bit-identical to the kernel-by-kernel path and measured slower than it (profiles/r05_small_forward_ab.log), so it is built
 * into this library only and taken with CS_SMALL_FORWARD=1: how many mini-batches took it, and how many of those gave up at
 * a grid barrier and were re-run kernel by kernel. */
CS_API int32_t cs_debug_small_forward_counters(cs_embedder* h, uint64_t* forwards, uint64_t* fallbacks);

/* Diagnostics: device milliseconds per launch of one dense layer on synthetic operands resident in HBM.
 * mode 0 exact-f32 MFMA, 1 split-f16 (128 x 128 / skinny kernels), 2 split-f16 wide kernel (N % 384 == 0);
 * epilogue 0 f32, 1 GELU -> split, 2 + residual, 3 LayerNorm-fused (mode 2, N = 384), 4 bias -> split (QKV);
 * ablation (mode 2): 0 none; epilogue 4 only: 1 no LDS-DMA, 2 no MFMA, 3 DMAs issued at the start of a k-step, 4-10 see
 * gemm_wide.hip; any epilogue: 192 / 384 = that block shape of the product kernel.  CS_DEBUG_GEMM_ZERO=1: all-zero operands
 * (the same instruction stream at the clock the chip holds on trivial data). */
CS_API int32_t cs_debug_gemm_time(int32_t device, int32_t mode, int32_t epilogue, uint32_t M, uint32_t N, uint32_t K,
                           uint32_t iters, int32_t ablation, double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* CODESEARCH_GPU_DIAG_H */
