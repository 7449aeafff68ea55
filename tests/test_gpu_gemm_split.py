"""Unit parity of the encoder's dense-layer kernels through
cs_debug_gemm: the split-f16 MFMA GEMM and the exact-f32 MFMA GEMM against float64 numpy.

Bar: the split form must stay within a small multiple of f32 rounding (its per-product error
bound is ~3 * 2^-22, codesearch_amd/csrc/split_f16.hpp) — asserted as max |err| <= 6e-7 *
sum_k |a||w| per output, next to the f32 kernel's own error on the same data."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F32, SPLIT, WIDE = 0, 1, 2  # WIDE: split-f16 on the persistent 128 x 384 one-accumulator kernel (gemm_wide.hip)


def run_gemm(lib, mode, epi, A, W, bias, resid=None):
    from codesearch_amd import _lib

    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), np.float32)
    flag = C.c_uint32(0)
    f32p = _lib.f32p
    rp = resid.ctypes.data_as(f32p) if resid is not None else None
    _lib.check_diag(lib.cs_debug_gemm(0, mode, epi, A.ctypes.data_as(f32p), W.ctypes.data_as(f32p),
                                 bias.ctypes.data_as(f32p), rp, out.ctypes.data_as(f32p), M, N, K, C.byref(flag)))
    return out, int(flag.value)


def reference(epi, A, W, bias, resid):
    acc = A.astype(np.float64) @ W.astype(np.float64).T + bias.astype(np.float64)
    if epi == 1:
        erf = np.vectorize(math.erf)
        acc = 0.5 * acc * (1.0 + erf(acc / math.sqrt(2.0)))
    if epi == 2:
        acc = acc + resid.astype(np.float64)
    return acc


@pytest.mark.parametrize("M,N,K", [(1, 128, 32), (300, 128, 64), (128, 384, 384), (257, 1536, 384),
                                   (1000, 384, 1536), (4096 + 77, 1152, 384)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_split_gemm_matches_float64_like_f32_does(diag_lib, M, N, K, epi):
    rng = np.random.default_rng(M * 7 + N * 3 + K + epi)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32) if epi == 2 else None
    ref = reference(epi, A, W, bias, resid)
    scale = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1.0  # sum |a||w| (+|bias|,|resid| ~ 1)
    got_s, flag = run_gemm(diag_lib, SPLIT, epi, A, W, bias, resid)
    got_f, _ = run_gemm(diag_lib, F32, epi, A, W, bias, resid)
    assert flag == 0
    err_s = np.abs(got_s - ref) / scale
    err_f = np.abs(got_f - ref) / scale
    assert err_f.max() < 4e-7, err_f.max()
    assert err_s.max() < 6e-7, (err_s.max(), err_f.max())
    # and the two kernels agree with each other far inside the 1e-4 bar on the embeddings
    assert np.abs(got_s - got_f).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,N,K", [(1, 384, 32), (130, 384, 64), (128, 384, 384), (257, 1536, 384), (1000, 384, 1536),
                                   (4096 + 77, 1152, 384),
                                   (128 * 300 + 5, 384, 96),     # 301 tiles over <= 256 persistent blocks
                                   (128 * 70, 1536, 384)])       # 280 tiles, four n-tiles per m-tile
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_wide_gemm_matches_float64_and_the_two_accumulator_kernels(diag_lib, M, N, K, epi):
    """gemm_wide.hip: one accumulator (w_hi scaled by 2^11 in registers), 128 x 384 tiles, persistent blocks,
    next tile's first stage prefetched under the epilogue — same bar as the 128 x 128 split kernels."""
    rng = np.random.default_rng(M * 11 + N * 5 + K + epi)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32) if epi == 2 else None
    got_w, flag = run_gemm(diag_lib, WIDE, epi, A, W, bias, resid)
    got_s, _ = run_gemm(diag_lib, SPLIT, epi, A, W, bias, resid)
    assert flag == 0
    if M * N * K <= 4096 * 1536 * 384:  # float64 reference where numpy finishes in seconds
        ref = reference(epi, A, W, bias, resid)
        scale = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1.0
        err = np.abs(got_w - ref) / scale
        assert err.max() < 6e-7, err.max()
    assert np.abs(got_w - got_s).max() < 2e-5 * max(1.0, np.abs(got_s).max())
    rows = rng.integers(0, M, 64)  # sampled rows against float64 at every size
    ref_rows = reference(epi, A[rows], W, bias, None if resid is None else resid[rows])
    sc = np.abs(A[rows]).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1.0
    assert (np.abs(got_w[rows] - ref_rows) / sc).max() < 6e-7


@pytest.mark.parametrize("M,K", [(1, 32), (130, 384), (1000, 1536), (128 * 300 + 5, 96)])
def test_wide_gemm_with_fused_layernorm(diag_lib, M, K):
    """N = 384: dense layer + bias + residual + LayerNorm in one kernel (accumulators start at (bias + resid) * 2^11;
    row statistics across the block's four column waves).  cs_debug_gemm epilogue 3 runs it in place over the
    residual with gamma = bias + 1, beta = -bias, checks that its f32 and split outputs agree, and returns the split one."""
    N = 384
    rng = np.random.default_rng(M + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    got, flag = run_gemm(diag_lib, WIDE, 3, A, W, bias, resid)
    assert flag == 0
    # the encoder's form: the residual arrives in split form (hi + lo / 2048, 2^-22 relative) in the buffer the
    # result overwrites, and no f32 copy is written (cs_debug_gemm epilogue 4)
    got_split_resid, flag4 = run_gemm(diag_lib, WIDE, 4, A, W, bias, resid)
    assert flag4 == 0 and np.abs(got_split_resid - got).max() < 2e-6
    rows = np.arange(M) if M <= 1000 else rng.integers(0, M, 256)
    v = A[rows].astype(np.float64) @ W.astype(np.float64).T + bias.astype(np.float64) + resid[rows].astype(np.float64)
    mu = v.mean(axis=1, keepdims=True)
    var = ((v - mu) ** 2).mean(axis=1, keepdims=True)
    ref = (v - mu) / np.sqrt(var + 1e-12) * (bias.astype(np.float64) + 1.0) - bias.astype(np.float64)
    assert np.abs(got[rows] - ref).max() < 1e-5, np.abs(got[rows] - ref).max()  # K = 1536: 5.7e-6 (the unfused path: the same order)


def test_wide_gemm_exact_and_range(diag_lib):
    """Integer data exact through the scaled accumulator; |w| up to 31 still inside the f16 range after the
    2^11 scaling; the activation range flag still raised by the epilogue's split."""
    rng = np.random.default_rng(6)
    M, N, K = 200, 384, 96
    A = rng.integers(-4, 5, (M, K)).astype(np.float32)
    W = rng.integers(-3, 4, (N, K)).astype(np.float32)
    W[:, 0] += np.arange(N, dtype=np.float32) % 7
    W[5, 7] = 31.0
    bias = np.arange(N, dtype=np.float32)
    ref = (A.astype(np.int64) @ W.astype(np.int64).T + bias.astype(np.int64)).astype(np.float32)
    got, flag = run_gemm(diag_lib, WIDE, 0, A, W, bias)
    assert flag == 0 and np.array_equal(got, ref)
    bias[3] = 9.0e4  # GELU(x + 9e4) leaves the f16 range: the split-form epilogue must say so
    _, flag = run_gemm(diag_lib, WIDE, 1, A, W, bias)
    assert flag == 1


def test_split_gemm_exact_on_f16_representable_data(diag_lib):
    """Small-integer operands are exact in f16, every product and partial sum is exact in f32:
    both kernels must return the integer result bit for bit (catches any k/row/col mapping slip,
    with an asymmetric W so a transposed C write cannot pass)."""
    rng = np.random.default_rng(5)
    M, N, K = 200, 256, 96
    A = rng.integers(-4, 5, (M, K)).astype(np.float32)
    W = rng.integers(-3, 4, (N, K)).astype(np.float32)
    W[:, 0] += np.arange(N, dtype=np.float32) % 7  # asymmetric in n
    bias = np.arange(N, dtype=np.float32)
    ref = (A.astype(np.int64) @ W.astype(np.int64).T + bias.astype(np.int64)).astype(np.float32)
    for mode in (SPLIT, F32):
        got, flag = run_gemm(diag_lib, mode, 0, A, W, bias)
        assert flag == 0 and np.array_equal(got, ref)


def test_split_gemm_small_and_mixed_magnitudes(diag_lib):
    """Values below the f16 normal range (|x| < 2^-14) live entirely in the scaled low part."""
    rng = np.random.default_rng(11)
    M, N, K = 130, 128, 128
    A = rng.standard_normal((M, K)).astype(np.float32)
    A[:, ::3] *= 1e-6
    A[:, 1::5] *= 300.0
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    W[::2, ::4] *= 1e-4
    bias = np.zeros(N, np.float32)
    ref = reference(0, A, W, bias, None)
    a64, w64 = np.abs(A).astype(np.float64), np.abs(W).astype(np.float64)
    scale = a64 @ w64.T + 1e-30
    # operands below 2^-14 keep 11 bits (split_f16.hpp): their products may be off by 2^-11 |a||w|
    tiny = (a64 * (a64 < 2.0 ** -14)) @ w64.T + a64 @ (w64 * (w64 < 2.0 ** -14)).T
    got, flag = run_gemm(diag_lib, SPLIT, 0, A, W, bias)
    assert flag == 0
    assert np.all(np.abs(got - ref) <= 4e-7 * scale + 2.0 ** -11 * tiny)
    assert (2.0 ** -11 * tiny / scale).max() < 1e-5  # and that allowance is itself small here


def test_split_range_flag(diag_lib):
    A = np.ones((4, 32), np.float32)
    W = np.ones((128, 32), np.float32)
    bias = np.zeros(128, np.float32)
    _, flag = run_gemm(diag_lib, SPLIT, 0, A, W, bias)
    assert flag == 0
    A[2, 5] = 7.0e4  # > 65504
    _, flag = run_gemm(diag_lib, SPLIT, 0, A, W, bias)
    assert flag == 1
    A[2, 5] = np.nan
    _, flag = run_gemm(diag_lib, SPLIT, 0, A, W, bias)
    assert flag == 1


def test_encoder_split_vs_f32_mode_and_fallback(gpu_lib, oracle):
    """Both arithmetic modes of the encoder meet the oracle; a mini-batch whose activations leave
    the f16 range is recomputed on the exact-f32 kernels (counted)."""
    from codesearch_amd import FastEmbedder, ModelType
    from codesearch_amd.bert_params import BertConfig, synth_params, synth_token_batch

    cfg = BertConfig(vocab_size=512, layers=2)
    ids, mask = synth_token_batch(cfg, 3, 5, 40, True)
    params = synth_params(cfg, 31)
    ref = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
    a = FastEmbedder(ModelType.BGESmallENV15, config=cfg, params=params, gemm_mode="split")
    b = FastEmbedder(ModelType.BGESmallENV15, config=cfg, params=params, gemm_mode="f32")
    ea, eb = a.embed_ids(ids, mask), b.embed_ids(ids, mask)
    np.testing.assert_allclose(ea, ref, atol=2e-5)
    np.testing.assert_allclose(eb, ref, atol=2e-5)
    np.testing.assert_allclose(ea, eb, atol=5e-6)
    assert a.debug_counters() == (1, 0, 0) and b.debug_counters() == (0, 1, 0)
    # blow one embedding row up so the first LayerNorm input is fine but an FFN activation is huge
    from codesearch_amd.bert_params import to_state_dict

    big = params.copy()
    to_state_dict(cfg, big)["encoder.layer.0.intermediate.dense.bias"][:8] = 3.0e5  # gelu(x + 3e5) > 65504
    c = FastEmbedder(ModelType.BGESmallENV15, config=cfg, params=big)
    ec = c.embed_ids(ids, mask)
    refc = oracle.bert_forward(cfg, big, ids, mask)["pooled"]
    np.testing.assert_allclose(ec, refc, atol=2e-5)
    s, f, fb = c.debug_counters()
    assert fb == 1 and f == 1 and s == 1
