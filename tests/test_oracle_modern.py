"""Pins the CS_ARCH_MODERN branch of the encoder oracle (oracle/bert_oracle.c: pre-norm layers, rotary positions with a global
and a local base, sliding-window local attention, GELU-gated feed-forward, final LayerNorm) to the committed golden vectors of
tests/golden/make_modern_golden.py — HF transformers' own ModernBertModel in float64 — and the layout / generator
identities.  CPU only."""
import os

import numpy as np
import pytest

from codesearch_amd.bert_params import (ARCH_MODERN, POOL_MEAN, BertConfig, param_count, synth_params, tensor_table,
                                        token_batch_with_lens)

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "modern_golden.npz"))


def case_cfg(name):
    m = GOLD[name + "/meta"]
    cfg = BertConfig(vocab_size=int(m[0]), hidden=int(m[1]), layers=int(m[2]), heads=int(m[3]), intermediate=int(m[4]),
                     max_position=int(m[5]), pooling=POOL_MEAN, arch=ARCH_MODERN, layer_norm_eps=1e-5, rotary_base=160000.0,
                     rotary_base_local=10000.0, global_every=3, local_window=int(m[11]))
    return cfg, int(m[6]), int(m[7]), [int(x) for x in GOLD[name + "/lens"]], int(m[9])


def test_modern_layout_and_generator_identity(oracle):
    cfg = BertConfig(vocab_size=512, hidden=128, layers=3, heads=2, intermediate=256, pooling=POOL_MEAN, arch=ARCH_MODERN,
                     layer_norm_eps=1e-5, rotary_base=160000.0, rotary_base_local=10000.0, global_every=3, local_window=64)
    bert = BertConfig(vocab_size=512, hidden=128, layers=3, heads=2, intermediate=256)
    H, I = cfg.hidden, cfg.intermediate
    # no position and no token-type table; one more [I, H] + [I] per layer than BERT; a final LayerNorm
    assert param_count(cfg) == param_count(bert) - bert.max_position * H - bert.type_vocab_size * H + cfg.layers * (I * H + I) + 2 * H
    assert oracle.bert_param_count(cfg) == param_count(cfg)
    assert np.array_equal(oracle.bert_synth_params(cfg, 9), synth_params(cfg, 9))
    names = [n for n, _, _ in tensor_table(cfg)]
    assert "embeddings.position_embeddings.weight" not in names and "embeddings.token_type_embeddings.weight" not in names
    assert names[-2:] == ["final_norm.weight", "final_norm.bias"]


@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"] if str(n) != "modern_large_shape"])
def test_oracle_matches_hf_modernbert(oracle, name):
    cfg, wseed, iseed, lens, L = case_cfg(name)
    params = synth_params(cfg, wseed)
    ids, mask = token_batch_with_lens(cfg, iseed, lens, L)
    r = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True)
    np.testing.assert_allclose(r["pooled"], GOLD[name + "/mean"], atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(r["pooled"], axis=1), 1.0, atol=1e-6)
    valid = mask.astype(bool)
    np.testing.assert_allclose(np.abs(r["hidden"][valid]).mean(), float(GOLD[name + "/last_absmean"]), rtol=1e-5)
    B, H = len(lens), cfg.hidden
    probe = np.array([r["hidden"][0, 0, 0], r["hidden"][B - 1, 1, 7], r["hidden"][0, mask[0].sum() - 1, H - 1]])
    np.testing.assert_allclose(probe, GOLD[name + "/last_probe"], atol=3e-5)
    np.testing.assert_allclose(r["hidden"][0, 0], GOLD[name + "/last_row0"], atol=3e-5)


def test_oracle_matches_at_the_published_shape(oracle):
    """28 x 1024, 16 heads of 64, intermediate 2624 padded to 2688, vocab 50368, window 64, every third layer global
    (lightonai/modernbert-embed-large's config.json), 2 x 200 tokens."""
    cfg, wseed, iseed, lens, L = case_cfg("modern_large_shape")
    params = oracle.bert_synth_params(cfg, wseed)
    ids, mask = token_batch_with_lens(cfg, iseed, lens, L)
    r = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True)
    # 28 pre-norm layers: the residual stream is never re-normalised, and fp32 rounding of the oracle itself adds up to 3.3e-5
    # on an embedding element against HF's float64 (the shorter cases above agree to 2e-6); the north-star tolerance is 1e-4
    np.testing.assert_allclose(r["pooled"], GOLD["modern_large_shape/mean"], atol=6e-5)
    # (a single token row of a 28-layer RANDOM-weight network amplifies rounding differences by ~1.4 per layer: 4e-3 here;
    # the mean over a row's tokens, which is what an embedding is, averages it down to the figure above)
    np.testing.assert_allclose(r["hidden"][0, 0], GOLD["modern_large_shape/last_row0"], atol=1e-2)


def test_local_layers_are_local_and_padding_does_not_leak(oracle):
    cfg, wseed, iseed, lens, L = case_cfg("dh32_L100")
    params = synth_params(cfg, wseed)
    ids, mask = token_batch_with_lens(cfg, iseed, lens, L)
    base = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
    ids2 = np.where(mask == 1, ids, 77).astype(np.int32)  # garbage in the padded positions changes nothing
    assert np.abs(oracle.bert_forward(cfg, params, ids2, mask)["pooled"] - base).max() < 1e-6
    # a model whose every layer is local (global_every larger than the layer count keeps only layer 0 global) differs from
    # one that is global throughout: the window is really applied
    all_global = BertConfig(**{**cfg.__dict__, "global_every": 1})
    assert np.abs(oracle.bert_forward(all_global, params, ids, mask)["pooled"] - base).max() > 1e-4


# ---- the ONNX export (what fastembed caches for the registry entry) -------------------------------------------------------

ONNX_CFG = BertConfig(vocab_size=48, hidden=64, layers=3, heads=2, intermediate=128, max_position=64, type_vocab_size=1,
                      pooling=POOL_MEAN, arch=ARCH_MODERN, layer_norm_eps=1e-5, rotary_base=160000.0, rotary_base_local=10000.0,
                      global_every=3, local_window=8)


def block_from_modernbert_state(cfg, sd):
    """The flat parameter block of `cfg` from a ModernBertModel state dict whose feed-forward is I_f <= cfg.intermediate wide:
    zero padding, zero biases, layer 0's (never read) attn_norm slot at one."""
    from codesearch_amd.bert_params import to_state_dict

    flat = np.zeros(param_count(cfg), np.float32)
    ours = to_state_dict(cfg, flat)  # views into flat
    H = cfg.hidden
    ours["embeddings.word_embeddings.weight"][:] = sd["embeddings.tok_embeddings.weight"]
    ours["embeddings.LayerNorm.weight"][:] = sd["embeddings.norm.weight"]
    ours["final_norm.weight"][:] = sd["final_norm.weight"]
    for l in range(cfg.layers):
        a, b = f"encoder.layer.{l}.", f"layers.{l}."
        ours[a + "attention.output.LayerNorm.weight"][:] = sd[b + "attn_norm.weight"] if l else 1.0
        for i, r in enumerate(("query", "key", "value")):
            ours[a + f"attention.self.{r}.weight"][:] = sd[b + "attn.Wqkv.weight"][i * H:(i + 1) * H]
        ours[a + "attention.output.dense.weight"][:] = sd[b + "attn.Wo.weight"]
        ours[a + "output.LayerNorm.weight"][:] = sd[b + "mlp_norm.weight"]
        If = sd[b + "mlp.Wi.weight"].shape[0] // 2
        ours[a + "intermediate.gate.weight"][:If] = sd[b + "mlp.Wi.weight"][:If]
        ours[a + "intermediate.dense.weight"][:If] = sd[b + "mlp.Wi.weight"][If:]
        ours[a + "output.dense.weight"][:, :If] = sd[b + "mlp.Wo.weight"]
    return flat


def load_onnx(gpu_lib, path, cfg):
    import ctypes as C

    from codesearch_amd import _lib

    c = cfg.to_c()
    out = np.full(param_count(cfg), np.nan, np.float32)
    rc = gpu_lib.cs_bert_params_from_onnx(str(path).encode(), C.byref(c), out.ctypes.data_as(_lib.f32p), out.size)
    return rc, out, gpu_lib.cs_last_error().decode()


def test_onnx_reader_reads_a_modernbert_export_written_by_torchs_own_exporter(gpu_lib, oracle):
    """tests/golden/modern_tiny_export.onnx (make_modern_onnx_fixture.py): transformers' own ModernBertModel through torch.onnx's
    TorchScript exporter — the four bias-free Linear weights of a layer as anonymous transposed initialisers in module order,
    LayerNorm weights under their state-dict names, the feed-forward 80 wide inside a block padded to 128.  The flat block must
    equal the one built from the state dict the file was exported from, bit for bit; and the CPU oracle run on that block
    must reproduce the exporting model's own output (stored with the state)."""
    from codesearch_amd import _lib

    gold = os.path.join(os.path.dirname(__file__), "golden")
    state = dict(np.load(os.path.join(gold, "modern_tiny_export_state.npz")))
    rc, got, err = load_onnx(gpu_lib, os.path.join(gold, "modern_tiny_export.onnx"), ONNX_CFG)
    assert rc == _lib.CS_OK, err
    assert np.array_equal(got, block_from_modernbert_state(ONNX_CFG, state))
    pooled = oracle.bert_forward(ONNX_CFG, got, state["query_ids"], state["query_mask"])["pooled"]
    np.testing.assert_allclose(pooled, state["query_pooled"], atol=2e-6)
    # a config with another layer count, or another width, does not match the file
    rc, _, err = load_onnx(gpu_lib, os.path.join(gold, "modern_tiny_export.onnx"), BertConfig(**{**ONNX_CFG.__dict__, "layers": 4}))
    assert rc == _lib.CS_ERR_BAD_ARG and "weight products" in err
    rc, _, err = load_onnx(gpu_lib, os.path.join(gold, "modern_tiny_export.onnx"), BertConfig(**{**ONNX_CFG.__dict__, "intermediate": 64}))
    assert rc == _lib.CS_ERR_DIM_MISMATCH and "mlp.Wi" in err
    rc, _, err = load_onnx(gpu_lib, os.path.join(gold, "bert_tiny_export.onnx"), ONNX_CFG)
    assert rc == _lib.CS_ERR_BAD_ARG and "tok_embeddings" in err


def test_onnx_reader_reads_an_opset_17_export(gpu_lib, tmp_path):
    """The same model exported at opset 17: LayerNorm arrives as ONE LayerNormalization node instead of the ReduceMean / Sub / Pow
    chain — the reader goes by initialiser names and weight products, not by how a norm is spelt."""
    import importlib.util

    from codesearch_amd import _lib

    spec = importlib.util.spec_from_file_location("make_modern_onnx_fixture", os.path.join(os.path.dirname(__file__), "golden", "make_modern_onnx_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = mod.write(str(tmp_path), "opset17", opset=17)
    from tests.onnx_dump import read

    assert any(op == "LayerNormalization" for op, _, _, _ in read(path)["nodes"])
    state = dict(np.load(os.path.join(str(tmp_path), "opset17_state.npz")))
    rc, got, err = load_onnx(gpu_lib, path, ONNX_CFG)
    assert rc == _lib.CS_OK, err
    assert np.array_equal(got, block_from_modernbert_state(ONNX_CFG, state))
