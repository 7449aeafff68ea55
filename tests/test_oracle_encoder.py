"""Pins the encoder oracle (oracle/bert_oracle.c) to the committed golden vectors produced
by HF transformers BertModel in float64 (tests/golden/make_encoder_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from codesearch_amd.bert_params import (POOL_CLS, POOL_MEAN, BertConfig, from_state_dict, param_count,
                                        synth_params, synth_token_batch, tensor_table, to_state_dict)

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_golden.npz"))


def case_cfg(name):
    m = GOLD[name + "/meta"]
    cfg = BertConfig(vocab_size=int(m[0]), hidden=int(m[1]), layers=int(m[2]), heads=int(m[3]),
                     intermediate=int(m[4]), max_position=int(m[5]))
    return cfg, int(m[6]), int(m[7]), int(m[8]), int(m[9]), bool(m[10])


def test_param_layout_and_generator_identity(oracle):
    cfg = BertConfig(vocab_size=512, layers=2)
    assert oracle.bert_param_count(cfg) == param_count(cfg)
    a, b = oracle.bert_synth_params(cfg, 7), synth_params(cfg, 7)
    assert np.array_equal(a, b)
    sd = to_state_dict(cfg, a)
    assert list(sd) == [n for n, _, _ in tensor_table(cfg)]
    assert np.array_equal(from_state_dict(cfg, sd), a)
    g = sd["encoder.layer.1.output.LayerNorm.weight"]
    assert abs(float(g.mean()) - 1.0) < 0.02 and float(g.std()) > 0.03
    assert param_count(BertConfig.bge_small()) == 33_212_160  # BGE-small w/o pooler


@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"] if str(n).startswith("tiny")])
def test_oracle_matches_hf_tiny(oracle, name):
    cfg, wseed, iseed, B, L, ragged = case_cfg(name)
    params = synth_params(cfg, wseed)
    ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
    for pooling, key in ((POOL_CLS, "cls"), (POOL_MEAN, "mean")):
        cfg.pooling = pooling
        r = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True, want_layers=True)
        np.testing.assert_allclose(r["pooled"], GOLD[f"{name}/{key}"], atol=2e-6)
        np.testing.assert_allclose(np.linalg.norm(r["pooled"], axis=1), 1.0, atol=1e-6)
    valid = mask.astype(bool)
    absmean = np.array([np.abs(h[valid]).mean() for h in r["layers"]])
    np.testing.assert_allclose(absmean, GOLD[name + "/layer_absmean"], rtol=1e-5)
    probe = np.array([[h[0, 0, 0], h[B - 1, 1, 7], h[0, mask[0].sum() - 1, 383]] for h in r["layers"]])
    np.testing.assert_allclose(probe, GOLD[name + "/layer_probe"], atol=2e-5)
    np.testing.assert_allclose(r["hidden"][0, 0], GOLD[name + "/last_row0"], atol=2e-5)


def test_oracle_matches_hf_full_bge_small_shape(oracle):
    """12-layer BGE-small shape; tolerance 1e-5 leaves margin under the 1e-4 target."""
    for name in ("full_dense", "full_ragged"):
        cfg, wseed, iseed, B, L, ragged = case_cfg(name)
        params = oracle.bert_synth_params(cfg, wseed)
        ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
        for pooling, key in ((POOL_CLS, "cls"), (POOL_MEAN, "mean")):
            cfg.pooling = pooling
            r = oracle.bert_forward(cfg, params, ids, mask, want_hidden=(key == "cls"))
            np.testing.assert_allclose(r["pooled"], GOLD[f"{name}/{key}"], atol=1e-5)
            if key == "cls":
                np.testing.assert_allclose(r["hidden"][0, 0], GOLD[name + "/last_row0"], atol=1e-4)
        # the synthetic model must still tell sequences apart, or the test is blind
        cls = GOLD[name + "/cls"]
        off_diag = (cls @ cls.T)[~np.eye(len(cls), dtype=bool)]
        assert off_diag.max() < 0.999, off_diag.max()


def test_padding_does_not_leak(oracle):
    """Batch-longest padding must not change a sequence's embedding (mask = -inf on keys)."""
    cfg = BertConfig(vocab_size=512, layers=2)
    params = synth_params(cfg, 5)
    ids, mask = synth_token_batch(cfg, 9, 3, 24, True)
    full = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
    n1 = int(mask[1].sum())
    alone = oracle.bert_forward(cfg, params, ids[1:2, :n1], mask[1:2, :n1])["pooled"]
    np.testing.assert_allclose(full[1], alone[0], atol=1e-6)


def test_checkpoint_loader_roundtrip(tmp_path):
    """Real-weight path: a safetensors BertModel checkpoint (with `bert.` prefix and the extra
    tensors HF exports) loads into the same flat block."""
    import json

    from safetensors.numpy import save_file

    from codesearch_amd.bert_params import config_from_hf, load_checkpoint

    hf = {"model_type": "bert", "vocab_size": 300, "hidden_size": 384, "num_hidden_layers": 2,
          "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 64,
          "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu"}
    cfg = config_from_hf(json.loads(json.dumps(hf)))
    flat = synth_params(cfg, 11)
    sd = {"bert." + k: np.ascontiguousarray(v) for k, v in to_state_dict(cfg, flat).items()}
    sd["bert.pooler.dense.weight"] = np.zeros((384, 384), np.float32)
    sd["bert.embeddings.position_ids"] = np.arange(64, dtype=np.int64)[None]
    path = str(tmp_path / "model.safetensors")
    save_file(sd, path)
    assert np.array_equal(load_checkpoint(path, cfg), flat)
    with pytest.raises(ValueError):
        config_from_hf({**hf, "model_type": "modernbert"})
    with pytest.raises(ValueError):  # NomicBert has its own keys (tests/test_oracle_nomic.py)
        config_from_hf({**hf, "model_type": "nomic_bert"})


def test_c_abi_checkpoint_loaders(tmp_path, gpu_lib):
    """cs_bert_config_from_dir / cs_bert_params_from_safetensors (csrc/checkpoint.cpp, host-only): a HF
    snapshot directory -> the same config and flat block as the Python loader, for F32, F16 and BF16
    files, with and without the `bert.` prefix; error texts for the usual failures."""
    import ctypes as C
    import json

    import torch
    from safetensors.numpy import save_file
    from safetensors.torch import save_file as save_torch

    from codesearch_amd import _lib
    from codesearch_amd.bert_params import config_from_hf

    hf = {"model_type": "bert", "vocab_size": 300, "hidden_size": 384, "num_hidden_layers": 2,
          "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 64,
          "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu",
          "architectures": ["BertModel"], "id2label": {"0": "LABEL_0"}, "torch_dtype": "float32",
          "note": "escapes \\ \" \u00e9 and nested {\"a\": [1, 2.5e3, true, null]}"}
    cfg = config_from_hf(hf)
    flat = synth_params(cfg, 11)
    sd = to_state_dict(cfg, flat)

    def load(d):
        c = _lib.BertConfig()
        _lib.check(gpu_lib.cs_bert_config_from_dir(str(d).encode(), POOL_MEAN, C.byref(c)))
        n = int(gpu_lib.cs_bert_param_count(C.byref(c)))
        out = np.empty(n, np.float32)
        _lib.check(gpu_lib.cs_bert_params_from_safetensors(str(d / "model.safetensors").encode(), C.byref(c),
                                                          out.ctypes.data_as(_lib.f32p), n))
        return c, out

    d32 = tmp_path / "f32"
    d32.mkdir()
    (d32 / "config.json").write_text(json.dumps(hf, indent=2))
    extra = {"bert." + k: np.ascontiguousarray(v) for k, v in sd.items()}
    extra["bert.pooler.dense.weight"] = np.zeros((384, 384), np.float32)
    extra["bert.embeddings.position_ids"] = np.arange(64, dtype=np.int64)[None]
    save_file(extra, str(d32 / "model.safetensors"), metadata={"format": "pt"})
    c, got = load(d32)
    assert (c.vocab_size, c.hidden, c.layers, c.heads, c.intermediate, c.max_position, c.type_vocab_size,
            c.pooling) == (300, 384, 2, 12, 1536, 64, 2, POOL_MEAN)
    assert abs(c.layer_norm_eps - 1e-12) < 1e-18 and np.array_equal(got, flat)

    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        d = tmp_path / name
        d.mkdir()
        (d / "config.json").write_text(json.dumps(hf))
        save_torch({k: torch.from_numpy(np.ascontiguousarray(v)).to(dt) for k, v in sd.items()},
                   str(d / "model.safetensors"))
        _, got = load(d)
        exp = torch.from_numpy(flat).to(dt).to(torch.float32).numpy()
        assert np.array_equal(got, exp), name

    def expect(fn, code, text):
        try:
            fn()
        except _lib.CsError as e:
            assert e.code == code and text in str(e), str(e)
        else:
            raise AssertionError("expected " + text)

    bad = tmp_path / "bad"
    bad.mkdir()
    expect(lambda: load(bad), _lib.CS_ERR_BAD_ARG, "cannot open")
    (bad / "config.json").write_text(json.dumps({**hf, "model_type": "deberta-v2"}))
    expect(lambda: load(bad), _lib.CS_ERR_UNSUPPORTED, "is not a BERT encoder")
    (bad / "config.json").write_text(json.dumps({**hf, "num_hidden_layers": 3}))
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(bad / "model.safetensors"))
    expect(lambda: load(bad), _lib.CS_ERR_BAD_ARG, "encoder.layer.2.attention.self.query.weight is missing")
    (bad / "config.json").write_text(json.dumps({**hf, "intermediate_size": 1024}))
    expect(lambda: load(bad), _lib.CS_ERR_DIM_MISMATCH, "has shape [1536, 384], config.json implies [1024, 384]")
    # pooling -1: the sentence-transformers pooling module of the snapshot decides
    def auto_pooling(d):
        c = _lib.BertConfig()
        _lib.check(gpu_lib.cs_bert_config_from_dir(str(d).encode(), -1, C.byref(c)))
        return c.pooling

    assert auto_pooling(d32) == POOL_CLS
    (d32 / "1_Pooling").mkdir()
    (d32 / "1_Pooling" / "config.json").write_text(json.dumps({"word_embedding_dimension": 384, "pooling_mode_cls_token": False,
                                                              "pooling_mode_mean_tokens": True}))
    assert auto_pooling(d32) == POOL_MEAN
    (d32 / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_cls_token": True, "pooling_mode_mean_tokens": False}))
    assert auto_pooling(d32) == POOL_CLS
    (bad / "model.safetensors").write_bytes(b"\x10\x00\x00\x00\x00\x00\x00\x00not json at all!")
    (bad / "config.json").write_text(json.dumps(hf))
    expect(lambda: load(bad), _lib.CS_ERR_BAD_ARG, "malformed header")


# ---- dynamically quantised Linears (the registry's *Q models; oracle: linear_q8 / cs_oracle_bert_forward_q8) ----------

def _onnx_dynamic_quantize(x):
    """ONNX DynamicQuantizeLinear (opset 11) in float32, restated independently of the oracle."""
    x = np.asarray(x, np.float32)
    lo, hi = np.minimum(np.float32(0), x.min()), np.maximum(np.float32(0), x.max())
    scale = np.float32(1) if hi == lo else np.float32((hi - lo) / np.float32(255))
    zp = np.float32(np.rint(np.clip(np.float32(0) - np.float32(lo / scale), 0, 255)))
    return np.clip(np.rint((x / scale).astype(np.float32)) + zp, 0, 255).astype(np.int64), scale, int(zp)


@pytest.mark.parametrize("per_channel,unsigned", [(False, True), (True, False), (True, True), (False, False)])
def test_quantised_linear_is_the_onnx_operator_chain(oracle, per_channel, unsigned):
    """DynamicQuantizeLinear -> MatMulInteger -> Cast -> Mul(x_scale * W_scale) -> Add(bias), bit for bit: the integer
    product is exact and the float stage is three roundings in a fixed order, so a numpy statement of the published operator
    definitions must reproduce the oracle's output exactly (onnxruntime itself is not installed here: parity unpinned)."""
    from codesearch_amd.bert_params import quantize_linear_weights

    rng = np.random.default_rng(11 + per_channel + 2 * unsigned)
    T, K, N = 37, 96, 40
    cfg = BertConfig(vocab_size=8, hidden=K, layers=1, heads=2, intermediate=N, max_position=4)
    # quantise a random [N, K] matrix the way the library's helper does it for every Linear (via a one-layer block)
    flat = synth_params(cfg, 3)
    sd = to_state_dict(cfg, flat)
    sd["encoder.layer.0.intermediate.dense.weight"][:] = (rng.standard_normal((N, K)) * 0.07).astype(np.float32)
    qflat, wscale = quantize_linear_weights(cfg, from_state_dict(cfg, sd), per_channel=per_channel, unsigned=unsigned)
    W = to_state_dict(cfg, qflat)["encoder.layer.0.intermediate.dense.weight"]
    sc = wscale[0, 4 * K:4 * K + N]
    d = np.rint(W / sc[:, None]).astype(np.int64)                      # W_q - W_zp
    assert np.abs(W / sc[:, None] - d).max() < 1e-3 and (d.max(axis=1) - d.min(axis=1)).max() <= 255
    x = (rng.standard_normal((T, K)) * rng.choice([0.2, 1.0, 3.0], size=(T, 1))).astype(np.float32)
    x[3, 5] = -7.25
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    got = oracle.linear_q8(x, W, sc, b)
    q, xs, xz = _onnx_dynamic_quantize(x)
    acc = (q - xz) @ d.T
    want = ((acc.astype(np.float32) * (xs * sc)[None, :].astype(np.float32)).astype(np.float32) + b[None, :]).astype(np.float32)
    assert np.array_equal(got, want)
    # degenerate tensor: all zeros -> scale 1, zero point 0, output = bias
    assert np.array_equal(oracle.linear_q8(np.zeros((2, K), np.float32), W, sc, b), np.tile(b, (2, 1)))


def test_quantised_forward_differs_from_the_f32_graph_and_is_call_dependent(oracle):
    """The quantised forward is a different function from the f32 graph of the same weights, and — DynamicQuantizeLinear
    ranging over the whole call — a sequence's embedding depends on what it was batched with."""
    from codesearch_amd.bert_params import quant_columns, quantize_linear_weights

    cfg = BertConfig(vocab_size=300, hidden=64, layers=2, heads=2, intermediate=128, max_position=32, pooling=POOL_MEAN)
    params, wscale = quantize_linear_weights(cfg, synth_params(cfg, 5), per_channel=False, unsigned=True)
    assert wscale.shape == (2, quant_columns(cfg)) and (wscale > 0).all()
    ids, mask = synth_token_batch(cfg, 9, 6, 24, True)
    q = oracle.bert_forward(cfg, params, ids, mask, wscale=wscale)["pooled"]
    f = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
    assert 1e-5 < np.abs(q - f).max() < 5e-2
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-5)
    alone = oracle.bert_forward(cfg, params, ids[:3], mask[:3], wscale=wscale)["pooled"]
    assert np.abs(alone - q[:3]).max() > 1e-6
