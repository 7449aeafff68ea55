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
        config_from_hf({**hf, "model_type": "nomic_bert"})
