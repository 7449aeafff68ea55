"""Pins the CS_ARCH_JINA / CS_ARCH_JINA_QKNORM branch of the encoder oracle (oracle/bert_oracle.c: ALiBi on the scores, a
GELU-gated feed-forward, optional LayerNorm on the query / key rows, no position table) to the committed golden vectors of
tests/golden/make_jina_golden.py (a float64 torch statement whose head slopes come from transformers' MPT ALiBi builder),
and the JinaBert checkpoint-name / config.json mapping.  CPU only."""
import os

import numpy as np
import pytest

from codesearch_amd.bert_params import (ARCH_JINA, ARCH_JINA_QKNORM, POOL_MEAN, BertConfig, alibi_slopes, config_from_hf,
                                        from_jina_state_dict, param_count, synth_params, synth_token_batch, tensor_table,
                                        to_state_dict)

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "jina_golden.npz"))


def case_cfg(name):
    m = GOLD[name + "/meta"]
    cfg = BertConfig(vocab_size=int(m[0]), hidden=int(m[1]), layers=int(m[2]), heads=int(m[3]), intermediate=int(m[4]),
                     max_position=int(m[5]), pooling=POOL_MEAN, arch=int(m[11]))
    return cfg, int(m[6]), int(m[7]), int(m[8]), int(m[9]), bool(m[10])


def test_jina_layout_and_generator_identity(oracle):
    bert = BertConfig(vocab_size=512, hidden=128, layers=2, heads=2, intermediate=256)
    H, I = bert.hidden, bert.intermediate
    for arch, extra in ((ARCH_JINA, 0), (ARCH_JINA_QKNORM, 4 * H)):
        cfg = BertConfig(vocab_size=512, hidden=128, layers=2, heads=2, intermediate=256, pooling=POOL_MEAN, arch=arch)
        # no position table; one more [I, H] + [I] per layer than BERT (+ four [H] vectors with the query / key LayerNorms)
        assert param_count(cfg) == param_count(bert) - bert.max_position * H + cfg.layers * (I * H + I + extra)
        assert oracle.bert_param_count(cfg) == param_count(cfg)
        assert np.array_equal(oracle.bert_synth_params(cfg, 9), synth_params(cfg, 9))
        names = [n for n, _, _ in tensor_table(cfg)]
        assert "embeddings.position_embeddings.weight" not in names
        assert ("encoder.layer.1.attention.self.layer_norm_k.bias" in names) == (arch == ARCH_JINA_QKNORM)
    # the LayerNorm slots are generated as LayerNorm parameters (gamma around 1)
    cfg = BertConfig(vocab_size=512, hidden=128, layers=1, heads=2, intermediate=256, pooling=POOL_MEAN, arch=ARCH_JINA_QKNORM)
    sd = to_state_dict(cfg, synth_params(cfg, 3))
    assert abs(float(sd["encoder.layer.0.attention.self.layer_norm_q.weight"].mean()) - 1.0) < 0.05
    assert abs(float(sd["encoder.layer.0.attention.self.layer_norm_k.bias"].mean())) < 0.05


@pytest.mark.parametrize("heads", [8, 12, 16])
def test_alibi_slopes_match_the_library_builder(oracle, heads):
    want = GOLD[f"slopes/{heads}"]
    assert np.array_equal(oracle.alibi_slopes(heads), want)
    assert np.array_equal(alibi_slopes(heads), want)
    if heads == 12:  # 2^-1 .. 2^-8, then 2^-0.5, 2^-1.5, 2^-2.5, 2^-3.5
        np.testing.assert_allclose(want, [2.0 ** -(i + 1) for i in range(8)] + [2.0 ** -(i + 0.5) for i in range(4)], rtol=1e-7)


@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"] if str(n) != "jina_code_shape"])
def test_oracle_matches_the_torch_statement(oracle, name):
    cfg, wseed, iseed, B, L, ragged = case_cfg(name)
    params = synth_params(cfg, wseed)
    ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
    r = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True, want_layers=True)
    np.testing.assert_allclose(r["pooled"], GOLD[name + "/mean"], atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(r["pooled"], axis=1), 1.0, atol=1e-6)
    valid = mask.astype(bool)
    absmean = np.array([np.abs(h[valid]).mean() for h in r["layers"]])
    np.testing.assert_allclose(absmean, GOLD[name + "/layer_absmean"], rtol=1e-5)
    H = cfg.hidden
    probe = np.array([[h[0, 0, 0], h[B - 1, 1, 7], h[0, mask[0].sum() - 1, H - 1]] for h in r["layers"]])
    np.testing.assert_allclose(probe, GOLD[name + "/layer_probe"], atol=2e-5)
    np.testing.assert_allclose(r["hidden"][0, 0], GOLD[name + "/last_row0"], atol=2e-5)


def test_oracle_matches_at_the_published_shape(oracle):
    """12 x 768, 12 heads of 64, intermediate 3072, vocab 61056 (jina-embeddings-v2-base-code's config.json), 4 x 128 tokens."""
    cfg, wseed, iseed, B, L, ragged = case_cfg("jina_code_shape")
    params = oracle.bert_synth_params(cfg, wseed)
    ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
    r = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True)
    np.testing.assert_allclose(r["pooled"], GOLD["jina_code_shape/mean"], atol=1e-5)
    np.testing.assert_allclose(r["hidden"][0, 0], GOLD["jina_code_shape/last_row0"], atol=1e-4)
    mean = GOLD["jina_code_shape/mean"]
    assert (mean @ mean.T)[~np.eye(len(mean), dtype=bool)].max() < 0.999  # the synthetic model tells sequences apart


def test_positions_matter_and_padding_does_not_leak(oracle):
    cfg, wseed, iseed, B, L, _ = case_cfg("dh64_L48")
    params = synth_params(cfg, wseed)
    ids, mask = synth_token_batch(cfg, iseed, B, L, True)
    base = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
    ids2 = np.where(mask == 1, ids, 77).astype(np.int32)  # garbage in the padded positions changes nothing
    assert np.abs(oracle.bert_forward(cfg, params, ids2, mask)["pooled"] - base).max() < 1e-6
    # with no position table, the ALiBi bias is the only thing that sees token order
    b = int(np.argmax(mask.sum(1)))
    ids3 = ids.copy()
    ids3[b, 3], ids3[b, 29] = ids[b, 29], ids[b, 3]
    assert ids3[b, 3] != ids[b, 3]
    moved = np.abs(oracle.bert_forward(cfg, params, ids3, mask)["pooled"][b] - base[b]).max()
    assert moved > 1e-5, moved


def test_jina_checkpoint_names_map_onto_the_flat_block():
    """config.json keys and parameter names of the two JinaBert modelling files: `mlp.up_gated_layer` (value rows first,
    qk-post-norm file) against `mlp.gated_layers` (activated rows first), and where the QK LayerNorm comes from."""
    hf = {"model_type": "bert", "position_embedding_type": "alibi", "feed_forward_type": "geglu", "hidden_act": "gelu",
          "vocab_size": 512, "hidden_size": 128, "num_attention_heads": 2, "num_hidden_layers": 2, "intermediate_size": 256,
          "max_position_embeddings": 8192, "type_vocab_size": 2, "layer_norm_eps": 1e-12,
          "auto_map": {"AutoModel": "jinaai/jina-bert-v2-qk-post-norm--modeling_bert.JinaBertModel"}}
    cfg = config_from_hf(hf)
    assert (cfg.arch, cfg.pooling, cfg.max_position, cfg.hidden, cfg.intermediate) == (ARCH_JINA_QKNORM, POOL_MEAN, 512, 128, 256)
    hf2 = dict(hf, auto_map={"AutoModel": "jinaai/jina-bert-implementation--modeling_bert.JinaBertModel"})
    cfg2 = config_from_hf(hf2)
    assert cfg2.arch == ARCH_JINA
    with pytest.raises(ValueError):
        config_from_hf(dict(hf, feed_forward_type="original"))
    rng = np.random.default_rng(5)
    for c, up_name, down_name, value_first in ((cfg, "mlp.up_gated_layer", "mlp.down_layer", True),
                                               (cfg2, "mlp.gated_layers", "mlp.wo", False)):
        flat = rng.standard_normal(param_count(c)).astype(np.float32)
        ours = to_state_dict(c, flat)
        theirs = {}
        for name, a in ours.items():
            if ".intermediate." in name or ".output.dense." in name or ".output.LayerNorm." in name:
                if "attention" in name:
                    theirs[name] = a
                continue
            theirs[name] = a
        for l in range(c.layers):
            p = f"encoder.layer.{l}."
            v, g = ours[p + "intermediate.dense.weight"], ours[p + "intermediate.gate.weight"]
            theirs[p + up_name + ".weight"] = np.concatenate([v, g] if value_first else [g, v])
            theirs[p + down_name + ".weight"] = ours[p + "output.dense.weight"]
            theirs[p + down_name + ".bias"] = ours[p + "output.dense.bias"]
            theirs[p + "mlp.layernorm.weight"] = ours[p + "output.LayerNorm.weight"]
            theirs[p + "mlp.layernorm.bias"] = ours[p + "output.LayerNorm.bias"]
        back = to_state_dict(c, from_jina_state_dict(c, theirs))
        for name, a in ours.items():
            if ".intermediate." in name and name.endswith(".bias"):
                assert not back[name].any()  # the checkpoint's up projection has no bias
            else:
                assert np.array_equal(back[name], a), name
        broken = dict(theirs)
        del broken["encoder.layer.1." + up_name + ".weight"]
        with pytest.raises(ValueError):
            from_jina_state_dict(c, broken)


# ---- the ONNX export (what fastembed caches for the registry entry) -------------------------------------------------------

ONNX_CFG = BertConfig(vocab_size=48, hidden=64, layers=2, heads=2, intermediate=128, max_position=512, pooling=POOL_MEAN,
                      arch=ARCH_JINA_QKNORM)


def load_onnx(gpu_lib, path, cfg):
    import ctypes as C

    from codesearch_amd import _lib

    c = cfg.to_c()
    out = np.full(param_count(cfg), np.nan, np.float32)
    rc = gpu_lib.cs_bert_params_from_onnx(str(path).encode(), C.byref(c), out.ctypes.data_as(_lib.f32p), out.size)
    return rc, out, gpu_lib.cs_last_error().decode()


def test_onnx_reader_reads_a_jinabert_export_written_by_torchs_own_exporter(gpu_lib, oracle):
    """tests/golden/jina_tiny_export.onnx (make_jina_onnx_fixture.py): a JinaBert-shaped module (the qk-post-norm modelling
    file's names) through torch.onnx's TorchScript exporter — Linear weights as anonymous transposed initialisers behind the
    Add of their named bias, the bias-free up_gated_layer identified only by its [H, 2I] shape and its position.  The flat
    block must equal the one built from the state dict the file was exported from, bit for bit, and the CPU oracle run on it
    must reproduce the exporting module's own output (stored with the state)."""
    from codesearch_amd import _lib

    gold = os.path.join(os.path.dirname(__file__), "golden")
    state = dict(np.load(os.path.join(gold, "jina_tiny_export_state.npz")))
    path = os.path.join(gold, "jina_tiny_export.onnx")
    rc, got, err = load_onnx(gpu_lib, path, ONNX_CFG)
    assert rc == _lib.CS_OK, err
    assert np.array_equal(got, from_jina_state_dict(ONNX_CFG, state))
    pooled = oracle.bert_forward(ONNX_CFG, got, state["query_ids"], state["query_mask"])["pooled"]
    np.testing.assert_allclose(pooled, state["query_pooled"], atol=2e-6)
    # the file's own tensors contradict a configuration without the query / key LayerNorms; another depth or width does not fit
    rc, _, err = load_onnx(gpu_lib, path, BertConfig(**{**ONNX_CFG.__dict__, "arch": ARCH_JINA}))
    assert rc == _lib.CS_ERR_DIM_MISMATCH and "query / key LayerNorm" in err
    rc, _, err = load_onnx(gpu_lib, path, BertConfig(**{**ONNX_CFG.__dict__, "layers": 3}))
    assert rc != _lib.CS_OK
    rc, _, err = load_onnx(gpu_lib, path, BertConfig(**{**ONNX_CFG.__dict__, "intermediate": 256}))
    assert rc == _lib.CS_ERR_BAD_ARG and "gated up projections" in err


def test_onnx_reader_reads_the_first_modelling_files_arrangement(gpu_lib, oracle, tmp_path):
    """jinaai/jina-bert-implementation's names (no query / key LayerNorm, mlp.gated_layers whose FIRST half goes through the GELU,
    mlp.wo), exported here by the fixture script's module in that arrangement."""
    import importlib.util

    from codesearch_amd import _lib

    spec = importlib.util.spec_from_file_location("make_jina_onnx_fixture", os.path.join(os.path.dirname(__file__), "golden", "make_jina_onnx_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = mod.write(str(tmp_path), "first_file", first_file=True)
    state = dict(np.load(os.path.join(str(tmp_path), "first_file_state.npz")))
    cfg = BertConfig(**{**ONNX_CFG.__dict__, "arch": ARCH_JINA})
    rc, got, err = load_onnx(gpu_lib, path, cfg)
    assert rc == _lib.CS_OK, err
    assert np.array_equal(got, from_jina_state_dict(cfg, state))
    pooled = oracle.bert_forward(cfg, got, state["query_ids"], state["query_mask"])["pooled"]
    np.testing.assert_allclose(pooled, state["query_pooled"], atol=2e-6)
    rc, _, err = load_onnx(gpu_lib, path, ONNX_CFG)
    assert rc == _lib.CS_ERR_DIM_MISMATCH and "holds no query / key LayerNorm" in err


def test_onnx_reader_reads_an_opset_17_export(gpu_lib, tmp_path):
    """Opset 17: LayerNormalization nodes instead of the decomposed chain; the Linear weights still anonymous behind their biases."""
    import importlib.util

    from codesearch_amd import _lib
    from tests.onnx_dump import read

    spec = importlib.util.spec_from_file_location("make_jina_onnx_fixture", os.path.join(os.path.dirname(__file__), "golden", "make_jina_onnx_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = mod.write(str(tmp_path), "opset17", opset=17)
    assert any(op == "LayerNormalization" for op, _, _, _ in read(path)["nodes"])
    state = dict(np.load(os.path.join(str(tmp_path), "opset17_state.npz")))
    rc, got, err = load_onnx(gpu_lib, path, ONNX_CFG)
    assert rc == _lib.CS_OK, err
    assert np.array_equal(got, from_jina_state_dict(ONNX_CFG, state))


def test_config_from_dir_takes_the_variant_from_the_weights_file(gpu_lib, tmp_path):
    """cs_bert_config_from_dir on a JinaBert directory: with no weights file the modelling file named by auto_map decides whether
    the query / key rows are LayerNorm'ed; a model.safetensors or an ONNX export next to config.json overrides it with what its own
    tensors say (host-only: no GPU involved)."""
    import ctypes as C
    import json
    import shutil

    from safetensors.numpy import save_file

    from codesearch_amd import _lib

    def arch_of(d):
        c = _lib.BertConfig()
        _lib.check(gpu_lib.cs_bert_config_from_dir(str(d).encode(), -1, C.byref(c)))
        return c.arch, c.pooling, c.max_position

    def config(repo):
        return {"model_type": "bert", "position_embedding_type": "alibi", "feed_forward_type": "geglu", "hidden_act": "gelu", "vocab_size": 48,
                "hidden_size": 64, "num_attention_heads": 2, "num_hidden_layers": 2, "intermediate_size": 128, "max_position_embeddings": 8192,
                "type_vocab_size": 2, "layer_norm_eps": 1e-12,
                "auto_map": {"AutoModel": repo + "--modeling_bert.JinaBertModel"}}

    gold = os.path.join(os.path.dirname(__file__), "golden")
    d = tmp_path / "m"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(config("jinaai/jina-bert-implementation")))
    assert arch_of(d) == (ARCH_JINA, POOL_MEAN, 512)                      # the config alone
    (d / "config.json").write_text(json.dumps(config("jinaai/jina-bert-v2-qk-post-norm")))
    assert arch_of(d)[0] == ARCH_JINA_QKNORM
    # a safetensors file without the query / key LayerNorms overrides the qk-post-norm config ...
    save_file({"encoder.layer.0.attention.self.query.weight": np.zeros((64, 64), np.float32)}, str(d / "model.safetensors"))
    assert arch_of(d)[0] == ARCH_JINA
    # ... and the exported graph, whose initialisers hold them, overrides the first modelling file's config
    (d / "model.safetensors").unlink()
    (d / "config.json").write_text(json.dumps(config("jinaai/jina-bert-implementation")))
    (d / "onnx").mkdir()
    shutil.copy(os.path.join(gold, "jina_tiny_export.onnx"), d / "onnx" / "model.onnx")
    assert arch_of(d)[0] == ARCH_JINA_QKNORM
    # an unreadable weights file leaves the decision with the config
    (d / "onnx" / "model.onnx").write_bytes(b"\x00\x01garbage")
    assert arch_of(d)[0] == ARCH_JINA
