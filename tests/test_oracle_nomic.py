"""Pins the CS_ARCH_NOMIC branch of the encoder oracle (oracle/bert_oracle.c: rotary Q / K, swiglu feed-forward, no
position table) to the committed golden vectors of tests/golden/make_nomic_golden.py (a float64 torch statement built on
transformers' rotate_half / apply_rotary_pos_emb), and the NomicBert checkpoint-name mapping.  CPU only."""
import os

import numpy as np
import pytest

from codesearch_amd.bert_params import (ARCH_NOMIC, POOL_MEAN, BertConfig, from_nomic_state_dict, nomic_config_from_hf,
                                        param_count, synth_params, synth_token_batch, tensor_table, to_state_dict)

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "nomic_golden.npz"))


def case_cfg(name):
    m = GOLD[name + "/meta"]
    cfg = BertConfig(vocab_size=int(m[0]), hidden=int(m[1]), layers=int(m[2]), heads=int(m[3]), intermediate=int(m[4]),
                     max_position=int(m[5]), pooling=POOL_MEAN, arch=ARCH_NOMIC, rotary_base=1000.0)
    return cfg, int(m[6]), int(m[7]), int(m[8]), int(m[9]), bool(m[10])


def test_nomic_layout_and_generator_identity(oracle):
    cfg = BertConfig(vocab_size=512, hidden=128, layers=2, heads=2, intermediate=256, pooling=POOL_MEAN, arch=ARCH_NOMIC,
                     rotary_base=1000.0)
    H, I = cfg.hidden, cfg.intermediate
    # no position table; one more [I, H] + [I] per layer than BERT
    bert = BertConfig(vocab_size=512, hidden=128, layers=2, heads=2, intermediate=256)
    assert param_count(cfg) == param_count(bert) - bert.max_position * H + cfg.layers * (I * H + I)
    assert oracle.bert_param_count(cfg) == param_count(cfg)
    assert np.array_equal(oracle.bert_synth_params(cfg, 9), synth_params(cfg, 9))
    names = [n for n, _, _ in tensor_table(cfg)]
    assert "embeddings.position_embeddings.weight" not in names
    assert names.index("encoder.layer.0.intermediate.gate.weight") == names.index("encoder.layer.0.intermediate.dense.bias") + 1


@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"] if str(n) != "nomic_shape"])
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
    """12 x 768, 12 heads of 64, n_inner 3072, rotary base 1000 (nomic-embed-text-v1.5's config.json), 4 x 128 tokens."""
    cfg, wseed, iseed, B, L, ragged = case_cfg("nomic_shape")
    params = oracle.bert_synth_params(cfg, wseed)
    ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
    r = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True)
    np.testing.assert_allclose(r["pooled"], GOLD["nomic_shape/mean"], atol=1e-5)
    np.testing.assert_allclose(r["hidden"][0, 0], GOLD["nomic_shape/last_row0"], atol=1e-4)
    mean = GOLD["nomic_shape/mean"]
    assert (mean @ mean.T)[~np.eye(len(mean), dtype=bool)].max() < 0.999  # the synthetic model tells sequences apart


def test_positions_matter_and_padding_does_not_leak(oracle):
    cfg, wseed, iseed, B, L, _ = case_cfg("dh64_L48")
    params = synth_params(cfg, wseed)
    ids, mask = synth_token_batch(cfg, iseed, B, L, True)
    base = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
    # garbage in the padded positions changes nothing
    ids2 = np.where(mask == 1, ids, 77).astype(np.int32)
    assert np.abs(oracle.bert_forward(cfg, params, ids2, mask)["pooled"] - base).max() < 1e-6
    # with no position table, the rotary map is the only thing that sees token order: swapping two inner tokens of a
    # full row must move the embedding
    b = int(np.argmax(mask.sum(1)))
    ids3 = ids.copy()
    ids3[b, 3], ids3[b, 9] = ids[b, 9], ids[b, 3]
    assert ids3[b, 3] != ids[b, 3]
    moved = np.abs(oracle.bert_forward(cfg, params, ids3, mask)["pooled"][b] - base[b]).max()
    assert moved > 1e-4, moved


def test_nomic_checkpoint_names_map_onto_the_flat_block():
    """The model repository's parameter names (emb_ln, encoder.layers.N.attn.Wqkv / out_proj, norm1, mlp.fc11 / fc12 / fc2,
    norm2; no Linear biases) and its config.json keys."""
    hf = {"model_type": "nomic_bert", "vocab_size": 512, "n_embd": 128, "n_head": 2, "n_layer": 2, "n_inner": 256,
          "n_positions": 8192, "type_vocab_size": 2, "layer_norm_epsilon": 1e-12, "rotary_emb_base": 1000,
          "rotary_emb_fraction": 1.0, "rotary_emb_interleaved": False, "rotary_emb_scale_base": None,
          "activation_function": "swiglu", "prenorm": False, "qkv_proj_bias": False, "mlp_fc1_bias": False}
    cfg = nomic_config_from_hf(hf)
    assert (cfg.arch, cfg.rotary_base, cfg.hidden, cfg.heads, cfg.intermediate, cfg.max_position, cfg.pooling) == \
        (ARCH_NOMIC, 1000.0, 128, 2, 256, 512, POOL_MEAN)
    flat = synth_params(cfg, 5)
    ours = to_state_dict(cfg, flat)
    sd = {"embeddings.word_embeddings.weight": ours["embeddings.word_embeddings.weight"],
          "embeddings.token_type_embeddings.weight": ours["embeddings.token_type_embeddings.weight"],
          "emb_ln.weight": ours["embeddings.LayerNorm.weight"], "emb_ln.bias": ours["embeddings.LayerNorm.bias"]}
    for l in range(cfg.layers):
        a, b = f"encoder.layer.{l}.", f"encoder.layers.{l}."
        sd[b + "attn.Wqkv.weight"] = np.concatenate([ours[a + f"attention.self.{r}.weight"] for r in ("query", "key", "value")])
        sd[b + "attn.out_proj.weight"] = ours[a + "attention.output.dense.weight"]
        sd[b + "mlp.fc11.weight"] = ours[a + "intermediate.dense.weight"]
        sd[b + "mlp.fc12.weight"] = ours[a + "intermediate.gate.weight"]
        sd[b + "mlp.fc2.weight"] = ours[a + "output.dense.weight"]
        for n, o in (("norm1", "attention.output.LayerNorm"), ("norm2", "output.LayerNorm")):
            sd[b + n + ".weight"], sd[b + n + ".bias"] = ours[a + o + ".weight"], ours[a + o + ".bias"]
    got = to_state_dict(cfg, from_nomic_state_dict(cfg, sd))
    for name, _, kind in tensor_table(cfg):
        if kind == "bias":
            assert not got[name].any(), name  # the checkpoint holds no Linear bias
        else:
            assert np.array_equal(got[name], ours[name]), name
    for bad in ({"rotary_emb_interleaved": True}, {"activation_function": "gelu"}, {"prenorm": True}, {"rotary_emb_fraction": 0.5}):
        with pytest.raises(ValueError):
            nomic_config_from_hf({**hf, **bad})


def nomic_snapshot(tmp, cfg, flat, dtype=None, with_biases=False):
    """A NomicBert HF snapshot directory (config.json with the repository's keys + model.safetensors with its names)."""
    import json

    from safetensors.numpy import save_file

    ours = to_state_dict(cfg, flat)
    sd = {"embeddings.word_embeddings.weight": ours["embeddings.word_embeddings.weight"],
          "embeddings.token_type_embeddings.weight": ours["embeddings.token_type_embeddings.weight"],
          "emb_ln.weight": ours["embeddings.LayerNorm.weight"], "emb_ln.bias": ours["embeddings.LayerNorm.bias"]}
    for l in range(cfg.layers):
        a, b = f"encoder.layer.{l}.", f"encoder.layers.{l}."
        sd[b + "attn.Wqkv.weight"] = np.concatenate([ours[a + f"attention.self.{r}.weight"] for r in ("query", "key", "value")])
        if with_biases:
            sd[b + "attn.Wqkv.bias"] = np.concatenate([ours[a + f"attention.self.{r}.bias"] for r in ("query", "key", "value")])
        for theirs, mine in (("attn.out_proj", "attention.output.dense"), ("mlp.fc11", "intermediate.dense"),
                             ("mlp.fc12", "intermediate.gate"), ("mlp.fc2", "output.dense")):
            sd[b + theirs + ".weight"] = ours[a + mine + ".weight"]
            if with_biases:
                sd[b + theirs + ".bias"] = ours[a + mine + ".bias"]
        for n, o in (("norm1", "attention.output.LayerNorm"), ("norm2", "output.LayerNorm")):
            sd[b + n + ".weight"], sd[b + n + ".bias"] = ours[a + o + ".weight"], ours[a + o + ".bias"]
    hf = {"model_type": "nomic_bert", "architectures": ["NomicBertModel"], "vocab_size": cfg.vocab_size, "n_embd": cfg.hidden,
          "n_head": cfg.heads, "n_layer": cfg.layers, "n_inner": cfg.intermediate, "n_positions": 8192,
          "type_vocab_size": cfg.type_vocab_size, "layer_norm_epsilon": cfg.layer_norm_eps, "rotary_emb_base": int(cfg.rotary_base),
          "rotary_emb_fraction": 1.0, "rotary_emb_interleaved": False, "rotary_emb_scale_base": None,
          "rotary_scaling_factor": None, "activation_function": "swiglu", "prenorm": False, "qkv_proj_bias": with_biases,
          "mlp_fc1_bias": with_biases, "mlp_fc2_bias": with_biases}
    tmp.mkdir(exist_ok=True)
    (tmp / "config.json").write_text(json.dumps(hf))
    save_file({k: np.ascontiguousarray(v if dtype is None else v.astype(dtype)) for k, v in sd.items()},
              str(tmp / "model.safetensors"), metadata={"format": "pt"})
    return hf


def test_c_abi_loader_reads_a_nomic_snapshot(tmp_path, gpu_lib):
    """cs_bert_config_from_dir / cs_bert_params_from_safetensors (csrc/checkpoint.cpp, host-only) on a NomicBert snapshot:
    the repository's config keys and tensor names (fused Wqkv cut into thirds, absent Linear biases left zero) give the
    config and flat block of the Python mapping; refusals are worded."""
    import ctypes as C
    import json

    from codesearch_amd import _lib

    cfg = BertConfig(vocab_size=300, hidden=384, layers=2, heads=12, intermediate=1536, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_NOMIC, rotary_base=1000.0)
    flat = synth_params(cfg, 21)

    def load(d, pooling=-1):
        c = _lib.BertConfig()
        _lib.check(gpu_lib.cs_bert_config_from_dir(str(d).encode(), pooling, C.byref(c)))
        n = int(gpu_lib.cs_bert_param_count(C.byref(c)))
        out = np.full(n, np.nan, np.float32)
        _lib.check(gpu_lib.cs_bert_params_from_safetensors(str(d / "model.safetensors").encode(), C.byref(c),
                                                          out.ctypes.data_as(_lib.f32p), n))
        return c, out

    hf = nomic_snapshot(tmp_path / "plain", cfg, flat)
    c, got = load(tmp_path / "plain")
    assert (c.arch, c.vocab_size, c.hidden, c.layers, c.heads, c.intermediate, c.max_position, c.type_vocab_size, c.pooling) == \
        (ARCH_NOMIC, 300, 384, 2, 12, 1536, 512, 2, POOL_MEAN)  # mean pooling without a 1_Pooling module: fastembed's choice
    assert c.rotary_base == 1000.0 and int(gpu_lib.cs_bert_param_count(C.byref(c))) == param_count(cfg)
    # what the Python mapping gives for the same file: Linear biases zero, everything else bit for bit
    from safetensors.numpy import load_file
    exp = from_nomic_state_dict(cfg, load_file(str(tmp_path / "plain" / "model.safetensors")))
    assert np.array_equal(got, exp)
    # a checkpoint that does hold Linear biases: taken
    nomic_snapshot(tmp_path / "biased", cfg, flat, with_biases=True)
    _, got = load(tmp_path / "biased")
    assert np.array_equal(got, flat)
    # f16 file
    nomic_snapshot(tmp_path / "f16", cfg, flat, dtype=np.float16)
    _, got16 = load(tmp_path / "f16")
    keep = exp != 0
    assert np.array_equal(got16[keep], exp.astype(np.float16).astype(np.float32)[keep])
    assert load(tmp_path / "plain", pooling=0)[0].pooling == 0  # an explicit pooling wins

    def expect(d, code, text):
        try:
            load(d)
        except _lib.CsError as e:
            assert e.code == code and text in str(e), str(e)
        else:
            raise AssertionError("expected " + text)

    bad = tmp_path / "bad"
    for change, text in (({"rotary_emb_interleaved": True}, "this nomic_bert configuration is not built"),
                         ({"activation_function": "gelu"}, "this nomic_bert configuration is not built"),
                         ({"rotary_scaling_factor": 2, "max_trained_positions": 64}, "this nomic_bert configuration is not built"),
                         ({"prenorm": True}, "this nomic_bert configuration is not built")):
        nomic_snapshot(bad, cfg, flat)
        (bad / "config.json").write_text(json.dumps({**hf, **change}))
        expect(bad, _lib.CS_ERR_UNSUPPORTED, text)
    # the 8k-context checkpoints' dynamic-NTK factor only changes the rotary base beyond max_trained_positions (2,048):
    # at the 512 positions this loader runs the table is the same, so such a config loads
    nomic_snapshot(bad, cfg, flat)
    (bad / "config.json").write_text(json.dumps({**hf, "rotary_scaling_factor": 2}))
    c2, got2 = load(bad)
    assert c2.max_position == 512 and np.array_equal(got2, exp)
    nomic_snapshot(bad, cfg, flat)
    (bad / "config.json").write_text(json.dumps({k: v for k, v in hf.items() if k != "n_head"}))
    expect(bad, _lib.CS_ERR_BAD_ARG, "lacks a nomic_bert size field")
    (bad / "config.json").write_text(json.dumps({**hf, "n_inner": 1024}))
    expect(bad, _lib.CS_ERR_DIM_MISMATCH, "encoder.layers.0.mlp.fc11.weight has shape [1536, 384], config.json implies [1024, 384]")
    (bad / "config.json").write_text(json.dumps({**hf, "n_layer": 3}))
    expect(bad, _lib.CS_ERR_BAD_ARG, "encoder.layers.2.attn.Wqkv.weight is missing")


def nomic_state_dict(cfg, flat):
    """The flat block under the model repository's tensor names (what nomic_snapshot writes into model.safetensors)."""
    ours = to_state_dict(cfg, flat)
    sd = {"embeddings.word_embeddings.weight": ours["embeddings.word_embeddings.weight"],
          "embeddings.token_type_embeddings.weight": ours["embeddings.token_type_embeddings.weight"],
          "emb_ln.weight": ours["embeddings.LayerNorm.weight"], "emb_ln.bias": ours["embeddings.LayerNorm.bias"]}
    for l in range(cfg.layers):
        a, b = f"encoder.layer.{l}.", f"encoder.layers.{l}."
        sd[b + "attn.Wqkv.weight"] = np.concatenate([ours[a + f"attention.self.{r}.weight"] for r in ("query", "key", "value")])
        for theirs, mine in (("attn.out_proj", "attention.output.dense"), ("mlp.fc11", "intermediate.dense"),
                             ("mlp.fc12", "intermediate.gate"), ("mlp.fc2", "output.dense")):
            sd[b + theirs + ".weight"] = ours[a + mine + ".weight"]
        for n, o in (("norm1", "attention.output.LayerNorm"), ("norm2", "output.LayerNorm")):
            sd[b + n + ".weight"], sd[b + n + ".bias"] = ours[a + o + ".weight"], ours[a + o + ".bias"]
    return sd


def load_nomic_onnx(gpu_lib, path, cfg):
    import ctypes as C

    from codesearch_amd import _lib

    c = cfg.to_c()
    out = np.full(param_count(cfg), np.nan, np.float32)
    rc = gpu_lib.cs_bert_params_from_onnx(str(path).encode(), C.byref(c), out.ctypes.data_as(_lib.f32p), out.size)
    return rc, out, gpu_lib.cs_last_error().decode()


def test_onnx_reader_reads_a_nomic_export_written_by_torchs_own_exporter(gpu_lib):
    """tests/golden/nomic_tiny_export.onnx (make_nomic_onnx_fixture.py): a NomicBert-shaped module through torch.onnx's
    TorchScript exporter — bias-free Linear weights as anonymous transposed initialisers, silu as Sigmoid + Mul, the
    attention products as MatMuls between activations.  The flat block must equal the one built from the state dict it was
    exported from, bit for bit (what fastembed caches for the registry's Nomic entries is such a file, embedder.rs:36-37)."""
    from codesearch_amd import _lib

    gold = os.path.join(os.path.dirname(__file__), "golden")
    cfg = BertConfig(vocab_size=48, hidden=64, layers=2, heads=2, intermediate=128, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_NOMIC, rotary_base=1000.0)
    rc, got, err = load_nomic_onnx(gpu_lib, os.path.join(gold, "nomic_tiny_export.onnx"), cfg)
    assert rc == _lib.CS_OK, err
    assert np.array_equal(got, from_nomic_state_dict(cfg, dict(np.load(os.path.join(gold, "nomic_tiny_export_state.npz")))))
    # a config with another layer count does not match the file's weight products
    rc, _, err = load_nomic_onnx(gpu_lib, os.path.join(gold, "nomic_tiny_export.onnx"), BertConfig(**{**cfg.__dict__, "layers": 3}))
    assert rc == _lib.CS_ERR_BAD_ARG and "weight products" in err


@pytest.mark.parametrize("quantized,per_channel,gate_first", [(False, False, False), (False, False, True), (True, False, False),
                                                              (True, True, True)])
def test_onnx_reader_finds_nomic_weights_by_graph_structure(tmp_path, gpu_lib, quantized, per_channel, gate_first):
    """The same layout at the models' width from tests/onnx_writer.py: f32 and dynamically quantised (the *Q entry's
    model_quantized.onnx: read as (q - zero_point) * scale, per tensor or per output channel), the gate found by the Sigmoid
    its product feeds whichever of fc11 / fc12 comes first in the file, F16 initialisers, a module prefix."""
    from codesearch_amd import _lib
    from tests import onnx_writer

    cfg = BertConfig(vocab_size=300, hidden=384, layers=2, heads=12, intermediate=1536, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_NOMIC, rotary_base=1000.0)
    flat = synth_params(cfg, 5)
    sd = nomic_state_dict(cfg, flat)
    deq = {}
    p = tmp_path / "model.onnx"
    p.write_bytes(onnx_writer.nomic_onnx(sd, cfg.layers, quantized=quantized, per_channel=per_channel, gate_first=gate_first,
                                         prefix="model." if gate_first else "", dequantized=deq))
    rc, got, err = load_nomic_onnx(gpu_lib, p, cfg)
    assert rc == _lib.CS_OK, err
    want_sd = dict(sd)
    want_sd.update(deq)
    want = from_nomic_state_dict(cfg, want_sd)   # (Linear biases: none in the file -> zero slots)
    assert np.array_equal(got, want)
    if quantized:
        assert np.abs(got - from_nomic_state_dict(cfg, sd)).max() < 2e-3 and not np.array_equal(got, from_nomic_state_dict(cfg, sd))
    if not quantized and not gate_first:  # F16 payloads
        p.write_bytes(onnx_writer.nomic_onnx(sd, cfg.layers, dtype=onnx_writer.FLOAT16))
        rc, got16, err = load_nomic_onnx(gpu_lib, p, cfg)
        assert rc == _lib.CS_OK, err
        h = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
        assert np.array_equal(got16, from_nomic_state_dict(cfg, h))


def test_onnx_reader_refusals_for_nomic(tmp_path, gpu_lib):
    """A file that is not an ONNX graph, a BERT export under a nomic_bert config, and an encoder family the reader has no
    branch for are refused with worded errors."""
    import ctypes as C

    from codesearch_amd import _lib
    cfg = BertConfig(vocab_size=300, hidden=384, layers=1, heads=12, intermediate=1536, pooling=POOL_MEAN, arch=ARCH_NOMIC,
                     rotary_base=1000.0)
    d = tmp_path / "snap"
    nomic_snapshot(d, cfg, synth_params(cfg, 1))
    (d / "model.safetensors").rename(d / "elsewhere.bin")
    (d / "model.onnx").write_bytes(b"\x08\x07")
    h = C.c_void_p()
    rc = gpu_lib.cs_embedder_create_from_dir(str(d).encode(), -1, 0, C.byref(h))
    assert rc == _lib.CS_ERR_BAD_ARG and "holds no graph" in gpu_lib.cs_last_error().decode()
    other = BertConfig(vocab_size=300, hidden=384, layers=1, heads=12, intermediate=1536, pooling=POOL_MEAN, arch=9)
    rc, _, err = load_nomic_onnx(gpu_lib, d / "model.onnx", other)
    assert rc == _lib.CS_ERR_UNSUPPORTED and "exports are read from ONNX files" in err
