"""Parity tests for the JinaBert encoder family (cs_bert_config.arch = CS_ARCH_JINA / CS_ARCH_JINA_QKNORM: the reference
registry's jina-embeddings-v2-base-code entry, /root/reference/src/embed/embedder.rs:40-41, :69, :92, :112): HIP kernels
through the C ABI — the BERT dense layers plus the ALiBi bias inside the attention kernels (attention_shx_body.hpp), the
GELU gate of the feed-forward (gemm_wide.hip GW_OUT_GEGLU / nomic.hip) and the query / key LayerNorm (nomic.hip) — against
the CPU oracle and the committed float64 golden vectors (tests/golden/make_jina_golden.py).  Needs an MI355X.

Bar as for the BERT encoders: within 2e-5 of the fp32 oracle, 3e-5 of the float64 golden (north_star: 1e-4)."""
import json
import os

import numpy as np
import pytest

from codesearch_amd.bert_params import (ARCH_JINA, ARCH_JINA_QKNORM, POOL_CLS, POOL_MEAN, BertConfig, synth_params,
                                        synth_token_batch, to_state_dict)

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "jina_golden.npz"))
TOL_ORACLE = 2e-5
TOL_GOLDEN = 3e-5


def case_cfg(name):
    m = GOLD[name + "/meta"]
    cfg = BertConfig(vocab_size=int(m[0]), hidden=int(m[1]), layers=int(m[2]), heads=int(m[3]), intermediate=int(m[4]),
                     max_position=int(m[5]), pooling=POOL_MEAN, arch=int(m[11]))
    return cfg, int(m[6]), int(m[7]), int(m[8]), int(m[9]), bool(m[10])


@pytest.fixture(scope="module")
def FE(gpu_lib):
    from codesearch_amd import FastEmbedder, ModelType

    assert gpu_lib.cs_device_count() >= 1
    return lambda cfg, **kw: FastEmbedder(ModelType.JinaEmbeddingsV2BaseCode, config=cfg, **kw)


@pytest.mark.parametrize("gemm_mode", ["split", "f32"])
@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"] if str(n) != "jina_code_shape"])
def test_small_cases_vs_golden_and_oracle(FE, oracle, name, gemm_mode):
    """Both arithmetic modes (split-f16 operands on the f16 MFMA, the default; the exact-f32 kernels), both head widths,
    with and without the query / key LayerNorm, 12 and 16 heads (the slope rule's two branches)."""
    cfg, wseed, iseed, B, L, ragged = case_cfg(name)
    ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
    params = synth_params(cfg, wseed)
    emb = FE(cfg, seed=wseed, gemm_mode=gemm_mode)
    got = emb.embed_ids(ids, mask)
    ref = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True)
    np.testing.assert_allclose(got, ref["pooled"], atol=TOL_ORACLE)
    np.testing.assert_allclose(got, GOLD[name + "/mean"], atol=TOL_GOLDEN)
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    hid = emb.last_hidden(B * L).reshape(B, L, cfg.hidden)
    valid = mask.astype(bool)
    np.testing.assert_allclose(hid[valid], ref["hidden"][valid], atol=2e-4)
    split, f32, _ = emb.debug_counters()
    assert (split, f32) == ((1, 0) if gemm_mode == "split" else (0, 1))
    emb.close()


def test_published_shape_vs_golden_and_oracle(FE, oracle):
    """jina-embeddings-v2-base-code's own shape (12 x 768, 12 heads of 64, intermediate 3072, vocab 61056, QK LayerNorm)."""
    cfg, wseed, iseed, B, L, ragged = case_cfg("jina_code_shape")
    ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
    emb = FE(cfg, seed=wseed)
    got = emb.embed_ids(ids, mask)
    ref = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, wseed), ids, mask)
    np.testing.assert_allclose(got, ref["pooled"], atol=TOL_ORACLE)
    np.testing.assert_allclose(got, GOLD["jina_code_shape/mean"], atol=TOL_GOLDEN)
    assert emb.debug_counters()[:2] == (1, 0)
    emb.close()


def test_registry_entry_builds_and_runs(FE, oracle):
    """ModelType::JinaEmbeddingsV2BaseCode -> the JinaBert config (embedder.rs:69, :92: 768 dimensions); two layers of it
    here, every dense-layer route by batch size: a few rows (skinny kernels), the reference's 32-chunk call (mid-size
    tiles), an indexing batch (persistent wide kernels with the GELU gate as the up projection's epilogue)."""
    from codesearch_amd import ModelType

    m = ModelType.JinaEmbeddingsV2BaseCode
    assert (m.dimensions(), m.bert_config().arch, m.short_name()) == (768, ARCH_JINA_QKNORM, "jina-code")
    cfg = m.bert_config()
    cfg.layers, cfg.vocab_size = 2, 2048
    emb = FE(cfg, seed=421)
    params = synth_params(cfg, 421)
    for B, L, ragged in ((1, 12, False), (32, 96, True), (128, 128, True)):
        ids, mask = synth_token_batch(cfg, 500 + B, B, L, ragged)
        got = emb.embed_ids(ids, mask)
        ref = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
        np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    assert emb.debug_counters()[1] == 0  # no exact-f32 fallback
    emb.close()


def test_minibatches_and_padding_do_not_change_an_embedding(FE, oracle):
    """The ALiBi bias of a pair of tokens is their own distance: embedding a row alone, in a longer padded batch, or behind
    other rows gives the same vector (the length-grouped runner relies on it)."""
    cfg, wseed, iseed, B, L, _ = case_cfg("dh64_L48")
    ids, mask = synth_token_batch(cfg, iseed, B, L, True)
    emb = FE(cfg, seed=wseed)
    base = emb.embed_ids(ids, mask)
    pad = 16
    ids2 = np.concatenate([ids, np.zeros((B, pad), np.int32)], axis=1)
    mask2 = np.concatenate([mask, np.zeros((B, pad), np.int32)], axis=1)
    np.testing.assert_allclose(emb.embed_ids(ids2, mask2), base, atol=2e-6)
    for b in range(B):
        np.testing.assert_allclose(emb.embed_ids(ids[b:b + 1], mask[b:b + 1])[0], base[b], atol=2e-6)
    emb.close()


def test_cls_pooling_and_refusals(FE, oracle):
    """CLS pooling on this family takes the full last layer (the CLS tail is BERT's); a quantised Jina model is refused with
    a worded error."""
    from codesearch_amd import CsError
    from codesearch_amd.bert_params import quantize_linear_weights

    cfg, wseed, iseed, B, L, _ = case_cfg("dh32_qkn_L64")
    cfg.pooling = POOL_CLS
    ids, mask = synth_token_batch(cfg, iseed, 80, L, False)  # 5,120 token rows: where BERT would take the CLS tail
    emb = FE(cfg, seed=wseed)
    got = emb.embed_ids(ids, mask)
    ref = oracle.bert_forward(cfg, synth_params(cfg, wseed), ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    emb.last_hidden(80 * L)  # the whole last layer was computed
    emb.close()
    small = BertConfig(vocab_size=512, hidden=384, layers=1, heads=12, intermediate=1536, pooling=POOL_MEAN)
    p, wscale = quantize_linear_weights(small, synth_params(small, 3))
    with pytest.raises(CsError, match="dynamic-quantisation mode is not built"):
        jina_small = BertConfig(**{**small.__dict__, "arch": ARCH_JINA})
        FE(jina_small, params=synth_params(jina_small, 3), wscale=wscale)


def jina_snapshot(d, cfg, flat, first_file=False):
    """A JinaBert snapshot directory as the model repository lays it out: config.json with its keys (alibi, geglu, the
    auto_map naming the modelling file) and model.safetensors with its tensor names — mlp.up_gated_layer (value rows first)
    / mlp.down_layer / mlp.layernorm, or the first modelling file's mlp.gated_layers (activated rows first) / mlp.wo."""
    from safetensors.numpy import save_file

    os.makedirs(d, exist_ok=True)
    repo = "jinaai/jina-bert-implementation" if cfg.arch == ARCH_JINA else "jinaai/jina-bert-v2-qk-post-norm"
    hf = {"model_type": "bert", "position_embedding_type": "alibi", "feed_forward_type": "geglu", "hidden_act": "gelu",
          "vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden, "num_attention_heads": cfg.heads,
          "num_hidden_layers": cfg.layers, "intermediate_size": cfg.intermediate, "max_position_embeddings": 8192,
          "type_vocab_size": cfg.type_vocab_size, "layer_norm_eps": cfg.layer_norm_eps, "emb_pooler": "mean",
          "auto_map": {"AutoConfig": repo + "--configuration_bert.JinaBertConfig", "AutoModel": repo + "--modeling_bert.JinaBertModel"}}
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(hf, f)
    ours = to_state_dict(cfg, flat)
    theirs = {}
    for name, a in ours.items():
        if (".intermediate." in name or ".output.dense." in name or ".output.LayerNorm." in name) and ".attention." not in name:
            continue
        theirs[name] = np.ascontiguousarray(a)
    up_name, down_name = ("mlp.gated_layers", "mlp.wo") if first_file else ("mlp.up_gated_layer", "mlp.down_layer")
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        v, g = ours[p + "intermediate.dense.weight"], ours[p + "intermediate.gate.weight"]
        theirs[p + up_name + ".weight"] = np.concatenate([g, v] if first_file else [v, g])
        theirs[p + down_name + ".weight"] = np.ascontiguousarray(ours[p + "output.dense.weight"])
        theirs[p + down_name + ".bias"] = np.ascontiguousarray(ours[p + "output.dense.bias"])
        theirs[p + "mlp.layernorm.weight"] = np.ascontiguousarray(ours[p + "output.LayerNorm.weight"])
        theirs[p + "mlp.layernorm.bias"] = np.ascontiguousarray(ours[p + "output.LayerNorm.bias"])
    save_file(theirs, os.path.join(d, "model.safetensors"))


@pytest.mark.parametrize("arch,first_file", [(ARCH_JINA_QKNORM, False), (ARCH_JINA, True)])
def test_embedder_from_a_jina_snapshot_directory(FE, oracle, tmp_path, arch, first_file):
    """FastEmbedder.from_dir = cs_embedder_create_from_dir + the directory's vocabulary: a JinaBert snapshot written here
    embeds texts like an embedder handed the mapped flat block, and like the oracle — for both modelling files' names."""
    from codesearch_amd import FastEmbedder
    from codesearch_amd.bert_params import from_jina_state_dict
    from codesearch_amd.pipeline import synth_code_texts, synth_vocab
    from codesearch_amd.tokenizer import WordPieceTokenizer

    vocab = synth_vocab(1024)
    cfg = BertConfig(vocab_size=1024, hidden=768, layers=2, heads=12, intermediate=3072, max_position=512, pooling=POOL_MEAN,
                     arch=arch)
    src = synth_params(cfg, 79)
    sd = to_state_dict(cfg, src)
    for l in range(cfg.layers):  # the checkpoint's up projection has no bias
        sd[f"encoder.layer.{l}.intermediate.dense.bias"][:] = 0
        sd[f"encoder.layer.{l}.intermediate.gate.bias"][:] = 0
    d = tmp_path / "snapshot"
    jina_snapshot(str(d), cfg, src, first_file)
    (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n")
    emb = FastEmbedder.from_dir(str(d))
    assert (emb.config.arch, emb.dimensions(), emb.config.pooling, emb.config.max_position) == (arch, 768, POOL_MEAN, 512)
    from safetensors.numpy import load_file
    flat = from_jina_state_dict(cfg, load_file(str(d / "model.safetensors")))
    assert np.array_equal(flat, src)  # (src's own bias slots were zeroed in place through sd's views)
    texts = synth_code_texts(vocab, 9, 3, mean_words=20) + ["fn main() { [SEP] }", ""]
    got = np.stack(emb.embed_batch(texts))
    ref_emb = FE(cfg, params=flat, tokenizer=WordPieceTokenizer(vocab, max_length=512))
    assert np.array_equal(got, np.stack(ref_emb.embed_batch(texts)))
    ids, mask = emb.tokenizer.encode_batch(texts)
    np.testing.assert_allclose(got, oracle.bert_forward(cfg, flat, ids, mask)["pooled"], atol=TOL_ORACLE)
    emb.close()
    ref_emb.close()


def test_two_stream_slices_and_the_longest_sequences(FE, oracle):
    """From 20,000 tokens a forward runs as two half-batches on two streams; and 512 positions — distances up to 511 under
    the steepest and the flattest slope."""
    cfg = BertConfig(vocab_size=512, hidden=768, layers=1, heads=12, intermediate=3072, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_JINA_QKNORM)
    emb = FE(cfg, seed=423)
    params = synth_params(cfg, 423)
    ids, mask = synth_token_batch(cfg, 700, 96, 256, True)  # 24,576 tokens
    got = emb.embed_ids(ids, mask, batch_size=96)
    np.testing.assert_allclose(got, oracle.bert_forward(cfg, params, ids, mask)["pooled"], atol=TOL_ORACLE)
    ids, mask = synth_token_batch(cfg, 701, 3, 512, True)
    got = emb.embed_ids(ids, mask)
    np.testing.assert_allclose(got, oracle.bert_forward(cfg, params, ids, mask)["pooled"], atol=TOL_ORACLE)
    assert emb.debug_counters()[1] == 0
    emb.close()


def test_randomised_shapes_against_the_oracle(FE, oracle):
    """Twelve seeded (batch, length, raggedness) draws over both head widths — odd lengths, single rows, lengths around the
    attention kernels' 32 / 64 / 128-key boundaries — one embedder per width, every draw against the oracle."""
    rng = np.random.default_rng(20261005)
    for hidden, heads, inter, arch in ((384, 12, 1536, ARCH_JINA), (768, 12, 3072, ARCH_JINA_QKNORM)):
        cfg = BertConfig(vocab_size=512, hidden=hidden, layers=2, heads=heads, intermediate=inter, max_position=512,
                         pooling=POOL_MEAN, arch=arch)
        emb = FE(cfg, seed=424)
        params = synth_params(cfg, 424)
        for draw in range(6):
            L = int(rng.choice([1, 2, 5, 31, 33, 63, 65, 100, 127, 129, 200, 257]))
            B = int(rng.integers(1, 40 if L <= 129 else 12))
            ragged = bool(rng.integers(0, 2)) and L >= 16
            ids, mask = synth_token_batch(cfg, 800 + 13 * draw + hidden, B, L, ragged)
            got = emb.embed_ids(ids, mask)
            ref = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
            np.testing.assert_allclose(got, ref, atol=TOL_ORACLE, err_msg=f"hidden {hidden} B {B} L {L} ragged {ragged}")
        assert emb.debug_counters()[1] == 0
        emb.close()


def test_texts_through_the_models_own_kind_of_tokenizer(FE, oracle, tmp_path):
    """jina-embeddings-v2-base-code ships a byte-level BPE tokenizer.json (not WordPiece): FastEmbedder.from_dir on a snapshot
    with such a file (trained here by the `tokenizers` library itself) tokenises with csrc/bpe.cpp — the ids the library
    gives — and embeds like the oracle on those ids: strings in, vectors out, for the registry's code model."""
    pytest.importorskip("tokenizers")
    from codesearch_amd import FastEmbedder
    from tests.test_bpe_tokenizer import build as build_bpe

    cfg = BertConfig(vocab_size=704, hidden=768, layers=2, heads=12, intermediate=3072, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_JINA_QKNORM)
    flat = synth_params(cfg, 83)
    for l in range(cfg.layers):
        sd = to_state_dict(cfg, flat)
        sd[f"encoder.layer.{l}.intermediate.dense.bias"][:] = 0
        sd[f"encoder.layer.{l}.intermediate.gate.bias"][:] = 0
    d = tmp_path / "snapshot"
    jina_snapshot(str(d), cfg, flat)
    tok = build_bpe(str(d / "tokenizer.json"), post="roberta")
    assert tok.get_vocab_size() <= cfg.vocab_size
    emb = FastEmbedder.from_dir(str(d))
    texts = ["def authenticate(user, password):\n    return check_hash(user.password_hash, password)",
             "fn main() { println!(\"{}\", 42); }", "SELECT id FROM users WHERE age >= 18;", "it's a naïve café", "", "x"]
    got = np.stack(emb.embed_batch(texts))
    encs = [tok.encode(t).ids for t in texts]
    L = max(len(e) for e in encs)
    ids = np.full((len(texts), L), tok.token_to_id("<pad>"), np.int32)
    mask = np.zeros((len(texts), L), np.int32)
    for i, e in enumerate(encs):
        ids[i, :len(e)] = e
        mask[i, :len(e)] = 1
    my_ids, my_mask = emb.tokenizer.encode_batch(texts)
    assert np.array_equal(my_ids, ids) and np.array_equal(my_mask, mask)
    np.testing.assert_allclose(got, oracle.bert_forward(cfg, flat, ids, mask)["pooled"], atol=TOL_ORACLE)
    emb.close()


def _fixture_module(name):
    import importlib.util

    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(__file__), "golden", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_embedder_from_a_fastembed_cache_of_the_onnx_export(FE, tmp_path):
    """What fastembed leaves on disk for the registry entry: config.json + onnx/model.onnx (no safetensors).  The file is
    written here by torch.onnx's exporter from the JinaBert-shaped module of tests/golden/make_jina_onnx_fixture.py at a width
    the kernels run (hidden 384, 12 heads, 2 layers); the embedder loaded from it — the variant (query / key LayerNorm) read
    off the file's own tensors, against a config.json that names the other modelling file — must reproduce the EXPORTING
    MODULE's own pooled output."""
    from codesearch_amd import FastEmbedder

    d = tmp_path / "cache"
    (d / "onnx").mkdir(parents=True)
    _fixture_module("make_jina_onnx_fixture").write(str(d / "onnx"), "model", dims=(384, 12, 2, 256, 64))
    state = np.load(str(d / "onnx" / "model_state.npz"))
    repo = "jinaai/jina-bert-implementation"   # (the config names the FIRST modelling file: the file's tensors decide)
    hf = {"model_type": "bert", "position_embedding_type": "alibi", "feed_forward_type": "geglu", "hidden_act": "gelu", "vocab_size": 64,
          "hidden_size": 384, "num_attention_heads": 12, "num_hidden_layers": 2, "intermediate_size": 256, "max_position_embeddings": 8192,
          "type_vocab_size": 2, "layer_norm_eps": 1e-12, "emb_pooler": "mean",
          "auto_map": {"AutoConfig": repo + "--configuration_bert.JinaBertConfig", "AutoModel": repo + "--modeling_bert.JinaBertModel"}}
    (d / "config.json").write_text(json.dumps(hf))
    (d / "vocab.txt").write_text("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(59)]) + "\n")
    emb = FastEmbedder.from_dir(str(d))
    assert (emb.config.arch, emb.dimensions(), emb.config.intermediate, emb.config.pooling) == (ARCH_JINA_QKNORM, 384, 256, POOL_MEAN)
    got = emb.embed_ids(state["query_ids"], state["query_mask"])
    np.testing.assert_allclose(got, state["query_pooled"], atol=TOL_GOLDEN)
    emb.close()
