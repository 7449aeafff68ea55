"""Parity tests for the NomicBert encoder family (cs_bert_config.arch = CS_ARCH_NOMIC: the reference registry's three
nomic-embed-text entries, /root/reference/src/embed/embedder.rs:30-35): HIP kernels through the C ABI — the BERT dense
layers and attention plus the rotary map and the feed-forward gate of csrc/nomic.hip — against the CPU oracle and the
committed float64 golden vectors (tests/golden/make_nomic_golden.py).  Needs an MI355X.

Bar as for the BERT encoders: within 2e-5 of the fp32 oracle, 3e-5 of the float64 golden (north_star: 1e-4)."""
import os

import numpy as np
import pytest

from codesearch_amd.bert_params import ARCH_NOMIC, POOL_CLS, POOL_MEAN, BertConfig, synth_params, synth_token_batch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "nomic_golden.npz"))
TOL_ORACLE = 2e-5
TOL_GOLDEN = 3e-5


def case_cfg(name):
    m = GOLD[name + "/meta"]
    cfg = BertConfig(vocab_size=int(m[0]), hidden=int(m[1]), layers=int(m[2]), heads=int(m[3]), intermediate=int(m[4]),
                     max_position=int(m[5]), pooling=POOL_MEAN, arch=ARCH_NOMIC, rotary_base=1000.0)
    return cfg, int(m[6]), int(m[7]), int(m[8]), int(m[9]), bool(m[10])


@pytest.fixture(scope="module")
def FE(gpu_lib):
    from codesearch_amd import FastEmbedder, ModelType

    assert gpu_lib.cs_device_count() >= 1
    return lambda cfg, **kw: FastEmbedder(ModelType.NomicEmbedTextV15, config=cfg, **kw)


@pytest.mark.parametrize("gemm_mode", ["split", "f32"])
@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"] if str(n) != "nomic_shape"])
def test_small_cases_vs_golden_and_oracle(FE, oracle, name, gemm_mode):
    """Both arithmetic modes: split-f16 operands on the f16 MFMA (the default) and the exact-f32 kernels (the fallback the
    split path takes when a value leaves the f16 range)."""
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
    """nomic-embed-text-v1.5's own shape (12 x 768, 12 heads of 64, n_inner 3072, vocab 30528, rotary base 1000)."""
    cfg, wseed, iseed, B, L, ragged = case_cfg("nomic_shape")
    ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
    emb = FE(cfg, seed=wseed)
    got = emb.embed_ids(ids, mask)
    ref = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, wseed), ids, mask)
    np.testing.assert_allclose(got, ref["pooled"], atol=TOL_ORACLE)
    np.testing.assert_allclose(got, GOLD["nomic_shape/mean"], atol=TOL_GOLDEN)
    assert emb.debug_counters()[:2] == (1, 0)
    emb.close()


def test_registry_entry_builds_and_runs(FE, oracle):
    """ModelType::NomicEmbedTextV1 / V15 / V15Q -> the NomicBert config (embedder.rs:64-66, :84-86: 768 dimensions); two
    layers of it here, every dense-layer route by batch size: a few rows (skinny kernels), the reference's 32-chunk call
    (mid-size tiles), an indexing batch (persistent wide kernels, N = 2I = 6,144 for the gated up projection)."""
    from codesearch_amd import ModelType

    for m in (ModelType.NomicEmbedTextV1, ModelType.NomicEmbedTextV15, ModelType.NomicEmbedTextV15Q):
        assert (m.dimensions(), m.bert_config().arch) == (768, ARCH_NOMIC)
    cfg = ModelType.NomicEmbedTextV15.bert_config()
    cfg.layers, cfg.vocab_size = 2, 2048
    emb = FE(cfg, seed=411)
    params = synth_params(cfg, 411)
    for B, L, ragged in ((1, 12, False), (32, 96, True), (128, 128, True)):
        ids, mask = synth_token_batch(cfg, 500 + B, B, L, ragged)
        got = emb.embed_ids(ids, mask)
        ref = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
        np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    assert emb.debug_counters()[1] == 0  # no exact-f32 fallback
    emb.close()


def test_minibatches_and_padding_do_not_change_an_embedding(FE, oracle):
    """A row's rotary angles are its own positions: embedding it alone, in a longer padded batch, or behind other rows
    gives the same vector (the length-grouped runner relies on it)."""
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
    """CLS pooling on this family takes the full last layer (the CLS tail is BERT's); a quantised Nomic model is refused
    with a worded error, and so are a missing rotary base and an unknown family."""
    from codesearch_amd import CsError
    from codesearch_amd.bert_params import quantize_linear_weights

    cfg, wseed, iseed, B, L, _ = case_cfg("dh32_L64")
    cfg.pooling = POOL_CLS
    ids, mask = synth_token_batch(cfg, iseed, 80, L, False)  # 5,120 token rows: where BERT would take the CLS tail
    emb = FE(cfg, seed=wseed)
    got = emb.embed_ids(ids, mask)
    ref = oracle.bert_forward(cfg, synth_params(cfg, wseed), ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    emb.last_hidden(80 * L)  # the whole last layer was computed
    emb.close()
    cfg.pooling = POOL_MEAN
    bad = BertConfig(**{**cfg.__dict__, "rotary_base": 0.0})
    with pytest.raises(CsError, match="rotary base"):
        FE(bad, seed=1)
    with pytest.raises(CsError, match="unknown encoder family"):
        FE(BertConfig(**{**cfg.__dict__, "arch": 7}), seed=1)
    small = BertConfig(vocab_size=512, hidden=384, layers=1, heads=12, intermediate=1536, pooling=POOL_MEAN)
    p, wscale = quantize_linear_weights(small, synth_params(small, 3))
    with pytest.raises(CsError, match="dynamic-quantisation mode is not built"):
        nomic_small = BertConfig(**{**small.__dict__, "arch": ARCH_NOMIC, "rotary_base": 1000.0})
        FE(nomic_small, params=synth_params(nomic_small, 3), wscale=wscale)


def test_embedder_from_a_nomic_snapshot_directory(FE, oracle, tmp_path):
    """FastEmbedder.from_dir = cs_embedder_create_from_dir + the directory's WordPiece vocabulary (the Nomic models use
    bert-base-uncased's): a NomicBert snapshot written here — config.json with the repository's keys, model.safetensors
    with its tensor names (fused Wqkv, no Linear biases), vocab.txt — embeds texts like an embedder handed the mapped
    flat block, and like the oracle."""
    from codesearch_amd import FastEmbedder
    from codesearch_amd.bert_params import from_nomic_state_dict
    from codesearch_amd.pipeline import synth_code_texts, synth_vocab
    from codesearch_amd.tokenizer import WordPieceTokenizer
    from tests.test_oracle_nomic import nomic_snapshot

    vocab = synth_vocab(1024)
    cfg = BertConfig(vocab_size=1024, hidden=768, layers=2, heads=12, intermediate=3072, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_NOMIC, rotary_base=1000.0)
    d = tmp_path / "snapshot"
    nomic_snapshot(d, cfg, synth_params(cfg, 78))
    (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n")
    emb = FastEmbedder.from_dir(str(d))
    assert (emb.config.arch, emb.config.rotary_base, emb.dimensions(), emb.config.pooling) == (ARCH_NOMIC, 1000.0, 768, POOL_MEAN)
    from safetensors.numpy import load_file
    flat = from_nomic_state_dict(cfg, load_file(str(d / "model.safetensors")))  # the file's Linear biases: none -> zero
    texts = synth_code_texts(vocab, 9, 3, mean_words=20) + ["fn main() { [SEP] }", ""]
    got = np.stack(emb.embed_batch(texts))
    ref_emb = FE(cfg, params=flat, tokenizer=WordPieceTokenizer(vocab, max_length=512))
    assert np.array_equal(got, np.stack(ref_emb.embed_batch(texts)))
    ids, mask = emb.tokenizer.encode_batch(texts)
    np.testing.assert_allclose(got, oracle.bert_forward(cfg, flat, ids, mask)["pooled"], atol=TOL_ORACLE)
    emb.close()
    ref_emb.close()


@pytest.mark.parametrize("hidden,heads,inter,B,L", [(384, 12, 1536, 96, 96), (768, 12, 3072, 48, 128), (384, 12, 1536, 40, 200)])
def test_gate_in_the_product_epilogue_equals_the_separate_kernel(FE, oracle, hidden, heads, inter, B, L):
    """At indexing batch sizes the gate is the up projection's epilogue (gemm_wide.hip GW_OUT_SWIGLU: the 128 x 384 block
    shape for 2I <= 4,096 columns, the 128 x 192 one above; a ragged last row tile in the third case).  Against the
    oracle, and against the same library with the stand-alone gate kernel (CS_NOMIC_GATE_FUSED=0 is read once per process,
    so the comparison runs in a child process through the C ABI as well)."""
    import subprocess
    import sys

    cfg = BertConfig(vocab_size=512, hidden=hidden, layers=2, heads=heads, intermediate=inter, max_position=512,
                     pooling=POOL_MEAN, arch=ARCH_NOMIC, rotary_base=1000.0)
    ids, mask = synth_token_batch(cfg, 600 + B, B, L, True)
    emb = FE(cfg, seed=412)
    got = emb.embed_ids(ids, mask)
    emb.close()
    ref = oracle.bert_forward(cfg, synth_params(cfg, 412), ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    code = (
        "import numpy as np, sys\n"
        "from codesearch_amd import FastEmbedder, ModelType\n"
        "from codesearch_amd.bert_params import ARCH_NOMIC, POOL_MEAN, BertConfig, synth_token_batch\n"
        f"cfg = BertConfig(vocab_size=512, hidden={hidden}, layers=2, heads={heads}, intermediate={inter}, max_position=512,"
        " pooling=POOL_MEAN, arch=ARCH_NOMIC, rotary_base=1000.0)\n"
        f"ids, mask = synth_token_batch(cfg, {600 + B}, {B}, {L}, True)\n"
        "emb = FastEmbedder(ModelType.NomicEmbedTextV15, config=cfg, seed=412)\n"
        "np.save(sys.argv[1], emb.embed_ids(ids, mask))\n")
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"nomic_unfused_{hidden}_{B}_{L}.npy")
    from codesearch_amd import _lib

    # (a laboratory knob: read by the diagnostic library only)
    env = dict(os.environ, CS_NOMIC_GATE_FUSED="0", CS_LIBCSGPU=_lib.DIAG_LIB_PATH)
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env, cwd=os.path.dirname(os.path.dirname(__file__)))
    unfused = np.load(out)
    os.remove(out)
    # the two routes differ by the silu's last bits (hardware exp2 / rcp against expf and a division)
    np.testing.assert_allclose(got, unfused, atol=2e-6)
    np.testing.assert_allclose(unfused, ref, atol=TOL_ORACLE)


def test_two_stream_slices_and_the_longest_sequences(FE, oracle):
    """From 20,000 tokens a forward runs as two half-batches on two streams (embedder.hip forward()): each slice has its
    own [T, 2I | I] stretch of the feed-forward workspace.  And 512 positions — every row of the rotary table."""
    cfg = BertConfig(vocab_size=512, hidden=768, layers=1, heads=12, intermediate=3072, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_NOMIC, rotary_base=1000.0)
    emb = FE(cfg, seed=413)
    params = synth_params(cfg, 413)
    ids, mask = synth_token_batch(cfg, 700, 96, 256, True)  # 24,576 tokens
    got = emb.embed_ids(ids, mask, batch_size=96)
    np.testing.assert_allclose(got, oracle.bert_forward(cfg, params, ids, mask)["pooled"], atol=TOL_ORACLE)
    ids, mask = synth_token_batch(cfg, 701, 3, 512, True)
    got = emb.embed_ids(ids, mask)
    np.testing.assert_allclose(got, oracle.bert_forward(cfg, params, ids, mask)["pooled"], atol=TOL_ORACLE)
    from codesearch_amd import CsError
    with pytest.raises(CsError):  # beyond max_position: no rotary row (and no truncation behind the tokenizer's back)
        emb.embed_ids(np.ones((1, 513), np.int32), np.ones((1, 513), np.int32))
    assert emb.debug_counters()[1] == 0
    emb.close()


def test_randomised_shapes_against_the_oracle(FE, oracle):
    """Twelve seeded (batch, length, raggedness) draws over both head widths — odd lengths, single rows, lengths around the
    attention kernels' 32 / 64 / 128-key boundaries — one embedder per width, every draw against the oracle."""
    rng = np.random.default_rng(20261004)
    for hidden, heads, inter in ((384, 12, 1536), (768, 12, 3072)):
        cfg = BertConfig(vocab_size=512, hidden=hidden, layers=2, heads=heads, intermediate=inter, max_position=512,
                         pooling=POOL_MEAN, arch=ARCH_NOMIC, rotary_base=1000.0)
        emb = FE(cfg, seed=414)
        params = synth_params(cfg, 414)
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


@pytest.mark.parametrize("quantized", [False, True])
def test_embedder_from_a_fastembed_cache_of_a_nomic_export(FE, oracle, tmp_path, quantized):
    """What `FastEmbedder::with_cache_dir` (embedder.rs:218-245) leaves on disk for a Nomic entry is the model's ONNX export
    (onnx/model.onnx; onnx/model_quantized.onnx for NomicEmbedTextV15Q, embedder.rs:36-37), not a safetensors snapshot:
    cs_embedder_create_from_dir reads the bias-free Linear weights off the graph's structure (csrc/onnx_reader.cpp).  The
    embedder from such a directory embeds like the one handed the same block, and like the oracle; the quantised file runs
    the f32 graph of its dequantised weights (the oracle is given those)."""
    import json

    from codesearch_amd import FastEmbedder
    from codesearch_amd.bert_params import from_nomic_state_dict
    from tests import onnx_writer
    from tests.test_oracle_nomic import nomic_snapshot, nomic_state_dict

    cfg = BertConfig(vocab_size=1024, hidden=768, layers=2, heads=12, intermediate=3072, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_NOMIC, rotary_base=1000.0)
    flat = synth_params(cfg, 81)
    cache = tmp_path / "models--nomic-ai--nomic-embed-text-v1.5" / "snapshots" / "abc"
    cache.mkdir(parents=True)
    nomic_snapshot(cache, cfg, flat)                     # config.json with the repository's keys ...
    (cache / "model.safetensors").unlink()               # ... but no PyTorch weights: fastembed fetched the ONNX file only
    (cache / "onnx").mkdir()
    sd = nomic_state_dict(cfg, flat)
    deq = {}
    rel = "onnx/model_quantized.onnx" if quantized else "onnx/model.onnx"
    (cache / rel).write_bytes(onnx_writer.nomic_onnx(sd, cfg.layers, quantized=quantized, per_channel=quantized, dequantized=deq))
    want_sd = dict(sd)
    want_sd.update(deq)
    block = from_nomic_state_dict(cfg, want_sd)
    ids, mask = synth_token_batch(cfg, 910, 6, 40, True)
    emb = FastEmbedder.from_dir(str(cache))
    assert (emb.config.arch, emb.dimensions(), emb.gemm_mode()) == (ARCH_NOMIC, 768, "split")
    got = emb.embed_ids(ids, mask)
    ref_emb = FE(cfg, params=block)
    assert np.array_equal(got, ref_emb.embed_ids(ids, mask))
    np.testing.assert_allclose(got, oracle.bert_forward(cfg, block, ids, mask)["pooled"], atol=TOL_ORACLE)
    emb.close()
    ref_emb.close()
