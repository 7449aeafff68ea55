"""The hand-written readers of untrusted model files — codesearch_amd/csrc/onnx_reader.cpp (protobuf wire format),
checkpoint.cpp (config.json, safetensors header + payload, tokenizer.json) and tokenizer.cpp (vocab.txt) — under
AddressSanitizer + UBSan on the CPU build (tests/cpp/parser_fuzz.cpp, `make -C tests/cpp asan`): every prefix
truncation and 1,000 seeded mutations of each kind of file must come back as CS_OK or as a CS_ERR_* with a message,
never as a crash, a sanitizer report or an unbounded allocation.  These are what FastEmbedder::with_cache_dir
(/root/reference/src/embed/embedder.rs:218-245) points the library at: files downloaded from a model hub."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
GOLD = os.path.join(ROOT, "tests", "golden")
TINY = "48 64 2 2 128 16"  # vocab hidden layers heads intermediate max_position of tests/golden/bert_tiny_export.onnx


@pytest.fixture(scope="module")
def fuzz():
    import fcntl

    with open(os.path.join(CPP, ".asan_build.lock"), "w") as lock:  # pytest-xdist workers build one at a time
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["make", "-s", "-C", CPP, "asan"], check=True)
    exe = os.path.join(CPP, "parser_fuzz_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=4096",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")

    def run(kind, path, seed=1, flips=1000, aux=TINY):
        r = subprocess.run([exe, kind, str(path), str(seed), str(flips), aux], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
        assert "intact file: accepted" in r.stdout, r.stdout
        return r.stdout

    return run


def test_onnx_reader_survives_truncations_and_mutations(fuzz):
    out = fuzz("onnx", os.path.join(GOLD, "bert_tiny_export.onnx"), seed=11)
    assert "0 crashes" in out


def test_quantised_onnx_survives_truncations_and_mutations(fuzz, tmp_path):
    """The dynamically quantised layout (INT8 / UINT8 weights + scales + zero points, the reference's default model)."""
    from codesearch_amd.bert_params import BertConfig, synth_params, to_state_dict
    from tests import onnx_writer

    cfg = BertConfig(vocab_size=64, hidden=128, heads=4, intermediate=256, layers=1, max_position=16)
    sd = to_state_dict(cfg, synth_params(cfg, 5))
    p = tmp_path / "model_quantized.onnx"
    p.write_bytes(onnx_writer.bert_onnx(sd, 1, "quantized", qdtype=onnx_writer.UINT8, per_channel=True, quantize_tables=True))
    fuzz("onnx", p, seed=21, flips=600, aux="64 128 1 4 256 16")
    # ... and the same through onnxruntime's transformer optimiser: QAttention, biases inside SkipLayerNormalization / BiasGelu
    p2 = tmp_path / "model_optimized.onnx"
    p2.write_bytes(onnx_writer.bert_onnx(sd, 1, "optimized_quantized", qdtype=onnx_writer.INT8, per_channel=False))
    fuzz("onnx", p2, seed=22, flips=400, aux="64 128 1 4 256 16")


def test_safetensors_reader_survives_truncations_and_mutations(fuzz, tmp_path):
    from safetensors.numpy import save_file

    st = np.load(os.path.join(GOLD, "bert_tiny_export_state.npz"))
    tensors = {k: np.ascontiguousarray(st[k]) for k in st.files if st[k].dtype == np.float32}
    p = tmp_path / "model.safetensors"
    save_file(tensors, str(p))
    fuzz("safetensors", p, seed=12)
    # the same payload as F16 and with a "bert." prefix: the other branches of the reader
    save_file({"bert." + k: v.astype(np.float16) for k, v in tensors.items()}, str(p))
    fuzz("safetensors", p, seed=13, flips=300)


def test_nomic_snapshot_readers_survive_truncations_and_mutations(fuzz, tmp_path):
    """The NomicBert branches of the same readers: config.json with the model repository's keys, model.safetensors with its
    names (rows of the fused Wqkv read at an offset, Linear biases optional)."""
    from codesearch_amd.bert_params import ARCH_NOMIC, POOL_MEAN, BertConfig, synth_params
    from tests.test_oracle_nomic import nomic_snapshot

    cfg = BertConfig(vocab_size=64, hidden=128, heads=4, intermediate=256, layers=1, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_NOMIC, rotary_base=1000.0)
    d = tmp_path / "nomic"
    nomic_snapshot(d, cfg, synth_params(cfg, 6))
    fuzz("safetensors", d / "model.safetensors", seed=31, flips=600, aux="64 128 1 4 256 512 1")
    nomic_snapshot(d, cfg, synth_params(cfg, 6), dtype=np.float16, with_biases=True)
    fuzz("safetensors", d / "model.safetensors", seed=32, flips=300, aux="64 128 1 4 256 512 1")
    out = fuzz("config_dir", d / "config.json", seed=33, flips=1500)
    assert "0 crashes" in out


def test_nomic_onnx_and_jina_snapshot_readers_survive_truncations_and_mutations(fuzz, tmp_path):
    """Round 5's reader branches: the structural NomicBert ONNX reader (a real exporter's file and the quantised layout of
    tests/onnx_writer.py) and the JinaBert safetensors names (rows of the [2I, H] up projection read at an offset) with its
    config.json keys."""
    from codesearch_amd.bert_params import ARCH_JINA_QKNORM, ARCH_NOMIC, POOL_MEAN, BertConfig, synth_params
    from tests import onnx_writer
    from tests.test_gpu_jina import jina_snapshot
    from tests.test_oracle_nomic import nomic_state_dict

    fuzz("onnx", os.path.join(GOLD, "nomic_tiny_export.onnx"), seed=41, aux="48 64 2 2 128 512 1")
    cfg = BertConfig(vocab_size=64, hidden=128, heads=4, intermediate=256, layers=1, max_position=512, pooling=POOL_MEAN,
                     arch=ARCH_NOMIC, rotary_base=1000.0)
    p = tmp_path / "model_quantized.onnx"
    p.write_bytes(onnx_writer.nomic_onnx(nomic_state_dict(cfg, synth_params(cfg, 7)), cfg.layers, quantized=True, per_channel=True))
    fuzz("onnx", p, seed=42, flips=600, aux="64 128 1 4 256 512 1")
    jcfg = BertConfig(vocab_size=64, hidden=128, heads=4, intermediate=256, layers=1, max_position=512, pooling=POOL_MEAN,
                      arch=ARCH_JINA_QKNORM)
    d = tmp_path / "jina"
    jina_snapshot(str(d), jcfg, synth_params(jcfg, 8))
    fuzz("safetensors", d / "model.safetensors", seed=43, flips=600, aux="64 128 1 4 256 512 3")
    out = fuzz("config_dir", d / "config.json", seed=44, flips=1500)
    assert "0 crashes" in out
    # ModernBERT: its own safetensors reader (whole tensors fetched, the feed-forward zero-padded) and config.json branch
    from codesearch_amd.bert_params import ARCH_MODERN
    from tests.test_gpu_modern import modern_snapshot

    mcfg = BertConfig(vocab_size=64, hidden=128, heads=4, intermediate=256, layers=2, max_position=512, type_vocab_size=1,
                      pooling=POOL_MEAN, arch=ARCH_MODERN, layer_norm_eps=1e-5, rotary_base=160000.0, rotary_base_local=10000.0,
                      global_every=3, local_window=64)
    md = tmp_path / "modern"
    modern_snapshot(str(md), mcfg, synth_params(mcfg, 9), file_intermediate=200)
    fuzz("safetensors", md / "model.safetensors", seed=45, flips=600, aux="64 128 2 4 256 512 4")
    out = fuzz("config_dir", md / "config.json", seed=46, flips=1500)
    assert "0 crashes" in out


def test_tokenizer_json_and_vocab_readers_survive(fuzz, tmp_path):
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    from codesearch_amd.pipeline import synth_vocab

    vocab = synth_vocab(600)
    tk = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
    tk.normalizer = normalizers.BertNormalizer(lowercase=True)
    tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", vocab["[CLS]"]), ("[SEP]", vocab["[SEP]"])])
    tk.enable_truncation(max_length=64)
    p = tmp_path / "tokenizer.json"
    tk.save(str(p))
    fuzz("tokenizer_json", p, seed=14)
    # the \\u-escaped form of the same document (surrogate pairs, escapes inside keys)
    esc = tmp_path / "escaped.json"
    esc.write_text(json.dumps(json.load(open(p, encoding="utf-8")), ensure_ascii=True))
    fuzz("tokenizer_json", esc, seed=15, flips=300)
    # a byte-level BPE tokenizer.json (bpe.cpp: merges, pre-tokenizer / post-processor objects, added tokens)
    from tests.test_bpe_tokenizer import build as build_bpe

    bp = tmp_path / "bpe.json"
    build_bpe(str(bp), post="template", digits=True)
    fuzz("tokenizer_json", bp, seed=16, flips=400)
    v = tmp_path / "vocab.txt"
    v.write_text("\n".join(t for t, _ in sorted(vocab.items(), key=lambda kv: kv[1])) + "\n", encoding="utf-8")
    fuzz("vocab", v, seed=16)


def test_unigram_tokenizer_json_reader_survives(fuzz, tmp_path):
    """A tokenizer.json with a SentencePiece-unigram model (csrc/unigram.cpp): the piece list, the base64 precompiled
    character map (a double-array trie walked with indices taken from the file) and the component lists, mutated."""
    pytest.importorskip("sentencepiece")
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_unigram_golden as G

    p = tmp_path / "tokenizer.json"
    G.build(str(p), "published")
    fuzz("tokenizer_json", p, seed=21, flips=200)
    # without the (large) character map the mutations land on the pieces, scores and component objects
    d = json.load(open(p, encoding="utf-8"))
    d["normalizer"]["normalizers"][0]["precompiled_charsmap"] = ""  # (an empty map normalises nothing)
    small = tmp_path / "small.json"
    small.write_text(json.dumps(d), encoding="utf-8")
    fuzz("tokenizer_json", small, seed=22, flips=300)


def test_config_json_reader_survives(fuzz, tmp_path):
    cfg = {"model_type": "bert", "hidden_act": "gelu", "vocab_size": 48, "hidden_size": 384, "num_hidden_layers": 2,
           "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 16, "type_vocab_size": 2,
           "layer_norm_eps": 1e-12, "position_embedding_type": "absolute", "architectures": ["BertModel"]}
    p = tmp_path / "config.json"
    p.write_text(json.dumps(cfg, indent=1))
    fuzz("config_dir", p, seed=17)


def test_jina_and_modernbert_onnx_exports_survive_truncations_and_mutations(fuzz):
    """The JinaBert and ModernBERT branches of the ONNX reader on files a real exporter wrote (tests/golden/make_jina_onnx_fixture.py,
    make_modern_onnx_fixture.py): named tensors, anonymous weight products found by shape and position, the zero-padded
    feed-forward of the ModernBERT block (the file is 80 wide, the configuration 128)."""
    out = fuzz("onnx", os.path.join(GOLD, "jina_tiny_export.onnx"), seed=51, flips=800, aux="48 64 2 2 128 512 3")
    assert "0 crashes" in out
    out = fuzz("onnx", os.path.join(GOLD, "modern_tiny_export.onnx"), seed=52, flips=800, aux="48 64 3 2 128 64 4")
    assert "0 crashes" in out
