"""What fastembed's cache holds (FastEmbedder::with_cache_dir,
/root/reference/src/embed/embedder.rs:218-245): an ONNX export of the model and a tokenizer.json.
cs_bert_params_from_onnx / cs_tokenizer_create_from_json / cs_tokenizer_create_from_dir are host-only,
so these run without a GPU."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from codesearch_amd import _lib
from codesearch_amd.bert_params import BertConfig, POOL_MEAN, config_from_hf, from_state_dict, synth_params, to_state_dict
from tests import onnx_writer

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_onnx(gpu_lib, path, cfg):
    c = cfg.to_c()
    n = int(gpu_lib.cs_bert_param_count(C.byref(c)))
    out = np.empty(n, np.float32)
    _lib.check(gpu_lib.cs_bert_params_from_onnx(str(path).encode(), C.byref(c), out.ctypes.data_as(_lib.f32p), n))
    return out


def load_onnx_q(gpu_lib, path, cfg):
    c = cfg.to_c()
    n = int(gpu_lib.cs_bert_param_count(C.byref(c)))
    cols = int(gpu_lib.cs_bert_quant_columns(C.byref(c)))
    out = np.empty(n, np.float32)
    wscale = np.zeros((cfg.layers, cols), np.float32)
    quantized = C.c_int32(-1)
    _lib.check(gpu_lib.cs_bert_params_from_onnx_q(str(path).encode(), C.byref(c), out.ctypes.data_as(_lib.f32p), n,
                                                  wscale.ctypes.data_as(_lib.f32p), wscale.size, C.byref(quantized)))
    return out, wscale, int(quantized.value)


def test_reads_a_file_written_by_torchs_own_exporter(gpu_lib):
    """tests/golden/bert_tiny_export.onnx was produced by torch.onnx.export from transformers.BertModel
    (make_onnx_fixture.py): named embeddings / LayerNorms / biases, transposed `onnx::MatMul_N` Linear
    weights, module prefix "bert.".  The flat block must equal the one built from the state dict it was
    exported from, bit for bit."""
    hf = {"model_type": "bert", "vocab_size": 48, "hidden_size": 64, "num_hidden_layers": 2, "num_attention_heads": 2,
          "intermediate_size": 128, "max_position_embeddings": 16, "type_vocab_size": 2, "layer_norm_eps": 1e-12,
          "hidden_act": "gelu"}
    cfg = config_from_hf(hf)
    sd = dict(np.load(os.path.join(GOLDEN, "bert_tiny_export_state.npz")))
    got = load_onnx(gpu_lib, os.path.join(GOLDEN, "bert_tiny_export.onnx"), cfg)
    assert np.array_equal(got, from_state_dict(cfg, sd))


@pytest.mark.parametrize("style,dtype,prefix", [("matmul", onnx_writer.FLOAT, ""), ("matmul", onnx_writer.FLOAT16, "bert."),
                                                ("matmul", onnx_writer.BFLOAT16, "0.auto_model."),
                                                ("gemm", onnx_writer.FLOAT, ""), ("fused", onnx_writer.FLOAT, ""),
                                                ("fused", onnx_writer.FLOAT16, ""), ("optimized", onnx_writer.FLOAT, ""),
                                                ("optimized", onnx_writer.FLOAT16, "bert.")])
def test_exporter_layouts_and_dtypes(gpu_lib, tmp_path, style, dtype, prefix):
    """The layouts exporters produce, at the BGE-small width (hidden 384, 2 layers): MatMul + Add with
    anonymous transposed weights (raw_data and packed float_data payloads, bias as either Add input), Gemm
    with transB, the fused Attention node of ORT-optimised files, and the fully optimised form (model_optimized.onnx:
    every bias swallowed by SkipLayerNormalization / BiasGelu, EmbedLayerNormalization); FLOAT / FLOAT16 / BFLOAT16."""
    import torch

    cfg = BertConfig(vocab_size=300, layers=2, max_position=64, pooling=POOL_MEAN)
    flat = synth_params(cfg, 77)
    sd = to_state_dict(cfg, flat)
    path = tmp_path / "model.onnx"
    path.write_bytes(onnx_writer.bert_onnx(sd, cfg.layers, style, dtype, prefix))
    got = load_onnx(gpu_lib, path, cfg)
    assert load_onnx_q(gpu_lib, path, cfg)[2] == 0   # not a quantised file
    if dtype == onnx_writer.FLOAT:
        exp = flat
    else:
        tdt = torch.float16 if dtype == onnx_writer.FLOAT16 else torch.bfloat16
        exp = torch.from_numpy(flat).to(tdt).to(torch.float32).numpy()
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("style", ["quantized", "optimized_quantized"])
@pytest.mark.parametrize("qdtype,per_channel,tables,prefix", [(onnx_writer.INT8, False, False, ""),
                                                               (onnx_writer.UINT8, False, True, ""),
                                                               (onnx_writer.INT8, True, True, "0.auto_model.")])
def test_dynamically_quantised_export(gpu_lib, tmp_path, qdtype, per_channel, tables, prefix, style):
    """The layout of the reference's DEFAULT model (ModelType::AllMiniLML6V2Q, embedder.rs:12-13: fastembed's
    model_quantized.onnx, written by onnxruntime's dynamic quantiser): W_quantized (INT8 symmetric / UINT8 asymmetric,
    per tensor or per output channel) + W_scale + W_zero_point behind DynamicQuantizeLinear -> MatMulInteger -> Cast ->
    Mul -> Add(bias), optionally a quantised word-embedding table — and the same through onnxruntime's transformer optimiser
    ("optimized_quantized", the BGE-small *Q entry's model_optimized.onnx: QAttention with a packed int8 [H, 3H] weight, the
    other biases inside SkipLayerNormalization / BiasGelu).  The loader must hand back exactly (q - zero_point) * scale
    for those tensors and every other parameter bit for bit."""
    cfg = BertConfig(vocab_size=300, layers=2, max_position=64, pooling=POOL_MEAN)
    flat = synth_params(cfg, 78)
    sd = to_state_dict(cfg, flat)
    deq = {}
    path = tmp_path / "model_quantized.onnx"
    path.write_bytes(onnx_writer.bert_onnx(sd, cfg.layers, style, prefix=prefix, qdtype=qdtype, per_channel=per_channel,
                                           quantize_tables=tables, dequantized=deq))
    got = to_state_dict(cfg, load_onnx(gpu_lib, path, cfg))
    assert len(deq) == 6 * cfg.layers + (1 if tables else 0)
    for name, want in sd.items():
        if name in deq:
            assert np.array_equal(got[name], deq[name]), name
            rel = np.abs(deq[name] - want).max() / np.abs(want).max()
            assert 0 < rel < (0.02 if not per_channel else 0.01), (name, rel)   # it IS quantised, and sanely so
        else:
            assert np.array_equal(got[name], want), name
    # ... and the column scales the dynamic-quantisation mode runs on (cs_bert_params_from_onnx_q): every Linear weight
    # is an integer multiple of its column's scale, the integers spanning at most 8 bits
    block, wscale, quantized = load_onnx_q(gpu_lib, path, cfg)
    assert quantized == 1 and np.array_equal(to_state_dict(cfg, block)[next(iter(deq))], got[next(iter(deq))])
    H, I = cfg.hidden, cfg.intermediate
    roles = ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
             "intermediate.dense", "output.dense")
    for l in range(cfg.layers):
        c0 = 0
        for role in roles:
            w = got[f"encoder.layer.{l}.{role}.weight"]
            sc = wscale[l, c0:c0 + w.shape[0]]
            c0 += w.shape[0]
            d = w / sc[:, None]
            assert np.abs(d - np.rint(d)).max() < 1e-3 and (np.rint(d).max(axis=1) - np.rint(d).min(axis=1)).max() <= 255
            if not (style == "optimized_quantized" and role.startswith("attention.self.")):   # (one packed tensor there)
                assert (np.unique(sc).size == 1) == (not per_channel)
        assert c0 == 5 * H + I
    # a quantised weight whose scale is missing is refused with a message, not read as garbage
    broken = onnx_writer.bert_onnx(sd, cfg.layers, "quantized", qdtype=qdtype).replace(b"onnx::MatMul_1001_scale", b"onnx::MatMul_1001_scalX")
    (tmp_path / "broken.onnx").write_bytes(broken)
    with pytest.raises(_lib.CsError) as e:
        load_onnx(gpu_lib, tmp_path / "broken.onnx", cfg)
    assert "MatMulInteger weight" in str(e.value) or "scale" in str(e.value)


def test_onnx_errors(gpu_lib, tmp_path):
    cfg = BertConfig(vocab_size=300, layers=2, max_position=64)
    sd = to_state_dict(cfg, synth_params(cfg, 78))

    def expect(path, c, code, text):
        with pytest.raises(_lib.CsError) as e:
            load_onnx(gpu_lib, path, c)
        assert e.value.code == code and text in str(e.value), str(e.value)

    expect(tmp_path / "absent.onnx", cfg, _lib.CS_ERR_BAD_ARG, "cannot open")
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff not protobuf")
    expect(bad, cfg, _lib.CS_ERR_BAD_ARG, "is not an ONNX file")
    good = tmp_path / "model.onnx"
    good.write_bytes(onnx_writer.bert_onnx(sd, cfg.layers))
    expect(good, BertConfig(vocab_size=300, layers=3, max_position=64), _lib.CS_ERR_BAD_ARG,
           "neither encoder.layer.2.attention.self.query.bias")
    expect(good, BertConfig(vocab_size=300, layers=2, max_position=64, intermediate=1024), _lib.CS_ERR_DIM_MISMATCH,
           "intermediate.dense.bias has 1536 elements, config.json implies 1024")
    trunc = tmp_path / "trunc.onnx"
    trunc.write_bytes(good.read_bytes()[:100_000])
    expect(trunc, cfg, _lib.CS_ERR_BAD_ARG, "is not an ONNX file")


def test_tokenizer_json_equals_vocab_route(gpu_lib, tmp_path):
    """tokenizer.json as the `tokenizers` library itself serialises a BERT WordPiece tokenizer (what fastembed
    loads) -> cs_tokenizer_create_from_json must tokenise exactly like the vocab.txt route and like the library."""
    pytest.importorskip("tokenizers")
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    from codesearch_amd.pipeline import synth_code_texts, synth_vocab
    from codesearch_amd.tokenizer import WordPieceTokenizer

    vocab = synth_vocab(2000)
    vocab["café"] = len(vocab)               # non-ASCII entries travel as \\u escapes or raw UTF-8
    vocab["\U0001f600"] = len(vocab)         # astral code point: a surrogate pair when escaped
    for lowercase in (True, False):
        tk = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
        tk.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None,
                                                   lowercase=lowercase)
        tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
        tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                          special_tokens=[("[CLS]", vocab["[CLS]"]), ("[SEP]", vocab["[SEP]"])])
        tk.add_special_tokens(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
        tk.enable_truncation(max_length=48)
        d = tmp_path / ("lc" if lowercase else "cased")
        d.mkdir()
        tk.save(str(d / "tokenizer.json"))
        escaped = json.dumps(json.load(open(d / "tokenizer.json", encoding="utf-8")), ensure_ascii=True)
        (d / "tokenizer_escaped.json").write_text(escaped)
        a = WordPieceTokenizer.from_tokenizer_json(str(d / "tokenizer.json"))
        b = WordPieceTokenizer.from_tokenizer_json(str(d / "tokenizer_escaped.json"))
        ref = WordPieceTokenizer(vocab, lowercase=lowercase, max_length=48)
        assert a.max_length == 48 and a.vocab_size() == len(vocab) == b.vocab_size()
        assert a.token_to_id("café") == vocab["café"] == b.token_to_id("café")
        assert b.token_to_id("\U0001f600") == vocab["\U0001f600"]
        texts = synth_code_texts(vocab, 40, 5, mean_words=30) + ["Café \U0001f600 fn main() { [SEP] }", "", "  \t"]
        ia, ma = a.encode_batch(texts)
        ib, mb = b.encode_batch(texts)
        ir, mr = ref.encode_batch(texts)
        assert np.array_equal(ia, ir) and np.array_equal(ma, mr) and np.array_equal(ib, ir)
        for t, row, m in zip(texts, ia, ma):
            assert row[: int(m.sum())].tolist() == tk.encode(t).ids
        # the directory route: tokenizer.json wins; truncation = min(requested or 512, model_max_length)
        (d / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": lowercase, "model_max_length": 32}))
        c = WordPieceTokenizer.from_dir(str(d))
        assert c.max_length == 32 and c.encode_batch(texts)[0].shape[1] <= 32
        os.remove(d / "tokenizer.json")
        os.remove(d / "tokenizer_escaped.json")
        (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n", encoding="utf-8")
        e = WordPieceTokenizer.from_dir(str(d), max_length=48)
        assert e.max_length == 32
        ref32 = WordPieceTokenizer(vocab, lowercase=lowercase, max_length=32)
        assert np.array_equal(e.encode_batch(texts)[0], ref32.encode_batch(texts)[0])


def test_tokenizer_json_refusals(gpu_lib, tmp_path):
    from codesearch_amd.pipeline import synth_vocab
    from codesearch_amd.tokenizer import WordPieceTokenizer

    vocab = synth_vocab(600)
    base = {"model": {"type": "WordPiece", "unk_token": "[UNK]", "continuing_subword_prefix": "##",
                      "max_input_chars_per_word": 100, "vocab": vocab},
            "normalizer": {"type": "BertNormalizer", "clean_text": True, "handle_chinese_chars": True,
                           "strip_accents": None, "lowercase": True}, "truncation": None}

    def expect(doc, code, text):
        p = tmp_path / "t.json"
        p.write_text(json.dumps(doc))
        with pytest.raises(_lib.CsError) as e:
            WordPieceTokenizer.from_tokenizer_json(str(p))
        assert e.value.code == code and text in str(e.value), str(e.value)

    p = tmp_path / "ok.json"
    p.write_text(json.dumps(base))
    assert WordPieceTokenizer.from_tokenizer_json(str(p)).max_length == 512  # fastembed's default truncation
    expect({**base, "model": {**base["model"], "type": "WordLevel"}}, _lib.CS_ERR_UNSUPPORTED, "WordPiece, Unigram and BPE are built")
    # (a BPE model is read by its own rules, tests/test_bpe_tokenizer.py: this file has no merges)
    expect({**base, "model": {**base["model"], "type": "BPE"}}, _lib.CS_ERR_BAD_ARG, "no BPE merges")
    # (a Unigram model is read by its own rules, tests/test_unigram_tokenizer.py: this vocabulary is not a [piece, score] list)
    expect({**base, "model": {**base["model"], "type": "Unigram"}}, _lib.CS_ERR_BAD_ARG, "no unigram vocabulary")
    expect({**base, "model": {**base["model"], "continuing_subword_prefix": "@@"}}, _lib.CS_ERR_UNSUPPORTED, "expected ##")
    expect({**base, "normalizer": {"type": "NFKC"}}, _lib.CS_ERR_UNSUPPORTED, "only BertNormalizer")
    expect({**base, "normalizer": {**base["normalizer"], "strip_accents": False}}, _lib.CS_ERR_UNSUPPORTED, "strip_accents")
    expect({"model": {"type": "WordPiece", "vocab": {}}}, _lib.CS_ERR_BAD_ARG, "no WordPiece vocabulary")
    gap = dict(vocab)
    gap["late"] = len(vocab) + 5                                        # holes in the id space are tolerated
    p.write_text(json.dumps({**base, "model": {**base["model"], "vocab": gap}}))
    t = WordPieceTokenizer.from_tokenizer_json(str(p))
    assert t.vocab_size() == len(vocab) + 6 and t.token_to_id("late") == len(vocab) + 5
