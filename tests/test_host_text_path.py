"""Host text path and caller logic: the reference's own known answers re-expressed, and the
WordPiece tokenizer against golden vectors from the `tokenizers` 0.22.2 library.  CPU only."""
import json
import os

import numpy as np

from codesearch_amd.batch import (BatchEmbedder, EmbeddingService, EmbeddingStats, clean_docstring,
                                  prepare_text)
from codesearch_amd.search import merge_variant_results, retrieval_limit, should_use_vector_only
from codesearch_amd.tokenizer import BertWordPieceTokenizer
from codesearch_amd.vector_store import Chunk, SearchResult

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tokenizer_golden.json")))
GOLD_SPECIAL = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tokenizer_golden_special.json")))


def test_clean_docstring_reference_cases():
    """/root/reference/src/embed/batch.rs:253-274 test_clean_docstring."""
    assert clean_docstring("/// This is a doc comment\n/// with multiple lines") == \
        "This is a doc comment with multiple lines"
    assert clean_docstring('"""This is a Python docstring"""') == '""This is a Python docstring""'
    assert clean_docstring("/**\n * JSDoc comment\n * with multiple lines\n */") == \
        "JSDoc comment with multiple lines"
    assert clean_docstring('"This is a quoted docstring"') == "This is a quoted docstring"
    assert clean_docstring("") == "" and clean_docstring("//! inner\r\n// plain") == "inner plain"


def test_prepare_text_reference_case():
    """batch.rs:276-314 test_prepare_text (the reference needs a downloaded model for this)."""
    c = Chunk('fn test() { println!("test"); }', 0, 1, "Function", "test.rs",
              context=["File: test.rs", "Function: test"], signature="fn test()", docstring="/// Test function")
    text = prepare_text(c)
    assert "Context: File: test.rs > Function: test" in text
    assert "Signature: fn test()" in text
    assert "Documentation: Test function" in text
    assert "Code:" in text
    assert text == ("Context: File: test.rs > Function: test\nSignature: fn test()\nName: test\n"
                    'Documentation: Test function\nCode:\nfn test() { println!("test"); }')
    g = Chunk("x", 0, 0, "Function", "a", signature="fn sort<T: Ord>(items: Vec<T>) -> Vec<T>")
    assert "Name: sort" in prepare_text(g)
    assert prepare_text(Chunk("body", 0, 0, "Other", "a")) == "Code:\nbody"
    assert "Name:" not in prepare_text(Chunk("b", 0, 0, "Other", "a", signature="lonely"))


def test_embedding_stats_reference_case():
    """batch.rs:238-251."""
    s = EmbeddingStats(100, 80, 20, 0, 1000)
    assert s.cache_hit_rate() == 0.2 and s.success_rate() == 0.8 and s.chunks_per_second() == 80.0


def test_wordpiece_matches_tokenizers_library():
    """cs_tokenizer_* (csrc/tokenizer.cpp, host-only C++) against outputs of the `tokenizers` 0.22.2
    wheel: the plain BERT pipeline, then the pipeline as fastembed configures it (special tokens as
    added tokens), cased and uncased, two truncation lengths, seeded random Unicode text."""
    for case in GOLD["cases"]:
        tok = BertWordPieceTokenizer(GOLD["vocab"], max_length=case["max_length"])
        ids, mask = tok.encode_batch(case["texts"])
        assert ids.tolist() == case["ids"] and mask.tolist() == case["mask"]
        assert ids.shape[1] <= case["max_length"]
    tok = BertWordPieceTokenizer(GOLD["vocab"])
    assert tok.encode("")[0] == 101 and tok.encode("")[-1] == 102  # BERT's [CLS]/[SEP] ids kept
    assert tok.vocab_size() == len(GOLD["vocab"]) and tok.token_to_id("[MASK]") == 103
    assert tok.token_to_id("no such token") == -1
    for case in GOLD_SPECIAL["cases"]:
        tok = BertWordPieceTokenizer(GOLD_SPECIAL["vocab"], lowercase=case["lowercase"], max_length=case["max_length"])
        ids, mask = tok.encode_batch(case["texts"])
        assert ids.tolist() == case["ids"] and mask.tolist() == case["mask"], case["texts"]
        # max_length passed per call overrides the handle's
        tok512 = BertWordPieceTokenizer(GOLD_SPECIAL["vocab"], lowercase=case["lowercase"])
        ids2, _ = tok512.encode_batch(case["texts"], case["max_length"])
        assert ids2.tolist() == case["ids"]


def test_tokenizer_abi_edges(tmp_path):
    from codesearch_amd._lib import CsError

    tok = BertWordPieceTokenizer(GOLD["vocab"])
    ids, mask = tok.encode_batch([])
    assert ids.shape == (0, 0)
    ids, mask = tok.encode_batch(["", "a"])
    assert ids.tolist() == [[101, 102, 0], [101, tok.token_to_id("a"), 102]] and mask.tolist() == [[1, 1, 0], [1, 1, 1]]
    # ill-formed UTF-8 reaches the tokenizer as U+FFFD, which clean_text drops
    import ctypes as C

    import numpy as np

    from codesearch_amd import _lib
    raw = b"a\xff\xe2\x82b \xf0\x9f"
    off = np.array([0, len(raw)], np.uint64)
    L = C.c_uint32()
    out = np.zeros((1, 8), np.int32)
    _lib.check(tok._lib.cs_tokenizer_encode_batch(tok.handle, raw, off.ctypes.data_as(_lib.u64p), 1, 0,
                                                  out.ctypes.data_as(_lib.i32p), None, 8, C.byref(L)))
    assert out[0, :L.value].tolist() == tok.encode("ab")
    # vocab.txt from a file, CRLF tolerated
    toks = sorted(GOLD["vocab"], key=GOLD["vocab"].get)
    path = tmp_path / "vocab.txt"
    path.write_bytes("\r\n".join(toks).encode("utf-8"))
    tf = BertWordPieceTokenizer.from_vocab_file(str(path))
    text = GOLD["cases"][0]["texts"][0]
    assert tf.encode(text) == tok.encode(text) and tf.vocab_size() == len(toks)
    for bad in ({"a": 0, "b": 1}, ):
        try:
            BertWordPieceTokenizer(bad)
            raise AssertionError("vocabulary without special tokens accepted")
        except CsError as e:
            assert "[PAD] [UNK] [CLS] [SEP]" in str(e)
    try:
        BertWordPieceTokenizer(GOLD["vocab"], max_length=1)
        raise AssertionError("max_length 1 accepted")
    except CsError:
        pass


def test_tokenizer_live_against_wheel_when_present():
    """Where the `tokenizers` wheel is importable (this image), a short seeded fuzz against it."""
    import random

    import pytest
    tk = pytest.importorskip("tokenizers")
    from tokenizers import AddedToken, Tokenizer
    from tokenizers.models import WordPiece
    from tokenizers.normalizers import BertNormalizer
    from tokenizers.pre_tokenizers import BertPreTokenizer
    from tokenizers.processors import BertProcessing

    vocab = GOLD_SPECIAL["vocab"]
    ref = Tokenizer(WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
    ref.normalizer = BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=True)
    ref.pre_tokenizer = BertPreTokenizer()
    ref.post_processor = BertProcessing(("[SEP]", vocab["[SEP]"]), ("[CLS]", vocab["[CLS]"]))
    ref.add_special_tokens([AddedToken(t, special=True, normalized=False)
                            for t in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")])
    ref.enable_truncation(64)
    ref.enable_padding(pad_id=0, pad_token="[PAD]")
    mine = BertWordPieceTokenizer(vocab, max_length=64)
    rng = random.Random(7)
    words = list(vocab)
    for _ in range(60):
        texts = []
        for _ in range(rng.randint(1, 12)):
            parts = []
            for _ in range(rng.randint(0, 40)):
                if rng.random() < 0.5:
                    parts.append(rng.choice(words))
                else:
                    parts.append("".join(chr(rng.choice((rng.randint(0x20, 0x24F), rng.randint(0x250, 0x2FFF),
                                                         rng.randint(0x3000, 0xD7FF), rng.randint(0xE000, 0x2FFFF))))
                                         for _ in range(rng.randint(1, 5))))
                parts.append(rng.choice(["", " ", "\t", "-"]))
            texts.append("".join(parts))
        enc = ref.encode_batch(texts)
        ids, mask = mine.encode_batch(texts)
        assert ids.tolist() == [e.ids for e in enc], texts
        assert mask.tolist() == [e.attention_mask for e in enc]


class _FakeEmbedder:
    """Counts encoder calls; the embedding is a pure function of the text."""

    def __init__(self):
        self.calls = []

    def embed_batch(self, texts):
        self.calls.append(list(texts))
        return [np.array([len(t), sum(map(ord, t)) % 997], np.float32) for t in texts]

    def embed_one(self, text):
        return self.embed_batch([text])[0]

    def dimensions(self):
        return 2


def test_batch_embedder_slices_and_service_cache_semantics():
    fe = _FakeEmbedder()
    chunks = [Chunk(f"content {i}", i, i, "Function", "f.rs") for i in range(70)]
    out = BatchEmbedder(fe).embed_chunks(chunks)  # batch.rs:94: slices of 32
    assert [len(c) for c in fe.calls] == [32, 32, 6] and len(out) == 70
    assert fe.calls[0][0] == "Code:\ncontent 0"
    fe = _FakeEmbedder()
    svc = EmbeddingService(fe, batch_size=256)
    first = svc.embed_chunks(chunks[:40])
    assert len(fe.calls) == 1 and svc.cache_misses == 40
    mixed = chunks[30:50] + chunks[:5]  # partial hit: order must be the caller's
    second = svc.embed_chunks(mixed)
    assert len(fe.calls) == 2 and len(fe.calls[1]) == 10  # only the 10 unseen hashes were encoded
    assert [ec.chunk.content for ec in second] == [c.content for c in mixed]
    for ec in second:
        assert np.array_equal(ec.embedding, fe.embed_one(prepare_text(ec.chunk)))
    n = len(fe.calls)
    qs = svc.embed_queries_batch(["alpha", "beta", "alpha2"])
    qs2 = svc.embed_queries_batch(["beta", "gamma", "alpha"])  # 2 cached, 1 new, order kept
    assert len(fe.calls) == n + 2 and fe.calls[-1] == ["gamma"]
    assert np.array_equal(qs2[0], qs[1]) and np.array_equal(qs2[2], qs[0])
    assert np.array_equal(svc.embed_query("gamma"), qs2[1]) and len(fe.calls) == n + 2


def _res(i, score):
    return SearchResult(i, "", "p", 0, 0, "Function", None, None, None, "", distance=1.0 - score, score=score)


def test_variant_merge_and_early_termination():
    """search/mod.rs:513-611."""
    a = [_res(1, 0.9), _res(2, 0.8), _res(3, 0.7)]
    b = [_res(2, 0.95), _res(4, 0.6), _res(1, 0.85)]
    m = merge_variant_results([a, b], 3)
    assert [(r.id, r.score) for r in m] == [(2, 0.95), (1, 0.9), (3, 0.7)]
    assert abs(m[0].distance - 0.05) < 1e-9  # distance travels with the winning score
    assert retrieval_limit(25, True, False) == 25
    assert retrieval_limit(25, False, True) == 100 and retrieval_limit(50, False, True) == 150
    assert retrieval_limit(25, False, False) == 200 and retrieval_limit(60, False, False) == 300
    hi = [_res(i, 0.9) for i in range(6)]
    assert should_use_vector_only(hi, False) and not should_use_vector_only(hi, True)
    assert not should_use_vector_only(hi[:4] + [_res(9, 0.8)], False)  # distance 0.2 in the top 5
    assert not should_use_vector_only([], False)
