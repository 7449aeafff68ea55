"""The SentencePiece-unigram text pipeline (csrc/unigram.cpp behind cs_tokenizer_create_from_json) against the
`tokenizers` library — the version the reference pins (Cargo.lock: tokenizers 0.22.2) — on tokenizer.json files of the
XLM-R kind, which the registry's multilingual entries ship (/root/reference/src/embed/embedder.rs:58,70).  No such file is on
disk (no network), so tests/golden/make_unigram_golden.py trains a small unigram model with `sentencepiece` (normalisation
rule nmt_nfkc: the real precompiled character map) and wraps it as transformers' XLM-R converter wraps the published ones,
in the hub's layout and in today's.  CPU only."""
import ctypes as C
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)

pytest.importorskip("sentencepiece")
pytest.importorskip("tokenizers")

import make_unigram_golden as G  # noqa: E402
import unigram_fuzz as F  # noqa: E402

from codesearch_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("unigram")
    mb = G.train()
    out = {}
    for style in ("published", "converter"):
        path = str(d / f"unigram_{style}.json")
        out[style] = (path, G.build(path, style, mb))
    return out


@pytest.mark.parametrize("style", ["published", "converter"])
def test_fixture_texts_match_the_library_and_the_committed_ids(lib, files, style):
    """Accents and combining sequences, width variants and ligatures (the precompiled map), CJK, Hangul, Arabic, Indic,
    emoji sequences, controls, literal special tokens, a literal U+2581, long runs: the ids of `tokenizers` — live, and
    as committed in tests/golden/unigram_golden.json (a drift of either library would show here)."""
    path, tok = files[style]
    golden = json.load(open(os.path.join(HERE, "golden", "unigram_golden.json"), encoding="utf-8"))
    texts = golden["texts"]
    assert texts == G.TEXTS
    got = F.encode_all(lib, path, texts)
    live = [tok.encode(t).ids for t in texts]
    assert got == live
    assert got == golden[style]["ids"]
    assert all(g[0] == 0 and g[-1] == 2 for g in got)           # <s> ... </s>
    assert got[0] == [0, 2]                                     # the empty text


def test_random_strings_match_the_library(lib):
    bad, total = F.mismatches(seed=3, n=1500)
    assert bad == 0, (bad, total)


def test_a_vocabulary_of_xlmr_size(lib, files, tmp_path):
    """XLM-R's vocabulary has 250,002 pieces: the same file with 120,000 more pieces (random lower-case strings, half of
    them word-initial, among them repeats of existing pieces with other scores) loads in a fraction of a second and
    still segments exactly as the library does."""
    import random
    import time

    from tokenizers import Tokenizer

    path, _ = files["published"]
    d = json.load(open(path, encoding="utf-8"))
    rng = random.Random(5)
    alpha = "abcdefghijklmnopqrstuvwxyz"
    extra = []
    for _ in range(120_000):
        w = ("▁" if rng.random() < 0.5 else "") + "".join(rng.choice(alpha) for _ in range(rng.randint(1, 9)))
        extra.append([w, -8.0 - rng.random() * 6])      # (repeats happen: a later entry wins, as in the library's map)
    mask = d["model"]["vocab"].pop()                     # <mask> stays last
    d["model"]["vocab"] += extra + [mask]
    for a in d["added_tokens"]:
        if a["content"] == "<mask>":
            a["id"] = len(d["model"]["vocab"]) - 1
    big = str(tmp_path / "big.json")
    json.dump(d, open(big, "w", encoding="utf-8"), ensure_ascii=False)
    texts = [" ".join("".join(rng.choice(alpha) for _ in range(rng.randint(1, 10))) for _ in range(100)) for _ in range(200)]
    want = [e.ids for e in Tokenizer.from_file(big).encode_batch(texts)]
    t0 = time.time()
    got = F.encode_all(lib, big, texts)
    assert got == want
    assert time.time() - t0 < 20.0


def test_truncation_and_handle_properties(lib, files):
    """max_length counts <s> and </s>; 0 takes the file's truncation.max_length; padding is <pad>; lookups by piece."""
    path, tok = files["published"]
    long_text = "word " * 400
    tok.enable_truncation(max_length=16)
    want = tok.encode(long_text).ids
    tok.enable_truncation(max_length=512)
    assert F.encode_all(lib, path, [long_text], max_length=16)[0] == want and len(want) == 16
    assert len(F.encode_all(lib, path, [long_text])[0]) == len(tok.encode(long_text).ids)
    h = C.c_void_p()
    _lib.check(lib.cs_tokenizer_create_from_json(path.encode(), 0, C.byref(h)))
    assert lib.cs_tokenizer_vocab_size(h) == tok.get_vocab_size()
    assert lib.cs_tokenizer_max_length(h) == 512
    for piece in ("<s>", "<pad>", "</s>", "<unk>", "<mask>", "▁"):
        assert lib.cs_tokenizer_token_to_id(h, piece.encode()) == tok.token_to_id(piece)
    assert lib.cs_tokenizer_token_to_id(h, b"no such piece at all") == -1
    # two texts of different lengths: the shorter row is padded with <pad> (id 1) under a zero mask
    import numpy as np

    texts = ["a", "a much longer text than the first one"]
    enc = [t.encode() for t in texts]
    offs = (C.c_uint64 * 3)(0, len(enc[0]), len(enc[0]) + len(enc[1]))
    L = C.c_uint32()
    _lib.check(lib.cs_tokenizer_encode_batch(h, b"".join(enc), offs, 2, 0, None, None, 0, C.byref(L)))
    ids = np.zeros((2, L.value), np.int32)
    mask = np.zeros((2, L.value), np.int32)
    _lib.check(lib.cs_tokenizer_encode_batch(h, b"".join(enc), offs, 2, 0, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                             mask.ctypes.data_as(C.POINTER(C.c_int32)), L.value, C.byref(L)))
    n0 = int(mask[0].sum())
    assert n0 < L.value and (ids[0, n0:] == tok.token_to_id("<pad>")).all() and mask[1].all()
    lib.cs_tokenizer_destroy(h)


def test_invalid_utf8_becomes_replacement_characters(lib, files):
    """The C boundary takes bytes: ill-formed sequences read as U+FFFD, as a Rust caller's from_utf8_lossy would hand them over."""
    path, tok = files["published"]
    raw = b"ok \xff\xfe bad \xe2\x82 cut \xf0\x9f\x98"
    h = C.c_void_p()
    _lib.check(lib.cs_tokenizer_create_from_json(path.encode(), 0, C.byref(h)))
    offs = (C.c_uint64 * 2)(0, len(raw))
    L = C.c_uint32()
    _lib.check(lib.cs_tokenizer_encode_batch(h, raw, offs, 1, 0, None, None, 0, C.byref(L)))
    import numpy as np

    ids = np.zeros(L.value, np.int32)
    _lib.check(lib.cs_tokenizer_encode_batch(h, raw, offs, 1, 0, ids.ctypes.data_as(C.POINTER(C.c_int32)), None, L.value, C.byref(L)))
    lib.cs_tokenizer_destroy(h)
    assert [int(x) for x in ids] == tok.encode(raw.decode("utf-8", errors="replace")).ids


def test_components_that_are_not_built_are_refused(lib, files, tmp_path):
    """A tokenizer.json asking for anything unigram.cpp does not restate fails at load with the reference's error prefix —
    never a silently different tokenisation."""
    path, _ = files["published"]
    base = json.load(open(path, encoding="utf-8"))

    def refuses(mutate, code):
        d = json.loads(json.dumps(base))
        mutate(d)
        p = tmp_path / "t.json"
        p.write_text(json.dumps(d), encoding="utf-8")
        h = C.c_void_p()
        st = lib.cs_tokenizer_create_from_json(str(p).encode(), 0, C.byref(h))
        assert st == code, (st, _lib.last_error())
        assert "Failed to initialize embedding model" in _lib.last_error()

    refuses(lambda d: d["normalizer"]["normalizers"].append({"type": "NFKC"}), _lib.CS_ERR_UNSUPPORTED)
    refuses(lambda d: d["normalizer"]["normalizers"].__setitem__(1, {"type": "Replace", "pattern": {"Regex": "a+"}, "content": "b"}),
            _lib.CS_ERR_UNSUPPORTED)
    refuses(lambda d: d["model"].__setitem__("byte_fallback", True), _lib.CS_ERR_UNSUPPORTED)
    refuses(lambda d: d.__setitem__("pre_tokenizer", {"type": "ByteLevel"}), _lib.CS_ERR_UNSUPPORTED)
    refuses(lambda d: d.__setitem__("post_processor", {"type": "RobertaProcessing"}), _lib.CS_ERR_UNSUPPORTED)
    refuses(lambda d: d["added_tokens"][4].__setitem__("single_word", True), _lib.CS_ERR_UNSUPPORTED)
    refuses(lambda d: d["model"].__setitem__("unk_id", 99999), _lib.CS_ERR_UNSUPPORTED)
    refuses(lambda d: d["model"].__setitem__("vocab", [["a", "x"]]), _lib.CS_ERR_BAD_ARG)
    refuses(lambda d: d["normalizer"]["normalizers"][0].__setitem__("precompiled_charsmap", "AAAA"), _lib.CS_ERR_BAD_ARG)


def _raw_json_with(path_in, path_out, marker, raw):
    """Re-writes a tokenizer.json with the ASCII `marker` replaced by the raw bytes `raw` (a JSON writer would never
    emit them; a downloaded file can hold anything, and checkpoint.cpp's JSON reader passes string bytes through)."""
    data = open(path_in, "rb").read()
    assert data.count(marker) >= 1
    open(path_out, "wb").write(data.replace(marker, raw))


@pytest.mark.parametrize("where", ["replace_content", "metaspace", "piece", "added", "charsmap"])
def test_ill_formed_utf8_in_the_file_is_refused_at_load_time(lib, files, tmp_path, where):
    """bytes the file splices into a text AFTER the input was sanitised (a Replace content, the Metaspace
    replacement, a vocabulary piece, an added token, a replacement string of the precompiled character map) must be
    well-formed UTF-8 themselves — a lone 0xE2 as Replace content made Strip walk past the end of the string at ENCODE
    time.  Such a file is refused when it is loaded, with the reference's wording."""
    import base64

    path, _ = files["converter"]  # Precompiled, Strip(right), Replace -> U+2581; Metaspace
    d = json.load(open(path, encoding="utf-8"))
    if where == "replace_content":
        d["normalizer"]["normalizers"].append({"type": "Replace", "pattern": {"String": "a"}, "content": "QQMARKQQ"})
        d["normalizer"]["normalizers"].append({"type": "Strip", "strip_left": True, "strip_right": True})
    elif where == "metaspace":
        pt = d["pre_tokenizer"]
        (pt["pretokenizers"][-1] if pt["type"] == "Sequence" else pt)["replacement"] = "QQMARKQQ"
    elif where == "piece":
        d["model"]["vocab"][10][0] = "QQMARKQQ"
    elif where == "added":
        d["added_tokens"].append({"id": 11, "content": "QQMARKQQ", "single_word": False, "lstrip": False, "rstrip": False,
                                  "normalized": False, "special": True})
    else:
        blob = bytearray(base64.b64decode(d["normalizer"]["normalizers"][0]["precompiled_charsmap"]))
        tsize = int.from_bytes(blob[:4], "little")
        pool = 4 + tsize
        # the first byte of the first multi-byte replacement in the pool becomes a lone continuation byte
        i = next(k for k in range(pool, len(blob)) if blob[k] >= 0xC2)
        blob[i] = 0x80
        d["normalizer"]["normalizers"][0]["precompiled_charsmap"] = base64.b64encode(bytes(blob)).decode()
    good = tmp_path / "good.json"
    good.write_text(json.dumps(d), encoding="utf-8")
    bad = tmp_path / "bad.json"
    if where == "charsmap":
        bad = good
    else:
        _raw_json_with(good, bad, b"QQMARKQQ", b"\xe2")
    h = C.c_void_p()
    st = lib.cs_tokenizer_create_from_json(str(bad).encode(), 0, C.byref(h))
    assert st == _lib.CS_ERR_BAD_ARG, (where, st)
    msg = _lib.last_error()
    assert msg.startswith("Failed to initialize embedding model") and "UTF-8" in msg, msg
    assert not h.value
