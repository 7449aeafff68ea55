"""The byte-level BPE text pipeline (csrc/bpe.cpp behind cs_tokenizer_create_from_json) against the `tokenizers` library — the
version the reference pins (Cargo.lock: tokenizers 0.22.2; the registry's JinaEmbeddingsV2BaseCode ships a RoBERTa-style
byte-level BPE tokenizer.json, /root/reference/src/embed/embedder.rs:40-41, :112).  No such file is on disk (no network), so
the fixtures are trained here with the library's own BpeTrainer on a small code-like corpus and saved through its own
serialiser, in the arrangements such files come in: RobertaProcessing or TemplateProcessing or no template, with and without
the prefix space, a Digits step in front of ByteLevel, ByteLevel without its regex, ignore_merges, the legacy ("a b") and the
current ([a, b]) merges serialisation.  Every comparison is id for id.  CPU only."""
import json
import os
import random
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

pytest.importorskip("tokenizers")

import unigram_fuzz as F  # noqa: E402  (encode_all: the C ABI round trip)

from codesearch_amd import _lib  # noqa: E402

SPECIALS = ["<s>", "<pad>", "</s>", "<unk>", "<mask>"]

CORPUS = [
    "def authenticate(user, password):\n    return check_hash(user.password_hash, password)\n",
    "fn main() {\n    let args: Vec<String> = std::env::args().collect();\n    println!(\"{:?}\", args);\n}\n",
    "class VectorStore:\n    \"\"\"stores 384-d embeddings\"\"\"\n    def search(self, query, k=10):\n        pass\n",
    "// it's the caller's job: don't free what you didn't allocate, they'll crash\n",
    "for (int i = 0; i < 1024; ++i) { sum += a[i] * b[i]; }  /* 3.14159 2024-10-05 */\n",
    "SELECT id, name FROM users WHERE age >= 18 AND name LIKE '%smith%';\n",
    "const π = 3.14; let naïve = \"café\"; // 你好，世界 \U0001F680\U0001F525\n",
    "\tif err != nil {\n\t\treturn fmt.Errorf(\"read %s: %w\", path, err)\n\t}\n",
] * 40

TEXTS = [
    "", " ", "  ", "a", " a", "a ", "hello world", "  leading and trailing  ", "tabs\tand\nnewlines\r\n\r\nhere",
    "it's I'm you're they've we'll he'd don't 'tis 'S 'RE O'Neill's", "x=1;y=22;z=333 v1.2.3 0x7fff 1e-9",
    "snake_case camelCase PascalCase SCREAMING_CASE kebab-case", "((nested [brackets] {and} <angles>))", "!!!???...,,,;;;",
    "emoji \U0001F680\U0001F525 and \U0001F469‍\U0001F469‍\U0001F467 zwj", "naïve café Ünïcödé ß ǆ",
    "你好，世界。日本語のテキスト 한국어", "مرحبا بالعالم שלום",
    "१२३ ٤٥٦ Ⅷ ½ ²", "a b c　de f", "line1\n\n\n   line2\n \n\t x", "trailing newline\n", "\n", "\n\n x",
    " \n", "x \n y", "<s>literal specials</s> in <mask> the <pad> text<unk>", "before<mask>after", "a <mask> b", "<s><s>", "</s>",
    "def f(x):\n    return x ** 2  # square\n", "std::vector<std::pair<int, float>> v{{1, 2.0f}};", "#include <stdio.h>\nint main(void){return 0;}",
    # runs of spaces and the placeholder tokens (the normalized added tokens of the ModernBERT arrangement)
    "class A:\n    def f(self):\n        if x:\n            return  y\n" + " " * 27 + "z" + " " * 24 + "|" + " " * 25,
    "ping |||IP_ADDRESS||| from|||EMAIL_ADDRESS|||  |||PHONE_NUMBER|||||| ||IP_ADDRESS|||", "   <mask>  x   </s>    ", "\t  \t   \n  \n   x",
    "a" * 300, "ab " * 200, "9" * 50, "'" * 7 + "s't're", "end with apostrophe'", "\x00\x01\x7f control", "� replacement ﻿ bom",
    # canonical (de)composition for the NFC variants: decomposed accents, reordering of marks, Hangul jamo, singletons, exclusions
    "cafe\u0301 nai\u0308ve A\u030a \u212b \u2126 \u1e9b\u0323 q\u0307\u0323 q\u0323\u0307", "\u1100\u1161\u11a8 \u1112\u1161\u11ab\uae00 \u1100\u1161 \u11a8",
    "\u0958 \u0915\u093c \u0f43 \u0f42\u0fb7 \ufb1f \u05f2\u05b7 \u2adc \u0344 \u0308\u0301", "a\u0301\u0301\u0328 e\u0328\u0301 \u0301a o\u0302\u0303 \u1ed7 D\u0307\u0323 \u1e0c\u0307",
]


# what the published ModernBERT tokenizer.json holds beside its specials (GPT-NeoX / OLMo lineage): plain added tokens,
# normalized and not special — three placeholders and the runs of 2 .. 24 spaces indented code is made of
MODERN_ADDED = ["|||IP_ADDRESS|||", "|||EMAIL_ADDRESS|||", "|||PHONE_NUMBER|||"] + [" " * n for n in range(24, 1, -1)]


def build(path, *, post="roberta", add_prefix_space=False, digits=None, use_regex=True, ignore_merges=False, legacy_merges=False,
          trim=True, nfc=False, modern_added=False):
    from tokenizers import Tokenizer, decoders, models, normalizers, pre_tokenizers, processors, trainers

    tok = Tokenizer(models.BPE())
    if nfc == "sequence":
        tok.normalizer = normalizers.Sequence([normalizers.NFC()])
    elif nfc:
        tok.normalizer = normalizers.NFC()   # ModernBERT's arrangement
    bl = pre_tokenizers.ByteLevel(add_prefix_space=add_prefix_space, use_regex=use_regex)
    tok.pre_tokenizer = bl if digits is None else pre_tokenizers.Sequence([pre_tokenizers.Digits(individual_digits=digits), bl])
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=700, special_tokens=SPECIALS, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                  show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    if modern_added:
        from tokenizers import AddedToken

        tok.add_tokens([AddedToken(t, normalized=True, special=False) for t in MODERN_ADDED])
    if post == "roberta":
        tok.post_processor = processors.RobertaProcessing(sep=("</s>", tok.token_to_id("</s>")), cls=("<s>", tok.token_to_id("<s>")),
                                                           trim_offsets=trim, add_prefix_space=add_prefix_space)
    elif post == "template":
        tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                            special_tokens=[("<s>", tok.token_to_id("<s>")), ("</s>", tok.token_to_id("</s>"))])
    elif post == "bytelevel":
        tok.post_processor = processors.ByteLevel(trim_offsets=trim)
    tok.save(path)
    doc = json.load(open(path, encoding="utf-8"))
    changed = False
    if ignore_merges:
        doc["model"]["ignore_merges"] = True
        changed = True
    if legacy_merges and doc["model"]["merges"] and isinstance(doc["model"]["merges"][0], list):
        doc["model"]["merges"] = [" ".join(m) for m in doc["model"]["merges"]]
        changed = True
    for a in doc.get("added_tokens", []):
        if a["content"] == "<mask>":
            a["lstrip"] = True   # RoBERTa's <mask> takes the space in front of it
            changed = True
    if changed:
        json.dump(doc, open(path, "w", encoding="utf-8"), ensure_ascii=False)
    return Tokenizer.from_file(path)


VARIANTS = {
    "roberta": dict(post="roberta"),
    "roberta_prefix_space": dict(post="roberta", add_prefix_space=True),
    "template_legacy_merges": dict(post="template", legacy_merges=True),
    "no_template": dict(post="bytelevel"),
    "digits_individual": dict(post="roberta", digits=True),
    "digits_contiguous_prefix": dict(post="template", digits=False, add_prefix_space=True),
    "no_regex": dict(post="roberta", use_regex=False),
    "ignore_merges": dict(post="roberta", ignore_merges=True),
    "nfc_template": dict(post="template", nfc=True),
    "nfc_sequence_prefix": dict(post="roberta", nfc="sequence", add_prefix_space=True),
    "modernbert_added_tokens": dict(post="template", nfc=True, modern_added=True),
    "normalized_added_no_normalizer": dict(post="roberta", modern_added=True),
}


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("bpe")
    return {name: (str(d / f"{name}.json"), build(str(d / f"{name}.json"), **kw)) for name, kw in VARIANTS.items()}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_fixture_texts_match_the_library(lib, files, name):
    path, tok = files[name]
    got = F.encode_all(lib, path, TEXTS, max_length=100000)   # (no truncation: the library object has none enabled)
    live = [tok.encode(t).ids for t in TEXTS]
    for t, g, w in zip(TEXTS, got, live):
        assert g == w, (name, t, g, w)
    if name.startswith("roberta"):
        assert all(g[0] == tok.token_to_id("<s>") and g[-1] == tok.token_to_id("</s>") for g in got)
        assert got[0] == [tok.token_to_id("<s>"), tok.token_to_id("</s>")]    # the empty text
    if name == "no_template":
        assert got[0] == []


def random_text(rng):
    pools = [
        "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", "0123456789", " \t\n\r", "  ", "_-+*/=<>!&|^%~?:;,.()[]{}\"'`#@$\\",
        "'s't're've'm'll'd", "äöüßéèêñçøåžšč", "你好世界日本語한국어",
        "\U0001F680\U0001F525\U0001F44D\U0001F3FD\U0001F469‍\U0001F4BB", "  　 ", "١٢٣४५६", "αβγδλπΣΩ", "<s></s><mask><pad><unk>",
    ]
    out = []
    for _ in range(rng.randint(1, 40)):
        pool = rng.choice(pools)
        if pool.startswith("<s>") and rng.random() < 0.7:
            out.append(rng.choice(SPECIALS))
        else:
            out.append("".join(rng.choice(pool) for _ in range(rng.randint(1, 6))))
    return "".join(out)


@pytest.mark.parametrize("name", list(VARIANTS))
def test_random_strings_match_the_library(lib, files, name):
    path, tok = files[name]
    rng = random.Random(sum(map(ord, name)) + 7)
    texts = [random_text(rng) for _ in range(1500)]
    got = F.encode_all(lib, path, texts, max_length=100000)
    for t, g in zip(texts, got):
        w = tok.encode(t).ids
        assert g == w, (name, t, g, w)


def test_truncation_and_batch_padding(lib, files):
    """max_length counts the template's tokens; truncation is on the right, before <eos>."""
    path, tok = files["roberta"]
    tok.enable_truncation(max_length=16)
    long_text = "let total = values.iter().map(|v| v * 2).sum::<i64>(); " * 6
    got = F.encode_all(lib, path, [long_text, "x"], max_length=16)
    assert got[0] == tok.encode(long_text).ids and len(got[0]) == 16
    assert got[1] == tok.encode("x").ids
    tok.no_truncation()


def test_refusals_are_worded(lib, tmp_path):
    import ctypes as C

    path = str(tmp_path / "t.json")
    build(path)
    doc = json.load(open(path, encoding="utf-8"))

    def expect(mutate, code, needle):
        d = json.loads(json.dumps(doc))
        mutate(d)
        p = str(tmp_path / "bad.json")
        json.dump(d, open(p, "w", encoding="utf-8"), ensure_ascii=False)
        h = C.c_void_p()
        rc = lib.cs_tokenizer_create_from_json(p.encode(), 0, C.byref(h))
        assert rc == code and needle in lib.cs_last_error().decode(), (rc, lib.cs_last_error().decode())

    expect(lambda d: d["model"].__setitem__("dropout", 0.1), _lib.CS_ERR_UNSUPPORTED, "dropout")
    expect(lambda d: d["model"].__setitem__("byte_fallback", True), _lib.CS_ERR_UNSUPPORTED, "byte_fallback")
    expect(lambda d: d["model"].__setitem__("end_of_word_suffix", "</w>"), _lib.CS_ERR_UNSUPPORTED, "end_of_word_suffix")
    expect(lambda d: d.__setitem__("normalizer", {"type": "NFKC"}), _lib.CS_ERR_UNSUPPORTED, "normalizer")
    expect(lambda d: d.__setitem__("pre_tokenizer", {"type": "Whitespace"}), _lib.CS_ERR_UNSUPPORTED, "pre_tokenizer")
    expect(lambda d: d["model"]["merges"].append(["zz", "qq"]), _lib.CS_ERR_BAD_ARG, "outside the vocabulary")
    expect(lambda d: d["model"].__setitem__("merges", ["a b c"]), _lib.CS_ERR_BAD_ARG, "not two tokens")
    expect(lambda d: d["model"].__setitem__("type", "WordLevel"), _lib.CS_ERR_UNSUPPORTED, "WordPiece, Unigram and BPE")
