"""Generates tests/golden/tokenizer_golden.json — known answers for the WordPiece text path.

Run in the build container:  python tests/golden/make_tokenizer_golden.py

Source of truth: the `tokenizers` 0.22.2 Python wheel — the same crate and version the
reference pins in Cargo.lock and reaches through fastembed — configured the way BERT-family
tokenizer.json files are: BertNormalizer(clean_text, handle_chinese_chars, strip_accents=None,
lowercase) / BertPreTokenizer / WordPiece("##", "[UNK]", 100) / [CLS] A [SEP] / truncation /
batch-longest padding.  The real bge-small vocab.txt is not reachable offline, so the
vocabulary is synthetic (BERT's special-token ids kept: [PAD]=0 [UNK]=100 [CLS]=101 [SEP]=102)
and is stored in the fixture together with the inputs and expected ids.
"""
import json
import os
import string

from tokenizers import Tokenizer
from tokenizers.models import WordPiece
from tokenizers.normalizers import BertNormalizer
from tokenizers.pre_tokenizers import BertPreTokenizer
from tokenizers.processors import BertProcessing


def build_vocab():
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += list(string.ascii_lowercase) + list(string.digits) + list(string.punctuation)
    toks += ["##" + c for c in string.ascii_lowercase + string.digits]
    words = """fn def return self let mut pub struct impl class import from for while if else match use
    async await const static void int float string vec result option error none some true false new
    print println main test function method value index search embed query chunk vector store build
    insert delete file path line code doc context signature name user data config parse token batch
    model hello world the a of to and in is it that this with as be on not or are by an at have has
    handle request response server client read write open close error authenticate calculate cosine
    similarity database connection rust python java type trait enum where loop break continue
    naive cafe resume uber strasse""".split()
    toks += [w for w in dict.fromkeys(words) if w not in toks]
    toks += ["##" + s for s in ["ing", "ed", "er", "s", "es", "tion", "ment", "able", "ly", "ize", "or", "al",
                                 "_", "ed_", "fn", "test", "able", "ness", "ity", "ch", "ex", "un", "re"]]
    toks += ["中", "文", "日", "本", "ß", "α", "β", "—", "“", "”", "é"]
    seen, out = set(), []
    for t in toks:
        if t not in seen:
            seen.add(t)
            out.append(t)
    return {t: i for i, t in enumerate(out)}


TEXTS = [
    "fn main() { println!(\"Hello, World!\"); }",
    "Context: File: test.rs > Function: test\nSignature: fn test()\nName: test\nDocumentation: Test function\nCode:\nfn test() { println!(\"test\"); }",
    "def calculate_cosine_similarity(a, b):\n    return dot(a, b) / (norm(a) * norm(b))",
    "Café naïve résumé Über Straße",
    "中文 mixed with English 日本語 text",
    "tabs\tand\r\nnewlines nbsp emspace",
    "control\x00chars\x07here�replacement​zero-width",
    "UPPERCASE lowerCamelCase snake_case kebab-case SCREAMING_SNAKE",
    "x" * 120 + " short",
    "unknownword zzzqqq handlers handling handled",
    "emoji \U0001F600 and symbols — “quoted” αβγ",
    "",
    "   ",
    "a",
    "pub struct VectorStore { env: Env, vectors: ArroyDatabase<Cosine>, next_id: u32 }",
    " ".join(["token"] * 600),
    "é combining accent and İ dotted capital I and ΣΣ sigma",
    "1234567890 3.14159 0xDEADBEEF 1e-12",
]


def main():
    vocab = build_vocab()
    tok = Tokenizer(WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
    tok.normalizer = BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=True)
    tok.pre_tokenizer = BertPreTokenizer()
    tok.post_processor = BertProcessing(("[SEP]", vocab["[SEP]"]), ("[CLS]", vocab["[CLS]"]))
    out = {"library": "tokenizers 0.22.2", "vocab": vocab, "cases": []}
    for max_len in (512, 16):
        tok.enable_truncation(max_len)
        tok.enable_padding(pad_id=vocab["[PAD]"], pad_token="[PAD]")
        for lo in range(0, len(TEXTS), 6):
            batch = TEXTS[lo:lo + 6]
            enc = tok.encode_batch(batch)
            out["cases"].append({"max_length": max_len, "texts": batch, "ids": [e.ids for e in enc],
                                 "mask": [e.attention_mask for e in enc]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tokenizer_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes; vocab", len(vocab))


# ---- second fixture: the tokenizer as fastembed configures it (special tokens registered as
# added tokens, so a literal "[SEP]" in the text becomes id 102), seeded random Unicode text that
# exercises every table of csrc/unicode_tables.inc, cased and uncased, two truncation lengths.
EXTRA_TOKENS = ["σ", "ς", "ω", "한", "ᄒ", "ᅡ", "##ᆫ", "##̇", "ǆ", "ss", "##ss", "😀", "ı", "к", "##к",
                "д", "я", "##я", "豈", "ᄀ", "Hello", "##World", "İ", "É"]
POOLS = [list(range(0x20, 0x7F)), list(range(0xA0, 0x250)), list(range(0x370, 0x400)), list(range(0x400, 0x460)),
         [0x9, 0xA, 0xD, 0x85, 0xA0, 0x2003, 0x200B, 0x3000, 0xFFFD, 0, 7, 0xAD, 0xFEFF, 0xE000],
         list(range(0x300, 0x370)), list(range(0x4E00, 0x4E20)) + [0xF900, 0xFA0E, 0x3400, 0x20000, 0x2F800],
         list(range(0xAC00, 0xAC40)) + [0xD7A3], list(range(0x2000, 0x2070)),
         list(range(0x1F600, 0x1F610)) + [0x1FAE0, 0x378], list(range(0x1E00, 0x2000)),
         [0x130, 0x131, 0x3A3, 0x3C2, 0x1C4, 0x1C5, 0xDF, 0x1E9E, 0x37E, 0x1FEF, 0x212A, 0x212B, 0x2126]]


def random_text(rng, words):
    parts = []
    for _ in range(rng.randint(0, 30)):
        r = rng.random()
        if r < 0.4:
            parts.append(rng.choice(words))
        elif r < 0.5:
            parts.append(rng.choice(words).upper())
        else:
            pool = rng.choice(POOLS)
            parts.append("".join(chr(rng.choice(pool)) for _ in range(rng.randint(1, 6))))
        parts.append(rng.choice(["", " ", " ", "\n", "_", "."]))
    return "".join(parts)


def main_special():
    import random

    from tokenizers import AddedToken

    vocab = build_vocab()
    for t in EXTRA_TOKENS:
        vocab.setdefault(t, len(vocab))
    words = list(vocab) + ["[CLS", "CLS]", "[[SEP]]", "[mask]"]
    rng = random.Random(20260227)
    out = {"library": "tokenizers 0.22.2", "vocab": vocab, "cases": []}
    for lowercase in (True, False):
        for max_len in (512, 24):
            tok = Tokenizer(WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
            tok.normalizer = BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None,
                                            lowercase=lowercase)
            tok.pre_tokenizer = BertPreTokenizer()
            tok.post_processor = BertProcessing(("[SEP]", vocab["[SEP]"]), ("[CLS]", vocab["[CLS]"]))
            tok.add_special_tokens([AddedToken(t, special=True, normalized=False)
                                    for t in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")])
            tok.enable_truncation(max_len)
            tok.enable_padding(pad_id=vocab["[PAD]"], pad_token="[PAD]")
            for _ in range(6):
                batch = [random_text(rng, words) for _ in range(6)]
                enc = tok.encode_batch(batch)
                out["cases"].append({"lowercase": lowercase, "max_length": max_len, "texts": batch,
                                     "ids": [e.ids for e in enc], "mask": [e.attention_mask for e in enc]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tokenizer_golden_special.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes; vocab", len(vocab), "cases", len(out["cases"]))


if __name__ == "__main__":
    main()
    main_special()
