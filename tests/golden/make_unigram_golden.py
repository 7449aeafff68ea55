#!/usr/bin/env python3
"""A SentencePiece-unigram tokenizer.json of the XLM-R kind (what the registry's multilingual models ship:
intfloat/multilingual-e5-small, sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2 — /root/reference/src/embed/
embedder.rs:58,70) built WITHOUT network: a small unigram model trained here by `sentencepiece` on a fixed text (a little
English and code plus a few multilingual lines; normalisation rule nmt_nfkc — the real precompiled character map), wrapped the way
transformers' XLMRobertaConverter wraps the published files.  `build(path, style)` writes the file; the tests compare the
C++ tokenizer with the `tokenizers` library on it.  Run as a script: writes tests/golden/unigram_golden.json (texts -> ids by
`tokenizers` 0.22, the version the reference pins) for the committed fixture."""
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

MULTI = [
    "Bonjour le monde, ça va très bien aujourd'hui ? L'élève a reçu un cadeau à Noël.",
    "Der schnelle braune Fuchs springt über den faulen Hund. Größe, Straße, Übermut.",
    "El niño comió una manzana y después se fue al colegio. ¿Dónde está la biblioteca?",
    "Быстрая коричневая лиса перепрыгивает через ленивую собаку. Привет, мир!",
    "Η γρήγορη καφέ αλεπού πηδάει πάνω από το τεμπέλικο σκυλί.",
    "快速的棕色狐狸跳过了懒狗。你好，世界！今天天气很好。",
    "素早い茶色の狐がのろまな犬を飛び越える。こんにちは世界。プログラミング言語",
    "빠른 갈색 여우가 게으른 개를 뛰어넘습니다. 안녕하세요 세계.",
    "الثعلب البني السريع يقفز فوق الكلب الكسول. مرحبا بالعالم.",
    "तेज़ भूरी लोमड़ी आलसी कुत्ते के ऊपर कूदती है। नमस्ते दुनिया।",
    "fn main() { let v: Vec<u32> = (0..10).collect(); println!(\"{:?}\", v); }",
    "def embed(self, texts: List[str]) -> np.ndarray: return self.model.encode(texts)",
    "ｆｕｌｌｗｉｄｔｈ　ｔｅｘｔ １２３ ＡＢＣ ﬁne ﬂow ½ ² ™ … — “quotes” ‘single’",
    "emoji \U0001F600 \U0001F44D\U0001F3FD \U0001F468‍\U0001F469‍\U0001F467 \U0001F1EB\U0001F1F7 and é ä ỗ ñ combining marks",
]

TEXTS = [
    "", " ", "hello world", "Hello   World  ", "  leading and trailing  ", "fn main() { println!(\"hi\"); }",
    "naïve café résumé", "élève ä ỗ é̂̃x", "ｆｕｌｌ　ｗｉｄｔｈ １２", "ﬁne ﬂow ½ ²",
    "快速的棕色狐狸", "こんにちは世界", "안녕하세요 세계", "한 글 jamo",
    "Привет, мир!", "مرحبا بالعالم", "नमस्ते दुनिया",
    "emoji \U0001F600 \U0001F44D\U0001F3FD \U0001F468‍\U0001F469‍\U0001F467 \U0001F1EB\U0001F1F7", "tab\tsep\nnew\r\nline\x0bvt\x0cff", "a▁b ▁c ▁▁d",
    "<s> literal </s> and <mask> and <unk><pad> x  <mask>y", "x" * 300, "zzzzqqqq  　end", "UPPER lower MiXeD 12345 67.89 1e-5",
    "https://example.com/path?query=1&x=y#frag user@example.com", "؀١ ‍​ zero​width  nbsp em ls",
    "def calculate(a, b):\n    return a + b  # sum", "SELECT * FROM chunks WHERE id = 7;", "…—“”‘’«»„‚",
    "\x00nul\x01ctl\x7fdel", "﻿bom � replacement ­ soft­hyphen", "กำ เก้า thai ཀ་ tibetan",
] + MULTI


ENGLISH = """
The quick brown fox jumps over the lazy dog while the developer searches the code base for a function definition.
Semantic code search embeds every chunk of source code as a vector and ranks chunks by cosine similarity to the query.
fn main() { let args: Vec<String> = std::env::args().collect(); println!("{:?}", args); }
pub fn embed_batch(&mut self, texts: Vec<String>) -> Result<Vec<Vec<f32>>> { self.model.embed(texts, None) }
def calculate(a, b): return a + b
class VectorStore: def __init__(self, path, dimensions): self.path = path; self.dimensions = dimensions
for (int i = 0; i < n; ++i) { sum += values[i] * weights[i]; }
SELECT id, path, start_line, end_line FROM chunks WHERE language = 'rust' ORDER BY score DESC LIMIT 10;
import numpy as np; x = np.zeros((batch, hidden), dtype=np.float32)
The index stores one embedding per chunk together with the file path, the line range and the kind of the chunk.
A tokenizer splits text into pieces, maps the pieces to integer ids and adds the special tokens of the model.
Error: failed to initialize embedding model; cannot open tokenizer.json in the model directory.
if let Some(result) = cache.get(&key) { return Ok(result.clone()); } else { cache.insert(key, value); }
const response = await fetch(url, { method: "POST", headers: { "Content-Type": "application/json" }, body });
Unicode normalisation maps full width letters, ligatures and compatibility characters to their plain forms.
https://example.com/docs/api/v1/search?q=vector+database&limit=25 user@example.com 192.168.0.1 2024-01-31T12:00:00Z
struct Point { x: f64, y: f64 } impl Display for Point { fn fmt(&self, f: &mut Formatter) -> fmt::Result { write!(f, "({}, {})", self.x, self.y) } }
""".strip().split("\n")


def corpus():
    """A FIXED training text (the fixture's ids depend on it): the multilingual lines and a little English and code."""
    return list(MULTI) * 8 + list(ENGLISH) * 8


def train(vocab_size=700):
    import sentencepiece as spm

    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(corpus()), model_writer=buf, vocab_size=vocab_size, model_type="unigram",
                                   character_coverage=0.9995, normalization_rule_name="nmt_nfkc", num_threads=1,
                                   minloglevel=2, input_sentence_size=0, shuffle_input_sentence=False, hard_vocab_limit=False)
    return buf.getvalue()


def build(path, style="published", model_bytes=None):
    """style 'published': the layout of the tokenizer.json files on the hub (Precompiled + Replace(' {2,}' -> ' '),
    WhitespaceSplit + Metaspace); 'converter': what today's transformers writes (Precompiled, Strip(right), Replace(' {2,}'
    -> U+2581), Metaspace alone)."""
    from sentencepiece import sentencepiece_model_pb2 as pb
    from tokenizers import AddedToken, Regex, Tokenizer, normalizers, pre_tokenizers, processors
    from tokenizers.models import Unigram

    m = pb.ModelProto()
    m.ParseFromString(model_bytes or train())
    vocab = [("<s>", 0.0), ("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0)]
    vocab += [(p.piece, p.score) for p in m.pieces[3:]]
    vocab += [("<mask>", 0.0)]
    tok = Tokenizer(Unigram(vocab, unk_id=3))
    charsmap = m.normalizer_spec.precompiled_charsmap
    if style == "published":
        tok.normalizer = normalizers.Sequence([normalizers.Precompiled(charsmap), normalizers.Replace(Regex(" {2,}"), " ")])
        tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(),
                                                     pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always")])
    else:
        tok.normalizer = normalizers.Sequence([normalizers.Precompiled(charsmap), normalizers.Strip(left=False, right=True),
                                               normalizers.Replace(Regex(" {2,}"), "▁")])
        tok.pre_tokenizer = pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always")
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                       special_tokens=[("<s>", 0), ("</s>", 2)])
    tok.add_special_tokens(["<s>", "<pad>", "</s>", "<unk>", AddedToken("<mask>", lstrip=True, special=True)])
    tok.enable_truncation(max_length=512)
    tok.save(path)
    return tok


def main():
    out = {}
    mb = train()
    for style in ("published", "converter"):
        path = os.path.join("/tmp", f"unigram_{style}.json")
        tok = build(path, style, mb)
        out[style] = {"ids": [tok.encode(t).ids for t in TEXTS]}
    out["texts"] = TEXTS
    import sentencepiece
    import tokenizers
    out["made_with"] = {"tokenizers": tokenizers.__version__, "sentencepiece": sentencepiece.__version__}
    with open(os.path.join(HERE, "unigram_golden.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=True)
    print("wrote", os.path.join(HERE, "unigram_golden.json"))


if __name__ == "__main__":
    sys.exit(main())
