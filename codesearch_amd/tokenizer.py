"""BERT WordPiece tokenizer for the text entry points of the embedder.

The reference tokenises inside fastembed with the `tokenizers` crate 0.22.2 (Cargo.lock),
configured from the model's tokenizer.json: BertNormalizer(clean_text, handle_chinese_chars,
strip_accents=None, lowercase) -> BertPreTokenizer -> WordPiece("##", [UNK], 100 chars) ->
[CLS] ... [SEP], truncation to 512, padding to the batch's longest sequence
(call site /root/reference/src/embed/embedder.rs:286-289).  This module restates that
published algorithm; tests/golden/tokenizer_golden.json pins it against the `tokenizers`
0.22.2 Python wheel (same crate, same version) on a committed vocabulary.
"""
from __future__ import annotations

import unicodedata
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np


def _is_control(c: str) -> bool:
    if c in "\t\n\r":
        return False
    return unicodedata.category(c) in ("Cc", "Cf", "Cn", "Co")


_WS = {0x9, 0xA, 0xB, 0xC, 0xD, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000} | set(
    range(0x2000, 0x200B))  # Unicode White_Space (Rust char::is_whitespace)


def _is_whitespace(c: str) -> bool:
    return ord(c) in _WS


def _is_chinese(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or
            0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0x2B920 <= cp <= 0x2CEAF or
            0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def _is_punct(c: str) -> bool:
    cp = ord(c)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(c).startswith("P")


class BertWordPieceTokenizer:
    def __init__(self, vocab: Dict[str, int], lowercase: bool = True, unk_token: str = "[UNK]",
                 cls_token: str = "[CLS]", sep_token: str = "[SEP]", pad_token: str = "[PAD]",
                 max_input_chars_per_word: int = 100, max_length: int = 512):
        self.vocab = vocab
        self.lowercase = lowercase
        self.unk_id = vocab[unk_token]
        self.cls_id = vocab[cls_token]
        self.sep_id = vocab[sep_token]
        self.pad_id = vocab[pad_token]
        self.max_chars = max_input_chars_per_word
        self.max_length = max_length

    @classmethod
    def from_vocab_file(cls, path: str, **kw) -> "BertWordPieceTokenizer":
        """vocab.txt: one token per line, id = line number."""
        with open(path, encoding="utf-8") as f:
            vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
        return cls(vocab, **kw)

    # -- BertNormalizer
    def normalize(self, text: str) -> str:
        out = []
        for c in text:  # clean_text
            cp = ord(c)
            if cp == 0 or cp == 0xFFFD or _is_control(c):
                continue
            out.append(" " if _is_whitespace(c) else c)
        text = "".join(out)
        out = []
        for c in text:  # handle_chinese_chars
            if _is_chinese(ord(c)):
                out += [" ", c, " "]
            else:
                out.append(c)
        text = "".join(out)
        if self.lowercase:  # strip_accents = None -> follows lowercase
            text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn")
            text = text.lower()
        return text

    # -- BertPreTokenizer: whitespace split (removed), punctuation isolated
    @staticmethod
    def pre_tokenize(text: str) -> List[str]:
        words: List[str] = []
        cur: List[str] = []
        for c in text:
            if _is_whitespace(c):
                if cur:
                    words.append("".join(cur))
                    cur = []
            elif _is_punct(c):
                if cur:
                    words.append("".join(cur))
                    cur = []
                words.append(c)
            else:
                cur.append(c)
        if cur:
            words.append("".join(cur))
        return words

    # -- WordPiece: greedy longest-match-first, "##" continuation, whole word -> [UNK] on failure
    def wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk_id]
        ids: List[int] = []
        start, n = 0, len(word)
        while start < n:
            end = n
            found = None
            while start < end:
                piece = word[start:end]
                if start > 0:
                    piece = "##" + piece
                if piece in self.vocab:
                    found = self.vocab[piece]
                    break
                end -= 1
            if found is None:
                return [self.unk_id]
            ids.append(found)
            start = end
        return ids

    def encode(self, text: str) -> List[int]:
        """-> ids with [CLS] ... [SEP], truncated to max_length."""
        ids: List[int] = []
        for w in self.pre_tokenize(self.normalize(text)):
            ids.extend(self.wordpiece(w))
        ids = ids[: self.max_length - 2]
        return [self.cls_id] + ids + [self.sep_id]

    def encode_batch(self, texts: Sequence[str], max_length: int = None) -> Tuple[np.ndarray, np.ndarray]:
        """-> (ids [n, L], mask [n, L]) int32, padded to the batch's longest sequence."""
        if max_length is not None and max_length != self.max_length:
            self.max_length = max_length
        enc = [self.encode(t) for t in texts]
        L = max((len(e) for e in enc), default=0)
        ids = np.full((len(enc), L), self.pad_id, np.int32)
        mask = np.zeros((len(enc), L), np.int32)
        for i, e in enumerate(enc):
            ids[i, : len(e)] = e
            mask[i, : len(e)] = 1
        return ids, mask
