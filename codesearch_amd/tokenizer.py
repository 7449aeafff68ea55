"""BERT WordPiece tokenizer for the text entry points of the embedder.

The reference tokenises inside fastembed with the `tokenizers` crate 0.22.2 (Cargo.lock),
configured from the model's tokenizer.json: special tokens cut out of the raw text ->
BertNormalizer(clean_text, handle_chinese_chars, strip_accents=None, lowercase) ->
BertPreTokenizer -> WordPiece("##", [UNK], 100 chars) -> [CLS] ... [SEP], truncation to 512,
padding to the batch's longest sequence (call site /root/reference/src/embed/embedder.rs:286-289).
The restatement of that algorithm is C++ (csrc/tokenizer.cpp) behind cs_tokenizer_* of the C ABI;
this module is its ctypes mirror.  tests/golden/tokenizer_golden*.json pin it against the
`tokenizers` 0.22.2 Python wheel (same crate, same version) on a committed vocabulary.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib


def pack_texts(texts: Sequence[str]):
    """-> (utf8 bytes, offsets[n+1] uint64): the text layout of the C ABI."""
    enc = [t.encode("utf-8", "replace") for t in texts]
    offsets = np.zeros(len(enc) + 1, np.uint64)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    return b"".join(enc), offsets


class WordPieceTokenizer:
    """The product tokenizer: cs_tokenizer_* of libcsgpu.so (csrc/tokenizer.cpp), the C++
    restatement of the `tokenizers` 0.22.2 BERT pipeline fastembed runs.  `vocab` is a
    {token: id} dict with contiguous ids, or use from_vocab_file for a vocab.txt.  A tokenizer.json
    (from_tokenizer_json / from_dir) may also hold a SentencePiece-unigram model — the XLM-R vocabulary of
    the registry's multilingual entries — which csrc/unigram.cpp runs behind the same handle."""

    def __init__(self, vocab: Dict[str, int] = None, lowercase: bool = True, max_length: int = 512,
                 vocab_file: str = None, tokenizer_json: str = None, model_dir: str = None):
        self._lib = _lib.load()
        h = C.c_void_p()
        if tokenizer_json is not None:   # the file fastembed loads; max_length 0 = the file's truncation length
            _lib.check(self._lib.cs_tokenizer_create_from_json(str(tokenizer_json).encode(), int(max_length or 0), C.byref(h)))
            max_length = 0
        elif model_dir is not None:      # tokenizer.json, else vocab.txt + tokenizer_config.json
            _lib.check(self._lib.cs_tokenizer_create_from_dir(str(model_dir).encode(), int(max_length or 0), C.byref(h)))
            max_length = 0
        elif vocab_file is not None:
            _lib.check(self._lib.cs_tokenizer_create_from_file(vocab_file.encode(), int(lowercase), max_length,
                                                               C.byref(h)))
        else:
            toks = sorted(vocab, key=vocab.get)
            if [vocab[t] for t in toks] != list(range(len(toks))):
                raise ValueError("vocabulary ids must be 0..n-1 without gaps (vocab.txt line numbers)")
            blob = "\n".join(toks).encode("utf-8") + b"\n"
            _lib.check(self._lib.cs_tokenizer_create(blob, len(blob), int(lowercase), max_length, C.byref(h)))
        self._h = h
        if not max_length:  # the handle's own truncation length (from the json / directory)
            max_length = int(self._lib.cs_tokenizer_max_length(h))
        self.max_length = max_length
        self.pad_id = int(self._lib.cs_tokenizer_pad_id(h))  # [PAD], or <pad> of a unigram tokenizer.json

    @classmethod
    def from_vocab_file(cls, path: str, **kw) -> "WordPieceTokenizer":
        return cls(vocab_file=path, **kw)

    @classmethod
    def from_tokenizer_json(cls, path: str, max_length: int = 0) -> "WordPieceTokenizer":
        return cls(tokenizer_json=path, max_length=max_length)

    @classmethod
    def from_dir(cls, model_dir: str, max_length: int = 0) -> "WordPieceTokenizer":
        return cls(model_dir=model_dir, max_length=max_length)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cs_tokenizer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def vocab_size(self) -> int:
        return int(self._lib.cs_tokenizer_vocab_size(self._h))

    def token_to_id(self, token: str) -> int:
        return int(self._lib.cs_tokenizer_token_to_id(self._h, token.encode("utf-8")))

    def encode_batch(self, texts: Sequence[str], max_length: int = None) -> Tuple[np.ndarray, np.ndarray]:
        """-> (ids [n, L], mask [n, L]) int32, padded to the batch's longest sequence."""
        blob, offsets = pack_texts(texts)
        n = len(texts)
        ml = int(max_length or 0)
        L = C.c_uint32()
        op = offsets.ctypes.data_as(_lib.u64p)
        stride = ml or self.max_length  # one pass: rows at the truncation length, then cut to the longest
        ids = np.empty((n, stride), np.int32)
        mask = np.empty((n, stride), np.int32)
        _lib.check(self._lib.cs_tokenizer_encode_batch(self._h, blob, op, n, ml, ids.ctypes.data_as(_lib.i32p),
                                                       mask.ctypes.data_as(_lib.i32p), stride, C.byref(L)))
        ids, mask = np.ascontiguousarray(ids[:, :L.value]), np.ascontiguousarray(mask[:, :L.value])
        return ids, mask

    def encode(self, text: str) -> List[int]:
        return self.encode_batch([text])[0][0].tolist()


BertWordPieceTokenizer = WordPieceTokenizer  # the name the earlier pure-Python class had
