"""End-to-end slice of the two callers of the hot path, on token ids:

  index  (src/index/mod.rs:626-762, :878): per file embed_chunks -> insert_chunks_with_ids,
         then one build_index()  ==>  embed mini-batches on the GPU, append the embeddings to
         the device-resident matrix without leaving HBM, build.
  search (src/search/mod.rs:483-511): embed_queries_batch -> store.search per variant
         ==>  embed the query batch, ONE batched search.

BASELINE.json configs[3]: 100k synthetic chunks embedded on-GPU, then 64 batched queries,
top-10.  Device buffers are torch tensors (plumbing only); all compute is libcsgpu.

The *_text_* variants start from strings, as the reference's callers do (embed_batch(Vec<String>),
src/embed/embedder.rs:249): cs_embedder_embed_texts tokenises mini-batch i+1 on host threads while
the device runs mini-batch i.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Dict

import numpy as np

from . import _lib
from .embedder import FastEmbedder
from .vector_store import VectorStore


def index_token_chunks(embedder: FastEmbedder, store: VectorStore, ids: np.ndarray, mask: np.ndarray,
                       batch_size: int = 0) -> Dict[str, float]:
    """Embed [n, L] token chunks and append them to `store` (embeddings never visit the host)."""
    import torch

    lib = _lib.load()
    n = ids.shape[0]
    dim = embedder.dimensions()
    if not store.sharded:  # E8 in place: the pooling kernel's stores land in the corpus rows themselves
        t0 = time.perf_counter()
        embedder.embed_ids_to_device(ids, mask, store.reserve_rows(n), batch_size)
        t1 = time.perf_counter()
        store.commit_rows(n)
        store.build_index()
        return {"embed_s": t1 - t0, "insert_build_s": time.perf_counter() - t1}
    dev = int(lib.cs_index_device(store.handle))
    buf = torch.empty((n, dim), dtype=torch.float32, device=f"cuda:{dev}")
    t0 = time.perf_counter()
    embedder.embed_ids_to_device(ids, mask, buf.data_ptr(), batch_size)
    t1 = time.perf_counter()
    _lib.check(lib.cs_index_add_device(store.handle, C.c_void_p(buf.data_ptr()), n, dim, None, None))
    store.build_index()
    t2 = time.perf_counter()
    del buf
    return {"embed_s": t1 - t0, "insert_build_s": t2 - t1}


def search_token_queries(embedder: FastEmbedder, store: VectorStore, q_ids: np.ndarray, q_mask: np.ndarray,
                         k: int):
    """Embed the query batch and run one batched search -> (cos, ids, counts, timings)."""
    t0 = time.perf_counter()
    q = embedder.embed_ids(q_ids, q_mask)
    t1 = time.perf_counter()
    cos, ids, counts = store.search_raw(q, k)
    t2 = time.perf_counter()
    return cos, ids, counts, {"embed_queries_s": t1 - t0, "search_s": t2 - t1}


def synth_vocab(size: int = 30522):
    """A vocab.txt-shaped vocabulary of `size` entries for benchmarks and tests (the real
    bge-small vocab.txt is not reachable offline): BERT's special-token ids, ASCII pieces, then
    generated lower-case words and "##" continuations.  -> {token: id}"""
    import itertools
    import string

    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += list(string.ascii_lowercase) + list(string.digits) + list(string.punctuation)
    toks += ["##" + c for c in string.ascii_lowercase + string.digits]
    seen = set(toks)
    cons, vow = "bcdfghjklmnprstvwz", "aeiou"
    for n in (2, 3, 4):
        for parts in itertools.product(*[(cons, vow)[i % 2] for i in range(n * 2 - (n > 2))]):
            w = "".join(parts)
            for t in (w, "##" + w):
                if t not in seen and len(toks) < size:
                    seen.add(t)
                    toks.append(t)
            if len(toks) >= size:
                break
        if len(toks) >= size:
            break
    assert len(toks) == size, len(toks)
    return {t: i for i, t in enumerate(toks)}


def synth_code_texts(vocab, n: int, seed: int, mean_words: int = 150):
    """n code-like chunks built from the vocabulary's words, identifiers (word_word, wordWord) and
    punctuation; ~1.5 tokens per word, so mean_words=150 gives ~230-token chunks."""
    rng = np.random.default_rng(seed)
    words = [t for t in vocab if t.isalpha() and len(t) >= 2]
    punct = list("(){}[];:,.=+-*/<>!&|")
    out = []
    for _ in range(n):
        k = int(rng.integers(mean_words // 2, mean_words * 3 // 2))
        idx = rng.integers(0, len(words), size=2 * k)
        kind = rng.integers(0, 10, size=k)
        parts = []
        for j in range(k):
            a, b = words[idx[2 * j]], words[idx[2 * j + 1]]
            c = kind[j]
            if c < 4:
                parts.append(a)
            elif c < 6:
                parts.append(a + "_" + b)
            elif c < 7:
                parts.append(a + b.capitalize())
            elif c < 9:
                parts.append(a + punct[idx[2 * j + 1] % len(punct)])
            else:
                parts.append("\n    " + a)
        out.append(" ".join(parts))
    return out


def index_text_chunks(embedder: FastEmbedder, store: VectorStore, texts, batch_size: int = 0) -> Dict[str, float]:
    """Tokenise + embed strings and append the embeddings to `store` without leaving HBM."""
    import torch

    lib = _lib.load()
    n, dim = len(texts), embedder.dimensions()
    if not store.sharded:  # E8 in place (the length-grouped mini-batches scatter their rows into the corpus in input order)
        t0 = time.perf_counter()
        embedder.embed_texts_to_device(texts, store.reserve_rows(n), batch_size)
        t1 = time.perf_counter()
        store.commit_rows(n)
        store.build_index()
        return {"embed_s": t1 - t0, "insert_build_s": time.perf_counter() - t1}
    dev = int(lib.cs_index_device(store.handle))
    buf = torch.empty((n, dim), dtype=torch.float32, device=f"cuda:{dev}")
    t0 = time.perf_counter()
    embedder.embed_texts_to_device(texts, buf.data_ptr(), batch_size)
    t1 = time.perf_counter()
    _lib.check(lib.cs_index_add_device(store.handle, C.c_void_p(buf.data_ptr()), n, dim, None, None))
    store.build_index()
    t2 = time.perf_counter()
    del buf
    return {"embed_s": t1 - t0, "insert_build_s": t2 - t1}


def search_text_queries(embedder: FastEmbedder, store: VectorStore, queries, k: int):
    """embed_queries_batch + one batched search -> (cos, ids, counts, timings)."""
    t0 = time.perf_counter()
    q = np.stack(embedder.embed_batch(queries))
    t1 = time.perf_counter()
    cos, ids, counts = store.search_raw(q, k)
    t2 = time.perf_counter()
    return cos, ids, counts, {"embed_queries_s": t1 - t0, "search_s": t2 - t1}
