"""End-to-end slice of the two callers of the hot path, on token ids:

  index  (src/index/mod.rs:626-762, :878): per file embed_chunks -> insert_chunks_with_ids,
         then one build_index()  ==>  embed mini-batches on the GPU, append the embeddings to
         the device-resident matrix without leaving HBM, build.
  search (src/search/mod.rs:483-511): embed_queries_batch -> store.search per variant
         ==>  embed the query batch, ONE batched search.

BASELINE.json configs[3]: 100k synthetic chunks embedded on-GPU, then 64 batched queries,
top-10.  Device buffers are torch tensors (plumbing only); all compute is libcsgpu.
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
