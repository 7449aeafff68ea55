#!/usr/bin/env python3
"""Latency of the query side of `codesearch search`: embed a few short query variants (embed_one /
embed_queries_batch, /root/reference/src/embed/mod.rs:164-226; the reference reports ~3-4 ms per query on
its CPU path), then one batched top-k search.  Host-buffer APIs end to end."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from codesearch_amd import BertConfig, FastEmbedder, ModelType, VectorStore  # noqa: E402
from codesearch_amd.bert_params import synth_token_batch  # noqa: E402

cfg = BertConfig.bge_small()
emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202)
store = VectorStore(None, cfg.hidden)
store.insert_synthetic(100_000, 77, 0)
store.build_index()
# the HIP runtime stalls once for ~36 ms a few hundred calls into a process: get past it before timing
_q = np.zeros((1, cfg.hidden), np.float32); _q[0, 0] = 1.0
for _ in range(600):
    store.search_raw(_q, 10)
for B, L in ((1, 16), (1, 64), (9, 16), (9, 64), (32, 128)):
    ids, mask = synth_token_batch(cfg, 5, B, L, False)
    for _ in range(10):
        q = emb.embed_ids(ids, mask)
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps):
        q = emb.embed_ids(ids, mask)
    t_embed = (time.perf_counter() - t0) / reps
    for _ in range(10):
        store.search_raw(q, 10)
    t0 = time.perf_counter()
    for _ in range(reps):
        store.search_raw(q, 10)
    t_search = (time.perf_counter() - t0) / reps
    fwd_ms, n = emb.profile_read()
    print(f"B={B} L={L}: embed {t_embed * 1e6:.0f} us (device {fwd_ms / max(n, 1) * 1e3:.0f} us), "
          f"search top-10 over 100k rows {t_search * 1e6:.0f} us", flush=True)
