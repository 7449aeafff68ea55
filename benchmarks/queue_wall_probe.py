#!/usr/bin/env python3
"""Where the wall time of eight queued 32-chunk calls goes (submit / first wait / other waits), BGE-small shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from codesearch_amd import BertConfig, FastEmbedder, ModelType
from codesearch_amd.bert_params import synth_token_batch

cfg = BertConfig.bge_small()
emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202)
ids, mask = synth_token_batch(cfg, 999, 256, 256, False)
emb.embed_ids(ids, mask)
for rep in range(4):
    t0 = time.perf_counter()
    ts = [emb.submit_ids(ids[lo:lo + 32], mask[lo:lo + 32]) for lo in range(0, 256, 32)]
    t1 = time.perf_counter()
    emb.wait(ts[0])
    t2 = time.perf_counter()
    for t in ts[1:]:
        emb.wait(t)
    t3 = time.perf_counter()
    emb.profile_read(reset=True)
    print(f"submit x8 {1e3*(t1-t0):.2f} ms, first wait {1e3*(t2-t1):.2f} ms, other seven waits {1e3*(t3-t2):.2f} ms")
t0 = time.perf_counter()
for _ in range(5):
    emb.embed_ids(ids, mask)
print(f"embed_ids 256 rows: {1e3*(time.perf_counter()-t0)/5:.2f} ms")
