#!/usr/bin/env python3
"""Device time of a forward over B short queries (16 tokens each: a query and its variants, /root/reference/src/search/mod.rs:508-611),
B = 1..12, BGE-small shape — for A/Bs of the small path's forms (CS_SMALL_WIDE_MIN_ROWS, CS_SMALL_FUSE through the environment)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codesearch_amd import BertConfig, FastEmbedder, ModelType  # noqa: E402
from codesearch_amd.bert_params import synth_token_batch  # noqa: E402

cfg = BertConfig.bge_small()
emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202)
out = []
for B in (1, 2, 4, 6, 8, 9, 10, 12):
    ids, mask = synth_token_batch(cfg, 5, B, 16, False)
    for _ in range(10):
        emb.embed_ids(ids, mask)
    emb.profile_read(reset=True)
    for _ in range(60):
        emb.embed_ids(ids, mask)
    ms, n = emb.profile_read()
    out.append(f"{B}x16: {ms / max(n, 1) * 1e3:.0f}")
print("device us:", "  ".join(out))
