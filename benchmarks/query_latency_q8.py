#!/usr/bin/env python3
"""Latency of one short query through the reference's DEFAULT model (AllMiniLML6V2Q shape: 6 layers, dynamically quantised —
/root/reference/src/embed/embedder.rs:12-13) next to the f32 graph of the same weights (CS_ENCODER_QUANT=0 in a second run)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from codesearch_amd import FastEmbedder, ModelType  # noqa: E402
from codesearch_amd.bert_params import POOL_MEAN, BertConfig, quantize_linear_weights, synth_params, synth_token_batch  # noqa: E402

cfg = BertConfig(vocab_size=30522, hidden=384, layers=6, heads=12, intermediate=1536, max_position=512, pooling=POOL_MEAN)
params, wscale = quantize_linear_weights(cfg, synth_params(cfg, 41), per_channel=False, unsigned=True)
emb = FastEmbedder(ModelType.AllMiniLML6V2Q, config=cfg, params=params, wscale=wscale)
print("mode", emb.gemm_mode())
for B, L in ((1, 16), (1, 64), (9, 16)):
    ids, mask = synth_token_batch(cfg, 5, B, L, False)
    for _ in range(20):
        emb.embed_ids(ids, mask)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        emb.embed_ids(ids, mask)
    t = (time.perf_counter() - t0) / reps
    fwd_ms, n = emb.profile_read()
    print(f"B={B} L={L}: embed {t * 1e6:.0f} us (device {fwd_ms / max(n, 1) * 1e3:.0f} us)")
