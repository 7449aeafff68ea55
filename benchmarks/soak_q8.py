#!/usr/bin/env python3
"""Soak of the dynamic-quantisation mode: random batch shapes, direct calls and queued unit mixes, against the oracle's
quantised forward with the flip-noise bars of tests/test_gpu_quantized.py.  usage: soak_q8.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from codesearch_amd import FastEmbedder, ModelType
from codesearch_amd.bert_params import BertConfig, POOL_CLS, POOL_MEAN, quantize_linear_weights, synth_params, synth_token_batch
from tests.oracle_lib import load_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
o = load_oracle()
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
t_end = time.time() + budget
cases = worst = 0
while time.time() < t_end:
    pooling = int(rng.choice([POOL_CLS, POOL_MEAN]))
    cfg = BertConfig(vocab_size=700, hidden=384, layers=int(rng.integers(1, 4)), heads=12, intermediate=1536, max_position=260, pooling=pooling)
    params, ws = quantize_linear_weights(cfg, synth_params(cfg, int(rng.integers(1, 10_000))), per_channel=bool(rng.integers(0, 2)),
                                         unsigned=bool(rng.integers(0, 2)))
    emb = FastEmbedder(ModelType.AllMiniLML6V2Q, config=cfg, params=params, wscale=ws)
    for _ in range(4):
        if rng.integers(0, 2):   # one call
            n, L = int(rng.integers(1, 70)), int(rng.choice([1, 2, 5, 31, 64, 100, 129, 256]))
            ids, mask = synth_token_batch(cfg, int(rng.integers(1, 10_000)), n, L, L > 2)
            got = [(emb.embed_ids(ids, mask, batch_size=n), ids, mask)]
        else:                    # queued unit mix
            subs = []
            for _ in range(int(rng.integers(2, 8))):
                L = int(rng.choice([1, 4, 19, 77, 200]))
                subs.append(synth_token_batch(cfg, int(rng.integers(1, 10_000)), int(rng.integers(1, 20)), L, L > 2))
            ts = [emb.submit_ids(i, m) for i, m in subs]
            got = [(emb.wait(t), i, m) for t, (i, m) in zip(ts, subs)]
        for g, ids, mask in got:
            want = o.bert_forward(cfg, params, ids, mask, wscale=ws)["pooled"]
            e = np.abs(g - want)
            assert np.isfinite(g).all() and e.max() < 8e-3 and e.mean() < 6e-4, (cfg.layers, ids.shape, e.max(), e.mean())
            worst = max(worst, float(e.max()))
            cases += 1
    emb.close()
print(f"soak_q8: {cases} calls ok, worst max |gpu - oracle| {worst:.2e}")
