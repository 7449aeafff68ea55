#!/usr/bin/env python3
"""Where the wall time of MANY SMALL queued calls goes (a file of a few chunks is one call in the reference:
/root/reference/src/embed/batch.rs:84-115): submit / first wait (the flush: one device batch) / the other waits, for both
GEMM modes of a quantised MiniLM-L6.  usage: queue_small_calls_probe.py [calls] [chunks_per_call] [tokens]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from codesearch_amd import FastEmbedder, ModelType
from codesearch_amd.bert_params import quantize_linear_weights, synth_params, synth_token_batch

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 64
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = int(sys.argv[3]) if len(sys.argv) > 3 else 128
mt = ModelType.AllMiniLML6V2Q
cfg = mt.bert_config()
params, wscale = quantize_linear_weights(cfg, synth_params(cfg, 202), per_channel=False, unsigned=True)
emb = FastEmbedder(mt, config=cfg, params=params, wscale=wscale)
ids, mask = synth_token_batch(cfg, 999, calls * per, L, False)
gc.disable()  # (a generation-2 collection of the interpreter landed inside one repetition in four: 40 ms that are not the library's)
for mode in ("q8", "split"):
    emb.set_gemm_mode(mode)
    for rep in range(4):
        emb.profile_read(reset=True)
        t0 = time.perf_counter()
        ts = [emb.submit_ids(ids[lo:lo + per], mask[lo:lo + per]) for lo in range(0, calls * per, per)]
        t1 = time.perf_counter()
        emb.wait(ts[0])
        t2 = time.perf_counter()
        for t in ts[1:]:
            emb.wait(t)
        t3 = time.perf_counter()
        ms, n = emb.profile_read()
        print(f"{mode}: submit x{calls} {1e3*(t1-t0):.2f} ms, first wait {1e3*(t2-t1):.2f} ms (device {ms:.2f} ms in {n} batches), "
              f"other waits {1e3*(t3-t2):.2f} ms -> {calls*per/(t3-t0):.0f} chunks/s")
