#!/usr/bin/env python3
"""The reference's call shape on a quantised model (the default one): slices of 32 chunks x 256 tokens
(/root/reference/src/embed/batch.rs:70,94), one call at a time vs eight submitted and collected through the queue
(each slice stays its own quantisation unit inside the shared device batch)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np

    from codesearch_amd import FastEmbedder, ModelType
    from codesearch_amd.bert_params import quantize_linear_weights, synth_params, synth_token_batch

    mt = ModelType.AllMiniLML6V2Q
    cfg = mt.bert_config()
    params, wscale = quantize_linear_weights(cfg, synth_params(cfg, 202), per_channel=False, unsigned=True)
    emb = FastEmbedder(mt, config=cfg, params=params, wscale=wscale)
    # --shape CALLS,CHUNKS,TOKENS: other call shapes (default: the reference's, 8 calls of 32 chunks x 256 tokens)
    calls, per, L = 8, 32, 256
    for a in sys.argv[1:]:
        if a.startswith("--shape="):
            calls, per, L = (int(x) for x in a[len("--shape="):].split(","))
    total = calls * per
    ids, mask = synth_token_batch(cfg, 999, total, L, False)
    out = {"shape": {"calls": calls, "chunks_per_call": per, "tokens": L}}
    for mode in ("q8", "split"):
        emb.set_gemm_mode(mode)
        emb.embed_ids(ids[:per], mask[:per])
        t0 = time.perf_counter()
        for _ in range(10):
            for lo in range(0, total, per):
                emb.embed_ids(ids[lo:lo + per], mask[lo:lo + per])
        one = (time.perf_counter() - t0) / 10

        def queued():
            ts = [emb.submit_ids(ids[lo:lo + per], mask[lo:lo + per]) for lo in range(0, total, per)]
            return [emb.wait(t) for t in ts]

        queued()
        emb.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(10):
            queued()
        q = (time.perf_counter() - t0) / 10
        ms, n = emb.profile_read()
        out[mode] = {"one_call_at_a_time_chunks_per_s": total / one, "queued_chunks_per_s": total / q,
                     "device_ms_per_8_calls_queued": ms / 10, "device_batches_per_8_calls": n / 10}
        if "--stages" in sys.argv:  # per-kernel-class microseconds per layer of the shared device batch (8 units)
            emb.profile_stages(True)
            queued()
            emb.profile_stages_read(reset=True)
            for _ in range(3):
                queued()
            st, nf = emb.profile_stages_read()
            emb.profile_stages(False)
            out[mode]["queued_stages_us_per_layer"] = {
                k: round(v / (cfg.layers if k not in ("embed_ln", "pool_normalize") else 1), 1) for k, v in st.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
