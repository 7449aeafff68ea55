#!/usr/bin/env python3
"""One short query through the quantised default model N times, for a rocprofv3 --kernel-trace run; --report DIR prints the last
forward's kernels (benchmarks/small_forward_trace.py's report)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.small_forward_trace import report  # noqa: E402


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    from codesearch_amd import FastEmbedder, ModelType
    from codesearch_amd.bert_params import POOL_MEAN, BertConfig, quantize_linear_weights, synth_params, synth_token_batch

    cfg = BertConfig(vocab_size=30522, hidden=384, layers=6, heads=12, intermediate=1536, max_position=512, pooling=POOL_MEAN)
    params, wscale = quantize_linear_weights(cfg, synth_params(cfg, 41), per_channel=False, unsigned=True)
    emb = FastEmbedder(ModelType.AllMiniLML6V2Q, config=cfg, params=params, wscale=wscale)
    ids, mask = synth_token_batch(cfg, 5, int(os.environ.get("B", 1)), int(os.environ.get("L", 16)), False)
    for _ in range(int(os.environ.get("REPS", 30))):
        emb.embed_ids(ids, mask)


if __name__ == "__main__":
    main()
