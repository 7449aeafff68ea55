#!/usr/bin/env python3
"""One short query through the encoder N times (B sequences of L tokens, BGE-small shape) — the workload of
benchmarks/query_latency.py's first lines — for a rocprofv3 --kernel-trace run; with --report DIR prints, from the trace of
the LAST forward, every kernel's duration and the gap in front of it."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(d):
    f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last forward: the kernels between the last two pooling kernels
    pools = [i for i, r in enumerate(rows) if "pool_normalize" in r["Kernel_Name"]]
    fwd = [r for r in rows[pools[-2] + 1:pools[-1] + 1] if "copyBuffer" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]]
    t0, prev_end = int(fwd[0]["Start_Timestamp"]), None
    tot_k = tot_gap = 0.0
    for r in fwd:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        tot_k += (e - s) / 1e3
        tot_gap += gap
        print(f'{r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]:60s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.2f}  gap {gap:5.2f}  grid {r.get("Grid_Size", r.get("Grid_Size_X", ""))}')
        prev_end = e
    print(f"# {len(fwd)} kernels, sum of durations {tot_k:.1f} us, sum of gaps {tot_gap:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    B = int(os.environ.get("B", 1))
    L = int(os.environ.get("L", 16))
    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import synth_token_batch

    cfg = BertConfig.bge_small()
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202)
    ids, mask = synth_token_batch(cfg, 5, B, L, False)
    for _ in range(int(os.environ.get("REPS", 30))):
        emb.embed_ids(ids, mask)


if __name__ == "__main__":
    main()
