"""Generates tests/golden/scan_golden.json — known answers for the cosine scan / top-k.

Run in the build container:  python tests/golden/make_scan_golden.py

Sources of truth, independent of both the HIP path and the C oracle:
  * the reference's own known-answer cases (transcribed as DATA: vectors + expected
    ordering), /root/reference/src/vectordb/store.rs:846-893 (4-d search case) and
    /root/reference/src/embed/batch.rs:326-340 (3-d cosine cases);
  * numpy float64 exhaustive cosine + lexsort (cos desc, id asc) on seeded corpora from
    the integer generator of include/cs_synth.h (numpy mirror codesearch_amd/synth.py).
The file also pins sha256 digests of generated matrices so the C, HIP and numpy
generators can be proven identical.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from codesearch_amd.synth import synth_planted, synth_rows  # noqa: E402


def exact_topk(corpus, q, k):
    c64 = corpus.astype(np.float64)
    q64 = q.astype(np.float64)
    dots = c64 @ q64
    norms = np.sqrt((c64 * c64).sum(1)) * np.sqrt((q64 * q64).sum())
    with np.errstate(invalid="ignore", divide="ignore"):
        cos = np.where(norms == 0, 0.0, dots / norms)
    order = np.lexsort((np.arange(len(cos)), -cos))
    top = order[:k]
    gap = float(cos[order[k - 1]] - cos[order[k]]) if len(cos) > k else None
    return top.astype(int).tolist(), cos[top].tolist(), gap


def main():
    out = {"generator": "include/cs_synth.h", "cases": [], "digests": []}

    # --- reference known answers (data only) ---------------------------------------
    rows = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    q = np.array([0.9, 0.1, 0.0, 0.0], np.float32)
    ids, cos, _ = exact_topk(rows, q, 2)
    out["reference_kat"] = {
        "store_rs_846_893": {"rows": rows.tolist(), "query": q.tolist(), "k": 2,
                             "expect_ids": ids, "expect_cos": cos,
                             "asserted_by_reference": "results[0] is row 0; score[0] > score[1]"},
        "batch_rs_326_340": [
            {"a": [1, 0, 0], "b": [1, 0, 0], "expect": 1.0, "tol": 1e-3},
            {"a": [1, 0, 0], "b": [0, 1, 0], "expect": 0.0, "tol": 1e-3},
            {"a": [1, 1, 0], "b": [1, 0, 0], "lo": 0.7, "hi": 0.72},
        ],
    }

    # --- seeded corpora ----------------------------------------------------------------
    for (n, dim, seed, nq) in [(1000, 384, 0xC0DE5EA, 4), (100000, 384, 0xC0DE5EA, 4),
                               (3000, 768, 77, 2), (2000, 1024, 78, 2), (500, 100, 79, 2)]:
        corpus = synth_rows(seed, 0, n, dim)
        out["digests"].append({"seed": seed, "first_row": 0, "n": n, "dim": dim,
                               "sha256": hashlib.sha256(corpus.tobytes()).hexdigest()})
        queries = synth_rows(seed + 1, 0, nq, dim)
        planted_rows = [(7 * (i + 1) * n) // 31 % n for i in range(nq)]
        planted = synth_planted(seed, seed + 2, planted_rows, dim)
        for qi in range(nq):
            for k in (1, 10, 25, 200):
                if k >= n:
                    continue
                ids, cos, gap = exact_topk(corpus, queries[qi], k)
                out["cases"].append({"n": n, "dim": dim, "seed": seed, "query_seed": seed + 1,
                                     "qi": qi, "kind": "random", "k": k, "ids": ids,
                                     "cos": cos, "gap_k_k1": gap})
            ids, cos, gap = exact_topk(corpus, planted[qi], 10)
            out["cases"].append({"n": n, "dim": dim, "seed": seed, "query_seed": seed + 2,
                                 "qi": qi, "kind": "planted", "planted_row": planted_rows[qi],
                                 "k": 10, "ids": ids, "cos": cos, "gap_k_k1": gap})
    # a slice that does not start at row 0 (shards regenerate their own range)
    part = synth_rows(0xC0DE5EA, 5_000_000, 64, 384)
    out["digests"].append({"seed": 0xC0DE5EA, "first_row": 5_000_000, "n": 64, "dim": 384,
                           "sha256": hashlib.sha256(part.tobytes()).hexdigest()})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scan_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
