#!/usr/bin/env python3
"""Soak for the int8 filter copy: seeded random corpora (width, size, data shape, tombstones, appends between builds, k,
query count) searched in batches and compared BIT FOR BIT with the same queries searched one at a time on the exact f32
streaming scan.  Not a test (minutes); prints the first disagreement and exits 1."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CS_FILTER_SINGLE_MIN_K"] = "0"   # single queries stay on the streaming scan: the yardstick
from codesearch_amd import VectorStore  # noqa: E402
from codesearch_amd.synth import synth_rows  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2026)
bad = 0
copies = {0: 0, 1: 0, 2: 0}
reruns = 0
for case in range(cases):
    dim = int(rng.choice([384, 384, 384, 768, 1024]))
    n_total = int(rng.choice([1100, 1152, 3000, 9999, 40_000, 130_000, 400_003]))
    if dim > 384:
        n_total = min(n_total, 130_000)
    nq = int(rng.choice([2, 3, 8, 9, 31, 33, 64, 65, 128, 200]))
    k = int(rng.choice([1, 10, 10, 47, 48, 100, 200, 256]))
    shape = rng.choice(["iso", "shared", "scaled", "outlier", "dups"])
    rows = synth_rows(50_000 + case, 0, n_total, dim).copy()
    if shape == "shared":
        mu = synth_rows(60_000 + case, 0, 1, dim)[0]
        rows += np.float32(1.5 * np.linalg.norm(rows[0]) / np.linalg.norm(mu)) * mu
    elif shape == "scaled":
        rows *= rng.choice([1e-8, 1e-3, 1.0, 1e4], size=(n_total, 1)).astype(np.float32)
    elif shape == "outlier":
        rows[:, rng.integers(0, dim, 2)] *= np.float32(6.0)
    elif shape == "dups":
        a = int(rng.integers(0, n_total - 600))
        rows[a:a + 600] = rows[a] + rng.normal(0, 0.02 * np.abs(rows[a]).mean(), (600, dim)).astype(np.float32)
    qs = synth_rows(70_000 + case, 0, nq, dim).copy()
    if shape == "shared":
        qs += np.float32(1.5 * np.linalg.norm(qs[0]) / np.linalg.norm(mu)) * mu
    qs[0] = rows[n_total // 3]
    qs[1] = rows[n_total - 1] * np.float32(0.5)
    st = VectorStore(None, dim)
    parts = sorted(set([n_total] + [int(x) for x in rng.integers(1, n_total, size=int(rng.integers(0, 3)))]))
    lo = 0
    for hi in parts:   # appends between builds: tiles fill up over several builds
        st.insert_embeddings(rows[lo:hi])
        st.build_index()
        lo = hi
    if case % 3 == 0:
        st.delete_chunks(sorted(set(int(x) for x in rng.integers(0, n_total, size=max(1, n_total // 50)))))
        st.build_index()
    cos, idx, cnt = st.search_raw(qs, k)
    for j in (range(nq) if nq <= 16 else list(range(0, nq, 7)) + [nq - 1]):
        c1, i1, n1 = st.search_raw(qs[j], k)
        if cnt[j] != n1[0] or idx[j].tolist() != i1[0].tolist() or cos[j].tobytes() != c1[0].tobytes():
            print("MISMATCH", dict(case=case, dim=dim, n=n_total, nq=nq, k=k, shape=str(shape), j=j), idx[j][:6], i1[0][:6])
            bad += 1
            break
    c, _, r = st.filter_state()
    copies[c] += 1
    reruns += r
    st.close()
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} mismatches; filter copy at the end of a case: int8 {copies[2]}, f16 {copies[1]}, none {copies[0]}; "
              f"int8 searches answered by the f16 copy: {reruns}", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
