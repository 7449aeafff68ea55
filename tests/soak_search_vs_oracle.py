#!/usr/bin/env python3
"""Soak: seeded random (rows, queries, k, tombstones) searches through the C ABI against the CPU oracle, plus random variant
merges against the host statement of search::search.  Not a test (minutes); prints the first disagreement and exits 1."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codesearch_amd import VectorStore  # noqa: E402
from codesearch_amd.search import merge_variant_results  # noqa: E402
from codesearch_amd.synth import synth_rows  # noqa: E402
from tests.oracle_lib import load_oracle  # noqa: E402

oracle = load_oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(7)
dim = 384
bad = 0
for case in range(cases):
    n = int(rng.choice([1, 2, 17, 64, 65, 500, 592, 1024, 1025, 2047, 4096, 4097, 9000, 30000]))
    nq = int(rng.choice([1, 1, 2, 4, 5, 9, 12]))
    k = int(rng.choice([1, 2, 10, 16, 17, 64, 65, 100, 200, 256, 300]))
    rows = synth_rows(10_000 + case, 0, n, dim)
    if n > 10:
        rows[rng.integers(0, n)] = rows[rng.integers(0, n)]  # an exact tie somewhere
        rows[rng.integers(0, n)] = 0.0
    qs = synth_rows(20_000 + case, 0, nq, dim)
    if n > 3:
        qs[0] = rows[3]
    st = VectorStore(None, dim)
    st.insert_embeddings(rows)
    dead = None
    if n > 20 and case % 3 == 0:
        ids = sorted(set(int(x) for x in rng.integers(0, n, size=max(1, n // 7))))
        st.delete_chunks(ids)
        dead = np.zeros((n + 31) // 32, np.uint32)
        for d in ids:
            dead[d >> 5] |= np.uint32(1 << (d & 31))
    st.build_index()
    cos, idx, cnt = st.search_raw(qs if nq > 1 else qs[0], k)
    for j in range(nq):
        ecos, eids = oracle.scan_topk(rows, qs[j], k, dead=dead, mode="omp")
        got_i, got_c = idx[j][: cnt[j]].tolist(), cos[j][: cnt[j]]
        ok = len(got_i) == len(eids) and np.allclose(got_c, ecos, atol=2e-6)
        if ok and got_i != eids.tolist():  # only float-order ties may differ: same cosine at the differing places
            ok = all(a == b or abs(float(ca) - float(cb)) < 1e-6 for a, b, ca, cb in zip(got_i, eids.tolist(), got_c, ecos)) \
                and len(set(got_i)) == len(got_i)
        if not ok:
            print("MISMATCH search", dict(case=case, n=n, nq=nq, k=k, j=j), got_i[:8], eids[:8].tolist())
            bad += 1
    if nq > 1 and nq <= 16:
        per = st.search_batch(qs, k)
        want = merge_variant_results(per, k)
        got, _ = st.search_variants(qs, k)
        if [r.score for r in got] != [r.score for r in want] or sorted(r.id for r in got) != sorted(r.id for r in want):
            print("MISMATCH variants", dict(case=case, n=n, nq=nq, k=k))
            bad += 1
    st.close()
    if case % 50 == 49:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print("done:", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
