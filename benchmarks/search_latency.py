#!/usr/bin/env python3
"""Search latency through the host-buffer API (cs_index_search: H2D query, kernels, D2H results, one sync)
at the corpus sizes the reference actually sees (its own benchmark indexes 592 chunks; its search path issues
<= 9 query variants with limit up to 200, src/search/mod.rs:494-511)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codesearch_amd import VectorStore  # noqa: E402
from codesearch_amd.synth import synth_rows  # noqa: E402


def timed(st, q, k, reps=300):
    for _ in range(20):
        st.search_raw(q, k)
    t0 = time.perf_counter()
    for _ in range(reps):
        st.search_raw(q, k)
    return (time.perf_counter() - t0) / reps * 1e6


for n in (592, 10_000, 100_000, 1_000_000):
    st = VectorStore(None, 384, device=0)
    st.insert_synthetic(n, 1234, 0)
    st.build_index()
    q1, q9 = synth_rows(99, 0, 1, 384), synth_rows(99, 0, 9, 384)
    # the HIP runtime stalls once for ~36 ms a few hundred calls into a process (seen at call ~345 of a plain
    # loop of searches): get past it before timing
    for q, k in ((q1, 10), (q9, 10), (q9, 200)) * 3:
        timed(st, q, k, reps=60)
    print(f"rows {n}: 1 query k=10 {timed(st, q1, 10):.1f} us; 9 queries k=10 {timed(st, q9, 10):.1f} us; "
          f"9 queries k=200 {timed(st, q9, 200):.1f} us", flush=True)
