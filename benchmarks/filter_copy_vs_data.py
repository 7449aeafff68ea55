#!/usr/bin/env python3
"""How the two filter copies cope with rows that are NOT isotropic noise.  Real sentence embeddings share a large common
component (mean pairwise cosine 0.4-0.8 for the BGE family) and have a few coordinates much larger than the rest; both
widen the int8 copy's error band (its scale is the tile's largest coordinate) and crowd scores around the k-th best.
Corpus models, 2M x 384 rows generated on the GPU:
  iso       unit Gaussian rows (the benchmark's synthetic corpus)
  shared    rows = normalise(c * mu + g): a common direction mu carrying cos^2 = c^2 / (1 + c^2) of every row
  outlier   `shared` plus three coordinates scaled by 8 (outlier dimensions)
  clustered 2,000 cluster centres, rows = normalise(centre + 0.35 g): dense neighbourhoods around every query
For each: batched searches through the int8 copy and through the f16 copy (CS_FILTER_INT8=0): ms per search, overflowed
searches, results compared bit for bit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from codesearch_amd import VectorStore  # noqa: E402

N, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000), 384
KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["iso", "shared", "outlier", "clustered"]
dev = "cuda:0"
g = torch.Generator(device=dev)
g.manual_seed(1234)


def rows_of(kind, n):
    x = torch.randn((n, D), device=dev, generator=g)
    if kind in ("shared", "outlier"):
        mu = torch.nn.functional.normalize(torch.randn((1, D), device=dev, generator=g), dim=1)
        x = x / np.sqrt(D) + 1.2 * mu          # cos to mu ~ 0.77: mean pairwise cosine ~ 0.59
    if kind == "outlier":
        x[:, [7, 100, 333]] *= 8.0
    if kind == "clustered":
        centres = torch.nn.functional.normalize(torch.randn((2000, D), device=dev, generator=g), dim=1)
        idx = torch.randint(0, 2000, (n,), device=dev, generator=g)
        x = centres[idx] + 0.35 * x / np.sqrt(D)
    return torch.nn.functional.normalize(x, dim=1).contiguous()


def run(kind, int8):
    os.environ["CS_FILTER_INT8"] = "1" if int8 else "0"
    os.environ["CS_FILTER_SINGLE_MIN_K"] = "0"
    g.manual_seed(99)
    st = VectorStore(None, D, capacity=N)
    x = rows_of(kind, N)
    st.insert_device(x.data_ptr(), N)
    torch.cuda.synchronize()
    if int8:  # the quantiser's scale statistic: largest unit coordinate per 128-row tile, in units of 1 / sqrt(dim)
        tmax = x[: N // 128 * 128].abs().reshape(-1, 128 * D).max(dim=1).values * np.sqrt(D)
        print(f"{kind}: tile max |u| sqrt(dim): median {tmax.median().item():.2f}, p99 {tmax.quantile(0.99).item():.2f}; "
              f"mean pairwise cosine {float((x[:2000] @ x[2000:4000].T).mean()):.3f}", flush=True)
    st.build_index()
    qs = rows_of(kind, 64)
    if kind == "clustered":
        qs = x[torch.arange(0, 64 * 1000, 1000, device=dev)] + 0.02 * torch.randn((64, D), device=dev, generator=g)
    qh = qs.cpu().numpy()
    out = {}
    for nq, k in ((8, 10), (9, 200), (64, 10)):
        q = np.ascontiguousarray(qh[:nq])
        st.search_raw(q, k)
        b0, f0 = st.debug_counters()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            cos, ids, cnt = st.search_raw(q, k)
        ms = (time.perf_counter() - t0) / reps * 1e3
        b1, f1 = st.debug_counters()
        out[(nq, k)] = (ms, f1 - f0, ids.copy(), cos.copy())
    st.close()
    del x
    return out


for kind in KINDS:
    a = run(kind, True)
    b = run(kind, False)
    for key in a:
        same = bool((a[key][2] == b[key][2]).all() and (a[key][3] == b[key][3]).all())
        print(f"{kind:9s} nq={key[0]:3d} k={key[1]:3d}  int8 {a[key][0]:7.3f} ms ({a[key][1]} of 20 searches overflowed)   "
              f"f16 {b[key][0]:7.3f} ms ({b[key][1]} overflowed)   identical {same}", flush=True)
