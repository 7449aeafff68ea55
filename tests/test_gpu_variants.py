"""The step right after the scan in `search::search` (/root/reference/src/search/mod.rs:508-611), on the
device: cs_index_search_variants / cs_merge_variants_device against the host statement of the same rule
(codesearch_amd/search.py::merge_variant_results + should_use_vector_only, itself a restatement of
mod.rs:513-611).  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

from codesearch_amd import _lib
from codesearch_amd.search import (EARLY_TERMINATION_TOP_N, HIGH_CONFIDENCE_THRESHOLD, merge_variant_results,
                                   should_use_vector_only)
from codesearch_amd.sharded import key_pack, key_unpack
from codesearch_amd.synth import synth_planted, synth_rows
from codesearch_amd.vector_store import SearchResult, cos_to_distance

pytestmark = pytest.mark.gpu


def vres(cid, cos):
    d = float(cos_to_distance(np.float32(cos)))
    return SearchResult(id=int(cid), score=float(np.float32(1.0) - np.float32(d)), distance=d, path="", content="",
                        start_line=0, end_line=0, kind="function", signature=None, docstring=None, context=None, hash="")


def host_merge(lists, limit):
    """lists: per variant [(id, cos), ...] best-first -> the reference's rule on the reference's score scale."""
    merged = merge_variant_results([[vres(i, c) for i, c in l] for l in lists], limit)
    return merged, should_use_vector_only(merged, False)


def device_merge(gpu_lib, lists, k, limit):
    import torch

    nv = len(lists)
    keys = np.zeros((nv, k), np.uint64)
    for v, l in enumerate(lists):
        if l:
            ids = np.array([i for i, _ in l], np.uint32)
            cos = np.array([c for _, c in l], np.float32)
            keys[v, : len(l)] = key_pack(cos, ids)
    dev = "cuda:0"
    d_in = torch.from_numpy(keys.view(np.int64)).to(dev)
    d_keys = torch.zeros(limit, dtype=torch.int64, device=dev)
    d_cos = torch.zeros(limit, dtype=torch.float32, device=dev)
    d_ids = torch.zeros(limit, dtype=torch.int32, device=dev)
    d_meta = torch.zeros(2, dtype=torch.int32, device=dev)
    vp = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
    _lib.check(gpu_lib.cs_merge_variants_device(0, vp(d_in), nv, k, limit, vp(d_keys), vp(d_cos), vp(d_ids), vp(d_meta),
                                                vp(d_meta, 4), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    cnt, flag = d_meta.cpu().numpy().tolist()
    kc, ki = key_unpack(d_keys.cpu().numpy().view(np.uint64))
    assert np.array_equal(kc[:cnt], d_cos.cpu().numpy()[:cnt]) and np.array_equal(ki[:cnt], d_ids.cpu().numpy().astype(np.uint32)[:cnt])
    return d_ids.cpu().numpy().astype(np.uint32)[:cnt], d_cos.cpu().numpy()[:cnt], bool(flag)


def assert_same(got_ids, got_cos, flag, merged, host_flag):
    assert len(got_ids) == len(merged)
    assert [vres(0, c).score for c in got_cos] == [r.score for r in merged]      # same scores, best first
    want = [r.id for r in merged]
    if got_ids.tolist() != want:  # equal scores: the reference's order is a HashMap's; ours is (cosine desc, id asc)
        assert sorted(got_ids.tolist()) == sorted(want)
        for a, b, r in zip(got_ids.tolist(), want, merged):
            if a != b:
                assert sum(1 for m in merged if m.score == r.score) > 1
    assert flag == host_flag


def test_reference_shaped_cases(gpu_lib):
    """Hand-sized cases in the style of the reference's own rank tests (rerank/mod.rs:273-337: ids 1..4,
    scores 0.9 / 0.8 / 0.7): duplicates across variants keep the best score, the limit cuts, empty variants
    and the early-termination predicate on both sides of its threshold."""
    c = lambda score: 2.0 * score - 1.0  # the cosine whose reference score (1 + cos) / 2 is `score`
    cases = [
        ([[(1, c(0.9)), (2, c(0.8)), (3, c(0.7))], [(2, c(0.95)), (1, c(0.6)), (4, c(0.5))]], 3, 3),
        ([[(1, c(0.9)), (2, c(0.8)), (3, c(0.7))], [(2, c(0.95)), (1, c(0.6)), (4, c(0.5))]], 3, 2),   # limit cuts
        ([[(7, c(0.99)), (8, c(0.98)), (9, c(0.97))], [(10, c(0.96)), (11, c(0.955)), (7, c(0.5))], []], 3, 5),
        ([[(7, c(0.99)), (8, c(0.98)), (9, c(0.97))], [(10, c(0.96)), (11, c(0.84)), (7, c(0.5))]], 3, 5),  # fifth at 0.16
        ([[(5, c(0.93))]], 1, 1),                                                                     # one confident hit
        ([[], []], 4, 4),                                                                             # nothing found
        ([[(0, 0.25), (1, 0.25), (2, -0.5)], [(1, 0.25), (0, 0.1), (0xFFFFFFFE, 0.0)]], 3, 3),        # ids 0 and 2^32-2, ties
    ]
    for lists, k, limit in cases:
        merged, hflag = host_merge(lists, limit)
        ids, cos, flag = device_merge(gpu_lib, lists, k, limit)
        assert_same(ids, cos, flag, merged, hflag)
    assert host_merge(cases[2][0], 5)[1] is True and host_merge(cases[3][0], 5)[1] is False
    assert host_merge(cases[4][0], 1)[1] is True and host_merge(cases[5][0], 4)[1] is False
    assert HIGH_CONFIDENCE_THRESHOLD == 0.15 and EARLY_TERMINATION_TOP_N == 5


@pytest.mark.parametrize("nv,k", [(9, 200), (1, 10), (16, 1024), (3, 25)])
def test_seeded_lists_against_host_merge(gpu_lib, nv, k):
    rng = np.random.default_rng(nv * 7919 + k)
    lists = []
    for v in range(nv):
        live = k if v % 3 else int(rng.integers(0, k + 1))
        ids = rng.choice(3 * k, size=live, replace=False)            # heavy overlap between variants
        cos = np.sort(rng.uniform(0.2, 0.999, size=live).astype(np.float32))[::-1]
        lists.append(list(zip(ids.tolist(), cos.tolist())))
    merged, hflag = host_merge(lists, k)
    ids, cos, flag = device_merge(gpu_lib, lists, k, k)
    assert_same(ids, cos, flag, merged, hflag)


def test_search_variants_end_to_end(gpu_lib, oracle):
    """cs_index_search_variants = nine per-variant searches (k = 200, the reference's retrieval limit) + the
    device merge, against per-variant searches merged on the host and against the oracle's scores."""
    from codesearch_amd import Chunk, EmbeddedChunk, VectorStore

    n, dim, k = 30_000, 384, 200
    st = VectorStore(None, dim)
    rows = oracle.synth_rows(4711, 0, n, dim)
    st.insert_chunks_with_ids([EmbeddedChunk(Chunk(f"fn f{i}() {{}}", i, i + 1, "Function", f"m{i % 13}.rs"), rows[i])
                               for i in range(n)])
    st.build_index()
    base = synth_planted(4711, 9, [12_345], dim)[0]
    noise = synth_rows(9, 0, 8, dim)
    variants = np.stack([base] + [base + 0.35 * np.linalg.norm(base) / np.linalg.norm(z) * z for z in noise]).astype(np.float32)
    per_variant = st.search_batch(variants, k)
    want = merge_variant_results(per_variant, k)
    got, flag = st.search_variants(variants, k)
    assert len(got) == len(want) == k
    assert [r.score for r in got] == [r.score for r in want]
    assert sorted(r.id for r in got) == sorted(r.id for r in want)
    assert got[0].id == 12_345 and got[0].path == "m8.rs"
    assert flag == should_use_vector_only(want, False)
    best = {}
    for v in variants:  # the oracle's view of "best cosine of a chunk over all variants"
        c, i = oracle.scan_topk(rows, v, k, mode="omp")
        for cc, ii in zip(c, i):
            best[int(ii)] = max(best.get(int(ii), -2.0), float(cc))
    top = sorted(best.items(), key=lambda t: (-t[1], t[0]))[:k]
    np.testing.assert_allclose([1.0 - 2.0 * r.distance for r in got], [c for _, c in top], atol=2e-6)
    sh = VectorStore(None, dim, devices=[0, 0, 0, 0], rows_per_stripe=4096)   # the same through four shards
    sh.insert_chunks_with_ids([EmbeddedChunk(Chunk(f"fn f{i}() {{}}", i, i + 1, "Function", f"m{i % 13}.rs"), rows[i])
                               for i in range(n)])
    sh.build_index()
    got4, flag4 = sh.search_variants(variants, k)
    assert [(r.id, r.score) for r in got4] == [(r.id, r.score) for r in got] and flag4 == flag
    sh.close()
    single, flag1 = st.search_variants(variants[0], 10)               # one variant: the plain search
    assert [r.id for r in single] == [r.id for r in st.search(variants[0], 10)] and flag1 is False
    from codesearch_amd import CsError
    with pytest.raises(CsError) as e:
        st.search_variants(np.zeros((17, dim), np.float32), 10)
    assert "at most 16 query variants" in str(e.value)
    st.close()
