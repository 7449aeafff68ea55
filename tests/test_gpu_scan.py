"""Parity tests proper: the HIP scan/top-k through the C ABI vs the CPU oracle, the
committed golden vectors and the reference's own known answers.  Needs an MI355X.

Bar (BASELINE.json north_star): top-k ids exact, cosine within 1e-4 (asserted far
tighter: 2e-6).  ids are compared exactly; a differing id is accepted only where the two
rows' float64 cosines differ by < 1e-6 (fp32 summation-order tie), SURVEY.md §7."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

from codesearch_amd.synth import synth_planted, synth_rows

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scan_golden.json")))
COS_TOL = 2e-6


@pytest.fixture(scope="module")
def VS(gpu_lib):
    from codesearch_amd import VectorStore

    assert gpu_lib.cs_device_count() >= 1, "no HIP device visible"
    return VectorStore


def assert_topk_equal(got_cos, got_ids, exp_cos, exp_ids, corpus=None, q=None, oracle=None):
    """ids exact; a differing id is accepted only as an fp32 summation-order tie: at every differing
    position the two rows' float64 cosines are within 1e-6, no id is returned twice, and an id present
    in only one of the lists (a tie across the k-th place) ties with the expected k-th cosine."""
    got_ids = list(map(int, got_ids))
    exp_ids = list(map(int, exp_ids))
    assert len(got_ids) == len(exp_ids)
    np.testing.assert_allclose(np.asarray(got_cos, np.float64), np.asarray(exp_cos, np.float64), atol=COS_TOL)
    if got_ids != exp_ids:
        assert corpus is not None, (got_ids, exp_ids)
        assert len(set(got_ids)) == len(got_ids), got_ids
        for a, b in zip(got_ids, exp_ids):
            if a != b:  # only an fp32-order tie may swap ids
                ca, cb = oracle.cosine_f64(q, corpus[a]), oracle.cosine_f64(q, corpus[b])
                assert abs(ca - cb) < 1e-6, (a, b, ca, cb)
        edge = oracle.cosine_f64(q, corpus[exp_ids[-1]])
        for x in set(got_ids) ^ set(exp_ids):
            cx = oracle.cosine_f64(q, corpus[x])
            assert abs(cx - edge) < 1e-6, (x, cx, edge)


# ---- the reference's own tests, re-expressed (store.rs:833-1028) --------------------------------

def test_reference_insert_and_search(VS, tmp_path):
    """store.rs:846-893 test_insert_and_search, 4-d vectors, k=2 (TempDir + test.db as there)."""
    from codesearch_amd import Chunk, EmbeddedChunk

    store = VS(tmp_path / "test.db", 4)
    assert store.dimensions == 4 and not store.is_indexed()  # store.rs:834-844
    chunks = [
        EmbeddedChunk(Chunk("fn authenticate() {}", 0, 1, "Function", "auth.rs"), [1.0, 0.0, 0.0, 0.0]),
        EmbeddedChunk(Chunk("fn calculate() {}", 2, 3, "Function", "math.rs"), [0.0, 1.0, 0.0, 0.0]),
    ]
    assert store.insert_chunks(chunks) == 2
    store.build_index()
    assert store.is_indexed()
    results = store.search([0.9, 0.1, 0.0, 0.0], 2)
    assert len(results) == 2
    assert "authenticate" in results[0].content
    assert results[0].score > results[1].score
    # numeric pin from the golden file (float64 exhaustive)
    kat = GOLDEN["reference_kat"]["store_rs_846_893"]
    cos = [1.0 - 2.0 * r.distance for r in results]
    np.testing.assert_allclose(cos, kat["expect_cos"], atol=COS_TOL)
    assert [r.id for r in results] == kat["expect_ids"]
    assert abs(results[0].score - (1 + kat["expect_cos"][0]) / 2) < 1e-6
    st = store.stats()
    assert st.total_chunks == 2 and st.total_files == 2 and st.indexed and st.dimensions == 4


def test_reference_error_texts(VS):
    from codesearch_amd import Chunk, CsError, EmbeddedChunk

    store = VS(None, 4)
    with pytest.raises(CsError) as e:  # store.rs:440-444
        store.search([1, 0, 0, 0], 1)
    assert str(e.value) == "Index not built. Call build_index() after inserting chunks."
    with pytest.raises(CsError) as e:  # store.rs:667-671
        store.insert_chunks_with_ids([EmbeddedChunk(Chunk("x", 0, 0, "Other", "a"), [1, 2, 3])])
    assert str(e.value) == "Embedding dimension mismatch: expected 4, got 3"
    with pytest.raises(CsError) as e:  # same text from the C ABI itself
        store.insert_embeddings(np.zeros((2, 3), np.float32))
    assert str(e.value) == "Embedding dimension mismatch: expected 4, got 3" and e.value.code == 2
    ids = store.insert_chunks_with_ids([EmbeddedChunk(Chunk("x", 0, 0, "Other", "a"), [1, 2, 3, 4]),
                                        EmbeddedChunk(Chunk("y", 0, 0, "Other", "b"), [4, 3, 2, 1])])
    assert ids == [0, 1] and store.next_id() == 2 and not store.is_indexed()
    store.build_index()
    with pytest.raises(CsError) as e:  # store.rs:432-438
        store.search([1, 0, 0], 1)
    assert str(e.value) == "Query embedding dimension mismatch: expected 4, got 3"
    # inserting after a build un-builds (store.rs:682); deleting too (store.rs:604-606)
    assert store.insert_chunks_with_ids([EmbeddedChunk(Chunk("z", 0, 0, "Other", "a"), [0, 0, 1, 0])]) == [2]
    assert not store.is_indexed()
    store.build_index()
    assert store.delete_chunks([1, 1, 99]) == 1 and not store.is_indexed()
    store.build_index()
    res = store.search([4, 3, 2, 1], 3)
    assert [r.id for r in res] == [0, 2] and len(store) == 2
    assert store.get_chunk(1) is None and store.get_chunk(0).content == "x"
    assert store.get_chunks_by_file() == {"a": [0, 2]}
    store.clear()
    assert store.next_id() == 0 and not store.is_indexed() and len(store) == 0


# ---- golden vectors ---------------------------------------------------------------------------

def test_generator_identity_on_gpu(VS):
    """The HIP generator writes byte-identical matrices to the C / numpy ones."""
    for d in GOLDEN["digests"]:
        store = VS(None, d["dim"])
        store.insert_synthetic(d["n"], d["seed"], d["first_row"])
        rows = store.read_rows(0, d["n"])
        assert hashlib.sha256(rows.tobytes()).hexdigest() == d["sha256"], d
        store.close()


def test_golden_cases(VS, oracle):
    stores = {}
    for case in GOLDEN["cases"]:
        n, dim, seed = case["n"], case["dim"], case["seed"]
        key = (n, dim, seed)
        if key not in stores:
            st = VS(None, dim)
            st.insert_synthetic(n, seed, 0)
            st.build_index()
            stores[key] = (st, oracle.synth_rows(seed, 0, n, dim))
        st, corpus = stores[key]
        if case["kind"] == "random":
            q = synth_rows(case["query_seed"], case["qi"], 1, dim)[0]
        else:
            q = synth_planted(seed, case["query_seed"], [case["planted_row"]] * (case["qi"] + 1), dim)[case["qi"]]
        cos, ids, counts = st.search_raw(q, case["k"])
        assert counts[0] == case["k"]
        assert_topk_equal(cos[0], ids[0], case["cos"], case["ids"], corpus, q, oracle)
        if case["kind"] == "planted":
            assert ids[0][0] == case["planted_row"]


# ---- HIP vs oracle on seeded inputs -----------------------------------------------------------

@pytest.mark.parametrize("dim", [384, 768, 1024, 100, 4])
def test_sizes_and_k_vs_oracle(VS, oracle, dim):
    for n in [1, 2, 7, 8, 9, 63, 64, 65, 1000, 4097]:
        corpus = oracle.synth_rows(1000 + n, 0, n, dim)
        st = VS(None, dim)
        st.insert_embeddings(corpus)
        st.build_index()
        q = synth_rows(2000 + n, 0, 1, dim)[0]
        for k in [1, 10, 25, 200, 256]:
            cos, ids, counts = st.search_raw(q, k)
            ecos, eids = oracle.scan_topk(corpus, q, k, mode="omp")
            assert counts[0] == len(eids) == min(k, n)
            assert_topk_equal(cos[0][: counts[0]], ids[0][: counts[0]], ecos, eids, corpus, q, oracle)
            if counts[0] < k:  # empty slots are marked
                assert (ids[0][counts[0]:] == 0xFFFFFFFF).all()
        st.close()


@pytest.mark.parametrize("nq", [1, 2, 3, 4, 5, 9, 17])
def test_batched_queries_vs_oracle(VS, oracle, nq):
    n, dim, k = 20011, 384, 25
    corpus = oracle.synth_rows(31, 0, n, dim)
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.build_index()
    qs = synth_rows(32, 0, nq, dim)
    cos, ids, counts = st.search_raw(qs, k)
    for i in range(nq):
        ecos, eids = oracle.scan_topk(corpus, qs[i], k, mode="omp")
        assert counts[i] == k
        assert_topk_equal(cos[i], ids[i], ecos, eids, corpus, qs[i], oracle)


def test_literal_oracle_agreement_100k(VS, oracle):
    """Config 2 shape at 1/10 scale against the LITERAL scalar restatement."""
    n, dim, k = 100_000, 384, 10
    st = VS(None, dim)
    st.insert_synthetic(n, 0xC0DE5EA, 0)
    st.build_index()
    corpus = oracle.synth_rows(0xC0DE5EA, 0, n, dim)
    q = synth_rows(0xC0DE5EB, 0, 1, dim)[0]
    cos, ids, counts = st.search_raw(q, k)
    ecos, eids = oracle.scan_topk(corpus, q, k, mode="literal")
    assert_topk_equal(cos[0], ids[0], ecos, eids, corpus, q, oracle)


def test_edges_ties_zero_rows_tombstones_id_base(VS, oracle):
    dim = 384
    base = synth_rows(11, 0, 6, dim)
    corpus = np.stack([base[0], base[1], base[0], np.zeros(dim, np.float32), base[0] * 2.0, base[2]] +
                      [base[3 + (i % 3)] for i in range(30)])
    n = len(corpus)
    st = VS(None, dim, id_base=1000)
    ids_in = st.insert_embeddings(corpus)
    assert ids_in.tolist() == list(range(1000, 1000 + n))
    st.build_index()
    q = base[0]
    cos, ids, counts = st.search_raw(q, 256)
    ecos, eids = oracle.scan_topk(corpus, q, 256, id_base=1000, mode="omp")
    assert counts[0] == n  # k > N: every live row exactly once
    assert ids[0][:3].tolist() == [1000, 1002, 1004]  # exact ties in id order
    assert cos[0][0] == cos[0][1] == cos[0][2]
    assert_topk_equal(cos[0][:n], ids[0][:n], ecos, eids)
    zero_at = ids[0][:n].tolist().index(1003)
    assert cos[0][zero_at] == 0.0  # zero-magnitude guard, batch.rs:320-322
    # tombstones
    assert st.delete_chunks([1000, 1004]) == 2
    st.build_index()
    dead = np.zeros((n + 31) // 32, np.uint32)
    dead[0] = (1 << 0) | (1 << 4)
    cos2, ids2, c2 = st.search_raw(q, 5)
    ecos2, eids2 = oracle.scan_topk(corpus, q, 5, dead=dead, id_base=1000, mode="omp")
    assert ids2[0][0] == 1002
    assert_topk_equal(cos2[0], ids2[0], ecos2, eids2)
    # rows appended after deletions keep getting fresh ids (ids are never reused)
    assert st.insert_embeddings(base[:1]).tolist() == [1000 + n]


def test_nan_rows_and_empty_index(VS):
    dim = 384
    corpus = np.zeros((3, dim), np.float32)
    corpus[0, 0] = np.nan
    corpus[1, 0] = 1.0
    corpus[2, 0] = np.inf
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.build_index()
    q = np.zeros(dim, np.float32)
    q[0] = 1.0
    cos, ids, counts = st.search_raw(q, 3)
    assert counts[0] == 1 and ids[0][0] == 1 and cos[0][0] == 1.0
    empty = VS(None, dim)
    empty.build_index()
    cos, ids, counts = empty.search_raw(q, 5)
    assert counts[0] == 0


def test_growth_keeps_rows(VS, oracle):
    dim = 384
    st = VS(None, dim, capacity=16)
    chunks = [oracle.synth_rows(5, i * 700, 700, dim) for i in range(5)]
    for c in chunks:
        st.insert_embeddings(c)
    allrows = np.concatenate(chunks)
    assert np.array_equal(st.read_rows(0, len(allrows)), allrows)
    st.build_index()
    q = synth_rows(6, 0, 1, dim)[0]
    cos, ids, _ = st.search_raw(q, 10)
    ecos, eids = oracle.scan_topk(allrows, q, 10, mode="omp")
    assert_topk_equal(cos[0], ids[0], ecos, eids, allrows, q, oracle)


def test_concurrent_searches_are_reentrant(VS, oracle):
    """search(&self) is called from rayon threads (src/search/mod.rs:508-511)."""
    n, dim, k = 50_000, 384, 10
    st = VS(None, dim)
    st.insert_synthetic(n, 77, 0)
    st.build_index()
    corpus = oracle.synth_rows(77, 0, n, dim)
    qs = synth_rows(78, 0, 8, dim)
    expect = [oracle.scan_topk(corpus, qs[i], k, mode="omp") for i in range(8)]
    errors = []

    def worker(i):
        try:
            for _ in range(5):
                cos, ids, _ = st.search_raw(qs[i], k)
                assert_topk_equal(cos[0], ids[0], expect[i][0], expect[i][1], corpus, qs[i], oracle)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


# ---- full size (BASELINE.json target: 10M x 384, top-10) ----------------------------------------

def _oracle_topk_by_slices(st, oracle, qs, n, kmax, seed, dim, slice_rows=1_000_000):
    """Exhaustive CPU answer for a corpus that only exists in HBM: pull it back in slices, scan each with
    the oracle at kmax, merge (top-k of a union = top-k of the per-slice top-ks).  The merged list is
    totally ordered (cosine desc, id asc), so its first k entries are the answer for every k <= kmax."""
    nsl = n // slice_rows
    pc = np.zeros((len(qs), nsl, kmax), np.float32)
    pi = np.zeros((len(qs), nsl, kmax), np.uint32)
    for s in range(nsl):
        rows = st.read_rows(s * slice_rows, slice_rows)
        if s in (0, 7):  # the slice really is what the host generator says
            assert np.array_equal(rows[:1000], synth_rows(seed, s * slice_rows, 1000, dim))
        for i in range(len(qs)):
            c, ii = oracle.scan_topk(rows, qs[i], kmax, id_base=s * slice_rows, mode="omp")
            pc[i, s], pi[i, s] = c, ii
        del rows
    return [oracle.merge_topk(pc[i], pi[i], np.full(nsl, kmax, np.uint32), kmax) for i in range(len(qs))]


@pytest.fixture(scope="module")
def full10m(VS, oracle):
    """BASELINE's target corpus — 10M x 384 generated in HBM — with the exhaustive CPU answer (oracle over 1M-row slices
    pulled back from HBM, one pass of 15 GB) for every query the full-size tests use: four planted + two random
    queries, and 16 of the 1,000 queries of configs[4]'s per-GPU workload."""
    n, dim, seed, kmax = 10_000_000, 384, 0xC0DE5EA, 200
    st = VS(None, dim, capacity=n)
    st.insert_synthetic(n, seed, 0)
    st.build_index()
    planted_rows = [123_456, 9_999_999, 0, 5_000_001]
    qs = np.concatenate([synth_planted(seed, seed + 2, planted_rows, dim), synth_rows(seed + 1, 0, 2, dim)])
    q1000 = synth_rows(seed + 9, 0, 1000, dim)
    sample = list(range(0, 1000, 67)) + [999]
    assert len(sample) == 16
    expect = _oracle_topk_by_slices(st, oracle, np.concatenate([qs, q1000[sample]]), n, kmax, seed, dim)
    yield dict(st=st, n=n, dim=dim, seed=seed, planted_rows=planted_rows, qs=qs, expect=expect[:len(qs)],
               q1000=q1000, sample=sample, expect1000=expect[len(qs):])
    st.close()


def test_full_size_10m_against_oracle_slices(VS, oracle, monkeypatch, full10m):
    """BASELINE's target: 10M x 384 generated in HBM, every search path against the exhaustive CPU oracle
    (examples/benchmark_models.rs:155-165 generalised to top-k).
    (a) planted queries: top-1 is the planted row at any size;
    (b) six queries in ONE call (filter + refine path) at k = 10;
    (c) the HEADLINE path — one query per call, k = 10 (and 25, 75): the primed one-block streaming f32 scan
        scan_topk_kernel<3,8,1,true,false> + prime pass, selected with CS_ROUTE_STREAM — for every query;
    (c') the DEFAULT route for the same searches (CS_ROUTE_COST): int8 filter + exact refine, proven by the
        debug counters, bit-identical to (c) at k = 10 / 25 / 75;
    (d) one query per call at k = 100 and 200 (the reference's retrieval limits, src/search/mod.rs:494-502):
        routed through filter + refine by default, and through the streaming scan on a second store
        created with CS_FILTER_SINGLE_MIN_K=0."""
    st, n, dim, seed = full10m["st"], full10m["n"], full10m["dim"], full10m["seed"]
    planted_rows, qs, expect = full10m["planted_rows"], full10m["qs"], full10m["expect"]
    b0, f0 = st.debug_counters()
    k = 10
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters() == (b0 + 1, f0)
    assert (counts == k).all()
    for i, r in enumerate(planted_rows):
        assert ids[i][0] == r and cos[i][0] > 0.85
    for i in range(len(qs)):
        assert (np.diff(cos[i]) <= 0).all() and len(set(ids[i].tolist())) == k
    for i in range(len(qs)):  # (b)
        ecos, eids = expect[i]
        assert ids[i].tolist() == eids[:k].tolist()
        np.testing.assert_allclose(cos[i], ecos[:k], atol=COS_TOL)
    # (c): the north-star kernel — selected explicitly (CS_ROUTE_STREAM): by default one query over this many rows goes
    # through the int8 filter (c')
    st.set_single_query_route(st.ROUTE_STREAM)
    streamed = {}
    for i in range(len(qs)):
        c1, i1, n1 = st.search_raw(qs[i], 10)
        assert n1[0] == 10 and i1[0].tolist() == expect[i][1][:10].tolist()
        np.testing.assert_allclose(c1[0], expect[i][0][:10], atol=COS_TOL)
        assert c1[0].tobytes() == cos[i].tobytes()  # and the two paths agree bit for bit
        streamed[(i, 10)] = (c1[0].copy(), i1[0].copy())
    for kk in (25, 75):
        for i in (0, 3, 5):
            c1, i1, n1 = st.search_raw(qs[i], kk)
            assert n1[0] == kk and i1[0].tolist() == expect[i][1][:kk].tolist()
            streamed[(i, kk)] = (c1[0].copy(), i1[0].copy())
    assert st.debug_counters() == (b0 + 1, f0)  # none of those took the batched path
    b = b0 + 1
    # (c') default routing (CS_ROUTE_COST): the reference's commonest searches — one query, k = 25 (src/server/mod.rs:547)
    # or limit * 3 (src/mcp/mod.rs:252) — take the int8 filter + exact refine over an index this large: same bits
    st.set_single_query_route(st.ROUTE_COST)
    assert st.filter_state()[0] == 2
    for (i, kk), (sc, si) in sorted(streamed.items()):
        c1, i1, n1 = st.search_raw(qs[i], kk)
        b += 1
        assert n1[0] == kk and i1[0].tolist() == si.tolist(), (i, kk)
        assert c1[0].tobytes() == sc.tobytes(), (i, kk)
    assert st.debug_counters() == (b, f0)   # every one of them went through filter + refine, none overflowed
    for kk in (100, 200):  # (d) default routing: filter + refine for one long-list query
        for i in (0, 4, 5):
            c1, i1, n1 = st.search_raw(qs[i], kk)
            b += 1
            assert n1[0] == kk and i1[0].tolist() == expect[i][1][:kk].tolist()
            np.testing.assert_allclose(c1[0], expect[i][0][:kk], atol=COS_TOL)
    assert st.debug_counters() == (b, f0)
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    st2 = VS(None, dim, capacity=n)
    st2.insert_synthetic(n, seed, 0)
    st2.build_index()
    for kk in (100, 200):  # (d) streaming f32 scan with long lists (two blocks per CU from k = 129)
        for i in (1, 5):
            c1, i1, n1 = st2.search_raw(qs[i], kk)
            assert n1[0] == kk and i1[0].tolist() == expect[i][1][:kk].tolist()
            np.testing.assert_allclose(c1[0], expect[i][0][:kk], atol=COS_TOL)
    assert st2.debug_counters() == (0, 0)
    st2.close()


def test_config5_per_gpu_workload_1000_queries_over_10m_rows(full10m):
    """BASELINE.json configs[4], one GPU's share: 1,000 batched queries, top-10, over that GPU's 10M x 384 rows (the
    256 x 256 f16 filter tiles + exact f32 refine).  16 sampled queries against the exhaustive CPU oracle; every query
    bit-equal — ids and cosines — to the single-query streaming scan, itself checked against the oracle above; the
    device-pointer call the sharded store issues (cs_index_search_device, > 16 queries) gives the same keys and
    reports no overflow."""
    import torch

    from codesearch_amd.sharded import key_unpack

    st, dim, q1000, sample = full10m["st"], full10m["dim"], full10m["q1000"], full10m["sample"]
    k = 10
    st.set_single_query_route(st.ROUTE_STREAM)  # the single-query searches below are the streaming-scan reference
    b0, f0 = st.debug_counters()
    cos, ids, counts = st.search_raw(q1000, k)
    assert st.debug_counters() == (b0 + 1, f0) and (counts == k).all()
    for j, i in enumerate(sample):
        ecos, eids = full10m["expect1000"][j]
        assert ids[i].tolist() == eids[:k].tolist()
        np.testing.assert_allclose(cos[i], ecos[:k], atol=COS_TOL)
    for i in range(len(q1000)):
        c1, i1, n1 = st.search_raw(q1000[i], k)  # scan_topk_kernel<3,8,1,true,false>
        assert n1[0] == k and i1[0].tolist() == ids[i].tolist() and c1[0].tobytes() == cos[i].tobytes(), i
    assert st.debug_counters() == (b0 + 1, f0)
    d_q = torch.from_numpy(q1000).to("cuda:0")
    keys = torch.zeros((1000, k), dtype=torch.int64, device="cuda:0")
    st.search_device(d_q.data_ptr(), 1000, k, d_keys=keys.data_ptr())
    assert st.search_status() is False
    kc, ki = key_unpack(keys.cpu().numpy().view(np.uint64))
    assert ki.tolist() == ids.tolist() and kc.tobytes() == cos.tobytes()
    st.set_single_query_route(st.ROUTE_COST)


def test_config1_1m_single_query_against_oracle(VS, oracle):
    """BASELINE configs[1]: one query, top-10 over 1M x 384 — the primed streaming scan at the size where
    the prime pass and the merge are 9 % of a search — and k = 100 / 200, all against the oracle on the
    whole corpus; the same searches on the default route (filter + refine), bit for bit; then one batched call."""
    n, dim, seed = 1_000_000, 384, 0xC0DE5EA
    st = VS(None, dim, capacity=n)
    st.insert_synthetic(n, seed, 0)
    st.delete_chunks([17, 999_999])
    st.build_index()
    corpus = st.read_rows(0, n)
    assert np.array_equal(corpus[-500:], synth_rows(seed, n - 500, 500, dim))
    dead = np.zeros((n + 31) // 32, np.uint32)
    for d in (17, 999_999):
        dead[d >> 5] |= np.uint32(1 << (d & 31))
    qs = np.concatenate([synth_rows(seed + 1, 0, 3, dim), synth_planted(seed, seed + 2, [999_998], dim)])
    expect = [oracle.scan_topk(corpus, q, 200, dead=dead, mode="omp") for q in qs]
    st.set_single_query_route(st.ROUTE_STREAM)   # configs[1] names the scan kernel: selected explicitly
    streamed = {}
    for kk in (10, 100, 200):
        for i in range(len(qs)):
            c1, i1, n1 = st.search_raw(qs[i], kk)
            assert n1[0] == kk
            assert_topk_equal(c1[0], i1[0], expect[i][0][:kk], expect[i][1][:kk], corpus, qs[i], oracle)
            streamed[(kk, i)] = (c1[0].copy(), i1[0].copy())
    assert st.debug_counters() == (0, 0)
    assert st.search_raw(qs[3], 10)[1][0][0] == 999_998
    # the default route at this size (CS_ROUTE_COST: the int8 filter from 32,768 rows on at k < 48): same bits, 0.13 vs 0.26 ms
    st.set_single_query_route(st.ROUTE_COST)
    for (kk, i), (sc, si) in sorted(streamed.items()):
        c1, i1, n1 = st.search_raw(qs[i], kk)
        assert i1[0].tolist() == si.tolist() and c1[0].tobytes() == sc.tobytes(), (kk, i)
    assert st.debug_counters() == (len(streamed), 0)
    b_before = st.debug_counters()[0]
    cos, ids, counts = st.search_raw(qs, 10)
    assert st.debug_counters() == (b_before + 1, 0)
    for i in range(len(qs)):
        assert_topk_equal(cos[i], ids[i], expect[i][0][:10], expect[i][1][:10], corpus, qs[i], oracle)


# ---- batched queries: the MFMA scoring + phased selection path (scan_mfma.hip) ------------------

@pytest.mark.parametrize("dim,n,nq,k", [(384, 100, 5, 10), (384, 5000, 33, 25), (384, 200_003, 64, 10),
                                        (384, 70_001, 100, 200), (768, 30_000, 40, 10), (384, 31, 7, 256)])
def test_mfma_batched_path_vs_oracle(VS, oracle, dim, n, nq, k):
    corpus = oracle.synth_rows(900 + nq, 0, n, dim)
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    dead_ids = [3, n // 2, n - 1] if n > 50 else []
    dead = None
    if dead_ids:
        st.delete_chunks(dead_ids)
        dead = np.zeros((n + 31) // 32, np.uint32)
        for d in dead_ids:
            dead[d >> 5] |= np.uint32(1 << (d & 31))
    st.build_index()
    qs = np.concatenate([synth_rows(901 + nq, 0, nq - 2, dim),
                         synth_planted(900 + nq, 77, [n // 3, (2 * n) // 3], dim)])
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters() == (1, 0)  # took the batched path, no overflow rerun
    for i in range(nq):
        ecos, eids = oracle.scan_topk(corpus, qs[i], k, dead=dead, mode="omp")
        assert counts[i] == len(eids)
        assert_topk_equal(cos[i][: counts[i]], ids[i][: counts[i]], ecos, eids, corpus, qs[i], oracle)
    assert ids[nq - 2][0] == n // 3 and ids[nq - 1][0] == (2 * n) // 3


def test_mfma_batched_overflow_falls_back_to_exact(VS, oracle):
    """Rows ordered so every later row beats all earlier ones for query 0: the candidate
    buffer overflows, the search is rerun on the list-based kernel, and is still exact."""
    n, dim, k, nq = 300_000, 384, 10, 8
    q = synth_rows(5, 0, nq, dim)
    u = synth_rows(6, 0, 1, dim)[0]
    u = u - (u @ q[0]) / (q[0] @ q[0]) * q[0]
    w = np.linspace(3.0, 0.5, n, dtype=np.float32)[:, None]
    corpus = (q[0][None, :] + w * u[None, :]).astype(np.float32)  # cos(q0, row_i) increases with i
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.build_index()
    cos, ids, counts = st.search_raw(q, k)
    assert st.debug_counters() == (1, 1)
    for i in range(nq):
        ecos, eids = oracle.scan_topk(corpus, q[i], k, mode="omp")
        assert_topk_equal(cos[i], ids[i], ecos, eids, corpus, q[i], oracle)


def test_mfma_batched_10m_equals_single_query_scans(full10m):
    """BASELINE configs 4/5 shape on one GPU: 64 (and 100) batched queries over 10M rows must
    equal 64 independent single-query scans (which are checked against the oracle above)."""
    st, dim, k, seed = full10m["st"], full10m["dim"], 10, full10m["seed"]
    st.set_single_query_route(st.ROUTE_STREAM)  # the single-query searches below are the streaming-scan reference
    b0, f0 = st.debug_counters()
    for nq in (64, 100):
        qs = synth_rows(seed + 5, 0, nq, dim)
        cos, ids, counts = st.search_raw(qs, k)
        assert (counts == k).all()
        for i in list(range(0, nq, 9)) + [nq - 1]:
            c1, i1, _ = st.search_raw(qs[i], k)
            assert ids[i].tolist() == i1[0].tolist()
            assert cos[i].tobytes() == c1[0].tobytes()  # refine = the single-query arithmetic
    assert st.debug_counters() == (b0 + 2, f0)
    st.set_single_query_route(st.ROUTE_COST)


# ---- batched queries: filter (split-f16 MFMA) + exact refine (scan_split.hip) ---------------------

@pytest.mark.parametrize("dim,n,nq,k", [(384, 50_000, 130, 10), (768, 20_001, 9, 25), (1024, 10_000, 33, 10),
                                        (384, 1000, 5, 256),
                                        # resident-query filter kernel at the wider models' dims: <1,12>, <2,12>, <1,16>
                                        (768, 60_000, 32, 10), (768, 30_003, 40, 200), (768, 40_000, 64, 10),
                                        (1024, 30_000, 8, 200), (1024, 50_001, 32, 10)])
def test_split_batched_path_is_bit_identical_to_single_query_scan(VS, dim, n, nq, k):
    """The refine step re-scores candidates with the single-query scan's arithmetic, so a batched
    search returns the same bits — ids AND cosines — as nq independent searches."""
    st = VS(None, dim)
    st.insert_synthetic(n, 4242 + dim, 0)
    st.delete_chunks([1, n // 2])
    st.build_index()
    qs = np.concatenate([synth_rows(77 + nq, 0, nq - 1, dim), synth_planted(4242 + dim, 5, [n // 4], dim)])
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters() == (1, 0)
    for i in range(nq):
        c1, i1, n1 = st.search_raw(qs[i], k)
        assert counts[i] == n1[0]
        assert ids[i].tolist() == i1[0].tolist()
        assert cos[i].tobytes() == c1[0].tobytes()
    assert ids[nq - 1][0] == n // 4


def test_split_batched_path_any_row_magnitude(VS, oracle):
    """The filter works on unit vectors, so rows far outside the f16 range (1e6), far below it
    (1e-9), zero rows and NaN/Inf rows are all handled exactly as by the single-query scan."""
    dim, n, nq, k = 384, 4000, 6, 12
    corpus = oracle.synth_rows(31, 0, n, dim).copy()
    scale = np.ones(n, np.float32)
    scale[0::7] = 1e6
    scale[1::7] = 1e-9
    scale[2::7] = 3e-5
    corpus *= scale[:, None]
    corpus[10] = 0.0
    corpus[11, 5] = np.nan
    corpus[12, 6] = np.inf
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.build_index()
    qs = synth_rows(32, 0, nq, dim) * np.float32(250.0)
    qs[1] *= np.float32(1e-7)
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters() == (1, 0)
    for i in range(nq):
        c1, i1, n1 = st.search_raw(qs[i], k)
        assert counts[i] == n1[0] and ids[i].tolist() == i1[0].tolist() and cos[i].tobytes() == c1[0].tobytes()
        ok = [r for r in range(n) if r not in (11, 12)]
        ecos, eids = oracle.scan_topk(corpus[ok], qs[i], k, mode="omp")
        assert_topk_equal(cos[i], ids[i], ecos, np.asarray(ok)[eids], corpus, qs[i], oracle)


def test_f32_mfma_batched_path_still_available(VS, oracle, monkeypatch):
    """CS_INDEX_SPLIT=0 keeps the batched path on the exact-f32 MFMA kernels (no second copy of
    the corpus in HBM)."""
    monkeypatch.setenv("CS_INDEX_SPLIT", "0")
    dim, n, nq, k = 384, 5000, 33, 25
    corpus = oracle.synth_rows(933, 0, n, dim)
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.build_index()
    qs = synth_rows(934, 0, nq, dim)
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters() == (1, 0)
    for i in range(nq):
        ecos, eids = oracle.scan_topk(corpus, qs[i], k, mode="omp")
        assert_topk_equal(cos[i], ids[i], ecos, eids, corpus, qs[i], oracle)


@pytest.mark.parametrize("nq", [129, 300, 1000])
def test_wide_filter_tiles_bit_identical(VS, oracle, nq):
    """More than 128 queries take the 256 x 256 filter tiles; results are the oracle's and those of the
    single-query scan, bit for bit (sampled queries)."""
    dim, n, k = 384, 300_001, 10
    st = VS(None, dim)
    st.insert_synthetic(n, 515, 0)
    dead_ids = [7, n - 1]
    st.delete_chunks(dead_ids)
    st.build_index()
    qs = np.concatenate([synth_rows(600 + nq, 0, nq - 1, dim), synth_planted(515, 6, [n - 2], dim)])
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters() == (1, 0)
    corpus = oracle.synth_rows(515, 0, n, dim)
    dead = np.zeros((n + 31) // 32, np.uint32)
    for d in dead_ids:
        dead[d >> 5] |= np.uint32(1 << (d & 31))
    for i in list(range(0, nq, max(1, nq // 16))) + [nq - 2, nq - 1]:
        c1, i1, n1 = st.search_raw(qs[i], k)
        assert counts[i] == n1[0] and ids[i].tolist() == i1[0].tolist() and cos[i].tobytes() == c1[0].tobytes()
        ecos, eids = oracle.scan_topk(corpus, qs[i], k, dead=dead, mode="omp")
        assert_topk_equal(cos[i], ids[i], ecos, eids, corpus, qs[i], oracle)
    assert ids[nq - 1][0] == n - 2


# ---- persistence interop: a store with a path survives a restart ---------------

def test_persistent_store_round_trip(VS, oracle, tmp_path):
    from codesearch_amd import CsError
    from codesearch_amd.vector_store import Chunk, EmbeddedChunk

    dim, n = 384, 300
    rows = oracle.synth_rows(88, 0, n, dim)
    chunks = [EmbeddedChunk(Chunk(f"fn f{i}() {{}}", i, i + 1, "Function", f"src/m{i % 7}.rs", hash=f"h{i}"), rows[i])
              for i in range(n)]
    db = tmp_path / "vectors.db"
    st = VS(db, dim)
    ids = st.insert_chunks_with_ids(chunks[:200])
    assert ids == list(range(200))
    st.build_index()
    st.delete_chunks([5, 17])
    st.insert_chunks_with_ids(chunks[200:])
    st.build_index()                      # second build appends rows 200..299 to the flat file
    q = rows[123] + 0.01 * rows[124]
    want = [(r.id, r.score, r.path) for r in st.search(q, 10)]
    size = st.db_size()
    assert size >= n * dim * 4
    st.close()

    st2 = VS(db, dim)                     # VectorStore::new on an existing directory, store.rs:139-170
    assert st2.is_indexed() and st2.next_id() == n and len(st2) == n - 2
    assert [(r.id, r.score, r.path) for r in st2.search(q, 10)] == want
    assert st2.get_chunk(5) is None and st2.get_chunk(6).hash == "h6"
    assert np.array_equal(st2.read_rows(0, n), rows)
    assert st2.stats().total_chunks == n - 2 and st2.db_size() == size
    new_id = st2.insert_chunks_with_ids([EmbeddedChunk(Chunk("x", 1, 2, "Function", "a.rs"), rows[0])])
    assert new_id == [n]                  # ids are never reused (store.rs:101)
    st2.close()

    st3 = VS(db, dim)                     # a delete is committed at once (store.rs:584-610), not at the next build
    assert st3.delete_chunks([6, 123]) == 2
    st3.close()                           # ... so a process that ends here
    st3 = VS(db, dim)
    assert st3.get_chunk(6) is None and st3.get_chunk(7).hash == "h7" and len(st3) == n - 4 and st3.next_id() == n
    assert all(r.id not in (6, 123) for r in st3.search(q, 10))
    st3.build_index()                     # rewrites chunks.jsonl without the deleted lines, appends nothing
    lines = [__import__("json").loads(l)["id"] for l in open(db / "chunks.jsonl")]
    assert 6 not in lines and 123 not in lines and len(lines) == len(set(lines)) == n - 4
    st3.insert_chunks_with_ids([EmbeddedChunk(Chunk("y", 1, 2, "Function", "b.rs"), rows[1])])
    st3.build_index()                     # only the new chunk is appended
    assert sum(1 for _ in open(db / "chunks.jsonl")) == len(lines) + 1
    st3.delete_chunks([n])
    st3.build_index()
    want = [(r.id, r.score, r.path) for r in st3.search(q, 10)]
    assert all(w[0] not in (5, 17, 6, 123, n) for w in want)
    st3.close()

    ro = VS.open_readonly(db, dim)        # store.rs:183-250
    assert [(r.id, r.score, r.path) for r in ro.search(q, 10)] == want
    with pytest.raises(CsError):
        ro.insert_chunks_with_ids(chunks[:1])
    ro.close()
    with pytest.raises(CsError) as e:     # a store written at 384 dims opened at 768
        VS(db, 768)
    assert "dimension mismatch" in str(e.value)


def test_single_query_through_filter_is_bit_identical(VS):
    """One query may take filter + refine — by default (CS_ROUTE_COST) from 32,768 rows on below k = 48 and from 300,000 rows
    on above (the measured crossovers, index.hip) when the int8 copy serves, at
    any size with CS_ROUTE_FILTER or cs_index_set_filter_min_queries(1); CS_ROUTE_STREAM pins the streaming scan.  Same
    bits on every route; the debug counters prove which route ran."""
    dim, n, k = 384, 200_000, 10
    st = VS(None, dim)
    st.insert_synthetic(n, 99, 0)
    st.build_index()
    qs = synth_rows(1234, 0, 5, dim)
    st.set_single_query_route(st.ROUTE_STREAM)
    base = [st.search_raw(q, k) for q in qs]
    assert st.debug_counters() == (0, 0)
    st.set_single_query_route(st.ROUTE_COST)      # the default: 200,000 rows are past the crossover
    for q, (c0, i0, n0) in zip(qs, base):
        c1, i1, n1 = st.search_raw(q, k)
        assert n1[0] == n0[0] and i1.tolist() == i0.tolist() and c1.tobytes() == c0.tobytes()
    assert st.debug_counters() == (5, 0)
    for kk, filtered in ((47, 1), (48, 0), (75, 0)):  # a long list over 200,000 rows is still ahead on the streaming scan
        before = st.debug_counters()[0]
        c1, i1, n1 = st.search_raw(qs[0], kk)
        assert st.debug_counters()[0] - before == filtered, kk
        st.set_single_query_route(st.ROUTE_STREAM)
        c0, i0, n0 = st.search_raw(qs[0], kk)
        st.set_single_query_route(st.ROUTE_COST)
        assert n1[0] == n0[0] and i1.tolist() == i0.tolist() and c1.tobytes() == c0.tobytes()
    small = VS(None, dim)
    small.insert_synthetic(20_000, 99, 0)         # below the crossover the default route streams ...
    small.build_index()
    sbase = [small.search_raw(q, k) for q in qs]
    assert small.debug_counters() == (0, 0)
    small.set_single_query_route(small.ROUTE_FILTER)   # ... unless told otherwise
    for q, (c0, i0, n0) in zip(qs, sbase):
        c1, i1, n1 = small.search_raw(q, k)
        assert n1[0] == n0[0] and i1.tolist() == i0.tolist() and c1.tobytes() == c0.tobytes()
    assert small.debug_counters() == (5, 0)
    small.set_single_query_route(small.ROUTE_COST)
    small.set_filter_min_queries(1)                    # round 2's knob still does the same
    c1, i1, n1 = small.search_raw(qs[0], k)
    assert i1.tolist() == sbase[0][1].tolist() and c1.tobytes() == sbase[0][0].tobytes()
    assert small.debug_counters() == (6, 0)


@pytest.mark.parametrize("dim", [384, 768, 1024])
def test_primed_scan_is_bit_identical(VS, oracle, dim, monkeypatch):
    """Large-k streaming scans start from a lower bound of the k-th best cosine taken from a
    pass over a corpus prefix (scan.hip, PRIME mode).  Forced on at test sizes; must equal the
    unprimed scan bit for bit — ties at the bound, a bound of exactly 0 (zero rows), tombstones
    inside the sample, and samples with fewer than k waves (no bound) included."""
    n = 12000
    rows = synth_rows(77, 0, n, dim)
    rows[5] = rows[2]; rows[900] = rows[2]; rows[4000] = rows[2] * 3.0   # exact ties, in and out of the sample
    rows[10:200] = 0.0                                                    # cosine exactly 0 inside the sample
    rows[3000:3300] = 0.0
    rows[300:2400:7] = 0.0
    qs = np.stack([rows[2], synth_rows(78, 0, 1, dim)[0], -rows[2], synth_rows(78, 5, 1, dim)[0]])

    def run(primed):
        monkeypatch.setenv("CS_INDEX_SPLIT", "0")          # 2..4 queries stay on the streaming scan
        monkeypatch.setenv("CS_SCAN_PRIME_MIN_K", "1" if primed else "0")
        monkeypatch.setenv("CS_SCAN_PRIME_MIN_ROWS", "1")
        monkeypatch.setenv("CS_SCAN_PRIME_ROWS", "2400")
        st = VS(None, dim)
        st.insert_embeddings(rows)
        st.build_index()
        out = []
        for dead in (False, True):
            if dead:
                assert st.delete_chunks(list(range(0, 120)) + [900, 5000]) == 122
                st.build_index()
            for k in (1, 10, 64, 200, 256):
                for nq in (1, 2, 4):
                    out.append((k, nq, dead) + tuple(st.search_raw(qs[:nq], k)))
        return out

    got, want = run(True), run(False)
    for (k, nq, dead, c1, i1, n1), (_, _, _, c0, i0, n0) in zip(got, want):
        assert n1.tolist() == n0.tolist(), (k, nq, dead)
        assert i1.tolist() == i0.tolist(), (k, nq, dead)
        assert c1.tobytes() == c0.tobytes(), (k, nq, dead)
    # and the unprimed result is the oracle's
    k, nq, dead, c0, i0, _ = want[3 * 3]  # k=200, nq=1, no tombstones
    assert (k, nq, dead) == (200, 1, False)
    ecos, eids = oracle.scan_topk(rows, qs[0], 200, mode="omp")
    assert_topk_equal(c0[0], i0[0], ecos, eids, rows, qs[0], oracle)


@pytest.mark.parametrize("dim", [384, 768, 100])
def test_large_k_up_to_cs_max_k(VS, oracle, dim, monkeypatch):
    """retrieval_limit = max(5 * max_results, 200) (src/search/mod.rs:494-502) passes 200 at the
    default and 500 at max_results = 100: every k up to CS_MAX_K = 1024 is served, by the streaming
    scan (primed and not), by filter + refine for several queries, by the exact-f32 MFMA path and
    by the any-dim kernel — all equal to the oracle; k = 1025 is refused."""
    from codesearch_amd import CsError, _lib

    assert _lib.CS_MAX_K == 1024
    n = 20_000
    corpus = oracle.synth_rows(4242, 0, n, dim)
    corpus[7] = corpus[3]; corpus[11_000] = corpus[3]
    qs = np.stack([corpus[3], synth_rows(4243, 0, 1, dim)[0], synth_rows(4243, 9, 1, dim)[0]])
    monkeypatch.setenv("CS_SCAN_PRIME_MIN_ROWS", "1")
    monkeypatch.setenv("CS_SCAN_PRIME_ROWS", "4096")
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.build_index()
    for k in (257, 500, 1000, 1024):
        want = [oracle.scan_topk(corpus, q, k, mode="omp") for q in qs]
        c1, i1, n1 = st.search_raw(qs[0], k)                    # one query: streaming scan
        assert n1[0] == k
        assert_topk_equal(c1[0], i1[0], want[0][0], want[0][1], corpus, qs[0], oracle)
        c3, i3, n3 = st.search_raw(qs, k)                       # three: filter + refine where the dim has it
        assert n3.tolist() == [k] * 3
        for j in range(3):
            assert_topk_equal(c3[j], i3[j], want[j][0], want[j][1], corpus, qs[j], oracle)
        assert i3[0].tolist() == i1[0].tolist() and c3[0].tobytes() == c1[0].tobytes()
    with pytest.raises(CsError) as e:
        st.search_raw(qs[0], 1025)
    assert "k must be in 1..1024" in str(e.value)
    # k larger than the corpus: every live row once, the rest marked empty
    small = VS(None, dim)
    small.insert_embeddings(corpus[:300])
    small.build_index()
    c, i, cnt = small.search_raw(qs, 1024)
    assert cnt.tolist() == [300] * 3 and (i[:, 300:] == 0xFFFFFFFF).all()
    for j in range(3):
        ecos, eids = oracle.scan_topk(corpus[:300], qs[j], 1024, mode="omp")
        assert_topk_equal(c[j][:300], i[j][:300], ecos, eids, corpus[:300], qs[j], oracle)
    if dim == 384:  # exact-f32 MFMA path (no filter copy) at 5+ queries
        monkeypatch.setenv("CS_INDEX_SPLIT", "0")
        st2 = VS(None, dim)
        st2.insert_embeddings(corpus)
        st2.build_index()
        q5 = np.concatenate([qs, synth_rows(4244, 0, 2, dim)])
        c5, i5, n5 = st2.search_raw(q5, 600)
        for j in range(5):
            ecos, eids = oracle.scan_topk(corpus, q5[j], 600, mode="omp")
            assert_topk_equal(c5[j], i5[j], ecos, eids, corpus, q5[j], oracle)


def test_randomised_paths_agree(VS, oracle, monkeypatch):
    """Seeded sweep over corpus size, query count, k, width and tombstones: the batched paths (filter + refine,
    phase 0 scored directly, candidates appended in batches) and the primed streaming scan must return the
    same bits as the unprimed single-query scan, which is checked against the oracle."""
    rng = np.random.default_rng(20261003)
    monkeypatch.setenv("CS_SCAN_PRIME_MIN_ROWS", "1")
    monkeypatch.setenv("CS_SCAN_PRIME_ROWS", "2048")
    monkeypatch.setenv("CS_SINGLE_BATCHED_MAX_ROWS", "0")  # single queries stay on the streaming scan at every size
    for case in range(36):
        dim = int(rng.choice([384, 384, 768, 1024]))
        n = int(rng.choice([1, 33, 700, 1024, 1025, 3000, 9000, 25_000]))
        nq = int(rng.choice([2, 3, 9, 31, 33, 64, 70]))
        k = int(rng.choice([1, 7, 10, 100, 200, 300, 1024]))
        rows = synth_rows(5000 + case, 0, n, dim)
        if n > 40:
            rows[n // 3] = rows[1]                       # exact tie
            rows[n // 2] = 0.0                           # zero row
            rows[5] *= 1e-6; rows[7] *= 1e5              # magnitudes the unit-vector filter must not care about
        qs = synth_rows(6000 + case, 0, nq, dim)
        qs[0] = rows[1] if n > 1 else qs[0]
        monkeypatch.setenv("CS_SCAN_PRIME_MIN_K", "1")
        st = VS(None, dim)
        st.insert_embeddings(rows)
        if n > 40 and case % 2:
            dead_ids = sorted(set(int(x) for x in rng.integers(0, n, size=n // 10)))
            assert st.delete_chunks(dead_ids) == len(dead_ids)
        st.build_index()
        cb, ib, nb = st.search_raw(qs, k)                # batched: filter + refine
        singles = [st.search_raw(q, k) for q in qs[:4]]  # primed streaming scan (prime forced on where n allows)
        monkeypatch.setenv("CS_SCAN_PRIME_MIN_K", "0")
        ref = VS(None, dim)                              # unprimed streaming scan
        ref.insert_embeddings(rows)
        if n > 40 and case % 2:
            ref.delete_chunks(dead_ids)
        ref.build_index()
        for j in range(min(4, nq)):
            c0, i0, n0 = ref.search_raw(qs[j], k)
            tag = (case, dim, n, nq, k, j)
            assert nb[j] == n0[0] == singles[j][2][0], tag
            assert ib[j].tolist() == i0[0].tolist() == singles[j][1][0].tolist(), tag
            assert cb[j].tobytes() == c0[0].tobytes() == singles[j][0][0].tobytes(), tag
        if case % 6 == 0:
            dead = None
            if n > 40 and case % 2:
                dead = np.zeros((n + 31) // 32, np.uint32)
                for d in dead_ids:
                    dead[d >> 5] |= np.uint32(1 << (d & 31))
            c0, i0, n0 = ref.search_raw(qs[0], k)
            ecos, eids = oracle.scan_topk(rows, qs[0], k, dead=dead, mode="omp")
            assert_topk_equal(c0[0][: n0[0]], i0[0][: n0[0]], ecos, eids, rows, qs[0], oracle)
        st.close(); ref.close()


@pytest.mark.parametrize("n", [1, 9, 592, 1024, 1025, 3000])
def test_small_corpus_single_query_both_routes(VS, oracle, monkeypatch, n):
    """One query over a corpus of the reference's own size takes the batched path (prep + direct scoring + select) up to
    1,024 rows and the streaming scan above; CS_SINGLE_BATCHED_MAX_ROWS=0 keeps it on the scan.  Both routes must
    return the oracle's answer and the same bits, with tombstones, an exact tie and a zero row in the corpus."""
    dim = 384
    rows = synth_rows(4242 + n, 0, n, dim)
    if n > 40:
        rows[n // 3] = rows[1]
        rows[n // 2] = 0.0
    q = synth_planted(4242 + n, 3, [min(1, n - 1)], dim)[0]
    dead_ids = [0, n // 5] if n > 40 else []
    stores = []
    for max_rows in ("0", None):
        if max_rows is None:
            monkeypatch.delenv("CS_SINGLE_BATCHED_MAX_ROWS", raising=False)
        else:
            monkeypatch.setenv("CS_SINGLE_BATCHED_MAX_ROWS", max_rows)
        st = VS(None, dim)
        st.insert_embeddings(rows)
        if dead_ids:
            assert st.delete_chunks(dead_ids) == len(set(dead_ids))
        st.build_index()
        stores.append(st)
    dead = None
    if dead_ids:
        dead = np.zeros((n + 31) // 32, np.uint32)
        for d in dead_ids:
            dead[d >> 5] |= np.uint32(1 << (d & 31))
    for k in (1, 10, 200):
        ecos, eids = oracle.scan_topk(rows, q, k, dead=dead, mode="omp")
        (c0, i0, n0), (c1, i1, n1) = stores[0].search_raw(q, k), stores[1].search_raw(q, k)
        assert n0[0] == n1[0] == len(eids)
        assert i0[0].tolist() == i1[0].tolist() and c0[0].tobytes() == c1[0].tobytes()
        assert_topk_equal(c1[0][: n1[0]], i1[0][: n1[0]], ecos, eids, rows, q, oracle)
    for st in stores:
        st.close()


@pytest.mark.parametrize("world,nq,k", [(8, 1, 10), (8, 9, 200), (8, 64, 10), (2, 5, 1), (8, 3, 1024), (4, 2, 700),
                                        (3, 1000, 10)])
def test_device_shard_merge_matches_host_statement(gpu_lib, world, nq, k):
    """cs_merge_topk_device (S4: what every rank runs on the all-gathered [world][nq][k] keys) against the
    host statement of the merge, with empty slots, duplicates across shards excluded by construction (ids are
    shard-disjoint) and the multi-level path (world * k > one sort: k = 700, 1024)."""
    import ctypes as C

    import torch

    from codesearch_amd import _lib
    from codesearch_amd.sharded import key_pack, key_unpack, merge_keys_host

    rng = np.random.default_rng(world * 1000 + nq + k)
    lists = np.zeros((world, nq, k), np.uint64)
    for w in range(world):
        for q in range(nq):
            live = int(rng.integers(0, k + 1)) if (w + q) % 3 else k      # some lists only partly filled
            cos = np.sort(rng.uniform(-1, 1, size=live).astype(np.float32))[::-1]
            if live > 2:
                cos[1] = cos[0]                                            # equal cosines: id order decides
            ids = (w * 1_000_000 + np.sort(rng.choice(900_000, size=live, replace=False))).astype(np.uint32)
            keys = np.sort(key_pack(cos, ids))[::-1]
            lists[w, q, :live] = keys
    want = merge_keys_host(lists, k)
    dev = "cuda:0"
    d_in = torch.from_numpy(lists.view(np.int64)).to(dev)
    d_keys = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    d_cos = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    d_ids = torch.zeros((nq, k), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros((nq,), dtype=torch.int32, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(gpu_lib.cs_merge_topk_device(0, vp(d_in), world, nq, k, vp(d_keys), vp(d_cos), vp(d_ids), vp(d_cnt), stream))
    torch.cuda.synchronize()
    got = d_keys.cpu().numpy().view(np.uint64)
    assert (got == want).all()
    wc, wi = key_unpack(want)
    live = want != 0
    assert (d_cnt.cpu().numpy().astype(np.uint32) == live.sum(axis=1)).all()
    assert (d_cos.cpu().numpy()[live] == wc[live]).all()
    assert (d_ids.cpu().numpy().view(np.uint32)[live] == wi[live]).all()
    assert (d_ids.cpu().numpy().view(np.uint32)[~live] == 0xFFFFFFFF).all()
