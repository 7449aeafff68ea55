"""The int8 filter copy (scan_filter.hip, "int8 filter copy"): a quarter-size quantised copy of the corpus feeds the
filter of batched searches of up to 128 queries; the refine step re-scores its candidates from the f32 rows, so the
results must stay bit-identical to the single-query exact scan — whatever the data does to the quantiser.  Needs an
MI355X.  The same shapes run with CS_FILTER_INT8=0 so that the f16 resident-query kernel stays covered."""
import numpy as np
import pytest

from codesearch_amd.synth import synth_planted, synth_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def VS(gpu_lib):
    from codesearch_amd import VectorStore

    assert gpu_lib.cs_device_count() >= 1, "no HIP device visible"
    return VectorStore


def _same_as_single_query_scans(st, qs, k):
    st.set_single_query_route(st.ROUTE_STREAM)  # the single-query searches below are the streaming-scan reference
    before = st.debug_counters()[1]
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters()[1] == before, "candidate buffer overflowed: the filter path was not what answered"
    for i in range(len(qs)):
        c1, i1, n1 = st.search_raw(qs[i], k)
        assert counts[i] == n1[0]
        assert ids[i].tolist() == i1[0].tolist(), i
        assert cos[i].tobytes() == c1[0].tobytes(), i
    return cos, ids, counts


@pytest.mark.parametrize("int8", ["1", "0"])
@pytest.mark.parametrize("dim,n,nq,k", [(384, 300_000, 2, 10), (384, 200_003, 9, 200), (384, 150_000, 33, 10),
                                        (384, 100_100, 64, 25), (384, 120_000, 100, 10), (384, 90_000, 128, 40),
                                        (768, 60_000, 9, 200), (768, 50_001, 64, 10), (768, 40_000, 96, 10),
                                        (1024, 50_001, 32, 10), (1024, 40_000, 64, 100),
                                        # several query tiles per row group (int8: 256 / 128 / 64 resident queries per
                                        # block at 384 / 768 / 1024; f16: the 256 x 256 tile kernel)
                                        (384, 50_000, 300, 10), (384, 40_000, 1000, 10), (384, 30_000, 129, 100),
                                        (768, 30_000, 200, 25), (1024, 20_000, 130, 10)])
def test_filter_copy_choice_does_not_change_a_bit(VS, monkeypatch, int8, dim, n, nq, k):
    monkeypatch.setenv("CS_FILTER_INT8", int8)
    # keep single queries on the streaming f32 scan: it is the yardstick here
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    st = VS(None, dim)
    st.insert_synthetic(n, 99 + dim, 0)
    st.delete_chunks([3, n // 2, n - 1])
    st.build_index()
    qs = np.concatenate([synth_rows(7 + nq, 0, nq - 1, dim), synth_planted(99 + dim, 5, [n // 3], dim)])
    cos, ids, _ = _same_as_single_query_scans(st, qs, k)
    assert ids[nq - 1][0] == n // 3


@pytest.mark.parametrize("dim,n,nq,k", [(384, 50_000, 300, 10), (384, 33_000, 1000, 10), (768, 20_100, 260, 25),
                                        (1024, 10_000, 257, 10)])
def test_int8_tile_kernel_for_query_counts_past_the_resident_limit(lab_lib, VS, monkeypatch, dim, n, nq, k):
    """Past 32 query tiles the int8 copy goes through the 256 x 256 tile kernel (score_filter256p_kernel<true>);
    CS_FILTER_INT8_RW_MAX_Q (a laboratory knob: the diagnostic library, lab_lib) lowers the switch-over so that a few hundred
    queries reach it.  Odd tile counts: the last
    256-row block holds one real 128-row tile."""
    monkeypatch.setenv("CS_FILTER_INT8_RW_MAX_Q", "128")
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    st = VS(None, dim)
    st.insert_synthetic(n, 12 + dim, 0)
    st.delete_chunks([7, n // 2])
    st.build_index()
    qs = np.concatenate([synth_rows(70 + nq, 0, nq - 1, dim), synth_planted(12 + dim, 5, [n - 1], dim)])
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters()[1] == 0
    for i in list(range(0, nq, 37)) + [nq - 1]:
        c1, i1, n1 = st.search_raw(qs[i], k)
        assert counts[i] == n1[0] and ids[i].tolist() == i1[0].tolist() and cos[i].tobytes() == c1[0].tobytes(), i
    assert ids[nq - 1][0] == n - 1


@pytest.mark.parametrize("rq", ["1", "0"])
@pytest.mark.parametrize("n,nq,k", [(70_000, 129, 10), (50_000, 257, 64), (33_000, 700, 10)])
def test_many_queries_both_resident_query_kernels(lab_lib, VS, monkeypatch, rq, n, nq, k):
    """Above 128 queries at dim 384 the int8 copy is scored by score_filter_rq8_kernel (eight waves per block, corpus
    fragments through registers); CS_FILTER_INT8_RQ=0 keeps score_filter_rw8_kernel<8, 3>.  Odd tile counts: the last
    256-row unit holds one real 128-row tile."""
    monkeypatch.setenv("CS_FILTER_INT8_RQ", rq)
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    st = VS(None, 384)
    st.insert_synthetic(n, 1234, 0)
    st.delete_chunks([5, n // 2, n - 3])
    st.build_index()
    qs = np.concatenate([synth_rows(80 + nq, 0, nq - 2, 384), synth_planted(1234, 5, [n - 1, n // 128 * 128 - 1], 384)])
    before = st.debug_counters()[1]
    cos, ids, counts = st.search_raw(qs, k)
    assert st.debug_counters()[1] == before
    for i in list(range(0, nq, 41)) + [nq - 2, nq - 1]:
        c1, i1, n1 = st.search_raw(qs[i], k)
        assert counts[i] == n1[0] and ids[i].tolist() == i1[0].tolist() and cos[i].tobytes() == c1[0].tobytes(), i
    assert ids[nq - 2][0] == n - 1 and ids[nq - 1][0] == n // 128 * 128 - 1


@pytest.mark.parametrize("q2", ["2", "0"])
@pytest.mark.parametrize("dim,n,nq,k", [(384, 200_000, 9, 200), (384, 150_000, 33, 10), (384, 120_000, 64, 100),
                                        (768, 60_000, 8, 200), (768, 50_000, 40, 25)])
def test_two_plane_queries_do_not_change_a_bit(lab_lib, VS, monkeypatch, q2, dim, n, nq, k):
    """CS_FILTER_INT8_Q2=2 takes every search of up to 64 queries through the two-plane query kernels (default: long
    lists of up to 32 queries only), =0 none: same bits either way; adversarial magnitudes ride along."""
    monkeypatch.setenv("CS_FILTER_INT8_Q2", q2)
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    st = VS(None, dim)
    st.insert_synthetic(n, 321 + dim, 0)
    st.delete_chunks([9, n // 3])
    st.build_index()
    qs = np.concatenate([synth_rows(17 + nq, 0, nq - 1, dim), synth_planted(321 + dim, 5, [n - 2], dim)])
    qs[0] *= np.float32(1e5)
    qs[1] = 0.0
    qs[1, 3] = -2.0        # one-hot: hi plane +-127, lo plane 0
    cos, ids, _ = _same_as_single_query_scans(st, qs, k)
    assert ids[nq - 1][0] == n - 2


def test_rows_that_stress_the_quantiser(VS, oracle, monkeypatch):
    """Outlier elements (one huge coordinate: the tile's scale collapses for everyone else), sparse rows, rows of
    wildly different magnitudes, zero rows, NaN / Inf rows inside a tile, thousands of near-duplicates of the query
    (cosines inside the quantisation band of each other around every k-th place)."""
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    dim, n, nq, k = 384, 40_000, 12, 50
    rng = np.random.default_rng(2024)
    corpus = oracle.synth_rows(515, 0, n, dim).copy()
    corpus[5::97, 17] = 40.0            # outlier coordinate: unit vector ~ e_17
    corpus[7::101] = 0.0
    corpus[7::101, ::64] = 1.0          # sparse rows
    corpus[11::53] *= np.float32(1e6)
    corpus[13::59] *= np.float32(1e-12)
    corpus[300] = 0.0
    corpus[301, 9] = np.nan
    corpus[302, 10] = np.inf
    corpus[303, 11] = -np.inf
    base = oracle.synth_rows(516, 0, 1, dim)[0]
    dup = np.arange(2000, 6000)         # 4,000 rows within ~1e-3 of each other in cosine to query 0
    corpus[dup] = base[None, :] + rng.normal(0, 2e-3, (len(dup), dim)).astype(np.float32)
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.delete_chunks([2500, 2501, 39_999])
    st.build_index()
    qs = synth_rows(517, 0, nq, dim).copy()
    qs[0] = base
    qs[1] = 0.0
    qs[1, 17] = 1.0                      # a one-hot query: its own scale is as coarse as it gets
    qs[2] *= np.float32(1e-7)
    qs[3] = corpus[5] * np.float32(3.0)
    cos, ids, counts = st.search_raw(qs, k)
    for i in range(nq):
        c1, i1, n1 = st.search_raw(qs[i], k)
        assert counts[i] == n1[0] and ids[i].tolist() == i1[0].tolist() and cos[i].tobytes() == c1[0].tobytes(), i
    bad = {301, 302, 303}
    assert not (set(ids[0][: counts[0]].tolist()) & bad)
    # and against the CPU oracle for the near-duplicate query
    ok = np.array([r for r in range(n) if r not in bad and r not in (2500, 2501, 39_999)])
    ecos, eids = oracle.scan_topk(corpus[ok], qs[0], k, mode="omp")
    np.testing.assert_allclose(cos[0], ecos, atol=2e-6)
    assert set(ids[0].tolist()) <= set(dup.tolist())


@pytest.mark.parametrize("n", [1025, 1100, 1151, 1152, 1153, 2047, 2048, 2049, 9_999])
def test_tail_rows_behind_the_last_complete_tile(VS, monkeypatch, n):
    """Only complete 128-row tiles are quantised; the rows behind them are candidates outright — planted best
    matches sit in the tail, in the last complete tile and right behind phase 0."""
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    dim, nq, k = 384, 5, 7
    st = VS(None, dim)
    st.insert_synthetic(n, 31, 0)
    st.delete_chunks([n - 2])
    st.build_index()
    plant = [n - 1, n // 128 * 128 - 1, 1024, 0, n - 2]  # tail, last complete tile, first row behind phase 0, ...
    qs = synth_planted(31, 5, plant, dim)
    cos, ids, counts = _same_as_single_query_scans(st, qs, k)
    for i, p in enumerate(plant):
        if p != n - 2:
            assert ids[i][0] == p
        else:
            assert p not in ids[i].tolist()  # tombstoned in the tail


def test_tiles_are_quantised_as_they_fill_up(VS, monkeypatch):
    """Appends after a build: the partial tile of the first build is quantised by the build that sees it full;
    rows in between are served from the tail path; clear() starts over; growth keeps the copy."""
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    dim, nq, k = 384, 9, 20
    st = VS(None, dim)
    total = 0
    for step, add in enumerate([1500, 37, 91, 4000, 128, 1, 20_000]):
        st.insert_synthetic(add, 77, total)
        total += add
        st.build_index()
        qs = np.concatenate([synth_rows(5 + step, 0, nq - 2, dim), synth_planted(77, 5, [total - 1, total // 2], dim)])
        cos, ids, _ = _same_as_single_query_scans(st, qs, k)
        assert ids[nq - 2][0] == total - 1 and ids[nq - 1][0] == total // 2
    st.clear()
    st.insert_synthetic(5000, 78, 0)
    st.build_index()
    qs = np.concatenate([synth_rows(55, 0, nq - 1, dim), synth_planted(78, 5, [4999], dim)])
    cos, ids, _ = _same_as_single_query_scans(st, qs, k)
    assert ids[nq - 1][0] == 4999


def test_int8_filter_over_10m_rows_k10_and_k200(VS, monkeypatch):
    """BASELINE's 10M x 384 through the int8 copy: 8 queries k = 10 and the reference's default hybrid shape
    (9 variants, retrieval limit 200) — bit-identical to the f16 copy's answers, no overflow."""
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    dim, n = 384, 10_000_000
    st = VS(None, dim)
    st.insert_synthetic(n, 4242, 0)
    st.build_index()
    for nq, k in ((8, 10), (9, 200), (64, 10), (128, 10), (300, 10)):
        qs = np.concatenate([synth_rows(1000 + nq, 0, nq - 1, dim), synth_planted(4242, 5, [n - 5], dim)])
        cos, ids, counts = st.search_raw(qs, k)
        assert st.debug_counters()[1] == 0
        assert ids[nq - 1][0] == n - 5
        for i in (0, nq - 1):
            c1, i1, n1 = st.search_raw(qs[i], k)
            assert ids[i].tolist() == i1[0].tolist() and cos[i].tobytes() == c1[0].tobytes()


def _shared_rows(rng, n, dim, c=1.2):
    """Rows with a large common component (mean pairwise cosine ~0.59): the shape of real sentence embeddings."""
    mu = rng.normal(size=(1, dim)).astype(np.float32)
    mu /= np.linalg.norm(mu)
    x = rng.normal(size=(n, dim)).astype(np.float32) / np.sqrt(dim) + np.float32(c) * mu
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def test_common_component_is_centred_out(VS, monkeypatch):
    """The int8 copy holds u - mu (mu = the mean unit row at the first build) and q . mu is added back per query: rows
    that share most of their norm — scores squeezed into a few hundredths around 0.6 — still filter without an overflow
    and bit-identically.  Appended tiles are centred on the SAME mu; clear() takes a new one."""
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    dim, n, nq = 384, 300_000, 9
    rng = np.random.default_rng(77)
    x = _shared_rows(rng, n + 40_000, dim)
    st = VS(None, dim)
    st.insert_embeddings(x[:n])
    st.build_index()
    assert st.filter_state()[0] == 2
    qs = _shared_rows(np.random.default_rng(78), nq, dim)
    qs[0] = x[12345] * np.float32(2.0)
    for k in (10, 200):
        cos, ids, _ = _same_as_single_query_scans(st, qs, k)
        assert ids[0][0] == 12345
    st.insert_embeddings(x[n:])           # more of the same, centred on the first build's mu
    st.build_index()
    qs[1] = x[n + 777]
    cos, ids, _ = _same_as_single_query_scans(st, qs, 50)
    assert ids[1][0] == n + 777 and st.filter_state()[0] == 2 and st.filter_state()[2] == 0
    # rows pointing the other way (tiles at -2 mu after the centring: a coarser scale) that are ALSO the best matches
    # of a query and come last: the phase that meets them holds more candidates than a buffer — the f16 copy answers
    st.insert_embeddings(-x[:30_000])
    st.build_index()
    qs[3] = -x[5]
    cos, ids, counts = st.search_raw(qs, 50)
    for i in range(nq):
        c1, i1, n1 = st.search_raw(qs[i], 50)
        assert ids[i].tolist() == i1[0].tolist() and cos[i].tobytes() == c1[0].tobytes(), i
    assert ids[3][0] == n + 40_000 + 5
    st.clear()
    st.insert_embeddings(-x[:50_000])
    st.build_index()
    qs[2] = -x[4242]
    cos, ids, _ = _same_as_single_query_scans(st, qs, 20)
    assert ids[2][0] == 4242


def test_outlier_coordinates_retire_the_int8_copy_at_build(VS, monkeypatch):
    """Three coordinates eight times the others' size: a tile's scale is set by them and the band would let a tenth of
    the corpus through.  The build measures that (spread statistic) and leaves the filter on the f16 copy."""
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    dim, n, nq, k = 384, 100_000, 8, 10
    rng = np.random.default_rng(5)
    x = rng.normal(size=(n, dim)).astype(np.float32)
    x[:, [7, 100, 333]] *= np.float32(8.0)
    st = VS(None, dim)
    st.insert_embeddings(x)
    st.build_index()
    copy, spread, reruns = st.filter_state()
    assert copy == 1 and spread > 7.0, (copy, spread)
    assert st.filter_copies()[1]          # the build found the int8 copy unfit and made the f16 one
    qs = rng.normal(size=(nq, dim)).astype(np.float32)
    qs[:, [7, 100, 333]] *= np.float32(8.0)
    qs[0] = x[999]
    cos, ids, _ = _same_as_single_query_scans(st, qs, k)
    assert ids[0][0] == 999
    iso = VS(None, dim)
    iso.insert_synthetic(50_000, 3, 0)
    iso.build_index()
    copy, spread, _ = iso.filter_state()
    assert copy == 2 and 2.5 < spread < 5.5, (copy, spread)
    has8, has16, nbytes = iso.filter_copies()   # the int8 copy serves: no f16 copy is built (5 bytes per element, not 7)
    assert has8 and not has16 and nbytes < 50_000 * dim * 2, (has8, has16, nbytes)


def test_two_overflows_through_the_int8_copy_retire_it(VS, oracle, monkeypatch):
    """Scores crowded inside the int8 band (thousands of near-duplicates of the query) overflow a candidate buffer: the
    host API answers that search from the f16 copy (band 0.001) instead of the list-based scan, counts a strike, and
    after two strikes the index filters on the f16 copy by itself.  Every answer stays exact."""
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    monkeypatch.setenv("CS_FILTER_INT8_MAX_SPREAD", "1000")   # keep the build-time check out of this test
    dim, n, nq, k = 384, 60_000, 4, 10
    rng = np.random.default_rng(11)
    corpus = oracle.synth_rows(41, 0, n, dim).copy()
    base = oracle.synth_rows(42, 0, 1, dim)[0]
    # 12,000 rows at cosines spread over 0.98 .. 1.0 to the query: all of them inside the int8 band of the 10th best (a
    # buffer holds 4,096), a few hundred inside the f16 copy's
    # (200 of them among the first 1,000 rows: phase 0 then leaves a bound near the top, and what a later phase lets
    # through is decided by the band, not by a bound that has not met these rows yet)
    dup = np.concatenate([np.arange(100, 300), np.arange(20_000, 31_800)])
    amp = rng.uniform(0.0, 0.2, (len(dup), 1)).astype(np.float32) * np.linalg.norm(base) / np.sqrt(dim)
    corpus[dup] = base[None, :] + amp * rng.normal(0, 1, (len(dup), dim)).astype(np.float32)
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.build_index()
    assert st.filter_state()[0] == 2 and st.filter_copies()[:2] == (True, False)
    qs = synth_rows(43, 0, nq, dim).copy()
    qs[0] = base
    for round_ in range(3):
        cos, ids, counts = st.search_raw(qs, k)
        ecos, eids = oracle.scan_topk(corpus, qs[0], k, mode="omp")
        np.testing.assert_allclose(cos[0], ecos, atol=2e-6)
        assert set(ids[0].tolist()) <= set(dup.tolist())
        for i in range(1, nq):
            c1, i1, _ = st.search_raw(qs[i], k)
            assert ids[i].tolist() == i1[0].tolist() and cos[i].tobytes() == c1[0].tobytes()
    copy, _, reruns = st.filter_state()
    assert copy == 1 and reruns == 2, (copy, reruns)   # strikes 2 of 3 searches: more than one in sixteen
    assert st.debug_counters()[1] == 0   # the f16 copy coped every time: no list-based rerun
    assert st.filter_copies()[1]         # ... built by the first search that overflowed


def test_filter_copy_allocation_failures_fall_back_without_corruption(lab_lib, VS, monkeypatch):
    """ADVICE r3 (index.hip grow): when a filter copy cannot be (re)allocated while the corpus grows, the stale smaller
    buffer must not survive.  CS_FAULT_INT8_ALLOC makes the int8 reallocation of a grow fail: the index drops the int8
    copy, the next build makes the f16 one, answers stay bit-identical to the single-query scans.  With the f16 allocation
    failing too (CS_FAULT_F16_ALLOC) searches take the exact paths — still the same bits, no error."""
    monkeypatch.setenv("CS_FILTER_SINGLE_MIN_K", "0")
    dim, nq, k = 384, 6, 10
    st = VS(None, dim)
    st.insert_synthetic(5_000, 9, 0)
    st.build_index()
    assert st.filter_state()[0] == 2 and st.filter_copies()[:2] == (True, False)
    monkeypatch.setenv("CS_FAULT_INT8_ALLOC", "1")
    st.insert_synthetic(120_000, 9, 5_000)            # grows past the first capacity: the int8 reallocation "fails"
    monkeypatch.delenv("CS_FAULT_INT8_ALLOC")
    st.build_index()
    assert st.filter_state()[0] == 1 and st.filter_copies()[:2] == (False, True)
    qs = synth_rows(10, 0, nq, dim).copy()
    qs[0] = synth_rows(9, 100_000, 1, dim)[0]
    cos, ids, _ = _same_as_single_query_scans(st, qs, k)
    assert ids[0][0] == 100_000
    # neither copy: a fresh index whose int8 copy is switched off and whose f16 copy cannot be allocated
    monkeypatch.setenv("CS_FILTER_INT8", "0")
    monkeypatch.setenv("CS_FAULT_F16_ALLOC", "1")
    bare = VS(None, dim)
    bare.insert_synthetic(50_000, 9, 0)
    bare.build_index()
    assert bare.filter_copies()[:2] == (False, False)
    qs[1] = synth_rows(9, 40_000, 1, dim)[0]
    # (the exact-f32 MFMA path sums in another order than the streaming scan: same ids, cosines to an ulp or two)
    cos, ids, counts = bare.search_raw(qs, k)
    assert ids[1][0] == 40_000
    for i in range(nq):
        c1, i1, _ = bare.search_raw(qs[i], k)
        assert ids[i].tolist() == i1[0].tolist(), i
        np.testing.assert_allclose(cos[i], c1[0], atol=1e-6)


@pytest.mark.parametrize("stage", [2, 3, 4, 5, 6])
def test_a_failing_copy_inside_grow_leaves_the_index_as_it_was(lab_lib, VS, monkeypatch, stage):
    """VERDICT r4 #13 (index.hip grow): a copy that fails between the new buffers' allocation and the pointer swap must
    neither leak them nor leave the index half-moved.  CS_FAULT_GROW_COPY=<stage> makes that stage's copy report a failure:
    the insert returns an error, free device memory is what it was, the index still answers from its old rows with the same
    bits, and the same insert succeeds once the fault is gone."""
    import torch

    from codesearch_amd import _lib

    dim, k = 384, 10
    st = VS(None, dim)
    st.insert_synthetic(5_000, 21, 0)
    st.build_index()
    q = synth_rows(22, 0, 1, dim)[0]
    cos0, ids0, _ = st.search_raw(q, k)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    monkeypatch.setenv("CS_FAULT_GROW_COPY", str(stage))
    with pytest.raises(_lib.CsError) as ei:
        st.insert_synthetic(120_000, 21, 5_000)   # grows past the first capacity
    assert "growing the index" in str(ei.value) and f"stage {stage}" in str(ei.value)
    monkeypatch.delenv("CS_FAULT_GROW_COPY")
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) < (8 << 20), (torch.cuda.mem_get_info()[0], free0)  # nothing leaked
    assert len(st) == 5_000
    cos1, ids1, _ = st.search_raw(q, k)
    assert ids1.tolist() == ids0.tolist() and cos1.tobytes() == cos0.tobytes()
    st.insert_synthetic(120_000, 21, 5_000)
    st.build_index()
    assert len(st) == 125_000
    pq = synth_rows(21, 100_000, 1, dim)[0]
    _, ids2, _ = st.search_raw(pq, k)
    assert ids2[0][0] == 100_000
