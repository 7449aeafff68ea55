"""Deleted rows are reclaimed at build time (cs_index_build; /root/reference/src/vectordb/store.rs:548-610 delete_chunks —
arroy drops deleted items at its next build — called by the incremental `index` for every changed file before it re-inserts
the file's chunks, /root/reference/src/index/mod.rs:525,544).

Bars: a store that lost half its rows answers exactly like a FRESH store of the survivors — the same rows in the same
order, so the same cosines bit for bit and the same ids through the survivors' numbering — on the streaming scan, on the
default route and on the batched filter; ids keep their meaning (never reused, store.rs:101; deleting a reclaimed id is a
no-op; rows appended afterwards continue the numbering); a second reclaim composes with the first; and the search stops
paying for the dead rows."""
import time

import numpy as np
import pytest

from codesearch_amd.synth import synth_planted, synth_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def VS(gpu_lib):
    from codesearch_amd import VectorStore

    assert gpu_lib.cs_device_count() >= 1, "no HIP device visible"
    return VectorStore


def _check_same(st, fresh, survivors, qs, k):
    """st (original ids) and fresh (ids 0 .. m-1 over the same rows in the same order): the same answers."""
    for route in (st.ROUTE_STREAM, st.ROUTE_COST):
        st.set_single_query_route(route)
        fresh.set_single_query_route(route)
        for q in qs[:3]:
            c0, i0, n0 = st.search_raw(q, k)
            c1, i1, n1 = fresh.search_raw(q, k)
            assert n0[0] == n1[0] and c0.tobytes() == c1.tobytes()
            assert i0[0][:n0[0]].tolist() == survivors[i1[0][:n1[0]]].tolist()
    c0, i0, n0 = st.search_raw(qs, k)           # batched: int8 filter + exact refine
    c1, i1, n1 = fresh.search_raw(qs, k)
    assert n0.tolist() == n1.tolist() and c0.tobytes() == c1.tobytes()
    for a, b, n in zip(i0, i1, n0):
        assert a[:n].tolist() == survivors[b[:n]].tolist()


@pytest.mark.parametrize("dim,n,k", [(384, 200_000, 10), (768, 60_000, 200), (384, 5_000, 25)])
def test_build_reclaims_deleted_rows_and_answers_like_a_fresh_store(VS, dim, n, k):
    rng = np.random.default_rng(n + dim)
    seed = 4000 + dim
    st = VS(None, dim)
    st.insert_synthetic(n, seed, 0)
    st.build_index()
    dead = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint32)
    survivors = np.setdiff1d(np.arange(n, dtype=np.uint32), dead)
    assert st.delete_chunks(dead.tolist()) == len(dead)
    assert st.stored_rows() == n and len(st) == n - len(dead)      # tombstones until the build
    st.build_index()
    assert st.stored_rows() == len(st) == len(survivors) and st.next_id() == n
    rows = synth_rows(seed, 0, n, dim)
    fresh = VS(None, dim)
    fresh.insert_embeddings(rows[survivors])
    fresh.build_index()
    qs = np.concatenate([synth_rows(seed + 1, 0, 6, dim), synth_planted(seed, seed + 2, [int(survivors[7]), int(dead[3])], dim)])
    _check_same(st, fresh, survivors, qs, k)
    planted = st.search_raw(qs[6], 1)[1][0][0]
    assert planted == survivors[7]                                  # a survivor keeps its id
    assert int(dead[3]) not in st.search_raw(qs[7], k)[1][0].tolist()
    # rows by id, wherever they live now; a reclaimed id is gone for good
    assert np.array_equal(st.read_rows(int(survivors[100]), 1)[0], rows[survivors[100]])
    with pytest.raises(Exception, match="reclaimed"):
        st.read_rows(int(dead[0]), 1)
    assert st.delete_chunks([int(dead[0]), int(dead[1])]) == 0
    # the numbering continues; a second round of deletes (over a table that is no longer the identity) composes
    more = synth_rows(seed + 9, 0, 3_000, dim)
    new_ids = st.insert_embeddings(more)
    assert new_ids.tolist() == list(range(n, n + 3_000))
    dead2 = np.concatenate([survivors[::3], np.arange(n, n + 3_000, 2, dtype=np.uint32)])
    assert st.delete_chunks(dead2.tolist()) == len(dead2)
    st.build_index()
    ids_now = np.setdiff1d(np.concatenate([survivors, np.arange(n, n + 3_000, dtype=np.uint32)]), dead2)
    assert st.stored_rows() == len(st) == len(ids_now) and st.next_id() == n + 3_000
    all_rows = np.concatenate([rows, more])
    fresh2 = VS(None, dim)
    fresh2.insert_embeddings(all_rows[ids_now])
    fresh2.build_index()
    _check_same(st, fresh2, ids_now, qs, k)
    for s in (st, fresh, fresh2):
        s.close()


def test_few_deletes_stay_tombstones(VS, monkeypatch):
    """Below the threshold (10 % of the stored rows; CS_INDEX_COMPACT_DEAD_PCT) a build moves nothing; 0 switches the reclaim off."""
    st = VS(None, 384)
    st.insert_synthetic(50_000, 77, 0)
    st.delete_chunks(list(range(0, 4_000)))
    st.build_index()
    assert st.stored_rows() == 50_000 and len(st) == 46_000
    st.delete_chunks(list(range(4_000, 5_000)))
    st.build_index()
    assert st.stored_rows() == len(st) == 45_000
    st.close()
    monkeypatch.setenv("CS_INDEX_COMPACT_DEAD_PCT", "0")
    st = VS(None, 384)
    st.insert_synthetic(50_000, 77, 0)
    st.delete_chunks(list(range(0, 40_000)))
    st.build_index()
    assert st.stored_rows() == 50_000 and len(st) == 10_000
    st.close()


def test_a_sharded_store_reclaims_per_shard(VS):
    """cs_shards_build builds every shard: each reclaims its own dead rows; ids (stripe-dealt) are untouched."""
    dim, n, k = 384, 120_000, 10
    st = VS(None, dim, devices=[0, 0, 0], rows_per_stripe=4096)
    st.insert_synthetic(n, 91, 0)
    st.build_index()
    rng = np.random.default_rng(5)
    dead = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint32)
    survivors = np.setdiff1d(np.arange(n, dtype=np.uint32), dead)
    st.delete_chunks(dead.tolist())
    st.build_index()
    assert st.stored_rows() == len(st) == len(survivors)
    fresh = VS(None, dim)
    fresh.insert_embeddings(synth_rows(91, 0, n, dim)[survivors])
    fresh.build_index()
    qs = synth_rows(92, 0, 5, dim)
    c0, i0, n0 = st.search_raw(qs, k)
    c1, i1, n1 = fresh.search_raw(qs, k)
    assert c0.tobytes() == c1.tobytes() and i0.tolist() == survivors[i1].tolist()
    st.close()
    fresh.close()


def test_searches_stop_paying_for_dead_rows(VS, monkeypatch):
    """4M rows, half of them deleted: with the reclaim switched off every search still streams all 4M; after a reclaiming
    build it streams 2M.  Streaming route (the north-star kernel), wall clock over 30 searches each."""
    dim, n, k = 384, 4_000_000, 10
    dead = list(range(0, n, 2))
    q = synth_rows(8, 0, 1, dim)[0]

    def timed(st):
        st.set_single_query_route(st.ROUTE_STREAM)
        for _ in range(5):
            out = st.search_raw(q, k)
        t0 = time.perf_counter()
        for _ in range(30):
            out = st.search_raw(q, k)
        return (time.perf_counter() - t0) / 30, out

    monkeypatch.setenv("CS_INDEX_COMPACT_DEAD_PCT", "0")
    a = VS(None, dim)
    a.insert_synthetic(n, 7, 0)
    a.delete_chunks(dead)
    a.build_index()
    t_dead, out_a = timed(a)
    assert a.stored_rows() == n
    a.close()
    monkeypatch.delenv("CS_INDEX_COMPACT_DEAD_PCT")
    b = VS(None, dim)
    b.insert_synthetic(n, 7, 0)
    b.delete_chunks(dead)
    t0 = time.perf_counter()
    b.build_index()
    t_build = time.perf_counter() - t0
    t_live, out_b = timed(b)
    assert b.stored_rows() == n // 2
    b.close()
    assert out_a[0].tobytes() == out_b[0].tobytes() and out_a[1].tolist() == out_b[1].tolist()
    print(f"\n4M x 384, half deleted: {t_dead * 1e6:.0f} us per search with tombstones, {t_live * 1e6:.0f} us after the reclaiming build "
          f"({t_live / t_dead:.2f}); that build took {t_build * 1e3:.0f} ms")
    assert t_live <= 0.6 * t_dead
