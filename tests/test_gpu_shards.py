"""cs_shards_* — the row-sharded VectorStore inside ONE process (what a Rust caller of
`VectorStore::search`, /root/reference/src/vectordb/store.rs:431-486, can reach) — and the non-blocking device
search it is built on.  One GPU is enough: N shards placed on device 0 run the same code as N devices
(streams, gather slots, event waits, id remap in the merge); the answer must be the single-index answer bit
for bit.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

from codesearch_amd.synth import synth_planted, synth_rows
from tests.test_gpu_scan import assert_topk_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def VS(gpu_lib):
    from codesearch_amd import VectorStore

    assert gpu_lib.cs_device_count() >= 1, "no HIP device visible"
    return VectorStore


@pytest.mark.parametrize("n,stripe,direct", [(80_000, 10_000, "1"),   # contiguous ranges: shard g = ids [g*S, (g+1)*S)
                                             (50_001, 1_000, "1"),    # seven rounds of stripes, ragged tail
                                             (50_001, 1_000, "0"),    # gather through hipMemcpyPeerAsync, forced
                                             (50_001, 1_000, ""),     # default: copy across devices, in place on the root
                                             (3_000, 4_096, "1")])    # everything on shard 0, seven empty shards
def test_eight_shards_equal_the_single_index(VS, oracle, monkeypatch, n, stripe, direct):
    if direct:
        monkeypatch.setenv("CS_SHARDS_DIRECT", direct)
    else:
        monkeypatch.delenv("CS_SHARDS_DIRECT", raising=False)
    dim, seed, shards = 384, 9090, 8
    single = VS(None, dim)
    single.insert_synthetic(n, seed, 0)
    sh = VS(None, dim, devices=[0] * shards, rows_per_stripe=stripe, capacity=n)
    assert sh.sharded and bool(sh._lib.cs_shards_direct_gather(sh.handle)) == (direct == "1")
    first = sh.insert_synthetic(n // 2, seed, 0)                      # two appends: the second starts mid-stripe
    rest = oracle.synth_rows(seed, n // 2, n - n // 2, dim)
    ids = sh.insert_embeddings(rest)                                  # host rows, ids contiguous (store.rs:659-685)
    assert first == 0 and ids.tolist() == list(range(n // 2, n)) and sh.next_id() == n
    lens = sh.shard_lens()
    assert sum(lens) == n and (stripe * shards > n or max(lens) - min(lens) <= stripe)
    assert np.array_equal(sh.read_rows(n // 2 - 3, 700), single.read_rows(n // 2 - 3, 700))
    dead = [5, n // 3, n - 1, stripe, stripe - 1] if n > 2 * stripe else [5, n - 1]
    assert sh.delete_chunks(dead + [n + 10]) == len(dead) == single.delete_chunks(dead + [n + 10])
    assert len(sh) == n - len(dead) and not sh.is_indexed()
    from codesearch_amd import CsError
    with pytest.raises(CsError) as e:
        sh.search_raw(synth_rows(1, 0, 1, dim), 10)
    assert str(e.value) == "Index not built. Call build_index() after inserting chunks."
    sh.build_index()
    single.build_index()
    planted = [0, n - 2, n // 2, min(n - 3, stripe + 1)]
    for nq in (1, 9, 1000):
        qs = np.concatenate([synth_rows(seed + nq, 0, nq - 1, dim), synth_planted(seed, 3, [planted[nq % 4]], dim)]) \
            if nq > 1 else synth_planted(seed, 3, [planted[1]], dim)
        for k in (10, 200):
            c1, i1, n1 = single.search_raw(qs, k)
            c8, i8, n8 = sh.search_raw(qs, k)
            assert n8.tolist() == n1.tolist()
            assert i8.tolist() == i1.tolist()          # global ids, (cosine desc, id asc)
            assert c8.tobytes() == c1.tobytes()
        assert i8[-1][0] == planted[nq % 4 if nq > 1 else 1]
    bitmap = np.zeros((n + 31) // 32, np.uint32)
    for d in dead:
        bitmap[d >> 5] |= np.uint32(1 << (d & 31))
    ecos, eids = oracle.scan_topk(single.read_rows(0, n), qs[0], 10, mode="omp", dead=bitmap)
    c8, i8, _ = sh.search_raw(qs[0], 10)
    assert i8[0].tolist() == eids.tolist() and np.abs(c8[0] - ecos).max() < 2e-6
    sh.clear()
    assert sh.next_id() == 0 and len(sh) == 0 and sum(sh.shard_lens()) == 0
    sh.close(); single.close()


def test_sharded_store_keeps_the_reference_surface(VS, tmp_path):
    """store.rs:846-893 on a store spread over four shards, with metadata and persistence."""
    from codesearch_amd import Chunk, EmbeddedChunk

    st = VS(tmp_path / "sh.db", 4, devices=[0, 0, 0, 0], rows_per_stripe=1)
    chunks = [EmbeddedChunk(Chunk("fn authenticate() {}", 0, 1, "Function", "auth.rs"), [1.0, 0.0, 0.0, 0.0]),
              EmbeddedChunk(Chunk("fn calculate() {}", 2, 3, "Function", "math.rs"), [0.0, 1.0, 0.0, 0.0]),
              EmbeddedChunk(Chunk("fn other() {}", 4, 5, "Function", "o.rs"), [0.0, 0.0, 1.0, 0.0])]
    assert st.insert_chunks_with_ids(chunks) == [0, 1, 2] and st.shard_lens() == [1, 1, 1, 0]
    st.build_index()
    res = st.search([0.9, 0.1, 0.0, 0.0], 2)
    assert [r.id for r in res] == [0, 1] and "authenticate" in res[0].content and res[0].score > res[1].score
    st.close()
    st2 = VS(tmp_path / "sh.db", 4, devices=[0, 0], rows_per_stripe=2)   # reopened over a different shard count
    assert st2.is_indexed() and [r.id for r in st2.search([0.0, 0.2, 0.9, 0.0], 3)] == [2, 1, 0]
    st2.close()


def test_device_search_never_waits_and_gated_rerun_is_exact(VS, oracle, gpu_lib):
    """cs_index_search_device only enqueues: (a) up to 16 queries the exact rerun is enqueued behind the
    filter path, gated on its overflow word, so an adversarial row order still yields the exact answer on the
    caller's stream with no host round trip; (b) above 16 queries the overflow is reported by
    cs_index_search_status, and a search that did not overflow reports nothing."""
    import torch

    from codesearch_amd import _lib

    n, dim, k, nq = 300_000, 384, 10, 8
    q = synth_rows(5, 0, 24, dim)
    u = synth_rows(6, 0, 1, dim)[0]
    u = u - (u @ q[0]) / (q[0] @ q[0]) * q[0]
    w = np.linspace(3.0, 0.5, n, dtype=np.float32)[:, None]
    corpus = (q[0][None, :] + w * u[None, :]).astype(np.float32)   # cos(q0, row_i) increases with i: overflow
    st = VS(None, dim)
    st.insert_embeddings(corpus)
    st.build_index()
    dev = "cuda:0"
    vp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def device_search(queries, kk):
        d_q = torch.from_numpy(np.ascontiguousarray(queries)).to(dev)
        m = len(queries)
        keys = torch.zeros((m, kk), dtype=torch.int64, device=dev)
        cos = torch.zeros((m, kk), dtype=torch.float32, device=dev)
        ids = torch.zeros((m, kk), dtype=torch.int32, device=dev)
        cnt = torch.zeros((m,), dtype=torch.int32, device=dev)
        _lib.check(gpu_lib.cs_index_search_device(st.handle, vp(d_q), m, dim, kk, vp(keys), vp(cos), vp(ids), vp(cnt), stream))
        torch.cuda.synchronize()
        return cos.cpu().numpy(), ids.cpu().numpy().astype(np.uint32), cnt.cpu().numpy()

    cos, ids, cnt = device_search(q[:nq], k)                       # (a)
    assert st.debug_counters() == (1, 1)
    for i in range(nq):  # rows are nearly collinear here: neighbours tie in f32, hence the tie-aware comparison
        ecos, eids = oracle.scan_topk(corpus, q[i], k, mode="omp")
        assert cnt[i] == k
        assert_topk_equal(cos[i], ids[i], ecos, eids, corpus, q[i], oracle)
    assert ids[0].min() >= n - 12                                  # query 0: the last rows, which the filter overflowed on
    ov = C.c_uint32(7)
    _lib.check(gpu_lib.cs_index_search_status(st.handle, stream, C.byref(ov)))
    assert ov.value == 0                                           # a gated call repaired itself: nothing to report
    cos, ids, cnt = device_search(q, k)                            # (b) 24 queries: no gated rerun
    _lib.check(gpu_lib.cs_index_search_status(st.handle, stream, C.byref(ov)))
    assert ov.value == 1
    _lib.check(gpu_lib.cs_index_search_status(st.handle, stream, C.byref(ov)))
    assert ov.value == 0                                           # reported once
    hc, hi, hn = st.search_raw(q, k)                               # the host-buffer API reruns by itself
    for i in (0, 5, 23):
        ecos, eids = oracle.scan_topk(corpus, q[i], k, mode="omp")
        assert_topk_equal(hc[i], hi[i], ecos, eids, corpus, q[i], oracle)
    st.close()
    # a benign corpus: nothing to report, and the device result is the host-API result bit for bit
    st = VS(None, dim)
    st.insert_synthetic(200_000, 31, 0)
    st.build_index()
    qs = synth_rows(32, 0, 40, dim)
    for m in (2, 16, 40):
        cos, ids, cnt = device_search(qs[:m], 25)
        hc, hi, hn = st.search_raw(qs[:m], 25)
        assert ids.tolist() == hi.tolist() and cos.tobytes() == hc.tobytes()
    _lib.check(gpu_lib.cs_index_search_status(st.handle, stream, C.byref(ov)))
    assert ov.value == 0 and st.debug_counters()[1] == 0
    st.close()


def test_concurrent_device_searches_on_one_stream_do_not_share_scratch(VS, oracle, gpu_lib):
    """Scratch is keyed by (stream, calling thread): threads that all pass the NULL stream stay independent."""
    import threading

    import torch

    from codesearch_amd import _lib

    n, dim, k = 60_000, 384, 10
    st = VS(None, dim)
    st.insert_synthetic(n, 77, 0)
    st.build_index()
    corpus = oracle.synth_rows(77, 0, n, dim)
    qs = synth_rows(78, 0, 12, dim)
    expect = [oracle.scan_topk(corpus, qs[i], k, mode="omp") for i in range(12)]
    errors = []
    vp = lambda t: C.c_void_p(t.data_ptr())

    def worker(t):
        try:
            torch.cuda.set_device(0)
            sub = qs[3 * t: 3 * t + 3]
            d_q = torch.from_numpy(np.ascontiguousarray(sub)).to("cuda:0")
            ids = torch.zeros((3, k), dtype=torch.int32, device="cuda:0")
            for _ in range(10):
                keys = torch.zeros((3, k), dtype=torch.int64, device="cuda:0")
                _lib.check(gpu_lib.cs_index_search_device(st.handle, vp(d_q), 3, dim, k, vp(keys), None, vp(ids), None, None))
                torch.cuda.synchronize()
                got = ids.cpu().numpy().astype(np.uint32)
                for j in range(3):
                    assert got[j].tolist() == expect[3 * t + j][1].tolist()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    st.close()


def test_shards_over_distinct_devices(VS, oracle, gpu_lib, monkeypatch):
    """cs_shards over min(8, cs_device_count()) DISTINCT devices (one shard on a one-GPU box): peer copies of the
    queries and keys, cross-device event waits, appends from another device's HBM.  1 / 9 / 1000 queries through the
    host API and through the asynchronous device-pointer API must equal the single index bit for bit, in both gather
    modes where a second device exists; rows appended with cs_shards_add_device from every device read back exactly."""
    import torch

    from codesearch_amd.sharded import key_unpack

    ndev = min(8, int(gpu_lib.cs_device_count()))
    devices = list(range(ndev))
    n, dim, seed, stripe = 120_000, 384, 555, 4_096
    corpus = oracle.synth_rows(seed, 0, n, dim)
    single = VS(None, dim)
    single.insert_embeddings(corpus)
    single.build_index()
    modes = ["", "1"] if ndev > 1 else [""]
    for mode in modes:
        if mode:
            monkeypatch.setenv("CS_SHARDS_DIRECT", mode)
        else:
            monkeypatch.delenv("CS_SHARDS_DIRECT", raising=False)
        sh = VS(None, dim, devices=devices, rows_per_stripe=stripe, capacity=n)
        assert sh.root_device() == 0 and bool(gpu_lib.cs_shards_direct_gather(sh.handle)) == (mode == "1")
        # appends: a third from the host, the rest from the HBM of each device in turn (cs_shards_add_device)
        third = n // 3
        assert sh.insert_embeddings(corpus[:third]).tolist() == list(range(third))
        lo, d = third, 0
        while lo < n:
            m = min(17_001, n - lo)
            t = torch.from_numpy(corpus[lo:lo + m]).to(f"cuda:{devices[d % ndev]}")
            with torch.cuda.device(devices[d % ndev]):
                ids = sh.insert_device(t.data_ptr(), m, src_device=devices[d % ndev],
                                       stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
            assert ids.tolist() == list(range(lo, lo + m))
            lo, d = lo + m, d + 1
        assert sh.next_id() == n and sum(sh.shard_lens()) == n
        assert np.array_equal(sh.read_rows(third - 5, 40_000), corpus[third - 5: third - 5 + 40_000])
        dead = [7, third + 1, n - 1]
        assert sh.delete_chunks(dead) == 3 == (single.delete_chunks(dead) if mode == modes[0] else 3)
        sh.build_index()
        if mode == modes[0]:
            single.build_index()
        torch.cuda.set_device(0)
        stream = torch.cuda.current_stream().cuda_stream
        for nq in (1, 9, 1000):
            qs = np.concatenate([synth_rows(seed + nq, 0, nq - 1, dim), synth_planted(seed, 3, [n - 2], dim)]) \
                if nq > 1 else synth_planted(seed, 3, [n - 2], dim)
            for k in (10, 200):
                c1, i1, n1 = single.search_raw(qs, k)
                c8, i8, n8 = sh.search_raw(qs, k)
                assert n8.tolist() == n1.tolist() and i8.tolist() == i1.tolist() and c8.tobytes() == c1.tobytes()
                # the same search enqueued three times back to back on the device-pointer API (no host waits between)
                d_q = torch.from_numpy(qs).to("cuda:0")
                outs = [torch.zeros((nq, k), dtype=torch.int64, device="cuda:0") for _ in range(3)]
                for o in outs:
                    sh.search_device(d_q.data_ptr(), nq, k, d_keys=o.data_ptr(), stream=stream)
                assert sh.search_status(stream) is False
                for o in outs:
                    kc, ki = key_unpack(o.cpu().numpy().view(np.uint64))
                    live = o.cpu().numpy() != 0
                    assert np.array_equal(ki[live], i1[live]) and np.array_equal(kc[live], c1[live])
            assert i8[-1][0] == n - 2
        sh.close()
    single.close()


def test_failed_append_leaves_the_sharded_store_consistent(VS, gpu_lib):
    """An append is all-or-nothing across the shards: a dimension mismatch or an id-space overflow is refused before
    any shard moved, and the stripe map (next_id, rows per shard, row contents) is what it was."""
    from codesearch_amd import CsError

    dim = 384
    sh = VS(None, dim, devices=[0, 0, 0], rows_per_stripe=100, capacity=1000)
    rows = synth_rows(9, 0, 450, dim)
    sh.insert_embeddings(rows)
    lens = sh.shard_lens()
    with pytest.raises(CsError) as e:
        sh.insert_embeddings(np.zeros((50, dim + 1), np.float32))
    assert str(e.value) == f"Embedding dimension mismatch: expected {dim}, got {dim + 1}"
    assert sh.next_id() == 450 and sh.shard_lens() == lens
    more = synth_rows(10, 0, 333, dim)
    assert sh.insert_embeddings(more).tolist() == list(range(450, 783))
    assert np.array_equal(sh.read_rows(0, 783), np.concatenate([rows, more]))
    sh.close()


def test_release_stream_frees_device_api_scratch(VS, gpu_lib):
    """cs_index_release_stream: the (stream, thread) scratch of a stream that is about to be destroyed is freed, and
    the handle keeps working — on that stream value again (new scratch) and on others."""
    import torch

    from codesearch_amd import _lib

    n, dim, k = 50_000, 384, 10
    st = VS(None, dim)
    st.insert_synthetic(n, 5, 0)
    st.build_index()
    qs = synth_rows(6, 0, 4, dim)
    want = st.search_raw(qs, k)[1]
    d_q = torch.from_numpy(qs).to("cuda:0")
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        s = torch.cuda.Stream()
        ids = torch.zeros((4, k), dtype=torch.int32, device="cuda:0")
        s.wait_stream(torch.cuda.current_stream())
        st.search_device(d_q.data_ptr(), 4, k, d_ids=ids.data_ptr(), stream=s.cuda_stream)
        _lib.check(gpu_lib.cs_index_release_stream(st.handle, C.c_void_p(s.cuda_stream)))   # synchronises s
        assert ids.cpu().numpy().astype(np.uint32).tolist() == want.tolist()
        del s
    assert st.debug_counters()[1] == 0
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20)   # nothing of three workspaces is left behind
    st.close()


def test_rank_backend_repairs_an_overflowed_batch_before_the_exchange(VS, oracle):
    """codesearch_amd/sharded.py (one process per GPU, BASELINE configs[4]): HipShardBackend.search_local asks
    cs_index_search_status after a > 16-query search and reruns an overflowed one in 16-query slices BEFORE its keys
    enter the all-gather.  Adversarial row order (cosine rising with the row number) with 24 queries; a benign corpus
    must not trigger a rerun."""
    import torch

    from codesearch_amd.sharded import ShardedVectorStore, key_unpack

    n, dim, k, nq = 300_000, 384, 10, 24
    q = synth_rows(5, 0, nq, dim)
    u = synth_rows(6, 0, 1, dim)[0]
    u = u - (u @ q[0]) / (q[0] @ q[0]) * q[0]
    w = np.linspace(3.0, 0.5, n, dtype=np.float32)[:, None]
    corpus = (q[0][None, :] + w * u[None, :]).astype(np.float32)
    sh = ShardedVectorStore(dim, n, rank=0, world=1, device=0)
    sh.store.insert_embeddings(corpus)
    sh.store.build_index()
    out = sh.search_device(torch.from_numpy(q).to("cuda:0"), nq, k)
    torch.cuda.synchronize()
    assert sh.backend.overflow_reruns == 1
    ids = out["ids"].cpu().numpy().astype(np.uint32).reshape(nq, k)
    cos = out["cos"].cpu().numpy().reshape(nq, k)
    kc, ki = key_unpack(out["keys"].cpu().numpy().view(np.uint64).reshape(nq, k))
    assert np.array_equal(ki, ids) and kc.tobytes() == cos.tobytes()
    for i in (0, 1, 15, 16, 23):
        ecos, eids = oracle.scan_topk(corpus, q[i], k, mode="omp")
        assert_topk_equal(cos[i], ids[i], ecos, eids, corpus, q[i], oracle)
    assert ids[0].min() >= n - 12
    sh.store.close()
    sh2 = ShardedVectorStore(dim, 100_000, rank=0, world=1, device=0)
    sh2.fill_synthetic(77)
    sh2.search_device(torch.from_numpy(q).to("cuda:0"), nq, k)
    torch.cuda.synchronize()
    assert getattr(sh2.backend, "overflow_reruns", 0) == 0
    sh2.store.close()
