"""Indexing-side entry points added around the encoder: the submission queue for the reference's 32-chunk call shape
(cs_embedder_submit_* / cs_embedder_wait*, /root/reference/src/embed/batch.rs:84-115) and the encoder replica set with
its index loop over a row-sharded store (cs_embedders_*, /root/reference/src/index/mod.rs:626-762; SURVEY.md §8e:
replicas only, no collective).  Needs an MI355X; every multi-GPU structure is exercised with several shards / replicas
placed on device 0, and over the distinct devices of the box where there are more."""
import threading

import numpy as np
import pytest

from codesearch_amd.bert_params import POOL_CLS, POOL_MEAN, BertConfig, synth_token_batch

pytestmark = pytest.mark.gpu

TOL_ORACLE = 2e-5
TOL_BATCHING = 2e-6   # the same row embedded in another mini-batch: f32 rounding only (test_gpu_encoder asserts 1e-6..2e-6)


@pytest.fixture(scope="module")
def libs(gpu_lib):
    assert gpu_lib.cs_device_count() >= 1
    import codesearch_amd as ca

    return ca


def test_queued_32_chunk_calls_equal_one_large_call(libs, oracle):
    """Eight slices of 32 submitted, then collected: ONE 256-row forward ran (debug counters), every ticket returns
    its own rows in its own order, equal to the single-call result up to batching noise and to the oracle."""
    cfg = BertConfig(vocab_size=2048, layers=2, pooling=POOL_CLS)
    emb = libs.FastEmbedder(libs.ModelType.BGESmallENV15, config=cfg, seed=11)
    ids, mask = synth_token_batch(cfg, 77, 256, 64, True)
    whole = emb.embed_ids(ids, mask)
    f0 = emb.debug_counters()[0]
    tickets = [emb.submit_ids(ids[lo:lo + 32], mask[lo:lo + 32]) for lo in range(0, 256, 32)]
    assert emb.queued_rows() == 256 and tickets == sorted(tickets)
    got = {t: emb.wait(t) for t in reversed(tickets)}            # collected out of order: the first wait runs everything
    assert emb.debug_counters()[0] == f0 + 1 and emb.queued_rows() == 0
    for j, t in enumerate(tickets):
        np.testing.assert_allclose(got[t], whole[32 * j: 32 * j + 32], atol=TOL_BATCHING)
    ref = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 11), ids[[0, 31, 32, 255]], mask[[0, 31, 32, 255]])["pooled"]
    np.testing.assert_allclose(np.stack([got[tickets[0]][0], got[tickets[0]][31], got[tickets[1]][0], got[tickets[7]][31]]),
                               ref, atol=TOL_ORACLE)
    with pytest.raises(libs.CsError):
        emb.wait(tickets[0])                                       # a ticket is consumed by the wait that returned it
    # ragged slices, a mask with a hole, an empty submission, a discarded one; results to the device
    import torch

    holes = mask[:5].copy()
    holes[2, 3] = 0
    ta = emb.submit_ids(ids[:5], holes)
    tb = emb.submit_ids(ids[5:5], mask[5:5])
    tc = emb.submit_ids(ids[100:117], mask[100:117])
    td = emb.submit_ids(ids[:3], mask[:3])
    emb.discard(td)
    d = torch.zeros((17, cfg.hidden), dtype=torch.float32, device="cuda:0")
    emb.wait_to_device(tc, d.data_ptr())
    np.testing.assert_allclose(d.cpu().numpy(), whole[100:117], atol=TOL_BATCHING)
    np.testing.assert_allclose(emb.wait(ta), emb.embed_ids(ids[:5], holes), atol=TOL_BATCHING)
    assert emb.wait(tb).shape == (0, cfg.hidden)
    with pytest.raises(libs.CsError):
        emb.wait(td)
    with pytest.raises(libs.CsError) as e:
        bad = ids[:2].copy()
        bad[1, 1] = cfg.vocab_size
        emb.submit_ids(bad, mask[:2])
    assert "outside vocabulary" in str(e.value)
    emb.close()


def test_queue_is_safe_from_several_threads_and_survives_a_shutdown_request(libs):
    """The reference's callers are worker threads behind Arc<Mutex<FastEmbedder>> (src/embed/mod.rs:41): submit / wait
    from four threads at once; and a wait interrupted by the shutdown flag (embedder.rs:280-282) loses nothing."""
    from codesearch_amd.embedder import request_shutdown

    cfg = BertConfig(vocab_size=2048, layers=2, pooling=POOL_MEAN)
    emb = libs.FastEmbedder(libs.ModelType.BGESmallENV15, config=cfg, seed=12)
    ids, mask = synth_token_batch(cfg, 78, 4 * 6 * 16, 48, True)
    whole = emb.embed_ids(ids, mask)
    errors = []

    def worker(w):
        try:
            for rep in range(6):
                lo = (w * 6 + rep) * 16
                t = emb.submit_ids(ids[lo:lo + 16], mask[lo:lo + 16])
                np.testing.assert_allclose(emb.wait(t), whole[lo:lo + 16], atol=TOL_BATCHING)
        except Exception as ex:  # pragma: no cover
            errors.append(ex)

    th = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    t = emb.submit_ids(ids[:40], mask[:40])
    request_shutdown(True)
    try:
        with pytest.raises(libs.CsError) as e:
            emb.wait(t)
        assert str(e.value) == "Embedding interrupted by shutdown request"
    finally:
        request_shutdown(False)
    assert emb.queued_rows() == 40
    np.testing.assert_allclose(emb.wait(t), whole[:40], atol=TOL_BATCHING)
    emb.close()


def test_batch_embedder_keeps_the_reference_call_shape_on_the_queue(libs, oracle):
    """BatchEmbedder::embed_chunks (batch.rs:84-115) with its 32-chunk slices, from strings: same vectors as one
    embed_batch call, chunk order kept."""
    from codesearch_amd.batch import BatchEmbedder, prepare_text
    from codesearch_amd.pipeline import synth_code_texts, synth_vocab
    from codesearch_amd.tokenizer import WordPieceTokenizer

    vocab = synth_vocab(2048)
    tok = WordPieceTokenizer(vocab, max_length=96)
    cfg = BertConfig(vocab_size=2048, layers=2, max_position=96, pooling=POOL_CLS)
    emb = libs.FastEmbedder(libs.ModelType.BGESmallENV15, config=cfg, seed=13, tokenizer=tok)
    chunks = [libs.Chunk(t, i, i + 3, "Function", f"f{i % 7}.rs") for i, t in enumerate(synth_code_texts(vocab, 150, 5, 30))]
    f0 = emb.debug_counters()[0]
    out = BatchEmbedder(emb).embed_chunks(chunks)
    assert emb.debug_counters()[0] == f0 + 1                      # 150 chunks, five slices: one device batch
    assert [ec.chunk.content for ec in out] == [c.content for c in chunks]
    direct = np.stack(emb.embed_batch([prepare_text(c) for c in chunks]))
    np.testing.assert_allclose(np.stack([ec.embedding for ec in out]), direct, atol=TOL_BATCHING)
    emb.close()


@pytest.mark.parametrize("stripe,n", [(64, 1000), (256, 2500), (4096, 700)])
def test_replicas_index_into_shards_like_the_single_pipeline(libs, oracle, stripe, n):
    """cs_embedders_index_ids: 8 shards and 2 replicas (all on device 0: same code as 8 GPUs) must leave in the sharded
    store the rows the single-embedder / single-index pipeline leaves — same ids, rows equal up to batching noise, and
    equal to the oracle on samples — and searches over the two stores agree."""
    from codesearch_amd.pipeline import index_token_chunks

    cfg = BertConfig(vocab_size=2048, layers=2, pooling=POOL_CLS)
    L = 40
    ids, mask = synth_token_batch(cfg, 4000 + stripe, n, L, True)
    single_e = libs.FastEmbedder(libs.ModelType.BGESmallENV15, config=cfg, seed=21)
    single = libs.VectorStore(None, cfg.hidden)
    index_token_chunks(single_e, single, ids, mask)
    reps = libs.EmbedderReplicas([0, 0], config=cfg, seed=21)
    sh = libs.VectorStore(None, cfg.hidden, devices=[0] * 8, rows_per_stripe=stripe)
    first = n // 3                                                # two calls: the second starts inside a stripe
    a = reps.index_ids(sh, ids[:first], mask[:first])
    b = reps.index_ids(sh, ids[first:], mask[first:])
    assert a.tolist() == list(range(first)) and b.tolist() == list(range(first, n)) and sh.next_id() == n
    used = [c[0] + c[1] for c in reps.replica_counters()]
    assert all(u > 0 for u in used) or n <= stripe                # both replicas worked unless one stripe holds it all
    sh.build_index()
    rows, want = sh.read_rows(0, n), single.read_rows(0, n)
    np.testing.assert_allclose(rows, want, atol=TOL_BATCHING)
    samp = [0, first - 1, first, n - 1]
    ref = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 21), ids[samp], mask[samp])["pooled"]
    np.testing.assert_allclose(rows[samp], ref, atol=TOL_ORACLE)
    q = single_e.embed_ids(ids[[5, n // 2]], mask[[5, n // 2]])
    c1, i1, _ = single.search_raw(q, 10)
    c8, i8, _ = sh.search_raw(q, 10)
    assert i8[:, 0].tolist() == [5, n // 2] == i1[:, 0].tolist()
    np.testing.assert_allclose(c8, c1, atol=1e-5)
    # the non-indexing entry point: embed over both replicas, row i = input i
    np.testing.assert_allclose(reps.embed_ids(ids[:600], mask[:600]), want[:600], atol=TOL_BATCHING)
    for x in (reps, sh, single, single_e):
        x.close()


def test_replica_index_failure_leaves_the_store_untouched(libs):
    """A bad token id (or a shutdown request) in ANY replica's share: nothing is appended anywhere."""
    cfg = BertConfig(vocab_size=2048, layers=2, pooling=POOL_CLS)
    ids, mask = synth_token_batch(cfg, 1, 900, 24, False)
    reps = libs.EmbedderReplicas([0, 0], config=cfg, seed=3)
    sh = libs.VectorStore(None, cfg.hidden, devices=[0] * 4, rows_per_stripe=100)
    reps.index_ids(sh, ids[:250], mask[:250])
    lens = sh.shard_lens()
    bad = ids.copy()
    bad[777, 3] = cfg.vocab_size + 5
    with pytest.raises(libs.CsError) as e:
        reps.index_ids(sh, bad[250:], mask[250:])
    assert "outside vocabulary" in str(e.value)
    assert sh.next_id() == 250 and sh.shard_lens() == lens
    assert reps.index_ids(sh, ids[250:], mask[250:]).tolist() == list(range(250, 900))
    sh.close()
    reps.close()


def test_replicas_on_distinct_devices(libs, oracle, gpu_lib):
    """One replica and one shard per visible device (up to 8): each GPU embeds its own stripes and writes them in
    place; a store with fewer replicas than shards sends the remainder over xGMI.  One device here = the degenerate
    case."""
    ndev = min(8, int(gpu_lib.cs_device_count()))
    cfg = BertConfig(vocab_size=2048, layers=2, pooling=POOL_CLS)
    n, L = 256 * ndev + 77, 32
    ids, mask = synth_token_batch(cfg, 9, n, L, True)
    params = oracle.bert_synth_params(cfg, 33)
    want = oracle.bert_forward(cfg, params, ids[:64], mask[:64])["pooled"]
    for replicas in ([*range(ndev)], [0]):
        reps = libs.EmbedderReplicas(replicas, config=cfg, seed=33)
        sh = libs.VectorStore(None, cfg.hidden, devices=list(range(ndev)), rows_per_stripe=256)
        assert reps.index_ids(sh, ids, mask).tolist() == list(range(n))
        sh.build_index()
        np.testing.assert_allclose(sh.read_rows(0, 64), want, atol=TOL_ORACLE)
        rows = sh.read_rows(0, n)
        np.testing.assert_allclose(np.linalg.norm(rows, axis=1), 1.0, atol=1e-5)
        c, i, _ = sh.search_raw(rows[[3, n - 1]], 5)
        assert i[:, 0].tolist() == [3, n - 1]
        sh.close()
        reps.close()
