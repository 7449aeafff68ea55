"""The one-launch forward of short queries (csrc/small_forward.hip) against the kernel-by-kernel path it replaces for
mini-batches of a few short sequences — the query side of `codesearch search` (EmbeddingService::embed_query /
embed_queries_batch, /root/reference/src/embed/mod.rs:164-226: one query and up to eight variants).

Bar: the SAME BITS.  The persistent kernel runs the same arithmetic (dense-layer tiles of gemm_sh_skinny_kernel, LayerNorm
of ln_row_core, attention of attention_shx_body) behind grid barriers with write-through (sc1) hand-offs instead of kernel
boundaries; any stale read of another block's data would show as a differing embedding.  Both paths are also held to the
oracle at the usual 2e-5."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _embed(emb, ids, mask, on, monkeypatch):
    monkeypatch.setenv("CS_SMALL_FORWARD", "1" if on else "0")
    return emb.embed_ids(ids, mask)


@pytest.mark.parametrize("B,L,ragged", [(1, 16, False), (1, 5, False), (9, 16, True), (3, 33, True), (2, 64, False),
                                        (1, 100, True), (7, 21, True), (12, 16, False), (1, 1, False), (4, 48, True)])
@pytest.mark.parametrize("pooling", ["cls", "mean"])
def test_one_launch_forward_gives_the_bits_of_the_kernel_by_kernel_path(monkeypatch, B, L, ragged, pooling):
    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import POOL_CLS, POOL_MEAN, synth_token_batch

    cfg = BertConfig(vocab_size=2048, layers=12, pooling=POOL_CLS if pooling == "cls" else POOL_MEAN)
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=311, device=0)
    ids, mask = synth_token_batch(cfg, 1000 + B * 64 + L, B, L, ragged)
    ref = _embed(emb, ids, mask, False, monkeypatch)
    assert emb.small_forward_counters() == (0, 0)
    for rep in range(3):
        got = _embed(emb, ids, mask, True, monkeypatch)
        assert got.tobytes() == ref.tobytes(), (rep, float(np.abs(got - ref).max()))
    ran, gave_up = emb.small_forward_counters()
    assert ran == 3 and gave_up == 0
    assert emb.debug_counters()[2] == 0
    emb.close()


def test_one_launch_forward_matches_the_oracle(monkeypatch):
    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import POOL_MEAN, synth_token_batch
    from tests.oracle_lib import load_oracle

    oracle = load_oracle()
    cfg = BertConfig(vocab_size=1024, layers=6, pooling=POOL_MEAN)
    emb = FastEmbedder(ModelType.AllMiniLML6V2, config=cfg, seed=77, device=0)
    ids, mask = synth_token_batch(cfg, 78, 9, 16, True)
    got = _embed(emb, ids, mask, True, monkeypatch)
    assert emb.small_forward_counters() == (1, 0)
    exp = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 77), ids, mask)["pooled"]
    assert float(np.abs(got - exp).max()) < 2e-5
    emb.close()


def test_one_launch_forward_under_other_gpu_work(monkeypatch):
    """The hand-offs under UNEVEN load (Guideline 16: idle chips and uniform load hide stale reads): a second thread keeps
    searches of a resident index running on its own stream while 200 one-launch forwards of changing inputs are each
    compared with the kernel-by-kernel result of the same inputs."""
    from codesearch_amd import BertConfig, FastEmbedder, ModelType, VectorStore
    from codesearch_amd.bert_params import POOL_CLS, synth_token_batch
    from codesearch_amd.synth import synth_rows

    cfg = BertConfig(vocab_size=4096, layers=12, pooling=POOL_CLS)
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=5, device=0)
    store = VectorStore(None, 384, device=0)
    store.insert_synthetic(2_000_000, 9, 0)
    store.build_index()
    stop = threading.Event()

    def load():
        q = synth_rows(10, 0, 8, 384)
        while not stop.is_set():
            store.search_raw(q, 10)

    th = threading.Thread(target=load, daemon=True)
    th.start()
    try:
        shapes = [(1, 16), (9, 16), (2, 40), (5, 24)]
        for it in range(200):
            B, L = shapes[it % len(shapes)]
            ids, mask = synth_token_batch(cfg, 9000 + it, B, L, it % 2 == 1)
            ref = _embed(emb, ids, mask, False, monkeypatch)
            got = _embed(emb, ids, mask, True, monkeypatch)
            assert got.tobytes() == ref.tobytes(), (it, B, L, float(np.abs(got - ref).max()))
    finally:
        stop.set()
        th.join()
    ran, gave_up = emb.small_forward_counters()
    assert ran == 200 and gave_up == 0
    emb.close()
    store.close()


def test_more_rows_than_the_bound_take_the_kernel_by_kernel_path(monkeypatch):
    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import POOL_CLS, synth_token_batch

    cfg = BertConfig(vocab_size=512, layers=2, pooling=POOL_CLS)
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=3, device=0)
    ids, mask = synth_token_batch(cfg, 4, 16, 64, False)  # 1,024 rows
    _embed(emb, ids, mask, True, monkeypatch)
    assert emb.small_forward_counters() == (0, 0)
    monkeypatch.setenv("CS_SMALL_FORWARD_MAX_ROWS", "64")
    ids, mask = synth_token_batch(cfg, 4, 4, 16, False)   # 64 rows: taken
    _embed(emb, ids, mask, True, monkeypatch)
    ids, mask = synth_token_batch(cfg, 4, 5, 16, False)   # 80 rows: not
    _embed(emb, ids, mask, True, monkeypatch)
    assert emb.small_forward_counters() == (1, 0)
    emb.close()
