"""Mini-batches of a few short sequences — the query side of `codesearch search` (EmbeddingService::embed_query /
embed_queries_batch, /root/reference/src/embed/mod.rs:164-226: one query and up to eight variants) — take the small path
(csrc/small_path.hip: LayerNorm as the dense layers' prologue, FFN-down in four K slices; default under 200 token rows) or,
opt-in (CS_SMALL_FORWARD=1), the same arithmetic as ONE kernel launch (csrc/small_forward.hip).

Bars: the small path against the oracle at the usual 2e-5 (and against the general small-batch kernels, CS_SMALL_PATH=0,
which sum FFN-down in another order); the one-launch form against the small path BIT FOR BIT — it runs the same arithmetic
behind grid barriers with write-through (sc1) hand-offs instead of kernel boundaries, so any stale read of another block's
data would show as a differing embedding."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _one_launch_forward_lives_in_the_diagnostic_library(lab_lib):
    """small_forward.hip is built into libcsgpu_diag.so only (it is bit-identical to the launch chain and slower): every
    embedder of this module is created through that library."""
    yield


@pytest.fixture(scope="module")
def oracle_lib_fixture():
    from tests.oracle_lib import load_oracle

    return load_oracle()


def _embed(emb, ids, mask, on, monkeypatch):
    # (the one-launch form repeats the bits of the launch chain with attention as its own launch: CS_SMALL_FUSE=0)
    monkeypatch.setenv("CS_SMALL_FUSE", "0")
    monkeypatch.setenv("CS_SMALL_FORWARD", "1" if on else "0")
    return emb.embed_ids(ids, mask)


@pytest.mark.parametrize("B,L,ragged", [(1, 16, False), (1, 5, False), (2, 16, True), (4, 8, True), (1, 32, False), (1, 1, False),
                                        (3, 10, True), (2, 7, True), (8, 4, True), (1, 17, False), (9, 16, True), (12, 16, False),
                                        (6, 11, True), (5, 31, True), (6, 32, False), (17, 11, True), (7, 21, True), (19, 10, True)])
@pytest.mark.parametrize("pooling", ["cls", "mean"])
def test_attention_inside_the_out_projection_matches_the_two_launches(monkeypatch, oracle_lib_fixture, B, L, ragged, pooling):
    """Sequences of up to 32 tokens of a 384-wide, 12-head model run E3 + E4 as ONE launch (sp_attn_proj_kernel: every
    out-projection block computes its rows' attention itself, 16 x 16 x 32 tiles, another summation order than
    attention_shx_kernel): within 2e-6 of the two-launch chain (CS_SMALL_FUSE=0) and within 2e-5 of the oracle — rows of
    several short sequences in one 16-row tile (a key counts only inside the query's sequence), padded keys, tiles whose
    sequences span 33 and more token rows (6 x 11: the four-tile form), sequences of 17-32 tokens, up to 190 rows."""
    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import POOL_CLS, POOL_MEAN, synth_token_batch

    oracle = oracle_lib_fixture
    cfg = BertConfig(vocab_size=2048, layers=4, pooling=POOL_CLS if pooling == "cls" else POOL_MEAN)
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=312, device=0)
    ids, mask = synth_token_batch(cfg, 2000 + B * 64 + L, B, L, ragged)
    monkeypatch.setenv("CS_SMALL_FORWARD", "0")
    monkeypatch.setenv("CS_SMALL_FUSE", "0")
    two = emb.embed_ids(ids, mask)
    monkeypatch.setenv("CS_SMALL_FUSE", "1")
    one = emb.embed_ids(ids, mask)
    again = emb.embed_ids(ids, mask)
    assert one.tobytes() == again.tobytes()
    assert float(np.abs(one - two).max()) < 2e-6, float(np.abs(one - two).max())
    exp = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 312), ids, mask)["pooled"]
    assert float(np.abs(one - exp).max()) < 2e-5
    assert emb.debug_counters()[2] == 0
    emb.close()


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


@pytest.mark.parametrize("hidden,heads,B,L", [(384, 12, 9, 16), (384, 12, 1, 7), (768, 12, 3, 20), (1024, 16, 2, 33)])
def test_small_path_and_one_launch_forward_match_the_oracle(monkeypatch, hidden, heads, B, L):
    """384-, 768- and 1024-wide models (head_dim 32 and 64): the small path against the oracle and against the general
    small-batch kernels; the one-launch form (384-wide models only: elsewhere the switch changes nothing) the same bits."""
    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import POOL_MEAN, synth_token_batch
    from tests.oracle_lib import load_oracle

    oracle = load_oracle()
    cfg = BertConfig(vocab_size=1024, hidden=hidden, heads=heads, intermediate=4 * hidden, layers=3, pooling=POOL_MEAN)
    mt = {384: ModelType.AllMiniLML6V2, 768: ModelType.BGEBaseENV15, 1024: ModelType.BGELargeENV15}[hidden]
    emb = FastEmbedder(mt, config=cfg, seed=77, device=0)
    ids, mask = synth_token_batch(cfg, 78, B, L, True)
    exp = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 77), ids, mask)["pooled"]
    small = _embed(emb, ids, mask, False, monkeypatch)
    assert float(np.abs(small - exp).max()) < 2e-5
    one = _embed(emb, ids, mask, True, monkeypatch)
    assert one.tobytes() == small.tobytes()
    assert emb.small_forward_counters() == ((1, 0) if hidden == 384 else (0, 0))
    monkeypatch.setenv("CS_SMALL_PATH", "0")
    general = _embed(emb, ids, mask, False, monkeypatch)
    monkeypatch.delenv("CS_SMALL_PATH")
    assert float(np.abs(general - exp).max()) < 2e-5 and float(np.abs(general - small).max()) < 2e-6
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


def test_200_rows_and_more_take_the_general_kernels(monkeypatch):
    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import POOL_CLS, synth_token_batch

    cfg = BertConfig(vocab_size=512, layers=2, pooling=POOL_CLS)
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=3, device=0)
    ids, mask = synth_token_batch(cfg, 4, 16, 64, False)  # 1,024 rows
    big = _embed(emb, ids, mask, True, monkeypatch)
    assert emb.small_forward_counters() == (0, 0)
    monkeypatch.setenv("CS_SMALL_PATH", "0")
    assert _embed(emb, ids, mask, True, monkeypatch).tobytes() == big.tobytes()  # the switch changes nothing up there
    monkeypatch.delenv("CS_SMALL_PATH")
    ids, mask = synth_token_batch(cfg, 4, 12, 16, False)  # 192 rows: the small path, and its one-launch form on request
    _embed(emb, ids, mask, True, monkeypatch)
    ids, mask = synth_token_batch(cfg, 4, 13, 16, False)  # 208 rows: not
    _embed(emb, ids, mask, True, monkeypatch)
    assert emb.small_forward_counters() == (1, 0)
    emb.close()
