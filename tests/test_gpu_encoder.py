"""Parity tests proper for the encoder: HIP kernels through the C ABI vs the CPU oracle and
the committed HF-transformers golden vectors.  Needs an MI355X.

Bar (BASELINE.json north_star): embeddings within 1e-4 of the reference path; asserted at
2e-5 against the fp32 oracle and 3e-5 against the float64 golden vectors."""
import os

import numpy as np
import pytest

from codesearch_amd.bert_params import POOL_CLS, POOL_MEAN, BertConfig, synth_params, synth_token_batch
from tests.test_gpu_scan import assert_topk_equal

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_golden.npz"))
TOL_ORACLE = 2e-5
TOL_GOLDEN = 3e-5


def case_cfg(name):
    m = GOLD[name + "/meta"]
    cfg = BertConfig(vocab_size=int(m[0]), hidden=int(m[1]), layers=int(m[2]), heads=int(m[3]),
                     intermediate=int(m[4]), max_position=int(m[5]))
    return cfg, int(m[6]), int(m[7]), int(m[8]), int(m[9]), bool(m[10])


@pytest.fixture(scope="module")
def FE(gpu_lib):
    from codesearch_amd import FastEmbedder, ModelType

    assert gpu_lib.cs_device_count() >= 1
    return lambda cfg, **kw: FastEmbedder(ModelType.BGESmallENV15, config=cfg, **kw)


def test_device_generated_params_match_host(FE, oracle):
    """Synthetic weights generated in HBM == the C / numpy generators: a model built from
    an uploaded block and one built from the seed give identical embeddings."""
    cfg = BertConfig(vocab_size=512, layers=2)
    ids, mask = synth_token_batch(cfg, 3, 4, 16, True)
    a = FE(cfg, seed=7).embed_ids(ids, mask)
    b = FE(cfg, params=synth_params(cfg, 7)).embed_ids(ids, mask)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"] if str(n).startswith("tiny")])
def test_tiny_cases_vs_golden_and_oracle(FE, oracle, name):
    cfg, wseed, iseed, B, L, ragged = case_cfg(name)
    ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
    params = synth_params(cfg, wseed)
    for pooling, key in ((POOL_CLS, "cls"), (POOL_MEAN, "mean")):
        cfg.pooling = pooling
        emb = FE(cfg, seed=wseed)
        got = emb.embed_ids(ids, mask)
        ref = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True)
        np.testing.assert_allclose(got, ref["pooled"], atol=TOL_ORACLE)
        np.testing.assert_allclose(got, GOLD[f"{name}/{key}"], atol=TOL_GOLDEN)
        np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)  # embedder.rs:460-463
        # full last_hidden_state of the valid tokens (pad rows are don't-care)
        hid = emb.last_hidden(B * L).reshape(B, L, cfg.hidden)
        valid = mask.astype(bool)
        np.testing.assert_allclose(hid[valid], ref["hidden"][valid], atol=1e-4)
        emb.close()


def test_full_bge_small_shape_vs_golden_and_oracle(FE, oracle):
    """12 layers, hidden 384, 12 heads, FFN 1536, vocab 30522 (BAAI/bge-small-en-v1.5)."""
    for name in ("full_dense", "full_ragged"):
        cfg, wseed, iseed, B, L, ragged = case_cfg(name)
        ids, mask = synth_token_batch(cfg, iseed, B, L, ragged)
        params = oracle.bert_synth_params(cfg, wseed)
        for pooling, key in ((POOL_CLS, "cls"), (POOL_MEAN, "mean")):
            cfg.pooling = pooling
            emb = FE(cfg, seed=wseed)
            got = emb.embed_ids(ids, mask)
            ref = oracle.bert_forward(cfg, params, ids, mask, want_hidden=True)
            np.testing.assert_allclose(got, ref["pooled"], atol=TOL_ORACLE)
            np.testing.assert_allclose(got, GOLD[f"{name}/{key}"], atol=TOL_GOLDEN)
            hid = emb.last_hidden(B * L).reshape(B, L, cfg.hidden)
            valid = mask.astype(bool)
            np.testing.assert_allclose(hid[valid], ref["hidden"][valid], atol=2e-4)
            emb.close()


@pytest.mark.parametrize("L", [1, 5, 31, 32, 33, 100, 129, 257, 512])
def test_sequence_length_edges(FE, oracle, L):
    cfg = BertConfig(vocab_size=512, layers=1)
    ids, mask = synth_token_batch(cfg, 40 + L, 3, L, L > 2)
    emb = FE(cfg, seed=9)
    got = emb.embed_ids(ids, mask)
    ref = oracle.bert_forward(cfg, synth_params(cfg, 9), ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)


def test_minibatching_and_padding_invariance(FE, oracle):
    """embed_batch_chunked semantics: results do not depend on the mini-batch size, and
    batch-longest padding does not leak into a sequence's embedding."""
    cfg = BertConfig(vocab_size=512, layers=2, pooling=POOL_MEAN)
    ids, mask = synth_token_batch(cfg, 77, 37, 48, True)
    emb = FE(cfg, seed=5)
    whole = emb.embed_ids(ids, mask, batch_size=64)
    parts = emb.embed_ids(ids, mask, batch_size=8)
    np.testing.assert_allclose(whole, parts, atol=1e-6)
    n1 = int(mask[3].sum())
    alone = emb.embed_ids(ids[3:4, :n1], mask[3:4, :n1])
    np.testing.assert_allclose(whole[3], alone[0], atol=2e-6)
    ref = oracle.bert_forward(cfg, synth_params(cfg, 5), ids, mask)["pooled"]
    np.testing.assert_allclose(whole, ref, atol=TOL_ORACLE)


def test_cancel_and_errors(FE):
    from codesearch_amd import CsError, FastEmbedder, ModelType
    from codesearch_amd import embedder as E

    cfg = BertConfig(vocab_size=512, layers=1)
    emb = FE(cfg, seed=1)
    ids, mask = synth_token_batch(cfg, 1, 4, 8, False)
    E.request_shutdown(True)
    try:
        with pytest.raises(CsError) as e:  # embedder.rs:280-282
            emb.embed_ids(ids, mask)
        assert str(e.value) == "Embedding interrupted by shutdown request" and e.value.code == 4
    finally:
        E.request_shutdown(False)
    bad = ids.copy()
    bad[0, 1] = 512
    with pytest.raises(CsError) as e:
        emb.embed_ids(bad, mask)
    assert str(e.value).startswith("Failed to generate embeddings:")  # embedder.rs:289
    with pytest.raises(CsError):
        emb.embed_ids(np.zeros((1, 513), np.int32), np.ones((1, 513), np.int32))
    assert emb.embed_ids(ids[:0], mask[:0]).shape == (0, 384)  # embedder.rs:271-273
    assert emb.dimensions() == 384 and emb.model_name() == "BAAI/bge-small-en-v1.5"
    with pytest.raises(CsError):  # an encoder family the library does not know
        FastEmbedder(ModelType.BGESmallENV15, config=BertConfig(arch=9), seed=1)
    assert ModelType.BGEBaseENV15.bert_config().hidden == 768 and ModelType.MxbaiEmbedLargeV1.bert_config().heads == 16


def test_reference_semantic_ordering_sanity(FE):
    """embedder.rs:487-506 asserts only an ordering: similar inputs embed closer than
    dissimilar ones.  With synthetic weights: a sequence vs a one-token edit vs a random one."""
    cfg = BertConfig(vocab_size=512, layers=2)
    ids, mask = synth_token_batch(cfg, 5, 3, 32, False)
    ids[1] = ids[0]
    ids[1, 7] = (ids[0, 7] + 1) % 512
    e = FE(cfg, seed=3).embed_ids(ids, mask)
    assert float(e[0] @ e[1]) > float(e[0] @ e[2])


def test_config3_shape_b256_l256_property(FE, oracle):
    """BASELINE.json configs[2]: B=256, L=256 on the full model.  The oracle would need
    minutes at this size, so check size-independent properties: unit norms, batch-position
    invariance (row i alone == row i in the batch) and 8 sampled rows against the oracle."""
    cfg = BertConfig.bge_small()
    B, L = 256, 256
    ids, mask = synth_token_batch(cfg, 999, B, L, True)
    emb = FE(cfg, seed=202)
    got = emb.embed_ids(ids, mask)
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    rows = [0, 1, 77, 128, 200, 255]
    sub = emb.embed_ids(ids[rows], mask[rows])
    np.testing.assert_allclose(sub, got[rows], atol=2e-6)
    ref = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 202), ids[rows], mask[rows])["pooled"]
    np.testing.assert_allclose(got[rows], ref, atol=TOL_ORACLE)


@pytest.mark.parametrize("hidden,heads,inter,layers,B,L", [(384, 12, 1536, 2, 40, 128),    # 5,120 tokens: just over the threshold
                                                          (384, 12, 1536, 3, 256, 16),    # shortest rows the tail takes
                                                          (384, 12, 1536, 2, 9, 512),     # longest rows
                                                          (768, 12, 3072, 1, 20, 256)])   # head_dim 64
def test_cls_tail_equals_the_full_last_layer(FE, oracle, hidden, heads, inter, layers, B, L):
    """CLS-pooled models: from 4,096 tokens per mini-batch the LAST layer runs only what the embedding reads (one query
    per sequence against every key, then B compact rows through the dense layers: cls_tail.hip).  Same embedding as the
    full layer: compared with the same rows embedded in slices below the threshold (the full last layer) and with the
    oracle; ragged masks, so padded keys are masked in the one-query attention too; mean pooling never takes the tail."""
    cfg = BertConfig(vocab_size=2048, hidden=hidden, heads=heads, intermediate=inter, layers=layers, max_position=512,
                     pooling=POOL_CLS)
    ids, mask = synth_token_batch(cfg, 100 + L, B, L, True)
    mask[1, 5:] = 0                                     # a very short row
    ids[1, 5:] = 0
    emb = FE(cfg, seed=9)
    got = emb.embed_ids(ids, mask, batch_size=B)        # one mini-batch of B x L >= 4,096 tokens: the tail
    step = max(1, 2048 // L)
    small = np.concatenate([emb.embed_ids(ids[i:i + step], mask[i:i + step], batch_size=step) for i in range(0, B, step)])
    np.testing.assert_allclose(got, small, atol=3e-6)
    rows = [0, 1, B // 2, B - 1]
    ref = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 9), ids[rows], mask[rows])["pooled"]
    np.testing.assert_allclose(got[rows], ref, atol=TOL_ORACLE)
    emb.close()
    cfg.pooling = POOL_MEAN
    emb = FE(cfg, seed=9)
    gm = emb.embed_ids(ids, mask, batch_size=B)
    refm = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 9), ids[rows], mask[rows])["pooled"]
    np.testing.assert_allclose(gm[rows], refm, atol=TOL_ORACLE)
    emb.close()


def test_cls_tail_falls_back_to_the_full_layer_and_guards_last_hidden(FE, oracle):
    """a CLS-pooled model whose rows are longer than the one-query attention kernel takes (512 keys) must run
    its last layer whole instead of failing the forward; and after a forward that DID take the tail, cs_embedder_last_hidden
    says so instead of returning the previous layer's rows."""
    from codesearch_amd._lib import CsError

    cfg = BertConfig(vocab_size=2048, layers=2, max_position=640, pooling=POOL_CLS)
    B, L = 8, 600                                       # 4,800 tokens: above the tail's threshold, L above its limit
    ids, mask = synth_token_batch(cfg, 77, B, L, True)
    emb = FE(cfg, seed=12)
    got = emb.embed_ids(ids, mask, batch_size=B)
    ref = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 12), ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    hid = emb.last_hidden(B * L)                        # the full layer ran: every token's last hidden state is there
    assert np.isfinite(hid).all()
    ids2, mask2 = synth_token_batch(cfg, 78, 32, 256, False)   # 8,192 tokens of 256: the tail runs
    emb.embed_ids(ids2, mask2, batch_size=32)
    with pytest.raises(CsError) as ei:
        emb.last_hidden(32 * 256)
    assert "CLS rows only" in str(ei.value)
    emb.close()


def test_end_to_end_index_then_search_vs_oracle_pipeline(FE, oracle):
    """BASELINE configs[3] at reduced size: chunks embedded on the GPU, appended to the
    device-resident matrix without leaving HBM, batched queries, top-10 — against the same
    pipeline built from the two oracles (encoder oracle -> scan oracle)."""
    from codesearch_amd import VectorStore
    from codesearch_amd.pipeline import index_token_chunks, search_token_queries

    cfg = BertConfig(vocab_size=2048, layers=2, pooling=POOL_CLS)
    n, L, nq, k = 1500, 32, 7, 10
    ids, mask = synth_token_batch(cfg, 4321, n, L, True)
    targets = [(i * 211) % n for i in range(nq)]
    q_ids, q_mask = ids[targets].copy(), mask[targets].copy()
    q_ids[:, 3] = (q_ids[:, 3] + 1) % cfg.vocab_size
    emb = FE(cfg, seed=17)
    store = VectorStore(None, cfg.hidden)
    index_token_chunks(emb, store, ids, mask, batch_size=256)
    assert store.is_indexed() and len(store) == n
    cos, rid, counts, _ = search_token_queries(emb, store, q_ids, q_mask, k)
    params = synth_params(cfg, 17)
    corpus = oracle.bert_forward(cfg, params, ids, mask)["pooled"]
    np.testing.assert_allclose(store.read_rows(0, n), corpus, atol=TOL_ORACLE)
    qv = oracle.bert_forward(cfg, params, q_ids, q_mask)["pooled"]
    for i in range(nq):
        ecos, eids = oracle.scan_topk(corpus, qv[i], k, mode="omp")
        assert counts[i] == k
        np.testing.assert_allclose(cos[i], ecos, atol=1e-4)  # north_star tolerance on scores
        if rid[i].tolist() != eids.tolist():  # ids may differ only where embeddings tie within tolerance
            for a, b in zip(rid[i], eids):
                if a != b:
                    assert abs(float(corpus[a] @ qv[i]) - float(corpus[b] @ qv[i])) < 5e-5


def test_config3_full_size_100k_chunks_64_queries(FE, oracle):
    """BASELINE.json configs[3] AT FULL SIZE: 100,000 synthetic chunks x 256 tokens embedded by the full 12-layer
    BGE-small-shaped encoder, appended to the device-resident matrix without leaving HBM, then 64 batched queries
    top-10.  Checked through what does not depend on size: every stored row is a unit vector; 8 sampled rows and 4
    query embeddings against the encoder oracle; every query — a one-token edit of a known chunk — retrieves its
    source chunk; and the 64 result lists against the scan oracle run over the stored matrix itself."""
    from codesearch_amd import VectorStore
    from codesearch_amd.pipeline import index_token_chunks

    cfg = BertConfig.bge_small()
    n, L, nq, k = 100_000, 256, 64, 10
    ids, mask = synth_token_batch(cfg, 31337, n, L, False)
    targets = [(i * 7919) % n for i in range(nq)]
    q_ids, q_mask = ids[targets].copy(), mask[targets].copy()
    q_ids[:, 5] = (q_ids[:, 5] + 1) % cfg.vocab_size
    emb = FE(cfg, seed=202)
    store = VectorStore(None, cfg.hidden, capacity=n)
    index_token_chunks(emb, store, ids, mask)
    assert store.is_indexed() and len(store) == n and store.next_id() == n
    corpus = store.read_rows(0, n)
    np.testing.assert_allclose(np.linalg.norm(corpus, axis=1), 1.0, atol=1e-5)
    params = oracle.bert_synth_params(cfg, 202)
    rows = [0, 255, 256, 31_337, 50_000, 65_535, 99_744, n - 1]   # first / last of mini-batches, the ragged last one
    ref = oracle.bert_forward(cfg, params, ids[rows], mask[rows])["pooled"]
    np.testing.assert_allclose(corpus[rows], ref, atol=TOL_ORACLE)
    qv = emb.embed_ids(q_ids, q_mask)
    qref = oracle.bert_forward(cfg, params, q_ids[:4], q_mask[:4])["pooled"]
    np.testing.assert_allclose(qv[:4], qref, atol=TOL_ORACLE)
    cos, rid, counts = store.search_raw(qv, k)
    assert (counts == k).all()
    assert rid[:, 0].tolist() == targets and (cos[:, 0] > 0.9).all()
    for i in range(nq):  # the search itself: exact against the oracle over the very rows the store holds
        ecos, eids = oracle.scan_topk(corpus, qv[i], k, mode="omp")
        assert_topk_equal(cos[i], rid[i], ecos, eids, corpus, qv[i], oracle)
    split, f32n, fallbacks = emb.debug_counters()
    assert split >= n // 256 and fallbacks == 0
    emb.close()
    store.close()


def test_text_entry_points_with_tokenizer(FE, oracle):
    """embed_batch / embed_one (embedder.rs:249-304) from strings: WordPiece on the host,
    batch-longest padding, encoder on the GPU — against tokenizer + encoder oracle."""
    import json

    from codesearch_amd.batch import BatchEmbedder, prepare_text
    from codesearch_amd.tokenizer import BertWordPieceTokenizer
    from codesearch_amd.vector_store import Chunk

    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tokenizer_golden.json")))
    tok = BertWordPieceTokenizer(G["vocab"])
    cfg = BertConfig(vocab_size=len(G["vocab"]), layers=2, pooling=POOL_MEAN)
    emb = FE(cfg, seed=23, tokenizer=tok)
    texts = ["fn main() { println!(\"hello world\"); }", "def calculate(a, b): return a", "struct"]
    got = emb.embed_batch(texts)
    ids, mask = tok.encode_batch(texts)
    ref = oracle.bert_forward(cfg, synth_params(cfg, 23), ids, mask)["pooled"]
    np.testing.assert_allclose(np.stack(got), ref, atol=TOL_ORACLE)
    one = emb.embed_one(texts[2])
    np.testing.assert_allclose(one, ref[2], atol=TOL_ORACLE)  # padding-invariant
    chunks = [Chunk(t, 0, 1, "Function", "a.rs", signature="fn main()") for t in texts]
    ecs = BatchEmbedder(emb).embed_chunks(chunks)
    ids2, mask2 = tok.encode_batch([prepare_text(c) for c in chunks])
    ref2 = oracle.bert_forward(cfg, synth_params(cfg, 23), ids2, mask2)["pooled"]
    np.testing.assert_allclose(np.stack([e.embedding for e in ecs]), ref2, atol=TOL_ORACLE)
    # mini-batches of 2 (cs_embedder_embed_texts tokenises batch i+1 while batch i runs): each is
    # padded to its own longest sequence, results are the same embeddings
    many = texts + ["Café naïve [SEP] résumé", "", "x" * 150, "中文 mixed"]
    chunked = np.stack(emb.embed_batch_chunked(many, 2))
    for lo in range(0, len(many), 2):
        i2, m2 = tok.encode_batch(many[lo:lo + 2])
        r2 = oracle.bert_forward(cfg, synth_params(cfg, 23), i2, m2)["pooled"]
        np.testing.assert_allclose(chunked[lo:lo + 2], r2, atol=TOL_ORACLE)
    np.testing.assert_allclose(np.stack(emb.embed_batch(many)), chunked, atol=2e-5)
    assert emb.embed_batch([]) == []
    from codesearch_amd import embedder as E
    from codesearch_amd._lib import CS_ERR_CANCELLED, CS_ERR_UNSUPPORTED, CsError
    E.request_shutdown(True)
    try:
        with pytest.raises(CsError) as ei:
            emb.embed_batch(many)
        assert ei.value.code == CS_ERR_CANCELLED and str(ei.value) == "Embedding interrupted by shutdown request"
    finally:
        E.request_shutdown(False)
    bare = FE(cfg, seed=23)
    with pytest.raises(CsError) as ei:
        bare.embed_batch(["a"])
    assert ei.value.code == CS_ERR_UNSUPPORTED


def test_text_entry_points_with_a_unigram_tokenizer(FE, oracle, tmp_path):
    """The registry's multilingual entries (multilingual-e5-small, paraphrase-multilingual-MiniLM: embedder.rs:58,70) are
    BERT encoders over a SentencePiece-unigram vocabulary: strings -> csrc/unigram.cpp on the host -> the encoder on the
    GPU, against the `tokenizers` library's ids through the encoder oracle (a small unigram model trained here:
    tests/golden/make_unigram_golden.py)."""
    pytest.importorskip("sentencepiece")
    tokenizers = pytest.importorskip("tokenizers")
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_unigram_golden as G

    from codesearch_amd.tokenizer import WordPieceTokenizer

    path = str(tmp_path / "tokenizer.json")
    hf = G.build(path, "published")
    tok = WordPieceTokenizer.from_tokenizer_json(path)
    assert tok.vocab_size() == hf.get_vocab_size() and tok.pad_id == hf.token_to_id("<pad>")
    cfg = BertConfig(vocab_size=tok.vocab_size(), layers=2, pooling=POOL_MEAN)
    emb = FE(cfg, seed=29, tokenizer=tok)
    texts = G.MULTI[:8] + ["fn main() { println!(\"hello\"); }", "", "x" * 90]
    got = np.stack(emb.embed_batch(texts))
    enc = hf.encode_batch(texts)
    L = max(len(e.ids) for e in enc)
    ids = np.full((len(texts), L), hf.token_to_id("<pad>"), np.int32)
    mask = np.zeros((len(texts), L), np.int32)
    for i, e in enumerate(enc):
        ids[i, :len(e.ids)] = e.ids
        mask[i, :len(e.ids)] = 1
    ref = oracle.bert_forward(cfg, synth_params(cfg, 29), ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    emb.close()


def test_text_pipeline_index_and_search(FE):
    """pipeline.index_text_chunks / search_text_queries: strings -> cs_embedder_embed_texts_device ->
    cs_index_add_device -> batched search; a prefix of a chunk must retrieve that chunk, and the result
    must equal the token-id pipeline on the same tokenisation."""
    from codesearch_amd import VectorStore
    from codesearch_amd.pipeline import (index_text_chunks, index_token_chunks, search_text_queries,
                                         synth_code_texts, synth_vocab)
    from codesearch_amd.tokenizer import WordPieceTokenizer

    vocab = synth_vocab(2048)
    tok = WordPieceTokenizer(vocab, max_length=64)
    cfg = BertConfig(vocab_size=2048, layers=2, max_position=64, pooling=POOL_MEAN)
    emb = FE(cfg, seed=5, tokenizer=tok)
    texts = synth_code_texts(vocab, 300, 9, mean_words=24)
    store = VectorStore(None, cfg.hidden)
    index_text_chunks(emb, store, texts, batch_size=128)
    assert store.stats().total_chunks == 0 and len(store) == 300  # vectors only: metadata is the caller's
    queries = [texts[i][: len(texts[i]) * 3 // 4] for i in (3, 150, 299)]
    cos, ids, counts, _ = search_text_queries(emb, store, queries, 5)
    assert ids[:, 0].tolist() == [3, 150, 299] and (counts == 5).all()
    # same rows as embedding consecutive mini-batches of token ids: cs_embedder_embed_texts groups a
    # window's texts by token count, and an embedding does not depend on its batch-mates or its padding
    store2 = VectorStore(None, cfg.hidden)
    for lo in range(0, 300, 128):
        i, m = tok.encode_batch(texts[lo:lo + 128])
        index_token_chunks(emb, store2, i, m)
    assert np.abs(store.read_rows(0, 300) - store2.read_rows(0, 300)).max() <= 1e-6


def test_embedder_from_hf_snapshot_directory(FE, oracle, tmp_path):
    """FastEmbedder.from_dir = cs_embedder_create_from_dir + vocab.txt tokenizer: a HF snapshot written here
    (config.json, bf16-free f32 model.safetensors with the `bert.` prefix, vocab.txt) embeds texts exactly
    like an embedder handed the same flat parameter block."""
    import json

    from safetensors.numpy import save_file

    from codesearch_amd import FastEmbedder
    from codesearch_amd.bert_params import to_state_dict
    from codesearch_amd.pipeline import synth_code_texts, synth_vocab
    from codesearch_amd.tokenizer import WordPieceTokenizer

    vocab = synth_vocab(1024)
    cfg = BertConfig(vocab_size=1024, layers=2, max_position=64, pooling=POOL_CLS)
    flat = synth_params(cfg, 77)
    d = tmp_path / "snapshot"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({
        "model_type": "bert", "vocab_size": 1024, "hidden_size": 384, "num_hidden_layers": 2,
        "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 64,
        "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu"}))
    save_file({"bert." + k: np.ascontiguousarray(v) for k, v in to_state_dict(cfg, flat).items()},
              str(d / "model.safetensors"))
    (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n")
    emb = FastEmbedder.from_dir(str(d))
    assert emb.dimensions() == 384 and emb.config.layers == 2 and emb.tokenizer.vocab_size() == 1024
    texts = synth_code_texts(vocab, 9, 3, mean_words=20)
    got = np.stack(emb.embed_batch(texts))
    ref_emb = FE(cfg, params=flat, tokenizer=WordPieceTokenizer(vocab, max_length=64))
    assert np.array_equal(got, np.stack(ref_emb.embed_batch(texts)))
    ids, mask = emb.tokenizer.encode_batch(texts)
    exp = oracle.bert_forward(cfg, flat, ids, mask)["pooled"]
    np.testing.assert_allclose(got, exp, atol=TOL_ORACLE)
    emb.close()


def test_embedder_from_fastembed_cache_layout(FE, oracle, tmp_path):
    """What `with_cache_dir` (embedder.rs:218-245) really finds in fastembed's cache: config.json, the ONNX
    export under onnx/model.onnx (exporter layout: anonymous transposed MatMul weights) and tokenizer.json as
    the `tokenizers` library serialises it.  FastEmbedder.from_dir on that directory must embed texts
    bit-identically to the safetensors + vocab.txt snapshot of the same model, and agree with the oracle."""
    import json

    pytest.importorskip("tokenizers")
    from safetensors.numpy import save_file
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    from codesearch_amd import FastEmbedder
    from codesearch_amd.bert_params import to_state_dict
    from codesearch_amd.pipeline import synth_code_texts, synth_vocab
    from tests import onnx_writer

    vocab = synth_vocab(1024)
    cfg = BertConfig(vocab_size=1024, layers=2, max_position=64, pooling=POOL_MEAN)
    flat = synth_params(cfg, 79)
    sd = to_state_dict(cfg, flat)
    config = {"model_type": "bert", "vocab_size": 1024, "hidden_size": 384, "num_hidden_layers": 2,
              "num_attention_heads": 12, "intermediate_size": 1536, "max_position_embeddings": 64,
              "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu"}
    cache = tmp_path / "models--Xenova--bge-small-en-v1.5" / "snapshots" / "abc"
    (cache / "onnx").mkdir(parents=True)
    (cache / "config.json").write_text(json.dumps(config))
    (cache / "onnx" / "model.onnx").write_bytes(onnx_writer.bert_onnx(sd, cfg.layers, "matmul"))
    tk = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
    tk.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=True)
    tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                      special_tokens=[("[CLS]", vocab["[CLS]"]), ("[SEP]", vocab["[SEP]"])])
    tk.add_special_tokens(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
    tk.save(str(cache / "tokenizer.json"))
    tk.enable_truncation(max_length=64)  # for the comparison below; the saved file carries no truncation
    (cache / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": True, "model_max_length": 512}))
    snap = tmp_path / "snapshot"
    snap.mkdir()
    (snap / "config.json").write_text(json.dumps(config))
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(snap / "model.safetensors"))
    (snap / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n")

    a = FastEmbedder.from_dir(str(cache), pooling=POOL_MEAN)
    b = FastEmbedder.from_dir(str(snap), pooling=POOL_MEAN)
    assert a.tokenizer.max_length == 64 == b.tokenizer.max_length  # min(512, model_max_length, max_position)
    texts = synth_code_texts(vocab, 11, 4, mean_words=25) + ["fn main() { [SEP] }"]
    ga, gb = np.stack(a.embed_batch(texts)), np.stack(b.embed_batch(texts))
    assert np.array_equal(ga, gb)
    ids, mask = a.tokenizer.encode_batch(texts)
    for t, row, m in zip(texts, ids, mask):
        assert row[: int(m.sum())].tolist() == tk.encode(t).ids
    exp = oracle.bert_forward(cfg, flat, ids, mask)["pooled"]
    np.testing.assert_allclose(ga, exp, atol=TOL_ORACLE)
    a.close(); b.close()


def test_multi_minibatch_id_calls_group_by_length(FE, oracle):
    """cs_embedder_embed_ids over several mini-batches: rows are grouped by mask length inside a window and
    each mini-batch is cut to its longest member; row i of the result is still sequence i, equal (to f32
    rounding) to the single-mini-batch result and to the oracle; masks with holes keep their columns."""
    cfg = BertConfig(vocab_size=512, layers=2, pooling=POOL_MEAN)
    emb = FE(cfg, seed=31)
    ids, mask = synth_token_batch(cfg, 77, 21, 48, True)  # ragged prefix masks
    mask[3, 5] = 0                                        # a hole: the row's length stays its last set bit
    one = emb.embed_ids(ids, mask, batch_size=32)         # one mini-batch, as given
    many = emb.embed_ids(ids, mask, batch_size=4)         # six mini-batches, grouped by length
    assert np.abs(one - many).max() <= 1e-6
    ref = oracle.bert_forward(cfg, synth_params(cfg, 31), ids, mask)["pooled"]
    np.testing.assert_allclose(many, ref, atol=TOL_ORACLE)


def test_minibatches_are_cut_by_tokens_not_rows(FE, oracle):
    """Inside a window the length-sorted rows are cut into mini-batches that hold the token budget of `batch` rows of 256
    tokens: short rows travel in mini-batches of up to 8 x batch rows (run_window, embedder.hip).  700 rows of 3 .. 200
    tokens at batch 16: fewer forwards than rows / batch, every row still its own embedding."""
    cfg = BertConfig(vocab_size=512, layers=2, pooling=POOL_CLS)
    emb = FE(cfg, seed=33)
    n, Lmax = 700, 200
    ids, mask = synth_token_batch(cfg, 79, n, Lmax, False)
    lens = np.random.default_rng(5).integers(3, Lmax + 1, n)
    lens[:40] = 3
    for i, ln in enumerate(lens):
        mask[i, ln:] = 0
        ids[i, ln:] = 0
    emb.profile_read(reset=True)
    got = emb.embed_ids(ids, mask, batch_size=16)
    _, forwards = emb.profile_read()
    assert forwards < (n + 15) // 16 * 0.8, forwards   # 44 mini-batches by rows; the 4,096-token budget needs ~25
    pick = np.concatenate([np.arange(0, 45), np.arange(100, n, 37)])
    ref = oracle.bert_forward(cfg, synth_params(cfg, 33), ids[pick], mask[pick])["pooled"]
    np.testing.assert_allclose(got[pick], ref, atol=TOL_ORACLE)
    alone = emb.embed_ids(ids[pick[:8]], mask[pick[:8]], batch_size=32)
    assert np.abs(alone - got[pick[:8]]).max() <= 2e-6


@pytest.mark.parametrize("B,L", [(12, 128), (20, 160), (40, 128), (40, 200)])
def test_mid_size_batches_split_k_layers(FE, oracle, B, L):
    """1,100 < tokens <= 10,240: FFN-down (and out-proj up to 2,560 tokens) run as three (two from 6,144) K slices whose partial
    slabs LayerNorm sums with bias and residual (launch_gemm_split_partial + layernorm_sum_kernel): same
    embeddings as the oracle, and the same run to run (fixed summation order)."""
    cfg = BertConfig(vocab_size=512, layers=2, pooling=POOL_MEAN)
    emb = FE(cfg, seed=41)
    ids, mask = synth_token_batch(cfg, 19, B, L, True)
    got = emb.embed_ids(ids, mask, batch_size=B)
    again = emb.embed_ids(ids, mask, batch_size=B)
    assert np.array_equal(got, again)
    ref = oracle.bert_forward(cfg, synth_params(cfg, 41), ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)


@pytest.mark.parametrize("hidden,heads,inter,L", [(768, 12, 3072, 40), (768, 12, 3072, 300), (1024, 16, 4096, 130)])
def test_head_dim_64_models(FE, oracle, hidden, heads, inter, L):
    """BERT-base / BERT-large shapes (BGE-base, BGE-large, mxbai-large in the registry, embedder.rs:7-96): 64-wide
    heads on attention_shx_kernel<2> (two K/V images per head, keys staged 128 at a time), ragged masks, more
    than one key super-tile and more than one query block — against the oracle."""
    cfg = BertConfig(vocab_size=512, hidden=hidden, layers=2, heads=heads, intermediate=inter, max_position=512,
                     pooling=POOL_MEAN)
    emb = FE(cfg, seed=53)
    B = 5
    ids, mask = synth_token_batch(cfg, 29, B, L, True)
    got = emb.embed_ids(ids, mask)
    assert got.shape == (B, hidden)
    ref = oracle.bert_forward(cfg, synth_params(cfg, 53), ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    emb.set_gemm_mode("f32")  # exact-f32 kernels (attention64_kernel): the mode a range overflow falls back to
    got32 = emb.embed_ids(ids, mask)
    np.testing.assert_allclose(got32, ref, atol=TOL_ORACLE)
    assert np.abs(got32 - got).max() < 5e-6


@pytest.mark.parametrize("hidden,heads,inter", [(384, 12, 1536), (768, 12, 3072)])
@pytest.mark.parametrize("qscale,kscale", [(8.0, 8.0), (30.0, 1.0), (0.02, 0.02), (1.0, 0.001)])
def test_attention_arithmetic_under_wide_and_narrow_scores(FE, oracle, hidden, heads, inter, qscale, kscale):
    """The attention kernel's round-5 arithmetic at its edges: query / key projections scaled so that the scores spread over
    hundreds of units (the running reference jumps by far more than the 4 the lazy rescaling tolerates, tile after tile, and the
    probabilities between rescales reach 2^4) or shrink to 1e-3 (the 2^-11-scaled query halves and the unscaled residual of the
    probabilities are f16 subnormals) — 512 tokens = four super-tiles per query block, ragged masks, both head widths, the
    whole forward against the oracle at the usual 2e-5."""
    from codesearch_amd.bert_params import to_state_dict

    cfg = BertConfig(vocab_size=512, hidden=hidden, heads=heads, intermediate=inter, layers=2, pooling=POOL_MEAN)
    flat = synth_params(cfg, 77)
    sd = to_state_dict(cfg, flat)  # views into flat
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}.attention.self."
        sd[p + "query.weight"] *= qscale
        sd[p + "query.bias"] *= qscale
        sd[p + "key.weight"] *= kscale
        sd[p + "key.bias"] *= kscale
    ids, mask = synth_token_batch(cfg, 515, 3, 512, True)
    emb = FE(cfg, params=flat)
    got = emb.embed_ids(ids, mask)
    ref = oracle.bert_forward(cfg, flat, ids, mask)["pooled"]
    np.testing.assert_allclose(got, ref, atol=TOL_ORACLE)
    short = emb.embed_ids(ids[:, :40], mask[:, :40])  # the few-rows path over the same weights
    np.testing.assert_allclose(short, oracle.bert_forward(cfg, flat, ids[:, :40], mask[:, :40])["pooled"], atol=TOL_ORACLE)
    emb.close()


def test_attention_loop_forms_are_bit_identical():
    """The attention tile loop's four forms (CS_ATTN_PIPE=0..3: rolled, two key tiles in flight, the super-tile written out,
    + early fragment requests) compute the same arithmetic in the same order: one process per form (the knob is read once),
    the same embeddings bit for bit — head_dim 32 and 64, an ALiBi model, a local-window model, ragged 300-token batches."""
    import hashlib
    import subprocess
    import sys

    script = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from codesearch_amd import FastEmbedder, ModelType
from codesearch_amd.bert_params import (ARCH_JINA_QKNORM, ARCH_MODERN, POOL_MEAN, BertConfig, synth_token_batch, token_batch_with_lens)
h = hashlib.sha256()
cfgs = [BertConfig(vocab_size=512, layers=2),
        BertConfig(vocab_size=512, hidden=768, heads=12, intermediate=3072, layers=1),
        BertConfig(vocab_size=512, hidden=768, heads=12, intermediate=3072, layers=1, pooling=POOL_MEAN, arch=ARCH_JINA_QKNORM),
        BertConfig(vocab_size=512, hidden=1024, heads=16, intermediate=1536, layers=3, max_position=512, type_vocab_size=1, pooling=POOL_MEAN,
                   arch=ARCH_MODERN, layer_norm_eps=1e-5, rotary_base=160000.0, rotary_base_local=10000.0, global_every=3, local_window=64)]
for i, cfg in enumerate(cfgs):
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=31 + i)
    if cfg.arch == ARCH_MODERN:
        ids, mask = token_batch_with_lens(cfg, 7, [300, 280, 255], 300)
    else:
        ids, mask = synth_token_batch(cfg, 7, 3, 300, True)
    h.update(np.ascontiguousarray(emb.embed_ids(ids, mask)).tobytes())
    emb.close()
print("DIGEST", h.hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from codesearch_amd import _lib

    # the four forms live in the diagnostic library (CS_ATTN_PIPE is a laboratory knob); the product library holds each head
    # width's default form only — run last, through libcsgpu.so itself: the same bits again
    digests = {}
    for form in ["0", "1", "2", "3", "product"]:
        env = dict(os.environ) if form == "product" else dict(os.environ, CS_ATTN_PIPE=form, CS_LIBCSGPU=_lib.DIAG_LIB_PATH)
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[form] = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]
    assert len(set(digests.values())) == 1, digests
