"""E8 in place (the pooling kernel "optionally writes straight into the corpus matrix row"; the index loop of
/root/reference/src/index/mod.rs:692-723 embeds a batch and inserts it): cs_index_reserve_rows hands out the address of the
next n corpus rows, cs_embedder_embed_*_device writes its pooled, normalised rows there, cs_index_commit_rows makes them
rows of the index.  Same bits as embedding to the host and inserting, no staging buffer, no device-to-device copy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pooled_rows_land_in_the_corpus(gpu_lib):
    from codesearch_amd import BertConfig, FastEmbedder, ModelType, VectorStore, _lib
    from codesearch_amd.bert_params import POOL_MEAN, synth_token_batch
    from codesearch_amd.pipeline import index_token_chunks
    from codesearch_amd.synth import synth_rows

    cfg = BertConfig(vocab_size=1024, layers=2, pooling=POOL_MEAN)
    emb = FastEmbedder(ModelType.AllMiniLML6V2, config=cfg, seed=5, device=0)
    ids, mask = synth_token_batch(cfg, 9, 300, 24, True)
    want = emb.embed_ids(ids, mask, batch_size=128)               # three mini-batches: 128 + 128 + 44
    st = VectorStore(None, 384, capacity=64)                      # the reservation has to grow the corpus, old rows kept
    first = synth_rows(3, 0, 50, 384)
    st.insert_embeddings(first)
    index_token_chunks(emb, st, ids, mask, batch_size=128)        # reserve -> embed in place -> commit -> build
    assert len(st) == 350 and st.next_id() == 350 and st.is_indexed()
    assert np.array_equal(st.read_rows(0, 50), first)
    assert st.read_rows(50, 300).tobytes() == want.tobytes()
    cos, got, cnt = st.search_raw(want[123], 3)
    assert got[0][0] == 50 + 123 and abs(cos[0][0] - 1.0) < 1e-6
    # a commit of rows that were never reserved is refused, and changes nothing
    with pytest.raises(_lib.CsError, match="not reserved"):
        st.commit_rows(1 << 20)
    assert len(st) == 350
    # the two-call form by hand, appended after a delete + reclaiming build (ids keep counting)
    st.delete_chunks(list(range(0, 200)))
    st.build_index()
    assert st.stored_rows() == 150
    emb.embed_ids_to_device(ids[:40], mask[:40], st.reserve_rows(40))
    new_ids = st.commit_rows(40)
    assert new_ids.tolist() == list(range(350, 390))
    st.build_index()
    # (the same call shape to the host: 40 sequences alone take other kernels than inside a mini-batch of 128 — last-bit differences)
    assert st.read_rows(350, 40).tobytes() == emb.embed_ids(ids[:40], mask[:40]).tobytes()
    emb.close()
    st.close()
