"""The model registry mirror (codesearch_amd.embedder.ModelType) against the reference's own unit tests,
/root/reference/src/embed/embedder.rs:333-437, re-expressed; plus which entries this build can run.  CPU only."""
import pytest

from codesearch_amd import CsError, ModelType
from codesearch_amd.bert_params import ARCH_JINA_QKNORM, ARCH_MODERN, ARCH_NOMIC, POOL_CLS, POOL_MEAN

M = ModelType


def test_model_type_dimensions():  # embedder.rs:335-352
    for m in (M.BGESmallENV15, M.BGESmallENV15Q, M.AllMiniLML6V2, M.AllMiniLML6V2Q, M.AllMiniLML12V2,
              M.MultilingualE5Small):
        assert m.dimensions() == 384
    for m in (M.BGEBaseENV15, M.NomicEmbedTextV1, M.NomicEmbedTextV15, M.JinaEmbeddingsV2BaseCode):
        assert m.dimensions() == 768
    for m in (M.BGELargeENV15, M.MxbaiEmbedLargeV1, M.ModernBertEmbedLarge):
        assert m.dimensions() == 1024


def test_model_type_names():  # embedder.rs:355-365
    assert M.BGESmallENV15.name_str() == "BAAI/bge-small-en-v1.5"
    assert M.AllMiniLML6V2.name_str() == "sentence-transformers/all-MiniLM-L6-v2"
    assert M.JinaEmbeddingsV2BaseCode.name_str() == "jinaai/jina-embeddings-v2-base-code"


def test_default_model_and_all():  # embedder.rs:368-378
    assert M.default() is M.AllMiniLML6V2Q and M.default().dimensions() == 384
    assert len(M.all()) == 16


def test_parse():  # embedder.rs:381-427
    expect = {"minilm-l6": M.AllMiniLML6V2, "minilm-l6-q": M.AllMiniLML6V2Q, "minilm-l12": M.AllMiniLML12V2,
              "minilm-l12-q": M.AllMiniLML12V2Q, "paraphrase-minilm": M.ParaphraseMLMiniLML12V2,
              "bge-small": M.BGESmallENV15, "bge-small-q": M.BGESmallENV15Q, "bge-base": M.BGEBaseENV15,
              "nomic-v1": M.NomicEmbedTextV1, "nomic-v1.5": M.NomicEmbedTextV15, "nomic-v1.5-q": M.NomicEmbedTextV15Q,
              "jina-code": M.JinaEmbeddingsV2BaseCode}
    for s, m in expect.items():
        assert M.parse(s) is m
    assert M.parse("invalid") is None
    # the second spelling of every entry (embedder.rs:178-195) and the round trip through short_name
    assert M.parse("BGESmallENV15") is M.BGESmallENV15 and M.parse("mxbaiembedlargev1") is M.MxbaiEmbedLargeV1
    for m in M.all():
        assert M.parse(m.short_name()) is m


def test_is_quantized():  # embedder.rs:430-437
    assert M.AllMiniLML6V2Q.is_quantized() and M.BGESmallENV15Q.is_quantized() and M.NomicEmbedTextV15Q.is_quantized()
    assert not M.AllMiniLML6V2.is_quantized() and not M.BGESmallENV15.is_quantized()


def test_gpu_runnable_architectures():
    """Which registry entries map onto the encoder kernels: BERT with absolute positions (head_dim 32 or 64)."""
    small = M.BGESmallENV15.bert_config()
    assert (small.hidden, small.layers, small.heads, small.intermediate, small.pooling) == (384, 12, 12, 1536, POOL_CLS)
    l6 = M.AllMiniLML6V2.bert_config()
    assert (l6.hidden, l6.layers, l6.pooling) == (384, 6, POOL_MEAN)
    base = M.BGEBaseENV15.bert_config()
    assert (base.hidden, base.layers, base.heads, base.intermediate) == (768, 12, 12, 3072)
    for m in (M.BGELargeENV15, M.MxbaiEmbedLargeV1):
        c = m.bert_config()
        assert (c.hidden, c.layers, c.heads, c.intermediate) == (1024, 24, 16, 4096)
    for m in (M.MultilingualE5Small, M.ParaphraseMLMiniLML12V2):  # BERT encoders over the XLM-R unigram vocabulary (embedder.rs:58,70)
        c = m.bert_config()
        assert (c.hidden, c.layers, c.heads, c.intermediate, c.vocab_size, c.pooling) == (384, 12, 12, 1536, 250037, POOL_MEAN)
    for m in (M.NomicEmbedTextV1, M.NomicEmbedTextV15, M.NomicEmbedTextV15Q):  # NomicBert: rotary positions, gated feed-forward
        c = m.bert_config()
        assert (c.arch, c.hidden, c.layers, c.heads, c.intermediate, c.vocab_size, c.rotary_base, c.pooling) == \
            (ARCH_NOMIC, 768, 12, 12, 3072, 30528, 1000.0, POOL_MEAN)
    c = M.JinaEmbeddingsV2BaseCode.bert_config()  # JinaBert: ALiBi, GELU-gated feed-forward, LayerNorm on Q / K rows
    assert (c.arch, c.hidden, c.layers, c.heads, c.intermediate, c.vocab_size, c.pooling) == \
        (ARCH_JINA_QKNORM, 768, 12, 12, 3072, 61056, POOL_MEAN)
    c = M.ModernBertEmbedLarge.bert_config()  # ModernBERT: pre-norm, two rotary bases, local / global layers, padded feed-forward width
    assert (c.arch, c.hidden, c.layers, c.heads, c.intermediate, c.vocab_size, c.rotary_base, c.rotary_base_local, c.local_window,
            c.global_every, c.layer_norm_eps, c.pooling) == (ARCH_MODERN, 1024, 28, 16, 2688, 50368, 160000.0, 10000.0, 64, 3, 1e-5, POOL_MEAN)
    for m in M.all():  # every registry entry has an encoder configuration
        m.bert_config()
    for m in M.all():  # what the configs produce is what the registry promises
        try:
            assert m.bert_config().hidden == m.dimensions()
        except CsError:
            pass
