"""Host-side mirror of the reference's FastEmbedder / ModelType over the C ABI.

Same method names and behaviour as /root/reference/src/embed/embedder.rs:7-322:
`with_cache_dir`, `embed_batch` (adaptive mini-batch 256/128/64, `CODESEARCH_BATCH_SIZE`),
`embed_batch_chunked` (shutdown poll between mini-batches), `embed_one`, `dimensions`,
`model_name`, `model_type`.  The device side starts from token ids; texts go through
cs_embedder_embed_texts with the WordPieceTokenizer (cs_tokenizer, csrc/tokenizer.cpp) the
caller attaches (the model's vocab.txt).
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import CsError, f32p, i32p
from .bert_params import ARCH_JINA_QKNORM, ARCH_MODERN, ARCH_NOMIC, POOL_CLS, POOL_MEAN, BertConfig
from .tokenizer import pack_texts


class ModelType(enum.Enum):
    """embedder.rs:7-198.  value = (short_name, HF name, dimensions, quantized)."""

    AllMiniLML6V2 = ("minilm-l6", "sentence-transformers/all-MiniLM-L6-v2", 384, False)
    AllMiniLML6V2Q = ("minilm-l6-q", "sentence-transformers/all-MiniLM-L6-v2 (quantized)", 384, True)
    AllMiniLML12V2 = ("minilm-l12", "sentence-transformers/all-MiniLM-L12-v2", 384, False)
    AllMiniLML12V2Q = ("minilm-l12-q", "sentence-transformers/all-MiniLM-L12-v2 (quantized)", 384, True)
    ParaphraseMLMiniLML12V2 = ("paraphrase-minilm", "sentence-transformers/paraphrase-MiniLM-L6-v2", 384, False)
    BGESmallENV15 = ("bge-small", "BAAI/bge-small-en-v1.5", 384, False)
    BGESmallENV15Q = ("bge-small-q", "BAAI/bge-small-en-v1.5 (quantized)", 384, True)
    BGEBaseENV15 = ("bge-base", "BAAI/bge-base-en-v1.5", 768, False)
    BGELargeENV15 = ("bge-large", "BAAI/bge-large-en-v1.5", 1024, False)
    NomicEmbedTextV1 = ("nomic-v1", "nomic-ai/nomic-embed-text-v1", 768, False)
    NomicEmbedTextV15 = ("nomic-v1.5", "nomic-ai/nomic-embed-text-v1.5", 768, False)
    NomicEmbedTextV15Q = ("nomic-v1.5-q", "nomic-ai/nomic-embed-text-v1.5 (quantized)", 768, True)
    JinaEmbeddingsV2BaseCode = ("jina-code", "jinaai/jina-embeddings-v2-base-code", 768, False)
    MultilingualE5Small = ("e5-multilingual", "intfloat/multilingual-e5-small", 384, False)
    MxbaiEmbedLargeV1 = ("mxbai-large", "mixedbread-ai/mxbai-embed-large-v1", 1024, False)
    ModernBertEmbedLarge = ("modernbert-large", "lightonai/modernbert-embed-large", 1024, False)

    @classmethod
    def default(cls) -> "ModelType":
        return cls.AllMiniLML6V2Q  # embedder.rs:12-13 (#[default])

    def dimensions(self) -> int:  # embedder.rs:76-96
        return self.value[2]

    def name_str(self) -> str:  # embedder.rs:98-117 (`name`)
        return self.value[1]

    def short_name(self) -> str:  # embedder.rs:132-151
        return self.value[0]

    def is_quantized(self) -> bool:  # embedder.rs:121-129
        return self.value[3]

    @classmethod
    def all(cls) -> List["ModelType"]:  # embedder.rs:155-174
        return list(cls)

    @classmethod
    def parse(cls, s: str) -> Optional["ModelType"]:
        """embedder.rs:177-197: short name or lower-cased enum name."""
        s = s.lower()
        for m in cls:
            if s == m.short_name() or s == _PARSE_ALIASES.get(m.name):
                return m
        return None

    def bert_config(self) -> BertConfig:
        """Encoder architecture this build can run on the GPU for the model (BERT family,
        head_dim 32 or 64).  Pooling = fastembed's default for the family (CLS for BGE, mean for
        MiniLM/E5), SURVEY.md §0 #4."""
        if self in (ModelType.BGESmallENV15, ModelType.BGESmallENV15Q):
            return BertConfig(pooling=POOL_CLS)
        if self in (ModelType.AllMiniLML6V2, ModelType.AllMiniLML6V2Q):
            return BertConfig(layers=6, pooling=POOL_MEAN)
        if self in (ModelType.ParaphraseMLMiniLML12V2, ModelType.MultilingualE5Small):
            # embedder.rs:58 maps the first entry to fastembed's ParaphraseMLMiniLML12V2 = paraphrase-multilingual-
            # MiniLM-L12-v2 whatever name() says (embedder.rs:103); both are BERT encoders (12 x 384, absolute positions)
            # over the ~250k-piece XLM-R SentencePiece vocabulary [3P-MEM: their config.json], mean pooling: the
            # tokenizer.json they ship is a unigram model, which csrc/unigram.cpp runs (cs_tokenizer_create_from_json)
            return BertConfig(vocab_size=250037, layers=12, pooling=POOL_MEAN)
        if self in (ModelType.AllMiniLML12V2, ModelType.AllMiniLML12V2Q):
            return BertConfig(layers=12, pooling=POOL_MEAN)
        if self is ModelType.BGEBaseENV15:     # BERT-base: 12 x 768, 12 heads of 64
            return BertConfig(hidden=768, layers=12, heads=12, intermediate=3072, pooling=POOL_CLS)
        if self in (ModelType.BGELargeENV15, ModelType.MxbaiEmbedLargeV1):  # BERT-large: 24 x 1024, 16 heads of 64
            return BertConfig(hidden=1024, layers=24, heads=16, intermediate=4096, pooling=POOL_CLS)
        if self in (ModelType.NomicEmbedTextV1, ModelType.NomicEmbedTextV15, ModelType.NomicEmbedTextV15Q):
            # NomicBert [3P-MEM: the model repositories' config.json]: 12 x 768, 12 heads of 64, n_inner 3072, vocab 30528
            # (bert-base-uncased's WordPiece vocabulary padded to a multiple of 64), rotary base 1000, swiglu, no position
            # table; mean pooling.  max_position = the 512 tokens fastembed's default InitOptions truncate to
            # (embedder.rs:238).  The quantised entry runs the f32 graph of its dequantised weights (encoder.hpp).
            return BertConfig(vocab_size=30528, hidden=768, layers=12, heads=12, intermediate=3072, max_position=512,
                              pooling=POOL_MEAN, arch=ARCH_NOMIC, rotary_base=1000.0)
        if self is ModelType.JinaEmbeddingsV2BaseCode:
            # JinaBert [3P-MEM: the model repository's config.json]: 12 x 768, 12 heads of 64, intermediate 3072, vocab 61056
            # (its own BPE vocabulary over code), ALiBi instead of a position table, GELU-gated feed-forward, LayerNorm on
            # the query / key rows (its auto_map names the "qk-post-norm" modelling file); mean pooling; 512 tokens as above.
            return BertConfig(vocab_size=61056, hidden=768, layers=12, heads=12, intermediate=3072, max_position=512,
                              pooling=POOL_MEAN, arch=ARCH_JINA_QKNORM)
        if self is ModelType.ModernBertEmbedLarge:
            # ModernBERT-large [3P-MEM: lightonai/modernbert-embed-large's config.json]: 28 x 1024, 16 heads of 64, intermediate
            # 2624 — carried as 2688 = 21 x 128, the kernels' tile: the extra rows / columns are zero —, vocab 50368, pre-norm,
            # rotary positions (base 160000 on every third layer, which attends globally; 10000 on the others, which see
            # 64 tokens either side), GELU-gated feed-forward, eps 1e-5; mean pooling; 512 tokens as above.
            return BertConfig(vocab_size=50368, hidden=1024, layers=28, heads=16, intermediate=2688, max_position=512,
                              type_vocab_size=1, layer_norm_eps=1e-5, pooling=POOL_MEAN, arch=ARCH_MODERN, rotary_base=160000.0,
                              rotary_base_local=10000.0, local_window=64, global_every=3)
        raise CsError(_lib.CS_ERR_UNSUPPORTED, f"Failed to initialize embedding model: {self.name_str()} has no encoder configuration")


# second spellings accepted by ModelType::parse (embedder.rs:178-195), verbatim
_PARSE_ALIASES = {
    "AllMiniLML6V2": "allminiml6v2", "AllMiniLML6V2Q": "allminiml6v2q",
    "AllMiniLML12V2": "allminiml12v2", "AllMiniLML12V2Q": "allminiml12v2q",
    "BGESmallENV15": "bgesmallenv15", "BGESmallENV15Q": "bgesmallenv15q",
    "BGEBaseENV15": "bgebaseenv15", "BGELargeENV15": "bgelargeenv15",
    "NomicEmbedTextV1": "nomicembedtextv1", "NomicEmbedTextV15": "nomicembedtextv15",
    "NomicEmbedTextV15Q": "nomicembedtextv15q", "JinaEmbeddingsV2BaseCode": "jinaembeddingsv2basecode",
    "MultilingualE5Small": "multilinguale5small", "MxbaiEmbedLargeV1": "mxbaiembedlargev1",
    "ModernBertEmbedLarge": "modernbertembedlarge",
}

_SHUTDOWN = C.c_int32(0)  # constants.rs:17-33 global shutdown flag


def request_shutdown(value: bool = True) -> None:
    _SHUTDOWN.value = 1 if value else 0


def is_shutdown_requested() -> bool:
    return bool(_SHUTDOWN.value)


class FastEmbedder:
    """embedder.rs:201-322.  `params` = flat f32 block in include/cs_bert_params.h order
    (see bert_params.from_state_dict for real checkpoints); None => synthetic weights from
    `seed`, generated on the device."""

    def __init__(self, model_type: ModelType = None, cache_dir=None, *, config: BertConfig = None,
                 params: np.ndarray = None, seed: int = 0, device: int = 0, tokenizer=None,
                 gemm_mode: Optional[str] = None, wscale: np.ndarray = None):
        self._lib = _lib.load()
        self._model_type = model_type or ModelType.default()
        self.config = config or self._model_type.bert_config()
        if cache_dir is not None:  # embedder.rs:224-229
            os.environ["FASTEMBED_CACHE_DIR"] = str(cache_dir)
        self.tokenizer = tokenizer
        ccfg = self.config.to_c()
        pptr = None
        if params is not None:
            params = np.ascontiguousarray(params, np.float32)
            need = int(self._lib.cs_bert_param_count(C.byref(ccfg)))
            if params.size != need:
                raise CsError(_lib.CS_ERR_BAD_ARG,
                              f"Failed to initialize embedding model: expected {need} parameters, got {params.size}")
            pptr = params.ctypes.data_as(f32p)
        h = C.c_void_p()
        if wscale is not None:
            # a dynamically quantised model (bert_params.quantize_linear_weights / a *Q file): params hold integer
            # multiples of the column scales wscale [layers, 5H + I]; runs as CS_GEMM_Q8_DYNAMIC ("q8") by default
            if pptr is None:
                raise CsError(_lib.CS_ERR_BAD_ARG, "Failed to initialize embedding model: a quantised model needs its parameters")
            wscale = np.ascontiguousarray(wscale, np.float32)
            _lib.check(self._lib.cs_embedder_create_quantized(C.byref(ccfg), pptr, wscale.ctypes.data_as(f32p), wscale.size,
                                                              device, C.byref(h)))
        else:
            _lib.check(self._lib.cs_embedder_create(C.byref(ccfg), pptr, seed, device, C.byref(h)))
        self._h = h
        if gemm_mode is not None:
            self.set_gemm_mode(gemm_mode)

    def set_gemm_mode(self, mode: str) -> None:
        """"split" (default: split-f16 operands on the f16 MFMA), "f32" (exact-f32 MFMA) or, for a quantised model (and
        its default), "q8": every Linear as onnxruntime's dynamic quantiser runs it (CS_GEMM_Q8_DYNAMIC)."""
        m = {"f32": _lib.CS_GEMM_F32, "split": _lib.CS_GEMM_SPLIT_F16, "q8": _lib.CS_GEMM_Q8_DYNAMIC}[mode]
        _lib.check(self._lib.cs_embedder_set_gemm_mode(self._h, m))

    def gemm_mode(self) -> str:
        """The arithmetic of the dense layers in force: "split", "f32" or, for a quantised model, "q8"."""
        return {_lib.CS_GEMM_F32: "f32", _lib.CS_GEMM_SPLIT_F16: "split", _lib.CS_GEMM_Q8_DYNAMIC: "q8"}[
            int(self._lib.cs_embedder_gemm_mode(self._h))]

    def debug_counters(self):
        """-> (split_forwards, f32_forwards, range_fallbacks)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.cs_embedder_debug_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def small_forward_counters(self):
        """-> (mini-batches that ran as ONE kernel launch (csrc/small_forward.hip: a few short sequences, the query side),
        how many of those gave up at a grid barrier and were re-run kernel by kernel).  The one-launch forward is built into
        the diagnostic library only (include/codesearch_gpu_diag.h): an embedder created through libcsgpu.so has no such path."""
        fn = getattr(self._lib, "cs_debug_small_forward_counters", None)
        if fn is None or fn.argtypes is None:
            raise RuntimeError("the one-launch forward lives in libcsgpu_diag.so only (tests: the lab_lib fixture)")
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(fn(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    @classmethod
    def from_dir(cls, model_dir: str, model_type: ModelType = None, pooling: int = -1, device: int = 0,
                 lowercase: Optional[bool] = None) -> "FastEmbedder":
        """A model directory as hf-hub leaves it in fastembed's cache — config.json, the weights as
        onnx/model.onnx | model.onnx | model_optimized.onnx (what fastembed downloads) or model.safetensors
        (the PyTorch snapshot), tokenizer.json or vocab.txt: cs_embedder_create_from_dir +
        cs_tokenizer_create_from_dir.  pooling -1 = what 1_Pooling/config.json says (CLS when absent);
        lowercase overrides the directory's own setting (vocab.txt route only)."""
        from .tokenizer import WordPieceTokenizer

        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._model_type = model_type or ModelType.default()
        ccfg = _lib.BertConfig()
        _lib.check(self._lib.cs_bert_config_from_dir(str(model_dir).encode(), pooling, C.byref(ccfg)))
        self.config = BertConfig(vocab_size=ccfg.vocab_size, hidden=ccfg.hidden, layers=ccfg.layers, heads=ccfg.heads,
                                 intermediate=ccfg.intermediate, max_position=ccfg.max_position,
                                 type_vocab_size=ccfg.type_vocab_size, layer_norm_eps=ccfg.layer_norm_eps,
                                 pooling=ccfg.pooling, arch=ccfg.arch, rotary_base=ccfg.rotary_base,
                                 rotary_base_local=ccfg.rotary_base_local, local_window=ccfg.local_window,
                                 global_every=ccfg.global_every)
        h = C.c_void_p()
        _lib.check(self._lib.cs_embedder_create_from_dir(str(model_dir).encode(), pooling, device, C.byref(h)))
        self._h = h
        d = str(model_dir)
        vocab = os.path.join(d, "vocab.txt")
        if os.path.exists(os.path.join(d, "tokenizer.json")) or (os.path.exists(vocab) and lowercase is None):
            self.tokenizer = WordPieceTokenizer.from_dir(d, max_length=self.config.max_position)
        elif os.path.exists(vocab):
            self.tokenizer = WordPieceTokenizer.from_vocab_file(vocab, lowercase=lowercase,
                                                                max_length=self.config.max_position)
        else:
            self.tokenizer = None
        return self

    # constructors named as in the reference
    @classmethod
    def new(cls, **kw) -> "FastEmbedder":
        return cls.with_model(ModelType.default(), **kw)

    @classmethod
    def with_model(cls, model_type: ModelType, **kw) -> "FastEmbedder":
        return cls.with_cache_dir(model_type, None, **kw)

    @classmethod
    def with_cache_dir(cls, model_type: ModelType, cache_dir, **kw) -> "FastEmbedder":
        return cls(model_type, cache_dir, **kw)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cs_embedder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- token-id entry points (what the device consumes) ---------------------------------
    def embed_ids(self, ids, mask, batch_size: int = 0) -> np.ndarray:
        """ids/mask [n, L] int32 -> [n, dim] float32 (pooled, L2-normalised)."""
        ids = np.ascontiguousarray(ids, np.int32)
        mask = np.ascontiguousarray(mask, np.int32)
        if ids.ndim != 2 or ids.shape != mask.shape:
            raise ValueError("ids and mask must both be [n, seq_len]")
        n, L = ids.shape
        out = np.empty((n, self.dimensions()), np.float32)
        _lib.check(self._lib.cs_embedder_embed_ids(self._h, ids.ctypes.data_as(i32p), mask.ctypes.data_as(i32p),
                                                   n, L, batch_size, out.ctypes.data_as(f32p),
                                                   C.cast(C.byref(_SHUTDOWN), i32p)))
        return out

    def embed_ids_to_device(self, ids, mask, d_out_ptr: int, batch_size: int = 0) -> None:
        """Same, leaving the [n, dim] result at the device address d_out_ptr."""
        ids = np.ascontiguousarray(ids, np.int32)
        mask = np.ascontiguousarray(mask, np.int32)
        n, L = ids.shape
        _lib.check(self._lib.cs_embedder_embed_ids_device(self._h, ids.ctypes.data_as(i32p),
                                                          mask.ctypes.data_as(i32p), n, L, batch_size,
                                                          C.c_void_p(d_out_ptr), C.cast(C.byref(_SHUTDOWN), i32p)))

    def last_hidden(self, n_tokens: int) -> np.ndarray:
        out = np.empty((n_tokens, self.dimensions()), np.float32)
        _lib.check(self._lib.cs_embedder_last_hidden(self._h, out.ctypes.data_as(f32p), n_tokens))
        return out

    # ---- text entry points (embedder.rs:249-304) --------------------------------------------
    def _require_tokenizer(self):
        if self.tokenizer is None:
            raise CsError(_lib.CS_ERR_UNSUPPORTED,
                          "Failed to generate embeddings: no tokenizer attached (pass tokenizer=...)")
        return self.tokenizer

    def embed_batch(self, texts: Sequence[str]) -> List[np.ndarray]:
        """embedder.rs:249-263: mini-batch from CODESEARCH_BATCH_SIZE or 256/128/64 by dims
        (batch 0 = that policy, applied inside cs_embedder_embed_texts)."""
        return self.embed_batch_chunked(texts, 0)

    def embed_batch_chunked(self, texts: Sequence[str], batch_size: int) -> List[np.ndarray]:
        """embedder.rs:266-295 through cs_embedder_embed_texts: each mini-batch tokenised on the
        host (padded batch-longest, as fastembed does) and run on the device, the shutdown flag
        polled between mini-batches."""
        texts = list(texts)
        if not texts:
            return []
        tok = self._require_tokenizer()
        blob, offsets = pack_texts(texts)
        emb = np.empty((len(texts), self.dimensions()), np.float32)
        _lib.check(self._lib.cs_embedder_embed_texts(self._h, tok.handle, blob, offsets.ctypes.data_as(_lib.u64p),
                                                     len(texts), batch_size, emb.ctypes.data_as(f32p),
                                                     C.cast(C.byref(_SHUTDOWN), i32p)))
        return [emb[i] for i in range(emb.shape[0])]

    def embed_texts_to_device(self, texts: Sequence[str], d_out_ptr: int, batch_size: int = 0) -> None:
        """Same, leaving the [n, dim] result at the device address d_out_ptr."""
        texts = list(texts)
        if not texts:
            return
        tok = self._require_tokenizer()
        blob, offsets = pack_texts(texts)
        _lib.check(self._lib.cs_embedder_embed_texts_device(self._h, tok.handle, blob,
                                                            offsets.ctypes.data_as(_lib.u64p), len(texts), batch_size,
                                                            C.c_void_p(d_out_ptr), C.cast(C.byref(_SHUTDOWN), i32p)))

    # ---- queueing entry points: the reference's 32-chunk call shape packed into full device batches ----
    def submit_texts(self, texts: Sequence[str]) -> int:
        """Queue a slice of texts (tokenised now, on this thread) -> ticket.  Nothing runs on the device until a
        wait needs it; then everything queued is embedded as full length-grouped mini-batches."""
        texts = list(texts)
        tok = self._require_tokenizer()
        blob, offsets = pack_texts(texts) if texts else (b"", np.zeros(1, np.uint64))
        t = C.c_uint64()
        _lib.check(self._lib.cs_embedder_submit_texts(self._h, tok.handle, blob, offsets.ctypes.data_as(_lib.u64p),
                                                      len(texts), C.byref(t)))
        self._ticket_rows = getattr(self, "_ticket_rows", {})
        self._ticket_rows[int(t.value)] = len(texts)
        return int(t.value)

    def submit_ids(self, ids, mask) -> int:
        ids = np.ascontiguousarray(ids, np.int32)
        mask = np.ascontiguousarray(mask, np.int32)
        if ids.ndim != 2 or ids.shape != mask.shape:
            raise ValueError("ids and mask must both be [n, seq_len]")
        t = C.c_uint64()
        _lib.check(self._lib.cs_embedder_submit_ids(self._h, ids.ctypes.data_as(i32p), mask.ctypes.data_as(i32p),
                                                    ids.shape[0], ids.shape[1], C.byref(t)))
        self._ticket_rows = getattr(self, "_ticket_rows", {})
        self._ticket_rows[int(t.value)] = ids.shape[0]
        return int(t.value)

    def wait(self, ticket: int) -> np.ndarray:
        """-> the ticket's [n, dim] embeddings, rows in submission order."""
        n = getattr(self, "_ticket_rows", {}).get(int(ticket), 0)
        out = np.empty((n, self.dimensions()), np.float32)
        _lib.check(self._lib.cs_embedder_wait(self._h, int(ticket), out.ctypes.data_as(f32p),
                                              C.cast(C.byref(_SHUTDOWN), i32p)))
        self._ticket_rows.pop(int(ticket), None)
        return out

    def wait_to_device(self, ticket: int, d_out_ptr: int) -> None:
        _lib.check(self._lib.cs_embedder_wait_device(self._h, int(ticket), C.c_void_p(d_out_ptr),
                                                     C.cast(C.byref(_SHUTDOWN), i32p)))
        getattr(self, "_ticket_rows", {}).pop(int(ticket), None)

    def discard(self, ticket: int) -> None:
        _lib.check(self._lib.cs_embedder_discard(self._h, int(ticket)))
        getattr(self, "_ticket_rows", {}).pop(int(ticket), None)

    def queued_rows(self) -> int:
        return int(self._lib.cs_embedder_queued_rows(self._h))

    def embed_one(self, text: str) -> np.ndarray:
        """embedder.rs:298-304."""
        r = self.embed_batch([text])
        if not r:
            raise CsError(_lib.CS_ERR_BAD_ARG, "No embedding generated")
        return r[0]

    def dimensions(self) -> int:
        return int(self._lib.cs_embedder_dim(self._h))

    def model_name(self) -> str:
        return self._model_type.name_str()

    def model_type(self) -> ModelType:
        return self._model_type

    def profile_read(self, reset: bool = True):
        """-> (forward_ms_total, forwards)"""
        ms, n = C.c_double(), C.c_uint64()
        _lib.check(self._lib.cs_embedder_profile_read(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, int(n.value)

    STAGES = ("embed_ln", "qkv_gemm", "attention", "out_proj_gemm", "layernorm_attn", "ffn_up_gemm",
              "ffn_down_gemm", "layernorm_ffn", "pool_normalize")  # CS_STAGE_* of codesearch_gpu.h

    def profile_stages(self, enable: bool) -> None:
        """While on, a forward runs on one stream with a HIP event after every kernel."""
        _lib.check(self._lib.cs_embedder_profile_stages(self._h, 1 if enable else 0))

    def profile_stages_read(self, reset: bool = True):
        """-> ({stage: microseconds per forward}, forwards)"""
        us = (C.c_double * len(self.STAGES))()
        n = C.c_uint64()
        _lib.check(self._lib.cs_embedder_profile_stages_read(self._h, us, C.byref(n), 1 if reset else 0))
        f = max(int(n.value), 1)
        return {name: us[i] / f for i, name in enumerate(self.STAGES)}, int(n.value)


class EmbedderReplicas:
    """cs_embedders_*: one encoder replica per GPU inside this process (replicas only, no
    collective), with the reference's index loop — embed_chunks then insert_chunks_with_ids,
    /root/reference/src/index/mod.rs:692-723 — over a row-sharded VectorStore: every replica embeds the chunks whose
    ids fall on the shards of its own device and the rows are appended without leaving HBM."""

    def __init__(self, devices: Sequence[int], model_type: ModelType = None, *, config: BertConfig = None,
                 params: np.ndarray = None, seed: int = 0, tokenizer=None, model_dir: str = None, pooling: int = -1):
        self._lib = _lib.load()
        self._model_type = model_type or ModelType.default()
        self.tokenizer = tokenizer
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        if model_dir is not None:
            _lib.check(self._lib.cs_embedders_create_from_dir(str(model_dir).encode(), pooling, devs, len(devices),
                                                              C.byref(h)))
            self.config = None
        else:
            self.config = config or self._model_type.bert_config()
            ccfg = self.config.to_c()
            pptr = None
            if params is not None:
                params = np.ascontiguousarray(params, np.float32)
                pptr = params.ctypes.data_as(f32p)
            _lib.check(self._lib.cs_embedders_create(C.byref(ccfg), pptr, seed, devs, len(devices), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cs_embedders_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._lib.cs_embedders_count(self._h))

    def dimensions(self) -> int:
        return int(self._lib.cs_embedders_dim(self._h))

    def replica_counters(self):
        """-> [(split_forwards, f32_forwards, range_fallbacks)] per replica."""
        out = []
        for i in range(len(self)):
            a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
            r = C.c_void_p(self._lib.cs_embedders_replica(self._h, i))
            _lib.check(self._lib.cs_embedder_debug_counters(r, C.byref(a), C.byref(b), C.byref(c)))
            out.append((int(a.value), int(b.value), int(c.value)))
        return out

    def embed_ids(self, ids, mask, batch_size: int = 0) -> np.ndarray:
        ids = np.ascontiguousarray(ids, np.int32)
        mask = np.ascontiguousarray(mask, np.int32)
        n, L = ids.shape
        out = np.empty((n, self.dimensions()), np.float32)
        _lib.check(self._lib.cs_embedders_embed_ids(self._h, ids.ctypes.data_as(i32p), mask.ctypes.data_as(i32p), n, L,
                                                    batch_size, out.ctypes.data_as(f32p), C.cast(C.byref(_SHUTDOWN), i32p)))
        return out

    def embed_batch(self, texts: Sequence[str], batch_size: int = 0) -> List[np.ndarray]:
        texts = list(texts)
        if not texts:
            return []
        blob, offsets = pack_texts(texts)
        emb = np.empty((len(texts), self.dimensions()), np.float32)
        _lib.check(self._lib.cs_embedders_embed_texts(self._h, self.tokenizer.handle, blob,
                                                      offsets.ctypes.data_as(_lib.u64p), len(texts), batch_size,
                                                      emb.ctypes.data_as(f32p), C.cast(C.byref(_SHUTDOWN), i32p)))
        return [emb[i] for i in range(emb.shape[0])]

    def index_ids(self, store, ids, mask, batch_size: int = 0) -> np.ndarray:
        """Embed [n, L] token chunks and append them to the sharded `store` -> the assigned ids."""
        ids = np.ascontiguousarray(ids, np.int32)
        mask = np.ascontiguousarray(mask, np.int32)
        n, L = ids.shape
        out = np.zeros(n, np.uint32)
        _lib.check(self._lib.cs_embedders_index_ids(self._h, store.handle, ids.ctypes.data_as(i32p),
                                                    mask.ctypes.data_as(i32p), n, L, batch_size,
                                                    out.ctypes.data_as(_lib.u32p), C.cast(C.byref(_SHUTDOWN), i32p)))
        return out

    def index_texts(self, store, texts: Sequence[str], batch_size: int = 0) -> np.ndarray:
        texts = list(texts)
        out = np.zeros(len(texts), np.uint32)
        if not texts:
            return out
        blob, offsets = pack_texts(texts)
        _lib.check(self._lib.cs_embedders_index_texts(self._h, self.tokenizer.handle, store.handle, blob,
                                                      offsets.ctypes.data_as(_lib.u64p), len(texts), batch_size,
                                                      out.ctypes.data_as(_lib.u32p), C.cast(C.byref(_SHUTDOWN), i32p)))
        return out
