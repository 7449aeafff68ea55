"""Host-side mirror of the callers between chunks and the encoder:
BatchEmbedder / prepare_text / clean_docstring (/root/reference/src/embed/batch.rs:60-231),
EmbeddingStats (batch.rs:12-44), and EmbeddingService with its content-hash and query caches
(src/embed/mod.rs:17-292, src/embed/cache.rs).  String work only; the arithmetic is the
FastEmbedder mirror's.  The reference's LMDB/moka stores are replaced by in-process dicts
(storage is out of scope); cache SEMANTICS are kept: lookup by chunk.hash, only misses reach
the encoder, results keep the caller's order (the reordering quirk of
CachedBatchEmbedder, SURVEY Appendix A, is deliberately not reproduced)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from .vector_store import Chunk, EmbeddedChunk


def _rust_lines(s: str) -> List[str]:
    """str::lines(): split on '\\n', drop one trailing '\\r' per line, no final empty line."""
    if not s:
        return []
    parts = s.split("\n")
    if parts and parts[-1] == "":
        parts.pop()
    return [p[:-1] if p.endswith("\r") else p for p in parts]


def _strip_prefix(s: str, p: str) -> Optional[str]:
    return s[len(p):] if s.startswith(p) else None


def clean_docstring(doc: str) -> str:
    """batch.rs:197-231."""
    out = []
    for line in _rust_lines(doc):
        trimmed = line.strip()
        if trimmed == "*/":
            cleaned = ""
        else:
            cleaned = trimmed
            for p in ("///", "//!", "//", "/**", "*", '"'):
                r = _strip_prefix(trimmed, p)
                if r is not None:
                    cleaned = r
                    break
            cleaned = cleaned.strip()
        cleaned = cleaned.strip()
        if cleaned:
            out.append(cleaned)
    result = " ".join(out)
    if result.endswith('"'):
        result = result[:-1]
    return result.strip()


def prepare_text(chunk: Chunk) -> str:
    """batch.rs:137-181: Context / Signature / Name / Documentation / Code."""
    parts = []
    if chunk.context:
        parts.append("Context: " + " > ".join(chunk.context))
    if chunk.signature is not None:
        sig = chunk.signature
        parts.append("Signature: " + sig)
        words = sig.split()
        if len(words) > 1:  # split_whitespace().nth(1)
            name = words[1].split("<")[0].split("(")[0].split("{")[0]
            parts.append("Name: " + name)
    if chunk.docstring is not None:
        cleaned = clean_docstring(chunk.docstring)
        if cleaned:
            parts.append("Documentation: " + cleaned)
    parts.append("Code:\n" + chunk.content)
    return "\n".join(parts)


@dataclass
class EmbeddingStats:
    """batch.rs:12-44."""

    total_chunks: int = 0
    embedded_chunks: int = 0
    cached_chunks: int = 0
    failed_chunks: int = 0
    total_time_ms: int = 0

    def cache_hit_rate(self) -> float:
        return 0.0 if self.total_chunks == 0 else self.cached_chunks / self.total_chunks

    def success_rate(self) -> float:
        return 0.0 if self.total_chunks == 0 else self.embedded_chunks / self.total_chunks

    def chunks_per_second(self) -> float:
        return 0.0 if self.total_time_ms == 0 else self.embedded_chunks / self.total_time_ms * 1000.0


class BatchEmbedder:
    """batch.rs:60-194.  `batch_size` defaults to the reference's 32; on the GPU a whole
    mini-batch (256) per call is the better setting (`with_batch_size`)."""

    def __init__(self, embedder, batch_size: int = 32):
        self.embedder = embedder
        self.batch_size = batch_size

    @classmethod
    def with_batch_size(cls, embedder, batch_size: int) -> "BatchEmbedder":
        return cls(embedder, batch_size)

    prepare_text = staticmethod(prepare_text)

    def embed_chunks(self, chunks: Sequence[Chunk]) -> List[EmbeddedChunk]:
        """batch.rs:84-115: the chunks go to the embedder in slices of `batch_size` (32).  On the GPU embedder every
        slice is SUBMITTED first and collected afterwards (cs_embedder_submit_texts / cs_embedder_wait): the first
        wait embeds all queued slices as full device batches, so the reference's call shape keeps the large-batch
        rate.  An embedder without a queue (any object with embed_batch) is called slice by slice as the reference
        does."""
        out: List[EmbeddedChunk] = []
        slices = [chunks[lo:lo + self.batch_size] for lo in range(0, len(chunks), self.batch_size)]
        if hasattr(self.embedder, "submit_texts") and len(slices) > 1:
            tickets = [self.embedder.submit_texts([prepare_text(c) for c in part]) for part in slices]
            try:
                for part in slices:
                    embs = self.embedder.wait(tickets[0])
                    tickets.pop(0)
                    out.extend(EmbeddedChunk(c, e) for c, e in zip(part, embs))
            finally:
                for t in tickets:  # a failed or interrupted wait: nothing of this call stays queued
                    try:
                        self.embedder.discard(t)
                    except Exception:
                        pass
            return out
        for part in slices:
            embs = self.embedder.embed_batch([prepare_text(c) for c in part])
            out.extend(EmbeddedChunk(c, e) for c, e in zip(part, embs))
        return out

    def embed_chunk(self, chunk: Chunk) -> EmbeddedChunk:
        return EmbeddedChunk(chunk, self.embedder.embed_one(prepare_text(chunk)))

    def dimensions(self) -> int:
        return self.embedder.dimensions()


class EmbeddingService:
    """src/embed/mod.rs:17-292: content-hash cache in front of the encoder + query cache."""

    def __init__(self, embedder, batch_size: int = 32, max_cache_entries: int = 200_000,
                 max_query_entries: int = 1000):
        self.batch_embedder = BatchEmbedder(embedder, batch_size)
        self._cache: Dict[str, np.ndarray] = {}
        self._query_cache: Dict[str, np.ndarray] = {}
        self.max_cache_entries = max_cache_entries
        self.max_query_entries = max_query_entries
        self.cache_hits = 0
        self.cache_misses = 0

    def embed_chunks(self, chunks: Sequence[Chunk]) -> List[EmbeddedChunk]:
        """mod.rs:86-161: lookup by chunk.hash, embed the misses in one go, keep order."""
        if not chunks:
            return []
        results: List[Optional[EmbeddedChunk]] = [None] * len(chunks)
        miss_idx = []
        for i, c in enumerate(chunks):
            hit = self._cache.get(c.hash)
            if hit is not None:
                results[i] = EmbeddedChunk(c, hit)
            else:
                miss_idx.append(i)
        self.cache_hits += len(chunks) - len(miss_idx)
        self.cache_misses += len(miss_idx)
        if miss_idx:
            embedded = self.batch_embedder.embed_chunks([chunks[i] for i in miss_idx])
            for i, ec in zip(miss_idx, embedded):
                results[i] = ec
                self._cache[ec.chunk.hash] = ec.embedding
            while len(self._cache) > self.max_cache_entries:  # evict_if_needed: oldest first
                self._cache.pop(next(iter(self._cache)))
        return results  # type: ignore[return-value]

    def embed_query(self, query: str) -> np.ndarray:
        """mod.rs:164-181."""
        hit = self._query_cache.get(query)
        if hit is not None:
            return hit
        e = self.batch_embedder.embedder.embed_one(query)
        self._put_query(query, e)
        return e

    def embed_queries_batch(self, queries: Sequence[str]) -> List[np.ndarray]:
        """mod.rs:184-226: one encoder call for all cache misses, original order kept."""
        if not queries:
            return []
        out: List[Optional[np.ndarray]] = [self._query_cache.get(q) for q in queries]
        miss = [i for i, e in enumerate(out) if e is None]
        if miss:
            embs = self.batch_embedder.embedder.embed_batch([queries[i] for i in miss])
            for i, e in zip(miss, embs):
                out[i] = e
                self._put_query(queries[i], e)
        return out  # type: ignore[return-value]

    def _put_query(self, q: str, e: np.ndarray) -> None:
        self._query_cache[q] = e
        while len(self._query_cache) > self.max_query_entries:
            self._query_cache.pop(next(iter(self._query_cache)))

    def dimensions(self) -> int:
        return self.batch_embedder.dimensions()
