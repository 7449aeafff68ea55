"""Host-side mirror of the reference's VectorStore over the C ABI.

Same method names, argument meaning and error behaviour as
/root/reference/src/vectordb/store.rs:94-750, so tests read like the reference's own
(`store.rs:833-1028`).  The vectors live in HBM behind `cs_index_*`; chunk metadata
(`ChunkMetadata`, store.rs:19-85) stays in host memory keyed by the u32 id, where the
reference keeps it in a second LMDB database.  A store opened with a path persists itself in a
flat form (vectors.f32 + sidecars); LMDB itself, MDB_MAP_FULL resizing and
page statistics are storage-engine concerns and out of scope.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import hashlib
import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import CsError, f32p, u32p, u64p


# ---- carrier types ---------------------------------------------------------------------------

@dataclass
class Chunk:
    """src/chunker/mod.rs:22-62 (fields the hot path and its metadata need)."""

    content: str
    start_line: int
    end_line: int
    kind: str  # ChunkKind Debug name, e.g. "Function"
    path: str
    context: List[str] = field(default_factory=list)
    signature: Optional[str] = None
    docstring: Optional[str] = None
    is_complete: bool = True
    split_index: Optional[int] = None
    hash: str = ""
    context_prev: Optional[str] = None
    context_next: Optional[str] = None

    def __post_init__(self):
        if not self.hash:  # Chunk::compute_hash, mod.rs:93-97
            self.hash = hashlib.sha256(self.content.encode("utf-8")).hexdigest()


@dataclass
class EmbeddedChunk:
    """src/embed/batch.rs:47-57."""

    chunk: Chunk
    embedding: Sequence[float]


@dataclass
class ChunkMetadata:
    """store.rs:19-85."""

    content: str
    path: str
    start_line: int
    end_line: int
    kind: str
    signature: Optional[str]
    docstring: Optional[str]
    context: Optional[str]
    hash: str
    context_prev: Optional[str] = None
    context_next: Optional[str] = None
    searchable_text: str = ""

    @staticmethod
    def from_embedded_chunk(ec: EmbeddedChunk) -> "ChunkMetadata":
        ch = ec.chunk
        parts = [p for p in (ch.signature, ch.docstring) if p is not None]
        parts += [ch.kind, ch.content]
        return ChunkMetadata(
            content=ch.content, path=ch.path, start_line=ch.start_line, end_line=ch.end_line,
            kind=ch.kind, signature=ch.signature, docstring=ch.docstring,
            context=" > ".join(ch.context) if ch.context else None, hash=ch.hash,
            context_prev=ch.context_prev, context_next=ch.context_next,
            searchable_text="\n".join(parts),
        )


@dataclass
class SearchResult:
    """store.rs:753-772."""

    id: int
    content: str
    path: str
    start_line: int
    end_line: int
    kind: str
    signature: Optional[str]
    docstring: Optional[str]
    context: Optional[str]
    hash: str
    distance: float
    score: float
    context_prev: Optional[str] = None
    context_next: Optional[str] = None


@dataclass
class StoreStats:
    """store.rs:784-792."""

    total_chunks: int
    total_files: int
    indexed: bool
    dimensions: int
    max_chunk_id: int


def cos_to_distance(cos):
    """arroy 0.5.0 Cosine distance (1 - cos)/2 (third-party)."""
    return (np.float32(1.0) - np.asarray(cos, np.float32)) * np.float32(0.5)


def cos_to_score(cos):
    """store.rs:478: score = 1.0 - distance."""
    return np.float32(1.0) - cos_to_distance(cos)


# ---- the store ---------------------------------------------------------------------------------

class VectorStore:
    """`VectorStore::new(db_path, dimensions)` — store.rs:110-176.

    `db_path = None`: a purely HBM-resident store.  With a path the store is persistent the way the
    reference's is across process restarts (it reopens its LMDB environment, `next_id = last_key + 1`,
    `indexed` = a built index exists, store.rs:139-170), in a flat form the GPU can ingest at full
    PCIe rate, written at `build_index()` next to where the reference keeps its
    `data.mdb`:
        <db_path>/vectors.f32         rows [next_id - id_base, dim] f32 little-endian, row = id - id_base
        <db_path>/vectors.meta.json   {"format", "dimensions", "id_base", "next_id", "removed": [ids], "built"}
        <db_path>/chunks.jsonl        one ChunkMetadata per line (the `chunks` database, store.rs:98)
    Rows appended after the last `build_index()` are in HBM only (the reference commits each insert
    to LMDB but cannot search it before the next build either).
    `device`, `capacity` and `id_base` are the GPU-side additions (shard placement).
    """

    FORMAT = 1

    def __init__(self, db_path, dimensions: int, device: int = 0, capacity: int = 0, id_base: int = 0,
                 devices: Optional[Sequence[int]] = None, rows_per_stripe: int = 65536):
        """devices=[d0, d1, ...]: the store is row-sharded over those GPUs inside this one process
        (cs_shards_*: ids stay contiguous, rows dealt in stripes of `rows_per_stripe`; a search broadcasts
        the queries, scans every shard and merges on d0).  Otherwise one cs_index on `device`."""
        self._lib = _lib.load()
        self.db_path = None if db_path is None else str(db_path)
        self.dimensions = int(dimensions)
        self.id_base = int(id_base)
        self.readonly = False
        if self.db_path is not None:  # reopening: reserve the persisted row count at once instead of growing by doubling
            try:
                with open(os.path.join(self.db_path, "vectors.meta.json")) as f:
                    capacity = max(int(capacity), int(json.load(f)["next_id"]) - self.id_base)
            except (OSError, ValueError, KeyError):
                pass
        handle = C.c_void_p()
        if devices is not None:
            if id_base:
                raise CsError(_lib.CS_ERR_BAD_ARG, "a sharded store hands out ids from 0")
            devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            _lib.check(self._lib.cs_shards_create(self.dimensions, len(devices), devs, int(rows_per_stripe), capacity,
                                                  C.byref(handle)))
            self._pfx = "cs_shards_"
        else:
            _lib.check(self._lib.cs_index_create(self.dimensions, capacity, device, id_base, C.byref(handle)))
            self._pfx = "cs_index_"
        self._h = handle
        self._meta: Dict[int, ChunkMetadata] = {}
        self._removed: set = set()
        self._persisted_rows = 0
        self._persisted_meta_ids: set = set()   # ids whose line is already in chunks.jsonl
        self._chunks_rewrite = False            # a persisted line was deleted: rewrite the file instead of appending
        if self.db_path is not None:
            os.makedirs(self.db_path, exist_ok=True)  # store.rs:113
            self._load()

    @classmethod
    def open_readonly(cls, db_path, dimensions: int, **kw) -> "VectorStore":
        """store.rs:183-250: same data, mutators refuse."""
        st = cls(db_path, dimensions, **kw)
        st.readonly = True
        return st

    # -- persistence
    def _paths(self):
        j = os.path.join
        return j(self.db_path, "vectors.f32"), j(self.db_path, "vectors.meta.json"), j(self.db_path, "chunks.jsonl")

    def _load(self) -> None:
        vec, meta, chunks = self._paths()
        if not (os.path.exists(vec) and os.path.exists(meta)):
            return
        m = json.load(open(meta))
        if m.get("format") != self.FORMAT:
            raise CsError(_lib.CS_ERR_BAD_ARG, f"unknown vector file format {m.get('format')!r} in {meta}")
        if int(m["dimensions"]) != self.dimensions:  # the reference would fail at the first insert/search
            raise CsError(_lib.CS_ERR_DIM_MISMATCH,
                          f"Embedding dimension mismatch: expected {self.dimensions}, got {m['dimensions']}")
        if int(m["id_base"]) != self.id_base:
            raise CsError(_lib.CS_ERR_BAD_ARG, f"store was written with id_base {m['id_base']}, opened with {self.id_base}")
        n = int(m["next_id"]) - self.id_base
        rows = np.memmap(vec, dtype="<f4", mode="r", shape=(n, self.dimensions)) if n else np.zeros((0, self.dimensions), np.float32)
        step = max(1, (256 << 20) // (4 * self.dimensions))  # 256 MB host pieces
        for lo in range(0, n, step):
            piece = np.ascontiguousarray(rows[lo:lo + step], np.float32)
            _lib.check(self._fn("add")(self._h, piece.ctypes.data_as(f32p), piece.shape[0], self.dimensions, None))
        self._persisted_rows = n
        removed = [int(i) for i in m.get("removed", [])]
        if removed:
            ids = np.ascontiguousarray(removed, np.uint32)
            _lib.check(self._fn("remove")(self._h, ids.ctypes.data_as(u32p), ids.size, None))
            self._removed = set(removed)
        if os.path.exists(chunks):
            for line in open(chunks):
                try:
                    d = json.loads(line)
                except ValueError:
                    self._chunks_rewrite = True  # a torn last line of an interrupted append
                    continue
                cid = int(d.pop("id"))
                if cid >= int(m["next_id"]):  # appended after the last commit point
                    self._chunks_rewrite = True
                    continue
                self._meta[cid] = ChunkMetadata(**d)
            self._persisted_meta_ids = set(self._meta)
        for cid in self._removed:  # deletes committed after the last build (delete_chunks) win over older lines
            if self._meta.pop(cid, None) is not None:
                self._chunks_rewrite = True
        if m.get("built"):
            _lib.check(self._fn("build")(self._h))

    def _persist(self) -> None:
        vec, meta, chunks = self._paths()
        n = self.next_id() - self.id_base
        with open(vec, "r+b" if os.path.exists(vec) else "w+b") as f:
            f.truncate(self._persisted_rows * self.dimensions * 4)  # drop anything past the last consistent state
            f.seek(0, 2)
            step = max(1, (256 << 20) // (4 * self.dimensions))
            for lo in range(self._persisted_rows, n, step):
                f.write(self._read_rows_for_file(lo, min(step, n - lo)).astype("<f4").tobytes())
        self._write_chunks(chunks)
        self._write_meta(meta, built=True)  # the meta file is the commit point
        self._persisted_rows = n

    def _read_rows_for_file(self, lo: int, cnt: int) -> np.ndarray:
        """Rows of ids [id_base + lo, + cnt) for the flat vector file, which stays indexed by id: a deleted id's row may
        have been reclaimed by the build (cs_index_build) — its slot in the file is zeros, and the meta file's removed
        list keeps it dead on reopening."""
        gone = sorted(i - self.id_base - lo for i in self._removed if lo <= i - self.id_base < lo + cnt)
        if not gone:
            return self.read_rows(lo, cnt)
        out = np.zeros((cnt, self.dimensions), np.float32)
        start = 0
        for g in gone + [cnt]:
            if g > start:
                out[start:g] = self.read_rows(lo + start, g - start)
            start = g + 1
        return out

    def _write_chunks(self, chunks: str) -> None:
        """chunks.jsonl: appended to when only new chunks arrived (the reference builds per file: rewriting N lines per
        build would be O(N) per file), rewritten atomically when a persisted line was deleted."""
        def line(cid):
            d = dataclasses.asdict(self._meta[cid])
            d["id"] = cid
            return json.dumps(d) + "\n"

        if self._chunks_rewrite or not os.path.exists(chunks):
            with open(chunks + ".tmp", "w") as f:
                for cid in sorted(self._meta):
                    f.write(line(cid))
            os.replace(chunks + ".tmp", chunks)
        else:
            new = sorted(set(self._meta) - self._persisted_meta_ids)
            if new:
                with open(chunks, "a") as f:
                    for cid in new:
                        f.write(line(cid))
        self._persisted_meta_ids = set(self._meta)
        self._chunks_rewrite = False

    def _write_meta(self, meta: str, built: bool) -> None:
        with open(meta + ".tmp", "w") as f:
            json.dump({"format": self.FORMAT, "dimensions": self.dimensions, "id_base": self.id_base,
                       "next_id": self.id_base + self._persisted_rows if not built else self.next_id(),
                       "removed": sorted(self._removed), "built": built}, f)
        os.replace(meta + ".tmp", meta)

    def db_size(self) -> int:
        """store.rs:741-749: bytes on disk."""
        if self.db_path is None:
            return 0
        return sum(os.path.getsize(p) for p in self._paths() if os.path.exists(p))

    def _fn(self, name: str):
        """cs_index_<name> or, for a store sharded over several GPUs, cs_shards_<name> (same signature)."""
        return getattr(self._lib, self._pfx + name)

    @property
    def sharded(self) -> bool:
        return self._pfx == "cs_shards_"

    def shard_lens(self) -> List[int]:
        """Live rows per shard (sharded stores only)."""
        if not self.sharded:
            return [len(self)]
        return [int(self._lib.cs_shards_shard_len(self._h, i)) for i in range(int(self._lib.cs_shards_count(self._h)))]

    def root_device(self) -> int:
        """The device whose HBM holds queries and results of the device-pointer search API."""
        return int(self._lib.cs_shards_root_device(self._h)) if self.sharded else int(self._lib.cs_index_device(self._h))

    def shard_handle(self, shard: int):
        """Borrowed cs_index of one shard (diagnostics: profile, counters)."""
        return C.c_void_p(self._lib.cs_shards_shard_index(self._h, shard)) if self.sharded else self._h

    def search_device(self, d_queries: int, nq: int, limit: int, d_keys: int = 0, d_cos: int = 0, d_ids: int = 0,
                      d_counts: int = 0, stream: int = 0) -> None:
        """cs_index_search_device / cs_shards_search_device: raw HBM pointers on root_device(), asynchronous on
        `stream`; never waits for the device."""
        p = lambda a: C.c_void_p(a) if a else None
        _lib.check(self._fn("search_device")(self._h, p(d_queries), nq, self.dimensions, limit, p(d_keys), p(d_cos),
                                             p(d_ids), p(d_counts), p(stream)))

    def search_status(self, stream: int = 0) -> bool:
        """True when a device search of more than 16 queries issued on `stream` overflowed a candidate buffer since
        the last call (rerun it in slices of <= 16 queries or through search_raw)."""
        ov = C.c_uint32(0)
        _lib.check(self._fn("search_status")(self._h, C.c_void_p(stream) if stream else None, C.byref(ov)))
        return bool(ov.value)

    def insert_device(self, d_rows: int, n: int, src_device: Optional[int] = None, stream: int = 0) -> np.ndarray:
        """Append n rows already in HBM (cs_index_add_device / cs_shards_add_device) -> ids."""
        self._writable()
        ids = np.zeros(n, np.uint32)
        st = C.c_void_p(stream) if stream else None
        if self.sharded:
            src = self.root_device() if src_device is None else int(src_device)
            _lib.check(self._lib.cs_shards_add_device(self._h, C.c_void_p(d_rows), src, n, self.dimensions,
                                                      ids.ctypes.data_as(u32p), st))
        else:
            _lib.check(self._lib.cs_index_add_device(self._h, C.c_void_p(d_rows), n, self.dimensions,
                                                     ids.ctypes.data_as(u32p), st))
        return ids

    def _writable(self):
        if self.readonly:
            raise CsError(_lib.CS_ERR_BAD_ARG, "store opened read-only (open_readonly)")

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self._fn("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- writes (`&mut self` in the reference)
    def insert_chunks_with_ids(self, chunks: Sequence[EmbeddedChunk]) -> List[int]:
        """store.rs:618-686 -> assigned ids (contiguous from next_id)."""
        self._writable()
        if not chunks:
            return []
        for ch in chunks:  # store.rs:666-672: first bad row aborts the transaction
            if len(ch.embedding) != self.dimensions:
                raise CsError(_lib.CS_ERR_DIM_MISMATCH,
                              f"Embedding dimension mismatch: expected {self.dimensions}, got {len(ch.embedding)}")
        rows = np.ascontiguousarray([ch.embedding for ch in chunks], dtype=np.float32)
        ids = np.zeros(len(chunks), np.uint32)
        _lib.check(self._fn("add")(self._h, rows.ctypes.data_as(f32p), len(chunks), self.dimensions,
                                          ids.ctypes.data_as(u32p)))
        for i, ch in zip(ids.tolist(), chunks):
            self._meta[i] = ChunkMetadata.from_embedded_chunk(ch)
        return ids.tolist()

    def insert_chunks(self, chunks: Sequence[EmbeddedChunk]) -> int:
        """store.rs:334-379 -> number inserted."""
        return len(self.insert_chunks_with_ids(chunks))

    def insert_embeddings(self, rows: np.ndarray) -> np.ndarray:
        """Vector-only append (no metadata) for bulk/bench use -> ids."""
        self._writable()
        rows = np.ascontiguousarray(rows, np.float32)
        if rows.ndim != 2:
            raise ValueError("rows must be [n, dim]")
        ids = np.zeros(rows.shape[0], np.uint32)
        _lib.check(self._fn("add")(self._h, rows.ctypes.data_as(f32p), rows.shape[0], rows.shape[1],
                                          ids.ctypes.data_as(u32p)))
        return ids

    def reserve_rows(self, n: int) -> int:
        """Room for n more rows -> the device address they are to be written at (cs_index_reserve_rows: E8 in place — an
        embedder's *_to_device call with this address stores its pooled rows straight into the corpus).  commit_rows(n)
        then makes them rows with the next n ids.  One cs_index only (a sharded store plans placement: cs_shards_plan_append)."""
        self._writable()
        if self.sharded:
            raise CsError(_lib.CS_ERR_UNSUPPORTED, "reserve_rows: a sharded store appends through cs_shards_add_device")
        p = C.c_void_p()
        _lib.check(self._lib.cs_index_reserve_rows(self._h, int(n), self.dimensions, C.byref(p)))
        return int(p.value or 0)

    def commit_rows(self, n: int) -> np.ndarray:
        ids = np.zeros(int(n), np.uint32)
        _lib.check(self._lib.cs_index_commit_rows(self._h, int(n), ids.ctypes.data_as(u32p)))
        return ids

    def insert_synthetic(self, n: int, seed: int, first_row: int = 0) -> int:
        """Generate n rows in HBM with include/cs_synth.h -> first id."""
        first = C.c_uint32()
        _lib.check(self._fn("add_synthetic")(self._h, n, seed, first_row, C.byref(first)))
        return int(first.value)

    def delete_chunks(self, chunk_ids: Sequence[int]) -> int:
        """store.rs:548-610 -> number deleted."""
        self._writable()
        ids = np.ascontiguousarray(chunk_ids, np.uint32)
        removed = C.c_uint64()
        _lib.check(self._fn("remove")(self._h, ids.ctypes.data_as(u32p), ids.size, C.byref(removed)))
        nxt = self.next_id()
        for i in ids.tolist():
            if self._meta.pop(i, None) is not None and i in self._persisted_meta_ids:  # store.rs:598
                self._chunks_rewrite = True
            if self.id_base <= i < nxt:
                self._removed.add(i)
        # The reference commits a delete to LMDB at once (store.rs:584-610).  Here the removed list of the persisted
        # state is committed at once too (atomic rewrite of the small meta file): a process that dies before the next
        # build_index() reopens without the deleted chunks.  Rows appended since the last build stay HBM-only until
        # that build, as arroy's tree does.
        if self.db_path is not None and os.path.exists(self._paths()[1]):
            m = json.load(open(self._paths()[1]))
            keep = int(m["next_id"])
            with open(self._paths()[1] + ".tmp", "w") as f:
                m["removed"] = sorted(i for i in self._removed if i < keep)
                json.dump(m, f)
            os.replace(self._paths()[1] + ".tmp", self._paths()[1])
        return int(removed.value)

    def build_index(self) -> None:
        """store.rs:386-430 (+ the flat vector file when the store has a path)."""
        self._writable()
        _lib.check(self._fn("build")(self._h))
        if self.db_path is not None:
            self._persist()

    def clear(self) -> None:
        """store.rs:690-707."""
        self._writable()
        _lib.check(self._fn("clear")(self._h))
        self._meta.clear()
        self._removed.clear()
        self._persisted_rows = 0
        if self.db_path is not None:
            for p in self._paths():
                if os.path.exists(p):
                    os.remove(p)

    # -- reads (`&self`)
    def is_indexed(self) -> bool:
        return bool(self._fn("is_built")(self._h))

    def next_id(self) -> int:
        return int(self._fn("next_id")(self._h))

    def __len__(self) -> int:
        return int(self._fn("len")(self._h))

    def stored_rows(self) -> int:
        """Rows the corpus matrix physically holds, tombstoned ones included: len(self) right after a build that reclaimed
        the deleted rows (cs_index_build, from 10 % dead rows on), more in between."""
        return int(self._fn("stored_rows")(self._h))

    def search_raw(self, queries, limit: int):
        """-> (cos [nq, limit] f32, ids [nq, limit] u32, counts [nq] u32); rows best-first."""
        q = np.ascontiguousarray(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, dim = q.shape
        cos = np.zeros((nq, max(limit, 1)), np.float32)
        ids = np.zeros((nq, max(limit, 1)), np.uint32)
        counts = np.zeros(nq, np.uint32)
        _lib.check(self._fn("search")(self._h, q.ctypes.data_as(f32p), nq, dim, limit,
                                             cos.ctypes.data_as(f32p), ids.ctypes.data_as(u32p),
                                             counts.ctypes.data_as(u32p)))
        return cos, ids, counts

    def search(self, query_embedding, limit: int) -> List[SearchResult]:
        """store.rs:431-486.  Results whose metadata is missing are skipped (store.rs:465)."""
        cos, ids, counts = self.search_raw(query_embedding, limit)
        return self._results(cos[0], ids[0], int(counts[0]))

    def search_batch(self, query_embeddings, limit: int) -> List[List[SearchResult]]:
        """One call for all query variants (the par_iter of src/search/mod.rs:508-511)."""
        cos, ids, counts = self.search_raw(query_embeddings, limit)
        return [self._results(cos[i], ids[i], int(counts[i])) for i in range(len(counts))]

    def search_variants(self, query_embeddings, limit: int):
        """search::search's vector leg (src/search/mod.rs:508-611) in one call: every variant searched for `limit`
        rows, union with a chunk keeping its best score, best `limit` distinct chunks best-first, merged on
        the device.  -> (results, high_confidence) where high_confidence is the early-termination predicate
        (top five all distance < 0.15)."""
        q = np.ascontiguousarray(query_embeddings, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, dim = q.shape
        cos = np.zeros(max(limit, 1), np.float32)
        ids = np.zeros(max(limit, 1), np.uint32)
        count, flag = C.c_uint32(), C.c_int32()
        _lib.check(self._fn("search_variants")(self._h, q.ctypes.data_as(f32p), nq, dim, limit,
                                                      cos.ctypes.data_as(f32p), ids.ctypes.data_as(u32p),
                                                      C.byref(count), C.byref(flag)))
        return self._results(cos, ids, int(count.value)), bool(flag.value)

    def _results(self, cos, ids, count) -> List[SearchResult]:
        out = []
        dist = cos_to_distance(cos[:count])
        for i in range(count):
            m = self._meta.get(int(ids[i]))
            if m is None:
                continue
            d = float(dist[i])
            out.append(SearchResult(
                id=int(ids[i]), content=m.content, path=m.path, start_line=m.start_line,
                end_line=m.end_line, kind=m.kind, signature=m.signature, docstring=m.docstring,
                context=m.context, hash=m.hash, distance=d, score=float(np.float32(1.0) - np.float32(d)),
                context_prev=m.context_prev, context_next=m.context_next))
        return out

    def get_chunk(self, chunk_id: int) -> Optional[ChunkMetadata]:
        """store.rs:709-713."""
        return self._meta.get(int(chunk_id))

    def get_chunk_as_result(self, chunk_id: int) -> Optional[SearchResult]:
        """store.rs:715-738 (distance/score 0.0, set by caller)."""
        m = self._meta.get(int(chunk_id))
        if m is None:
            return None
        return SearchResult(id=int(chunk_id), content=m.content, path=m.path, start_line=m.start_line,
                            end_line=m.end_line, kind=m.kind, signature=m.signature, docstring=m.docstring,
                            context=m.context, hash=m.hash, distance=0.0, score=0.0,
                            context_prev=m.context_prev, context_next=m.context_next)

    def get_chunks_by_file(self) -> Dict[str, List[int]]:
        """store.rs:529-543: path -> chunk ids."""
        out: Dict[str, List[int]] = {}
        for i, m in self._meta.items():
            out.setdefault(m.path, []).append(i)
        return out

    def stats(self) -> StoreStats:
        """store.rs:488-523."""
        files = {m.path for m in self._meta.values()}
        return StoreStats(total_chunks=len(self._meta), total_files=len(files), indexed=self.is_indexed(),
                          dimensions=self.dimensions, max_chunk_id=max(self._meta.keys(), default=0))

    def read_rows(self, first_row: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dimensions), np.float32)
        _lib.check(self._fn("read_rows")(self._h, first_row, n, out.ctypes.data_as(f32p)))
        return out

    # -- kernel timing (bench.py)
    def profile(self, enable: bool) -> None:
        _lib.check(self._lib.cs_index_profile(self._h, 1 if enable else 0))

    def profile_read(self, reset: bool = True):
        """-> (scan_ms_total, scan_launches, merge_ms_total)."""
        s, m, n = C.c_double(), C.c_double(), C.c_uint64()
        _lib.check(self._lib.cs_index_profile_read(self._h, C.byref(s), C.byref(n), C.byref(m), 1 if reset else 0))
        return s.value, int(n.value), m.value

    def set_filter_min_queries(self, n: int) -> None:
        """Searches of >= n queries use the f16 filter + exact refine path (default 2; 1 = always)."""
        _lib.check(self._lib.cs_index_set_filter_min_queries(self._h, int(n)))

    ROUTE_COST, ROUTE_STREAM, ROUTE_FILTER = 0, 1, 2

    def set_single_query_route(self, route: int) -> None:
        """How one query over a large index is answered (cs_index_set_single_query_route): ROUTE_COST (default: the int8
        filter + exact refine from 32,768 rows on for lists of k < 48 and from 300,000 rows on from k = 48 — the measured
        crossovers, index.hip single_int8_min_rows / single_int8_min_rows_long; CS_FILTER_SINGLE_MIN_ROWS /
        CS_FILTER_SINGLE_MIN_ROWS_LONG override them), ROUTE_STREAM (always the f32 streaming scan), ROUTE_FILTER.  Same bits."""
        if self.sharded:
            for g in range(int(self._lib.cs_shards_count(self._h))):
                _lib.check(self._lib.cs_index_set_single_query_route(self.shard_handle(g), int(route)))
        else:
            _lib.check(self._lib.cs_index_set_single_query_route(self._h, int(route)))

    def debug_counters(self):
        """-> (batched_searches, batched_fallbacks)"""
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.cs_index_debug_counters(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def filter_state(self):
        """-> (copy, spread, int8_reruns): which copy feeds the batched filter (0 none, 1 f16, 2 int8), the spread statistic
        of the last build and the searches the f16 copy answered after the int8 copy overflowed (cs_index_filter_state)."""
        if self.sharded:
            raise ValueError("filter_state is per index: use shard_handle()")
        c, sp, n = C.c_int32(), C.c_float(), C.c_uint64()
        _lib.check(self._lib.cs_index_filter_state(self._h, C.byref(c), C.byref(sp), C.byref(n)))
        return int(c.value), float(sp.value), int(n.value)

    def filter_copies(self):
        """-> (has_int8, has_f16, bytes): which filter copies of the corpus exist in HBM right now and what they occupy
        (cs_index_filter_copies).  An index whose int8 copy serves holds no f16 copy: it is built by the first search
        that needs it (the int8 copy retired) or at a build whose int8 copy does not serve."""
        if self.sharded:
            raise ValueError("filter_copies is per index: use shard_handle()")
        a, b, n = C.c_int32(), C.c_int32(), C.c_uint64()
        _lib.check(self._lib.cs_index_filter_copies(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return bool(a.value), bool(b.value), int(n.value)

    @property
    def handle(self):
        return self._h


# ---- embedding-cache value format --------------------------------------------------

def encode_cached_embedding(vec) -> bytes:
    """The value bytes of the reference's persistent embedding cache
    (`Database<Str, SerdeBincode<Vec<f32>>>`, /root/reference/src/embed/cache.rs:283-285):
    bincode 1.x of a Vec<f32> = u64 little-endian length, then the f32 little-endian elements."""
    v = np.ascontiguousarray(vec, "<f4")
    return int(v.size).to_bytes(8, "little") + v.tobytes()


def decode_cached_embedding(buf: bytes) -> np.ndarray:
    """Inverse of encode_cached_embedding; raises on a length/size mismatch."""
    if len(buf) < 8:
        raise ValueError("bincode Vec<f32>: missing length prefix")
    n = int.from_bytes(buf[:8], "little")
    if len(buf) != 8 + 4 * n:
        raise ValueError(f"bincode Vec<f32>: length {n} does not match {len(buf) - 8} payload bytes")
    return np.frombuffer(buf, "<f4", count=n, offset=8).astype(np.float32)
