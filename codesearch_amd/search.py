"""Host-side mirror of the step right after the scan in `search::search`
(/root/reference/src/search/mod.rs:494-611): retrieval limit, the cross-variant merge
(dedup by id keeping the best score, top retrieval_limit, score-descending) and the
early-termination predicate.  Rank arithmetic on <= 9 x 200 items — host work by design."""
from __future__ import annotations

from typing import List, Sequence

from .vector_store import SearchResult

HIGH_CONFIDENCE_THRESHOLD = 0.15  # mod.rs:598: distance < 0.15 (cos > 0.7 under arroy's Cosine)
EARLY_TERMINATION_TOP_N = 5       # mod.rs:599


def retrieval_limit(max_results: int, vector_only: bool, is_identifier_query: bool) -> int:
    """mod.rs:494-502."""
    if vector_only:
        return max_results
    if is_identifier_query:
        return max(max_results * 3, 100)
    return max(max_results * 5, 200)


def merge_variant_results(per_variant: Sequence[Sequence[SearchResult]], limit: int) -> List[SearchResult]:
    """mod.rs:513-590: keep, per chunk id, the result with the highest score across query
    variants; take the `limit` best; sort by score descending."""
    best = {}
    for results in per_variant:
        for r in results:
            cur = best.get(r.id)
            if cur is None or r.score > cur.score:
                best[r.id] = r
    merged = sorted(best.values(), key=lambda r: -r.score)[:limit]
    return merged


def should_use_vector_only(results: Sequence[SearchResult], vector_only: bool) -> bool:
    """mod.rs:601-611: skip FTS when the top-5 all have distance < 0.15."""
    if vector_only:
        return False
    top = list(results[:EARLY_TERMINATION_TOP_N])
    return bool(top) and all(r.distance < HIGH_CONFIDENCE_THRESHOLD for r in top)


# ---- result fusion (/root/reference/src/rerank/mod.rs:14-241) -----------------------------------
import dataclasses
from typing import Dict, Optional, Tuple

import numpy as np

DEFAULT_RRF_K = 20.0      # rerank/mod.rs:15
EXACT_MATCH_RRF_K = 5.0   # rerank/mod.rs:18


@dataclasses.dataclass
class FusedResult:
    """rerank/mod.rs:21-36."""
    chunk_id: int
    rrf_score: float
    vector_score: Optional[float] = None
    fts_score: Optional[float] = None
    vector_rank: Optional[int] = None
    fts_rank: Optional[int] = None


def _rrf_term(k: float, rank0: int) -> np.float32:
    # `1.0 / (k + rank as f32 + 1.0)` in f32, rerank/mod.rs:59
    return np.float32(1.0) / (np.float32(k) + np.float32(rank0) + np.float32(1.0))


def _sorted_desc(results: List[FusedResult]) -> List[FusedResult]:
    # rerank/mod.rs:101-105: sort by rrf_score descending (stable; the reference's input order is a
    # HashMap's, i.e. unspecified between equal scores — here first-seen order)
    return sorted(results, key=lambda r: -r.rrf_score)


def rrf_fusion(vector_results: Sequence[SearchResult], fts_results: Sequence[Tuple[int, float]],
               k: float = DEFAULT_RRF_K) -> List[FusedResult]:
    """rerank/mod.rs:48-108.  fts_results: (chunk_id, bm25 score) in rank order (FtsResult)."""
    acc: Dict[int, FusedResult] = {}
    for rank, r in enumerate(vector_results):
        e = acc.setdefault(r.id, FusedResult(r.id, np.float32(0.0)))
        e.rrf_score = np.float32(e.rrf_score + _rrf_term(k, rank))
        e.vector_score, e.vector_rank = r.score, rank + 1
    for rank, (cid, score) in enumerate(fts_results):
        e = acc.setdefault(cid, FusedResult(cid, np.float32(0.0)))
        e.rrf_score = np.float32(e.rrf_score + _rrf_term(k, rank))
        e.fts_score, e.fts_rank = score, rank + 1
    out = _sorted_desc(list(acc.values()))
    for e in out:
        e.rrf_score = float(e.rrf_score)
    return out


def vector_only(vector_results: Sequence[SearchResult]) -> List[FusedResult]:
    """rerank/mod.rs:111-124: pass-through, rrf_score = the vector score."""
    return [FusedResult(r.id, r.score, r.score, None, rank + 1, None) for rank, r in enumerate(vector_results)]


def rrf_fusion_with_exact(vector_results: Sequence[SearchResult], fts_results: Sequence[Tuple[int, float]],
                          exact_results: Sequence[Tuple[int, float]], vector_k: float = DEFAULT_RRF_K,
                          fts_k: float = DEFAULT_RRF_K, exact_k: float = EXACT_MATCH_RRF_K) -> List[FusedResult]:
    """rerank/mod.rs:139-241: three-way fusion; exact identifier matches get the smaller k.  The
    fts_score of a result is the mean of its FTS and exact scores when both exist (:216-221) and its
    fts_rank falls back to the exact rank (:229)."""
    acc: Dict[int, list] = {}  # id -> [rrf, vscore, fscore, escore, vrank, frank, erank]

    def entry(cid):
        return acc.setdefault(cid, [np.float32(0.0), None, None, None, None, None, None])

    for rank, r in enumerate(vector_results):
        e = entry(r.id)
        e[0] = np.float32(e[0] + _rrf_term(vector_k, rank)); e[1] = r.score; e[4] = rank + 1
    for rank, (cid, score) in enumerate(fts_results):
        e = entry(cid)
        e[0] = np.float32(e[0] + _rrf_term(fts_k, rank)); e[2] = score; e[5] = rank + 1
    for rank, (cid, score) in enumerate(exact_results):
        e = entry(cid)
        e[0] = np.float32(e[0] + _rrf_term(exact_k, rank)); e[3] = score; e[6] = rank + 1
    out = []
    for cid, (rrf, vs, fs, es, vr, fr, er) in acc.items():
        if fs is not None and es is not None:
            comb = float((np.float32(fs) + np.float32(es)) / np.float32(2.0))
        else:
            comb = fs if fs is not None else es
        out.append(FusedResult(cid, float(rrf), vs, comb, vr, fr if fr is not None else er))
    return _sorted_desc(out)
