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
