"""The reference's own fusion tests (/root/reference/src/rerank/mod.rs:273-337) re-expressed against
the host mirror, plus the three-way variant's field rules (:139-241)."""
import numpy as np

from codesearch_amd.search import (DEFAULT_RRF_K, EXACT_MATCH_RRF_K, rrf_fusion, rrf_fusion_with_exact, vector_only)
from codesearch_amd.vector_store import SearchResult


def vres(cid, score):
    return SearchResult(id=cid, score=score, distance=0.0, path=f"file_{cid}.rs", content=f"content {cid}",
                        start_line=1, end_line=10, kind="function", signature=None, docstring=None, context=None,
                        hash="")


def test_rrf_fusion_basic():  # rerank/mod.rs:273-306
    fused = rrf_fusion([vres(1, 0.9), vres(2, 0.8), vres(3, 0.7)], [(2, 10.0), (1, 8.0), (4, 6.0)], 20.0)
    by = {r.chunk_id: r for r in fused}
    assert by[1].vector_rank and by[1].fts_rank and by[2].vector_rank and by[2].fts_rank
    assert by[4].vector_rank is None and by[4].fts_rank is not None
    assert [r.chunk_id for r in fused][:2] in ([1, 2], [2, 1])  # equal sums 1/21 + 1/22
    assert fused[-1].chunk_id in (3, 4)
    assert all(fused[i].rrf_score >= fused[i + 1].rrf_score for i in range(len(fused) - 1))


def test_rrf_score_calculation():  # rerank/mod.rs:308-324
    fused = rrf_fusion([vres(1, 0.9)], [(1, 10.0)], 20.0)
    assert len(fused) == 1
    assert abs(fused[0].rrf_score - (1.0 / 21.0 + 1.0 / 21.0)) < 1e-4
    f32 = np.float32(1.0) / np.float32(21.0)
    assert fused[0].rrf_score == float(np.float32(f32 + f32))  # f32 arithmetic, as in Rust


def test_vector_only():  # rerank/mod.rs:326-336
    res = vector_only([vres(1, 0.9), vres(2, 0.8)])
    assert len(res) == 2 and res[0].chunk_id == 1 and res[0].rrf_score == 0.9 and res[0].fts_score is None


def test_three_way_exact_boost():
    assert DEFAULT_RRF_K == 20.0 and EXACT_MATCH_RRF_K == 5.0  # rerank/mod.rs:15-18
    fused = rrf_fusion_with_exact([vres(1, 0.9), vres(2, 0.8)], [(2, 4.0), (3, 2.0)], [(3, 9.0)])
    by = {r.chunk_id: r for r in fused}
    # exact match at rank 1 with k = 5 outweighs a rank-1 vector hit: 1/6 + 1/22 > 1/21
    assert fused[0].chunk_id == 3
    assert abs(by[3].rrf_score - (1 / 6 + 1 / 22)) < 1e-6
    assert by[3].fts_score == (2.0 + 9.0) / 2.0 and by[3].fts_rank == 2       # :216-221, :229
    assert by[2].fts_score == 4.0 and by[1].fts_score is None and by[1].fts_rank is None
    only_exact = rrf_fusion_with_exact([], [], [(7, 1.5)])
    assert only_exact[0].fts_score == 1.5 and only_exact[0].fts_rank == 1
