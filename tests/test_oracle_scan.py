"""Pins the scan/top-k oracle: reference known answers, committed golden vectors, an
independent float64 check, and the cross-language identity of the synthetic generator.
CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from codesearch_amd.synth import synth_planted, synth_rows

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scan_golden.json")))


def test_reference_kat_store_insert_and_search(oracle):
    """/root/reference/src/vectordb/store.rs:846-893: rows e0,e1; query (0.9,0.1,0,0);
    k=2 -> first hit is row 0 and score[0] > score[1]."""
    kat = GOLDEN["reference_kat"]["store_rs_846_893"]
    rows = np.array(kat["rows"], np.float32)
    q = np.array(kat["query"], np.float32)
    cos, ids = oracle.scan_topk(rows, q, kat["k"])
    assert ids.tolist() == kat["expect_ids"] == [0, 1]
    assert cos[0] > cos[1]
    np.testing.assert_allclose(cos, kat["expect_cos"], atol=1e-6)
    np.testing.assert_allclose(cos, [0.993884, 0.110432], atol=1e-6)
    # score = 1 - distance, distance = (1 - cos)/2  (store.rs:477-478 + arroy Cosine)
    score = 1.0 - (1.0 - cos) / 2.0
    assert score[0] > score[1] and 0.0 <= score[1] <= 1.0


def test_reference_kat_cosine_3d(oracle):
    """/root/reference/src/embed/batch.rs:326-340."""
    for case in GOLDEN["reference_kat"]["batch_rs_326_340"]:
        c = oracle.cosine(case["a"], case["b"])
        if "expect" in case:
            assert abs(c - case["expect"]) < case["tol"]
        else:
            assert case["lo"] < c < case["hi"]
    assert oracle.cosine([0, 0, 0], [1, 0, 0]) == 0.0  # zero guard, batch.rs:320-322


def test_generator_identity_numpy_c_and_digest(oracle):
    for d in GOLDEN["digests"]:
        if d["n"] * d["dim"] > 2_000_000:
            a = oracle.synth_rows(d["seed"], d["first_row"], d["n"], d["dim"])
        else:
            a = oracle.synth_rows(d["seed"], d["first_row"], d["n"], d["dim"])
            b = synth_rows(d["seed"], d["first_row"], d["n"], d["dim"])
            assert np.array_equal(a, b)
        assert hashlib.sha256(a.tobytes()).hexdigest() == d["sha256"]
    p_c = oracle.synth_planted(5, 6, [3, 9], 384)
    p_n = synth_planted(5, 6, [3, 9], 384)
    assert np.array_equal(p_c, p_n)
    assert abs(float(a.std()) - 0.577) < 0.01


@pytest.mark.parametrize("mode", ["literal", "omp", "f64"])
def test_golden_cases(oracle, mode):
    cache = {}
    for case in GOLDEN["cases"]:
        n, dim, seed = case["n"], case["dim"], case["seed"]
        if n > 3000 and mode == "literal" and case["k"] not in (10,):
            continue  # keep the scalar literal scan to a few seconds
        key = (n, dim, seed)
        if key not in cache:
            cache[key] = oracle.synth_rows(seed, 0, n, dim)
        corpus = cache[key]
        if case["kind"] == "random":
            q = synth_rows(case["query_seed"], case["qi"], 1, dim)[0]
        else:
            q = synth_planted(seed, case["query_seed"], [case["planted_row"]] * (case["qi"] + 1), dim)[case["qi"]]
        cos, ids = oracle.scan_topk(corpus, q, case["k"], mode=mode)
        assert len(ids) == case["k"]
        assert ids.tolist() == case["ids"], (case["n"], case["k"], case["gap_k_k1"])
        np.testing.assert_allclose(cos, case["cos"], atol=2e-6 if mode != "f64" else 1e-12)
        if case["kind"] == "planted":
            assert ids[0] == case["planted_row"] and cos[0] > 0.85


def test_ties_zero_rows_k_gt_n_tombstones(oracle):
    dim = 8
    base = synth_rows(11, 0, 6, dim)
    corpus = np.stack([base[0], base[1], base[0], np.zeros(dim, np.float32), base[0] * 2.0, base[2]])
    q = base[0]
    cos, ids = oracle.scan_topk(corpus, q, 10)
    # k > N: every live row returned once; ties (rows 0,2,4 all cos==1) in id order
    assert len(ids) == 6
    assert ids[:3].tolist() == [0, 2, 4]
    assert cos[0] == cos[1] == cos[2]
    zero_pos = ids.tolist().index(3)
    assert cos[zero_pos] == 0.0
    # tombstone rows 0 and 4
    dead = np.zeros(1, np.uint32)
    dead[0] = (1 << 0) | (1 << 4)
    cos2, ids2 = oracle.scan_topk(corpus, q, 3, dead=dead)
    assert ids2[0] == 2 and 0 not in ids2 and 4 not in ids2
    for mode in ("omp", "f64"):
        c3, i3 = oracle.scan_topk(corpus, q, 3, dead=dead, mode=mode)
        assert i3.tolist() == ids2.tolist()
    # id_base shifts ids only
    c4, i4 = oracle.scan_topk(corpus, q, 3, id_base=1000)
    assert i4.tolist() == [1000, 1002, 1004]
    # k = 0 and empty corpus
    assert len(oracle.scan_topk(corpus, q, 0)[1]) == 0
    assert len(oracle.scan_topk(np.zeros((0, dim), np.float32), q, 5)[1]) == 0


def test_nan_rows_never_selected(oracle):
    dim = 4
    corpus = np.array([[np.nan, 0, 0, 0], [1, 0, 0, 0], [np.inf, 0, 0, 0]], np.float32)
    cos, ids = oracle.scan_topk(corpus, np.array([1, 0, 0, 0], np.float32), 3)
    assert ids.tolist() == [1]


def test_merge_matches_whole_scan(oracle):
    corpus = oracle.synth_rows(21, 0, 5000, 384)
    q = synth_rows(22, 0, 1, 384)[0]
    k = 10
    whole_c, whole_i = oracle.scan_topk(corpus, q, k, mode="omp")
    parts_c = np.zeros((4, k), np.float32)
    parts_i = np.zeros((4, k), np.uint32)
    counts = np.zeros(4, np.uint32)
    for s in range(4):
        lo, hi = s * 1250, (s + 1) * 1250
        c, i = oracle.scan_topk(corpus[lo:hi], q, k, id_base=lo, mode="omp")
        parts_c[s, : len(c)], parts_i[s, : len(i)], counts[s] = c, i, len(i)
    mc, mi = oracle.merge_topk(parts_c, parts_i, counts, k)
    assert mi.tolist() == whole_i.tolist()
    assert np.array_equal(mc, whole_c)
