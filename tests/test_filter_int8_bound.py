"""The error bound of the int8 filter copy (codesearch_amd/csrc/scan_filter.hip, "int8 filter copy" and q8_threshold),
restated in numpy and checked numerically on the CPU: for every (row, query) pair the cosine must lie within the band
the kernels subtract from tau — otherwise a row that beats tau could be dropped before the exact refine sees it.  The
quantiser below follows corpus_q8_kernel / prep_queries_kernel step by step in float32; the cosine is float64."""
import numpy as np
import pytest

F = np.float32


def _quantise_tile(rows, mu):
    """rows: [128, dim] f32 -> (a int8-valued f32, inv_t, hB, E, N) as corpus_q8_kernel."""
    nrm = np.sqrt((rows.astype(F) ** 2).sum(axis=1, dtype=F)).astype(F)
    u = np.where(nrm[:, None] == 0, F(0), rows / np.where(nrm == 0, F(1), nrm)[:, None]).astype(F) - mu
    mx = np.abs(u).max()
    inv = F(127.0) / mx if mx > 0 else F(1.0)
    t = (u * inv).astype(F)
    a = np.clip(np.rint(t), -127, 127).astype(F)
    hB = F(0.5001) * np.abs(a).sum(axis=1).max()
    E = F(np.sqrt(((t - a) ** 2).sum(axis=1, dtype=F).max())) * F(1.001) + F(1e-3)
    N = F(np.sqrt((a * a).sum(axis=1, dtype=F).max())) * F(1.0001)
    return a, inv, hB, E, N


def _quantise_query(q, mu):
    m = F(np.sqrt((q.astype(F) ** 2).sum(dtype=F)))
    v = (q / m).astype(F) if m > 0 else np.zeros_like(q, F)
    mx = np.abs(v).max()
    inv = F(127.0) / mx if (mx > 0 and np.isfinite(mx)) else F(1.0)
    t = (v * inv).astype(F)
    b = np.clip(np.rint(t), -127, 127).astype(F)
    return (b, inv, F(0.5001) * np.abs(b).sum(), F(np.sqrt((b * b).sum(dtype=F))) * F(1.0001),
            F(np.sqrt(((t - b) ** 2).sum(dtype=F))) * F(1.001) + F(1e-3), F((v * mu).sum(dtype=F)))


def _quantise_query_two_planes(q, mu):
    """prep_queries_kernel's second quantisation: B = rint(v * 128 inv) = 128 hi + lo; the MFMAs compute a . hi and a . lo."""
    m = F(np.sqrt((q.astype(F) ** 2).sum(dtype=F)))
    v = (q / m).astype(F) if m > 0 else np.zeros_like(q, F)
    mx = np.abs(v).max()
    inv = (F(127.0) / mx if (mx > 0 and np.isfinite(mx)) else F(1.0)) * F(128.0)
    t = (v * inv).astype(F)
    B = np.clip(np.rint(t), -16256, 16256).astype(F)
    hi = np.rint(B * F(0.0078125)).astype(F)
    lo = (B - F(128.0) * hi).astype(F)
    assert np.abs(hi).max() <= 127 and np.abs(lo).max() <= 64
    return (hi, lo, inv, F(0.5001) * np.abs(B).sum(dtype=F) * F(1.0001), F(np.sqrt((B * B).sum(dtype=F))) * F(1.0001),
            F(np.sqrt(((t - B) ** 2).sum(dtype=F))) * F(1.001) + F(1e-3), F((v * mu).sum(dtype=F)))


def _check(rows, queries, dim):
    n = rows.shape[0] // 128 * 128
    rows = rows[:n]
    unit = rows.astype(np.float64)
    nr = np.linalg.norm(unit, axis=1)
    unit = np.where(nr[:, None] == 0, 0.0, unit / np.where(nr == 0, 1.0, nr)[:, None])
    mu = unit[: min(n, 1 << 20)].mean(axis=0).astype(F)          # unit_mean_kernel (any vector keeps the bound)
    slack = F(dim) * F(1.1920929e-07) + F(4.0e-5)
    worst = -np.inf
    for q in queries:
        b, inv_q, hA, nb, Dq, qmu = _quantise_query(q, mu)
        planes = _quantise_query_two_planes(q, mu)
        qn = np.linalg.norm(q.astype(np.float64))
        for t0 in range(0, n, 128):
            a, inv_t, hB, E, N = _quantise_tile(rows[t0:t0 + 128], mu)
            I = a.astype(np.float64) @ b.astype(np.float64)                      # the MFMA: exact integers
            cos = (unit[t0:t0 + 128] @ q.astype(np.float64)) / qn if qn > 0 else np.zeros(128)
            band_units = min(hA, nb * E) + min(hB, N * Dq) + min(F(0.2501) * F(dim), E * Dq) + F(4.0)
            # the kernel keeps a row iff I > (tau - slack - q.mu) inv_q inv_t - band_units; a row with cos >= tau must pass,
            # i.e. (cos - slack - q.mu) inv_q inv_t - band_units < I for every row
            lhs = (cos - float(slack) - float(qmu)) * float(inv_q) * float(inv_t) - float(band_units)
            worst = max(worst, float((lhs - I).max()))
            assert (lhs < I).all(), (t0, float((lhs - I).max()))
            # ... and with the query in two planes (the finer unit; q8_threshold's relative guard included)
            hi, lo, inv2, hA2, nb2, Dq2, _ = planes
            I2 = 128.0 * (a.astype(np.float64) @ hi.astype(np.float64)) + a.astype(np.float64) @ lo.astype(np.float64)
            assert np.abs(I2).max() < 2 ** 31
            lead = (cos - float(slack) - float(qmu)) * float(inv2) * float(inv_t)
            band2 = min(hA2, nb2 * E) + min(hB, N * Dq2) + min(F(0.2501) * F(dim), E * Dq2) + F(4.0)
            assert (lead - float(band2) - np.abs(lead) * 4.0e-7 < I2).all(), t0
    return worst


@pytest.mark.parametrize("dim", [384, 768])
def test_band_covers_every_pair(dim):
    rng = np.random.default_rng(7)
    n = 512
    rows = rng.normal(size=(n, dim)).astype(F)
    rows[0:64] *= F(1e6)
    rows[64:128] *= F(1e-9)
    rows[130] = 0
    rows[131, :] = 0
    rows[131, 5] = 3.0                       # one-hot row
    rows[140:150, ::17] *= F(25.0)           # outlier coordinates inside a tile
    rows[256:384] = rows[256] + rng.normal(0, 1e-3, (128, dim)).astype(F)   # a tile of near-duplicates
    rows[384:512] += F(4.0) * rng.normal(size=(1, dim)).astype(F)           # a tile with a big common component
    queries = [rng.normal(size=dim).astype(F) for _ in range(6)]
    queries.append(rows[256].copy())
    queries.append(rows[131].copy())         # one-hot query
    queries.append((rows[400] * F(1e-7)).astype(F))
    sparse = np.zeros(dim, F)
    sparse[::64] = 1.0
    queries.append(sparse)
    worst = _check(rows, np.stack(queries), dim)
    assert worst < 0.0


def test_band_is_tight_enough_to_be_useful():
    """For evenly spread rows the Cauchy-Schwarz side makes the band ~0.7 of the L1 side: about 0.017 in cosine units."""
    dim = 384
    rng = np.random.default_rng(3)
    rows = rng.normal(size=(128, dim)).astype(F)
    mu = np.zeros(dim, F)
    a, inv_t, hB, E, N = _quantise_tile(rows, mu)
    b, inv_q, hA, nb, Dq, _ = _quantise_query(rng.normal(size=dim).astype(F), mu)
    l1 = (hA + hB + F(0.2501) * dim) / (inv_q * inv_t)
    cs = (min(hA, nb * E) + min(hB, N * Dq) + min(F(0.2501) * dim, E * Dq)) / (inv_q * inv_t)
    assert 0.010 < cs < 0.022 and cs < 0.8 * l1, (float(cs), float(l1))
