"""Host restatement of the split-f16 operand format (codesearch_amd/csrc/split_f16.hpp) in
numpy: the reconstruction bound the kernels rely on, checked on CPU.  The GPU tests
(test_gpu_gemm_split.py) check the kernels themselves."""
import numpy as np

LO_SCALE = np.float32(2048.0)
MIN_NORMAL = np.float32(2.0 ** -14)


def split(x):
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)  # may be an f16 subnormal (the f16 MFMA consumes those exactly)
    lo = ((x - hi.astype(np.float32)) * LO_SCALE).astype(np.float16)
    return hi, lo


def test_reconstruction_error_bound():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000) * s for s in (1e-7, 1e-4, 1e-2, 1.0, 50.0, 6e3)]).astype(np.float32)
    hi, lo = split(x)
    rec = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
    err = np.abs(rec - x.astype(np.float64))
    ax = np.abs(x).astype(np.float64)
    # |e| <= 2^-22 |x| above the f16 normal threshold; below it hi and lo are subnormals on a 2^-24 /
    # 2^-35 grid: |e| <= 2^-36
    bound = np.where(ax >= 2.0 ** -14, ax * 2.0 ** -22, 2.0 ** -36) + 2.0 ** -40
    assert np.all(err <= bound)


def test_three_product_sum_is_f32_grade():
    rng = np.random.default_rng(1)
    K = 1536
    a = rng.standard_normal((64, K)).astype(np.float32)
    w = (rng.standard_normal((48, K)) * 0.05).astype(np.float32)
    ah, al = split(a)
    wh, wl = split(w)
    f = lambda v: v.astype(np.float64)
    approx = f(ah) @ f(wh).T + (f(ah) @ f(wl).T + f(al) @ f(wh).T) / 2048.0
    exact = f(a) @ f(w).T
    scale = np.abs(f(a)) @ np.abs(f(w)).T
    assert (np.abs(approx - exact) / scale).max() < 3 * 2.0 ** -22
