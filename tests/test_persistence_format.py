"""CPU tests of the on-disk formats of SURVEY.md §8f-4: the embedding-cache value encoding
(bincode Vec<f32>, /root/reference/src/embed/cache.rs:283-285)."""
import numpy as np
import pytest

from codesearch_amd.vector_store import decode_cached_embedding, encode_cached_embedding


def test_bincode_vec_f32_round_trip_and_layout():
    v = np.array([1.0, -2.5, 3.25e-3], np.float32)
    b = encode_cached_embedding(v)
    # bincode 1.x, default options: u64 LE length then the elements
    assert b[:8] == (3).to_bytes(8, "little") and len(b) == 8 + 12
    assert b[8:12] == np.float32(1.0).tobytes()
    assert np.array_equal(decode_cached_embedding(b), v)
    e = np.random.default_rng(0).standard_normal(384).astype(np.float32)
    assert np.array_equal(decode_cached_embedding(encode_cached_embedding(e)), e)
    assert decode_cached_embedding(encode_cached_embedding([])).size == 0


def test_bincode_vec_f32_rejects_truncated_values():
    b = encode_cached_embedding(np.ones(4, np.float32))
    with pytest.raises(ValueError):
        decode_cached_embedding(b[:-1])
    with pytest.raises(ValueError):
        decode_cached_embedding(b[:5])
