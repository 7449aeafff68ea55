/*
 * cs_synth.h — counter-based synthetic data generator shared by the library
 * (cs_index_add_synthetic, synthetic-weight embedders), the oracle and the numpy mirror
 * (codesearch_amd/synth.py).  Any element is computable from (seed, flat index) alone,
 * so an 80M-row corpus can be produced in place on 8 GPUs and any slice re-derived on
 * the host for checking.
 *
 * Integer arithmetic only, then one exact int->float conversion and one exact
 * power-of-two scale: the value is bit-identical in C, HIP and numpy, with no libm and
 * no dependence on FMA contraction.
 *
 *   h = mix64(seed + idx * GOLDEN)            (splitmix64 finaliser)
 *   v = sum of the four 16-bit fields of h - 131070      in [-131070, 131070]
 *   value = (float)v * 2^-16                  Irwin-Hall(4): mean 0, sd ~0.577, |x| < 2
 *
 * Rows are deliberately NOT unit-norm (norm ~ 0.577*sqrt(dim)): the scan kernel has to
 * compute row norms exactly as the reference's cosine_similarity does
 * (examples/benchmark_models.rs:323-328).
 */
#ifndef CS_SYNTH_H
#define CS_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CS_SYNTH_FN __host__ __device__ static inline
#else
#define CS_SYNTH_FN static inline
#endif

#define CS_SYNTH_GOLDEN 0x9E3779B97F4A7C15ull

CS_SYNTH_FN uint64_t cs_synth_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

CS_SYNTH_FN int32_t cs_synth_int(uint64_t seed, uint64_t idx) {
    uint64_t h = cs_synth_mix64(seed + idx * CS_SYNTH_GOLDEN);
    int32_t v = (int32_t)(h & 0xFFFFu) + (int32_t)((h >> 16) & 0xFFFFu) +
                (int32_t)((h >> 32) & 0xFFFFu) + (int32_t)(h >> 48);
    return v - 131070;
}

/* Corpus / query element. */
CS_SYNTH_FN float cs_synth_value(uint64_t seed, uint64_t idx) {
    return (float)cs_synth_int(seed, idx) * (1.0f / 65536.0f);
}

/* Model-weight element: same stream scaled by 2^-shift (shift chosen per tensor so
 * activations stay O(1) through the encoder; see codesearch_amd/csrc/bert_params.h). */
CS_SYNTH_FN float cs_synth_weight(uint64_t seed, uint64_t idx, int shift) {
    return (float)cs_synth_int(seed, idx) * (1.0f / 65536.0f) * (1.0f / (float)(1u << shift));
}

/* Uniform integer in [0, n) (token ids, sequence lengths). */
CS_SYNTH_FN uint32_t cs_synth_below(uint64_t seed, uint64_t idx, uint32_t n) {
    uint64_t h = cs_synth_mix64(seed + idx * CS_SYNTH_GOLDEN);
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}

/* "Planted" query: row `row` of corpus(seed_c) plus half-amplitude noise from seed_q.
 * Both terms are multiples of 2^-17 below 4 in magnitude, so the sum is exact in f32;
 * its cosine with the planted row is ~0.894 while unrelated rows stay near
 * N(0, 1/dim), which makes top-1 known by construction at any corpus size. */
CS_SYNTH_FN float cs_synth_planted(uint64_t seed_c, uint64_t seed_q, uint64_t row,
                                   uint32_t dim, uint64_t qi, uint32_t col) {
    return cs_synth_value(seed_c, row * dim + col) +
           0.5f * cs_synth_value(seed_q, qi * dim + col);
}

#endif /* CS_SYNTH_H */
