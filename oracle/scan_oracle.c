/*
 * scan_oracle.c — CPU restatement of the reference's brute-force cosine scan.
 * TEST INFRASTRUCTURE ONLY (see cs_oracle.h).  Compile with -ffp-contract=off so the
 * literal functions keep one rounding per operation, as rustc emits for
 * `a.iter().zip(b).map(|(x, y)| x * y).sum()` (no FMA contraction in Rust).
 *
 * Follows, in /root/reference:
 *   examples/benchmark_models.rs:323-328  cosine_similarity (dot, two magnitudes, divide)
 *   src/embed/batch.rs:316-324            the same with the zero-magnitude guard
 *   examples/benchmark_models.rs:155-165  linear scan keeping a strict-`>` best
 *   src/vectordb/store.rs:464-483         results best-first, id + distance/score
 */
#include "cs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/cs_synth.h"

/* benchmark_models.rs:323-328 / batch.rs:316-324 */
float cs_oracle_cosine(const float* a, const float* b, size_t dim) {
    float dot = 0.0f, sa = 0.0f, sb = 0.0f;
    for (size_t i = 0; i < dim; ++i) dot += a[i] * b[i];
    for (size_t i = 0; i < dim; ++i) sa += a[i] * a[i];
    for (size_t i = 0; i < dim; ++i) sb += b[i] * b[i];
    float mag_a = sqrtf(sa), mag_b = sqrtf(sb);
    if (mag_a == 0.0f || mag_b == 0.0f) return 0.0f; /* batch.rs:320-322 */
    return dot / (mag_a * mag_b);
}

double cs_oracle_cosine_f64(const float* a, const float* b, size_t dim) {
    double dot = 0.0, sa = 0.0, sb = 0.0;
    for (size_t i = 0; i < dim; ++i) {
        dot += (double)a[i] * (double)b[i];
        sa += (double)a[i] * (double)a[i];
        sb += (double)b[i] * (double)b[i];
    }
    if (sa == 0.0 || sb == 0.0) return 0.0;
    return dot / (sqrt(sa) * sqrt(sb));
}

static inline int is_dead(const uint32_t* dead, uint64_t row) {
    return dead && ((dead[row >> 5] >> (row & 31)) & 1u);
}

/* Sorted best-first list under (cos desc, id asc).  Rows arrive in ascending id, so a
 * newcomer that ties the current worst does not displace it: the `>` of
 * benchmark_models.rs:160. */
typedef struct { float c; uint32_t id; } hit32;
typedef struct { double c; uint32_t id; } hit64;

static inline int better32(float c, uint32_t id, hit32 o) {
    return c > o.c || (c == o.c && id < o.id);
}

static uint32_t insert32(hit32* list, uint32_t cnt, uint32_t k, float c, uint32_t id) {
    if (c != c) return cnt; /* NaN never wins a `>` */
    if (cnt == k && !better32(c, id, list[k - 1])) return cnt;
    uint32_t pos = cnt < k ? cnt : k - 1;
    while (pos > 0 && better32(c, id, list[pos - 1])) {
        list[pos] = list[pos - 1];
        --pos;
    }
    list[pos].c = c;
    list[pos].id = id;
    return cnt < k ? cnt + 1 : k;
}

uint32_t cs_oracle_scan_topk(const float* corpus, uint64_t n, uint32_t dim, const float* q,
                             uint32_t k, const uint32_t* dead, uint32_t id_base,
                             float* out_cos, uint32_t* out_ids) {
    if (k == 0) return 0;
    hit32* list = (hit32*)malloc(sizeof(hit32) * k);
    uint32_t cnt = 0;
    for (uint64_t r = 0; r < n; ++r) { /* benchmark_models.rs:158 */
        if (is_dead(dead, r)) continue;
        float c = cs_oracle_cosine(q, corpus + r * dim, dim);
        cnt = insert32(list, cnt, k, c, id_base + (uint32_t)r);
    }
    for (uint32_t i = 0; i < cnt; ++i) { out_cos[i] = list[i].c; out_ids[i] = list[i].id; }
    free(list);
    return cnt;
}

uint32_t cs_oracle_scan_topk_f64(const float* corpus, uint64_t n, uint32_t dim,
                                 const float* q, uint32_t k, const uint32_t* dead,
                                 uint32_t id_base, double* out_cos, uint32_t* out_ids) {
    if (k == 0) return 0;
    hit64* list = (hit64*)malloc(sizeof(hit64) * k);
    uint32_t cnt = 0;
    for (uint64_t r = 0; r < n; ++r) {
        if (is_dead(dead, r)) continue;
        double c = cs_oracle_cosine_f64(q, corpus + r * dim, dim);
        uint32_t id = id_base + (uint32_t)r;
        if (c != c) continue;
        if (cnt == k && !(c > list[k - 1].c || (c == list[k - 1].c && id < list[k - 1].id)))
            continue;
        uint32_t pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && (c > list[pos - 1].c || (c == list[pos - 1].c && id < list[pos - 1].id))) {
            list[pos] = list[pos - 1];
            --pos;
        }
        list[pos].c = c;
        list[pos].id = id;
        if (cnt < k) ++cnt;
    }
    for (uint32_t i = 0; i < cnt; ++i) { out_cos[i] = list[i].c; out_ids[i] = list[i].id; }
    free(list);
    return cnt;
}

/* ---- tuned port: the fair CPU ceiling beside the literal loop ----------------------- */

__attribute__((target_clones("avx512f", "avx2", "default")))
static void dot_ss8(const float* q, const float* x, uint32_t dim, float* dot, float* ss) {
    float d[16] = {0}, s[16] = {0};
    uint32_t i = 0;
    for (; i + 16 <= dim; i += 16)
        for (int j = 0; j < 16; ++j) {
            d[j] += q[i + j] * x[i + j];
            s[j] += x[i + j] * x[i + j];
        }
    float dd = 0.0f, sq = 0.0f;
    for (int j = 0; j < 16; ++j) { dd += d[j]; sq += s[j]; }
    for (; i < dim; ++i) { dd += q[i] * x[i]; sq += x[i] * x[i]; }
    *dot = dd;
    *ss = sq;
}

int cs_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

uint32_t cs_oracle_merge_topk(const float* cos, const uint32_t* ids, const uint32_t* counts,
                              uint32_t nlists, uint32_t k, float* out_cos,
                              uint32_t* out_ids) {
    if (k == 0) return 0;
    hit32* list = (hit32*)malloc(sizeof(hit32) * k);
    uint32_t cnt = 0;
    for (uint32_t l = 0; l < nlists; ++l)
        for (uint32_t i = 0; i < counts[l]; ++i)
            cnt = insert32(list, cnt, k, cos[(size_t)l * k + i], ids[(size_t)l * k + i]);
    for (uint32_t i = 0; i < cnt; ++i) { out_cos[i] = list[i].c; out_ids[i] = list[i].id; }
    free(list);
    return cnt;
}

uint32_t cs_oracle_scan_topk_omp(const float* corpus, uint64_t n, uint32_t dim,
                                 const float* q, uint32_t k, const uint32_t* dead,
                                 uint32_t id_base, int threads, float* out_cos,
                                 uint32_t* out_ids) {
    if (k == 0) return 0;
    int nt = threads > 0 ? threads : cs_oracle_num_threads();
    if ((uint64_t)nt > n) nt = n ? (int)n : 1;
    float qs = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) qs += q[i] * q[i];
    const float qmag = sqrtf(qs);
    float* pc = (float*)malloc(sizeof(float) * (size_t)nt * k);
    uint32_t* pi = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nt * k);
    uint32_t* pn = (uint32_t*)calloc((size_t)nt, sizeof(uint32_t));
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
        hit32* list = (hit32*)malloc(sizeof(hit32) * k);
        uint32_t cnt = 0;
        for (uint64_t r = lo; r < hi; ++r) {
            if (is_dead(dead, r)) continue;
            float dot, ss;
            dot_ss8(q, corpus + r * dim, dim, &dot, &ss);
            float xm = sqrtf(ss);
            float c = (qmag == 0.0f || xm == 0.0f) ? 0.0f : dot / (qmag * xm);
            cnt = insert32(list, cnt, k, c, id_base + (uint32_t)r);
        }
        for (uint32_t i = 0; i < cnt; ++i) {
            pc[(size_t)t * k + i] = list[i].c;
            pi[(size_t)t * k + i] = list[i].id;
        }
        pn[t] = cnt;
        free(list);
    }
    uint32_t cnt = cs_oracle_merge_topk(pc, pi, pn, (uint32_t)nt, k, out_cos, out_ids);
    free(pc); free(pi); free(pn);
    return cnt;
}

/* ---- synthetic data --------------------------------------------------------------- */

void cs_oracle_synth_rows(uint64_t seed, uint64_t first_row, uint64_t n, uint32_t dim,
                          float* out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t r = 0; r < (int64_t)n; ++r)
        for (uint32_t c = 0; c < dim; ++c)
            out[(uint64_t)r * dim + c] = cs_synth_value(seed, (first_row + (uint64_t)r) * dim + c);
}

void cs_oracle_synth_planted(uint64_t seed_c, uint64_t seed_q, const uint64_t* rows,
                             uint64_t nq, uint32_t dim, float* out) {
    for (uint64_t i = 0; i < nq; ++i)
        for (uint32_t c = 0; c < dim; ++c)
            out[i * dim + c] = cs_synth_planted(seed_c, seed_q, rows[i], dim, i, c);
}
