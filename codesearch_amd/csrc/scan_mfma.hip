// scan_mfma.hip — batched-query cosine scan on the f32 MFMA: the
// [rows, dim] x [dim, Q] product is never materialised.  Replaces the reference's
// `variants.par_iter().map(|e| store.search(e, limit))` (src/search/mod.rs:508-511) and
// serves BASELINE.json configs 4/5 (64 and 1000 batched queries).
//
// Scoring: one wave owns a 32-row corpus tile; the row tile is the MFMA A operand and a
// resident tile of 32*NQT queries (LDS) is the B operand (v_mfma_f32_32x32x2_f32, exact
// f32).  Rows stream HBM -> registers (coalesced 128 B row segments) -> a wave-private
// LDS chunk -> ds_read_b128 fragments; rows are read once per query tile.
//
// Selection: no per-wave lists.  The scan runs in row-ordered PHASES of geometrically
// growing size; a phase appends every (row, query) whose cosine beats the query's
// threshold tau to a per-query candidate buffer, then a select kernel folds candidates into
// the running best-k ("carry") and raises tau to the k-th best seen so far.  tau is always
// the k-th best of rows ALREADY scanned, hence a true lower bound of the final k-th best:
// nothing that belongs to the top-k is ever dropped, and because phases go in ascending id
// order a later row that ties tau loses the (cosine desc, id asc) tie-break, so `c > tau`
// is exact.  Expected candidates per phase are ~ k * growth; if a buffer still overflows
// (adversarially ordered data) the caller reruns those queries on the list-based kernel.
#include "scan.hpp"
#include "block_select.hpp"

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MB_WAVES = 8;             // waves per block (1 block per CU)
constexpr int MB_THREADS = MB_WAVES * 64;
constexpr int MB_KC = 32;               // K chunk (floats) staged per step
constexpr int MB_RS = 36;               // padded row stride of the staged chunk (floats)
constexpr int MB_SEL_THREADS = 1024;
constexpr int MB_SEL_CAP = 4096;  // keys sorted per chunk: k carried (<= 1024) + up to 4096 - k new

__device__ __forceinline__ float half_sum32(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// |row| for rows [first, first+n): same lane layout as the single-query scan (32 lanes x
// float4 per 128 floats), one row per half-wave.
__global__ void __launch_bounds__(256)
row_norms_kernel(const float* __restrict__ corpus, uint64_t first, uint64_t n, uint32_t dim,
                 float* __restrict__ norms) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const uint64_t pair = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t r = first + pair * 2 + half;
    if (pair * 2 >= n) return;
    const bool ok = (pair * 2 + half) < n;
    float ss = 0.0f;
    if (ok) {
        const float* p = corpus + r * dim;
        for (uint32_t c = l32 * 4; c < dim; c += 128) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
            ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss);
            ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
        }
    }
    ss = half_sum32(ss);
    if (ok && l32 == 0) norms[r] = sqrtf(ss);
}

// Score rows [row_lo, row_hi) against queries [q0, q0 + 32*NQT) and append candidates.
// grid = (query tiles, row blocks); dynamic LDS = queries [32*NQT][dim+4] | waves x [32][36].
template <int NQT, bool NT>
__global__ void __launch_bounds__(MB_THREADS)
score_append_kernel(const float* __restrict__ corpus, const float* __restrict__ norms,
                    uint64_t row_lo, uint64_t row_hi, uint32_t dim,
                    const float* __restrict__ queries, uint32_t nq,
                    const float* __restrict__ tau, const uint32_t* __restrict__ dead,
                    RowIds id_base, uint64_t* __restrict__ cand, uint32_t* __restrict__ cnt,
                    uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t QS = dim + 4;  // (dim+4) % 64 == 4 for dim in {384, 768, 1024}: conflict-free b128
    float* Qs = smem;                                  // [32*NQT][QS]
    float* stage = smem + (size_t)32 * NQT * QS;       // [MB_WAVES][32][MB_RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t q0 = blockIdx.x * 32 * NQT;

    // resident query tile (+ zero rows past nq)
    for (uint32_t idx = tid; idx < 32u * NQT * (dim / 4); idx += MB_THREADS) {
        const uint32_t j = idx / (dim / 4), c4 = idx % (dim / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + j < nq) v = *reinterpret_cast<const f32x4*>(queries + (size_t)(q0 + j) * dim + c4 * 4);
        *reinterpret_cast<f32x4*>(Qs + (size_t)j * QS + c4 * 4) = v;
    }
    __syncthreads();
    float qmag[NQT], thr[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const float* qr = Qs + (size_t)(t * 32 + l31) * QS;
        float s = 0.0f;
        for (uint32_t c = 0; c < dim; ++c) s = fmaf(qr[c], qr[c], s);
        qmag[t] = sqrtf(s);
        const uint32_t q = q0 + t * 32 + l31;
        thr[t] = (q < nq) ? tau[q] : __builtin_huge_valf();  // padded queries never append
    }

    float* my = stage + (size_t)wave * 32 * MB_RS;
    const uint64_t ntiles = (row_hi - row_lo + 31) / 32;
    const uint64_t gw = (uint64_t)blockIdx.y * MB_WAVES + wave;
    const uint64_t nw = (uint64_t)gridDim.y * MB_WAVES;
    const uint32_t nchunks = dim / MB_KC;
    // staging map: instruction t covers rows 8t..8t+7, lane -> (row 8t + lane/8, 16 B piece lane%8)
    const int srow = lane >> 3, spiece = lane & 7;

    for (uint64_t tile = gw; tile < ntiles; tile += nw) {
        const uint64_t r0 = row_lo + tile * 32;
        const float* src[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint64_t r = r0 + 8 * t + srow;
            r = r < row_hi ? r : row_hi - 1;  // tail: re-read the last row, masked at append
            src[t] = corpus + r * dim + spiece * 4;
        }
        f32x16 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        f32x4 g[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if constexpr (NT) g[t] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src[t]));
            else g[t] = *reinterpret_cast<const f32x4*>(src[t]);
        }
        for (uint32_t kc = 0; kc < nchunks; ++kc) {
            // publish chunk kc to the wave-private LDS tile (LDS ops of one wave retire in order)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *reinterpret_cast<f32x4*>(my + (8 * t + srow) * MB_RS + spiece * 4) = g[t];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (kc + 1 < nchunks) {  // next chunk in flight under this chunk's MFMAs
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4* p = reinterpret_cast<const f32x4*>(src[t] + (kc + 1) * MB_KC);
                    if constexpr (NT) g[t] = __builtin_nontemporal_load(p);
                    else g[t] = *p;
                }
            }
            f32x4 a[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                a[c] = *reinterpret_cast<const f32x4*>(my + l31 * MB_RS + 16 * h + 4 * c);
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                const float* qr = Qs + (size_t)(t * 32 + l31) * QS + kc * MB_KC + 16 * h;
                f32x4 b[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const f32x4*>(qr + 4 * c);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][e], b[c][e], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // acc[t][r] = dot(row r0 + (r&3) + 8*(r>>2) + 4h, query q0 + 32t + (lane&31))
        float xm[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const uint64_t rr = r0 + 8 * gq + 4 * h;  // 4 consecutive rows
            if (rr + 3 < row_hi) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(norms + rr);
                xm[4 * gq] = v.x; xm[4 * gq + 1] = v.y; xm[4 * gq + 2] = v.z; xm[4 * gq + 3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xm[4 * gq + e] = (rr + e < row_hi) ? norms[rr + e] : 0.0f;
            }
        }
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const uint32_t q = q0 + t * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint64_t row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float c = (qmag[t] == 0.0f || xm[r] == 0.0f) ? 0.0f : acc[t][r] / (qmag[t] * xm[r]);
                if (c > thr[t] && row < row_hi) {  // rare, divergent, short
                    if (!dead || !((dead[row >> 5] >> (row & 31)) & 1u)) {
                        const uint32_t pos = atomicAdd(&cnt[(size_t)q * kCntStride], 1u);
                        if (pos < cap) cand[(size_t)q * cap + pos] = key_pack(c, id_base.of(row));
                    }
                }
            }
        }
    }
}

// Fold a query's candidates into its running best-k.  One block per query.
//   carry[q][k]   : best k keys so far (0 = empty), best first; updated in place
//   tau[q]        : cosine of the k-th best if k rows have been seen, else -inf
//   cnt[q]        : reset to 0;  overflow[0] |= 1 if cnt[q] > cap
// When `final_out` the decoded results are also written.
__global__ void __launch_bounds__(MB_SEL_THREADS)
select_candidates_kernel(const uint64_t* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap,
                         uint32_t k, uint64_t* __restrict__ carry, float* __restrict__ tau,
                         uint32_t* __restrict__ overflow, int final_out,
                         uint64_t* __restrict__ out_keys, float* __restrict__ out_cos,
                         uint32_t* __restrict__ out_ids, uint32_t* __restrict__ out_counts) {
    __shared__ __attribute__((aligned(16))) uint64_t a[MB_SEL_CAP];
    __shared__ uint32_t live;
    __shared__ uint32_t sel_slots[66];
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    uint32_t n = cnt[(size_t)q * kCntStride];
    if (n > cap) {
        // [1] is the sticky word cs_index_search_status reports: only searches of more than kGatedMaxQ queries (one
        // block per query here) rely on it — smaller ones carry the gated exact rerun and repair themselves
        if (tid == 0) { atomicOr(overflow, 1u); if (gridDim.x > kGatedMaxQ) atomicOr(overflow + 1, 1u); }
        n = cap;
    }
    const uint64_t* src = cand + (size_t)q * cap;
    for (uint32_t i = tid; i < MB_SEL_CAP; i += MB_SEL_THREADS) a[i] = (i < k) ? carry[(size_t)q * k + i] : 0ull;
    if (tid == 0) live = 0;
    __syncthreads();
    const uint32_t room = MB_SEL_CAP - k;
    for (uint32_t done = 0; done < n || done == 0; done += room) {
        const uint32_t take = (n - done) < room ? (n - done) : room;
        // sort size: the k carried keys + this chunk, padded with zero keys to a power of two (a few
        // dozen candidates sort in 64..256 slots, a k = 200 phase in 1024)
        uint32_t nsort = 64;
        while (nsort < k + take) nsort <<= 1;
        for (uint32_t i = tid; k + i < nsort; i += MB_SEL_THREADS) a[k + i] = (i < take) ? src[done + i] : 0ull;
        if (nsort > 256) {  // hundreds of candidates: bracket the k-th key first, sort only what is above it
            __syncthreads();
            nsort = block_select_topk<MB_SEL_THREADS, MB_SEL_CAP / MB_SEL_THREADS>(a, k + take, k, tid, sel_slots);
        }
        uint32_t prev_stride = 128;  // block barrier only around cross-segment stages (block_bitonic_desc, scan.hip)
        for (uint32_t size = 2; size <= nsort; size <<= 1)
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                if (stride >= 128 || prev_stride >= 128) {
                    __syncthreads();
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                prev_stride = stride;
                for (uint32_t t = tid; t < nsort / 2; t += MB_SEL_THREADS) {
                    const uint32_t i = 2 * t - (t & (stride - 1)), j = i + stride;
                    const uint64_t x = a[i], y = a[j];
                    if ((x < y) == ((i & size) == 0)) { a[i] = y; a[j] = x; }
                }
            }
        __syncthreads();
        if (n == 0) break;
    }
    for (uint32_t i = tid; i < k; i += MB_SEL_THREADS) {
        const uint64_t key = a[i];
        carry[(size_t)q * k + i] = key;
        if (final_out) {
            if (key) atomicAdd(&live, 1u);
            if (out_keys) out_keys[(size_t)q * k + i] = key;
            if (out_cos) out_cos[(size_t)q * k + i] = key ? key_cos(key) : 0.0f;
            if (out_ids) out_ids[(size_t)q * k + i] = key ? key_id(key) : 0xffffffffu;
        }
    }
    if (tid == 0) {
        const uint64_t kth = a[k - 1];
        tau[q] = kth ? key_cos(kth) : -__builtin_huge_valf();
        cnt[(size_t)q * kCntStride] = 0;
    }
    __syncthreads();
    if (final_out && out_counts && tid == 0) out_counts[q] = live;
}

int32_t launch_select_candidates(const BatchedState& st, uint32_t nq, uint32_t cap, uint32_t k, bool last,
                                 uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids,
                                 uint32_t* d_out_counts, hipStream_t stream) {
    hipLaunchKernelGGL(select_candidates_kernel, dim3(nq), dim3(MB_SEL_THREADS), 0, stream, st.d_cand, st.d_cnt,
                       cap, k, st.d_carry, st.d_tau, st.d_overflow, last ? 1 : 0, d_out_keys, d_out_cos, d_out_ids,
                       d_out_counts);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

__global__ void init_batched_state_kernel(float* tau, uint32_t* cnt, uint64_t* carry, uint32_t nq,
                                          uint32_t k, uint32_t* overflow) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) { tau[i] = -__builtin_huge_valf(); cnt[(size_t)i * kCntStride] = 0; }
    if (i < nq * k) carry[i] = 0ull;
    if (i == 0) { overflow[2] += overflow[0]; overflow[0] = 0; }
}

// ---- host side ----------------------------------------------------------------------------

size_t batched_lds_bytes(uint32_t dim, int nqt) {
    return ((size_t)32 * nqt * (dim + 4) + (size_t)MB_WAVES * 32 * MB_RS) * sizeof(float);
}

bool batched_supported(uint32_t dim) {
    return (dim == 384 || dim == 768) && batched_lds_bytes(dim, 1) <= 160 * 1024 - 256;
}

uint32_t batched_cap(uint32_t k) {
    const uint32_t c = 64 * k;
    return c < 4096 ? 4096 : c;
}


int32_t launch_row_norms(const float* d_corpus, uint64_t first, uint64_t n, uint32_t dim,
                         float* d_norms, hipStream_t stream) {
    if (n == 0) return CS_OK;
    if (dim % 4) return fail(CS_ERR_UNSUPPORTED, "row norms need dim %% 4 == 0");
    const uint64_t pairs = (n + 1) / 2;
    hipLaunchKernelGGL(row_norms_kernel, dim3((uint32_t)((pairs + 3) / 4)), dim3(256), 0, stream, d_corpus,
                       first, n, dim, d_norms);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_scan_batched(const BatchedState& st, const float* d_corpus, const float* d_norms,
                            uint64_t n_rows, uint32_t dim, const float* d_queries, uint32_t nq, uint32_t k,
                            const uint32_t* d_dead, RowIds id_base, int num_cus, uint64_t* d_out_keys,
                            float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_counts,
                            hipStream_t stream) {
    const uint32_t cap = batched_cap(k);
    int nqt = nq > 32 ? 2 : 1;
    if (batched_lds_bytes(dim, nqt) > 160 * 1024 - 256) nqt = 1;
    const uint32_t qtiles = (nq + 32 * nqt - 1) / (32 * nqt);
    const size_t lds = batched_lds_bytes(dim, nqt);
    static PerDeviceOnce attr_set;  // function attributes are per device
    CS_TRY(attr_set.run([&]() -> int32_t {
        const void* fns[4] = {reinterpret_cast<const void*>(score_append_kernel<1, true>),
                              reinterpret_cast<const void*>(score_append_kernel<1, false>),
                              reinterpret_cast<const void*>(score_append_kernel<2, true>),
                              reinterpret_cast<const void*>(score_append_kernel<2, false>)};
        for (const void* f : fns)
            CS_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        return CS_OK;
    }));
    if (lds > 160 * 1024 - 256) return fail(CS_ERR_UNSUPPORTED, "query tile does not fit LDS at dim %u", dim);
    {
        const uint32_t n = nq * k > nq ? nq * k : nq;
        hipLaunchKernelGGL(init_batched_state_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, st.d_tau,
                           st.d_cnt, st.d_carry, nq, k, st.d_overflow);
    }
    uint64_t done = 0;
    uint64_t phase = n_rows < cap ? n_rows : cap;  // phase 0: tau = -inf, every row is a candidate
    const uint32_t growth = 16;
    do {
        const uint64_t lo = done, hi = done + phase;
        if (hi > lo) {
            const uint64_t tiles = (hi - lo + 31) / 32;
            uint64_t rb = (tiles + MB_WAVES - 1) / MB_WAVES;
            uint64_t max_rb = (uint64_t)num_cus / qtiles;
            if (max_rb < 1) max_rb = 1;
            if (rb > max_rb) rb = max_rb;
            dim3 grid(qtiles, (uint32_t)rb);
            // one query tile: rows are read once -> non-temporal; several: let L2/MALL share them
#define CS_LAUNCH_SCORE(NQT_, NT_)                                                                     \
    hipLaunchKernelGGL((score_append_kernel<NQT_, NT_>), grid, dim3(MB_THREADS), lds, stream, d_corpus, \
                       d_norms, lo, hi, dim, d_queries, nq, st.d_tau, d_dead, id_base, st.d_cand,      \
                       st.d_cnt, cap)
            if (nqt == 1) { if (qtiles == 1) CS_LAUNCH_SCORE(1, true); else CS_LAUNCH_SCORE(1, false); }
            else { if (qtiles == 1) CS_LAUNCH_SCORE(2, true); else CS_LAUNCH_SCORE(2, false); }
#undef CS_LAUNCH_SCORE
            CS_HIP(hipGetLastError());
        }
        done = hi;
        const bool last = done >= n_rows;
        CS_TRY(launch_select_candidates(st, nq, cap, k, last, d_out_keys, d_out_cos, d_out_ids, d_out_counts, stream));
        phase = done * growth;
        if (phase > n_rows - done) phase = n_rows - done;
    } while (done < n_rows);
    return CS_OK;
}

}  // namespace cs
