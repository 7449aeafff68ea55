// gemm_q8_dev.hpp — device helpers shared by the dynamic-quantisation kernels (gemm_q8.hip, gemm_q8_slab.hip): range slots,
// DynamicQuantizeLinear's parameters, the FFN-up output range from three extremes of y, the re-quantised GELU byte table.
#pragma once
#include "encoder.hpp"
#include "gemm_epilogue.hpp"
#include "gemm_q8.hpp"
#include "split_f16.hpp"

namespace cs {

typedef int q8_i32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- the range of a quantisation unit ------------------------------------------------------------------------------
// lo <= 0 <= hi always (the graph's range includes zero), so both start from +0.0f = all bits zero and move by integer
// atomics on the float's bits: non-negative floats order like their bits (atomicMax), negative floats like their bits
// reversed (atomicMax on the unsigned pattern finds the most negative).
// A range only widens, so a value that does not beat what the slot already shows (a plain load, possibly stale) can be
// dropped without the atomic (the load is agent-scope, so it is not served from another XCD's stale line): after the first few waves almost every update is — 32,768 waves hammering two addresses
// took 380 us per tensor before this check.
__device__ __forceinline__ void q8_range_update(uint32_t* slot, float lo, float hi) {
    if (lo < 0.0f) {
        const uint32_t b = __float_as_uint(lo);
        if (b > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, b);
    }
    if (hi > 0.0f) {
        const uint32_t b = __float_as_uint(hi);
        if (b > __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot + 1, b);
    }
}

// (x_scale, x_zp) of a unit from its range: DynamicQuantizeLinear's arithmetic, f32, one rounding per operation.
__device__ __forceinline__ void q8_params_of(float lo, float hi, float& xs, float& xz) {
    xs = hi == lo ? 1.0f : __fdiv_rn(__fsub_rn(hi, lo), 255.0f);
    const float z = __fsub_rn(0.0f, __fdiv_rn(lo, xs));
    xz = rintf(fminf(fmaxf(z, 0.0f), 255.0f));  // round half to even
}
__device__ __forceinline__ void q8_params(const uint32_t* slot, float& xs, float& xz) {
    q8_params_of(__uint_as_float(slot[0]), __uint_as_float(slot[1]), xs, xz);
}

// ---- the range of GELU(y) from three extremes of y ---------------------------------------------------------------
// x Phi(x) rises for x > c = -0.75179..., falls for x < c, and is <= 0 exactly where x <= 0.  So over a tensor
//   hi = max(0, gelu(max y)),   lo = min(0, gelu(a), gelu(b)),  a = the largest y <= c,  b = the smallest y >= c
// and the range pass of FFN-up only has to track max y, a and b (three compares per element) instead of evaluating
// the GELU (~18 VALU slots per element, 100M elements per layer at 65,536 rows).  The three travel as order-preserving
// unsigned keys, all "larger is better" (b negated), so an all-zero slot means "none yet" for each.
constexpr float kGeluArgMin = -0.7517916f;
__device__ __forceinline__ uint32_t q8_key(float x) {
    const uint32_t b = __float_as_uint(x);
    return b ^ (uint32_t)(((int32_t)b >> 31) | (int32_t)0x80000000);
}
__device__ __forceinline__ float q8_unkey(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}
__device__ __forceinline__ void q8_key_update(uint32_t* word, float x) {
    const uint32_t k = q8_key(x);
    if (k > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, k);
}
__device__ __forceinline__ void q8_params_gelu(const uint32_t* slot, float& xs, float& xz) {
    float lo = 0.0f, hi = 0.0f;
    if (slot[2]) hi = fmaxf(hi, sh_gelu_erf(q8_unkey(slot[2])));
    if (slot[3]) lo = fminf(lo, sh_gelu_erf(q8_unkey(slot[3])));
    if (slot[4]) lo = fminf(lo, sh_gelu_erf(-q8_unkey(slot[4])));
    q8_params_of(lo, hi, xs, xz);
}

enum { Q8_EPI_GELU_RANGE = 100, Q8_EPI_GELU_Q8 = 101 };
struct Q8Requant {
    uint32_t* range;       // (lo, hi) of the output tensor: widened by pass 1, read by pass 2
    int8_t* out;           // [M][N] s8 (pass 2)
    Q8RowMeta* rmeta_out;  // [M]: rowsum zeroed by pass 1, accumulated by pass 2; xs / za written by pass 2
    uint32_t use_table;    // pass 2 of the row-block kernel, one unit: the output byte by table lookup (CS_Q8_GELU_TABLE=0: direct)
};

// ---- FFN-up store pass: the re-quantised GELU byte as a table lookup in the y domain ----------------------------------------
// Per output the store pass evaluated erf-GELU, a reciprocal multiply, the tie test, rint, clamp: ~43 of the ~48 VALU
// instructions it spends per element, on a kernel whose vector issue is busy 0.72 of its time
// (profiles/r04_q8_sq_counters.txt).  With ONE quantisation unit the output parameters (g_scale, g_zp) are the same for the
// whole launch, so the byte is a FUNCTION of y alone:  B(y) = clamp(rint(gelu(y) / g_scale) + g_zp - 128)  — a step function,
// falling one step at a time up to the GELU's minimum at c = -0.7518 and rising one step at a time behind it, and two steps
// are never closer than g_scale / max|gelu'| = g_scale / 1.13 in y.  Every block therefore builds, once, a table over
// buckets of width 0.8 g_scale: { thr, byte_left | byte_right << 8 } — at most one step per bucket, its threshold found by
// bisection on B ITSELF (the function below: the arithmetic the direct form runs), so a lookup returns what the direct form
// returns except inside the few-ulp neighbourhood of a threshold where the f32 pipeline is not monotone (a 1e-5 fraction
// of values, two orders below the boundary flips between any two erf implementations).  Lookup: one fma, a conversion, an
// unsigned min, a shift-add, ds_read_b64, a shift, a compare, a select.  The table covers [QG_YL, max y]; everything below
// QG_YL reads the last entry (checked constant by the builder); a configuration the table cannot represent (more than
// QG_NB buckets: a degenerate range; two steps in one bucket: the bucket around c when a rounding boundary falls inside its
// 1e-4-wide dip) makes the builder return 0 and the block runs the direct form.
constexpr int QG_NB = 2048;
constexpr float QG_YL = -16.0f;
struct Q8GeluEntry { float thr; uint32_t w; };  // byte = low 8 bits of (y >= thr ? w >> 8 : w)
constexpr int QG_LDS = QG_NB * (int)sizeof(Q8GeluEntry);

__device__ __forceinline__ int q8_gelu_byte(float y, float gs, float rgs, float gz128) {  // the direct form (s8: uint8 - 128)
    const float v = sh_gelu_erf(y);
    const float t = v * rgs;
    float rt = rintf(t);
    // |t - rint(t)| <= 0.5: within 1e-3 of a tie exactly when it exceeds 0.499
    if (fabsf(t - rt) > 0.499f) rt = rintf(__fdiv_rn(v, gs));
    return (int)fminf(fmaxf(__fadd_rn(rt, gz128), -128.0f), 127.0f);
}

// All `nthreads` threads of the block.  Returns 1 / bucket width, or 0 when the block must run the direct form.  `ok` is a
// word of LDS the caller has zeroed (behind a barrier).
__device__ __forceinline__ float q8_build_gelu_table(Q8GeluEntry* tbl, uint32_t* ok_bad, float gs, float rgs, float gz128, float ymax,
                                                     int nthreads) {
    const float w = gs * 0.8f, span = ymax - QG_YL;
    if (!(w > 0.0f) || !(span > 0.0f) || !(span / w < (float)(QG_NB - 4))) return 0.0f;   // (block-uniform)
    const float inv_w = __fdiv_rn(1.0f, w);
    const int nb = (int)(span * inv_w) + 2;          // buckets a value <= ymax can index
    bool bad = false;
    for (int i = threadIdx.x; i < nb; i += nthreads) {
        // a value indexes bucket i when floor((y - YL) inv_w) == i in f32: inside [left, right] with room for those roundings
        const float left = QG_YL + ((float)i - 0.02f) * w, right = QG_YL + ((float)i + 1.02f) * w;
        const int bl = q8_gelu_byte(left, gs, rgs, gz128), br = q8_gelu_byte(right, gs, rgs, gz128);
        Q8GeluEntry e;
        e.thr = INFINITY;
        e.w = (uint32_t)(bl & 0xff) | ((uint32_t)(bl & 0xff) << 8);
        if (left <= kGeluArgMin && right >= kGeluArgMin) {   // the turning point: flat unless a rounding boundary sits in the dip
            if (bl != br || q8_gelu_byte(kGeluArgMin, gs, rgs, gz128) != bl) bad = true;
        } else if (bl != br) {
            if (br - bl != 1 && bl - br != 1) bad = true;
            float lo = left, hi = right;                      // B(lo) == bl, B(hi) == br
            for (int it = 0; it < 40; ++it) {
                const float mid = 0.5f * (lo + hi);
                if (!(mid > lo && mid < hi)) break;
                if (q8_gelu_byte(mid, gs, rgs, gz128) == bl) lo = mid; else hi = mid;
            }
            e.thr = hi;
            e.w = (uint32_t)(bl & 0xff) | ((uint32_t)(br & 0xff) << 8);
        }
        tbl[i] = e;
    }
    if (threadIdx.x == 0) {  // everything below QG_YL: one byte (gelu(y) is -0.5 |y| 1.5e-8 there: far inside one step)
        const int b0 = q8_gelu_byte(QG_YL, gs, rgs, gz128);
        if (q8_gelu_byte(-3.0e4f, gs, rgs, gz128) != b0 || q8_gelu_byte(QG_YL - 1.0f, gs, rgs, gz128) != b0) bad = true;
        Q8GeluEntry e;
        e.thr = INFINITY;
        e.w = (uint32_t)(b0 & 0xff) | ((uint32_t)(b0 & 0xff) << 8);
        tbl[QG_NB - 1] = e;
    }
    if (bad) atomicOr(ok_bad, 1u);
    __syncthreads();
    return *ok_bad ? 0.0f : inv_w;
}

}  // namespace
}  // namespace cs
