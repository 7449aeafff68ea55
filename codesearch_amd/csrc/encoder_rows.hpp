// encoder_rows.hpp — the row arithmetic of the encoder's LayerNorm (encoder.hip E1 / E4 / E6) as a device function two
// kernels share: the row kernels of encoder.hip (one launch per LayerNorm) and the one-launch forward of short queries
// (small_forward.hip), where the same LayerNorm runs as the prologue of the dense layer that reads it.  One wave per token
// row; NPL = H / 64 values per lane, held as pairs of consecutive columns (v[2p], v[2p+1] = columns 2*lane + 128*p + {0,1}).
#pragma once

#include "common.hpp"

namespace cs {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ int ln_col(int lane, int i) { return 2 * lane + 128 * (i >> 1) + (i & 1); }

// o = (v - mean) / sqrt(var + eps) * g + b over the wave's row (two passes: the mean, then the variance of the deviations)
template <int NPL>
__device__ __forceinline__ void ln_row_core(const float (&v)[NPL], const float* __restrict__ g, const float* __restrict__ b,
                                            float eps, int lane, float (&o)[NPL]) {
    constexpr float invH = 1.0f / (64.0f * NPL);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) s += v[i];
    const float mean = wave_sum(s) * invH;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    const float var = wave_sum(q) * invH;
    const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int p = 0; p < NPL / 2; ++p) {
        const int c = ln_col(lane, 2 * p);
        const float2 gv = *reinterpret_cast<const float2*>(g + c);
        const float2 bv = *reinterpret_cast<const float2*>(b + c);
        o[2 * p] = (v[2 * p] - mean) * inv * gv.x + bv.x;
        o[2 * p + 1] = (v[2 * p + 1] - mean) * inv * gv.y + bv.y;
    }
}

// The same for R rows at once (one wave, row r's values in v[r]): per row exactly ln_row_core's operations in its order —
// only interleaved, so that the R reduction chains (twelve dependent cross-lane steps each) overlap instead of queueing.
template <int NPL, int R>
__device__ __forceinline__ void ln_rows_core(const float (&v)[R][NPL], const float* __restrict__ g, const float* __restrict__ b,
                                             float eps, int lane, float (&o)[R][NPL]) {
    constexpr float invH = 1.0f / (64.0f * NPL);
    // (gamma / beta requested before the reductions: left in the last loop, where they are used, their memory round trip stood
    // behind the two shuffle chains of every short kernel that normalises a row as its prologue)
    float2 gv[NPL / 2], bv[NPL / 2];
#pragma unroll
    for (int p = 0; p < NPL / 2; ++p) {
        gv[p] = *reinterpret_cast<const float2*>(g + ln_col(lane, 2 * p));
        bv[p] = *reinterpret_cast<const float2*>(b + ln_col(lane, 2 * p));
    }
    asm volatile("" : : "v"(gv[0].x), "v"(bv[0].x));  // (keeps the requests up here)
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) s[r] += v[r][i];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], m, 64);
    float mean[R], q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = s[r] * invH;
        q[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { const float d = v[r][i] - mean[r]; q[r] = fmaf(d, d, q[r]); }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] += __shfl_xor(q[r], m, 64);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float var = q[r] * invH;
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int p = 0; p < NPL / 2; ++p) {
            o[r][2 * p] = (v[r][2 * p] - mean[r]) * inv * gv[p].x + bv[p].x;
            o[r][2 * p + 1] = (v[r][2 * p + 1] - mean[r]) * inv * gv[p].y + bv[p].y;
        }
    }
}

}  // namespace cs
