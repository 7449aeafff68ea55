// gemm_epilogue.hpp — the epilogue the 128 x 128-tile dense-layer kernels share (gemm_split.hip: split-f16 operands;
// gemm_q8.hip: dynamic-quantised int8 operands): the block's C tile sits in LDS as [128 m][128 n] f32 and leaves as
//   SH_OUT_F32 / SH_OUT_F32_RESID / SH_OUT_PARTIAL   f32 rows (+ bias, + residual)
//   SH_OUT_SPLIT / SH_OUT_SPLIT_GELU                 split-f16 lines (+ bias, erf-GELU)
// with every global access a full 16 B per lane on consecutive lanes.  Device code only.
#pragma once
#include "encoder.hpp"
#include "split_f16.hpp"

namespace cs {

// erf-GELU.  ocml's erff costs ~34 VALU instructions per element with both of its branches taken in
// a wave (polynomial below |x| = 1, accurate-exp form above), and the FFN-up epilogue applies it to
// 64 elements per thread.  This branch-free form, erf(t) = 1 - 2^(-t q(t)) for t = |x| with q a
// degree-9 minimax fit of -log2(erfc(t))/t on [0, 4] (erf(4) = 1 - 1.5e-8), is 14: |error| <=
// 1.2e-7 absolute on erf (ocml: ~6e-8), i.e. <= 0.6e-7 |v| on GELU — below the split GEMM's own
// error.  The exact-f32 kernels keep erff.
__device__ __forceinline__ float sh_erf_fast(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    float q = 7.569788067485206e-07f;
    q = fmaf(q, t, -1.6365151168429293e-05f);
    q = fmaf(q, t, 0.00015192339196801186f);
    q = fmaf(q, t, -0.0007679605041630566f);
    q = fmaf(q, t, 0.002005203627049923f);
    q = fmaf(q, t, 0.0003252939786761999f);
    q = fmaf(q, t, -0.028044508770108223f);
    q = fmaf(q, t, 0.1484302133321762f);
    q = fmaf(q, t, 0.9184240698814392f);
    q = fmaf(q, t, 1.6279078722000122f);
    const float e = 1.0f - __builtin_amdgcn_exp2f(-(q * t));
    return __builtin_copysignf(e, x);
}
__device__ __forceinline__ float sh_gelu_erf(float v) {
    return 0.5f * v * (1.0f + sh_erf_fast(v * 0.70710678118654752440f));
}

// The split-form outputs (300-400 MB per launch) are stored non-temporal: each byte is touched once by this
// block, while the XCD's L2 is holding the weights and the A tiles its sibling n-tile blocks are about to
// read (QKV 193 -> 181 us, FFN-up 284 -> 270 us in the encoder).  The 100 MB f32 tensors keep the default
// policy on both sides: they are served from the Infinity Cache (a non-temporal residual load made the
// LayerNorm that follows 8.5 us slower for 2 us gained here; the policy on LayerNorm's and attention's own
// accesses cost the encoder +10 %).
template <int EPI, bool FULL, int WM>
__device__ __forceinline__ void gemm_sh_epilogue(const float* ctile, const float* __restrict__ bias,
                                                 const float* resid, float* C, _Float16* __restrict__ Cs,
                                                 uint32_t M, uint32_t N, uint32_t m0, uint32_t n0,
                                                 uint32_t* __restrict__ flag) {
    const int tid = threadIdx.x;
    if (EPI == SH_OUT_SPLIT_GELU || EPI == SH_OUT_SPLIT) {
        bool ovf = false;
        const int c8 = tid & 15;  // 8 consecutive n per thread
        const sh_f32x4 b0 = *reinterpret_cast<const sh_f32x4*>(bias + n0 + c8 * 8);
        const sh_f32x4 b1 = *reinterpret_cast<const sh_f32x4*>(bias + n0 + c8 * 8 + 4);
        const size_t nchunks = N / 32;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = (tid >> 4) + 8 * WM * it;
            const sh_f32x4 v0 = *reinterpret_cast<const sh_f32x4*>(ctile + row * 128 + c8 * 8);
            const sh_f32x4 v1 = *reinterpret_cast<const sh_f32x4*>(ctile + row * 128 + c8 * 8 + 4);
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, b;
                ovf |= sh_split(EPI == SH_OUT_SPLIT_GELU ? sh_gelu_erf(v0[e] + b0[e]) : v0[e] + b0[e], a, b);
                hi[e] = a; lo[e] = b;
                ovf |= sh_split(EPI == SH_OUT_SPLIT_GELU ? sh_gelu_erf(v1[e] + b1[e]) : v1[e] + b1[e], a, b);
                hi[4 + e] = a; lo[4 + e] = b;
            }
            if (FULL || m0 + row < M) {
                _Float16* dst = Cs + ((size_t)(m0 + row) * nchunks + (n0 >> 5) + (c8 >> 2)) * 64 + (c8 & 3) * 8;
                __builtin_nontemporal_store(hi, reinterpret_cast<f16x8*>(dst));
                __builtin_nontemporal_store(lo, reinterpret_cast<f16x8*>(dst + 32));
            }
        }
        if (ovf && flag) atomicOr(flag, 1u);
    } else {
        const int c4 = tid & 31;
        sh_f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (EPI != SH_OUT_PARTIAL) bv = *reinterpret_cast<const sh_f32x4*>(bias + n0 + c4 * 4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            sh_f32x4 rs[8];
            if (EPI == SH_OUT_F32_RESID) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const uint32_t row = m0 + (tid >> 5) + 4 * WM * (half * 8 + it);
                    rs[it] = *reinterpret_cast<const sh_f32x4*>(resid + (size_t)((FULL || row < M) ? row : M - 1) * N + n0 + c4 * 4);
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = (tid >> 5) + 4 * WM * (half * 8 + it);
                sh_f32x4 v = *reinterpret_cast<const sh_f32x4*>(ctile + row * 128 + c4 * 4);
                v += bv;
                if (EPI == SH_OUT_F32_RESID) v += rs[it];
#ifdef SH_ABLATE_NO_STORE
                asm volatile("" ::"v"(v));
#else
                if (FULL || m0 + row < M) {
                    // the 100 MB f32 outputs (QKV in f32 mode, pre-LayerNorm sums) fit the Infinity Cache and
                    // the next kernel reads them: default policy; the split outputs (3-4x larger) stream out
                    *reinterpret_cast<sh_f32x4*>(C + (size_t)(m0 + row) * N + n0 + c4 * 4) = v;
                }
#endif
            }
        }
    }
}

}  // namespace cs
