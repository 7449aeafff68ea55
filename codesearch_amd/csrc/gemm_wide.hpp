// gemm_wide.hpp — what the two forms of the persistent wide dense-layer kernel share (gemm_wide.hip: main loop on
// v_mfma_f32_16x16x32_f16; gemm_wide32.hip: the same block, stage image and epilogues on v_mfma_f32_32x32x16_f16).
#pragma once

#include "encoder.hpp"
#include "split_f16.hpp"

namespace cs {

constexpr int GW_BM = 128;
constexpr int GW_A_BYTES = GW_BM * 128;            // one k-chunk (32 k, hi + lo) of 128 A rows
constexpr int GW_PARAM_FLOATS = 4096;                // bias of up to 4,096 columns (LayerNorm: bias | gamma | beta of 384)
constexpr int GW_STATS = 5 * GW_BM * 4;            // LayerNorm epilogue: [4 column groups][128 rows] f32 partial sums + [128] row statistic
// Two block shapes share the kernel: WCN = 4 column waves -> 8 waves, 128 x 384 outputs, one block per CU (whole rows
// at N = 384: the LayerNorm epilogue); WCN = 2 -> 4 waves, 128 x 192 outputs, 80 KiB of LDS, TWO blocks per CU, so
// one block's epilogue (VALU conversions + stores) runs under the other's MFMAs.
// WRN = 4 (with WCN = 2; r06, diagnostic library only): 8 waves as 4 x 2, 256 x 192 outputs, one block per CU — a k-step stages
// 32 KiB of A + 24 KiB of W for the outputs the 128 x 384 block stages 64 KiB for (and two 128 x 192 blocks 80 KiB).  Built to test
// whether the 5 GB of operand tiles an encoder layer's four products pull from L2 into LDS bound them (748 us per layer = 6.7 TB/s,
// the rate MI355X_MICROARCH.md gives that path): they do not — the block is level with the others (profiles/r06_gemm_tall_block_ab.log).
template <int WCN, int WRN = 2>
struct GwGeom {
    static constexpr int BM = 64 * WRN;
    static constexpr int A_BYTES = BM * 128;                     // one k-chunk (32 k, hi + lo) of the block's A rows
    static constexpr int BN = 96 * WCN;
    static constexpr int WAVES = WRN * WCN;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int W_BYTES = BN * 128;
    static constexpr int STAGE = A_BYTES + W_BYTES;              // 65,536 | 40,960 | 57,344 (256 x 192)
    // WCN == 4 keeps the layer's bias (and the LayerNorm's gamma / beta) in LDS for the block's lifetime: the epilogue then
    // issues no global LOAD, so nothing in it waits on vmcnt (which retires in issue order: a load issued behind the
    // previous strip's stores, or behind the next tile's first DMAs, waits for all of them)
    static constexpr int PARAMS = WCN == 4 ? GW_PARAM_FLOATS * 4 : 0;
    static constexpr int LDS = 2 * STAGE + (WCN == 4 ? GW_STATS : 0) + PARAMS;
    static constexpr int A_PIECES = (BM / 8) / WAVES;            // LDS-DMA pieces (8 rows x 128 B) of A per wave and stage: 2 | 4 | 4
    static constexpr int W_PIECES = (BN / 8) / WAVES;            // ... of W: 6 | 6 | 3
    static constexpr int PIECES = A_PIECES + W_PIECES;           // 8 | 10 | 7
    // a wave's private epilogue patch inside the free stage buffer ([16 rows][100 floats] = 6,400 B used)
    static constexpr int PATCH = WAVES * 8192 <= STAGE ? 8192 : STAGE / WAVES / 16 * 16;
    static_assert(PATCH >= 6400, "the stage buffer must hold every wave's epilogue patch");
    static_assert(WRN == 2 || WCN == 2, "256-row blocks are built 192 columns wide");
};
constexpr uint32_t GW_LN_RESID_SPLIT = 1u, GW_LN_NO_F32 = 2u;  // ln_flags of the LayerNorm epilogue (launch_gemm_wide_ln)
constexpr int GW_OUT_LN = 16;  // epilogue: + bias + residual, LayerNorm over the 384 columns, store f32 AND split form

// this wave's LDS-DMA pieces of a stage: element offsets from A / W (32 bits: a [65536, 1536] operand is 2^27.6 elements)
template <int WCN, int WRN = 2>
struct GwSrc {
    uint32_t a[GwGeom<WCN, WRN>::A_PIECES];
    uint32_t w[GwGeom<WCN, WRN>::W_PIECES];
};

__device__ __forceinline__ float gw_erf_fast(float x) {  // gemm_epilogue.hpp sh_erf_fast
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
__device__ __forceinline__ float gw_gelu(float v) { return 0.5f * v * (1.0f + gw_erf_fast(v * 0.70710678118654752440f)); }
// v * sigmoid(v): hardware exp2 and reciprocal (~1 ulp each)
__device__ __forceinline__ float gw_silu(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

// gemm_wide32.hip: the 32 x 32 x 16 form of the kernel, same arguments as the launch inside gemm_wide.hip.  wcn = 4 | 2.
int32_t gemm_wide32_launch(int wcn, int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid,
                           float* C, _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s,
                           const float* ln_g, const float* ln_b, float ln_eps, uint32_t ln_flags);

}  // namespace cs
