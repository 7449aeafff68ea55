// small_path.hpp — launch interface of the small-batch dense layers with a LayerNorm prologue (small_path.hip).
#pragma once

#include "encoder.hpp"

namespace cs {

constexpr uint32_t SP_MAX_ROWS = 256;  // token rows this path takes (workspace: 5 x SP_MAX_ROWS x H floats)

struct SpLnGemmArgs {
    // prologue 0: rows of Y [T, H] f32
    const float* Y;
    // prologue 1: parts [4][T][H] + parts_bias [H] + X [T, H] (the residual)
    const float* parts;
    const float* parts_bias;
    const float* X;
    // prologue 2: the embedding gather
    const int32_t* ids;
    const float *word, *pos, *type0;
    uint32_t L, vocab;
    // LayerNorm, and where the block of column tile 0 writes the normalised rows (never the buffer `X` is read from)
    const float *ln_g, *ln_b;
    float eps;
    float* Xout;
    // the product: W [N][H/32][64] split form, bias [N] -> Cs [T][N/32][64]
    const _Float16* W;
    const float* bias;
    _Float16* Cs;
    uint32_t T, N;
    uint32_t* flag;
};

bool small_path_supported(uint32_t H, uint32_t I, uint32_t T);
// epi: SH_OUT_SPLIT | SH_OUT_SPLIT_GELU; pro: 0 | 1 | 2 (above)
int32_t launch_sp_ln_gemm(int epi, int pro, const SpLnGemmArgs& a, uint32_t H, hipStream_t s);
// parts[ks][T][N] = A[:, K slice ks] W[:, K slice ks]^T for the four quarters of K = 4 H (A [T][K/32][64], W [N][K/32][64])
int32_t launch_sp_partial(const _Float16* A, const _Float16* W, float* parts, uint32_t T, uint32_t N, uint32_t H, hipStream_t s);

// E3 + E4 in one launch for sequences of up to 32 tokens of a 384-wide, 12-head model (sp_attn_proj_kernel): C [T][384] =
// attention(qkvs) W^T + bias + resid; qkvs [T][36][64] split form (Q heads | K heads | V heads), mask [T] (row-major [B][L]),
// W [384][12][64] split form
bool sp_attn_proj_supported(uint32_t H, uint32_t heads, uint32_t T, uint32_t L);
int32_t launch_sp_attn_proj(const _Float16* qkvs, const int32_t* mask, const _Float16* W, const float* bias, const float* resid, float* C,
                            uint32_t T, uint32_t L, uint32_t H, uint32_t heads, uint32_t* flag, hipStream_t s);

}  // namespace cs
