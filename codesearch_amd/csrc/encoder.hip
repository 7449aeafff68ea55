// encoder.hip — the BERT forward pass of the embedding path as hand-written gfx950 kernels
// (SURVEY.md §8a E1-E8).  Replaces what /root/reference reaches through
// `self.model.embed(text_refs, None)` (src/embed/embedder.rs:286-289): fastembed -> ONNX
// Runtime CPU executing the BAAI/bge-small-en-v1.5 graph, then pooling + L2 normalise.
//
//   E1  embed_ln_kernel      word + type + position gather, LayerNorm            (HBM)
//   E2/E4/E5/E6 gemm_f32_kernel  y = x W^T + b [+gelu | +residual]; exact-f32 MFMA
//                            v_mfma_f32_32x32x2_f32, 128x128x32 tiles through LDS  (MFMA)
//   E3  attention_kernel     per (batch, head, 128 queries): S^T = K Q^T and O^T = V^T P on
//                            the f32 MFMA with the query on the lane, so the online
//                            softmax is in-lane; K/V of the head staged in LDS    (MFMA/LDS)
//   E4/E6 layernorm_kernel   in-place row LayerNorm (bias + residual already added)  (HBM)
//   E7/E8 pool_normalize_kernel  CLS or masked-mean pooling, v / (|v| + 1e-12)      (HBM)
//
// All arithmetic is fp32: the f32-input MFMA is bit-for-bit an fmaf chain
// (cdna_hip_programming.md §3), which keeps the 1e-4 embedding tolerance with margin.
#include "encoder.hpp"
#include "encoder_rows.hpp"
#include "split_f16.hpp"

#include <cstdlib>

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float kMaskMin = -3.4028234663852886e38f;  // HF get_extended_attention_mask

// ---- E1: embeddings + LayerNorm; E4/E6: LayerNorm -------------------------------------------
// One wave per token row; NPL = H / 64 values per lane, held as pairs of consecutive columns
// (v[2p], v[2p+1] = columns 2*lane + 128*p + {0,1}) so the row can also be written in split-f16
// form (split_f16.hpp) with one 4-byte store per pair and plane.  The arithmetic is ln_row_core (encoder_rows.hpp).
template <int NPL>
__device__ __forceinline__ bool ln_row(float (&v)[NPL], const float* __restrict__ g,
                                       const float* __restrict__ b, float eps, int lane,
                                       float* __restrict__ out, _Float16* __restrict__ outs, float& lo, float& hi) {
    float ov[NPL];
    ln_row_core<NPL>(v, g, b, eps, lane, ov);
    bool ovf = false;
#pragma unroll
    for (int p = 0; p < NPL / 2; ++p) {
        const int c = ln_col(lane, 2 * p);
        float2 o;
        o.x = ov[2 * p];
        o.y = ov[2 * p + 1];
        *reinterpret_cast<float2*>(out + c) = o;
        lo = fminf(lo, fminf(o.x, o.y));  // (the row's range: see ln_range_out)
        hi = fmaxf(hi, fmaxf(o.x, o.y));
        if (outs) {
            f16x2 hi, lo;
            _Float16 a, bb;
            ovf |= sh_split(o.x, a, bb); hi[0] = a; lo[0] = bb;
            ovf |= sh_split(o.y, a, bb); hi[1] = a; lo[1] = bb;
            _Float16* d = outs + (c >> 5) * 64 + (c & 31);
            *reinterpret_cast<f16x2*>(d) = hi;
            *reinterpret_cast<f16x2*>(d + 32) = lo;
        }
    }
    return ovf;
}

// Dynamic-quantised models quantise the LayerNorm output next (gemm_q8.hip): the block's (lo, hi) over its four rows,
// zero included, goes to range_out[2 blockIdx.x ..] so that the range pass over the tensor (100 MB read at 65,536 rows)
// is a reduction over one pair per block instead.  All 256 threads call.  rows != 0 (several quantisation units in the
// batch, whose borders fall on any row): one pair per token ROW, range_out[2 t ..].
__device__ __forceinline__ void ln_range_out(float lo, float hi, float* __restrict__ range_out, uint32_t rows, uint32_t t, uint32_t T) {
    __shared__ float s_lo[4], s_hi[4];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, m, 64));
        hi = fmaxf(hi, __shfl_xor(hi, m, 64));
    }
    if (rows) {
        if ((threadIdx.x & 63) == 0 && t < T) {
            range_out[2 * (size_t)t] = lo;
            range_out[2 * (size_t)t + 1] = hi;
        }
        return;
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        range_out[2 * (size_t)blockIdx.x] = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
        range_out[2 * (size_t)blockIdx.x + 1] = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    }
}

template <int NPL>
__global__ void __launch_bounds__(256)
embed_ln_kernel(const int32_t* __restrict__ ids, const float* __restrict__ word,
                const float* __restrict__ pos, const float* __restrict__ type0,
                const float* __restrict__ g, const float* __restrict__ b, float eps, uint32_t T,
                uint32_t L, uint32_t vocab, float* __restrict__ x, _Float16* __restrict__ xs,
                uint32_t* __restrict__ flag, float* __restrict__ range_out, uint32_t range_rows) {
    constexpr int H = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    float lo = 0.0f, hi = 0.0f;
    if (t < T) {
    uint32_t id = (uint32_t)ids[t];
    if (id >= vocab) id = 0;  // host validates; never index out of the table
    const float* we = word + (size_t)id * H;
    const float* pe = pos ? pos + (size_t)(t % L) * H : nullptr;  // null: no position table (CS_ARCH_NOMIC)
    float v[NPL];
#pragma unroll
    for (int p = 0; p < NPL / 2; ++p) {
        const int c = ln_col(lane, 2 * p);
        const float2 w2 = *reinterpret_cast<const float2*>(we + c);
        const float2 t2 = *reinterpret_cast<const float2*>(type0 + c);
        const float2 p2 = pe ? *reinterpret_cast<const float2*>(pe + c) : make_float2(0.0f, 0.0f);
        v[2 * p] = (w2.x + t2.x) + p2.x;  // BertEmbeddings: (inputs + token_type) + position
        v[2 * p + 1] = (w2.y + t2.y) + p2.y;
    }
    const bool ovf = ln_row<NPL>(v, g, b, eps, lane, x + (size_t)t * H, xs ? xs + (size_t)t * H * 2 : nullptr, lo, hi);
    if (ovf && flag) atomicOr(flag, 1u);
    }
    if (range_out) ln_range_out(lo, hi, range_out, range_rows, t, T);
}

template <int NPL>
__global__ void __launch_bounds__(256)
layernorm_kernel(float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                 float eps, uint32_t T, _Float16* __restrict__ xs, uint32_t* __restrict__ flag,
                 float* __restrict__ range_out, uint32_t range_rows) {
    constexpr int H = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    float lo = 0.0f, hi = 0.0f;
    if (t < T) {
    float* row = x + (size_t)t * H;
    float v[NPL];
#pragma unroll
    for (int p = 0; p < NPL / 2; ++p) {
        const float2 r2 = *reinterpret_cast<const float2*>(row + ln_col(lane, 2 * p));
        v[2 * p] = r2.x;
        v[2 * p + 1] = r2.y;
    }
    const bool ovf = ln_row<NPL>(v, g, b, eps, lane, row, xs ? xs + (size_t)t * H * 2 : nullptr, lo, hi);
    if (ovf && flag) atomicOr(flag, 1u);
    }
    if (range_out) ln_range_out(lo, hi, range_out, range_rows, t, T);
}

// LayerNorm of src into dst (and its split form): the pre-norm families keep src, the residual stream, as it is
template <int NPL>
__global__ void __launch_bounds__(256)
layernorm_to_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ g, const float* __restrict__ b,
                    float eps, uint32_t T, _Float16* __restrict__ xs, uint32_t* __restrict__ flag) {
    constexpr int H = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float* row = src + (size_t)t * H;
    float v[NPL];
#pragma unroll
    for (int p = 0; p < NPL / 2; ++p) {
        const float2 r2 = *reinterpret_cast<const float2*>(row + ln_col(lane, 2 * p));
        v[2 * p] = r2.x;
        v[2 * p + 1] = r2.y;
    }
    float lo = 0.0f, hi = 0.0f;
    const bool ovf = ln_row<NPL>(v, g, b, eps, lane, dst + (size_t)t * H, xs ? xs + (size_t)t * H * 2 : nullptr, lo, hi);
    if (ovf && flag) atomicOr(flag, 1u);
}

// LayerNorm over x + bias + sum of the split-K partial slabs (the epilogue of a launch_gemm_split_partial
// GEMM moved here: slab order is fixed, so the sum is deterministic).
template <int NPL>
__global__ void __launch_bounds__(256)
layernorm_sum_kernel(float* __restrict__ x, const float* __restrict__ parts, uint32_t nparts,
                     const float* __restrict__ bias, const float* __restrict__ g, const float* __restrict__ b,
                     float eps, uint32_t T, _Float16* __restrict__ xs, uint32_t* __restrict__ flag) {
    constexpr int H = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float* row = x + (size_t)t * H;
    float v[NPL];
#pragma unroll
    for (int p = 0; p < NPL / 2; ++p) {
        const int c = ln_col(lane, 2 * p);
        float2 acc = *reinterpret_cast<const float2*>(parts + (size_t)t * H + c);
        for (uint32_t s = 1; s < nparts; ++s) {
            const float2 q = *reinterpret_cast<const float2*>(parts + ((size_t)s * T + t) * H + c);
            acc.x += q.x;
            acc.y += q.y;
        }
        const float2 bv = *reinterpret_cast<const float2*>(bias + c);
        const float2 r2 = *reinterpret_cast<const float2*>(row + c);
        v[2 * p] = (acc.x + bv.x) + r2.x;      // (A W^T + bias) + residual, as the fused epilogue computes it
        v[2 * p + 1] = (acc.y + bv.y) + r2.y;
    }
    float lo = 0.0f, hi = 0.0f;
    const bool ovf = ln_row<NPL>(v, g, b, eps, lane, row, xs ? xs + (size_t)t * H * 2 : nullptr, lo, hi);
    if (ovf && flag) atomicOr(flag, 1u);
}

// ---- E2/E4/E5/E6: C[M,N] = A[M,K] W[N,K]^T + bias (+ epilogue) --------------------------------
// 128x128 block tile, BK = 32, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of 32x32.
// LDS rows are padded to 36 floats so the ds_read_b128 fragment reads (16-lane groups reading
// 16 distinct rows) are bank-conflict free.  Fragment k order: lane half h holds
// k = 16h .. 16h+15 of the stage for BOTH operands, MFMA step s pairs k = s with k = 16 + s;
// the sum over k is the same set in a fixed order.
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2 };
constexpr int GBM = 128, GBN = 128, GBK = 32, GLS = 36;

__device__ __forceinline__ float gelu_erf(float v) {
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

// Epilogue of both GEMM kernels.  C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Full tiles take a branch-free path: residuals are loaded 16 at a time before use (a
// per-element `if (row < M)` makes hipcc wait on every load separately).
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[2][2], const float* __restrict__ bias,
                                              const float* __restrict__ resid, float* __restrict__ C,
                                              uint32_t M, uint32_t N, uint32_t m0, uint32_t n0, int wr,
                                              int wc, int l31, int h) {
    const bool full = (m0 + GBM) <= M;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t col = n0 + wc * 64 + j * 32 + l31;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t rbase = m0 + wr * 64 + i * 32 + 4 * h;
            if (full) {
                float rs[16];
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rs[r] = resid[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * N + col];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bv;
                    if (EPI == EPI_GELU) v = gelu_erf(v);
                    if (EPI == EPI_RESID) v = v + rs[r];
                    C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * N + col] = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < M) {
                        float v = acc[i][j][r] + bv;
                        if (EPI == EPI_GELU) v = gelu_erf(v);
                        if (EPI == EPI_RESID) v = v + resid[(size_t)row * N + col];
                        C[(size_t)row * N + col] = v;
                    }
                }
            }
        }
    }
}

template <int EPI>
__global__ void __launch_bounds__(256, 2)
gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                const float* __restrict__ bias, const float* __restrict__ resid,
                float* __restrict__ C, uint32_t M, uint32_t N, uint32_t K) {
    __shared__ __attribute__((aligned(16))) float lds[2 * GBM * GLS];
    float* As = lds;
    float* Ws = lds + GBM * GLS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // global -> register staging: 4 float4 of A and 4 of W per thread per stage
    f32x4 ga[4], gw[4];
    auto load_stage = [&](uint32_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 3, c4 = idx & 7;
            // rows past M re-read row M-1: they only feed output rows that are never stored
            const uint32_t m = (m0 + row < M) ? m0 + row : M - 1;
            ga[i] = *reinterpret_cast<const f32x4*>(A + (size_t)m * K + k0 + c4 * 4);
            gw[i] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + row) * K + k0 + c4 * 4);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<f32x4*>(As + row * GLS + c4 * 4) = ga[i];
            *reinterpret_cast<f32x4*>(Ws + row * GLS + c4 * 4) = gw[i];
        }
    };

    load_stage(0);
    store_stage();
    __syncthreads();
    for (uint32_t k0 = 0; k0 < K; k0 += GBK) {
        const bool more = (k0 + GBK) < K;
        if (more) load_stage(k0 + GBK);  // in flight while this stage computes
        f32x4 a[2][4], b[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[t][c] = *reinterpret_cast<const f32x4*>(As + (wr * 64 + t * 32 + l31) * GLS + 16 * h + 4 * c);
                b[t][c] = *reinterpret_cast<const f32x4*>(Ws + (wc * 64 + t * 32 + l31) * GLS + 16 * h + 4 * c);
            }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][c][e], b[j][c][e], acc[i][j], 0, 0, 0);
        __syncthreads();
        if (more) {
            store_stage();
            __syncthreads();
        }
    }

    gemm_epilogue<EPI>(acc, bias, resid, C, M, N, m0, n0, wr, wc, l31, h);
}

// Software-pipelined variant: two LDS stage buffers, ONE barrier per K-step, and the fragments
// of the next 8-k chunk are read from LDS while the current chunk's 16 MFMAs run, so a wave
// has no MFMA-free window between stages (blocks on a CU run in lock-step, so such windows
// coincide and idle the matrix pipe: 29 % on the plain kernel per SQ_VALU_MFMA_BUSY_CYCLES).
// Stage s+1 goes global -> registers during stage s-1.. s, is written to the other buffer in
// the middle of stage s, and the barrier sits before the first read of that buffer.
template <int EPI>
__global__ void __launch_bounds__(256, 2)
gemm_f32_pipe_kernel(const float* __restrict__ A, const float* __restrict__ W,
                     const float* __restrict__ bias, const float* __restrict__ resid,
                     float* __restrict__ C, uint32_t M, uint32_t N, uint32_t K) {
    extern __shared__ __attribute__((aligned(16))) float plds[];  // [2][A 128x36 | W 128x36]
    constexpr int STAGE = 2 * GBM * GLS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ga[4], gw[4];
    const float* asrc[4];
    const float* wsrc[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 3, c4 = idx & 7;
        // rows past M re-read row M-1: they only feed output rows that are never stored
        const uint32_t m = (m0 + row < M) ? m0 + row : M - 1;
        asrc[i] = A + (size_t)m * K + c4 * 4;
        wsrc[i] = W + (size_t)(n0 + row) * K + c4 * 4;
        soff[i] = row * GLS + c4 * 4;
    }
    auto load_stage = [&](uint32_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ga[i] = *reinterpret_cast<const f32x4*>(asrc[i] + k0);
            gw[i] = *reinterpret_cast<const f32x4*>(wsrc[i] + k0);
        }
    };
    auto store_stage = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(buf + soff[i]) = ga[i];
            *reinterpret_cast<f32x4*>(buf + GBM * GLS + soff[i]) = gw[i];
        }
    };
    const int aoff = (wr * 64 + l31) * GLS + 16 * h;
    const int boff = GBM * GLS + (wc * 64 + l31) * GLS + 16 * h;
    auto frags = [&](const float* buf, int c, f32x4 (&a)[2], f32x4 (&b)[2]) {
        a[0] = *reinterpret_cast<const f32x4*>(buf + aoff + 4 * c);
        a[1] = *reinterpret_cast<const f32x4*>(buf + aoff + 32 * GLS + 4 * c);
        b[0] = *reinterpret_cast<const f32x4*>(buf + boff + 4 * c);
        b[1] = *reinterpret_cast<const f32x4*>(buf + boff + 32 * GLS + 4 * c);
    };
    auto mfma_chunk = [&](const f32x4 (&a)[2], const f32x4 (&b)[2]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    };

    const uint32_t nstages = K / GBK;
    load_stage(0);
    store_stage(plds);
    __syncthreads();
    if (nstages > 1) load_stage(GBK);
    f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
    frags(plds, 0, fa0, fb0);
    for (uint32_t s = 0; s < nstages; ++s) {
        float* cur = plds + (s & 1) * STAGE;
        float* nxt = plds + ((s + 1) & 1) * STAGE;
        const bool more = s + 1 < nstages;
        frags(cur, 1, fa1, fb1);
        mfma_chunk(fa0, fb0);
        frags(cur, 2, fa0, fb0);
        mfma_chunk(fa1, fb1);
        if (more) {
            store_stage(nxt);                              // stage s+1 -> the other buffer
            if (s + 2 < nstages) load_stage((s + 2) * GBK);  // stage s+2 -> registers
        }
        frags(cur, 3, fa1, fb1);
        mfma_chunk(fa0, fb0);
        __syncthreads();                                   // nxt complete; cur fully consumed
        if (more) frags(nxt, 0, fa0, fb0);
        mfma_chunk(fa1, fb1);
    }

    gemm_epilogue<EPI>(acc, bias, resid, C, M, N, m0, n0, wr, wc, l31, h);
}

// ---- E3: attention, head_dim 32 ---------------------------------------------------------------
// Block = (query block of 128, head, batch); wave w owns queries qb*128 + w*32 + (lane&31).
// S^T tile (keys x queries) = K_tile (A: [key][d]) x Q^T (B: [d][query]); then
// O^T (d x queries) += V_tile^T (A: [d][key]) x P^T (B: [key][query]) where P^T is the S^T
// accumulator itself: its row (key) index sits on (register, half) exactly as the A/B k index
// of the next MFMA needs, so probabilities never leave registers.
constexpr int AKS = 36;  // padded K row (floats)

template <bool SPLIT>
__global__ void __launch_bounds__(256)
attention_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ mask,
                 float* __restrict__ ctx, _Float16* __restrict__ ctxs, uint32_t* __restrict__ flag,
                 uint32_t L, uint32_t H, float scale, const float* __restrict__ alibi, uint32_t window) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t Lp = (L + 31) & ~31u;
    float* Ks = smem;                 // [Lp][36]
    float* Vs = Ks + (size_t)Lp * AKS;  // [Lp][32]
    float* madd = Vs + (size_t)Lp * 32; // [Lp]
    int& last_valid = *reinterpret_cast<int*>(madd + Lp);  // all LDS in the one dynamic region
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t qb = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const size_t row3 = (size_t)3 * H;
    const float* base = qkv + (size_t)b * L * row3 + head * 32;

    if (tid == 0) last_valid = 0;
    __syncthreads();
    for (uint32_t idx = tid; idx < Lp * 8; idx += 256) {
        const uint32_t key = idx >> 3, c4 = idx & 7;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (key < L) {
            kv = *reinterpret_cast<const f32x4*>(base + key * row3 + H + c4 * 4);
            vv = *reinterpret_cast<const f32x4*>(base + key * row3 + 2 * H + c4 * 4);
        }
        *reinterpret_cast<f32x4*>(Ks + key * AKS + c4 * 4) = kv;
        *reinterpret_cast<f32x4*>(Vs + key * 32 + c4 * 4) = vv;
    }
    for (uint32_t key = tid; key < Lp; key += 256) {
        const bool ok = key < L && mask[(size_t)b * L + key] != 0;
        madd[key] = ok ? 0.0f : kMaskMin;
        if (ok) atomicMax(&last_valid, (int)key);
    }
    __syncthreads();
    const uint32_t ntiles = (uint32_t)last_valid / 32 + 1;  // trailing all-masked tiles add exp(min - m) = 0

    const uint32_t query = qb * 128 + wave * 32 + l31;
    float qf[16];
    {
        const bool ok = query < L;
        const float* qp = base + (size_t)(ok ? query : 0) * row3 + 16 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 4 * c);
            qf[4 * c + 0] = ok ? t.x : 0.f; qf[4 * c + 1] = ok ? t.y : 0.f;
            qf[4 * c + 2] = ok ? t.z : 0.f; qf[4 * c + 3] = ok ? t.w : 0.f;
        }
    }
    f32x16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.0f;
    float m = -__builtin_huge_valf(), lsum = 0.0f;

    for (uint32_t kt = 0; kt < ntiles; ++kt) {
        const float* kr = Ks + (size_t)(kt * 32 + l31) * AKS + 16 * h;
        f32x4 ka[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) ka[c] = *reinterpret_cast<const f32x4*>(kr + 4 * c);
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[c][e], qf[4 * c + e], st, 0, 0, 0);
        // st[r] = S^T[key = kt*32 + (r&3) + 8*(r>>2) + 4h][query = lane&31]
        float tmax = -__builtin_huge_valf();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            st[r] = st[r] * scale + madd[key];
            if (alibi) st[r] += alibi[head] * -fabsf((float)query - (float)key);  // JinaBert: -slope_h |i - j|
            if (window && (query > key ? query - key : key - query) > window) st[r] = kMaskMin;  // ModernBERT: a local layer
            tmax = fmaxf(tmax, st[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(m, tmax);
        const float alpha = expf(m - mnew);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = expf(st[r] - mnew);
            psum += st[r];
        }
        lsum = lsum * alpha + psum;
        m = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float vt = Vs[(size_t)key * 32 + l31];  // A = V^T[d = lane&31][key]
            ot = __builtin_amdgcn_mfma_f32_32x32x2f32(vt, st[r], ot, 0, 0, 0);
        }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / lsum;
    // ot[r] = O^T[d = (r&3) + 8*(r>>2) + 4h][query]; 4 consecutive d per register quad
    if (query < L) {
        if (SPLIT) {
            // head_dim 32 = one k-chunk of the output projection: [row][head][32 hi | 32 lo]
            _Float16* op = ctxs + (((size_t)b * L + query) * (H / 32) + head) * 64 + 4 * h;
            bool ovf = false;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a, bb;
                    ovf |= sh_split(ot[4 * g + e] * inv, a, bb);
                    hi[e] = a; lo[e] = bb;
                }
                *reinterpret_cast<f16x4*>(op + 8 * g) = hi;
                *reinterpret_cast<f16x4*>(op + 32 + 8 * g) = lo;
            }
            if (ovf && flag) atomicOr(flag, 1u);
        } else {
            float* op = ctx + ((size_t)b * L + query) * H + head * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o = {ot[4 * g] * inv, ot[4 * g + 1] * inv, ot[4 * g + 2] * inv, ot[4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(op + 8 * g) = o;
            }
        }
    }
}

// ---- E3 for head_dim 64 (exact f32): the kernel above with keys staged 128 at a time ---------------------------
// K rows of 64 + 4 floats and V rows of 64 floats for a whole 512-token sequence would need 270 KB of LDS; a
// super-tile of 128 keys needs 67.6 KB.  The online softmax state and the two 32-row tiles of O^T carry over.
constexpr int AKS64 = 68;  // padded K row (floats)
constexpr int AKT64 = 128; // keys per super-tile

__global__ void __launch_bounds__(256)
attention64_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ mask, float* __restrict__ ctx,
                   uint32_t L, uint32_t H, float scale, const float* __restrict__ alibi, uint32_t window) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t Lp = (L + 31) & ~31u;
    float* Ks = smem;                          // [128][68]
    float* Vs = Ks + (size_t)AKT64 * AKS64;    // [128][64]
    float* madd = Vs + (size_t)AKT64 * 64;     // [Lp]
    int& last_valid = *reinterpret_cast<int*>(madd + Lp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t qb = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const size_t row3 = (size_t)3 * H;
    const float* base = qkv + (size_t)b * L * row3 + head * 64;

    if (tid == 0) last_valid = 0;
    __syncthreads();
    for (uint32_t key = tid; key < Lp; key += 256) {
        const bool ok = key < L && mask[(size_t)b * L + key] != 0;
        madd[key] = ok ? 0.0f : kMaskMin;
        if (ok) atomicMax(&last_valid, (int)key);
    }
    __syncthreads();
    const uint32_t ntiles = (uint32_t)last_valid / 32 + 1;

    const uint32_t query = qb * 128 + wave * 32 + l31;
    float qf[32];  // d = 32 h + j  (the same permutation of d as the K fragments below)
    {
        const bool ok = query < L;
        const float* qp = base + (size_t)(ok ? query : 0) * row3 + 32 * h;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 4 * c);
            qf[4 * c + 0] = ok ? t.x : 0.f; qf[4 * c + 1] = ok ? t.y : 0.f;
            qf[4 * c + 2] = ok ? t.z : 0.f; qf[4 * c + 3] = ok ? t.w : 0.f;
        }
    }
    f32x16 ot[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[c][r] = 0.0f;
    float m = -__builtin_huge_valf(), lsum = 0.0f;

    for (uint32_t st = 0; st * 4 < ntiles; ++st) {
        __syncthreads();  // the previous super-tile is consumed
        for (uint32_t idx = tid; idx < (uint32_t)AKT64 * 16; idx += 256) {
            const uint32_t row = idx >> 4, c4 = idx & 15, key = st * AKT64 + row;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < L) {
                kv = *reinterpret_cast<const f32x4*>(base + key * row3 + H + c4 * 4);
                vv = *reinterpret_cast<const f32x4*>(base + key * row3 + 2 * H + c4 * 4);
            }
            *reinterpret_cast<f32x4*>(Ks + row * AKS64 + c4 * 4) = kv;
            *reinterpret_cast<f32x4*>(Vs + row * 64 + c4 * 4) = vv;
        }
        __syncthreads();
        const uint32_t kt_end = ntiles - st * 4 < 4 ? ntiles - st * 4 : 4;
        for (uint32_t kl = 0; kl < kt_end; ++kl) {
            const uint32_t kt = st * 4 + kl;
            const float* kr = Ks + (size_t)(kl * 32 + l31) * AKS64 + 32 * h;
            f32x16 stt;
#pragma unroll
            for (int r = 0; r < 16; ++r) stt[r] = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 ka = *reinterpret_cast<const f32x4*>(kr + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    stt = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[4 * c + e], stt, 0, 0, 0);
            }
            float tmax = -__builtin_huge_valf();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                stt[r] = stt[r] * scale + madd[key];
                if (alibi) stt[r] += alibi[head] * -fabsf((float)query - (float)key);  // JinaBert: -slope_h |i - j|
                if (window && (query > key ? query - key : key - query) > window) stt[r] = kMaskMin;  // ModernBERT: a local layer
                tmax = fmaxf(tmax, stt[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(m, tmax);
            const float alpha = expf(m - mnew);
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                stt[r] = expf(stt[r] - mnew);
                psum += stt[r];
            }
            lsum = lsum * alpha + psum;
            m = mnew;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[c][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t krow = kl * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;  // key inside the super-tile
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float vt = Vs[(size_t)krow * 64 + 32 * c + l31];  // A = V^T[d = 32 c + (lane & 31)][key]
                    ot[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vt, stt[r], ot[c], 0, 0, 0);
                }
            }
        }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / lsum;
    // ot[c][r] = O^T[d = 32 c + (r&3) + 8*(r>>2) + 4h][query]
    if (query < L) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float* op = ctx + ((size_t)b * L + query) * H + head * 64 + 32 * c + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o = {ot[c][4 * g] * inv, ot[c][4 * g + 1] * inv, ot[c][4 * g + 2] * inv, ot[c][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(op + 8 * g) = o;
            }
        }
    }
}

// ---- E7/E8: pooling + L2 normalise ------------------------------------------------------------
// One wave per sequence.  CLS: row 0.  Mean: sum(mask*h) / max(sum mask, 1e-9).
template <int NPL>
__global__ void __launch_bounds__(256)
pool_normalize_kernel(const float* __restrict__ x, const int32_t* __restrict__ mask, uint32_t B,
                      uint32_t L, int pooling, float* __restrict__ out) {
    constexpr int H = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float v[NPL];
    if (pooling == CS_POOL_CLS) {
        const float* r = x + (size_t)b * L * H;
#pragma unroll
        for (int i = 0; i < NPL; ++i) v[i] = r[lane + 64 * i];
    } else {
#pragma unroll
        for (int i = 0; i < NPL; ++i) v[i] = 0.0f;
        float cnt = 0.0f;
        for (uint32_t t = 0; t < L; ++t) {
            if (!mask[(size_t)b * L + t]) continue;  // wave-uniform
            cnt += 1.0f;
            const float* r = x + ((size_t)b * L + t) * H;
#pragma unroll
            for (int i = 0; i < NPL; ++i) v[i] += r[lane + 64 * i];
        }
        cnt = fmaxf(cnt, 1e-9f);
#pragma unroll
        for (int i = 0; i < NPL; ++i) v[i] /= cnt;
    }
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) ss = fmaf(v[i], v[i], ss);
    const float den = sqrtf(wave_sum(ss)) + 1e-12f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) out[(size_t)b * H + lane + 64 * i] = v[i] / den;
}

// Mean pooling with a block per sequence (the kernel above walks a sequence with ONE wave and a mask load in front of
// every row: 145 us for 128 x 256 tokens, all of it latency).  Wave w takes tokens 64 w .. 64 w + 63 of every 256: one
// load of their 64 mask words, a ballot, then the rows that count in batches of four independent loads; the four waves'
// partial sums are added in wave order (a fixed order: the result does not depend on timing).
template <int NPL>
__global__ void __launch_bounds__(256)
mean_pool_normalize_kernel(const float* __restrict__ x, const int32_t* __restrict__ mask, uint32_t L,
                           float* __restrict__ out) {
    constexpr int H = 64 * NPL;
    __shared__ float part[4][H];
    __shared__ float pcnt[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t b = blockIdx.x;
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = 0.0f;
    float cnt = 0.0f;
    for (uint32_t t0 = wave * 64; t0 < L; t0 += 256) {
        const uint32_t t = t0 + lane;
        unsigned long long live = __ballot(t < L && mask[(size_t)b * L + t] != 0);
        cnt += (float)__popcll(live);
        while (live) {
            uint32_t tok[4];
            int n = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                tok[u] = 0;
                if (live) { tok[u] = t0 + (uint32_t)__builtin_ctzll(live); live &= live - 1; n = u + 1; }
            }
            float r[4][NPL];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < n) {
                    const float* p = x + ((size_t)b * L + tok[u]) * H;
#pragma unroll
                    for (int i = 0; i < NPL; ++i) r[u][i] = p[lane + 64 * i];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < n) {
#pragma unroll
                    for (int i = 0; i < NPL; ++i) v[i] += r[u][i];
                }
        }
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) part[wave][lane + 64 * i] = v[i];
    if (lane == 0) pcnt[wave] = cnt;
    __syncthreads();
    if (wave != 0) return;
    cnt = fmaxf(pcnt[0] + pcnt[1] + pcnt[2] + pcnt[3], 1e-9f);
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = (((part[0][c] + part[1][c]) + part[2][c]) + part[3][c]) / cnt;
        ss = fmaf(v[i], v[i], ss);
    }
    const float den = sqrtf(wave_sum(ss)) + 1e-12f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) out[(size_t)b * H + lane + 64 * i] = v[i] / den;
}

// ---- synthetic parameters, generated in HBM ---------------------------------------------------
__global__ void synth_params_kernel(float* __restrict__ out, cs_bert_config cfg, cs_bert_offsets off,
                                    uint64_t seed) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < off.total; e += stride)
        out[e] = cs_bert_synth_param(&cfg, &off, seed, e);
}

// ---- launchers --------------------------------------------------------------------------------

template <int NPL>
static void launch_rows(int which, const EncoderLaunch& a, hipStream_t s) {
    const uint32_t T = a.T;
    if (which == 0)
        hipLaunchKernelGGL(embed_ln_kernel<NPL>, dim3((T + 3) / 4), dim3(256), 0, s, a.ids, a.word, a.pos,
                           a.type0, a.g, a.b, a.eps, T, a.L, a.vocab, a.x, static_cast<_Float16*>(a.xs), a.flag, a.range_out, a.range_rows ? 1u : 0u);
    else if (which == 1)
        hipLaunchKernelGGL(layernorm_kernel<NPL>, dim3((T + 3) / 4), dim3(256), 0, s, a.x, a.g, a.b, a.eps, T,
                           static_cast<_Float16*>(a.xs), a.flag, a.range_out, a.range_rows ? 1u : 0u);
    else if (which == 4)
        hipLaunchKernelGGL(layernorm_to_kernel<NPL>, dim3((T + 3) / 4), dim3(256), 0, s, a.src, a.x, a.g, a.b, a.eps, T,
                           static_cast<_Float16*>(a.xs), a.flag);
    else if (which == 3)
        hipLaunchKernelGGL(layernorm_sum_kernel<NPL>, dim3((T + 3) / 4), dim3(256), 0, s, a.x, a.parts, a.nparts,
                           a.bias, a.g, a.b, a.eps, T, static_cast<_Float16*>(a.xs), a.flag);
    else if (a.pooling != CS_POOL_CLS && a.L >= 16)
        hipLaunchKernelGGL(mean_pool_normalize_kernel<NPL>, dim3(a.B), dim3(256), 0, s, a.x, a.mask, a.L, a.out);
    else
        hipLaunchKernelGGL(pool_normalize_kernel<NPL>, dim3((a.B + 3) / 4), dim3(256), 0, s, a.x, a.mask, a.B,
                           a.L, a.pooling, a.out);
}

int32_t launch_row_kernel(int which, const EncoderLaunch& a, uint32_t H, hipStream_t s) {
    switch (H) {
        case 384: launch_rows<6>(which, a, s); break;
        case 768: launch_rows<12>(which, a, s); break;
        case 1024: launch_rows<16>(which, a, s); break;
        default: return fail(CS_ERR_UNSUPPORTED, "hidden size %u not supported (384/768/1024)", H);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

static int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = cs_lab_env("CS_GEMM_VARIANT");  // 0 = plain, 1 = pipelined (default)
        v = e ? std::atoi(e) : 1;
    }
    return v;
}

int32_t launch_gemm(int epi, const float* A, const float* W, const float* bias, const float* resid,
                    float* C, uint32_t M, uint32_t N, uint32_t K, hipStream_t s) {
    if (N % GBN || K % GBK) return fail(CS_ERR_UNSUPPORTED, "GEMM N=%u K=%u must be multiples of 128/32", N, K);
    dim3 grid(N / GBN, (M + GBM - 1) / GBM);
    if (gemm_variant() == 0) {
        if (epi == EPI_BIAS) hipLaunchKernelGGL(gemm_f32_kernel<EPI_BIAS>, grid, dim3(256), 0, s, A, W, bias, resid, C, M, N, K);
        else if (epi == EPI_GELU) hipLaunchKernelGGL(gemm_f32_kernel<EPI_GELU>, grid, dim3(256), 0, s, A, W, bias, resid, C, M, N, K);
        else hipLaunchKernelGGL(gemm_f32_kernel<EPI_RESID>, grid, dim3(256), 0, s, A, W, bias, resid, C, M, N, K);
    } else {
        constexpr size_t lds = 2 * 2 * GBM * GLS * sizeof(float);  // 73,728 B
        static PerDeviceOnce attr_set;  // function attributes are per device
        CS_TRY(attr_set.run([&]() -> int32_t {
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_pipe_kernel<EPI_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_pipe_kernel<EPI_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_pipe_kernel<EPI_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            return CS_OK;
        }));
        if (epi == EPI_BIAS) hipLaunchKernelGGL(gemm_f32_pipe_kernel<EPI_BIAS>, grid, dim3(256), lds, s, A, W, bias, resid, C, M, N, K);
        else if (epi == EPI_GELU) hipLaunchKernelGGL(gemm_f32_pipe_kernel<EPI_GELU>, grid, dim3(256), lds, s, A, W, bias, resid, C, M, N, K);
        else hipLaunchKernelGGL(gemm_f32_pipe_kernel<EPI_RESID>, grid, dim3(256), lds, s, A, W, bias, resid, C, M, N, K);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

size_t attention_lds_bytes(uint32_t L) {
    const uint32_t Lp = (L + 31) & ~31u;
    return (size_t)Lp * (AKS + 32 + 1) * sizeof(float) + 16;
}

static int32_t launch_attention_impl(const float* qkv, const int32_t* mask, float* ctx, _Float16* ctxs,
                                     uint32_t* flag, uint32_t B, uint32_t L, uint32_t H, uint32_t heads,
                                     hipStream_t s, const float* alibi, uint32_t window) {
    if (heads && H % heads == 0 && H / heads == 64 && !ctxs) {  // exact-f32 mode, 64-wide heads
        const size_t Lp = (L + 31) & ~31u;
        const size_t lds64 = ((size_t)AKT64 * (AKS64 + 64) + Lp + 4) * sizeof(float);
        static PerDeviceOnce attr64;  // function attributes are per device
        CS_TRY(attr64.run([&]() -> int32_t {
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention64_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            return CS_OK;
        }));
        hipLaunchKernelGGL(attention64_kernel, dim3((L + 127) / 128, heads, B), dim3(256), lds64, s, qkv, mask, ctx, L, H,
                           1.0f / sqrtf(64.0f), alibi, window);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    if (H / heads != 32 || H % heads)
        return fail(CS_ERR_UNSUPPORTED, "head_dim %u not supported (32 or 64)", heads ? H / heads : 0);
    const size_t lds = attention_lds_bytes(L);
    if (lds > 160 * 1024 - 64) return fail(CS_ERR_UNSUPPORTED, "sequence length %u exceeds the LDS-resident K/V limit", L);
    static PerDeviceOnce attr_set;  // function attributes are per device
    CS_TRY(attr_set.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        return CS_OK;
    }));
    dim3 grid((L + 127) / 128, heads, B);
    const float scale = 1.0f / sqrtf(32.0f);
    if (ctxs) return fail(CS_ERR_UNSUPPORTED, "the f32 attention kernel writes f32 context rows only");
    hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), lds, s, qkv, mask, ctx, ctxs, flag, L, H, scale, alibi, window);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_attention(const float* qkv, const int32_t* mask, float* ctx, uint32_t B, uint32_t L,
                         uint32_t H, uint32_t heads, hipStream_t s, const float* alibi, uint32_t window) {
    return launch_attention_impl(qkv, mask, ctx, nullptr, nullptr, B, L, H, heads, s, alibi, window);
}

int32_t launch_synth_params(float* d_out, const cs_bert_config& cfg, uint64_t seed, hipStream_t s) {
    cs_bert_offsets off;
    cs_bert_layout(&cfg, &off);
    hipLaunchKernelGGL(synth_params_kernel, dim3(2048), dim3(256), 0, s, d_out, cfg, off, seed);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
