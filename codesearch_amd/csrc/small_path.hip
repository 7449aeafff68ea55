// small_path.hip — the dense layers of a forward over A FEW SHORT SEQUENCES (under 200 token rows: the query side of
// `codesearch search`, EmbeddingService::embed_query / embed_queries_batch, /root/reference/src/embed/mod.rs:164-226),
// kernel by kernel but with fewer, shorter kernels than the general small-batch path of gemm_split.hip:
//   * LayerNorm is the PROLOGUE of the dense layer that reads it (sp_ln_gemm_kernel): every block normalises the 16 rows
//     its tile needs itself — 4 rows per wave, their reduction chains interleaved (ln_rows_core) — and keeps them in LDS
//     in split form; the block of column tile 0 also writes the rows (the residual stream).  Two launches per layer and
//     the embedding kernel disappear (86 -> 62 launches per 12-layer forward).
//   * FFN-down is cut into four K slices (sp_partial_kernel): 96 blocks of 24 KB of weights + 24 KB of activations each
//     instead of 24 blocks pulling 196 KB through one CU's load path (10 us per layer in the round-5 trace); the four
//     partial slabs are summed in slab order, with bias and residual, by the LayerNorm prologue that follows
//     (the arithmetic of layernorm_sum_kernel, encoder.hip) — deterministic, and the order the one-launch form
//     (small_forward.hip) uses too, so the two stay bit-identical.
// A tile is gemm_sh_skinny_kernel<EPI, 1, 1>'s: K split over the block's four waves (chunks w, w + 4, ...), three MFMAs per
// chunk, partial tiles summed (w0 + w1) + (w2 + w3).
#include "small_path.hpp"

#include "encoder_rows.hpp"
#include "gemm_epilogue.hpp"
#include "split_f16.hpp"

namespace cs {

namespace {

template <int NPL>
struct SpGeom {
    static constexpr int H = 64 * NPL;
    static constexpr int AROW = H * 4 + 16;  // a split row (H / 32 lines of 128 B) + 16 B: conflict-free ds_read_b128 over 16 rows
    static constexpr int U = NPL / 2;         // k-chunks per wave of a K = H product (and of a quarter of K = 4 H)
};

}  // namespace

// PRO: 0 LayerNorm of Y's rows | 1 LayerNorm of (slab sum + bias) + residual | 2 embedding gather + LayerNorm
template <int NPL, int EPI, int PRO>
__global__ void __launch_bounds__(256)
sp_ln_gemm_kernel(SpLnGemmArgs a) {
    using G = SpGeom<NPL>;
    constexpr int H = G::H, U = G::U;
    __shared__ __attribute__((aligned(16))) char aimg[16 * G::AROW];
    __shared__ float red[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t n0 = blockIdx.x * 16, m0 = blockIdx.y * 16, T = a.T;
    const bool leader = blockIdx.x == 0;
    bool ovf = false;

    // this wave's W fragments leave first: they depend on nothing
    constexpr uint32_t kchunks = H / 32;
    const _Float16* wp = a.W + (size_t)(n0 + l15) * kchunks * 64 + 8 * g;
    f16x8 wh[U], wl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        wh[u] = *reinterpret_cast<const f16x8*>(wp + (size_t)(wave + 4 * u) * 64);
        wl[u] = *reinterpret_cast<const f16x8*>(wp + (size_t)(wave + 4 * u) * 64 + 32);
    }
    const int em = tid >> 4, en = tid & 15;
    const float bias_v = a.bias[n0 + en];

    // ---- prologue: the tile's 16 rows -> LayerNorm -> split form in LDS (4 rows per wave) ----
    float v[4][NPL];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const uint32_t t_raw = m0 + 4 * wave + rr, t = t_raw < T ? t_raw : T - 1;
        if constexpr (PRO == 2) {
            uint32_t id = (uint32_t)a.ids[t];
            if (id >= a.vocab) id = 0;
            const float* we = a.word + (size_t)id * H;
            const float* pe = a.pos + (size_t)(t % a.L) * H;
#pragma unroll
            for (int p = 0; p < NPL / 2; ++p) {
                const int c = ln_col(lane, 2 * p);
                const float2 w2 = *reinterpret_cast<const float2*>(we + c);
                const float2 t2 = *reinterpret_cast<const float2*>(a.type0 + c);
                const float2 p2 = *reinterpret_cast<const float2*>(pe + c);
                v[rr][2 * p] = (w2.x + t2.x) + p2.x;
                v[rr][2 * p + 1] = (w2.y + t2.y) + p2.y;
            }
        } else if constexpr (PRO == 1) {
#pragma unroll
            for (int p = 0; p < NPL / 2; ++p) {
                const int c = ln_col(lane, 2 * p);
                float2 acc = *reinterpret_cast<const float2*>(a.parts + (size_t)t * H + c);
#pragma unroll
                for (uint32_t s = 1; s < 4; ++s) {
                    const float2 q = *reinterpret_cast<const float2*>(a.parts + ((size_t)s * T + t) * H + c);
                    acc.x += q.x;
                    acc.y += q.y;
                }
                const float2 bv = *reinterpret_cast<const float2*>(a.parts_bias + c);
                const float2 r2 = *reinterpret_cast<const float2*>(a.X + (size_t)t * H + c);
                v[rr][2 * p] = (acc.x + bv.x) + r2.x;      // layernorm_sum_kernel's order (encoder.hip)
                v[rr][2 * p + 1] = (acc.y + bv.y) + r2.y;
            }
        } else {
#pragma unroll
            for (int p = 0; p < NPL / 2; ++p) {
                const float2 r2 = *reinterpret_cast<const float2*>(a.Y + (size_t)t * H + ln_col(lane, 2 * p));
                v[rr][2 * p] = r2.x;
                v[rr][2 * p + 1] = r2.y;
            }
        }
    }
    float ov[4][NPL];
    ln_rows_core<NPL, 4>(v, a.ln_g, a.ln_b, a.eps, lane, ov);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = 4 * wave + rr;
        const uint32_t t = m0 + row;
        char* irow = aimg + row * G::AROW;
#pragma unroll
        for (int p = 0; p < NPL / 2; ++p) {
            const int c = ln_col(lane, 2 * p);
            f16x2 hi, lo;
            _Float16 x0, x1;
            ovf |= sh_split(ov[rr][2 * p], x0, x1); hi[0] = x0; lo[0] = x1;
            ovf |= sh_split(ov[rr][2 * p + 1], x0, x1); hi[1] = x0; lo[1] = x1;
            *reinterpret_cast<f16x2*>(irow + (c >> 5) * 128 + (c & 31) * 2) = hi;
            *reinterpret_cast<f16x2*>(irow + (c >> 5) * 128 + 64 + (c & 31) * 2) = lo;
            // (PRO 1 reads X's row as the residual above: the leader overwrites it only after its own read — other blocks
            // read the same row concurrently, so the new row goes to Xout, a different buffer)
            if (leader && t < T) *reinterpret_cast<float2*>(a.Xout + (size_t)t * H + c) = make_float2(ov[rr][2 * p], ov[rr][2 * p + 1]);
        }
    }
    __syncthreads();

    // ---- the tile ----
    sh_f32x4v hh = {0.f, 0.f, 0.f, 0.f}, xx = {0.f, 0.f, 0.f, 0.f};
    const char* irow = aimg + l15 * G::AROW + g * 16;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(irow + (wave + 4 * u) * 128);
        const f16x8 al = *reinterpret_cast<const f16x8*>(irow + (wave + 4 * u) * 128 + 64);
        hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[u], hh, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[u], xx, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[u], xx, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * g + r][l15] = fmaf(xx[r], kShLoInv, hh[r]);
    __syncthreads();
    const uint32_t row = m0 + em, col = n0 + en;
    if (row < T) {
        float o = (red[0][em][en] + red[1][em][en]) + (red[2][em][en] + red[3][em][en]) + bias_v;
        if (EPI == SH_OUT_SPLIT_GELU) o = sh_gelu_erf(o);
        _Float16 hi, lo;
        ovf |= sh_split(o, hi, lo);
        _Float16* dst = a.Cs + ((size_t)row * (a.N / 32) + (col >> 5)) * 64 + (col & 31);
        dst[0] = hi;
        dst[32] = lo;
    }
    if (ovf && a.flag) atomicOr(a.flag, 1u);
}

// One K slice (a quarter of K = 4 H) of a 16 x 16 tile of C = A W^T: raw sums into slab blockIdx.z of parts[4][T][N].
template <int NPL>
__global__ void __launch_bounds__(256)
sp_partial_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ parts, uint32_t T, uint32_t N) {
    constexpr int U = SpGeom<NPL>::U;
    constexpr uint32_t kchunks = 8 * NPL, kq = kchunks / 4;  // K = 4 H = 256 NPL
    __shared__ float red[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t n0 = blockIdx.x * 16, m0 = blockIdx.y * 16, ks = blockIdx.z;
    const uint32_t r = m0 + l15;
    const _Float16* ap = A + (size_t)(r < T ? r : T - 1) * kchunks * 64 + 8 * g;
    const _Float16* wp = W + (size_t)(n0 + l15) * kchunks * 64 + 8 * g;
    f16x8 ah[U], al[U], wh[U], wl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t off = (size_t)(ks * kq + wave + 4 * u) * 64;
        wh[u] = *reinterpret_cast<const f16x8*>(wp + off);
        wl[u] = *reinterpret_cast<const f16x8*>(wp + off + 32);
        ah[u] = *reinterpret_cast<const f16x8*>(ap + off);
        al[u] = *reinterpret_cast<const f16x8*>(ap + off + 32);
    }
    sh_f32x4v hh = {0.f, 0.f, 0.f, 0.f}, xx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], wh[u], hh, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], wl[u], xx, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], wh[u], xx, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][4 * g + q][l15] = fmaf(xx[q], kShLoInv, hh[q]);
    __syncthreads();
    const int em = tid >> 4, en = tid & 15;
    const uint32_t row = m0 + em;
    if (row < T)
        parts[((size_t)ks * T + row) * N + n0 + en] = (red[0][em][en] + red[1][em][en]) + (red[2][em][en] + red[3][em][en]);
}

bool small_path_supported(uint32_t H, uint32_t I, uint32_t T) {
    return (H == 384 || H == 768 || H == 1024) && I == 4 * H && T >= 1 && T <= SP_MAX_ROWS;
}

template <int NPL>
static void sp_launch_ln_gemm(int epi, int pro, const SpLnGemmArgs& a, hipStream_t s) {
    const dim3 grid(a.N / 16, (a.T + 15) / 16);
#define SP_L(E, P) hipLaunchKernelGGL((sp_ln_gemm_kernel<NPL, E, P>), grid, dim3(256), 0, s, a)
    if (epi == SH_OUT_SPLIT) { if (pro == 0) SP_L(SH_OUT_SPLIT, 0); else if (pro == 1) SP_L(SH_OUT_SPLIT, 1); else SP_L(SH_OUT_SPLIT, 2); }
    else { if (pro == 0) SP_L(SH_OUT_SPLIT_GELU, 0); else if (pro == 1) SP_L(SH_OUT_SPLIT_GELU, 1); else SP_L(SH_OUT_SPLIT_GELU, 2); }
#undef SP_L
}

int32_t launch_sp_ln_gemm(int epi, int pro, const SpLnGemmArgs& a, uint32_t H, hipStream_t s) {
    if ((epi != SH_OUT_SPLIT && epi != SH_OUT_SPLIT_GELU) || pro < 0 || pro > 2 || a.N % 32 || a.T == 0)
        return fail(CS_ERR_BAD_ARG, "bad small-path dense layer (epilogue %d, prologue %d, N %u)", epi, pro, a.N);
    switch (H) {
        case 384: sp_launch_ln_gemm<6>(epi, pro, a, s); break;
        case 768: sp_launch_ln_gemm<12>(epi, pro, a, s); break;
        case 1024: sp_launch_ln_gemm<16>(epi, pro, a, s); break;
        default: return fail(CS_ERR_UNSUPPORTED, "hidden size %u not supported (384/768/1024)", H);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_sp_partial(const _Float16* A, const _Float16* W, float* parts, uint32_t T, uint32_t N, uint32_t H, hipStream_t s) {
    const dim3 grid(N / 16, (T + 15) / 16, 4);
    switch (H) {
        case 384: hipLaunchKernelGGL(sp_partial_kernel<6>, grid, dim3(256), 0, s, A, W, parts, T, N); break;
        case 768: hipLaunchKernelGGL(sp_partial_kernel<12>, grid, dim3(256), 0, s, A, W, parts, T, N); break;
        case 1024: hipLaunchKernelGGL(sp_partial_kernel<16>, grid, dim3(256), 0, s, A, W, parts, T, N); break;
        default: return fail(CS_ERR_UNSUPPORTED, "hidden size %u not supported (384/768/1024)", H);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
