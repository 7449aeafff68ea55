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

#include <algorithm>
#include <cstdlib>

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
// WIDE (many row tiles: a query AND its variants): a block = 16 rows x 64 columns, a wave = one 16-column tile over ALL of K — a
// quarter of the blocks, so a quarter of the redundant LayerNorm prologues (at 144 rows the 72 column tiles of a row tile read the
// same four partial slabs 72 times: 80 MB per launch).  The wave keeps the four partial sums the four waves of the narrow form hold
// (chunks v, v + 4, ... for v = 0..3, in that order) and adds them the same way: the same bits.
template <int NPL, int EPI, int PRO, bool WIDE = false>
__global__ void __launch_bounds__(256)
sp_ln_gemm_kernel(SpLnGemmArgs a) {
    using G = SpGeom<NPL>;
    constexpr int H = G::H, U = G::U;
    __shared__ __attribute__((aligned(16))) char aimg[16 * G::AROW];
    __shared__ float red[WIDE ? 1 : 4][16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t n0 = WIDE ? blockIdx.x * 64 + wave * 16 : blockIdx.x * 16, m0 = blockIdx.y * 16, T = a.T;
    const bool leader = blockIdx.x == 0;
    bool ovf = false;

    // this wave's W fragments leave first: they depend on nothing
    constexpr uint32_t kchunks = H / 32;
    constexpr int NWF = WIDE ? 4 * U : U;  // narrow: chunks wave, wave + 4, ...; wide: every chunk, fragment v * U + u = chunk v + 4 u
    const _Float16* wp = a.W + (size_t)(n0 + l15) * kchunks * 64 + 8 * g;
    f16x8 wh[NWF], wl[NWF];
#pragma unroll
    for (int i = 0; i < NWF; ++i) {
        const int chunk = WIDE ? (i / U) + 4 * (i % U) : wave + 4 * i;
        wh[i] = *reinterpret_cast<const f16x8*>(wp + (size_t)chunk * 64);
        wl[i] = *reinterpret_cast<const f16x8*>(wp + (size_t)chunk * 64 + 32);
    }
    const int em = tid >> 4, en = tid & 15;
    const float bias_v = WIDE ? a.bias[n0 + l15] : a.bias[n0 + en];

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
    const char* irow = aimg + l15 * G::AROW + g * 16;
    if constexpr (WIDE) {
        float part[4][4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            sh_f32x4v hh = {0.f, 0.f, 0.f, 0.f}, xx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(irow + (v + 4 * u) * 128);
                const f16x8 al = *reinterpret_cast<const f16x8*>(irow + (v + 4 * u) * 128 + 64);
                hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[v * U + u], hh, 0, 0, 0);
                xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[v * U + u], xx, 0, 0, 0);
                xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[v * U + u], xx, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) part[v][r] = fmaf(xx[r], kShLoInv, hh[r]);
        }
        // C/D layout: column n0 + l15, rows m0 + 4 g + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t row = m0 + 4 * g + r, col = n0 + l15;
            if (row < T) {
                float o = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]) + bias_v;
                if (EPI == SH_OUT_SPLIT_GELU) o = sh_gelu_erf(o);
                _Float16 hi, lo;
                ovf |= sh_split(o, hi, lo);
                _Float16* dst = a.Cs + ((size_t)row * (a.N / 32) + (col >> 5)) * 64 + (col & 31);
                dst[0] = hi;
                dst[32] = lo;
            }
        }
        if (ovf && a.flag) atomicOr(a.flag, 1u);
        return;
    }
    sh_f32x4v hh = {0.f, 0.f, 0.f, 0.f}, xx = {0.f, 0.f, 0.f, 0.f};
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
// WIDE (sp_ln_gemm_kernel's note): a block = 16 rows x 64 columns, a wave = one column tile over the whole slice, keeping the four
// partial sums of the narrow form's four waves and adding them the same way.
template <int NPL, bool WIDE = false>
__global__ void __launch_bounds__(256)
sp_partial_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ parts, uint32_t T, uint32_t N) {
    constexpr int U = SpGeom<NPL>::U;
    constexpr uint32_t kchunks = 8 * NPL, kq = kchunks / 4;  // K = 4 H = 256 NPL
    __shared__ float red[WIDE ? 1 : 4][16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t n0 = WIDE ? blockIdx.x * 64 + wave * 16 : blockIdx.x * 16, m0 = blockIdx.y * 16, ks = blockIdx.z;
    const uint32_t r = m0 + l15;
    const _Float16* ap = A + (size_t)(r < T ? r : T - 1) * kchunks * 64 + 8 * g;
    const _Float16* wp = W + (size_t)(n0 + l15) * kchunks * 64 + 8 * g;
    constexpr int NF = WIDE ? 4 * U : U;  // narrow: chunks wave, wave + 4, ... of the slice; wide: all, fragment v * U + u = chunk v + 4 u
    f16x8 ah[NF], al[NF], wh[NF], wl[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int chunk = WIDE ? (i / U) + 4 * (i % U) : wave + 4 * i;
        const size_t off = (size_t)(ks * kq + chunk) * 64;
        wh[i] = *reinterpret_cast<const f16x8*>(wp + off);
        wl[i] = *reinterpret_cast<const f16x8*>(wp + off + 32);
        ah[i] = *reinterpret_cast<const f16x8*>(ap + off);
        al[i] = *reinterpret_cast<const f16x8*>(ap + off + 32);
    }
    if constexpr (WIDE) {
        float part[4][4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            sh_f32x4v hh = {0.f, 0.f, 0.f, 0.f}, xx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[v * U + u], wh[v * U + u], hh, 0, 0, 0);
                xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[v * U + u], wl[v * U + u], xx, 0, 0, 0);
                xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[v * U + u], wh[v * U + u], xx, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) part[v][q] = fmaf(xx[q], kShLoInv, hh[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // C/D layout: column n0 + l15, rows m0 + 4 g + q
            const uint32_t row = m0 + 4 * g + q;
            if (row < T) parts[((size_t)ks * T + row) * N + n0 + l15] = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
        }
        return;
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

// ---- attention as the out-projection's prologue (sequences of up to 32 tokens, hidden 384 = 12 heads of 32) ---------------------------
// The out-projection of the small path is gemm_sh_skinny_kernel's tile: a block = one 16 x 16 output tile, its four waves split
// K = 384 into the chunks w, w + 4, w + 8 — which are the HEADS w, w + 4, w + 8 (head_dim 32 = one 32-k chunk).  So a wave can
// compute, for the block's 16 rows, exactly the three heads' attention output it then multiplies: no attention launch, no context
// tensor in HBM, no exchange between the waves before the partial tiles meet.  Every block recomputes its rows' attention (the
// trick the LayerNorm prologues already play); with at most 64 keys that is 12-24 MFMAs and one softmax step per head.
//   The keys of a block: the token rows of every sequence its 16 rows touch, kb .. kb + 16 NT - 1 (NT = 2 tiles for sequences of
//     up to 16 tokens, 4 up to 32); a key counts for a query when it lies in the query's own sequence and its mask word is set.
//   S^T (keys x queries) = K (A: lane = key, its 32 d) x Q^T (B: lane = query), one 16-key tile per MFMA.
//   The lane of query n holds keys 4 g + r of each tile: as the next MFMA's k index that is, per pair of tiles, the order (g, j) ->
//     key 4 g + j | 16 + 4 g + (j - 4), which V^T follows through two transposing LDS reads per 16-d tile (ds_read_b64_tr_b16: a
//     16-lane group turns a 4-key x 16-d block into "lane d holds its 4 keys"; V's 128-byte lines [32 hi | 32 lo] are staged as they
//     lie by LDS-DMA, wave-private: the wave's own vmcnt orders them).
//   O^T (d x queries) leaves lane (query, g) holding d = 4 g + r | 16 + 4 g + r of the head: again the k order of the product that
//     follows, and W's fragment is read in that order (two 8-byte pieces per plane instead of one 16-byte piece).
// Every global operand of the block — Q, K, W fragments, mask words, bias and residual of the epilogue — is requested before anything
// is waited for: one memory round trip per launch.  Arithmetic: the split products of the other kernels (hi x hi + (hi x lo + lo x
// hi) 2^-11), softmax in the exp2 domain with the additive mask of attention_shx_body.hpp, probabilities split like any activation.
// Not the bits of attention_shx_kernel (32 x 32 tiles, another summation order): equal to it within 2e-6 at the embedding
// (tests/test_gpu_small_forward.py).
constexpr float kSpLog2e = 1.4426950408889634f, kSpMasked = -3.0e38f;
typedef short sp_s16x4 __attribute__((ext_vector_type(4)));
union SpFrag8 { f16x8 v; uint2 d[2]; };
union SpFragTr { f16x8 v; sp_s16x4 q[2]; };

template <int NT>
__global__ void __launch_bounds__(256)
sp_attn_proj_kernel(const _Float16* __restrict__ qkvs, const int32_t* __restrict__ mask, const _Float16* __restrict__ W,
                    const float* __restrict__ bias, const float* __restrict__ resid, float* __restrict__ C, uint32_t T, uint32_t L,
                    uint32_t* __restrict__ flag) {
    constexpr uint32_t H = 384, NH = 12, NCH = 3 * NH;  // chunks of a qkv row: Q heads | K heads | V heads
    constexpr int VHEAD = NT * 16 * 128;                 // a head's V lines of the block's keys
    extern __shared__ __attribute__((aligned(16))) char vimg[];  // [wave][its head][key][32 hi | 32 lo]
    __shared__ float red[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const uint32_t kb = m0 / L * L;  // the first token row of the first sequence this tile touches
    const float scale_log2e = 0.17677669529663687f * kSpLog2e;  // 1 / sqrt(32)
    char* vw = vimg + wave * 3 * VHEAD;
    // V of this wave's three heads -> LDS as it lies (8 keys x 128 B per instruction; keys past T re-read row T - 1: masked)
#pragma unroll
    for (int hi = 0; hi < 3; ++hi) {
        const uint32_t head = wave + 4 * hi;
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {
            const uint32_t key = kb + 8 * i + (lane >> 3), kc = key < T ? key : T - 1;
            sh_glds16(qkvs + ((size_t)kc * NCH + 2 * NH + head) * 64 + (lane & 7) * 8, vw + hi * VHEAD + i * 1024);
        }
    }
    const uint32_t qrow = m0 + l15, qr = qrow < T ? qrow : T - 1;
    // the epilogue's operands, requested now: (row, col) of this thread's output element
    const int em = tid >> 4, en = tid & 15;
    const uint32_t orow = m0 + em, ocol = n0 + en;
    const float e_bias = bias[ocol], e_resid = resid[(size_t)(orow < T ? orow : T - 1) * H + ocol];
    // the mask words of this lane's keys: 4 g + r of every tile
    int32_t mw[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t key = kb + 16 * t + 4 * g + r;
            mw[t][r] = mask[key < T ? key : T - 1];
        }
    typedef __attribute__((address_space(3))) sp_s16x4* lds_s16x4_p;
    // every operand of the three heads requested before anything is waited for
    // (W's fragment in the k order the attention output will have: row n0 + l15, chunk `head`, pieces 4 g and 16 + 4 g of each plane)
    f16x8 qh[3], ql[3], kh[3][NT], kl[3][NT];
    SpFrag8 wh[3], wl[3];
#pragma unroll
    for (int hi = 0; hi < 3; ++hi) {
        const uint32_t head = wave + 4 * hi;
        const _Float16* qp = qkvs + ((size_t)qr * NCH + head) * 64 + 8 * g;
        qh[hi] = *reinterpret_cast<const f16x8*>(qp);
        ql[hi] = *reinterpret_cast<const f16x8*>(qp + 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint32_t key = kb + 16 * t + l15, kc = key < T ? key : T - 1;
            const _Float16* kp = qkvs + ((size_t)kc * NCH + NH + head) * 64 + 8 * g;
            kh[hi][t] = *reinterpret_cast<const f16x8*>(kp);
            kl[hi][t] = *reinterpret_cast<const f16x8*>(kp + 32);
        }
        const _Float16* wp = W + ((size_t)(n0 + l15) * (H / 32) + head) * 64 + 4 * g;
        wh[hi].d[0] = *reinterpret_cast<const uint2*>(wp);
        wh[hi].d[1] = *reinterpret_cast<const uint2*>(wp + 16);
        wl[hi].d[0] = *reinterpret_cast<const uint2*>(wp + 32);
        wl[hi].d[1] = *reinterpret_cast<const uint2*>(wp + 48);
    }
    // the additive mask of this lane's query for its keys
    const uint32_t seq0 = qr / L * L;  // the query's sequence: token rows seq0 .. seq0 + L - 1
    float madd[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t key = kb + 16 * t + 4 * g + r;
            const bool ok = key < T && key >= seq0 && key < seq0 + L && mw[t][r] != 0;
            madd[t][r] = ok ? 0.0f : kSpMasked;
        }
    sh_f32x4v hh = {0.f, 0.f, 0.f, 0.f}, xx = {0.f, 0.f, 0.f, 0.f};
    bool ovf = false;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's V lines have landed (nobody else reads them), and the rest with them
#pragma unroll
    for (int hi = 0; hi < 3; ++hi) {
        float sc[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sh_f32x4v a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[hi][t], qh[hi], a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[hi][t], ql[hi], b, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[hi][t], qh[hi], b, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[t][r] = fmaf(fmaf(b[r], kShLoInv, a[r]), scale_log2e, madd[t][r]);
        }
        float mx = sc[0][0];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float ps = 0.0f;
        sh_f32x4 pv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[t][r] = __builtin_amdgcn_exp2f(sc[t][r] - mx);
                ps += pv[t][r];
            }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        const float inv = 1.0f / ps;
        f16x8 ph[NT / 2], pl[NT / 2];
#pragma unroll
        for (int s2 = 0; s2 < NT / 2; ++s2) {
            uint32_t pmx = 0;
            sh_split8(pv[2 * s2], pv[2 * s2 + 1], ph[s2], pl[s2], pmx);  // element j: key 32 s2 + (4 g + j | 16 + 4 g + (j - 4))
        }
        sh_f32x4 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            sh_f32x4v a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < NT / 2; ++s2) {
                // lane 4 q + p of a 16-lane group supplies key row 32 s2 + 4 g + q (+ 16), columns 16 dt + 4 p .. + 3 of a plane
                const char* vb = vw + hi * VHEAD + (32 * s2 + 4 * g + (l15 >> 2)) * 128 + (16 * dt + 4 * (l15 & 3)) * 2;
                SpFragTr vh, vl;
                vh.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vb));
                vh.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vb + 16 * 128));
                vl.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vb + 64));
                vl.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vb + 16 * 128 + 64));
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh.v, ph[s2], a, 0, 0, 0);
                b = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh.v, pl[s2], b, 0, 0, 0);
                b = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl.v, ph[s2], b, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] = fmaf(b[r], kShLoInv, a[r]) * inv;  // row qrow, d = 16 dt + 4 g + r of the head
        }
        f16x8 ah, al;
        uint32_t omx = 0;
        sh_split8(o[0], o[1], ah, al, omx);  // element j: k = 32 head + (4 g + j | 16 + 4 g + (j - 4))
        ovf |= sh_split_overflowed(omx);
        hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[hi].v, hh, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[hi].v, xx, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[hi].v, xx, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][4 * g + q][l15] = fmaf(xx[q], kShLoInv, hh[q]);
    __syncthreads();
    if (orow < T)
        C[(size_t)orow * H + ocol] = (red[0][em][en] + red[1][em][en]) + (red[2][em][en] + red[3][em][en]) + e_bias + e_resid;
    if (ovf && flag) atomicOr(flag, 1u);
}

// sequences of L <= 32 tokens in a 384-wide, 12-head model, any number of them the small path takes: a 16-row tile then touches at
// most 2 L <= 32 (L <= 16) or 64 consecutive token rows
bool sp_attn_proj_supported(uint32_t H, uint32_t heads, uint32_t T, uint32_t L) {
    return H == 384 && heads == 12 && T >= 1 && T <= SP_MAX_ROWS && L >= 1 && L <= 32 && T % L == 0;
}

int32_t launch_sp_attn_proj(const _Float16* qkvs, const int32_t* mask, const _Float16* W, const float* bias, const float* resid, float* C,
                            uint32_t T, uint32_t L, uint32_t H, uint32_t heads, uint32_t* flag, hipStream_t s) {
    if (!sp_attn_proj_supported(H, heads, T, L))
        return fail(CS_ERR_UNSUPPORTED, "attention + out-projection in one launch: %u rows of %u tokens, hidden %u, %u heads not built", T, L, H, heads);
    const dim3 grid(H / 16, (T + 15) / 16);
    // the most consecutive token rows a 16-row tile's sequences span: first row of the first one .. last row of the last one
    uint32_t span = 0;
    for (uint32_t m0 = 0; m0 < T; m0 += 16) {
        const uint32_t last = std::min(m0 + 15, T - 1), kb = m0 / L * L, e = (last / L + 1) * L;
        span = std::max(span, e - kb);
    }
    if (span <= 32) {
        hipLaunchKernelGGL(sp_attn_proj_kernel<2>, grid, dim3(256), 4 * 3 * 2 * 16 * 128, s, qkvs, mask, W, bias, resid, C, T, L, flag);
    } else {
        static PerDeviceOnce attr;  // 96 KiB of dynamic LDS: above the default limit
        CS_TRY(attr.run([&]() -> int32_t {
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sp_attn_proj_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * 4 * 16 * 128));
            return CS_OK;
        }));
        hipLaunchKernelGGL(sp_attn_proj_kernel<4>, grid, dim3(256), 4 * 3 * 4 * 16 * 128, s, qkvs, mask, W, bias, resid, C, T, L, flag);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

bool small_path_supported(uint32_t H, uint32_t I, uint32_t T) {
    return (H == 384 || H == 768 || H == 1024) && I == 4 * H && T >= 1 && T <= SP_MAX_ROWS;
}

// rows from which the dense layers take the wide form (a block = 16 x 64): CS_SMALL_WIDE_MIN_ROWS (laboratory knob; 0 = never)
static uint32_t sp_wide_min_rows() {
    static const uint32_t v = [] { const char* e = cs_lab_env("CS_SMALL_WIDE_MIN_ROWS"); return e ? (uint32_t)std::atol(e) : 128u; }();
    return v;
}

template <int NPL>
static void sp_launch_ln_gemm(int epi, int pro, const SpLnGemmArgs& a, hipStream_t s) {
    if (NPL == 6 && sp_wide_min_rows() && a.T >= sp_wide_min_rows() && a.N % 64 == 0) {
        const dim3 grid(a.N / 64, (a.T + 15) / 16);
#define SP_W(E, P) hipLaunchKernelGGL((sp_ln_gemm_kernel<6, E, P, true>), grid, dim3(256), 0, s, a)
        if (epi == SH_OUT_SPLIT) { if (pro == 0) SP_W(SH_OUT_SPLIT, 0); else if (pro == 1) SP_W(SH_OUT_SPLIT, 1); else SP_W(SH_OUT_SPLIT, 2); }
        else { if (pro == 0) SP_W(SH_OUT_SPLIT_GELU, 0); else if (pro == 1) SP_W(SH_OUT_SPLIT_GELU, 1); else SP_W(SH_OUT_SPLIT_GELU, 2); }
#undef SP_W
        return;
    }
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
    if (H == 384 && sp_wide_min_rows() && T >= sp_wide_min_rows() && N % 64 == 0) {
        hipLaunchKernelGGL((sp_partial_kernel<6, true>), dim3(N / 64, (T + 15) / 16, 4), dim3(256), 0, s, A, W, parts, T, N);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
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
