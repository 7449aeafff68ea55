// ffn_fused.hip — the encoder's feed-forward block (SURVEY.md §8a E5 + E6) as ONE persistent kernel per 128 token rows:
//     X_out = LayerNorm( GELU(X W1^T + b1) W2^T + b2 + X ) * gamma + beta
// on the f16 MFMA with split-f16 operands (split_f16.hpp) and the one-accumulator form of gemm_wide.hip.
//
// Why.  As two kernels (gemm_wide<GELU> then gemm_wide<LayerNorm>) the [T, 1536] intermediate makes a round trip
// through HBM in split form — 403 MB written by the first kernel's epilogue (through an LDS patch so that the stores are
// whole lines) and 403 MB pulled back by the second kernel's LDS-DMA fill, which the ablations of DESIGN.md §3.3a call
// "the other half" of that kernel — and each of the four n-tiles of the first GEMM pays a store epilogue.  Here the
// intermediate never leaves the CU: for every chunk of 128 intermediate columns a block
//   (up)    accumulates H[128, 128] = X_tile W1_chunk^T over K = 384 in 32 registers per lane (12 k-steps),
//   (GELU)  applies bias + erf-GELU, splits the values into (hi, lo) and writes them into a 64-KiB LDS image laid out
//           exactly like an A-operand stage (so the down product reads it with the same fragment addresses),
//   (down)  accumulates C[128, 384] += H W2[:, chunk]^T in the 96 accumulator registers per lane of the wide kernel
//           (4 k-steps whose A operand is that image and whose only staged operand is W2),
// and after the last chunk runs the LayerNorm epilogue of gemm_wide.hip (accumulators started at (b2 + residual) 2^11).
// Block = 8 waves as 2 (rows) x 4 (columns): wave tile 64 x 32 in the up product, 64 x 96 in the down product.
// LDS = 64 KiB (H image) + two 48-KiB stage buffers (up step: 16 KiB of X | 16 KiB of W1; down step: 48 KiB of W2) =
// the CU's 160 KiB.  Operands arrive by LDS-DMA issued between the MFMA groups of the previous step, one barrier per
// step, as in gemm_wide.hip.  X is re-streamed once per chunk (12 x 192 KiB per tile, L2 hits).
#include <cstdlib>

#include "encoder.hpp"
#include "split_f16.hpp"

namespace cs {

namespace {

constexpr int FF_BM = 128, FF_N = 384, FF_NC = 128;
constexpr int FF_KU = FF_N / 32;                    // k-steps of the up product (K = hidden = 384): 12
constexpr int FF_KD = FF_NC / 32;                   // k-steps of the down product per chunk: 4
constexpr int FF_HBYTES = FF_BM * FF_NC * 4;        // 65,536: [4 k-chunks][128 rows][128 B]
constexpr int FF_PLANE = FF_BM * 128;               // 16,384: one k-chunk of 128 rows
constexpr int FF_STAGE = 48 * 1024;
constexpr int FF_LDS = FF_HBYTES + 2 * FF_STAGE;    // 163,840
constexpr int FF_THREADS = 512;

__device__ __forceinline__ float ff_erf_fast(float x) {  // gemm_wide.hip gw_erf_fast (same bits)
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
__device__ __forceinline__ float ff_gelu(float v) { return 0.5f * v * (1.0f + ff_erf_fast(v * 0.70710678118654752440f)); }

}  // namespace

// ABL (diagnostics): 0 = product; 1 = no GELU arithmetic (H = bias + acc, timing only); 2 = no LDS-DMA after the first
// stage of a tile (timing only).
template <int ABL>
__global__ void __launch_bounds__(FF_THREADS, 2)
ffn_fused_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W1, const float* __restrict__ b1,
                 const _Float16* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ ln_g,
                 const float* __restrict__ ln_b, float ln_eps, float* X, _Float16* Xs, uint32_t M, uint32_t nchunks,
                 uint32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const hbuf = lds;
    char* const stage0 = lds + FF_HBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t mtiles = (M + FF_BM - 1) / FF_BM;
    const uint32_t kc2 = nchunks * FF_KD;  // k-chunks of a W2 row (intermediate / 32)

    // ---- LDS-DMA sources (element offsets; the 16-B slot permutation of the LDS image goes into the source address)
    const int drow = lane >> 3;
    // Rows of a piece: image row r = 8 q + drow for piece number q of the operand; the slot permutation depends on
    // (r >> 1) & 7 = (4 q + (drow >> 1)) & 7, i.e. on the PARITY of q only — so a lane needs two source offsets per row
    // stride (even / odd pieces) and the rest of a piece's address is wave-uniform (scalar) arithmetic.
    const uint32_t sl_even = (uint32_t)(((lane & 7) ^ ((drow >> 1) & 7)) * 8), sl_odd = (uint32_t)(((lane & 7) ^ ((4 + (drow >> 1)) & 7)) * 8);
    uint32_t src_a[2];                                                 // X rows are clamped per lane on the last tile
    const uint32_t w1_lane[2] = {drow * (FF_KU * 64) + sl_even, drow * (FF_KU * 64) + sl_odd};
    const uint32_t w2_lane[2] = {drow * (kc2 * 64) + sl_even, drow * (kc2 * 64) + sl_odd};
    auto tile_src = [&](uint32_t m0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t r = (wave * 2 + p) * 8 + drow;
            const uint32_t row = m0 + r < M ? m0 + r : M - 1;  // rows past M re-read row M-1 (never stored)
            src_a[p] = row * (FF_KU * 64) + (p ? sl_odd : sl_even);
        }
    };
    // up step (chunk c, k-step ks): pieces 0, 1 = X rows, 2, 3 = W1 rows of the chunk
    auto dma_up = [&](int p, uint32_t c, uint32_t ks, char* buf) {
        if (p < 2) {
            sh_glds16(A + (src_a[p < 2 ? p : 0] + ks * 64), buf + (wave * 2 + p) * 1024);
        } else {
            const uint32_t q = wave * 2 + (p - 2);  // piece number = 8-row group of the chunk's 128 W1 rows
            const uint32_t uni = (c * FF_NC + q * 8) * (FF_KU * 64) + ks * 64;
            sh_glds16(W1 + uni + w1_lane[p & 1], buf + FF_PLANE + q * 1024);
        }
    };
    // down step: pieces 0..5 = W2 rows, k-chunk kq of the intermediate
    auto dma_down = [&](int p, uint32_t kq, char* buf) {
        const uint32_t q = wave * 6 + p;
        const uint32_t uni = q * 8 * (kc2 * 64) + kq * 64;
        sh_glds16(W2 + uni + w2_lane[p & 1], buf + q * 1024);
    };

    const int swz = (l15 >> 1) & 7;
    const uint32_t a_off = (wr * 64 + l15) * 128;
    const uint32_t s_hi = (g ^ swz) * 16, s_lo = ((4 + g) ^ swz) * 16;
    const uint32_t w1_off = FF_PLANE + (wc * 32 + l15) * 128, w2_off = (wc * 96 + l15) * 128;

    uint32_t mt = blockIdx.x;
    if (mt >= mtiles) return;
    tile_src(mt * FF_BM);
#pragma unroll
    for (int p = 0; p < 4; ++p) dma_up(p, 0, 0, stage0);

    uint32_t mx = 0;  // packed maximum of |hi| bit patterns (sh_split8), over H and the outputs
    while (mt < mtiles) {
        const uint32_t m0 = mt * FF_BM;
        sh_f32x4v acc[4][6];
        {
            // the accumulators START at (b2 + residual) * 2^11; the residual is the tile's own X rows, read in split form
            // (hi * 2^11 + lo' is exact in f32 and already on the accumulators' scale) while stage 0 is in flight
            // (lane-derived addresses of the cold parts of the tile — this one, the GELU writes, the epilogue — are rebuilt
            // from an opaque copy of the lane id where they are used: hoisted out of the tile loop they would stay live
            // across the k loops, whose 240 registers have no room for them)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int l15 = lane_o & 15, g = lane_o >> 4;
            const char* rbase = reinterpret_cast<const char*>(A);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t row = m0 + wr * 64 + 16 * i + l15;
                const uint32_t rr = row < M ? row : M - 1;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const uint32_t col = wc * 96 + 16 * j + 4 * g;
                    const sh_f32x4 bv = *reinterpret_cast<const sh_f32x4*>(b2 + col);
                    const char* lp = rbase + (size_t)((rr * (FF_N / 32) + (col >> 5)) * 128u + (col & 31) * 2u);
                    const f16x4 rh = *reinterpret_cast<const f16x4*>(lp);
                    const f16x4 rl = *reinterpret_cast<const f16x4*>(lp + 64);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf((float)rh[r], kShLoScale, (float)rl[r]) + bv[r] * kShLoScale;
                }
            }
        }
        __syncthreads();  // stage 0 has landed (vmcnt(0) precedes the barrier)

        uint32_t buf = 0;  // stage buffer of the current step
        for (uint32_t c = 0; c < nchunks; ++c) {
            // ---------------- up: U[64 x 32 per wave] = X_tile W1_chunk^T ----------------
            sh_f32x4v u[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) u[i][j] = sh_f32x4v{0.f, 0.f, 0.f, 0.f};
            for (uint32_t ks = 0; ks < FF_KU; ++ks) {
                const char* cur = stage0 + buf * FF_STAGE;
                char* nxt = stage0 + (buf ^ 1) * FF_STAGE;
                const bool last_up = ks + 1 == FF_KU;
                f16x8 ah[4], al[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(cur + a_off + i * 2048 + s_hi);
                    al[i] = *reinterpret_cast<const f16x8*>(cur + a_off + i * 2048 + s_lo);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(cur + w1_off + j * 2048 + s_hi);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(cur + w1_off + j * 2048 + s_lo);
                    const f16x8 whs = wh * (_Float16)2048.0f;  // exact: |w_hi| < 32 (sh_weights_fit_wide)
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whs, ah[i], u[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ABL != 2) {  // next step's pieces: three issue points per j (0..5)
                        if (last_up) dma_down(3 * j, c * FF_KD, nxt);
                        else dma_up(j, c, ks + 1, nxt);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[i], u[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ABL != 2) {
                        if (last_up) dma_down(3 * j + 1, c * FF_KD, nxt);
                        else dma_up(2 + j, c, ks + 1, nxt);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[i], u[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ABL != 2 && last_up) dma_down(3 * j + 2, c * FF_KD, nxt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();  // the next step's stage has landed; every wave is done reading this one
                buf ^= 1;
            }
            // ---------------- GELU: H = split(gelu(U / 2^11 + b1)) into the LDS image ----------------
            // A lane holds, per (i, j), row 16 i + l15 of the wave's 64 and the four columns 16 j + 4 g .. + 3 of the wave's
            // 32 = k-chunk plane `wc` of the image, elements 16 j + 4 g ..: hi at byte 2 e of the row's line, lo at 64 + 2 e.
            {
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const int l15 = lane_o & 15, g = lane_o >> 4;
                const int swz = (l15 >> 1) & 7;
                sh_f32x4 bv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const sh_f32x4*>(b1 + c * FF_NC + wc * 32 + 16 * j + 4 * g);
                char* hrow = hbuf + wc * FF_PLANE + (wr * 64 + l15) * 128;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sh_f32x4 v0, v1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v0[r] = fmaf(u[i][0][r], kShLoInv, bv[0][r]);
                        v1[r] = fmaf(u[i][1][r], kShLoInv, bv[1][r]);
                        if (ABL != 1) { v0[r] = ff_gelu(v0[r]); v1[r] = ff_gelu(v1[r]); }
                    }
                    f16x8 hi, lo;
                    sh_split8(v0, v1, hi, lo, mx);
                    const f16x4 h0 = {hi[0], hi[1], hi[2], hi[3]}, h1 = {hi[4], hi[5], hi[6], hi[7]};
                    const f16x4 l0 = {lo[0], lo[1], lo[2], lo[3]}, l1 = {lo[4], lo[5], lo[6], lo[7]};
                    char* row = hrow + i * 2048 + (g & 1) * 8;
                    const int sg = g >> 1;  // logical 16-B slot of j = 0: sg (hi), 4 + sg (lo); j = 1: 2 + sg, 6 + sg
                    *reinterpret_cast<f16x4*>(row + ((sg ^ swz) * 16)) = h0;
                    *reinterpret_cast<f16x4*>(row + (((2 + sg) ^ swz) * 16)) = h1;
                    *reinterpret_cast<f16x4*>(row + (((4 + sg) ^ swz) * 16)) = l0;
                    *reinterpret_cast<f16x4*>(row + (((6 + sg) ^ swz) * 16)) = l1;
                }
            }
            __syncthreads();  // the image is complete
            // ---------------- down: C[64 x 96 per wave] += H W2[:, chunk]^T ----------------
            for (uint32_t kd = 0; kd < FF_KD; ++kd) {
                const char* cur = stage0 + buf * FF_STAGE;
                char* nxt = stage0 + (buf ^ 1) * FF_STAGE;
                const char* hcur = hbuf + kd * FF_PLANE;
                const bool last_down = kd + 1 == FF_KD;
                const bool more_chunks = c + 1 < nchunks;
                f16x8 ah[4], al[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(hcur + a_off + i * 2048 + s_hi);
                    al[i] = *reinterpret_cast<const f16x8*>(hcur + a_off + i * 2048 + s_lo);
                }
                f16x8 wh = *reinterpret_cast<const f16x8*>(cur + w2_off + s_hi);
                f16x8 wl = *reinterpret_cast<const f16x8*>(cur + w2_off + s_lo);
                auto issue = [&](int p) {  // piece p (0..5) of the next step
                    if (ABL == 2) return;
                    if (!last_down) dma_down(p, c * FF_KD + kd + 1, nxt);
                    else if (more_chunks && p < 4) dma_up(p, c + 1, 0, nxt);
                };
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f16x8 whn = wh, wln = wl;
                    if (j < 5) {
                        whn = *reinterpret_cast<const f16x8*>(cur + w2_off + (j + 1) * 2048 + s_hi);
                        wln = *reinterpret_cast<const f16x8*>(cur + w2_off + (j + 1) * 2048 + s_lo);
                    }
                    const f16x8 whs = wh * (_Float16)2048.0f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whs, ah[i], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < 3) issue(2 * j);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[i], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < 3) issue(2 * j + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[i], acc[i][j], 0, 0, 0);
                    wh = whn;
                    wl = wln;
                }
                __syncthreads();
                buf ^= 1;
            }
        }
        // 16 steps per chunk: `buf` is 0 again, and buffer 0 is free (the last step read buffer 1 and the H image).
        // Next tile of this block: its first stage flies into buffer 0 while the epilogue runs.
        const uint32_t nmt = mt + gridDim.x;
        if (nmt < mtiles) {
            tile_src(nmt * FF_BM);
#pragma unroll
            for (int p = 0; p < 4; ++p) dma_up(p, 0, 0, stage0);
        }

        // ---- epilogue: LayerNorm over the 384 columns (gemm_wide.hip GW_OUT_LN), patches in the H image, statistics in
        // stage buffer 1 ----
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o, l15 = lane_o & 15, g = lane_o >> 4;  // shadow the hot-loop copies (see the tile's start)
        const bool full = m0 + FF_BM <= M;
        float* patch = reinterpret_cast<float*>(hbuf + wave * 8192);  // [16 rows][100 floats]
        constexpr int PS = 100;
        float* stats = reinterpret_cast<float*>(stage0 + FF_STAGE);    // [4][128] partial sums, [128] row statistic
        float* rowstat = stats + 4 * FF_BM;
        float mean[4] = {0.f, 0.f, 0.f, 0.f};
        constexpr float invN = 1.0f / (float)FF_N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] *= kShLoInv;
        auto reduce_rows = [&](bool second) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = 0.0f;
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d = second ? acc[i][j][r] - mean[i] : acc[i][j][r];
                        t = second ? fmaf(d, d, t) : t + d;
                    }
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                if (g == 0) stats[wc * FF_BM + wr * 64 + 16 * i + l15] = t;
            }
            __syncthreads();
            const int t512 = wave * 64 + lane;
            if (t512 < FF_BM) {
                const float tot = (stats[t512] + stats[FF_BM + t512]) + (stats[2 * FF_BM + t512] + stats[3 * FF_BM + t512]);
                rowstat[t512] = second ? 1.0f / sqrtf(tot * invN + ln_eps) : tot * invN;
            }
            __syncthreads();
        };
        reduce_rows(false);
#pragma unroll
        for (int i = 0; i < 4; ++i) mean[i] = rowstat[wr * 64 + 16 * i + l15];
        reduce_rows(true);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float inv = rowstat[wr * 64 + 16 * i + l15];
            uint32_t strip_zero = 0;  // keeps the gamma / beta loads of different strips apart, and ordinary (gemm_wide.hip)
            asm volatile("" : "+s"(strip_zero));
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const sh_f32x4 gj = *reinterpret_cast<const sh_f32x4*>(ln_g + strip_zero + wc * 96 + 16 * j + 4 * g);
                const sh_f32x4 bj = *reinterpret_cast<const sh_f32x4*>(ln_b + strip_zero + wc * 96 + 16 * j + 4 * g);
                sh_f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (acc[i][j][r] - mean[i]) * inv * gj[r] + bj[r];
                *reinterpret_cast<sh_f32x4*>(patch + l15 * PS + 16 * j + 4 * g) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int pidx = lane + 64 * t;
                const int prow = pidx / 12, q = pidx - prow * 12;
                const sh_f32x4 v0 = *reinterpret_cast<const sh_f32x4*>(patch + prow * PS + q * 8);
                const sh_f32x4 v1 = *reinterpret_cast<const sh_f32x4*>(patch + prow * PS + q * 8 + 4);
                const uint32_t m = wr * 64 + 16 * i + prow;
                const uint32_t col = wc * 96 + q * 8;
                const bool live = full || m0 + m < M;
                if (X && live) {
                    float* o = X + (size_t)(m0 + m) * FF_N + col;
                    *reinterpret_cast<sh_f32x4*>(o) = v0;
                    *reinterpret_cast<sh_f32x4*>(o + 4) = v1;
                }
                f16x8 hi, lo;
                sh_split8(v0, v1, hi, lo, mx);
                if (live) {
                    _Float16* dst = Xs + ((size_t)(m0 + m) * (FF_N / 32) + (col >> 5)) * 64 + (col & 31);
                    *reinterpret_cast<f16x8*>(dst) = hi;
                    *reinterpret_cast<f16x8*>(dst + 32) = lo;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next strip
        }
        mt = nmt;
        // the next tile's GELU phase rewrites the H image (patches): every wave is past its own patch by then, and the
        // statistics in stage buffer 1 are rewritten by a DMA only after the barriers of the next tile's first steps
    }
    if (flag && sh_split_overflowed(mx)) atomicOr(flag, 1u);
}

bool ffn_fused_supported(uint32_t hidden, uint32_t intermediate) {
    return hidden == FF_N && intermediate % FF_NC == 0 && intermediate >= FF_NC && intermediate <= 8192;
}

int g_ffn_fused_ablation = 0;  // diagnostics only

// X_out = LayerNorm(GELU(A W1^T + b1) W2^T + b2 + A) * gamma + beta for M token rows of hidden 384.
// A / Xs: split form [M][12][64] (Xs may be A itself: a block reads its rows for the last time before it writes them);
// W1 [I][12][64], W2 [384][I/32][64] split weights with |w| < 31.98; X (optional): the f32 copy of the output.
int32_t launch_ffn_fused(const _Float16* A, const _Float16* W1, const float* b1, const _Float16* W2, const float* b2,
                         const float* gamma, const float* beta, float eps, float* X, _Float16* Xs, uint32_t M,
                         uint32_t intermediate, uint32_t* d_flag, hipStream_t s) {
    if (!ffn_fused_supported(FF_N, intermediate)) return fail(CS_ERR_UNSUPPORTED, "fused FFN needs hidden 384 and an intermediate size that is a multiple of 128");
    if (M == 0) return CS_OK;
    static PerDeviceOnce attr_set;
    static int cus = 256;
    CS_TRY(attr_set.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS));
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        return CS_OK;
    }));
    const uint32_t mtiles = (M + FF_BM - 1) / FF_BM;
    const uint32_t grid = mtiles < (uint32_t)cus ? mtiles : (uint32_t)cus;
    const uint32_t nchunks = intermediate / FF_NC;
#define FF_LAUNCH(V) hipLaunchKernelGGL((ffn_fused_kernel<V>), dim3(grid), dim3(FF_THREADS), FF_LDS, s, A, W1, b1, W2, b2, gamma, beta, eps, X, Xs, M, nchunks, d_flag)
    switch (g_ffn_fused_ablation) {
        case 1: FF_LAUNCH(1); break;
        case 2: FF_LAUNCH(2); break;
        default: FF_LAUNCH(0); break;
    }
#undef FF_LAUNCH
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
