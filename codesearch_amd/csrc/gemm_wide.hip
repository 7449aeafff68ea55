// gemm_wide.hip — the encoder's dense layers at indexing batch sizes:
// C[M,N] = A[M,K] W[N,K]^T + bias on the f16 MFMA with split-f16 operands (split_f16.hpp), as a PERSISTENT
// kernel over 128 x 384 output tiles with ONE accumulator per output.
//
// Why another GEMM.  The 128 x 128 kernels of gemm_split.hip are bound by operand delivery, not by the matrix
// pipe (a stage costs ~1100 cycles of the CU's global->LDS address path against 768 cycles of
// MFMA, and the two add because every wave issues its LDS-DMAs back to back after the k-step barrier).  Two
// changes remove that bound:
//   * One accumulator instead of two.  sum a*w = sum a_hi*w_hi + 2^-11 sum (a_hi*w_lo' + a_lo'*w_hi) needed a
//     second accumulator set only because the two sums carry different scales.  Multiplying the w_hi fragment
//     by 2^11 IN REGISTERS (exact: a power of two; four v_pk_mul_f16 per fragment) puts all three products on
//     the scale 2^11, so they accumulate into one f32 tile and C = acc * 2^-11.  Same three MFMAs per product,
//     same per-product error bound (~3 * 2^-22), half the accumulator registers.  Needs |w_hi| * 2^11 to stay
//     finite in f16, i.e. |w| < 31.98: checked once per model (sh_weights_fit_wide), else the 128 x 128
//     kernels run.
//   * With 64 accumulator registers freed, a block of 8 waves owns 128 x 384 outputs (wave tile 64 x 96 =
//     4 x 6 MFMA tiles of 16 x 16): a k-step stages 16 KiB of A + 48 KiB of W for 576 MFMAs — 0.83 address-path
//     cycles per MFMA cycle instead of 1.46 — and every N of a BERT encoder whose hidden size is a multiple of
//     384 (384, 1152, 1536; 768, 2304, 3072) is a whole number of tiles, N = 384 a single one.
// The LDS-DMAs of stage k+1 are issued one at a time between the MFMA groups of stage k (eight per wave, all
// within the first two thirds of the step), so the address path works while the matrix pipe does.  The
// kernel is persistent (one block per CU): the first stage of a block's NEXT tile is in flight while the current
// tile's epilogue runs, in the stage buffer the epilogue does not use.
// LDS: two stages of 64 KiB.  The MFMAs take the weight fragment as their FIRST operand, so the accumulator of a
// 16 x 16 tile holds four consecutive columns of one row per lane, and a wave's 64 x 96 tile is whole 128-byte
// output lines: each wave finishes its own rows through a private 6.4 KB patch of the free stage buffer — no block
// barrier in the epilogue (the staged three-pass epilogue this replaces cost a quarter of a tile's time).
#include <cstdlib>

#include "encoder.hpp"

#include <algorithm>
#include <type_traits>
#include <vector>
#include "gemm_wide.hpp"

namespace cs {

// GW_OUT_SWIGLU / GW_OUT_GEGLU (encoder.hpp): the gated up projection of a NomicBert (silu gate) / JinaBert (erf-GELU gate) feed-forward.  W's rows (and the bias) are value and gate
// rows interleaved in groups of 16 — raw columns 32 u .. 32 u + 15 are fc11's rows 16 u .. 16 u + 15, the next sixteen
// fc12's — so a wave's six 16-column MFMA tiles are value, gate, value, gate, value, gate of the SAME 48 gated columns and a
// lane holds a value and its gate in the same register slot of neighbouring tiles: the epilogue stores
// value * silu(gate) in split form, [M][N/64][64], and the 2 x 4 bytes per element of the raw projection never reach HBM.
// (A wave owns 48 gated columns = one and a half 128-byte lines: a third of the lines are completed by two waves.)

namespace {

struct GwAcc { sh_f32x4v c[4][6]; };

}  // namespace

// ABL (diagnostics, cs_debug_gemm_time): 0 = the product kernel; 1 = no LDS-DMA after a tile's first stage (the MFMA +
// LDS-read ceiling); 2 = no MFMA (fill + LDS reads only); 3 = the eight DMAs of a stage issued back to back at the
// start of the step instead of between the MFMA groups; 4 = 1 without the k-step barrier; 5 = 4 with the fragment
// reads hoisted out of the k loop (the pure MFMA rate); 6 = 1 without the epilogue's stores.
// ABL 7: the product kernel plus one (s_memtime, s_memrealtime) pair per block at its first and after its last tile,
// written to a buffer nothing else reads: the in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6).
// Only libcsgpu_diag.so (-DCS_DIAGNOSTICS) instantiates ABL != 0 and holds the stamp buffer; the product library compiles
// the ABL = 0 kernels alone.
#ifdef CS_DIAGNOSTICS
__device__ uint64_t g_gw_stamps[8 * 512];  // per block: clk0, real0, clk1, real1, main-loop cycles, epilogue cycles, tiles
#endif

// Round 4 measured, and did not keep, three re-arrangements of WHEN this kernel's memory instructions issue (diagnostic
// builds of commit "Wide GEMM experiments", logs profiles/r04_gemm_dma_schedule_ab.log, r04_gemm_stagger_by_cu_ab.log,
// r04_gemm_role_split_ab.log): other DMA slots for waves 4-7 than for their SIMD partners 0-3 (four schedules: 0 ... +5 %),
// the next tile's first barrier waiting with vmcnt(24) so the epilogue's stores drain under the first k-step (0 %), the
// two blocks of a CU half a tile out of phase with the pair taken from HW_ID (0 ... +4 %), and loader waves that never
// store beside storing waves that never wait on vmcnt (+7 %).  The no-DMA ablation already runs as fast as the full
// kernel; on all-zero operands the same instruction stream is 15-20 % faster (r04_gemm_zero_vs_random.log): the rest
// is the clock the chip holds under this load, not the schedule.
template <int EPI, int ABL = 0, int WCN = 4, int WRN = 2>
__global__ void __launch_bounds__((GwGeom<WCN, WRN>::THREADS), 2)
gemm_wide_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W, const float* __restrict__ bias,
                 const float* resid, float* C, _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t kchunks,
                 uint32_t* __restrict__ flag, uint32_t total_slots, const float* __restrict__ ln_g,
                 const float* __restrict__ ln_b, float ln_eps, uint32_t ln_flags) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using G = GwGeom<WCN, WRN>;
    static_assert(EPI != GW_OUT_LN || (WCN == 4 && WRN == 2), "the LayerNorm epilogue needs whole rows in one block");
    constexpr int BM = G::BM, A_BYTES = G::A_BYTES;
    constexpr int GW_BN = G::BN, GW_STAGE = G::STAGE, AP = G::A_PIECES, WP = G::W_PIECES, NP = G::PIECES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCN, wc = wave % WCN;
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t mtiles = (M + BM - 1) / BM, ntiles = N / GW_BN;

    // LDS image of a stage: row r of a tile = one 128-B line, logical 16-B slot c at physical slot c ^ ((r >> 1) & 7)
    // (split_f16.hpp); the LDS-DMA destination is lane-linear, so the permutation goes into the source address.
    const int drow = lane >> 3;                  // row of a piece (8 rows x 128 B) this lane fetches
    auto src_of = [&](uint32_t row_in_tile, uint32_t grow) {
        const int c = (lane & 7) ^ ((row_in_tile >> 1) & 7);
        return (grow * kchunks * 64 + c * 8) * 2;  // bytes from the tensor's start
    };
    auto tile_src = [&](uint32_t m0, uint32_t n0, GwSrc<WCN, WRN>& s) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const uint32_t r = (wave * AP + p) * 8 + drow;
            s.a[p] = src_of(r, (m0 + r < M) ? m0 + r : M - 1);  // rows past M re-read row M-1 (never stored)
        }
#pragma unroll
        for (int p = 0; p < WP; ++p) {
            const uint32_t r = (wave * WP + p) * 8 + drow;
            s.w[p] = src_of(r, n0 + r);
        }
    };
    // (buffer loads: split_f16.hpp, sh_blds16 — the operand reads' waits stay counted while the next k-step is on its way)
    const sh_rsrc a_rsrc = sh_make_rsrc(A, M * kchunks * 128u), w_rsrc = sh_make_rsrc(W, N * kchunks * 128u);
    auto dma = [&](const GwSrc<WCN, WRN>& s, int p, uint32_t kc, uint32_t bufoff) {
        if (p < AP) sh_blds16(a_rsrc, s.a[p < AP ? p : 0], kc * 128, lds + bufoff + (wave * AP + p) * 1024);
        else sh_blds16(w_rsrc, s.w[p >= AP ? p - AP : 0], kc * 128, lds + bufoff + A_BYTES + (wave * WP + (p - AP)) * 1024);
    };

    const int swz = (l15 >> 1) & 7;
    const uint32_t a_off = (wr * 64 + l15) * 128, w_off = A_BYTES + (wc * 96 + l15) * 128;
    const uint32_t s_hi = (g ^ swz) * 16, s_lo = ((4 + g) ^ swz) * 16;

    auto valid = [&](uint32_t slot, uint32_t& mt, uint32_t& nt) { return sh_tile_of_block(slot, mtiles, ntiles, mt, nt); };
    auto next_valid = [&](uint32_t slot, uint32_t& mt, uint32_t& nt) {
        while (slot < total_slots && !valid(slot, mt, nt)) slot += gridDim.x;
        return slot;
    };

    uint32_t mt = 0, nt = 0;
    uint32_t slot = next_valid(blockIdx.x, mt, nt);
    if (slot >= total_slots) return;
    float* const pbias = reinterpret_cast<float*>(lds + 2 * GW_STAGE + GW_STATS);  // (WCN == 4) [N] bias, LayerNorm: + gamma, beta
    constexpr bool lds_params = WCN == 4;  // the launcher sends N > GW_PARAM_FLOATS to the 128 x 192 shape
    if constexpr (lds_params) {
        for (uint32_t i = tid; i < N / 4; i += G::THREADS) {
            reinterpret_cast<sh_f32x4*>(pbias)[i] = reinterpret_cast<const sh_f32x4*>(bias)[i];
            if (EPI == GW_OUT_LN) {
                reinterpret_cast<sh_f32x4*>(pbias + N)[i] = reinterpret_cast<const sh_f32x4*>(ln_g)[i];
                reinterpret_cast<sh_f32x4*>(pbias + 2 * N)[i] = reinterpret_cast<const sh_f32x4*>(ln_b)[i];
            }
        }
        __syncthreads();
    }
    // the bias of this lane's four columns of tile j (n0: first column of the block's n-tile)
    auto bias_of = [&](uint32_t n0, int j) -> sh_f32x4 {
        const uint32_t c = n0 + wc * 96 + 16 * j + 4 * g;
        // (a compile-time choice: a runtime select between an LDS and a global pointer becomes a FLAT load, which counts
        // on vmcnt AND lgkmcnt and is waited for with both at zero)
        if constexpr (lds_params) return *reinterpret_cast<const sh_f32x4*>(pbias + c);
        else return *reinterpret_cast<const sh_f32x4*>(bias + c);
    };
    GwSrc<WCN, WRN> src;
    tile_src(mt * BM, nt * GW_BN, src);
    uint32_t buf = 0;  // stage buffer (0 | 1) that holds stage 0 of the current tile
#pragma unroll
    for (int p = 0; p < NP; ++p) dma(src, p, sh_kc_rot(nt, ntiles, kchunks), 0);

    [[maybe_unused]] uint64_t t_clk = 0, t_real = 0, t_main = 0, t_epi = 0, t_mark = 0, n_tiles = 0;
    if (ABL == 7) { t_clk = __builtin_amdgcn_s_memtime(); t_real = __builtin_amdgcn_s_memrealtime(); t_mark = t_clk; }
    while (slot < total_slots) {
        const uint32_t m0 = mt * BM, n0 = nt * GW_BN;
        const uint32_t rot = sh_kc_rot(nt, ntiles, kchunks);  // see sh_mainloop16: n-tiles of an m-tile walk K out of phase
        GwAcc acc;
        // The accumulators START at bias * 2^11 (LayerNorm: (bias + residual) * 2^11), the scale the products arrive on: the
        // epilogue needs neither registers nor loads for them.
        if (EPI == GW_OUT_LN) {
            // The residual is read while stage 0 is in flight.  The form (f32 | split) is decided ONCE, outside the unrolled
            // loops, and every load of a half tile is issued before the first is used: with the choice inside the loops hipcc
            // branches around each load and waits vmcnt(0) behind it — 24 dependent round trips per tile and wave
            // (cdna_hip_programming.md §5, Projection GEMM item 4(c)).
            const char* rbase = reinterpret_cast<const char*>(resid);
            if (ln_flags & GW_LN_RESID_SPLIT) {
                // the residual stream in split form only (same 4 B per element: the row's line [32 hi | 32 lo] of the
                // 32-column chunk): hi * 2^11 + lo' is exact in f32 and already on the accumulators' scale
#pragma unroll
                for (int ih = 0; ih < 2; ++ih) {
                    f16x4 rh[2][6], rl[2][6];
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const uint32_t row = m0 + wr * 64 + 16 * (2 * ih + ii) + l15;
                        const uint32_t rrow = (row < M ? row : M - 1) * (GW_BN / 32);
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            const uint32_t col = wc * 96 + 16 * j + 4 * g;
                            const char* lp = rbase + (size_t)((rrow + (col >> 5)) * 128u + (col & 31) * 2u);
                            rh[ii][j] = *reinterpret_cast<const f16x4*>(lp);
                            rl[ii][j] = *reinterpret_cast<const f16x4*>(lp + 64);
                        }
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            const sh_f32x4 bv = bias_of(0, j);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc.c[2 * ih + ii][j][r] = fmaf((float)rh[ii][j][r], kShLoScale, (float)rl[ii][j][r]) + bv[r] * kShLoScale;
                        }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t row = m0 + wr * 64 + 16 * i + l15;
                    // 32-bit byte offset from a uniform base (a [65536, 384] f32 tensor is 100 MB): one VGPR per row
                    const uint32_t off = ((row < M ? row : M - 1) * GW_BN + wc * 96 + 4 * g) * 4u;
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc.c[i][j] = *reinterpret_cast<const sh_f32x4*>(rbase + (size_t)(off + 64u * j));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc.c[i][j] = (bias_of(0, j) + acc.c[i][j]) * kShLoScale;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const sh_f32x4 bv = bias_of(n0, j) * kShLoScale;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc.c[i][j] = bv;
            }
        }
        __syncthreads();  // stage 0 has landed (vmcnt(0) precedes the barrier)

        {
            constexpr bool DMA_ON = ABL == 0 || ABL == 2 || ABL >= 7;  // ABL 1, 4, 5, 6: no DMA after a tile's first stage
            for (uint32_t kc = 0; kc < kchunks; ++kc) {
                const char* cur = lds + ((buf + kc) & 1) * GW_STAGE;
                const uint32_t nb = ((buf + kc + 1) & 1) * GW_STAGE;
                const bool more = kc + 1 < kchunks;
                uint32_t kn = rot + kc + 1;
                kn = kn >= kchunks ? kn - kchunks : kn;
                if (ABL == 5) cur = lds;  // same addresses every step: the reads hoist out of the loop (pure MFMA rate)
                f16x8 ah[4], al[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(cur + a_off + i * 2048 + s_hi);
                    al[i] = *reinterpret_cast<const f16x8*>(cur + a_off + i * 2048 + s_lo);
                }
                f16x8 wh = *reinterpret_cast<const f16x8*>(cur + w_off + s_hi);
                f16x8 wl = *reinterpret_cast<const f16x8*>(cur + w_off + s_lo);
                if (ABL == 3 && more) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) dma(src, p, kn, nb);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f16x8 whn = wh, wln = wl;
                    if (j < 5) {
                        whn = *reinterpret_cast<const f16x8*>(cur + w_off + (j + 1) * 2048 + s_hi);
                        wln = *reinterpret_cast<const f16x8*>(cur + w_off + (j + 1) * 2048 + s_lo);
                    }
                    const f16x8 whs = wh * (_Float16)2048.0f;  // exact: |w_hi| < 32 (sh_weights_fit_wide)
                    if (ABL == 2) {
                        asm volatile("" ::"v"(whs), "v"(wl), "v"(wh));
#pragma unroll
                        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ABL != 2) acc.c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whs, ah[i], acc.c[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (DMA_ON && more && 2 * j < NP) dma(src, 2 * j, kn, nb);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ABL != 2) acc.c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[i], acc.c[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (DMA_ON && more && 2 * j + 1 < NP) dma(src, 2 * j + 1, kn, nb);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ABL != 2) acc.c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[i], acc.c[i][j], 0, 0, 0);
                    wh = whn;
                    wl = wln;
                }
                if (ABL != 4 && ABL != 5) __syncthreads();  // stage kc+1 has landed; every wave is done reading stage kc
            }
        }
        if (ABL == 4 || ABL == 5) __syncthreads();

        if (ABL == 7) { const uint64_t t = __builtin_amdgcn_s_memtime(); t_main += t - t_mark; t_mark = t; }
        // next tile of this block: its first stage flies into the buffer the epilogue does not use
        const uint32_t ebuf = (buf + kchunks - 1) & 1;  // buffer of the last stage
        uint32_t nmt = 0, nnt = 0;
        const uint32_t nslot = next_valid(slot + gridDim.x, nmt, nnt);
        if (nslot < total_slots) {
            tile_src(nmt * BM, nnt * GW_BN, src);
#pragma unroll
            for (int p = 0; p < NP; ++p) dma(src, p, sh_kc_rot(nnt, ntiles, kchunks), (ebuf ^ 1) * GW_STAGE);
        }

        // ---- epilogue: per wave, through a private LDS patch — no block barrier ------------------------------------
        // The MFMAs take the WEIGHT fragment as their first operand, so a lane holds, per 16 x 16 tile, FOUR CONSECUTIVE
        // COLUMNS n = 16 j + 4 g + r of ONE ROW m = 16 i + l15.  A wave's 64 x 96 tile is whole 128-byte lines of the
        // output (96 columns = three 32-column chunks of the split form, or three lines of f32), so each wave finishes
        // its own rows alone: per strip of 16 rows it writes its values into a private 6.4 KB patch of the free stage
        // buffer (ds_write_b128, rows padded to 400 B: conflict-free) and reads them back in line order, 8 consecutive
        // columns per lane, so every global store is 16 B per lane on consecutive lanes of a line.  (Storing straight
        // from the accumulators, 8 B per lane and plane, was measured: 2.4x slower — partial-line writes.)
        const bool full = m0 + BM <= M;
        uint32_t mx = 0;  // packed maximum of |hi| bit patterns (sh_split8)
        float* patch = reinterpret_cast<float*>(lds + ebuf * GW_STAGE + wave * G::PATCH);  // [16 rows][100 floats]
        constexpr int PS = 100;
        float mean[4] = {0.f, 0.f, 0.f, 0.f};
        float* rowstat = reinterpret_cast<float*>(lds + 2 * GW_STAGE) + 4 * GW_BM;
        if (EPI == GW_OUT_LN) {
            // One n-tile = whole rows: v = acc / 2^11 (bias and residual are in there), then LayerNorm over the 384
            // columns (two passes like encoder.hip ln_row).  A row's columns sit in 4 waves (wc) x 6 tiles x 4 lanes (g).
            constexpr float invN = 1.0f / (float)GW_BN;
            float* stats = reinterpret_cast<float*>(lds + 2 * GW_STAGE);  // [4][128] partial sums, [128] row statistic
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc.c[i][j] *= kShLoInv;
            auto reduce_rows = [&](bool second) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = 0.0f;
#pragma unroll
                    for (int j = 0; j < 6; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = second ? acc.c[i][j][r] - mean[i] : acc.c[i][j][r];
                            t = second ? fmaf(d, d, t) : t + d;
                        }
                    t += __shfl_xor(t, 16, 64);
                    t += __shfl_xor(t, 32, 64);
                    if (g == 0) stats[wc * GW_BM + wr * 64 + 16 * i + l15] = t;
                }
                __syncthreads();
                if (tid < GW_BM) {
                    const float tot = (stats[tid] + stats[GW_BM + tid]) + (stats[2 * GW_BM + tid] + stats[3 * GW_BM + tid]);
                    rowstat[tid] = second ? 1.0f / sqrtf(tot * invN + ln_eps) : tot * invN;
                }
                __syncthreads();
            };
            reduce_rows(false);
#pragma unroll
            for (int i = 0; i < 4; ++i) mean[i] = rowstat[wr * 64 + 16 * i + l15];
            reduce_rows(true);
        }
#pragma unroll
        for (int i = 0; i < (ABL == 6 ? 0 : 4); ++i) {
            // final values of strip i into the patch (row l15, columns 16 j + 4 g .. + 3)
            float inv = 1.0f;
            if (EPI == GW_OUT_LN) inv = rowstat[wr * 64 + 16 * i + l15];
            // gamma / beta per strip from the block's LDS copy (held in registers across the strips they would cost the 48
            // registers the accumulators need; as global loads each strip's would sit behind the previous strip's stores
            // in the vmcnt queue and wait for them)
            if constexpr (EPI == GW_OUT_SWIGLU || EPI == GW_OUT_GEGLU) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    sh_f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gt = acc.c[i][2 * jj + 1][r] * kShLoInv;
                        v[r] = (acc.c[i][2 * jj][r] * kShLoInv) * (EPI == GW_OUT_GEGLU ? gw_gelu(gt) : gw_silu(gt));
                    }
                    *reinterpret_cast<sh_f32x4*>(patch + l15 * PS + 16 * jj + 4 * g) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // 16 rows x 6 pieces of 8 gated columns = 96 pieces
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int pidx = lane + 64 * t;
                    if (pidx < 96) {
                        const int prow = pidx / 6, q = pidx - prow * 6;
                        const sh_f32x4 v0 = *reinterpret_cast<const sh_f32x4*>(patch + prow * PS + q * 8);
                        const sh_f32x4 v1 = *reinterpret_cast<const sh_f32x4*>(patch + prow * PS + q * 8 + 4);
                        const uint32_t m = wr * 64 + 16 * i + prow;
                        const uint32_t col = (n0 >> 1) + wc * 48 + q * 8;  // gated column
                        f16x8 hi, lo;
                        sh_split8(v0, v1, hi, lo, mx);
                        if (full || m0 + m < M) {
                            _Float16* dst = Cs + ((size_t)(m0 + m) * (N / 64) + (col >> 5)) * 64 + (col & 31);
                            __builtin_nontemporal_store(hi, reinterpret_cast<f16x8*>(dst));
                            __builtin_nontemporal_store(lo, reinterpret_cast<f16x8*>(dst + 32));
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                continue;
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                sh_f32x4 v;
                if constexpr (EPI == GW_OUT_LN) {
                    const uint32_t c = wc * 96 + 16 * j + 4 * g;
                    const sh_f32x4 gj = *reinterpret_cast<const sh_f32x4*>(pbias + GW_BN + c);
                    const sh_f32x4 bj = *reinterpret_cast<const sh_f32x4*>(pbias + 2 * GW_BN + c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (acc.c[i][j][r] - mean[i]) * inv * gj[r] + bj[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc.c[i][j][r] * kShLoInv;  // the bias is in there (accumulator start)
                        if (EPI == SH_OUT_SPLIT_GELU) v[r] = gw_gelu(v[r]);
                    }
                }
                *reinterpret_cast<sh_f32x4*>(patch + l15 * PS + 16 * j + 4 * g) = v;
            }
            // wave-local: the writes above are visible to this wave's reads below once lgkmcnt drains
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // 16 rows x 12 pieces of 8 columns = 192 pieces, three per lane
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int pidx = lane + 64 * t;
                const int prow = pidx / 12, q = pidx - prow * 12;  // row of the strip, 8-column piece of the 96
                const sh_f32x4 v0 = *reinterpret_cast<const sh_f32x4*>(patch + prow * PS + q * 8);
                const sh_f32x4 v1 = *reinterpret_cast<const sh_f32x4*>(patch + prow * PS + q * 8 + 4);
                const uint32_t m = wr * 64 + 16 * i + prow;
                const uint32_t col = n0 + wc * 96 + q * 8;
                const bool live = full || m0 + m < M;
                if (EPI == SH_OUT_F32 || EPI == SH_OUT_F32_RESID || EPI == GW_OUT_LN) {
                    if (live) {
                        float* o = C + (size_t)(m0 + m) * N + col;
                        sh_f32x4 o0 = v0, o1 = v1;
                        if (EPI == SH_OUT_F32_RESID) {
                            o0 += *reinterpret_cast<const sh_f32x4*>(resid + (size_t)(m0 + m) * N + col);
                            o1 += *reinterpret_cast<const sh_f32x4*>(resid + (size_t)(m0 + m) * N + col + 4);
                        }
                        if (!(EPI == GW_OUT_LN && (ln_flags & GW_LN_NO_F32))) {
                            *reinterpret_cast<sh_f32x4*>(o) = o0;
                            *reinterpret_cast<sh_f32x4*>(o + 4) = o1;
                        }
                    }
                }
                if (EPI == SH_OUT_SPLIT || EPI == SH_OUT_SPLIT_GELU || EPI == GW_OUT_LN) {
                    f16x8 hi, lo;
                    sh_split8(v0, v1, hi, lo, mx);
                    if (live) {
                        _Float16* dst = Cs + ((size_t)(m0 + m) * (N / 32) + (col >> 5)) * 64 + (col & 31);
                        if (ABL == 10) {
                            // timing only (wrong addresses, same bytes): every store instruction writes 64 consecutive
                            // 16-B chunks = eight whole 128-B lines, what a line-ordered patch would give
                            const size_t chunk = ((((size_t)(mt * ntiles + nt) * G::WAVES + wave) * 4 + i) * 6 + 2 * t) * 64 + lane;
                            __builtin_nontemporal_store(hi, reinterpret_cast<f16x8*>(Cs + chunk * 8));
                            __builtin_nontemporal_store(lo, reinterpret_cast<f16x8*>(Cs + (chunk + 64) * 8));
                        } else if (ABL == 8) {  // epilogue arithmetic and LDS passes, no global stores
                            asm volatile("" ::"v"(hi), "v"(lo));
                        } else if (EPI == GW_OUT_LN || ABL == 9) {  // the next GEMM reads it from L2 / MALL: default policy
                            *reinterpret_cast<f16x8*>(dst) = hi;
                            *reinterpret_cast<f16x8*>(dst + 32) = lo;
                        } else {
                            __builtin_nontemporal_store(hi, reinterpret_cast<f16x8*>(dst));
                            __builtin_nontemporal_store(lo, reinterpret_cast<f16x8*>(dst + 32));
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next strip
        }
        if (ABL == 6) {  // keep the accumulators alive without storing them
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(acc.c[i][j]));
        }
        if (flag && sh_split_overflowed(mx)) atomicOr(flag, 1u);
        if (ABL == 7) { const uint64_t t = __builtin_amdgcn_s_memtime(); t_epi += t - t_mark; t_mark = t; ++n_tiles; }
        buf = ebuf ^ 1;
        slot = nslot;
        mt = nmt;
        nt = nnt;
    }
#ifdef CS_DIAGNOSTICS
    if (ABL == 7 && tid == 0 && blockIdx.x < 512) {
        g_gw_stamps[8 * blockIdx.x + 0] = t_clk;
        g_gw_stamps[8 * blockIdx.x + 1] = t_real;
        g_gw_stamps[8 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
        g_gw_stamps[8 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
        g_gw_stamps[8 * blockIdx.x + 4] = t_main;
        g_gw_stamps[8 * blockIdx.x + 5] = t_epi;
        g_gw_stamps[8 * blockIdx.x + 6] = n_tiles;
    }
#endif
}

#ifdef CS_DIAGNOSTICS
// median over blocks of (shader cycles) / (100 MHz reference ticks) of the last ABL 7 launch, in GHz; main_cycles /
// epi_cycles: wave 0's cycles per tile in the k loop (first barrier to last) and in the epilogue (next tile's first
// DMAs + conversions + stores), medians over blocks
double gemm_wide_read_clock_ghz(double* main_cycles, double* epi_cycles) {
    static uint64_t h[8 * 512];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gw_stamps), sizeof h) != hipSuccess) return 0.0;
    std::vector<double> v, m, e;
    for (int b = 0; b < 512; ++b) {
        const uint64_t* p = h + 8 * b;
        if (p[3] > p[1] && p[6] > 0) {
            v.push_back((double)(p[2] - p[0]) / (double)(p[3] - p[1]) * 0.1);
            m.push_back((double)p[4] / (double)p[6]);
            e.push_back((double)p[5] / (double)p[6]);
        }
    }
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end()); std::sort(m.begin(), m.end()); std::sort(e.begin(), e.end());
    if (main_cycles) *main_cycles = m[m.size() / 2];
    if (epi_cycles) *epi_cycles = e[e.size() / 2];
    return v[v.size() / 2];
}
#endif  // CS_DIAGNOSTICS

// true when every w_hi of a split weight matrix times 2^11 stays finite in f16 (|w_hi| <= 31.98)
__global__ void __launch_bounds__(256)
gw_weight_range_kernel(const _Float16* __restrict__ w, uint64_t nlines, uint32_t* __restrict__ bad) {
    bool b = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines * 4; i += (uint64_t)gridDim.x * blockDim.x) {
        const f16x8 hi = *reinterpret_cast<const f16x8*>(w + (i >> 2) * 64 + (i & 3) * 8);  // the 32 hi values of a line
#pragma unroll
        for (int e = 0; e < 8; ++e) b |= !(fabsf((float)hi[e]) <= 31.98f);
    }
    if (b) atomicOr(bad, 1u);
}

int32_t sh_weights_fit_wide(const _Float16* d_wsplit, uint64_t n_f16, uint32_t* d_scratch_flag, bool* ok, hipStream_t s) {
    CS_HIP(hipMemsetAsync(d_scratch_flag, 0, sizeof(uint32_t), s));
    const uint64_t nlines = n_f16 / 64;
    hipLaunchKernelGGL(gw_weight_range_kernel, dim3(1024), dim3(256), 0, s, d_wsplit, nlines, d_scratch_flag);
    CS_HIP(hipGetLastError());
    uint32_t v = 1;
    CS_HIP(hipMemcpyAsync(&v, d_scratch_flag, sizeof v, hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    CS_HIP(hipMemsetAsync(d_scratch_flag, 0, sizeof(uint32_t), s));
    *ok = v == 0;
    return CS_OK;
}

bool gemm_wide_supported(uint32_t N, uint32_t K) { return N % 192 == 0 && K % 32 == 0 && N > 0 && K > 0; }

#ifdef CS_DIAGNOSTICS
int g_gemm_wide_ablation = 0;  // diagnostics only (cs_debug_gemm_time)
int g_gemm_wide_shape = 0;     // diagnostics only: 192 / 384 overrides CS_GEMM_WIDE_SHAPE
int g_gemm_wide_mfma = 0;      // diagnostics only: 16 / 32 overrides CS_GEMM_WIDE_MFMA
#endif

template <int WCN, int WRN = 2>
static int32_t gemm_wide_launch(int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid, float* C,
                                _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s,
                                const float* ln_g, const float* ln_b, float ln_eps, uint32_t ln_flags) {
    using G = GwGeom<WCN, WRN>;
    static PerDeviceOnce attr_set;  // function attributes are per device
    static int cus = 256;
    CS_TRY(attr_set.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_F32, 0, WCN, WRN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_F32_RESID, 0, WCN, WRN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT_GELU, 0, WCN, WRN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 0, WCN, WRN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<GW_OUT_SWIGLU, 0, WCN, WRN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<GW_OUT_GEGLU, 0, WCN, WRN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        if constexpr (WCN == 4 && WRN == 2)
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<GW_OUT_LN, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8)
            cus = n / 8 * 8;  // whole XCD octets: slot -> XCD mapping survives the persistent stride
        if (cs_lab_env("CS_GEMM_WIDE_DEBUG")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_wide_kernel<SH_OUT_SPLIT, 0, WCN, WRN>, G::THREADS, G::LDS);
            fprintf(stderr, "gemm_wide<WCN=%d, WRN=%d>: %d threads, %d B LDS, occupancy %d blocks per CU\n", WCN, WRN, G::THREADS, G::LDS, nb);
        }
        return CS_OK;
    }));
    const uint32_t mtiles = (M + G::BM - 1) / G::BM, ntiles = N / G::BN;
    const uint32_t slots = sh_grid_blocks(mtiles, ntiles);
    const uint32_t resident = (uint32_t)cus * ((WCN == 4 || WRN == 4) ? 1u : 2u);  // persistent grid: every block resident
    const uint32_t grid = slots < resident ? slots : resident;
    const uint32_t kc = K / 32;
#define GW_LAUNCH(E, V) hipLaunchKernelGGL((gemm_wide_kernel<E, V, WCN, WRN>), dim3(grid), dim3(G::THREADS), G::LDS, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag, slots, ln_g, ln_b, ln_eps, ln_flags)
#ifdef CS_DIAGNOSTICS
    if constexpr (WCN == 4 && WRN == 2) {
        if (g_gemm_wide_ablation && epi == SH_OUT_SPLIT) {
            static PerDeviceOnce abl_attr;  // function attributes are per device
            CS_TRY(abl_attr.run([&]() -> int32_t {
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 3, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 5, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 6, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 7, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 9, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<SH_OUT_SPLIT, 10, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
                return CS_OK;
            }));
            switch (g_gemm_wide_ablation) {
                case 1: GW_LAUNCH(SH_OUT_SPLIT, 1); break;
                case 2: GW_LAUNCH(SH_OUT_SPLIT, 2); break;
                case 4: GW_LAUNCH(SH_OUT_SPLIT, 4); break;
                case 5: GW_LAUNCH(SH_OUT_SPLIT, 5); break;
                case 6: GW_LAUNCH(SH_OUT_SPLIT, 6); break;
                case 7: GW_LAUNCH(SH_OUT_SPLIT, 7); break;
                case 8: GW_LAUNCH(SH_OUT_SPLIT, 8); break;
                case 9: GW_LAUNCH(SH_OUT_SPLIT, 9); break;
                case 10: GW_LAUNCH(SH_OUT_SPLIT, 10); break;
                default: GW_LAUNCH(SH_OUT_SPLIT, 3); break;
            }
            CS_HIP(hipGetLastError());
            return CS_OK;
        }
    }
#endif  // CS_DIAGNOSTICS
    if (epi == SH_OUT_F32) GW_LAUNCH(SH_OUT_F32, 0);
    else if (epi == SH_OUT_F32_RESID) GW_LAUNCH(SH_OUT_F32_RESID, 0);
    else if (epi == SH_OUT_SPLIT) GW_LAUNCH(SH_OUT_SPLIT, 0);
    else if (epi == SH_OUT_SPLIT_GELU) GW_LAUNCH(SH_OUT_SPLIT_GELU, 0);
    else if (epi == GW_OUT_SWIGLU) GW_LAUNCH(GW_OUT_SWIGLU, 0);
    else if (epi == GW_OUT_GEGLU) GW_LAUNCH(GW_OUT_GEGLU, 0);
    else if (epi == GW_OUT_LN) {
        if constexpr (WCN == 4 && WRN == 2) GW_LAUNCH(GW_OUT_LN, 0);
        else return fail(CS_ERR_BAD_ARG, "the LayerNorm epilogue needs the 128 x 384 block");
    } else return fail(CS_ERR_BAD_ARG, "unknown epilogue %d", epi);
#undef GW_LAUNCH
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// Block shape: 128 x 384 / one block per CU where N allows (and always for the LayerNorm epilogue); 128 x 192 / two
// blocks per CU for the other multiples of 192 or with CS_GEMM_WIDE_SHAPE=192 (measured level with each other on every
// layer shape of the encoder: QKV 187 vs 190 us, FFN-up 270 vs 279, FFN-down 223 vs 237).
static int32_t gemm_wide_impl(int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid, float* C,
                              _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s,
                              const float* ln_g, const float* ln_b, float ln_eps, int shape = 0, uint32_t ln_flags = 0) {
    if (!gemm_wide_supported(N, K)) return fail(CS_ERR_UNSUPPORTED, "wide split GEMM needs N %% 192 == 0 and K %% 32 == 0 (N=%u K=%u)", N, K);
    if ((uint64_t)M * (K / 32) * 128 >= (1ull << 32) || (uint64_t)N * (K / 32) * 128 >= (1ull << 32))  // (32-bit buffer offsets in the kernel)
        return fail(CS_ERR_UNSUPPORTED, "wide split GEMM: an operand of 4 GiB or more (M=%u N=%u K=%u)", M, N, K);
    if (M == 0) return CS_OK;
    static const int shape_env = [] { const char* e = cs_lab_env("CS_GEMM_WIDE_SHAPE"); return e ? std::atoi(e) : 0; }();
    // default: 128 x 384, except the bias -> split-store layer (the QKV projection), which four boxes measured 3-4 % faster
    // as 128 x 192 tiles, two blocks per CU (169-173 vs 175-181 us at 65,536 x 1,152 x 384; FFN-up level: profiles/
    // r04_gemm_stagger_by_cu_ab.log code 1000, r04_gemm_role_split_ab.log)
    const int want = shape ? shape : g_gemm_wide_shape ? g_gemm_wide_shape : shape_env ? shape_env : (epi == SH_OUT_SPLIT ? 192 : 384);
    const bool big = epi == GW_OUT_LN || g_gemm_wide_ablation || (want == 384 && N % 384 == 0 && N <= (uint32_t)GW_PARAM_FLOATS);
#ifdef CS_DIAGNOSTICS
    // MFMA shape of the main loop: 16 x 16 x 32 (this file) | 32 x 32 x 16 (gemm_wide32.hip, diagnostic library only: parity-green,
    // 2-8 % slower per layer, profiles/r05_gemm_mfma_shape_ab.log; CS_GEMM_WIDE_MFMA=32)
    static const int mfma_env = [] { const char* e = cs_lab_env("CS_GEMM_WIDE_MFMA"); return e ? std::atoi(e) : 0; }();
    const int mfma = g_gemm_wide_mfma ? g_gemm_wide_mfma : mfma_env;
    if (mfma == 32 && !g_gemm_wide_ablation && epi != GW_OUT_SWIGLU && epi != GW_OUT_GEGLU)
        return gemm_wide32_launch(big ? 4 : 2, epi, A, W, bias, resid, C, Cs, M, N, K, d_flag, s, ln_g, ln_b, ln_eps, ln_flags);
#endif
    if (big) return gemm_wide_launch<4>(epi, A, W, bias, resid, C, Cs, M, N, K, d_flag, s, ln_g, ln_b, ln_eps, ln_flags);
#ifdef CS_DIAGNOSTICS
    // 256 x 192, one block per CU (GwGeom<2, 4>; diagnostic library only): 0.875x / 0.70x the operand bytes per flop of the default
    // shapes, parity-green, and level with them on every layer shape (profiles/r06_gemm_tall_block_ab.log) — the wide GEMMs are not
    // bound by what L2 can deliver to LDS
    if (want == 256) return gemm_wide_launch<2, 4>(epi, A, W, bias, resid, C, Cs, M, N, K, d_flag, s, ln_g, ln_b, ln_eps, ln_flags);
#endif
    return gemm_wide_launch<2>(epi, A, W, bias, resid, C, Cs, M, N, K, d_flag, s, ln_g, ln_b, ln_eps, ln_flags);
}

int32_t launch_gemm_wide(int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid, float* C,
                         _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s, int shape) {
    if (epi == GW_OUT_LN) return fail(CS_ERR_BAD_ARG, "the LayerNorm epilogue goes through launch_gemm_wide_ln");
    return gemm_wide_impl(epi, A, W, bias, resid, C, Cs, M, N, K, d_flag, s, nullptr, nullptr, 0.0f, shape);
}

// X[M,384] = LayerNorm(A W^T + bias + resid) * gamma + beta, written as f32 (X; may alias resid) and in split
// form (Xs): a dense layer with N = 384, its residual add and the LayerNorm behind it (E4 / E6) in one kernel.
// resid_split != null: the residual is read from that split-form tensor instead of `resid` (it may be Xs itself: a
// block reads its own rows before it writes them); X == null: no f32 copy is written — the residual stream then lives
// in split form only (hi + lo / 2048 carries x to 2^-22 relative) and the layer writes 100 MB less per 65,536 rows.
int32_t launch_gemm_wide_ln(const _Float16* A, const _Float16* W, const float* bias, const float* resid, const float* gamma,
                            const float* beta, float eps, float* X, _Float16* Xs, uint32_t M, uint32_t K, uint32_t* d_flag,
                            hipStream_t s, const _Float16* resid_split) {
    if (!resid && !resid_split) return fail(CS_ERR_BAD_ARG, "the LayerNorm epilogue needs a residual");
    const uint32_t flags = (resid_split ? GW_LN_RESID_SPLIT : 0u) | (X ? 0u : GW_LN_NO_F32);
    const float* r = resid_split ? reinterpret_cast<const float*>(resid_split) : resid;
    return gemm_wide_impl(GW_OUT_LN, A, W, bias, r, X, Xs, M, 384, K, d_flag, s, gamma, beta, eps, 0, flags);
}

}  // namespace cs
