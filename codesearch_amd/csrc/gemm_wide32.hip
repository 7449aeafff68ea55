// gemm_wide32.hip — gemm_wide.hip's persistent dense-layer kernel with its main loop on v_mfma_f32_32x32x16_f16
// (cdna_hip_programming.md §5.4 rule 28: "build both at the same output tile per wave and keep the faster
// by wall, on random data").  Everything outside the MFMA shape is the 16 x 16 x 32 kernel's: the 128 x 384 / 128 x 192
// block, the two-stage LDS image filled by LDS-DMA with the swizzle on the source address, the one-accumulator trick
// (w_hi * 2^11 in registers), persistent blocks with the next tile's first stage in flight under the epilogue, the
// per-wave LDS patch that turns accumulators into whole-line stores, the LayerNorm epilogue at N = 384.
//
// What the shape changes.  A wave's 64 x 96 tile is 2 x 3 tiles of 32 x 32 (the same 96 accumulator registers).  Per
// 32-k chunk a wave still reads its whole operand panels — 64 + 96 rows of 128 B = twenty ds_read_b128 — because that is
// set by the wave tile, not by the instruction; what halves is the number of MFMA instructions (36 of 32 cycles instead
// of 72 of 16), i.e. the issue slots the matrix instructions take from the SIMD's two waves.
//   operand lane map (32x32x16): lane l, r = l & 31, h = l >> 5 holds k = 8 h + j of the 16-k step: logical 16-B slot
//   2 s + h (hi) / 4 + 2 s + h (lo) of the row's 128-B line for step s = 0, 1 of the chunk;
//   accumulator map with W as the FIRST operand: lane holds row m = l & 31 and columns n = 8 q + 4 h + (0..3), q = reg >> 2
//   — four consecutive columns per register quad, as in the 16 x 16 form, so the patch epilogue carries over with a strip
//   = one 32 x 32 tile (32 rows x one 128-byte output line).
// GW_OUT_SWIGLU stays on the 16 x 16 x 32 kernel (its strips want a wave's 48 gated columns together).
#include <cstdlib>

#include "gemm_wide.hpp"

namespace cs {

namespace {
struct GwAcc32 { sh_f32x16 c[2][3]; };
}  // namespace

template <int EPI, int WCN>
__global__ void __launch_bounds__(GwGeom<WCN>::THREADS, 2)
gemm_wide32_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W, const float* __restrict__ bias,
                   const float* resid, float* C, _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t kchunks,
                   uint32_t* __restrict__ flag, uint32_t total_slots, const float* __restrict__ ln_g,
                   const float* __restrict__ ln_b, float ln_eps, uint32_t ln_flags) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using G = GwGeom<WCN>;
    static_assert(EPI != GW_OUT_LN || WCN == 4, "the LayerNorm epilogue needs whole rows in one block");
    constexpr int GW_BN = G::BN, GW_STAGE = G::STAGE, AP = G::A_PIECES, WP = G::W_PIECES, NP = G::PIECES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCN, wc = wave % WCN;
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t mtiles = (M + GW_BM - 1) / GW_BM, ntiles = N / GW_BN;

    // stage image and its fill: gemm_wide.hip
    const int drow = lane >> 3;
    auto src_of = [&](uint32_t row_in_tile, uint32_t grow) {
        const int c = (lane & 7) ^ ((row_in_tile >> 1) & 7);
        return grow * kchunks * 64 + c * 8;
    };
    auto tile_src = [&](uint32_t m0, uint32_t n0, GwSrc<WCN>& s) {
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
    auto dma = [&](const GwSrc<WCN>& s, int p, uint32_t kc, uint32_t bufoff) {
        if (p < AP) sh_glds16(A + (s.a[p < AP ? p : 0] + kc * 64), lds + bufoff + (wave * AP + p) * 1024);
        else sh_glds16(W + (s.w[p >= AP ? p - AP : 0] + kc * 64), lds + bufoff + GW_A_BYTES + (wave * WP + (p - AP)) * 1024);
    };

    // rows wr*64 + 32 it + l31 / wc*96 + 32 jt + l31: (row >> 1) & 7 = (l31 >> 1) & 7 for every it, jt
    const int swz = (l31 >> 1) & 7;
    const uint32_t a_off = (wr * 64 + l31) * 128, w_off = GW_A_BYTES + (wc * 96 + l31) * 128;
    const uint32_t s_hi0 = ((0 + h) ^ swz) * 16, s_hi1 = ((2 + h) ^ swz) * 16;
    const uint32_t s_lo0 = ((4 + h) ^ swz) * 16, s_lo1 = ((6 + h) ^ swz) * 16;

    auto valid = [&](uint32_t slot, uint32_t& mt, uint32_t& nt) { return sh_tile_of_block(slot, mtiles, ntiles, mt, nt); };
    auto next_valid = [&](uint32_t slot, uint32_t& mt, uint32_t& nt) {
        while (slot < total_slots && !valid(slot, mt, nt)) slot += gridDim.x;
        return slot;
    };

    uint32_t mt = 0, nt = 0;
    uint32_t slot = next_valid(blockIdx.x, mt, nt);
    if (slot >= total_slots) return;
    float* const pbias = reinterpret_cast<float*>(lds + 2 * GW_STAGE + GW_STATS);  // (WCN == 4) [N] bias, LayerNorm: + gamma, beta
    constexpr bool lds_params = WCN == 4;
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
    // the bias of this lane's four columns 8 q + 4 h .. + 3 of tile jt (n0: first column of the block's n-tile)
    auto bias_of = [&](uint32_t n0, int jt, int q) -> sh_f32x4 {
        const uint32_t c = n0 + wc * 96 + 32 * jt + 8 * q + 4 * h;
        if constexpr (lds_params) return *reinterpret_cast<const sh_f32x4*>(pbias + c);
        else return *reinterpret_cast<const sh_f32x4*>(bias + c);
    };
    GwSrc<WCN> src;
    tile_src(mt * GW_BM, nt * GW_BN, src);
    uint32_t buf = 0;  // stage buffer (0 | 1) that holds stage 0 of the current tile
#pragma unroll
    for (int p = 0; p < NP; ++p) dma(src, p, sh_kc_rot(nt, ntiles, kchunks), 0);

    while (slot < total_slots) {
        const uint32_t m0 = mt * GW_BM, n0 = nt * GW_BN;
        const uint32_t rot = sh_kc_rot(nt, ntiles, kchunks);
        GwAcc32 acc;
        // accumulators start at bias * 2^11 (LayerNorm: (bias + residual) * 2^11), the scale the products arrive on
        if (EPI == GW_OUT_LN) {
            const char* rbase = reinterpret_cast<const char*>(resid);
            if (ln_flags & GW_LN_RESID_SPLIT) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    f16x4 rh[3][4], rl[3][4];
                    const uint32_t row = m0 + wr * 64 + 32 * it + l31;
                    const uint32_t rrow = (row < M ? row : M - 1) * (GW_BN / 32);
#pragma unroll
                    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t col = wc * 96 + 32 * jt + 8 * q + 4 * h;
                            const char* lp = rbase + (size_t)((rrow + (col >> 5)) * 128u + (col & 31) * 2u);
                            rh[jt][q] = *reinterpret_cast<const f16x4*>(lp);
                            rl[jt][q] = *reinterpret_cast<const f16x4*>(lp + 64);
                        }
#pragma unroll
                    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const sh_f32x4 bv = bias_of(0, jt, q);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc.c[it][jt][4 * q + r] = fmaf((float)rh[jt][q][r], kShLoScale, (float)rl[jt][q][r]) + bv[r] * kShLoScale;
                        }
                }
            } else {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const uint32_t row = m0 + wr * 64 + 32 * it + l31;
                    const uint32_t off = ((row < M ? row : M - 1) * GW_BN + wc * 96 + 4 * h) * 4u;
                    sh_f32x4 rv[3][4];
#pragma unroll
                    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) rv[jt][q] = *reinterpret_cast<const sh_f32x4*>(rbase + (size_t)(off + 128u * jt + 32u * q));
#pragma unroll
                    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const sh_f32x4 bv = bias_of(0, jt, q);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc.c[it][jt][4 * q + r] = (bv[r] + rv[jt][q][r]) * kShLoScale;
                        }
                }
            }
        } else {
#pragma unroll
            for (int jt = 0; jt < 3; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const sh_f32x4 bv = bias_of(n0, jt, q) * kShLoScale;
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc.c[it][jt][4 * q + r] = bv[r];
                }
        }
        __syncthreads();  // stage 0 has landed (vmcnt(0) precedes the barrier)

        for (uint32_t kc = 0; kc < kchunks; ++kc) {
            const char* cur = lds + ((buf + kc) & 1) * GW_STAGE;
            const uint32_t nb = ((buf + kc + 1) & 1) * GW_STAGE;
            const bool more = kc + 1 < kchunks;
            uint32_t kn = rot + kc + 1;
            kn = kn >= kchunks ? kn - kchunks : kn;
            // activation fragments of both 16-k steps (the second step's are not needed before the fourth MFMA group)
            f16x8 ah[2][2], al[2][2];  // [it][s]
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                ah[it][0] = *reinterpret_cast<const f16x8*>(cur + a_off + it * 4096 + s_hi0);
                al[it][0] = *reinterpret_cast<const f16x8*>(cur + a_off + it * 4096 + s_lo0);
            }
            f16x8 wh = *reinterpret_cast<const f16x8*>(cur + w_off + s_hi0);
            f16x8 wl = *reinterpret_cast<const f16x8*>(cur + w_off + s_lo0);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                ah[it][1] = *reinterpret_cast<const f16x8*>(cur + a_off + it * 4096 + s_hi1);
                al[it][1] = *reinterpret_cast<const f16x8*>(cur + a_off + it * 4096 + s_lo1);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {  // (16-k step s, column tile jt) = (j / 3, j % 3)
                const int s = j / 3, jt = j % 3;
                f16x8 whn = wh, wln = wl;
                if (j < 5) {
                    const int sn = (j + 1) / 3, jtn = (j + 1) % 3;
                    whn = *reinterpret_cast<const f16x8*>(cur + w_off + jtn * 4096 + (sn ? s_hi1 : s_hi0));
                    wln = *reinterpret_cast<const f16x8*>(cur + w_off + jtn * 4096 + (sn ? s_lo1 : s_lo0));
                }
                const f16x8 whs = wh * (_Float16)2048.0f;  // exact: |w_hi| < 32 (sh_weights_fit_wide)
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    acc.c[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whs, ah[it][s], acc.c[it][jt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more && 2 * j < NP) dma(src, 2 * j, kn, nb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    acc.c[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah[it][s], acc.c[it][jt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more && 2 * j + 1 < NP) dma(src, 2 * j + 1, kn, nb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    acc.c[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al[it][s], acc.c[it][jt], 0, 0, 0);
                wh = whn;
                wl = wln;
            }
            __syncthreads();  // stage kc+1 has landed; every wave is done reading stage kc
        }

        // next tile of this block: its first stage flies into the buffer the epilogue does not use
        const uint32_t ebuf = (buf + kchunks - 1) & 1;  // buffer of the last stage
        uint32_t nmt = 0, nnt = 0;
        const uint32_t nslot = next_valid(slot + gridDim.x, nmt, nnt);
        if (nslot < total_slots) {
            tile_src(nmt * GW_BM, nnt * GW_BN, src);
#pragma unroll
            for (int p = 0; p < NP; ++p) dma(src, p, sh_kc_rot(nnt, ntiles, kchunks), (ebuf ^ 1) * GW_STAGE);
        }

        // ---- epilogue: per wave, one 32 x 32 tile (32 rows x one 128-byte line) at a time through a private LDS patch ----
        const bool full = m0 + GW_BM <= M;
        uint32_t mx = 0;  // packed maximum of |hi| bit patterns (sh_split8)
        constexpr int PS = 36;  // patch row stride in floats: ds_write_b128 of 8 consecutive rows hit 8 distinct bank quads
        float* patch = reinterpret_cast<float*>(lds + ebuf * GW_STAGE + wave * 8192);  // [32 rows][36 floats]
        float mean[2] = {0.f, 0.f};
        float* rowstat = reinterpret_cast<float*>(lds + 2 * GW_STAGE) + 4 * GW_BM;
        if (EPI == GW_OUT_LN) {
            constexpr float invN = 1.0f / (float)GW_BN;
            float* stats = reinterpret_cast<float*>(lds + 2 * GW_STAGE);  // [4][128] partial sums, [128] row statistic
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < 3; ++jt) acc.c[it][jt] *= kShLoInv;
            auto reduce_rows = [&](bool second) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    float t = 0.0f;
#pragma unroll
                    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float d = second ? acc.c[it][jt][r] - mean[it] : acc.c[it][jt][r];
                            t = second ? fmaf(d, d, t) : t + d;
                        }
                    t += __shfl_xor(t, 32, 64);
                    if (h == 0) stats[wc * GW_BM + wr * 64 + 32 * it + l31] = t;
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
            for (int it = 0; it < 2; ++it) mean[it] = rowstat[wr * 64 + 32 * it + l31];
            reduce_rows(true);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float inv = 1.0f;
            if (EPI == GW_OUT_LN) inv = rowstat[wr * 64 + 32 * it + l31];
#pragma unroll
            for (int jt = 0; jt < 3; ++jt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sh_f32x4 v;
                    if constexpr (EPI == GW_OUT_LN) {
                        const uint32_t c = wc * 96 + 32 * jt + 8 * q + 4 * h;
                        const sh_f32x4 gj = *reinterpret_cast<const sh_f32x4*>(pbias + GW_BN + c);
                        const sh_f32x4 bj = *reinterpret_cast<const sh_f32x4*>(pbias + 2 * GW_BN + c);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (acc.c[it][jt][4 * q + r] - mean[it]) * inv * gj[r] + bj[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc.c[it][jt][4 * q + r] * kShLoInv;  // the bias is in there (accumulator start)
                            if (EPI == SH_OUT_SPLIT_GELU) v[r] = gw_gelu(v[r]);
                        }
                    }
                    *reinterpret_cast<sh_f32x4*>(patch + l31 * PS + 8 * q + 4 * h) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // 32 rows x 4 pieces of 8 columns = 128 pieces, two per lane; four consecutive lanes = one row's line
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int pidx = lane + 64 * t;
                    const int prow = pidx >> 2, q = pidx & 3;
                    const sh_f32x4 v0 = *reinterpret_cast<const sh_f32x4*>(patch + prow * PS + q * 8);
                    const sh_f32x4 v1 = *reinterpret_cast<const sh_f32x4*>(patch + prow * PS + q * 8 + 4);
                    const uint32_t m = wr * 64 + 32 * it + prow;
                    const uint32_t col = n0 + wc * 96 + 32 * jt + q * 8;
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
                            if (EPI == GW_OUT_LN) {  // the next GEMM reads it from L2 / MALL: default policy
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
                __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next tile
            }
        }
        if (flag && sh_split_overflowed(mx)) atomicOr(flag, 1u);
        buf = ebuf ^ 1;
        slot = nslot;
        mt = nmt;
        nt = nnt;
    }
}

template <int WCN>
static int32_t gemm_wide32_launch_t(int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid, float* C,
                                    _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s,
                                    const float* ln_g, const float* ln_b, float ln_eps, uint32_t ln_flags) {
    using G = GwGeom<WCN>;
    static PerDeviceOnce attr_set;  // function attributes are per device
    static int cus = 256;
    CS_TRY(attr_set.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide32_kernel<SH_OUT_F32, WCN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide32_kernel<SH_OUT_F32_RESID, WCN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide32_kernel<SH_OUT_SPLIT_GELU, WCN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide32_kernel<SH_OUT_SPLIT, WCN>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        if constexpr (WCN == 4)
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide32_kernel<GW_OUT_LN, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8)
            cus = n / 8 * 8;  // whole XCD octets: slot -> XCD mapping survives the persistent stride
        return CS_OK;
    }));
    const uint32_t mtiles = (M + GW_BM - 1) / GW_BM, ntiles = N / G::BN;
    const uint32_t slots = sh_grid_blocks(mtiles, ntiles);
    const uint32_t resident = (uint32_t)cus * (WCN == 4 ? 1u : 2u);  // persistent grid: every block resident
    const uint32_t grid = slots < resident ? slots : resident;
    const uint32_t kc = K / 32;
#define GW32_LAUNCH(E) hipLaunchKernelGGL((gemm_wide32_kernel<E, WCN>), dim3(grid), dim3(G::THREADS), G::LDS, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag, slots, ln_g, ln_b, ln_eps, ln_flags)
    if (epi == SH_OUT_F32) GW32_LAUNCH(SH_OUT_F32);
    else if (epi == SH_OUT_F32_RESID) GW32_LAUNCH(SH_OUT_F32_RESID);
    else if (epi == SH_OUT_SPLIT) GW32_LAUNCH(SH_OUT_SPLIT);
    else if (epi == SH_OUT_SPLIT_GELU) GW32_LAUNCH(SH_OUT_SPLIT_GELU);
    else if (epi == GW_OUT_LN) {
        if constexpr (WCN == 4) GW32_LAUNCH(GW_OUT_LN);
        else return fail(CS_ERR_BAD_ARG, "the LayerNorm epilogue needs the 128 x 384 block");
    } else return fail(CS_ERR_BAD_ARG, "epilogue %d is not built on the 32 x 32 x 16 form", epi);
#undef GW32_LAUNCH
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t gemm_wide32_launch(int wcn, int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid,
                           float* C, _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s,
                           const float* ln_g, const float* ln_b, float ln_eps, uint32_t ln_flags) {
    if (wcn == 4) return gemm_wide32_launch_t<4>(epi, A, W, bias, resid, C, Cs, M, N, K, d_flag, s, ln_g, ln_b, ln_eps, ln_flags);
    return gemm_wide32_launch_t<2>(epi, A, W, bias, resid, C, Cs, M, N, K, d_flag, s, ln_g, ln_b, ln_eps, ln_flags);
}

}  // namespace cs
