// split_f16.hpp — the "split-f16" operand format and the shared MFMA main loop built on it.
//
// gfx950 has no reduced-precision fast path for f32 matmul inputs (no xf32): the exact-f32
// MFMA runs at 1/16 of the f16/bf16 MFMA rate (MI355X_MICROARCH.md § Matrix cores).  The dense
// products of this path (the encoder's QKV/FFN GEMMs, the batched query x corpus scan) therefore
// run on v_mfma_f32_32x32x16_f16 with every f32 operand x stored as TWO f16 values
//
//     x  =  hi  +  lo / 2048  +  e,     hi = rn_f16(x),  lo = rn_f16((x - hi) * 2048)
//     |e| <= 2^-22 |x|  for |x| >= 2^-14;   |e| <= 2^-36 below (hi and lo are then f16 subnormals,
//     which the f16 MFMA consumes exactly: verified at cs_embedder_create, sh_denorm_selftest)
//
// and a product sum  sum a*w  evaluated as  sum a_hi*w_hi  +  2^-11 * sum (a_hi*w_lo + a_lo*w_hi)
// with both sums accumulated in f32 by the MFMA (three f16 MFMAs in place of sixteen-cycles-each
// f32 ones: 3/16 of the matrix-pipe time).  Dropped: a_lo*w_lo <= 2^-22 |a w|.  Per-product error
// is <= ~3 * 2^-22 |a w|, the same order as the rounding of an f32 fmaf chain at K ~ 10^3.
// Values with |x| > 65504 do not fit: producers raise an overflow flag that the host checks
// (cs_embedder falls back to the exact-f32 kernels; the scan uses the result only as a filter).
//
// Memory layout of a logical [rows, K] f32 matrix in split form ("sh" layout), K % 32 == 0:
//     [rows][K/32][64] f16  —  per 32-element k-chunk one 128-B line: 32 hi, then 32 lo
// so a row is K*4 bytes (the f32 size) and every k-chunk of a row is one full cache line.
#pragma once

#include "common.hpp"

namespace cs {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float sh_f32x4 __attribute__((ext_vector_type(4)));
typedef float sh_f32x16 __attribute__((ext_vector_type(16)));

constexpr float kShLoScale = 2048.0f;
constexpr float kShLoInv = 1.0f / 2048.0f;
constexpr float kShMinNormal = 6.103515625e-05f;  // 2^-14
constexpr float kShMax = 65504.0f;

// One value -> (hi, lo).  Returns true when the value does not fit the format.  hi may be an f16
// subnormal: the f16 MFMA takes subnormal inputs at full precision under the kernels' default
// denormal mode (measured: benchmarks/denorm_probe; cs_embedder_create re-checks it on the device
// with sh_denorm_selftest and otherwise stays on the exact-f32 kernels).
__device__ __forceinline__ bool sh_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    // The stored bits must be the ONLY source of the value lo is computed against.  When x is the
    // result of a multiply (GELU's last product), hipcc may produce the stored hi with
    // v_fma_mixlo_f16 — f16(a*b) rounded once from the exact product — while `(float)hi` below is
    // derived from a separate conversion of the f32-rounded x; for x within an f32 ulp of an f16
    // rounding tie (1 element in 4096) the two differ and the pair is off by a whole f16 ulp.
    asm volatile("" : "+v"(hi));
    lo = (_Float16)((x - (float)hi) * kShLoScale);  // x - hi is exact in f32
    return !(fabsf(x) <= kShMax);                    // also true for NaN
}

// Eight values at once, same results bit for bit: packed conversions (v_cvt_pk_f16_f32) and packed f32 arithmetic
// (v_pk_add_f32 / v_pk_mul_f32) take the split from ~7 VALU instructions per value to ~3.5 — the dense layers'
// epilogues are VALU-bound on it.  Range check: `mx` accumulates the packed unsigned maximum of
// |hi| bit patterns (as integers |f16| orders inf = 0x7C00 below every NaN); sh_split_overflowed(mx) at the end.
typedef float sh_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short sh_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sh_split8(sh_f32x4 v0, sh_f32x4 v1, f16x8& hi, f16x8& lo, uint32_t& mx) {
    asm volatile("" : "+v"(v0), "+v"(v1));  // the f32 values are the only source of hi (see sh_split)
    uint32_t* hw = reinterpret_cast<uint32_t*>(&hi);
    uint32_t* lw = reinterpret_cast<uint32_t*>(&lo);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const sh_f32x2 x = p < 2 ? sh_f32x2{v0[2 * p], v0[2 * p + 1]} : sh_f32x2{v1[2 * p - 4], v1[2 * p - 3]};
        f16x2 h = __builtin_convertvector(x, f16x2);
        asm volatile("" : "+v"(h));
        const sh_f32x2 hf = {(float)h[0], (float)h[1]};
        const sh_f32x2 t = (x - hf) * kShLoScale;  // x - hi is exact in f32
        const f16x2 l = __builtin_convertvector(t, f16x2);
        hw[p] = __builtin_bit_cast(uint32_t, h);
        lw[p] = __builtin_bit_cast(uint32_t, l);
        const sh_u16x2 a = __builtin_bit_cast(sh_u16x2, hw[p] & 0x7fff7fffu), b = __builtin_bit_cast(sh_u16x2, mx);
        mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(a, b));
    }
}
__device__ __forceinline__ bool sh_split_overflowed(uint32_t mx) { return (mx & 0xffffu) >= 0x7c00u || (mx >> 16) >= 0x7c00u; }

// ---- 128 x 128 x 32 tile main loop --------------------------------------------------------
// Block = 256 threads = 4 waves as 2 (m) x 2 (n); a wave owns 64 x 64 = 2 x 2 MFMA tiles of
// 32 x 32.  Stage = one k-chunk (32 k) of 128 A rows and 128 W rows = 2 x 16 KiB, brought in
// by global_load_lds_dwordx4 (no VGPR round trip, no ds_write); two stage buffers = 64 KiB,
// two blocks per CU.  LDS image: row r of a tile = 128 B = eight 16-B slots, logical slot c
// (c = 0..3: hi k 8c..8c+7; c = 4..7: lo) stored at physical slot c ^ ((r >> 1) & 7): the
// 16-lane groups of ds_read_b128 then touch 16 distinct 4-bank slots (conflict-free).  The
// LDS-DMA destination is lane-linear, so the permutation is applied to the per-lane SOURCE
// address (cdna_hip_programming.md §5.4 rule 21).
constexpr int SH_BM = 128, SH_BN = 128;
constexpr int SH_TILE_BYTES = 128 * 128;          // one operand tile of one stage
constexpr int SH_STAGE_BYTES = 2 * SH_TILE_BYTES; // A tile | W tile
constexpr int SH_LDS_BYTES = 2 * SH_STAGE_BYTES;  // 65,536

struct ShAcc {
    sh_f32x16 hh[2][2];  // sum a_hi * w_hi
    sh_f32x16 xx[2][2];  // sum a_hi * w_lo' + a_lo' * w_hi   (primes: scaled by 2048)
};

__device__ __forceinline__ void sh_glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// The same with the non-temporal cache policy (aux = 2): for bytes ONE CU reads once (a streamed corpus);
// never for operands other CUs re-read from L2 (MI355X_MICROARCH.md, price list row nt-weights).
__device__ __forceinline__ void sh_glds16_nt(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
}

// A: split rows [m0, m0+128) of a [M][kc][64] matrix (rows >= M re-read row M-1: they only feed
// outputs that are never stored); W likewise with N rows.  acc must be zero-initialised by the
// caller or carry a previous partial sum.  All 256 threads must call.
// kc_rot: the block walks the k-chunks starting at chunk kc_rot (mod kchunks).  Blocks that share an
// A tile (the n-tiles of one m-tile run concurrently on one XCD) are given different rotations, so
// at any moment they read DIFFERENT lines of the tile: one of them misses to HBM, the others hit
// the XCD's L2 a stage later.  In lockstep they would all miss on the same lines at the same time
// (measured: the L2 does not merge them; fill rate 2x lower).
__device__ __forceinline__ void sh_mainloop(const _Float16* __restrict__ A, uint32_t M, uint32_t m0,
                                            const _Float16* __restrict__ W, uint32_t N, uint32_t n0,
                                            uint32_t kchunks, char* lds, ShAcc& acc, uint32_t kc_rot = 0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // staging: wave w moves tile rows [32w, 32w+32) of both operands, 8 rows per instruction
    const _Float16* asrc[4];
    const _Float16* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
#ifdef SH_ABLATE_LINEAR
        const int c = (lane & 7);
#else
        const int c = (lane & 7) ^ ((row >> 1) & 7);
#endif
#ifdef SH_ABLATE_SAMEA
        const uint32_t am = row;
#else
        const uint32_t am = (m0 + row < M) ? m0 + row : M - 1;
#endif
        const uint32_t wn = (n0 + row < N) ? n0 + row : N - 1;
#ifdef SH_ABLATE_TILED  // timing-only address pattern: the 8 rows of one instruction contiguous (1 KiB)
        asrc[i] = A + ((size_t)(am >> 3) * kchunks * 8 + (am & 7)) * 64 + c * 8;
        wsrc[i] = W + ((size_t)(wn >> 3) * kchunks * 8 + (wn & 7)) * 64 + c * 8;
#else
        asrc[i] = A + (size_t)am * kchunks * 64 + c * 8;
        wsrc[i] = W + (size_t)wn * kchunks * 64 + c * 8;
#endif
    }
    auto stage = [&](uint32_t kc, char* buf) {
#ifndef SH_ABLATE_NO_LOAD  // (diagnostic builds only: benchmarks/gemm_probe.hip)
        char* dst = buf + wave * 32 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#ifdef SH_ABLATE_TILED
            sh_glds16(asrc[i] + (size_t)kc * 512, dst + i * 1024);
            sh_glds16(wsrc[i] + (size_t)kc * 512, dst + SH_TILE_BYTES + i * 1024);
#else
            sh_glds16(asrc[i] + (size_t)kc * 64, dst + i * 1024);
            sh_glds16(wsrc[i] + (size_t)kc * 64, dst + SH_TILE_BYTES + i * 1024);
#endif
        }
#endif
    };

    // fragment addresses: MFMA step s (k 16s..16s+15), lane half h -> logical slot 2s + h (hi), 4 + 2s + h (lo)
    const int swz = (l31 >> 1) & 7;
    const int arow = (wr * 64 + l31) * 128, wrow = SH_TILE_BYTES + (wc * 64 + l31) * 128;
    int sl_hi[2], sl_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        sl_hi[s] = ((2 * s + h) ^ swz) * 16;
        sl_lo[s] = ((4 + 2 * s + h) ^ swz) * 16;
    }

    uint32_t kr = kc_rot % kchunks;  // chunk index of the stage being issued
    auto next_chunk = [&]() { const uint32_t c = kr; kr = kr + 1 == kchunks ? 0 : kr + 1; return c; };
    stage(next_chunk(), lds);
    __syncthreads();
    for (uint32_t kc = 0; kc < kchunks; ++kc) {
        char* cur = lds + (kc & 1) * SH_STAGE_BYTES;
        if (kc + 1 < kchunks) stage(next_chunk(), lds + ((kc + 1) & 1) * SH_STAGE_BYTES);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 ah[2], al[2], wh[2], wl[2];
#ifdef SH_ABLATE_NO_LDSREAD
#pragma unroll
            for (int t = 0; t < 2; ++t) { ah[t] = al[t] = wh[t] = wl[t] = f16x8{}; }
#else
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(cur + arow + t * 32 * 128 + sl_hi[s]);
                al[t] = *reinterpret_cast<const f16x8*>(cur + arow + t * 32 * 128 + sl_lo[s]);
                wh[t] = *reinterpret_cast<const f16x8*>(cur + wrow + t * 32 * 128 + sl_hi[s]);
                wl[t] = *reinterpret_cast<const f16x8*>(cur + wrow + t * 32 * 128 + sl_lo[s]);
            }
#endif
#ifdef SH_ABLATE_NO_MFMA
#pragma unroll
            for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(ah[t]), "v"(al[t]), "v"(wh[t]), "v"(wl[t]));
#else
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc.hh[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc.hh[i][j], 0, 0, 0);
                    acc.xx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc.xx[i][j], 0, 0, 0);
                    acc.xx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc.xx[i][j], 0, 0, 0);
                }
#endif
        }
        __syncthreads();  // stage kc+1 has landed (vmcnt(0) precedes the barrier); cur is free
    }
}

// ---- sh_mainloop on v_mfma_f32_16x16x32_f16 -------------------------------------------------------------
// Same tiles, staging, LDS image and barrier protocol as sh_mainloop; a wave's 64 x 64 tile is 4 x 4 MFMA
// tiles of 16 x 16 and one MFMA consumes the whole 32-k line of a row (A/B operand: lane = row | col
// (lane & 15), k-group lane >> 4 of 8 values = 16-B piece g (hi) or 4 + g (lo) of the line).  Same MFMA
// cycles and the same LDS read bytes per flop as the 32 x 32 x 16 form; MI355X_MICROARCH.md (DVFS
// give-back, item 7) measures ~1.12-1.15x the FLOP/s for this shape on random data because the chip
// holds a higher clock under it.  Accumulator layout: 4 registers per tile, n = lane & 15,
// m = 4 (lane >> 4) + r.
typedef float sh_f32x4v __attribute__((ext_vector_type(4)));
struct ShAcc16 {
    sh_f32x4v hh[4][4];
    sh_f32x4v xx[4][4];
};

__device__ __forceinline__ void sh_acc16_zero(ShAcc16& acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc.hh[i][j][r] = 0.0f; acc.xx[i][j][r] = 0.0f; }
}

// kc_begin / kc_count: the block's share of the k-chunks (split-K launches); rows are always kchunks long.
__device__ __forceinline__ void sh_mainloop16(const _Float16* __restrict__ A, uint32_t M, uint32_t m0,
                                              const _Float16* __restrict__ W, uint32_t N, uint32_t n0,
                                              uint32_t kchunks, char* lds, ShAcc16& acc, uint32_t kc_rot = 0,
                                              uint32_t kc_begin = 0, uint32_t kc_count = 0) {
    if (kc_count == 0) kc_count = kchunks;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const _Float16* asrc[4];
    const _Float16* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const uint32_t am = (m0 + row < M) ? m0 + row : M - 1;
        const uint32_t wn = (n0 + row < N) ? n0 + row : N - 1;
        asrc[i] = A + (size_t)am * kchunks * 64 + c * 8;
        wsrc[i] = W + (size_t)wn * kchunks * 64 + c * 8;
    }
    auto stage = [&](uint32_t kc, char* buf) {
        char* dst = buf + wave * 32 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sh_glds16(asrc[i] + (size_t)kc * 64, dst + i * 1024);
            sh_glds16(wsrc[i] + (size_t)kc * 64, dst + SH_TILE_BYTES + i * 1024);
        }
    };
    // rows wr*64 + 16 i + l15: (row >> 1) & 7 = (l15 >> 1) for every i (16 i, 64 wr are multiples of 16)
    const int swz = (l15 >> 1) & 7;
    const int arow = (wr * 64 + l15) * 128, wrow = SH_TILE_BYTES + (wc * 64 + l15) * 128;
    const int s_hi = (g ^ swz) * 16, s_lo = ((4 + g) ^ swz) * 16;

    uint32_t kr = kc_rot % kc_count;  // position inside the block's share
    auto next_chunk = [&]() { const uint32_t c = kc_begin + kr; kr = kr + 1 == kc_count ? 0 : kr + 1; return c; };
    stage(next_chunk(), lds);
    __syncthreads();
    for (uint32_t kc = 0; kc < kc_count; ++kc) {
        char* cur = lds + (kc & 1) * SH_STAGE_BYTES;
        if (kc + 1 < kc_count) stage(next_chunk(), lds + ((kc + 1) & 1) * SH_STAGE_BYTES);
        f16x8 ah[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(cur + arow + i * 16 * 128 + s_hi);
            al[i] = *reinterpret_cast<const f16x8*>(cur + arow + i * 16 * 128 + s_lo);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(cur + wrow + j * 16 * 128 + s_hi);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(cur + wrow + j * 16 * 128 + s_lo);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc.hh[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wh, acc.hh[i][j], 0, 0, 0);
                acc.xx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wl, acc.xx[i][j], 0, 0, 0);
                acc.xx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], wh, acc.xx[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
}

// C tile to LDS as [128 m][128 n] f32 (see sh_acc_to_lds).  Followed by a barrier.
__device__ __forceinline__ void sh_acc16_to_lds(const ShAcc16& acc, float* ctile) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = wr * 64 + i * 16 + 4 * g + r;
                ctile[m * 128 + wc * 64 + j * 16 + l15] = fmaf(acc.xx[i][j][r], kShLoInv, acc.hh[i][j][r]);
            }
    __syncthreads();
}

// ---- 3-stage variant: (64*WM) x 128 x 32 tiles, 2*WM waves, up to 3 stages in flight ----------
// WM = 4: 256 x 128 tile, 8 waves (two per SIMD), one block per CU: 1.33x the MFMA work per byte
// staged into LDS of the 128 x 128 tile, and the LDS-DMA of stages k+1 and k+2 stays in flight
// under the MFMAs of stage k (counted vmcnt, raw s_barrier: cdna_hip_programming.md §5
// "Pipelining across barriers").  One barrier per k-step: it publishes stage k (each wave has
// waited for its own pieces) and retires every read of stage k-1, whose buffer the loads issued
// right after it overwrite.
template <int WM>
struct ShGeom {
    static constexpr int WAVES = 2 * WM;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int BM = 64 * WM;
    static constexpr int A_BYTES = BM * 128;
    static constexpr int W_BYTES = 128 * 128;
    static constexpr int STAGE = A_BYTES + W_BYTES;
    static constexpr int NSTAGE = 3;
    static constexpr int LDS = NSTAGE * STAGE;
    static constexpr int A_PER_WAVE = (BM / 8) / WAVES;   // glds instructions per wave per stage
    static constexpr int W_PER_WAVE = 16 / WAVES;
    static constexpr int NL = A_PER_WAVE + W_PER_WAVE;
};

template <int N>
__device__ __forceinline__ void sh_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WM>
__device__ __forceinline__ void sh_mainloop3(const _Float16* __restrict__ A, uint32_t M, uint32_t m0,
                                             const _Float16* __restrict__ W, uint32_t N, uint32_t n0,
                                             uint32_t kchunks, char* lds, ShAcc& acc, uint32_t kc_rot = 0) {
    using G = ShGeom<WM>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const _Float16* asrc[G::A_PER_WAVE];
    const _Float16* wsrc[G::W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G::A_PER_WAVE; ++i) {
        const int row = (wave * G::A_PER_WAVE + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const uint32_t am = (m0 + row < M) ? m0 + row : M - 1;
        asrc[i] = A + (size_t)am * kchunks * 64 + c * 8;
    }
#pragma unroll
    for (int i = 0; i < G::W_PER_WAVE; ++i) {
        const int row = (wave * G::W_PER_WAVE + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const uint32_t wn = (n0 + row < N) ? n0 + row : N - 1;
        wsrc[i] = W + (size_t)wn * kchunks * 64 + c * 8;
    }
    auto stage = [&](uint32_t kc, char* buf) {
#pragma unroll
        for (int i = 0; i < G::A_PER_WAVE; ++i)
            sh_glds16(asrc[i] + (size_t)kc * 64, buf + (wave * G::A_PER_WAVE + i) * 1024);
#pragma unroll
        for (int i = 0; i < G::W_PER_WAVE; ++i)
            sh_glds16(wsrc[i] + (size_t)kc * 64, buf + G::A_BYTES + (wave * G::W_PER_WAVE + i) * 1024);
    };

    const int swz = (l31 >> 1) & 7;
    const int arow = (wr * 64 + l31) * 128, wrow = G::A_BYTES + (wc * 64 + l31) * 128;
    int sl_hi[2], sl_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        sl_hi[s] = ((2 * s + h) ^ swz) * 16;
        sl_lo[s] = ((4 + 2 * s + h) ^ swz) * 16;
    }

    // Fragment sets are double-buffered in registers: the reads of the next k16-step are in
    // flight under the 12 MFMAs of the current one, and the k-step barrier sits between the two
    // MFMA groups of a stage, so a wave never parks on LDS latency with its matrix pipe idle.
    // The reads are inline asm with hand-counted lgkmcnt (hipcc's own bookkeeping waits
    // lgkmcnt(0) for loop-carried LDS loads, which would expose the whole latency every k-step;
    // cdna_hip_programming.md §5.7 form (iii): "=v" loads, wait-only statement, sched_barrier).
    struct Frags { f16x8 ah[2], al[2], wh[2], wl[2]; };
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;  // LDS byte address (shared aperture low bits)
    auto load_frags = [&](uint32_t buf_off, int s, Frags& f) {
        const uint32_t a_hi = lds_base + buf_off + arow + sl_hi[s], a_lo = lds_base + buf_off + arow + sl_lo[s];
        const uint32_t w_hi = lds_base + buf_off + wrow + sl_hi[s], w_lo = lds_base + buf_off + wrow + sl_lo[s];
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.ah[0]) : "v"(a_hi));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.wh[0]) : "v"(w_hi));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.wl[0]) : "v"(w_lo));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.al[0]) : "v"(a_lo));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.wh[1]) : "v"(w_hi));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.wl[1]) : "v"(w_lo));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.ah[1]) : "v"(a_hi));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.al[1]) : "v"(a_lo));
    };
    auto mfma_group = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc.hh[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.wh[j], acc.hh[i][j], 0, 0, 0);
                acc.xx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.wl[j], acc.xx[i][j], 0, 0, 0);
                acc.xx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.wh[j], acc.xx[i][j], 0, 0, 0);
            }
    };
#define SH_LGKM_WAIT(N)                                            \
    do {                                                           \
        asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                         \
    } while (0)

    uint32_t kr = kc_rot % kchunks;  // chunk index of the stage being issued (see sh_mainloop)
    auto next_chunk = [&]() { const uint32_t c = kr; kr = kr + 1 == kchunks ? 0 : kr + 1; return c; };
    stage(next_chunk(), lds);
    if (kchunks > 1) stage(next_chunk(), lds + G::STAGE);
    if (kchunks > 2) stage(next_chunk(), lds + 2 * G::STAGE);
    if (kchunks > 2) sh_wait_vmcnt<2 * G::NL>();
    else if (kchunks > 1) sh_wait_vmcnt<G::NL>();
    else sh_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    load_frags(0, 0, f0);
    uint32_t cur = 0;  // buffer index of stage kc
    for (uint32_t kc = 0; kc + 1 < kchunks; ++kc) {  // (the last stage is peeled)
        load_frags(cur * G::STAGE, 1, f1);
        SH_LGKM_WAIT(8);  // f0 (the older 8 reads) is back; f1 stays in flight under the MFMAs
        mfma_group(f0);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t nxt = cur + 1 == 3 ? 0 : cur + 1;
        if (kc + 2 < kchunks) sh_wait_vmcnt<G::NL>();  // stage kc+1 landed; kc+2 may still fly
        else sh_wait_vmcnt<0>();
        SH_LGKM_WAIT(0);  // f1 is back = this wave's last reads of stage kc
        __builtin_amdgcn_s_barrier();
        if (kc + 3 < kchunks) stage(next_chunk(), lds + cur * G::STAGE);  // overwrite stage kc's buffer
        load_frags(nxt * G::STAGE, 0, f0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(f1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    load_frags(cur * G::STAGE, 1, f1);
    SH_LGKM_WAIT(8);
    mfma_group(f0);
    __builtin_amdgcn_sched_barrier(0);
    SH_LGKM_WAIT(0);
    mfma_group(f1);
#undef SH_LGKM_WAIT
    __syncthreads();  // every wave is done reading the stage buffers (callers reuse them)
}

// ---- sh_mainloop3 with the LDS-DMA issue spread through the MFMA stream -----------------------------
// Measured (benchmarks/gemm_probe ablations, QKV shape): the CU's address path accepts one 1-KiB
// LDS-DMA instruction per ~35 cycles even when every line hits, i.e. a 128x128x32 stage costs ~1100
// cycles of it against 768 cycles of MFMA, and a wave that issues its DMAs back to back right after the
// k-step barrier (as sh_mainloop and sh_mainloop3 do, all waves at the same moment) sits in instruction
// issue for that long with the matrix pipe idle: load time and MFMA time ADD (loads only 146 us, MFMA +
// LDS reads only 125 us, kernel 253 us).  Here each wave issues ONE DMA after every third MFMA, half of
// a stage's pieces behind the barrier and half in front of the next one, so the address path works
// while the matrix pipe does.  Ring protocol and vmcnt counts as in sh_mainloop3 (a stage is still
// complete one full k-step before the barrier that publishes it).
template <int WM>
__device__ __forceinline__ void sh_mainloop3i(const _Float16* __restrict__ A, uint32_t M, uint32_t m0,
                                              const _Float16* __restrict__ W, uint32_t N, uint32_t n0,
                                              uint32_t kchunks, char* lds, ShAcc& acc, uint32_t kc_rot = 0) {
    using G = ShGeom<WM>;
    static_assert(G::A_PER_WAVE == 4 && G::W_PER_WAVE == 2, "DMA interleave below is written for 256 x 128 tiles");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // this wave's six pieces of a stage: p = 0..3 A rows, p = 4,5 W rows; LDS offsets inside a stage
    const _Float16* src[6];
    uint32_t dst[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const uint32_t am = (m0 + row < M) ? m0 + row : M - 1;
        src[i] = A + (size_t)am * kchunks * 64 + c * 8;
        dst[i] = (wave * 4 + i) * 1024;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const uint32_t wn = (n0 + row < N) ? n0 + row : N - 1;
        src[4 + i] = W + (size_t)wn * kchunks * 64 + c * 8;
        dst[4 + i] = G::A_BYTES + (wave * 2 + i) * 1024;
    }
    // piece order of a stage: first half {A0, W0, A1}, second half {A2, W1, A3}
    constexpr int kOrder[6] = {0, 4, 1, 2, 5, 3};
    auto dma = [&](int q, uint32_t kc, uint32_t slot) {
        const int p = kOrder[q];
        sh_glds16(src[p] + (size_t)kc * 64, lds + slot * G::STAGE + dst[p]);
    };

    const int swz = (l31 >> 1) & 7;
    const int arow = (wr * 64 + l31) * 128, wrow = G::A_BYTES + (wc * 64 + l31) * 128;
    int sl_hi[2], sl_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        sl_hi[s] = ((2 * s + h) ^ swz) * 16;
        sl_lo[s] = ((4 + 2 * s + h) ^ swz) * 16;
    }
    struct Frags { f16x8 ah[2], al[2], wh[2], wl[2]; };
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    auto load_frags = [&](uint32_t buf_off, int s, Frags& f) {
        const uint32_t a_hi = lds_base + buf_off + arow + sl_hi[s], a_lo = lds_base + buf_off + arow + sl_lo[s];
        const uint32_t w_hi = lds_base + buf_off + wrow + sl_hi[s], w_lo = lds_base + buf_off + wrow + sl_lo[s];
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.ah[0]) : "v"(a_hi));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.wh[0]) : "v"(w_hi));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.wl[0]) : "v"(w_lo));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.al[0]) : "v"(a_lo));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.wh[1]) : "v"(w_hi));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.wl[1]) : "v"(w_lo));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.ah[1]) : "v"(a_hi));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.al[1]) : "v"(a_lo));
    };
    // 12 MFMAs; when `issue`, DMA pieces q0, q0+1, q0+2 of (kc, slot) go out after MFMAs 3, 6 and 9
    auto mfma_group = [&](const Frags& f, bool issue, int q0, uint32_t kc, uint32_t slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc.hh[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.wh[j], acc.hh[i][j], 0, 0, 0);
                acc.xx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.wl[j], acc.xx[i][j], 0, 0, 0);
                acc.xx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.wh[j], acc.xx[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (issue && (i * 2 + j) < 3) dma(q0 + i * 2 + j, kc, slot);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
#define SH_LGKM_WAIT(N)                                            \
    do {                                                           \
        asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                         \
    } while (0)

    uint32_t kr = kc_rot % kchunks;
    auto next_chunk = [&]() { const uint32_t c = kr; kr = kr + 1 == kchunks ? 0 : kr + 1; return c; };
    // prologue: stages 0..2 whole
    for (uint32_t st = 0; st < 3 && st < kchunks; ++st) {
        const uint32_t c = next_chunk();
#pragma unroll
        for (int q = 0; q < 6; ++q) dma(q, c, st);
    }
    if (kchunks > 2) sh_wait_vmcnt<2 * G::NL>();
    else if (kchunks > 1) sh_wait_vmcnt<G::NL>();
    else sh_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    load_frags(0, 0, f0);
    uint32_t cur = 0;        // slot of stage kc
    uint32_t half_kc = 0;    // chunk of the stage whose second half is still to be issued
    bool half_open = false;  // ... and whether there is one
    for (uint32_t kc = 0; kc + 1 < kchunks; ++kc) {  // (the last stage is peeled)
        load_frags(cur * G::STAGE, 1, f1);
        SH_LGKM_WAIT(8);  // f0 is back; f1 stays in flight under the MFMAs
        // second half of stage kc+2 into the slot stage kc-1 used (free since the previous barrier)
        const uint32_t prv = cur == 0 ? 2 : cur - 1;
        mfma_group(f0, half_open, 3, half_kc, prv);
        half_open = false;
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t nxt = cur + 1 == 3 ? 0 : cur + 1;
        if (kc + 2 < kchunks) sh_wait_vmcnt<G::NL>();  // stage kc+1 landed; kc+2 may still fly
        else sh_wait_vmcnt<0>();
        SH_LGKM_WAIT(0);  // f1 is back = this wave's last reads of stage kc
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        load_frags(nxt * G::STAGE, 0, f0);
        __builtin_amdgcn_sched_barrier(0);
        // first half of stage kc+3 into stage kc's slot
        const bool more = kc + 3 < kchunks;
        if (more) { half_kc = next_chunk(); half_open = true; }
        mfma_group(f1, more, 0, half_kc, cur);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    load_frags(cur * G::STAGE, 1, f1);
    SH_LGKM_WAIT(8);
    mfma_group(f0, false, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    SH_LGKM_WAIT(0);
    mfma_group(f1, false, 0, 0, 0);
#undef SH_LGKM_WAIT
    __syncthreads();  // every wave is done reading the stage buffers (callers reuse them)
}

__device__ __forceinline__ void sh_acc_zero(ShAcc& acc) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc.hh[i][j][r] = 0.0f; acc.xx[i][j][r] = 0.0f; }
}

// The tile's f32 values into LDS as [128 m][128 n] (64 KiB, reusing the stage buffers; the main
// loop's final barrier has retired every read of them).  C/D map of the 32x32 MFMA:
// n = lane & 31, m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Followed by a barrier.
// (with 2*WM waves the tile is [64*WM m][128 n]: the same code, wr = wave >> 1 runs to WM-1)
__device__ __forceinline__ void sh_acc_to_lds(const ShAcc& acc, float* ctile) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                ctile[m * 128 + wc * 64 + j * 32 + l31] = fmaf(acc.xx[i][j][r], kShLoInv, acc.hh[i][j][r]);
            }
    __syncthreads();
}

// XCD-aware tile order (cdna_hip_programming.md T1): blocks b and b + 8 share an XCD's L2, so
// each XCD walks its own m-tiles with the n-tiles of one m-tile back to back (the A tile is
// fetched into one L2; W is small and resident in all eight).  Returns false for padding blocks.
__device__ __forceinline__ bool sh_tile_of_block(uint32_t b, uint32_t mtiles, uint32_t ntiles,
                                                 uint32_t& mt, uint32_t& nt) {
    const uint32_t xcd = b & 7, q = b >> 3;
    mt = (q / ntiles) * 8 + xcd;
    nt = q % ntiles;
    return mt < mtiles;
}
// Rotation of n-tile nt's k-walk: spread the n-tiles of an m-tile evenly over the k-chunks.
__device__ __forceinline__ uint32_t sh_kc_rot(uint32_t nt, uint32_t ntiles, uint32_t kchunks) {
    const uint32_t stride = kchunks / ntiles > 0 ? kchunks / ntiles : 1;
    return (nt * stride) % kchunks;
}
inline uint32_t sh_grid_blocks(uint32_t mtiles, uint32_t ntiles) { return ((mtiles + 7) / 8) * 8 * ntiles; }

// ---- host-visible launchers (gemm_split.hip) ----------------------------------------------
// True when the f16 MFMA on the current device multiplies f16 subnormal inputs exactly.
int32_t sh_denorm_selftest(bool* ok, hipStream_t s);
// rows x K f32 -> split layout; flag (device u32, may be null) is OR-ed with 1 on overflow;
// d_row_norm (may be null): row r is divided by d_row_norm[r] first (zero norm -> zero row).
int32_t launch_split_rows(const float* d_src, _Float16* d_dst, uint64_t rows, uint32_t K, uint32_t* d_flag,
                          hipStream_t s, const float* d_row_norm = nullptr);

}  // namespace cs
