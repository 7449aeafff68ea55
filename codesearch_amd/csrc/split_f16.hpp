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

// ---- 128 x 128 x 32 tile main loop on v_mfma_f32_16x16x32_f16 ----------------------------
// Block = 256 threads = 4 waves as 2 (m) x 2 (n); a wave owns 64 x 64 = 4 x 4 MFMA tiles of
// 16 x 16, and one MFMA consumes the whole 32-k line of a row (A/B operand: lane = row | col
// (lane & 15), k-group lane >> 4 of 8 values = 16-B piece g (hi) or 4 + g (lo) of the line).
// Stage = one k-chunk (32 k) of 128 A rows and 128 W rows = 2 x 16 KiB, brought in by
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write); two stage buffers = 64 KiB,
// two blocks per CU.  LDS image: row r of a tile = 128 B = eight 16-B slots, logical slot c
// (c = 0..3: hi k 8c..8c+7; c = 4..7: lo) stored at physical slot c ^ ((r >> 1) & 7): the
// 16-lane groups of ds_read_b128 then touch 16 distinct 4-bank slots (conflict-free).  The
// LDS-DMA destination is lane-linear, so the permutation is applied to the per-lane SOURCE
// address (cdna_hip_programming.md §5.4 rule 21).
// The 32 x 32 x 16 MFMA form of this loop and the 256 x 128 three-stage rings (rounds 1-2,
// profiles/r01d_gemm_tile_variants.log) executed the same MFMA cycles and LDS bytes per flop and
// measured 0.87-0.89x the FLOP/s on random data (MI355X_MICROARCH.md, DVFS give-back item 7: the
// chip holds a higher clock under the 16 x 16 x 32 shape); they are no longer built.
// Accumulator layout: 4 registers per tile, n = lane & 15, m = 4 (lane >> 4) + r.
constexpr int SH_BM = 128, SH_BN = 128;
constexpr int SH_TILE_BYTES = 128 * 128;          // one operand tile of one stage
constexpr int SH_STAGE_BYTES = 2 * SH_TILE_BYTES; // A tile | W tile
constexpr int SH_LDS_BYTES = 2 * SH_STAGE_BYTES;  // 65,536

__device__ __forceinline__ void sh_glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// The same request as a BUFFER load (buffer_load_dwordx4 ... lds): base + a 32-bit per-lane offset + a wave-uniform offset.  Worth
// the descriptor wherever other waits stand between the request and its covering vmcnt: the compiler files global_load_lds
// under FLAT ("may touch LDS or memory, may complete out of order") and, until a vmcnt wait has covered it, answers every wait it
// inserts itself — the lgkmcnt in front of each MFMA's fragment reads included — with a full drain; buffer loads are counted.
// (Reads past the descriptor's size return zeros instead of faulting.)
typedef __amdgpu_buffer_rsrc_t sh_rsrc;
__device__ __forceinline__ sh_rsrc sh_make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void sh_blds16(sh_rsrc r, uint32_t lane_off, uint32_t wave_off, void* lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)lane_off, (int)wave_off, 0, 0);
}
// The same with the non-temporal cache policy (aux = 2): for bytes ONE CU reads once (a streamed corpus);
// never for operands other CUs re-read from L2 (MI355X_MICROARCH.md, price list row nt-weights).
__device__ __forceinline__ void sh_glds16_nt(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
}

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

// A: split rows [m0, m0+128) of a [M][kc][64] matrix (rows >= M re-read row M-1: they only feed
// outputs that are never stored); W likewise with N rows.  acc must be zero-initialised by the
// caller or carry a previous partial sum.  All 256 threads must call.
// kc_rot: the block walks the k-chunks starting at chunk kc_rot (mod its share).  Blocks that share an
// A tile (the n-tiles of one m-tile run concurrently on one XCD) are given different rotations, so
// at any moment they read DIFFERENT lines of the tile: one of them misses to HBM, the others hit
// the XCD's L2 a stage later.  In lockstep they would all miss on the same lines at the same time
// (measured: the L2 does not merge them; fill rate 2x lower).
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

// The tile's f32 values into LDS as [128 m][128 n] (64 KiB, reusing the stage buffers; the main
// loop's final barrier has retired every read of them).  Followed by a barrier.
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

template <int N>
__device__ __forceinline__ void sh_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
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
