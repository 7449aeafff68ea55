// gemm_split.hip — the encoder's dense layers on the f16 MFMA with split-f16 operands
// (split_f16.hpp), SURVEY.md §8a E2/E4/E5/E6:  C[M,N] = A[M,K] W[N,K]^T + bias, then
//   SH_OUT_F32        store f32                                  (QKV projection)
//   SH_OUT_F32_RESID  + residual, store f32 (may be in place)    (attention output / FFN down)
//   SH_OUT_SPLIT_GELU erf-GELU, store in split form              (FFN up; only a GEMM reads it)
// 128x128x32 tiles by LDS-DMA, three v_mfma_f32_16x16x32_f16 per f32 product block, C tile
// staged through LDS so every global access of the epilogue is a full 16 B per lane on
// consecutive lanes.  Replaces the fp32 arithmetic ONNX Runtime does for
// /root/reference/src/embed/embedder.rs:286-289 to within ~3 * 2^-22 per product.
#include <cstdlib>
#include <type_traits>

#include "encoder.hpp"
#include "gemm_epilogue.hpp"
#include "split_f16.hpp"

// Diagnostic builds (benchmarks/gemm_probe.hip) define SH_STAMP to record s_memtime at five points
// of a block's life; the product build compiles it to nothing.
#ifndef SH_STAMP
#define SH_STAMP(i)
#endif

namespace cs {

// 128 x 128 tiles on v_mfma_f32_16x16x32_f16 (sh_mainloop16), two blocks per CU.
template <int EPI>
__global__ void __launch_bounds__(256, 2)
gemm_sh16_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                 const float* __restrict__ bias, const float* resid, float* C,
                 _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t kchunks,
                 uint32_t* __restrict__ flag, uint32_t ksplit) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t mt, nt;
    // split-K (SH_OUT_PARTIAL): gridDim.x = tile slots x ksplit; slice s takes chunks [s, s+1) * kchunks / ksplit
    // and writes its raw partial tile into slab s of C ([ksplit][M][N])
    const uint32_t slots = gridDim.x / ksplit, slice = blockIdx.x / slots;
    if (!sh_tile_of_block(blockIdx.x % slots, (M + SH_BM - 1) / SH_BM, N / SH_BN, mt, nt)) return;
    const uint32_t m0 = mt * SH_BM, n0 = nt * SH_BN;
    SH_STAMP(0);
    ShAcc16 acc;
    sh_acc16_zero(acc);
    const uint32_t kc_begin = (uint32_t)((uint64_t)kchunks * slice / ksplit);
    const uint32_t kc_count = (uint32_t)((uint64_t)kchunks * (slice + 1) / ksplit) - kc_begin;
    if (EPI == SH_OUT_PARTIAL) C += (size_t)slice * M * N;
    sh_mainloop16(A, M, m0, W, N, n0, kchunks, lds, acc, sh_kc_rot(nt, N / SH_BN, kc_count), kc_begin, kc_count);
    SH_STAMP(1);
    float* ctile = reinterpret_cast<float*>(lds);
    sh_acc16_to_lds(acc, ctile);
    SH_STAMP(2);
    if (m0 + SH_BM <= M) gemm_sh_epilogue<EPI, true, 2>(ctile, bias, resid, C, Cs, M, N, m0, n0, flag);
    else gemm_sh_epilogue<EPI, false, 2>(ctile, bias, resid, C, Cs, M, N, m0, n0, flag);
    SH_STAMP(3);
}

// ---- small M (query-side batches: a few short sequences) -------------------------------------------------
// The tiled kernels above are throughput kernels: with a handful of 128-row tiles they leave 95 % of the
// chip idle and walk K one exposed memory latency per stage (FFN-down: 48 stages), ~1 ms per forward of a
// single short query.  Here a block owns ONE 16 x 16 output tile and its four waves split K between them
// (wave w takes k-chunks w, w+4, ...): operands go global -> VGPR straight into the 16x16x32 MFMA fragment
// layout (a row's 128-B line per 32 k is exactly four lanes' hi pieces + four lanes' lo pieces), every load
// of a wave is in flight at once, and the four partial tiles meet in 4 KiB of LDS.  Weights are read once
// per 16 rows of activations; grid = (N / 16) x ceil(M / 16) blocks.
template <int EPI, int RT, int CT>
__global__ void __launch_bounds__(256)
gemm_sh_skinny_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                      const float* __restrict__ bias, const float* resid, float* C,
                      _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t kchunks,
                      uint32_t* __restrict__ flag) {
    // a block owns RT x CT output tiles of 16 x 16 (1 x 1 up to 64 rows; 2 x 2 above: half the operand
    // bytes per flop); each of its four waves computes all of them over a quarter of K
    __shared__ float red[4][RT][CT][16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t n0 = blockIdx.x * 16 * CT, m0 = blockIdx.y * 16 * RT;
    const _Float16* ap[RT];
    const _Float16* wp[CT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const uint32_t r = m0 + 16 * t + l15;
        ap[t] = A + (size_t)(r < M ? r : M - 1) * kchunks * 64 + 8 * g;  // rows past M re-read row M-1 (never stored)
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) wp[t] = W + (size_t)(n0 + 16 * t + l15) * kchunks * 64 + 8 * g;
    sh_f32x4v hh[RT][CT], xx[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) { hh[i][j] = sh_f32x4v{0.f, 0.f, 0.f, 0.f}; xx[i][j] = sh_f32x4v{0.f, 0.f, 0.f, 0.f}; }
    // the epilogue's operands — bias and residual of this thread's output elements — are requested with the tile's operands: read
    // behind the partial tiles' meeting they were a second memory round trip in a kernel that lives for one or two
    const int em = tid >> 4, en = tid & 15;
    float e_bias[CT], e_res[RT][CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) e_bias[j] = bias[n0 + 16 * j + en];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const uint32_t row = m0 + 16 * i + em;
            e_res[i][j] = EPI == SH_OUT_F32_RESID ? resid[(size_t)(row < M ? row : M - 1) * N + n0 + 16 * j + en] : 0.0f;
        }
    // groups of U of this wave's chunks: every 16-B load of a group in flight before its MFMAs.  With one
    // tile per block K = 384 is one group of 3 chunks per wave and K = 1536 one group of 12: one memory
    // latency per GEMM.
    const uint32_t mine = kchunks > (uint32_t)wave ? (kchunks - wave + 3) / 4 : 0;  // chunks wave, wave+4, ...
    auto run_groups = [&](auto ucount) {
        constexpr int U = decltype(ucount)::value;
        for (uint32_t i0 = 0; i0 < mine; i0 += U) {
            f16x8 ah[U][RT], al[U][RT], wh[U][CT], wl[U][CT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = i0 + u < mine ? i0 + u : mine - 1;  // past the end: reload the last one (unused)
                const size_t off = (size_t)(wave + 4 * i) * 64;
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    wh[u][t] = *reinterpret_cast<const f16x8*>(wp[t] + off);
                    wl[u][t] = *reinterpret_cast<const f16x8*>(wp[t] + off + 32);
                }
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    ah[u][t] = *reinterpret_cast<const f16x8*>(ap[t] + off);
                    al[u][t] = *reinterpret_cast<const f16x8*>(ap[t] + off + 32);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u < mine) {  // wave-uniform
#pragma unroll
                    for (int i = 0; i < RT; ++i)
#pragma unroll
                        for (int j = 0; j < CT; ++j) {
                            hh[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u][i], wh[u][j], hh[i][j], 0, 0, 0);
                            xx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u][i], wl[u][j], xx[i][j], 0, 0, 0);
                            xx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u][i], wh[u][j], xx[i][j], 0, 0, 0);
                        }
                }
            }
        }
    };
    if (RT * CT == 1 && kchunks > 16) run_groups(std::integral_constant<int, 12>{});
    else run_groups(std::integral_constant<int, (RT * CT == 1 ? 4 : 3)>{});
    // C/D layout of the 16x16 MFMA: n = lane & 15, m = 4 (lane >> 4) + r
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i][j][4 * g + r][l15] = fmaf(xx[i][j][r], kShLoInv, hh[i][j][r]);
    __syncthreads();
    const int m = em, n = en;
    bool ovf = false;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const uint32_t row = m0 + 16 * i + m;
        if (row >= M) break;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const uint32_t col = n0 + 16 * j + n;
            float v = (red[0][i][j][m][n] + red[1][i][j][m][n]) + (red[2][i][j][m][n] + red[3][i][j][m][n]) + e_bias[j];
            if (EPI == SH_OUT_F32 || EPI == SH_OUT_F32_RESID) {
                if (EPI == SH_OUT_F32_RESID) v += e_res[i][j];
                C[(size_t)row * N + col] = v;
            } else {
                if (EPI == SH_OUT_SPLIT_GELU) v = sh_gelu_erf(v);
                _Float16 hi, lo;
                ovf |= sh_split(v, hi, lo);
                _Float16* dst = Cs + ((size_t)row * (N / 32) + (col >> 5)) * 64 + (col & 31);
                dst[0] = hi;
                dst[32] = lo;
            }
        }
    }
    if (ovf && flag) atomicOr(flag, 1u);
}

// rows x K f32 -> split layout; one thread per 8 consecutive k.  With row_norm, row r is divided
// by row_norm[r] first (a zero norm gives a zero row): unit rows for the scan's filter operand.
__global__ void __launch_bounds__(256)
split_rows_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, uint64_t n8, uint32_t K,
                  const float* __restrict__ row_norm, uint32_t* __restrict__ flag) {
    bool ovf = false;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t k8 = K / 8;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint64_t row = i / k8;
        const uint32_t c = (uint32_t)(i % k8);  // 8-element group within the row
        sh_f32x4 v0 = *reinterpret_cast<const sh_f32x4*>(src + i * 8);
        sh_f32x4 v1 = *reinterpret_cast<const sh_f32x4*>(src + i * 8 + 4);
        if (row_norm) {
            const float nrm = row_norm[row];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = nrm == 0.0f ? 0.0f : v0[e] / nrm;
                v1[e] = nrm == 0.0f ? 0.0f : v1[e] / nrm;
            }
        }
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 a, b;
            ovf |= sh_split(v0[e], a, b); hi[e] = a; lo[e] = b;
            ovf |= sh_split(v1[e], a, b); hi[4 + e] = a; lo[4 + e] = b;
        }
        _Float16* d = dst + (row * (K / 32) + (c >> 2)) * 64 + (c & 3) * 8;
        *reinterpret_cast<f16x8*>(d) = hi;
        *reinterpret_cast<f16x8*>(d + 32) = lo;
    }
    if (ovf && flag) atomicOr(flag, 1u);
}

// a = 2^-24 (the smallest f16 subnormal) at k = 0 of every row, b = 1024: every output must be 2^-14
// (x2: both lane halves hold a k = 0 element of their own 8-k group).
__global__ void denorm_selftest_kernel(uint32_t* ok) {
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.0f; b[j] = (_Float16)0.0f; }
    a[0] = (_Float16)5.9604645e-08f;
    b[0] = (_Float16)1024.0f;
    sh_f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    bool good = true;
#pragma unroll
    for (int r = 0; r < 16; ++r) good &= c[r] == 2.0f * 1024.0f * 5.9604645e-08f;
    if (!good) atomicAnd(ok, 0u);
}

int32_t sh_denorm_selftest(bool* ok, hipStream_t s) {
    uint32_t* d = nullptr;
    uint32_t hv = 1;
    CS_HIP(hipMalloc(&d, sizeof(uint32_t)));
    hipError_t e = hipMemcpyAsync(d, &hv, sizeof hv, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(denorm_selftest_kernel, dim3(1), dim3(64), 0, s, d);
        e = hipMemcpyAsync(&hv, d, sizeof hv, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(CS_ERR_HIP, "f16 subnormal self-test failed to run: %s", hipGetErrorString(e));
    *ok = hv == 1;
    return CS_OK;
}

int32_t launch_split_rows(const float* d_src, _Float16* d_dst, uint64_t rows, uint32_t K, uint32_t* d_flag,
                          hipStream_t s, const float* d_row_norm) {
    if (K % 32) return fail(CS_ERR_UNSUPPORTED, "split-f16 layout needs K %% 32 == 0 (K = %u)", K);
    const uint64_t n8 = rows * (K / 8);
    if (n8 == 0) return CS_OK;
    const uint64_t want = (n8 + 255) / 256;
    hipLaunchKernelGGL(split_rows_kernel, dim3((uint32_t)(want < 8192 ? want : 8192)), dim3(256), 0, s, d_src,
                       d_dst, n8, K, d_row_norm, d_flag);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_gemm_split(int epi, const _Float16* A, const _Float16* W, const float* bias, const float* resid,
                          float* C, _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag,
                          hipStream_t s) {
    if (N % SH_BN || K % 32) return fail(CS_ERR_UNSUPPORTED, "split GEMM N=%u K=%u must be multiples of 128/32", N, K);
    if (M == 0) return CS_OK;
    const uint32_t kc = K / 32;
    static int skinny_max_m = -1;
    if (skinny_max_m < 0) {
        const char* e = cs_lab_env("CS_GEMM_SKINNY_MAX_M");  // 0 disables the small-M kernel
        skinny_max_m = e ? std::atoi(e) : 1100;
    }
    if ((int64_t)M <= skinny_max_m) {
#define CS_SKINNY(RT_, CT_)                                                                                                \
    do {                                                                                                                   \
        const dim3 gs(N / (16 * CT_), (M + 16 * RT_ - 1) / (16 * RT_));                                                    \
        if (epi == SH_OUT_F32) hipLaunchKernelGGL((gemm_sh_skinny_kernel<SH_OUT_F32, RT_, CT_>), gs, dim3(256), 0, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag); \
        else if (epi == SH_OUT_F32_RESID) hipLaunchKernelGGL((gemm_sh_skinny_kernel<SH_OUT_F32_RESID, RT_, CT_>), gs, dim3(256), 0, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag); \
        else if (epi == SH_OUT_SPLIT) hipLaunchKernelGGL((gemm_sh_skinny_kernel<SH_OUT_SPLIT, RT_, CT_>), gs, dim3(256), 0, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag); \
        else hipLaunchKernelGGL((gemm_sh_skinny_kernel<SH_OUT_SPLIT_GELU, RT_, CT_>), gs, dim3(256), 0, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag); \
    } while (0)
        // measured per forward (device us), tiled / 1x1 / 2x2 tiles per block: 144 rows 1096 / 520 / 559, 256 rows
        // 1152 / 748 / 660, 512 rows 1160 / 1096 / 711, 1024 rows 1209 / 1883 / 1072, 2048 rows 1248 / 3450 / 1601
        // (4 x 4 tiles per block: 1017 at 512 rows, 1297 at 1024: never the fastest)
        static int wide_from = -1;
        if (wide_from < 0) {
            const char* e = cs_lab_env("CS_GEMM_SKINNY_WIDE_M");  // rows from which a block owns 2 x 2 tiles
            wide_from = e ? std::atoi(e) : 200;
        }
        if ((int64_t)M < wide_from) CS_SKINNY(1, 1);
        else CS_SKINNY(2, 2);
#undef CS_SKINNY
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    {  // 128 x 128 tiles on the 16x16x32 MFMA
        static PerDeviceOnce attr16;  // function attributes are per device
        CS_TRY(attr16.run([&]() -> int32_t {
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sh16_kernel<SH_OUT_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sh16_kernel<SH_OUT_F32_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sh16_kernel<SH_OUT_SPLIT_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sh16_kernel<SH_OUT_SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
            return CS_OK;
        }));
        const dim3 grid16(sh_grid_blocks((M + SH_BM - 1) / SH_BM, N / SH_BN));
        if (epi == SH_OUT_F32) hipLaunchKernelGGL(gemm_sh16_kernel<SH_OUT_F32>, grid16, dim3(256), SH_LDS_BYTES, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag, 1u);
        else if (epi == SH_OUT_F32_RESID) hipLaunchKernelGGL(gemm_sh16_kernel<SH_OUT_F32_RESID>, grid16, dim3(256), SH_LDS_BYTES, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag, 1u);
        else if (epi == SH_OUT_SPLIT) hipLaunchKernelGGL(gemm_sh16_kernel<SH_OUT_SPLIT>, grid16, dim3(256), SH_LDS_BYTES, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag, 1u);
        else hipLaunchKernelGGL(gemm_sh16_kernel<SH_OUT_SPLIT_GELU>, grid16, dim3(256), SH_LDS_BYTES, s, A, W, bias, resid, C, Cs, M, N, kc, d_flag, 1u);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
}

// Split-K form of C = A W^T for layers whose K walk is what bounds them (FFN-down at a few thousand token
// rows: 48 stages per block, too few blocks to overlap them): `ksplit` K slices per output tile, raw f32
// partial tiles into Cpart[ksplit][M][N]; bias, residual and the sum are LayerNorm's (launch_row_kernel 3).
int32_t launch_gemm_split_partial(const _Float16* A, const _Float16* W, float* Cpart, uint32_t M, uint32_t N,
                                  uint32_t K, uint32_t ksplit, hipStream_t s) {
    if (N % SH_BN || K % 32 || ksplit == 0 || K / 32 < ksplit)
        return fail(CS_ERR_UNSUPPORTED, "split-K GEMM N=%u K=%u ksplit=%u", N, K, ksplit);
    if (M == 0) return CS_OK;
    static PerDeviceOnce attr;  // function attributes are per device
    CS_TRY(attr.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sh16_kernel<SH_OUT_PARTIAL>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
        return CS_OK;
    }));
    const dim3 grid(sh_grid_blocks((M + SH_BM - 1) / SH_BM, N / SH_BN) * ksplit);
    hipLaunchKernelGGL(gemm_sh16_kernel<SH_OUT_PARTIAL>, grid, dim3(256), SH_LDS_BYTES, s, A, W, nullptr, nullptr, Cpart,
                       nullptr, M, N, K / 32, nullptr, ksplit);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
