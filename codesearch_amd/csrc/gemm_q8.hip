// gemm_q8.hip — the encoder's dense layers as onnxruntime's dynamic quantiser rewrites them (SURVEY.md §8a E2/E4/E5/E6 for
// the registry's *Q models; the reference's DEFAULT model is one: ModelType::AllMiniLML6V2Q,
// /root/reference/src/embed/embedder.rs:12-13,367-372 — fastembed runs its model_quantized.onnx through ONNX Runtime).
// In those files every Linear is
//     x -> DynamicQuantizeLinear -> MatMulInteger(x_q, W_q, x_zp, W_zp) -> Cast(f32) -> Mul(x_scale * W_scale) -> Add(bias)
// with the published (ONNX opset 10 / 11) definitions
//     DynamicQuantizeLinear, per tensor, uint8:  lo = min(0, min x), hi = max(0, max x),  x_scale = (hi - lo) / 255 (1 if hi == lo),
//         x_zp = sat_u8(round_half_even(0 - lo / x_scale)),   x_q = sat_u8(round_half_even(x / x_scale) + x_zp)
//     MatMulInteger:  acc[m][n] = sum_k (x_q[m][k] - x_zp) * (W_q[k][n] - W_zp[n])   in int32, exact
// Here: a range pass (q8_minmax_kernel), a quantising pass (q8_quantize_kernel: the same f32 division and rounding, so the
// same bytes from the same activations), and the product on v_mfma_i32_16x16x64_i8 — exact, so the int32 accumulators ARE
// MatMulInteger's.  Both operands are kept as SIGNED bytes for the MFMA: a = x_q - 128, b = (W_q - W_zp) - c[n] with c[n]
// chosen per column so the 8-bit span fits [-128, 127]; with za = x_zp - 128, zw = -c[n]
//     sum_k (a - za)(b - zw)  =  sum_k a b  -  zw rowsum[m]  -  za colsum[n]  +  K za zw
// puts the zero points back in the epilogue (int32 throughout).  One f16 MFMA product of the split-f16 kernels (three
// MFMAs per 32 k) becomes one int8 MFMA per 64 k: a sixth of the matrix-pipe time, and a quarter of the operand bytes.
// The tile leaves through the epilogue the split-f16 kernels use (gemm_epilogue.hpp).
#include <cstdlib>
#include <utility>

#include "encoder.hpp"
#include "encoder_rows.hpp"
#include "gemm_epilogue.hpp"
#include "gemm_q8.hpp"
#include "gemm_q8_dev.hpp"
#include "split_f16.hpp"

namespace cs {

namespace {


// 16 bytes of the source -> up to 8 values.  f32 rows: unit u = 4 consecutive floats.  Split-f16 lines [rows][K/32][64]:
// unit u = piece u % 4 of line u / 4: 8 hi halves at +0, their 8 lo halves 64 B further on; x = hi + lo / 2048.
template <int SRC>
struct Q8Unit {
    static constexpr int N = SRC == Q8_SRC_F32 ? 4 : 8;
    float v[N];
    __device__ __forceinline__ void load(const void* src, uint64_t u) {
        if (SRC == Q8_SRC_F32) {
            const sh_f32x4 t = reinterpret_cast<const sh_f32x4*>(src)[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = t[e];
        } else {
            const _Float16* line = reinterpret_cast<const _Float16*>(src) + (u >> 2) * 64 + (u & 3) * 8;
            const f16x8 h = *reinterpret_cast<const f16x8*>(line);
            const f16x8 l = *reinterpret_cast<const f16x8*>(line + 32);
#pragma unroll
            for (int e = 0; e < N; ++e) v[e % N] = (float)h[e] + (float)l[e] * kShLoInv;
        }
    }
};

template <int SRC>
__global__ void __launch_bounds__(256)
q8_minmax_kernel(const void* __restrict__ src, uint32_t T, uint32_t K, uint32_t* __restrict__ range) {
    constexpr int UN = Q8Unit<SRC>::N;
    const uint32_t upr = K / UN;  // units per row
    const uint64_t units = (uint64_t)T * upr;
    // one unit (slot 0): four loads in flight per lane, the block's four waves meet in LDS, one update per block
    __shared__ float s_lo[4], s_hi[4];
    float lo = 0.0f, hi = 0.0f;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; u + 3 * stride < units; u += 4 * stride) {
        Q8Unit<SRC> x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i].load(src, u + i * stride);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < UN; ++e) { lo = fminf(lo, x[i].v[e]); hi = fmaxf(hi, x[i].v[e]); }
    }
    for (; u < units; u += stride) {
        Q8Unit<SRC> x;
        x.load(src, u);
#pragma unroll
        for (int e = 0; e < UN; ++e) { lo = fminf(lo, x.v[e]); hi = fmaxf(hi, x.v[e]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0)
        q8_range_update(range, fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3])),
                        fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3])));
}

// Several units in the tensor (row_slot): a block owns 32 consecutive rows, a wave eight of them one after the other —
// rows of one unit are contiguous, so a wave folds its rows into one (slot, lo, hi) and the block's four into mostly one
// update.  (A grid-stride walk as above would hand every lane rows of every unit: millions of range updates, 50 ms.)
template <int SRC>
__global__ void __launch_bounds__(256)
q8_minmax_units_kernel(const void* __restrict__ src, uint32_t T, uint32_t K, uint32_t* __restrict__ range,
                       const uint32_t* __restrict__ row_slot) {
    constexpr int UN = Q8Unit<SRC>::N;
    constexpr uint32_t NONE = 0xffffffffu;
    __shared__ uint32_t s_slot[4];
    __shared__ float s_lo[4], s_hi[4];
    const uint32_t upr = K / UN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t cur = NONE;
    float lo = 0.0f, hi = 0.0f;
    for (int i = 0; i < 8; ++i) {
        const uint32_t row = blockIdx.x * 32 + wave * 8 + i;
        if (row >= T) break;
        const uint32_t slot = row_slot[row];
        if (slot >> 31) continue;  // outside its call's own padded length: not part of the tensor the reference quantises
        float rlo = 0.0f, rhi = 0.0f;
        for (uint32_t uu = lane; uu < upr; uu += 64) {
            Q8Unit<SRC> x;
            x.load(src, (uint64_t)row * upr + uu);
#pragma unroll
            for (int e = 0; e < UN; ++e) { rlo = fminf(rlo, x.v[e]); rhi = fmaxf(rhi, x.v[e]); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            rlo = fminf(rlo, __shfl_xor(rlo, o));
            rhi = fmaxf(rhi, __shfl_xor(rhi, o));
        }
        if (slot != cur) {
            if (cur != NONE && lane == 0) q8_range_update(range + Q8_RANGE_WORDS * (size_t)cur, lo, hi);
            cur = slot;
            lo = rlo;
            hi = rhi;
        } else {
            lo = fminf(lo, rlo);
            hi = fmaxf(hi, rhi);
        }
    }
    if (lane == 0) { s_slot[wave] = cur; s_lo[wave] = lo; s_hi[wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < 4; ++w) {
            if (s_slot[w] == NONE) continue;
            float l = s_lo[w], h = s_hi[w];
            for (int v = w + 1; v < 4; ++v)
                if (s_slot[v] == s_slot[w]) { l = fminf(l, s_lo[v]); h = fmaxf(h, s_hi[v]); s_slot[v] = NONE; }
            q8_range_update(range + Q8_RANGE_WORDS * (size_t)s_slot[w], l, h);
        }
    }
}

// Quantisation units inside one device batch (several calls of the reference embedded together): row (b, t) belongs to
// unit seq_unit[b]; positions t >= unit_len[unit] lie outside that call's own padded length (bit 31: quantised, never ranged).
__global__ void __launch_bounds__(256)
q8_row_slot_kernel(const uint32_t* __restrict__ seq_unit, const uint32_t* __restrict__ unit_len, uint32_t T, uint32_t L,
                   uint32_t* __restrict__ row_slot) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= T) return;
    const uint32_t u = seq_unit[r / L];
    row_slot[r] = u | ((r % L) >= unit_len[u] ? 0x80000000u : 0u);
}

// The producers of a tensor (LayerNorm, attention) leave one (lo, hi) per block or wave: their reduction is the range pass.
// Several blocks, each folding a slice of the pairs and widening the slot with at most two atomics (the slot starts from
// (+0, +0): the forward zeroes every range slot before its first kernel) — one block of 1,024 threads took 7.3 us on
// 16-24 thousand pairs, three times per layer.
constexpr int Q8_RR_BLOCKS = 32;
__global__ void __launch_bounds__(256)
q8_range_reduce_kernel(const float* __restrict__ pairs, uint32_t n, uint32_t* __restrict__ slot) {
    __shared__ float s_lo[4], s_hi[4];
    float lo = 0.0f, hi = 0.0f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float2 p = reinterpret_cast<const float2*>(pairs)[i];
        lo = fminf(lo, p.x);
        hi = fmaxf(hi, p.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, s_lo[w]); hi = fmaxf(hi, s_hi[w]); }
        q8_range_update(slot, lo, hi);
    }
}

// The same with several units in the batch: the pairs lie sequence by sequence (pps per sequence), the sequences of a unit
// are consecutive; block u reduces unit u's.  rows != 0: a pair is a token ROW (pps = the batch's padded length) and only
// positions below the unit's own padded length belong to its tensor.
__global__ void __launch_bounds__(256)
q8_range_reduce_units_kernel(const float* __restrict__ pairs, uint32_t pps, uint32_t rows, const uint32_t* __restrict__ seq_unit,
                             const uint32_t* __restrict__ unit_len, uint32_t B, uint32_t* __restrict__ range) {
    __shared__ uint32_t s_first, s_end;
    __shared__ float s_lo[4], s_hi[4];
    const uint32_t u = blockIdx.x;
    if (threadIdx.x == 0) { s_first = B; s_end = 0; }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += 256)
        if (seq_unit[b] == u) { atomicMin(&s_first, b); atomicMax(&s_end, b + 1); }
    __syncthreads();
    const uint32_t first = s_first, end = s_end;
    float lo = 0.0f, hi = 0.0f;
    if (first < end) {
        const uint32_t own = rows ? unit_len[u] : pps;
        const uint64_t n = (uint64_t)(end - first) * pps;
        for (uint64_t i = threadIdx.x; i < n; i += 256) {
            if (rows && (uint32_t)(i % pps) >= own) continue;
            const float2 p = reinterpret_cast<const float2*>(pairs)[(uint64_t)first * pps + i];
            lo = fminf(lo, p.x);
            hi = fmaxf(hi, p.y);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t* slot = range + Q8_RANGE_WORDS * (size_t)u;
        slot[0] = __float_as_uint(fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3])));
        slot[1] = __float_as_uint(fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3])));
    }
}


// A block owns Q8_RB whole rows, so a row's sum of stored bytes meets in LDS.
constexpr int Q8_RB = 8;

template <int SRC>
__global__ void __launch_bounds__(256)
q8_quantize_kernel(const void* __restrict__ src, uint32_t T, uint32_t K, const uint32_t* __restrict__ range,
                   const uint32_t* __restrict__ row_slot, int8_t* __restrict__ xq, Q8RowMeta* __restrict__ rmeta) {
    constexpr int UN = Q8Unit<SRC>::N;
    __shared__ float r_xs[Q8_RB], r_xz[Q8_RB];
    __shared__ int r_sum[Q8_RB];
    const uint32_t row0 = blockIdx.x * Q8_RB;
    const uint32_t rows = T - row0 < (uint32_t)Q8_RB ? T - row0 : (uint32_t)Q8_RB;
    if (threadIdx.x < rows) {
        const uint32_t slot = row_slot ? (row_slot[row0 + threadIdx.x] & 0x7fffffffu) : 0u;
        float xs, xz;
        q8_params(range + Q8_RANGE_WORDS * (size_t)slot, xs, xz);
        r_xs[threadIdx.x] = xs;
        r_xz[threadIdx.x] = xz;
        r_sum[threadIdx.x] = 0;
    }
    __syncthreads();
    const uint32_t upr = K / UN;
    for (uint32_t c = threadIdx.x; c < rows * upr; c += 256) {
        const uint32_t r = c / upr, cc = c - r * upr;
        const uint64_t u = (uint64_t)(row0 + r) * upr + cc;
        Q8Unit<SRC> x;
        x.load(src, u);
        const float xs = r_xs[r], xz = r_xz[r];
        int sum = 0;
        uint32_t packed[UN / 4];
#pragma unroll
        for (int w = 0; w < UN / 4; ++w) {
            uint32_t p = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // x_q = sat_u8(round_half_even(x / x_scale) + x_zp); stored minus 128
                const float q = fminf(fmaxf(__fadd_rn(rintf(__fdiv_rn(x.v[4 * w + e], xs)), xz), 0.0f), 255.0f);
                const int a = (int)q - 128;
                sum += a;
                p |= (uint32_t)(a & 0xff) << (8 * e);
            }
            packed[w] = p;
        }
        // position of these UN bytes in the row: f32 source, unit cc = k 4cc..; split source, line cc / 4 (32 k), piece cc % 4 (8 k)
        const size_t kbyte = SRC == Q8_SRC_F32 ? (size_t)cc * 4 : (size_t)(cc >> 2) * 32 + (cc & 3) * 8;
        uint32_t* dst = reinterpret_cast<uint32_t*>(xq + (size_t)(row0 + r) * K + kbyte);
#pragma unroll
        for (int w = 0; w < UN / 4; ++w) dst[w] = packed[w];
        atomicAdd(&r_sum[r], sum);
    }
    __syncthreads();
    if (threadIdx.x < rows) {
        Q8RowMeta m;
        m.xs = r_xs[threadIdx.x];
        m.za = (int)r_xz[threadIdx.x] - 128;
        m.rowsum = r_sum[threadIdx.x];
        m.pad = 0;
        rmeta[row0 + threadIdx.x] = m;
    }
}

// One wave per weight row n: the integers d = round(w / scale) (= W_q - W_zp of the file), re-centred so they fit s8.
__global__ void __launch_bounds__(64)
q8_pack_weight_kernel(const float* __restrict__ W, const float* __restrict__ scale, const float* __restrict__ bias,
                      uint32_t N, uint32_t K, int8_t* __restrict__ wq, Q8ColMeta* __restrict__ cmeta,
                      uint32_t* __restrict__ bad) {
    const uint32_t n = blockIdx.x;
    if (n >= N) return;
    const int lane = threadIdx.x;
    const float sc = scale[n];
    const float* w = W + (size_t)n * K;
    int dmin = 0x7fffffff, dmax = -0x7fffffff;
    bool off_grid = false;
    for (uint32_t k = lane; k < K; k += 64) {
        const float t = sc != 0.0f ? __fdiv_rn(w[k], sc) : 0.0f;
        const float d = rintf(t);
        off_grid |= !(fabsf(t - d) <= 0.01f) || (sc == 0.0f && w[k] != 0.0f);
        const int di = (int)fminf(fmaxf(d, -1.0e6f), 1.0e6f);
        dmin = di < dmin ? di : dmin;
        dmax = di > dmax ? di : dmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int a = __shfl_xor(dmin, o), b = __shfl_xor(dmax, o);
        dmin = a < dmin ? a : dmin;
        dmax = b > dmax ? b : dmax;
    }
    if (__any(off_grid) && lane == 0) atomicOr(bad, 2u);
    if (dmax - dmin > 255) {
        if (lane == 0) atomicOr(bad, 1u);
        return;
    }
    const int c = dmin + 128;  // b = d - c lies in [-128, 127]
    int sum = 0;
    for (uint32_t k = lane; k < K; k += 64) {
        const int d = sc != 0.0f ? (int)rintf(__fdiv_rn(w[k], sc)) : 0;
        const int b = d - c;
        wq[(size_t)n * K + k] = (int8_t)b;
        sum += b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) {
        Q8ColMeta m;
        m.ws = sc;
        m.zw = -c;
        m.colsum = sum;
        m.bias = bias ? bias[n] : 0.0f;
        cmeta[n] = m;
    }
}

// ---- the product ----------------------------------------------------------------------------------------------------
// 128 x 128 tiles, four waves as 2 x 2 (a wave owns 64 x 64 = 4 x 4 MFMA tiles of 16 x 16), two blocks per CU; stage =
// one 128-B line (128 k) of 128 activation rows and 128 weight rows by LDS-DMA into the swizzled image of split_f16.hpp
// (sh_mainloop16: same staging, same fragment addresses — a line's slots 0-3 are the first v_mfma_i32_16x16x64_i8 k-step,
// slots 4-7 the second, where the split-f16 line holds its hi and lo planes).
// Two more ways out of the tile, for a layer whose output is GELU'd and then quantised again (FFN-up -> FFN-down): the
// int8 product is a sixth of the split-f16 one on the matrix pipe, so computing it TWICE is cheaper than carrying the
// f32-class GELU tensor through HBM (403 MB out, 403 MB back through a range pass, 403 MB back through a quantising
// pass at 65,536 rows): pass 1 (Q8_EPI_GELU_RANGE) stores nothing and only widens the output tensor's range; pass 2
// (Q8_EPI_GELU_Q8) recomputes the same values — same instruction sequence, same bits — and stores them already
// quantised with that range, with their row sums: the next layer's operand (100 MB).
static uint32_t gelu_table_on_env() {
    const char* e = cs_lab_env("CS_Q8_GELU_TABLE");  // (read per launch: A/B scripts and tests flip it mid-process)
    return !(e && e[0] == '0');
}

template <int EPI>
__global__ void __launch_bounds__(256, 2)
gemm_q8_kernel(const int8_t* __restrict__ A, const int8_t* __restrict__ W, const Q8RowMeta* __restrict__ rmeta,
               const Q8ColMeta* __restrict__ cmeta, const float* __restrict__ bias, const float* resid, float* C,
               _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* __restrict__ flag,
               int32_t* __restrict__ acc_dbg, Q8Requant rq, uint32_t total_slots) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t kchunks = K / 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int swz = (l15 >> 1) & 7;
    const int arow = (wr * 64 + l15) * 128, wrow = SH_TILE_BYTES + (wc * 64 + l15) * 128;
    const int s0 = (g ^ swz) * 16, s1 = ((4 + g) ^ swz) * 16;
    const int Ki = (int)K;
    // FFN-up passes: the output tensor's parameters (store pass) / this block's extremes of y over all its tiles (range pass)
    float gs = 1.0f, gz = 0.0f, rgs = 1.0f;
    if (EPI == Q8_EPI_GELU_Q8) {
        q8_params_gelu(rq.range, gs, gz);
        rgs = __fdiv_rn(1.0f, gs);
    }
    float ymax = -INFINITY, ya = -INFINITY, yb = INFINITY;  // max y, largest y <= c, smallest y >= c (q8_params_gelu)
    // Persistent: a block walks tile slots b, b + grid, ... (a multiple of 8 apart: the same XCD, sh_tile_of_block) — a
    // tile is three to twelve k-steps and a short epilogue, too little to pay a workgroup launch for.
    for (uint32_t slot = blockIdx.x; slot < total_slots; slot += gridDim.x) {
    uint32_t mt, nt;
    if (!sh_tile_of_block(slot, (M + SH_BM - 1) / SH_BM, N / SH_BN, mt, nt)) continue;
    const uint32_t m0 = mt * SH_BM, n0 = nt * SH_BN;
    const int8_t* asrc[4];
    const int8_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const uint32_t am = (m0 + row < M) ? m0 + row : M - 1;  // rows past M re-read the last one: never stored
        asrc[i] = A + (size_t)am * K + c * 16;
        wsrc[i] = W + (size_t)(n0 + row) * K + c * 16;
    }
    auto stage = [&](uint32_t kc, char* buf) {
        char* dst = buf + wave * 32 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sh_glds16(asrc[i] + (size_t)kc * 128, dst + i * 1024);
            sh_glds16(wsrc[i] + (size_t)kc * 128, dst + SH_TILE_BYTES + i * 1024);
        }
    };
    q8_i32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = q8_i32x4{0, 0, 0, 0};

    uint32_t kr = sh_kc_rot(nt, N / SH_BN, kchunks);
    auto next_chunk = [&]() { const uint32_t c = kr; kr = kr + 1 == kchunks ? 0 : kr + 1; return c; };
    stage(next_chunk(), lds);
    __syncthreads();
    for (uint32_t kc = 0; kc < kchunks; ++kc) {
        char* cur = lds + (kc & 1) * SH_STAGE_BYTES;
        if (kc + 1 < kchunks) stage(next_chunk(), lds + ((kc + 1) & 1) * SH_STAGE_BYTES);
        q8_i32x4 a0[4], a1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[i] = *reinterpret_cast<const q8_i32x4*>(cur + arow + i * 16 * 128 + s0);
            a1[i] = *reinterpret_cast<const q8_i32x4*>(cur + arow + i * 16 * 128 + s1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const q8_i32x4 w0 = *reinterpret_cast<const q8_i32x4*>(cur + wrow + j * 16 * 128 + s0);
            const q8_i32x4 w1 = *reinterpret_cast<const q8_i32x4*>(cur + wrow + j * 16 * 128 + s1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[i], w0, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[i], w1, acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();  // stage kc+1 has landed (vmcnt(0) precedes the barrier); cur is free
    }

    Q8ColMeta cm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cm[j] = cmeta[n0 + wc * 64 + j * 16 + l15];
    if constexpr (EPI == Q8_EPI_GELU_RANGE || EPI == Q8_EPI_GELU_Q8) {
        float bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bj[j] = bias[n0 + wc * 64 + j * 16 + l15];
        int8_t* tile8 = reinterpret_cast<int8_t*>(lds);  // [128 m][128 n] s8 (the stage buffers are free)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = wr * 64 + i * 16 + 4 * g + r;
                const Q8RowMeta rm = rmeta[(m0 + m < M) ? m0 + m : M - 1];
                const int pm = rm.rowsum - Ki * rm.za;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int corr = acc[i][j][r] - __mul24(cm[j].zw, pm) - __mul24(rm.za, cm[j].colsum);
                    const float y = __fadd_rn(__fmul_rn((float)corr, __fmul_rn(rm.xs, cm[j].ws)), bj[j]);
                    if (EPI == Q8_EPI_GELU_RANGE) {
                        ymax = fmaxf(ymax, y);
                        ya = y <= kGeluArgMin ? fmaxf(ya, y) : ya;
                        yb = y >= kGeluArgMin ? fminf(yb, y) : yb;
                    } else {
                        // sat_u8(round_half_even(v / scale) + zp): the quotient by a reciprocal, the true division only
                        // where the two could round apart (|v / scale| <= ~255: they differ by < 1e-4)
                        const float v = sh_gelu_erf(y);
                        const float t = v * rgs;
                        float rt = rintf(t);
                        if (fabsf(fabsf(t - rt) - 0.5f) < 1.0e-3f) rt = rintf(__fdiv_rn(v, gs));
                        const float q = fminf(fmaxf(__fadd_rn(rt, gz), 0.0f), 255.0f);
                        tile8[m * 128 + wc * 64 + j * 16 + l15] = (int8_t)((int)q - 128);
                    }
                }
            }
        if (EPI == Q8_EPI_GELU_RANGE) {
            // (rows past M repeat row M - 1: the same values once more, harmless in a range; reduced after the last tile)
            if (nt == 0 && tid < SH_BM && m0 + tid < M) rq.rmeta_out[m0 + tid].rowsum = 0;
        } else {
            __syncthreads();
            // 16 B per lane on consecutive lanes: a tile row is 8 lanes; their sum of bytes goes to the row's total
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = (tid >> 3) + 32 * pass, c16 = tid & 7;
                const q8_i32x4 v = *reinterpret_cast<const q8_i32x4*>(tile8 + row * 128 + c16 * 16);
                int sum = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum += (int)(int8_t)(((uint32_t)v[w] >> (8 * e)) & 0xffu);
                sum += __shfl_xor(sum, 1);
                sum += __shfl_xor(sum, 2);
                sum += __shfl_xor(sum, 4);
                if (m0 + row < M) {
                    *reinterpret_cast<q8_i32x4*>(rq.out + (size_t)(m0 + row) * N + n0 + c16 * 16) = v;
                    if (c16 == 0) atomicAdd(&rq.rmeta_out[m0 + row].rowsum, sum);
                }
            }
            if (nt == 0 && tid < SH_BM && m0 + tid < M) {
                rq.rmeta_out[m0 + tid].xs = gs;
                rq.rmeta_out[m0 + tid].za = (int)gz - 128;
            }
            __syncthreads();  // tile8 is read: the next tile may stage over it
        }
        continue;
    }
    // zero points back in, Cast + Mul(x_scale * W_scale): the tile as f32 into LDS (the stage buffers are free)
    float* ctile = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = wr * 64 + i * 16 + 4 * g + r;
            const Q8RowMeta rm = rmeta[(m0 + m < M) ? m0 + m : M - 1];
            const int pm = rm.rowsum - Ki * rm.za;  // -zw rowsum - za colsum + K za zw = -zw (rowsum - K za) - za colsum
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (24-bit multiplies, full rate: |pm| <= 2^8 K, |colsum| <= 2^7 K, K <= 4,096 at create)
                const int corr = acc[i][j][r] - __mul24(cm[j].zw, pm) - __mul24(rm.za, cm[j].colsum);
                ctile[m * 128 + wc * 64 + j * 16 + l15] = __fmul_rn((float)corr, __fmul_rn(rm.xs, cm[j].ws));
                if (acc_dbg && m0 + m < M)  // cs_debug_gemm_q8: the MatMulInteger result itself
                    acc_dbg[(size_t)(m0 + m) * N + n0 + wc * 64 + j * 16 + l15] = corr;
            }
        }
    __syncthreads();
    constexpr int SEPI = (EPI == Q8_EPI_GELU_RANGE || EPI == Q8_EPI_GELU_Q8) ? SH_OUT_F32 : EPI;  // (never reached for those)
    if (m0 + SH_BM <= M) gemm_sh_epilogue<SEPI, true, 2>(ctile, bias, resid, C, Cs, M, N, m0, n0, flag);
    else gemm_sh_epilogue<SEPI, false, 2>(ctile, bias, resid, C, Cs, M, N, m0, n0, flag);
    __syncthreads();  // ctile is read: the next tile may stage over it
    }  // tile slots
    if (EPI == Q8_EPI_GELU_RANGE) {
        __shared__ float s_r[3][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ymax = fmaxf(ymax, __shfl_xor(ymax, o));
            ya = fmaxf(ya, __shfl_xor(ya, o));
            yb = fminf(yb, __shfl_xor(yb, o));
        }
        if (lane == 0) { s_r[0][wave] = ymax; s_r[1][wave] = ya; s_r[2][wave] = yb; }
        __syncthreads();
        if (tid == 0) {
            const float m3 = fmaxf(fmaxf(s_r[0][0], s_r[0][1]), fmaxf(s_r[0][2], s_r[0][3]));
            const float a3 = fmaxf(fmaxf(s_r[1][0], s_r[1][1]), fmaxf(s_r[1][2], s_r[1][3]));
            const float b3 = fminf(fminf(s_r[2][0], s_r[2][1]), fminf(s_r[2][2], s_r[2][3]));
            if (m3 > -INFINITY) q8_key_update(rq.range + 2, m3);
            if (a3 > -INFINITY) q8_key_update(rq.range + 3, a3);
            if (b3 < INFINITY) q8_key_update(rq.range + 4, -b3);
        }
    }
}


// ---- a few token rows (query-side forwards): one launch per Linear --------------------------------------------------------
// A one-query forward is latency-bound: range reduction, quantising pass and product are three dependent launches per
// Linear, seventeen per layer (0.55 ms against 0.20 for the split-f16 path's skinny kernels; 0.25 with this kernel).  Here a block owns ONE
// 16 x 16 output tile (as gemm_sh_skinny_kernel): it reduces the (lo, hi) pairs its input's producer left, quantises
// its 16 activation rows over all of K into LDS — the same division and rounding, so the same bytes as
// q8_quantize_kernel — and its four waves split K between them, weights global -> VGPR in MFMA operand order, every load of
// a wave in flight at once; the four partial tiles meet in LDS.  With the FFN-up launch leaving the (lo, hi) of what each
// block stored, a layer is seven launches.  One quantisation unit only.
// WIDE (many row tiles: a query and its variants, N >= 1024): a block = 16 rows x 64 columns, a wave = one column tile over all of K —
// a quarter of the blocks redo the range reduction and the quantising pass of the same 16 rows.  Integer sums: the same bits.
template <int EPI, int SRC, bool WIDE = false>
__global__ void __launch_bounds__(256)
gemm_q8_skinny_kernel(const void* __restrict__ src, const float* __restrict__ range_pairs, uint32_t n_pairs,
                      const int8_t* __restrict__ W, const Q8ColMeta* __restrict__ cmeta, const float* resid, float* C,
                      _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* __restrict__ flag,
                      float* __restrict__ range_out, const float* __restrict__ ln_g = nullptr, const float* __restrict__ ln_b = nullptr,
                      float ln_eps = 0.0f, float* __restrict__ ln_xout = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [16][K + 16] s8: the block's quantised rows
    __shared__ float s_lo[4], s_hi[4];
    __shared__ int s_rowsum[16];
    __shared__ int red[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t n0 = WIDE ? blockIdx.x * 64 + wave * 16 : blockIdx.x * 16, m0 = blockIdx.y * 16;
    const uint32_t astride = K + 16;
    // this wave's weight fragments first: k-steps wave, wave + 4, ... of 64 — WIDE: every k-step — (they do not depend on the activations)
    const uint32_t ksteps = K / 64;
    const uint32_t mine = WIDE ? ksteps : (ksteps > (uint32_t)wave ? (ksteps - wave + 3) / 4 : 0);
    const uint32_t kfirst = WIDE ? 0u : (uint32_t)wave, kstride = WIDE ? 1u : 4u;
    const int8_t* wp = W + (size_t)(n0 + l15) * K + g * 16;
    constexpr int U = 6;  // K = 1536: six steps per wave; beyond that the loop runs again
    q8_i32x4 wf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t i = (uint32_t)u < mine ? u : (mine ? mine - 1 : 0);
        wf[u] = mine ? *reinterpret_cast<const q8_i32x4*>(wp + (size_t)(kfirst + kstride * i) * 64) : q8_i32x4{0, 0, 0, 0};
    }
    // the epilogue's operands and the block's raw activation rows are requested NOW, with the weights and the pairs: the range, the
    // quantising pass and the epilogue used to wait for memory one after the other (three round trips per launch, ~1.5 us each, in a
    // path that is seven dependent launches per layer)
    const int em = tid >> 4, en = tid & 15;
    constexpr int ET = WIDE ? 4 : 1;  // column tiles of the block: thread (em, en) finishes element (em, en) of each
    const uint32_t erow = m0 + em, ecol0 = (WIDE ? blockIdx.x * 64 : blockIdx.x * 16) + en;
    Q8ColMeta cmv[ET];
    float e_residv[ET];
#pragma unroll
    for (int t = 0; t < ET; ++t) {
        cmv[t] = cmeta[ecol0 + 16 * t];
        e_residv[t] = EPI == SH_OUT_F32_RESID ? resid[(size_t)(erow < M ? erow : M - 1) * N + ecol0 + 16 * t] : 0.0f;
    }
    // 16 rows x K / 16 slots of 16 k; slot sidx = tid + 256 s (s < 6: K <= 1536; beyond, the loop below reads the rest as before)
    const uint32_t spr = K / 16, nslots = 16 * spr;
    constexpr int PS = SRC == Q8_SRC_LN ? 1 : 6;
    sh_f32x4 raw[PS][4];   // Q8_SRC_F32: 16 floats | split source: (hi 0-7, hi 8-15, lo 0-7, lo 8-15) as 4 x 16 bytes
    // Q8_SRC_LN (K = 384, M <= 16: the block's rows are the whole tensor): the rows are normalised here — a wave takes four of them,
    // layernorm_kernel's arithmetic (ln_rows_core) — their range is the tensor's, the blocks of column tile 0 write them to ln_xout
    float lnv[4][6];
    if constexpr (SRC == Q8_SRC_LN) {
        float v[4][6];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const uint32_t t_raw = m0 + 4 * wave + rr, t = t_raw < M ? t_raw : M - 1;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float2 r2 = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(src) + (size_t)t * 384 + ln_col(lane, 2 * p));
                v[rr][2 * p] = r2.x;
                v[rr][2 * p + 1] = r2.y;
            }
        }
        ln_rows_core<6, 4>(v, ln_g, ln_b, ln_eps, lane, lnv);
    }
#pragma unroll
    for (int ps = 0; ps < (SRC == Q8_SRC_LN ? 0 : PS); ++ps) {
        const uint32_t sidx = tid + 256 * ps;
        if (sidx < nslots) {
            const uint32_t row = sidx / spr, sl = sidx - row * spr;
            const uint32_t m = (m0 + row < M) ? m0 + row : M - 1;  // rows past M repeat the last one: never stored
            if (SRC == Q8_SRC_F32) {
                const sh_f32x4* p = reinterpret_cast<const sh_f32x4*>(reinterpret_cast<const float*>(src) + (size_t)m * K + sl * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) raw[ps][q] = p[q];
            } else {  // 16 k = half a 32-k line: 16 hi halves, their 16 lo halves 64 B further on
                const _Float16* p = reinterpret_cast<const _Float16*>(src) + ((size_t)m * (K / 32) + (sl >> 1)) * 64 + (sl & 1) * 16;
                raw[ps][0] = *reinterpret_cast<const sh_f32x4*>(p);
                raw[ps][1] = *reinterpret_cast<const sh_f32x4*>(p + 8);
                raw[ps][2] = *reinterpret_cast<const sh_f32x4*>(p + 32);
                raw[ps][3] = *reinterpret_cast<const sh_f32x4*>(p + 40);
            }
        }
    }
    // the tensor's range from its producer's pairs
    float lo = 0.0f, hi = 0.0f;
    if constexpr (SRC == Q8_SRC_LN) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const uint32_t t = m0 + 4 * wave + rr;
#pragma unroll
            for (int i = 0; i < 6; ++i) {  // (rows past M are copies of row M - 1: they widen nothing)
                lo = fminf(lo, lnv[rr][i]);
                hi = fmaxf(hi, lnv[rr][i]);
            }
            if (blockIdx.x == 0 && t < M) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<float2*>(ln_xout + (size_t)t * 384 + ln_col(lane, 2 * p)) = make_float2(lnv[rr][2 * p], lnv[rr][2 * p + 1]);
            }
        }
    } else {
        for (uint32_t i = tid; i < n_pairs; i += 256) {
            const float2 p = reinterpret_cast<const float2*>(range_pairs)[i];
            lo = fminf(lo, p.x);
            hi = fmaxf(hi, p.y);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; }
    if (tid < 16) s_rowsum[tid] = 0;
    __syncthreads();
    float xs, xz;
    q8_params_of(fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3])), fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3])), xs, xz);
    const int za = (int)xz - 128;
    auto quantise_slot = [&](uint32_t sidx, const sh_f32x4 (&r4)[4]) {
        const uint32_t row = sidx / spr, sl = sidx - row * spr;
        float v[16];
        if (SRC == Q8_SRC_F32) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = r4[q][e];
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f16x8 h = __builtin_bit_cast(f16x8, r4[q]);
                const f16x8 l = __builtin_bit_cast(f16x8, r4[2 + q]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * q + e] = (float)h[e] + (float)l[e] * kShLoInv;
            }
        }
        q8_i32x4 packed;
        int sum = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t pw = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = fminf(fmaxf(__fadd_rn(rintf(__fdiv_rn(v[4 * w + e], xs)), xz), 0.0f), 255.0f);
                const int a = (int)q - 128;
                sum += a;
                pw |= (uint32_t)(a & 0xff) << (8 * e);
            }
            packed[w] = (int)pw;
        }
        *reinterpret_cast<q8_i32x4*>(lds + row * astride + sl * 16) = packed;
        atomicAdd(&s_rowsum[row], sum);
    };
    if constexpr (SRC == Q8_SRC_LN) {  // the normalised rows from registers: columns 2 lane + 128 p + {0, 1} of row 4 wave + rr
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * wave + rr;
            int sum = 0;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                uint32_t pw = 0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float q = fminf(fmaxf(__fadd_rn(rintf(__fdiv_rn(lnv[rr][2 * p + e], xs)), xz), 0.0f), 255.0f);
                    const int a = (int)q - 128;
                    sum += a;
                    pw |= (uint32_t)(a & 0xff) << (8 * e);
                }
                *reinterpret_cast<uint16_t*>(lds + row * astride + ln_col(lane, 2 * p)) = (uint16_t)pw;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if (lane == 0) s_rowsum[row] = sum;
        }
    }
#pragma unroll
    for (int ps = 0; ps < (SRC == Q8_SRC_LN ? 0 : PS); ++ps)
        if (tid + 256 * ps < nslots) quantise_slot(tid + 256 * ps, raw[ps]);
    for (uint32_t sidx = tid + 256 * PS; SRC != Q8_SRC_LN && sidx < nslots; sidx += 256) {  // (K > 1536)
        const uint32_t row = sidx / spr, sl = sidx - row * spr;
        const uint32_t m = (m0 + row < M) ? m0 + row : M - 1;
        sh_f32x4 r4[4];
        if (SRC == Q8_SRC_F32) {
            const sh_f32x4* p = reinterpret_cast<const sh_f32x4*>(reinterpret_cast<const float*>(src) + (size_t)m * K + sl * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) r4[q] = p[q];
        } else {
            const _Float16* p = reinterpret_cast<const _Float16*>(src) + ((size_t)m * (K / 32) + (sl >> 1)) * 64 + (sl & 1) * 16;
            r4[0] = *reinterpret_cast<const sh_f32x4*>(p);
            r4[1] = *reinterpret_cast<const sh_f32x4*>(p + 8);
            r4[2] = *reinterpret_cast<const sh_f32x4*>(p + 32);
            r4[3] = *reinterpret_cast<const sh_f32x4*>(p + 40);
        }
        quantise_slot(sidx, r4);
    }
    __syncthreads();
    q8_i32x4 acc = {0, 0, 0, 0};
    for (uint32_t i0 = 0; i0 < mine; i0 += U) {
        if (i0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = i0 + u < mine ? i0 + u : mine - 1;
                wf[u] = *reinterpret_cast<const q8_i32x4*>(wp + (size_t)(kfirst + kstride * i) * 64);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u < mine) {  // wave-uniform
                const q8_i32x4 a = *reinterpret_cast<const q8_i32x4*>(lds + l15 * astride + (size_t)(kfirst + kstride * (i0 + u)) * 64 + g * 16);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, wf[u], acc, 0, 0, 0);
            }
        }
    }
    // C/D layout of the 16x16 MFMA: n = lane & 15, m = 4 (lane >> 4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * g + r][l15] = acc[r];
    __syncthreads();
    const int m = em, n = en;
    const uint32_t row = erow;
    float rlo = 0.0f, rhi = 0.0f;
    bool ovf = false;
#pragma unroll
    for (int t = 0; t < ET; ++t) {
        const uint32_t col = ecol0 + 16 * t;
        const Q8ColMeta cm = cmv[t];
        const int total = WIDE ? red[t][m][n] : (red[0][m][n] + red[1][m][n]) + (red[2][m][n] + red[3][m][n]);
        const int corr = total - __mul24(cm.zw, s_rowsum[m] - (int)K * za) - __mul24(za, cm.colsum);
        float v = __fadd_rn(__fmul_rn((float)corr, __fmul_rn(xs, cm.ws)), cm.bias);
        if (EPI == SH_OUT_F32 || EPI == SH_OUT_F32_RESID) {
            if (row < M) {
                if (EPI == SH_OUT_F32_RESID) v += e_residv[t];
                C[(size_t)row * N + col] = v;
            }
        } else {
            if (EPI == SH_OUT_SPLIT_GELU) v = sh_gelu_erf(v);
            _Float16 h16, l16;
            ovf |= sh_split(v, h16, l16);
            if (row < M) {
                _Float16* dst = Cs + ((size_t)row * (N / 32) + (col >> 5)) * 64 + (col & 31);
                dst[0] = h16;
                dst[32] = l16;
                const float stored = fmaf((float)l16, kShLoInv, (float)h16);  // what the next Linear's quantiser will read
                rlo = fminf(rlo, stored);
                rhi = fmaxf(rhi, stored);
            }
        }
    }
    if (EPI != SH_OUT_F32 && EPI != SH_OUT_F32_RESID && ovf && flag) atomicOr(flag, 1u);
    if (range_out) {  // this block's (lo, hi) of what it stored, for the Linear that follows
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            rlo = fminf(rlo, __shfl_xor(rlo, o));
            rhi = fmaxf(rhi, __shfl_xor(rhi, o));
        }
        __syncthreads();
        if (lane == 0) { s_lo[wave] = rlo; s_hi[wave] = rhi; }
        __syncthreads();
        if (tid == 0) {
            const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            range_out[2 * b] = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
            range_out[2 * b + 1] = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
        }
    }
}


// ---- K = 384 (hidden 384: MiniLM, BGE-small): activations in registers, weights streamed a whole n-tile ahead ------------
// In gemm_q8_kernel a 128 x 128 tile with K = 384 is three k-steps, each waiting a full L2 round trip for the next stage
// with two blocks per CU to cover it: 9 us per tile where the MFMAs are 0.8 (profiles/r04_q8_minilm_l6_kernel_stats.csv).
// Here a block of eight waves owns a 128-row block of the activations and walks n-tiles: its A fragments — 64 rows x 384
// k per wave = 96 registers — are loaded once per row block straight from global memory in MFMA operand order, and only
// weights pass through LDS: one n-tile of W with ALL of K is 48 KiB, double-buffered, so tile t + 1 lands under the whole
// of tile t (its MFMAs and its epilogue) and no k-step waits on memory; the tile's column metadata (with the bias in it)
// rides along as two more DMA instructions, the row block's metadata sits in LDS too, so the loop issues no ordinary
// load at all.  Wave tile 64 x 32 (waves as 2 x 4); the C tile leaves in two 64-row halves through 32 KiB of LDS (or as
// the 16 KiB s8 tile of the re-quantising pass).  One block per CU (134 KiB of LDS, <= 256 registers at two waves per SIMD).
constexpr int QR_KC = 3;
constexpr int QR_WTILE = 128 * 128 * QR_KC;          // one n-tile of weights over all of K
constexpr int QR_OUT_BYTES = 64 * 128 * 4;           // half a C tile in f32
constexpr int QR_CM_BYTES = 128 * 16;                // an n-tile's column metadata (scale, zero point, column sum, bias)
constexpr int QR_RM_BYTES = 128 * 16;                // the row block's metadata
constexpr int QR_LDS = 2 * QR_WTILE + 2 * QR_CM_BYTES + QR_RM_BYTES + QR_OUT_BYTES;  // 137,216
constexpr int QR_THREADS = 512;

// rows [mrow0, mrow0 + 64) x columns [n0, n0 + 128) from ctile [64][128] f32 (bias already in), 512 threads
template <int EPI, bool FULL>
__device__ __forceinline__ void q8_rows_store(const float* ctile, const float* resid, float* C, _Float16* __restrict__ Cs,
                                              uint32_t M, uint32_t N, uint32_t mrow0, uint32_t n0, uint32_t* __restrict__ flag) {
    const int tid = threadIdx.x;
    if (EPI == SH_OUT_SPLIT || EPI == SH_OUT_SPLIT_GELU) {
        bool ovf = false;
        const int c8 = tid & 15;
        const size_t nchunks = N / 32;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = (tid >> 4) + 32 * pass;
            const sh_f32x4 v0 = *reinterpret_cast<const sh_f32x4*>(ctile + row * 128 + c8 * 8);
            const sh_f32x4 v1 = *reinterpret_cast<const sh_f32x4*>(ctile + row * 128 + c8 * 8 + 4);
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, b;
                ovf |= sh_split(EPI == SH_OUT_SPLIT_GELU ? sh_gelu_erf(v0[e]) : v0[e], a, b);
                hi[e] = a; lo[e] = b;
                ovf |= sh_split(EPI == SH_OUT_SPLIT_GELU ? sh_gelu_erf(v1[e]) : v1[e], a, b);
                hi[4 + e] = a; lo[4 + e] = b;
            }
            if (FULL || mrow0 + row < M) {
                _Float16* dst = Cs + ((size_t)(mrow0 + row) * nchunks + (n0 >> 5) + (c8 >> 2)) * 64 + (c8 & 3) * 8;
                __builtin_nontemporal_store(hi, reinterpret_cast<f16x8*>(dst));
                __builtin_nontemporal_store(lo, reinterpret_cast<f16x8*>(dst + 32));
            }
        }
        if (ovf && flag) atomicOr(flag, 1u);
    } else {
        const int c4 = tid & 31;
        sh_f32x4 rs[4];
        if (EPI == SH_OUT_F32_RESID) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const uint32_t row = mrow0 + (tid >> 5) + 16 * pass;
                rs[pass] = *reinterpret_cast<const sh_f32x4*>(resid + (size_t)((FULL || row < M) ? row : M - 1) * N + n0 + c4 * 4);
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = (tid >> 5) + 16 * pass;
            sh_f32x4 v = *reinterpret_cast<const sh_f32x4*>(ctile + row * 128 + c4 * 4);
            if (EPI == SH_OUT_F32_RESID) v += rs[pass];
            if (FULL || mrow0 + row < M) *reinterpret_cast<sh_f32x4*>(C + (size_t)(mrow0 + row) * N + n0 + c4 * 4) = v;
        }
    }
}

// SRC = QR_PREQUANT: A is the quantised tensor (q8_quantize_kernel's output) with its row metadata.  SRC = Q8_SRC_F32 /
// Q8_SRC_SPLIT: A is the f32-class tensor itself and in_range its range slot — the block quantises its own 128 rows on the
// way in (cooperatively, into the W buffer the first tile does not use yet; every wave then takes its fragments from that
// image): the quantising pass over the tensor, its 25 MB of output and their re-read disappear (25-28 us per Linear at
// 65,536 rows).
// MU (with SRC >= 0): the tensor holds SEVERAL quantisation units (queued calls sharing a device batch): row_slot [M]
// names each row's unit (bit 31: the row lies beyond its call's own padded length — quantised with the unit's
// parameters, never part of a range), in_range and rq.range are the units' slot arrays.  Every row is quantised with its
// own unit's parameters; the FFN-up range pass keeps ONE unit's extremes per wave at a time and hands them to that unit's
// slot whenever the rows it walks change unit.  Rows that are not the current unit's are marked in the LDS row metadata
// (x_scale = NaN: their y drops out of every max, min and comparison), so a row block of one unit — the usual case — runs
// the one-unit code per element.
constexpr int QR_PREQUANT = -1;
struct Q8RowOut { float gs, gz, rgs; uint32_t slot; };  // MU: a row's output parameters (FFN-up store pass) and slot word
constexpr int QR_LDS_MU = QR_LDS + 128 * 16;
template <int EPI, int SRC = QR_PREQUANT, bool MU = false>
__global__ void __launch_bounds__(QR_THREADS, 2)
gemm_q8_rows_kernel(const void* __restrict__ Asrc, const int8_t* __restrict__ W, const Q8RowMeta* __restrict__ rmeta,
                    const Q8ColMeta* __restrict__ cmeta, const float* resid, float* C,
                    _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t* __restrict__ flag, Q8Requant rq,
                    uint32_t parts, uint32_t total_units, const uint32_t* __restrict__ in_range,
                    const uint32_t* __restrict__ row_slot) {
    static_assert(!MU || SRC != QR_PREQUANT, "several units: quantise-on-load only");
    const int8_t* A = reinterpret_cast<const int8_t*>(Asrc);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr uint32_t K = 128 * QR_KC;
    constexpr bool REQ = EPI == Q8_EPI_GELU_RANGE || EPI == Q8_EPI_GELU_Q8;
    char* cmbuf = lds + 2 * QR_WTILE;                          // [2][128] Q8ColMeta
    Q8RowMeta* lrow = reinterpret_cast<Q8RowMeta*>(cmbuf + 2 * QR_CM_BYTES);  // [128]: xs, -za, rowsum - K za
    char* obuf = cmbuf + 2 * QR_CM_BYTES + QR_RM_BYTES;
    Q8RowOut* lout = reinterpret_cast<Q8RowOut*>(lds + QR_LDS);  // [128], MU only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    const int swz = (l15 >> 1) & 7;
    const int s0 = (g ^ swz) * 16, s1 = ((4 + g) ^ swz) * 16;
    const int wrow = (wc * 32 + l15) * 128;
    const uint32_t ntiles = N / 128, per = (ntiles + parts - 1) / parts;
    // this wave's six LDS-DMA instructions of a W tile (of 48: chunk q / 16, rows 8 (q % 16) ..): per-lane source offsets
    uint32_t woff[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int q = wave * 6 + t, c = q >> 4, row = (q & 15) * 8 + (lane >> 3);
        woff[t] = (uint32_t)row * K + c * 128 + (((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    auto issue_w = [&](uint32_t nt, int b) {
        const int8_t* src = W + (size_t)nt * 128 * K;
        char* buf = lds + b * QR_WTILE;
#pragma unroll
        for (int t = 0; t < 6; ++t) sh_glds16(src + woff[t], buf + (wave * 6 + t) * 1024);
        if (wave < 2)  // 128 columns x 16 B of metadata: lane l of wave w moves column 64 w + l
            sh_glds16(reinterpret_cast<const char*>(cmeta + (size_t)nt * 128 + wave * 64) + lane * 16, cmbuf + b * QR_CM_BYTES + wave * 1024);
    };
    float gs = 1.0f, gz = 0.0f, rgs = 1.0f;
    // one unit's store pass: the output byte by table (q8_build_gelu_table); tb_inv_w == 0: the direct form
    Q8GeluEntry* gtbl = reinterpret_cast<Q8GeluEntry*>(lds + QR_LDS);
    float tb_inv_w = 0.0f, tb_c0 = 0.0f;
    if (EPI == Q8_EPI_GELU_Q8 && !MU) {
        q8_params_gelu(rq.range, gs, gz);
        rgs = __fdiv_rn(1.0f, gs);
        if (rq.use_table && rq.range[2]) {
            uint32_t* ok_bad = reinterpret_cast<uint32_t*>(lds);  // (W buffer 0: nothing has been issued into it yet)
            if (threadIdx.x == 0) *ok_bad = 0u;
            __syncthreads();
            tb_inv_w = q8_build_gelu_table(gtbl, ok_bad, gs, rgs, gz - 128.0f, q8_unkey(rq.range[2]), QR_THREADS);
            tb_c0 = -QG_YL * tb_inv_w;
            __syncthreads();  // (ok_bad has been read by every thread before the first W tile lands on it)
        }
    }
    float ymax = -INFINITY, ya = -INFINITY, yb = INFINITY;
    float ycen = 0.0f, yhw = INFINITY;  // wave-uniform: centre and half-width (padded) of the wave's (a, b) so far
    float xs = 1.0f, xz = 0.0f, rxs = 1.0f;  // SRC >= 0: DynamicQuantizeLinear's parameters of the input tensor
    if (SRC != QR_PREQUANT && !MU) {
        q8_params(in_range, xs, xz);
        rxs = __fdiv_rn(1.0f, xs);
    }
    // MU, range pass: the unit whose extremes (ymax, ya, yb) the wave is holding; they go to its slot when the unit changes
    constexpr uint32_t NO_SLOT = 0xffffffffu;
    uint32_t cur_slot = NO_SLOT;
    uint32_t blk_first = 0, blk_last = 0;
    // (block-uniform: every wave of the block is at the same unit — the eight meet in LDS and one thread updates the slot;
    // with an update per wave, 2,048 waves read the same two cache lines at the same moment: 137 us per launch instead of
    // 116 at 65,536 rows, eight units)
    auto flush_slot = [&]() {
        if (cur_slot != NO_SLOT) {
            float* s_r = reinterpret_cast<float*>(obuf);  // [3][8] (the range pass stages nothing in obuf)
            float m3 = ymax, a3 = ya, b3 = yb;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                m3 = fmaxf(m3, __shfl_xor(m3, o));
                a3 = fmaxf(a3, __shfl_xor(a3, o));
                b3 = fminf(b3, __shfl_xor(b3, o));
            }
            if (lane == 0) { s_r[wave] = m3; s_r[8 + wave] = a3; s_r[16 + wave] = b3; }
            __syncthreads();
            if (tid == 0) {
                m3 = -INFINITY; a3 = -INFINITY; b3 = INFINITY;
                for (int w = 0; w < 8; ++w) { m3 = fmaxf(m3, s_r[w]); a3 = fmaxf(a3, s_r[8 + w]); b3 = fminf(b3, s_r[16 + w]); }
                uint32_t* slot = rq.range + Q8_RANGE_WORDS * (size_t)cur_slot;
                if (m3 > -INFINITY) q8_key_update(slot + 2, m3);
                if (a3 > -INFINITY) q8_key_update(slot + 3, a3);
                if (b3 < INFINITY) q8_key_update(slot + 4, -b3);
            }
            __syncthreads();
        }
        ymax = -INFINITY; ya = -INFINITY; yb = INFINITY;
        ycen = 0.0f; yhw = INFINITY;
    };

    for (uint32_t unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
        const uint32_t mt = unit / parts, nt0 = (unit % parts) * per;
        const uint32_t nt1 = nt0 + per < ntiles ? nt0 + per : ntiles;
        if (nt0 >= nt1) continue;
        const uint32_t m0 = mt * 128;
        // the wave's 64 rows x 384 k as MFMA A operands: row l15 of row group i, bytes 16 g .. of k-step s of chunk c
        q8_i32x4 a0[QR_KC][4], a1[QR_KC][4];
        if constexpr (SRC == QR_PREQUANT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t row = m0 + wr * 64 + i * 16 + l15;
                const int8_t* p = A + (size_t)(row < M ? row : M - 1) * K + g * 16;
#pragma unroll
                for (int c = 0; c < QR_KC; ++c) {
                    a0[c][i] = *reinterpret_cast<const q8_i32x4*>(p + c * 128);
                    a1[c][i] = *reinterpret_cast<const q8_i32x4*>(p + c * 128 + 64);
                }
            }
            // the row block's metadata into LDS: x_scale, x zero point, rowsum - K za
            if (tid < 128) {
                Q8RowMeta rm = rmeta[m0 + tid < M ? m0 + tid : M - 1];
                rm.rowsum -= (int)K * rm.za;
                rm.za = -rm.za;  // (as y_of takes it)
                lrow[tid] = rm;
            }
            issue_w(nt0, 0);
            __syncthreads();  // W tile nt0 has landed (vmcnt(0) precedes the barrier); obuf and the other W buffer are free
        } else {
            if constexpr (MU) {
                const bool live = m0 + tid < M;
                uint32_t sw = 0;
                if (tid < 128) sw = row_slot[live ? m0 + tid : M - 1];
                issue_w(nt0, 0);  // (behind the slot load, ahead of the loads that depend on it)
                if (tid < 128) {  // the row's unit: its input parameters (and, store pass, its output parameters)
                    const uint32_t sl = sw & 0x7fffffffu;
                    float rxs_, rxz_;
                    q8_params(in_range + Q8_RANGE_WORDS * (size_t)sl, rxs_, rxz_);
                    Q8RowMeta rm;
                    rm.xs = rxs_; rm.za = (int)rxz_ - 128; rm.rowsum = 0; rm.pad = __float_as_uint(__fdiv_rn(1.0f, rxs_));
                    lrow[tid] = rm;
                    Q8RowOut ro;
                    ro.gs = 1.0f; ro.gz = 0.0f; ro.rgs = 1.0f;
                    ro.slot = live ? sw : NO_SLOT;  // rows past M: in no unit's range
                    if (EPI == Q8_EPI_GELU_Q8) {
                        q8_params_gelu(rq.range + Q8_RANGE_WORDS * (size_t)sl, ro.gs, ro.gz);
                        ro.rgs = __fdiv_rn(1.0f, ro.gs);
                    }
                    lout[tid] = ro;
                }
            } else {
                if (tid < 128) lrow[tid].rowsum = 0;
                issue_w(nt0, 0);
            }
            __syncthreads();
            // 3 chunks x 128 rows x 8 slots of 16 k: six slots per thread, quantised into the image of W buffer 1
            char* abuf = lds + QR_WTILE;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int sidx = tid + QR_THREADS * u, c = sidx >> 10, row = (sidx & 1023) >> 3, sl = sidx & 7;
                const uint32_t m = (m0 + row < M) ? m0 + row : M - 1;  // rows past M repeat the last one: never stored
                float v[16];
                if (SRC == Q8_SRC_F32) {
                    const sh_f32x4* p = reinterpret_cast<const sh_f32x4*>(reinterpret_cast<const float*>(Asrc) + (size_t)m * K + c * 128 + sl * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const sh_f32x4 t = p[q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e];
                    }
                } else {  // 16 k = half a 32-k line: 16 hi halves, their 16 lo halves 64 B further on
                    const _Float16* p = reinterpret_cast<const _Float16*>(Asrc) + ((size_t)m * (K / 32) + c * 4 + (sl >> 1)) * 64 + (sl & 1) * 16;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f16x8 h = *reinterpret_cast<const f16x8*>(p + 8 * q);
                        const f16x8 l = *reinterpret_cast<const f16x8*>(p + 32 + 8 * q);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[8 * q + e] = (float)h[e] + (float)l[e] * kShLoInv;
                    }
                }
                if constexpr (MU) {  // this row's unit
                    const Q8RowMeta rq_ = lrow[row];
                    xs = rq_.xs;
                    xz = (float)(rq_.za + 128);
                    rxs = __uint_as_float(rq_.pad);
                }
                q8_i32x4 packed;
                int sum = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t pw = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // sat_u8(round_half_even(x / x_scale) + x_zp): the quotient by a reciprocal, the true division only where the
                        // two could round apart (|x / x_scale| <= 255: they differ by < 1e-4)
                        const float t = v[4 * w + e] * rxs;
                        float rt = rintf(t);
                        if (fabsf(fabsf(t - rt) - 0.5f) < 1.0e-3f) rt = rintf(__fdiv_rn(v[4 * w + e], xs));
                        const float q = fminf(fmaxf(__fadd_rn(rt, xz), 0.0f), 255.0f);
                        const int a = (int)q - 128;
                        sum += a;
                        pw |= (uint32_t)(a & 0xff) << (8 * e);
                    }
                    packed[w] = (int)pw;
                }
                *reinterpret_cast<q8_i32x4*>(abuf + c * 16384 + row * 128 + ((sl ^ ((row >> 1) & 7)) * 16)) = packed;
                atomicAdd(&lrow[row].rowsum, sum);
            }
            __syncthreads();  // the image and the row sums are complete (and W tile nt0 has landed)
            if constexpr (MU && EPI == Q8_EPI_GELU_RANGE) {  // the units this row block holds (almost always one)
                const uint32_t last_row = M - 1 - m0 < 127u ? M - 1 - m0 : 127u;
                blk_first = __builtin_amdgcn_readfirstlane(lout[0].slot & 0x7fffffffu);
                blk_last = __builtin_amdgcn_readfirstlane(lout[last_row].slot & 0x7fffffffu);
                if (blk_first == blk_last && cur_slot != blk_first) {
                    flush_slot();
                    cur_slot = blk_first;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < QR_KC; ++c) {
                    const char* p = abuf + c * 16384 + (wr * 64 + i * 16 + l15) * 128;
                    a0[c][i] = *reinterpret_cast<const q8_i32x4*>(p + s0);
                    a1[c][i] = *reinterpret_cast<const q8_i32x4*>(p + s1);
                }
            Q8RowMeta rmine;
            if (tid < 128) {
                if constexpr (MU) {
                    rmine = lrow[tid];
                } else {
                    rmine.xs = xs; rmine.za = (int)xz - 128; rmine.rowsum = lrow[tid].rowsum; rmine.pad = 0;
                }
                rmine.rowsum -= (int)K * rmine.za;
                rmine.za = -rmine.za;  // (as y_of takes it)
                if constexpr (MU && EPI == Q8_EPI_GELU_RANGE) {
                    // range pass: a row that is not part of the row block's (one) unit — beyond its call's own padded length, or past
                    // M — yields NaN; with several units in the block the marks are set per unit and tile (below)
                    const uint32_t last_row = M - 1 - m0 < 127u ? M - 1 - m0 : 127u;
                    const uint32_t bf = lout[0].slot & 0x7fffffffu, bl = lout[last_row].slot & 0x7fffffffu;
                    lout[tid].gs = rmine.xs;
                    if (bf == bl && lout[tid].slot != bf) rmine.xs = __builtin_nanf("");
                }
            }
            __syncthreads();  // fragments are in registers (W buffer 1 is free for tile nt0 + 1), row sums read
            if (tid < 128) lrow[tid] = rmine;
            __syncthreads();
        }
        for (uint32_t nt = nt0; nt < nt1; ++nt) {
            const uint32_t n0 = nt * 128;
            const int b = (nt - nt0) & 1;
            char* cur = lds + b * QR_WTILE;
            const Q8ColMeta* lcm = reinterpret_cast<const Q8ColMeta*>(cmbuf + b * QR_CM_BYTES);
            if (nt + 1 < nt1) issue_w(nt + 1, b ^ 1);  // lands under this tile's MFMAs and epilogue
            q8_i32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = q8_i32x4{0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < QR_KC; ++c) {
                q8_i32x4 w0[2], w1[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    w0[j] = *reinterpret_cast<const q8_i32x4*>(cur + c * 16384 + wrow + j * 16 * 128 + s0);
                    w1[j] = *reinterpret_cast<const q8_i32x4*>(cur + c * 16384 + wrow + j * 16 * 128 + s1);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[c][i], w0[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[c][i], w1[j], acc[i][j], 0, 0, 0);
                    }
            }
            // y_of reads the accumulators through inline asm, which the compiler's hazard recogniser does not pad: an MFMA's
            // result must not be read for up to 12 wait states (8-pass XDL; cdna_hip_programming.md 5.7).  Every later read of
            // acc goes through this statement's outputs, so none is scheduled above it.
            asm volatile("s_nop 15"
                         : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),
                           "+v"(acc[3][0]), "+v"(acc[3][1]));
            // y = float(acc with the zero points back in) * (x_scale * W_scale) + bias
            Q8ColMeta cm[2];
            float csc[2];  // one unit, parameters known to the kernel: x_scale * W_scale per column, once per tile
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                cm[j] = lcm[wc * 32 + j * 16 + l15];
                cm[j].zw = -cm[j].zw;
                csc[j] = __fmul_rn(xs, cm[j].ws);
            }
            // the zero points back in: acc - zw rowsum' - za colsum as two v_mad_i32_i24 (|operands| < 2^17; lrow holds -za, cm
            // -zw) — through __mul24 the compiler spent five instructions per element on this, three of them sign extensions
            auto y_of = [&](const Q8RowMeta& rm, int i, int r, int j) {
                int corr;
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(corr) : "v"(cm[j].zw), "v"(rm.rowsum), "v"(acc[i][j][r]));
                asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(corr) : "v"(rm.za), "v"(cm[j].colsum));
                const float sc = (SRC != QR_PREQUANT && !MU) ? csc[j] : __fmul_rn(rm.xs, cm[j].ws);
                return __fadd_rn(__fmul_rn((float)corr, sc), cm[j].bias);
            };
            if constexpr (EPI == Q8_EPI_GELU_RANGE) {
                // max y per element; the two neighbours a, b of the GELU's minimum (q8_params_gelu) only when this tile holds a
                // value inside the window (a, b) the wave has so far — a handful of tiles per wave: three instructions per
                // element instead of eight.  The window test is padded by 1e-5 (rounding of its centre): never misses.
                // MU: the rows that are not the folded unit's carry x_scale = NaN in the row metadata (set at the head of the row
                // block): their y is NaN, which fmaxf / fminf and both comparisons drop — no maximum, neither side of c, outside
                // the window — so the fold itself is the one-unit code.
                auto fold = [&]() {
                    float ys[32], off = INFINITY;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const Q8RowMeta rm = lrow[wr * 64 + i * 16 + 4 * g + r];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float y = y_of(rm, i, r, j);
                                ys[(i * 4 + r) * 2 + j] = y;
                                ymax = fmaxf(ymax, y);
                                off = fminf(off, fabsf(y - ycen));
                            }
                        }
                    if (!(yhw < INFINITY) || __any(off < yhw)) {
#pragma unroll
                        for (int e = 0; e < 32; ++e) {
                            ya = ys[e] <= kGeluArgMin ? fmaxf(ya, ys[e]) : ya;
                            yb = ys[e] >= kGeluArgMin ? fminf(yb, ys[e]) : yb;
                        }
                        float wa = ya, wb = yb;  // the wave's window
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            wa = fmaxf(wa, __shfl_xor(wa, o));
                            wb = fminf(wb, __shfl_xor(wb, o));
                        }
                        ycen = 0.5f * (wa + wb);             // (NaN / inf while a side is still empty: yhw stays inf)
                        yhw = 0.5f * (wb - wa) + 1.0e-5f;
                    }
                };
                if constexpr (MU) {
                    for (uint32_t su = blk_first;; ++su) {  // almost always ONE round: a row block holds one unit
                        if (blk_first != blk_last) {
                            // several units in this row block (rare): a round per unit, the row metadata re-marked for each
                            // (block-uniform, so the barriers are safe)
                            if (su != cur_slot) {
                                flush_slot();
                                cur_slot = su;
                            }
                            __syncthreads();
                            if (tid < 128) lrow[tid].xs = lout[tid].slot == su ? lout[tid].gs : __builtin_nanf("");
                            __syncthreads();
                        }
                        fold();
                        if (su == blk_last) break;
                    }
                } else {
                    fold();
                }
                if (nt == 0 && tid < 128 && m0 + tid < M) rq.rmeta_out[m0 + tid].rowsum = 0;
            } else if constexpr (REQ) {
                int8_t* tile8 = reinterpret_cast<int8_t*>(obuf);  // [128 m][128 n] s8
                if (!MU && tb_inv_w != 0.0f) {  // (block-uniform) one unit: the byte by table
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const Q8RowMeta rm = lrow[wr * 64 + i * 16 + 4 * g + r];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float y = y_of(rm, i, r, j);
                                uint32_t idx = (uint32_t)(int)fmaf(y, tb_inv_w, tb_c0);   // y < QG_YL: negative -> the last entry
                                idx = idx < (uint32_t)(QG_NB - 1) ? idx : (uint32_t)(QG_NB - 1);
                                const Q8GeluEntry e = gtbl[idx];
                                const uint32_t sel = y >= e.thr ? e.w >> 8 : e.w;
                                tile8[(wr * 64 + i * 16 + 4 * g + r) * 128 + wc * 32 + j * 16 + l15] = (int8_t)sel;
                            }
                        }
                } else
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const Q8RowMeta rm = lrow[wr * 64 + i * 16 + 4 * g + r];
                        if constexpr (MU) {  // the row's unit's output parameters
                            const Q8RowOut ro = lout[wr * 64 + i * 16 + 4 * g + r];
                            gs = ro.gs; gz = ro.gz; rgs = ro.rgs;
                        }
                        const float gz128 = gz - 128.0f;  // (the stored byte is the uint8 minus 128: folded into the zero point, exact)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float v = sh_gelu_erf(y_of(rm, i, r, j));
                            const float t = v * rgs;
                            float rt = rintf(t);
                            // |t - rint(t)| <= 0.5: within 1e-3 of a tie exactly when it exceeds 0.499
                            if (fabsf(t - rt) > 0.499f) rt = rintf(__fdiv_rn(v, gs));
                            const float q = fminf(fmaxf(__fadd_rn(rt, gz128), -128.0f), 127.0f);
                            tile8[(wr * 64 + i * 16 + 4 * g + r) * 128 + wc * 32 + j * 16 + l15] = (int8_t)(int)q;
                        }
                    }
                {
                    __syncthreads();
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        const int idx = tid + QR_THREADS * pass, row = idx >> 3, c16 = idx & 7;
                        const q8_i32x4 v = *reinterpret_cast<const q8_i32x4*>(tile8 + row * 128 + c16 * 16);
                        int sum = 0;
#pragma unroll
                        for (int w = 0; w < 4; ++w)
#pragma unroll
                            for (int e = 0; e < 4; ++e) sum += (int)(int8_t)(((uint32_t)v[w] >> (8 * e)) & 0xffu);
                        sum += __shfl_xor(sum, 1);
                        sum += __shfl_xor(sum, 2);
                        sum += __shfl_xor(sum, 4);
                        if (m0 + row < M) {
                            *reinterpret_cast<q8_i32x4*>(rq.out + (size_t)(m0 + row) * N + n0 + c16 * 16) = v;
                            if (c16 == 0) atomicAdd(&rq.rmeta_out[m0 + row].rowsum, sum);
                        }
                    }
                    if (nt == 0 && tid < 128 && m0 + tid < M) {
                        if constexpr (MU) { gs = lout[tid].gs; gz = lout[tid].gz; }
                        rq.rmeta_out[m0 + tid].xs = gs;
                        rq.rmeta_out[m0 + tid].za = (int)gz - 128;
                    }
                }
            } else {
                float* ctile = reinterpret_cast<float*>(obuf);  // [64 m][128 n] f32: one half of the tile at a time
                constexpr int SEPI = REQ ? SH_OUT_F32 : EPI;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (half) __syncthreads();  // the first half has been read
                    if (wr == half) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const Q8RowMeta rm = lrow[wr * 64 + i * 16 + 4 * g + r];
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    ctile[(i * 16 + 4 * g + r) * 128 + wc * 32 + j * 16 + l15] = y_of(rm, i, r, j);
                            }
                    }
                    __syncthreads();
                    const uint32_t mrow0 = m0 + half * 64;
                    if (mrow0 + 64 <= M) q8_rows_store<SEPI, true>(ctile, resid, C, Cs, M, N, mrow0, n0, flag);
                    else if (mrow0 < M) q8_rows_store<SEPI, false>(ctile, resid, C, Cs, M, N, mrow0, n0, flag);
                }
            }
            __syncthreads();  // W tile nt + 1 has landed; every wave is done with this tile's weights and with obuf
        }
    }
    if (EPI == Q8_EPI_GELU_RANGE && MU) {
        flush_slot();
    } else if (EPI == Q8_EPI_GELU_RANGE) {
        float* s_r = reinterpret_cast<float*>(obuf);  // [3][8]
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ymax = fmaxf(ymax, __shfl_xor(ymax, o));
            ya = fmaxf(ya, __shfl_xor(ya, o));
            yb = fminf(yb, __shfl_xor(yb, o));
        }
        if (lane == 0) { s_r[wave] = ymax; s_r[8 + wave] = ya; s_r[16 + wave] = yb; }
        __syncthreads();
        if (tid == 0) {
            float m3 = -INFINITY, a3 = -INFINITY, b3 = INFINITY;
            for (int w = 0; w < 8; ++w) { m3 = fmaxf(m3, s_r[w]); a3 = fmaxf(a3, s_r[8 + w]); b3 = fminf(b3, s_r[16 + w]); }
            if (m3 > -INFINITY) q8_key_update(rq.range + 2, m3);
            if (a3 > -INFINITY) q8_key_update(rq.range + 3, a3);
            if (b3 < INFINITY) q8_key_update(rq.range + 4, -b3);
        }
    }
}

// W [N][K] -> [K / 64][N][64]: the order gemm_q8_ln_kernel streams it in (one thread per 16-byte piece)
__global__ void __launch_bounds__(256)
q8_stage_major_kernel(const int8_t* __restrict__ w, int8_t* __restrict__ out, uint32_t N, uint32_t K) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, per_row = K / 16;
    if (i >= N * per_row) return;
    const uint32_t n = i / per_row, p = i % per_row;
    *reinterpret_cast<q8_i32x4*>(out + ((size_t)(p / 4) * N + n) * 64 + (p % 4) * 16) = *reinterpret_cast<const q8_i32x4*>(w + (size_t)n * K + p * 16);
}

// ---- N = 384 layers (out-proj, FFN-down of the 384-wide models) with their residual add and LayerNorm in the epilogue --------
// The row-block kernel above hands a row's 384 outputs to eight waves and three n-tiles: the LayerNorm behind E4 / E6 had to be
// its own kernel (34 us per layer each at 65,536 rows: 100 MB read, 100 MB written — the two of them a tenth of the forward).
// Here a WAVE owns 16 whole rows: W is the MFMA's first operand, so a lane holds, for each of the 24 column tiles, four
// CONSECUTIVE columns of ONE row (row l15, columns 16 j + 4 g ..) — 96 accumulators; the four lanes of a row meet by two
// shuffles for its statistics and the row leaves normalised in 16-byte stores, with the (lo, hi) the next quantisation wants.
// A block = 8 waves = 128 rows; its activations are MFMA operands in registers — K / 64 x 16 bytes per lane (SRC = Q8_SRC_SPLIT:
// loaded from the split-f16 tensor and quantised on the way in with the tensor's parameters; QR_PREQUANT: the s8 tensor FFN-up
// left, with its row metadata, loaded stage by stage three stages ahead) — and W streams through a ring of five 24-KiB stages
// ([384 n][64 k], one MFMA k-step for all 24 tiles) by LDS-DMA with one bare barrier per stage; fragment reads run eight tiles
// ahead of their MFMAs across those barriers (the loop's notes; profiles/r06_q8_ln_kernel.log).  y, y + residual are the row-block kernel's bits
// (same operations in the same order); the LayerNorm sums a row in another order than layernorm_kernel (in-lane over 96
// values, then across four lanes), so its output may differ from the two-kernel path in the last bit.
// One quantisation unit only (several units: the two-kernel path).  CS_Q8_LN_FUSED=0 restores it.
#ifndef CS_Q8_LN_RESID_PREFETCH
#define CS_Q8_LN_RESID_PREFETCH 1   // (0: the residual read tile by tile inside the epilogue loop, for A/B)
#endif
#ifdef CS_Q8_STAMPS
#define QN_STAMP(v)                                                                      \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : : "memory");     \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define QN_STAMP(v) do { } while (0)
#endif
#ifndef CS_Q8_LN_PF
#define CS_Q8_LN_PF 8   // weight fragments requested ahead of their MFMA (4 registers each)
#endif
#ifndef CS_Q8_LN_DIAG
#define CS_Q8_LN_DIAG 0
#endif
#ifndef CS_Q8_LN_RPRE
#define CS_Q8_LN_RPRE 16  // of the 24 residual tiles of a row, those requested inside the k loop (4 registers each; the rest behind it)
#endif
// the vector-memory requests made at the top of stage i (i < 0: the prologue's weight stages): what the counted waits add up
// residual tiles requested before stage i's top.  K = 384: evenly from stage 0 on (the loop is short); K = 1536: half of them over
// stages ks / 3 .. ks - 4 and half over the last three, where the activation fragments' registers have drained (requested evenly,
// ten tiles already make the allocator spill freshly loaded ones — a wait for memory in the middle of the loop)
constexpr int qn_rpre_upto(int i, int ks) {
    constexpr int R = CS_Q8_LN_RPRE;
    if (i <= 0) return 0;
    if (i >= ks) return R;
    if (ks <= 6) return i * R / ks;
    const int i0 = ks / 3, i1 = ks - 3;
    if (i <= i0) return 0;
    if (i <= i1) return (i - i0) * (R / 2) / (i1 - i0);
    return R / 2 + (i - i1) * (R - R / 2) / 3;
}
constexpr int qn_top_requests(int i, int ks, int ah, int dpw, bool roll) {
    return i < 0 ? dpw : (i + ah < ks ? dpw + (roll ? 1 : 0) : 0) + (qn_rpre_upto(i + 1, ks) - qn_rpre_upto(i, ks));
}
constexpr int qn_younger_requests(int st, int ks, int ah, int dpw, bool roll) {  // those certainly behind stage st's last weight request
    int n = 0;
    for (int i = st - ah + 1; i <= st - 1; ++i) n += qn_top_requests(i, ks, ah, dpw, roll);
    return n;
}
// a loop whose index is a constant expression in its body (immediates of s_waitcnt, bounds of inner loops): the stage loop is
// too large for "#pragma unroll" to be relied on, and left rolled it indexes the register arrays dynamically (scratch memory)
template <class F, int... I>
__device__ __forceinline__ void qn_static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
constexpr int QN_N = 384;
constexpr int QN_STAGE = QN_N * 64;                    // 24,576 B
// (the metadata sits at the bottom of LDS: every read of it is then one base register + an immediate offset; above 64 KiB the
// offsets do not fit the instruction and the compiler keeps — and spills — an address register per (array, tile))
constexpr int QN_OFF_CM = 0;                           // ws [384] | -zw [384] | colsum [384] | bias [384]
constexpr int QN_OFF_LN = QN_OFF_CM + 4 * QN_N * 4;    // gamma [384] | beta [384]
constexpr int QN_OFF_W = QN_OFF_LN + 2 * QN_N * 4;     // 9,216: the ring of weight stages
// (lng: gamma / beta read from global memory in the last pass instead of from LDS — three stages then fit twice into a CU's 160 KiB)
constexpr int qn_lds(int nst, bool lng = false) { return (lng ? QN_OFF_LN : QN_OFF_W) + nst * QN_STAGE + 64; }  // 8 waves, 5 stages: 132,160; 4 waves, 3 stages: 79,936; the last 64 B: the waves' (lo, hi)

// NW waves per block (8, the default: 128 rows, ONE block per CU, a ring of NST = 5 stages; 4: 64 rows, TWO blocks per CU with
// three stages each — the same eight waves per CU as two independent blocks; measured slower, see launch_gemm_q8_ln)
template <int SRC, int KS, int NW, int NST>  // KS = K / 64 stages: 6 (K = 384) | 24 (K = 1536)
__global__ void __launch_bounds__(64 * NW, 2)
gemm_q8_ln_kernel(const void* __restrict__ Asrc, const Q8RowMeta* __restrict__ rmeta, const uint32_t* __restrict__ in_range,
                  const int8_t* __restrict__ W, const Q8ColMeta* __restrict__ cmeta, float* X, const float* __restrict__ ln_g,
                  const float* __restrict__ ln_b, float eps, uint32_t M, float* __restrict__ range_out, uint32_t* __restrict__ range_slot,
                  uint32_t stage_major) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr uint32_t K = 64 * KS;
    constexpr int QN_THREADS = 64 * NW, QN_NST = NST;
    constexpr bool LNG = NW == 4;      // gamma / beta from global memory (their LDS goes to the third weight stage)
    constexpr int OFF_W = LNG ? QN_OFF_LN : QN_OFF_W;
    constexpr int DPW = 24 / NW;       // LDS-DMA instructions of a stage per wave (16 weight rows x 64 B each)
    constexpr uint32_t RG = 16 * NW;   // rows per group
    float* l_ws = reinterpret_cast<float*>(lds + QN_OFF_CM);
    int* l_nzw = reinterpret_cast<int*>(l_ws + QN_N);
    int* l_cs = l_nzw + QN_N;
    float* l_bias = reinterpret_cast<float*>(l_cs + QN_N);
    float* l_g = reinterpret_cast<float*>(lds + QN_OFF_LN);
    float* l_b = l_g + QN_N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    for (int n = tid; n < QN_N; n += QN_THREADS) {
        const Q8ColMeta cm = cmeta[n];
        l_ws[n] = cm.ws; l_nzw[n] = -cm.zw; l_cs[n] = cm.colsum; l_bias[n] = cm.bias;
        if (!LNG) { l_g[n] = ln_g[n]; l_b[n] = ln_b[n]; }
    }
    float xs = 1.0f, xz = 0.0f, rxs = 1.0f;
    if (SRC == Q8_SRC_SPLIT) {
        q8_params(in_range, xs, xz);
        rxs = __fdiv_rn(1.0f, xs);
    }
    // this wave's DPW LDS-DMA instructions of a stage: 16 rows x 64 B each; piece p of row n sits at slot p ^ ((n >> 2) & 2).
    // (ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32 — MI355X_MICROARCH.md, LDS table:
    // with 64-byte rows the 16 lanes of a group are rows r, r + 12 of one 16-byte column and rows r + 4, r + 8 of the next; flipping
    // bit 1 of the slot on rows 8-15 puts the four on four different slots.  The r04-r06 form, p ^ ((n >> 2) & 3), left two of them on
    // the same banks: every fragment read took eight LDS cycles instead of four.)
    uint32_t woff[DPW];
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
        const int row = (wave * DPW + t) * 16 + (lane >> 2);
        woff[t] = (uint32_t)row * (stage_major ? 64u : K) + (((lane & 3) ^ ((row >> 2) & 2)) * 16);
    }
    // W as launch_q8_stage_major leaves it ([K / 64][384][64]: a stage is 24 KiB in a row and every request of 16 rows x 64 B one
    // KiB of whole 128-byte lines) or row-major ([384][K]: the same request takes 64 B out of each of 16 lines — the L2 moves
    // twice the bytes, and the k loop of K = 1536 ran at the rate its weight stream arrived, 16 B per cycle and CU)
    const uint32_t st_stride = stage_major ? (uint32_t)QN_N * 64u : 64u;
    // (a buffer load, not global_load_lds: the compiler files the latter under FLAT — "may touch LDS or memory, may complete out of
    // order" — and from then on answers every wait it inserts itself, the fragment reads' lgkmcnt included, with a full drain)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(W), 0, (int)(QN_N * K), 0x00020000);
    auto issue = [&](uint32_t st) {
        char* buf = lds + OFF_W + (st % QN_NST) * QN_STAGE;
#pragma unroll
        for (int t = 0; t < DPW; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(buf + (wave * DPW + t) * 1024), 16, (int)woff[t],
                                                     (int)(st * st_stride), 0, 0);
    };
    const int frag = l15 * 64 + ((g ^ ((l15 >> 2) & 2)) * 16);  // this lane's 16 bytes of tile j of a stage: + 1024 j
    const uint32_t groups = (M + RG - 1) / RG;
#ifdef CS_Q8_STAMPS
    unsigned long long c_pro = 0, c_wait = 0, c_loop = 0, c_e1 = 0, c_e2 = 0, c_e3 = 0, t0 = 0, t1 = 0, t2 = 0;
    uint32_t c_groups = 0;
#endif
    for (uint32_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        QN_STAMP(t0);
        const uint32_t row = grp * RG + wave * 16 + l15;           // this lane's row (the four lanes l15 + 16 g share it)
        const uint32_t rowc = row < M ? row : M - 1;
        __syncthreads();  // every wave is done with the previous group's last stages (and the metadata is in LDS)
        // (opaque to the optimiser: otherwise the 3 x KS DMA source addresses are loop invariants it keeps — and spills — as
        // 64-bit registers; recomputing one costs an add)
#pragma unroll
        for (int t = 0; t < DPW; ++t) asm volatile("" : "+v"(woff[t]));
        // The weight ring: stage s lives in slot s % NST; AH = NST - 2 stages are requested ahead of the one being read, so the
        // request made at "top of stage s" overwrites stage s - 2, whose last fragment every wave had consumed before it arrived
        // at that barrier (only reads of stage s - 1 are still in flight there).  The barrier of stage s stands PF tiles BEFORE the stage's first MFMA (its first fragment read
        // is the next instruction), is a bare s_barrier behind a counted vmcnt wait — __syncthreads() carries a fence the
        // compiler answers with vmcnt(0), which through round 5 drained every request in flight at every stage — and the fragment
        // reads run PF tiles ahead of their MFMAs across it.  (Rounds 4-5: six reads, a full wait, six MFMAs where registers
        // allowed, else — K = 1536, whose 24 activation fragments were all loaded up front — one read, a full wait, one MFMA:
        // an LDS round trip per MFMA, 2,200 cycles per stage against the 770 of its MFMAs.)
        constexpr int AH = QN_NST - 2;
        constexpr int PF = CS_Q8_LN_PF;
        constexpr int T = 24 * KS;
        constexpr bool ROLL = SRC != Q8_SRC_SPLIT;  // the s8 row arrives stage by stage, AH stages ahead of its MFMAs
        q8_i32x4 a[KS];
        int rowsum = 0;
        float rxs_row = xs;
        int za = 0;
        const int8_t* a8 = reinterpret_cast<const int8_t*>(Asrc) + (size_t)rowc * K + g * 16;
        if constexpr (ROLL) {  // (first: the oldest request in flight, nothing waits behind it)
            const Q8RowMeta rm = rmeta[rowc];
            rxs_row = rm.xs; za = rm.za; rowsum = rm.rowsum;
        }
#pragma unroll
        for (int st = 0; st < AH && st < KS; ++st) issue(st);
        if constexpr (SRC == Q8_SRC_SPLIT) {
            const _Float16* src = reinterpret_cast<const _Float16*>(Asrc) + (size_t)rowc * (K / 32) * 64;
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                const _Float16* p = src + (2 * st + (g >> 1)) * 64 + (g & 1) * 16;
                const f16x8 h0 = *reinterpret_cast<const f16x8*>(p), h1 = *reinterpret_cast<const f16x8*>(p + 8);
                const f16x8 o0 = *reinterpret_cast<const f16x8*>(p + 32), o1 = *reinterpret_cast<const f16x8*>(p + 40);
                q8_i32x4 packed;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    // four values -> one dword (gemm_q8_slab.hip's form of the same arithmetic: the reciprocal quotient lies within
                    // 255 * 2^-23 of the true one, so only within 1e-4 of a tie does the true division decide; the integer in [0, 255]
                    // converts and packs in one instruction, the row sum is a dot product with ones)
                    float v[4], rt[4], d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = 4 * w + e;
                        v[e] = idx < 8 ? (float)h0[idx] + (float)o0[idx] * kShLoInv : (float)h1[idx - 8] + (float)o1[idx - 8] * kShLoInv;
                        const float t = v[e] * rxs;
                        rt[e] = rintf(t);
                        d[e] = t - rt[e];
                    }
                    if (fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3]))) > 0.4999f) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (fabsf(d[e]) > 0.4999f) rt[e] = rintf(__fdiv_rn(v[e], xs));
                    }
                    uint32_t pw = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        pw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(__fadd_rn(rt[e], xz), 0.0f, 255.0f), e, pw);
                    pw ^= 0x80808080u;
                    rowsum = __builtin_amdgcn_sdot4((int)pw, 0x01010101, rowsum, false);
                    packed[w] = (int)pw;
                }
                a[st] = packed;
            }
            rowsum += __shfl_xor(rowsum, 16);
            rowsum += __shfl_xor(rowsum, 32);
            za = (int)xz - 128;
        } else {
#pragma unroll
            for (int st = 0; st < AH && st < KS; ++st) a[st] = *reinterpret_cast<const q8_i32x4*>(a8 + st * 64);
        }
        const int rowsum_c = rowsum - (int)K * za, nza = -za;
        q8_i32x4 acc[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) acc[j] = q8_i32x4{0, 0, 0, 0};
#ifdef CS_Q8_STAMPS
        asm volatile("" : "+v"(a[0]));
        QN_STAMP(t1);
        c_pro += t1 - t0; ++c_groups;
#endif
        // The row's residual: CS_Q8_LN_RPRE of its 24 tiles are requested stage by stage inside the k loop (their registers are
        // the ones the up-front activation fragments held through round 5), the rest behind it — the epilogue used to begin with
        // a memory round trip for all 24 (15-20 thousand cycles per group of the stamps' 70).
        float* xrow = X + (size_t)rowc * QN_N + 4 * g;
#if CS_Q8_LN_RESID_PREFETCH
        sh_f32x4 rpre[24];
#endif
        // top of stage st: this wave's share of it has landed (requests are served in order: behind the LAST of stage st's DPW
        // there are at most AH - 1 younger stages, each with — ROLL — one activation load that the compiler places anywhere among
        // its stage's DPW; the prologue's activation loads may all stand in front of stage 0's last request), then everyone's;
        // request stage st + AH and its activations
        auto top = [&](auto st_) {
            constexpr int st = decltype(st_)::value;
            __builtin_amdgcn_sched_barrier(0);
#ifdef CS_Q8_STAMPS
            QN_STAMP(t2);
#endif
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(qn_younger_requests(st, KS, AH, DPW, ROLL)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#ifdef CS_Q8_STAMPS
            { unsigned long long tw; QN_STAMP(tw); c_wait += tw - t2; }
#endif
            if constexpr (st + AH < KS) {
#if !(defined(CS_Q8_STAMPS) && CS_Q8_LN_DIAG == 1)  // (diagnostic: 1 = no weight requests inside the loop)
                issue(st + AH);
#endif
                if constexpr (ROLL) a[st + AH] = *reinterpret_cast<const q8_i32x4*>(a8 + (st + AH) * 64);
            }
#if CS_Q8_LN_RESID_PREFETCH
#pragma unroll
            for (int j = qn_rpre_upto(st, KS); j < qn_rpre_upto(st + 1, KS); ++j) rpre[j] = *reinterpret_cast<const sh_f32x4*>(xrow + 16 * j);
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        auto wfrag = [&](int u) {  // tile u % 24 of stage u / 24
            return *reinterpret_cast<const q8_i32x4*>(lds + OFF_W + ((u / 24) % QN_NST) * QN_STAGE + frag + (u % 24) * 1024);
        };
        top(std::integral_constant<int, 0>{});
        q8_i32x4 wq[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) wq[u] = wfrag(u);
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
        auto tile = [&](int st, int j) {  // MFMA of tile j of stage st, behind the request for the fragment PF tiles on
            const int t = 24 * st + j, u = t + PF;
            const q8_i32x4 w = wq[t % PF];
#if defined(CS_Q8_STAMPS) && CS_Q8_LN_DIAG == 2  // (diagnostic: 2 = neither fragment reads nor MFMAs, 3 = no fragment reads)
            return;
#elif defined(CS_Q8_STAMPS) && CS_Q8_LN_DIAG == 3
            if (u < PF) wq[t % PF] = wfrag(u);
#else
            if (u < T) wq[t % PF] = wfrag(u);
#endif
            acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, a[st], acc[j], 0, 0, 0);
            // (pinned: one read, one MFMA — left alone the scheduler gathers the reads, 4 registers each, in front)
            if (u < T) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        };
        qn_static_for(std::make_integer_sequence<int, KS>{}, [&](auto st_) {
            constexpr int st = decltype(st_)::value;
#pragma unroll
            for (int j = 0; j < 24 - PF; ++j) tile(st, j);
            if constexpr (st + 1 < KS) top(std::integral_constant<int, st + 1>{});  // the next fragment requested is the next stage's first
#pragma unroll
            for (int j = 24 - PF; j < 24; ++j) tile(st, j);
        });
        __builtin_amdgcn_sched_barrier(0);
        // (the accumulators are read through inline asm below: see the row-block kernel's note on MFMA results and s_nop)
        asm volatile("s_nop 15"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                       "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]),
                       "+v"(acc[16]), "+v"(acc[17]), "+v"(acc[18]), "+v"(acc[19]), "+v"(acc[20]), "+v"(acc[21]), "+v"(acc[22]), "+v"(acc[23]));
#ifdef CS_Q8_STAMPS
        QN_STAMP(t2);
        c_loop += t2 - t1;
#endif
        // y = float(acc with the zero points back in) * (x_scale * W_scale) + bias, + residual: kept in the accumulator registers
        // (the residual tiles not requested inside the k loop, all at once: read tile by tile inside the loop below — whose
        // scheduling fences keep two tiles' loads in flight — the epilogue paid a memory round trip per pair of tiles.
        // CS_Q8_LN_RESID_PREFETCH=0 at compile time restores that form.)
#if CS_Q8_LN_RESID_PREFETCH
#pragma unroll
        for (int j = CS_Q8_LN_RPRE; j < 24; ++j) rpre[j] = *reinterpret_cast<const sh_f32x4*>(xrow + 16 * j);
        __builtin_amdgcn_sched_barrier(0);
#endif
        float vf[96];  // the row's values this lane holds (scalars: partial updates of the accumulator tuples made the allocator spill)
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            const int c0 = 16 * j + 4 * g;
            const sh_f32x4 ws4 = *reinterpret_cast<const sh_f32x4*>(l_ws + c0);
            const q8_i32x4 nzw4 = *reinterpret_cast<const q8_i32x4*>(l_nzw + c0);
            const q8_i32x4 cs4 = *reinterpret_cast<const q8_i32x4*>(l_cs + c0);
            const sh_f32x4 b4 = *reinterpret_cast<const sh_f32x4*>(l_bias + c0);
#if CS_Q8_LN_RESID_PREFETCH
            const sh_f32x4 r4 = rpre[j];
#else
            const sh_f32x4 r4 = *reinterpret_cast<const sh_f32x4*>(xrow + 16 * j);
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int corr;
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(corr) : "v"(nzw4[r]), "v"(rowsum_c), "v"(acc[j][r]));
                asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(corr) : "v"(nza), "v"(cs4[r]));
                const float y = __fadd_rn(__fmul_rn((float)corr, __fmul_rn(rxs_row, ws4[r])), b4[r]);
                const float v = y + r4[r];
                sum += v;
                vf[4 * j + r] = v;
            }
            // (the tile's values are pinned here: left alone, the optimiser runs the integer half and the loads of all 24 tiles
            // first and the float half behind them, and the allocator spills ~160 values per lane)
            asm volatile("" : "+v"(sum), "+v"(vf[4 * j]), "+v"(vf[4 * j + 1]), "+v"(vf[4 * j + 2]), "+v"(vf[4 * j + 3]));
            if (j % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
#ifdef CS_Q8_STAMPS
        asm volatile("" : "+v"(sum));
        QN_STAMP(t1);
        c_e1 += t1 - t2;
#endif
        const float mean = sum * (1.0f / (float)QN_N);
        float qv = 0.0f;
#pragma unroll
        for (int j = 0; j < 24; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = vf[4 * j + r] - mean; qv = fmaf(d, d, qv); }
        qv += __shfl_xor(qv, 16);
        qv += __shfl_xor(qv, 32);
        const float inv = 1.0f / sqrtf(qv * (1.0f / (float)QN_N) + eps);
#ifdef CS_Q8_STAMPS
        { float iv = inv; asm volatile("" : "+v"(iv)); }
        QN_STAMP(t2);
        c_e2 += t2 - t1;
#endif
        float lo = 0.0f, hi = 0.0f;
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            const int c0 = 16 * j + 4 * g;
            const sh_f32x4 g4 = *reinterpret_cast<const sh_f32x4*>((LNG ? ln_g : l_g) + c0);
            const sh_f32x4 b4 = *reinterpret_cast<const sh_f32x4*>((LNG ? ln_b : l_b) + c0);
            sh_f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = (vf[4 * j + r] - mean) * inv * g4[r] + b4[r];
                lo = fminf(lo, o[r]);
                hi = fmaxf(hi, o[r]);
            }
            if (row < M) *reinterpret_cast<sh_f32x4*>(xrow + 16 * j) = o;
            if (j % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (range_out || range_slot) {  // the (lo, hi) of this wave's 16 rows (rows past M: none of theirs)
            if (row >= M) { lo = 0.0f; hi = 0.0f; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo = fminf(lo, __shfl_xor(lo, o));
                hi = fmaxf(hi, __shfl_xor(hi, o));
            }
            if (range_slot) {
                // straight into the range slot of the tensor's next quantisation: no pair buffer, no reduction launch behind the
                // kernel.  One update per BLOCK (the waves meet in LDS): a wave each — 4,096 contended atomics on two addresses —
                // cost the kernel 36 us.
                float* s_r = reinterpret_cast<float*>(lds + OFF_W + QN_NST * QN_STAGE);  // [NW][2], behind the ring
                if (lane == 0) { s_r[2 * wave] = lo; s_r[2 * wave + 1] = hi; }
                __syncthreads();
                if (tid == 0) {
#pragma unroll
                    for (int w = 1; w < NW; ++w) { lo = fminf(lo, s_r[2 * w]); hi = fmaxf(hi, s_r[2 * w + 1]); }
                    q8_range_update(range_slot, lo, hi);
                }
                // (the next group's first barrier stands between these reads and the next writes of s_r)
            } else if (lane == 0) {
                range_out[2 * ((size_t)grp * NW + wave)] = lo;
                range_out[2 * ((size_t)grp * NW + wave) + 1] = hi;
            }
        }
#ifdef CS_Q8_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QN_STAMP(t1);
        c_e3 += t1 - t2;
#endif
    }
#ifdef CS_Q8_STAMPS
    if (blockIdx.x == 3 && (tid == 0 || tid == 64 * (NW - 1)))
        printf("q8 ln SRC %d KS %d wave %d: %u groups, prologue %llu  k loop %llu (of it waiting %llu)  y+resid %llu  variance %llu  normalise+store %llu\n", SRC, KS, wave,
               c_groups, c_pro, c_loop, c_wait, c_e1, c_e2, c_e3);
#endif
}

}  // namespace

uint32_t q8_gelu_table_on() { return gelu_table_on_env(); }

int32_t launch_q8_pack_weight(const float* d_W, const float* d_scale, const float* d_bias, uint32_t N, uint32_t K, int8_t* d_wq,
                              Q8ColMeta* d_cmeta, uint32_t* d_bad, hipStream_t s) {
    if (N == 0 || K == 0) return CS_OK;
    hipLaunchKernelGGL(q8_pack_weight_kernel, dim3(N), dim3(64), 0, s, d_W, d_scale, d_bias, N, K, d_wq, d_cmeta, d_bad);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_q8_row_slots(const uint32_t* d_seq_unit, const uint32_t* d_unit_len, uint32_t T, uint32_t L, uint32_t* d_row_slot,
                            hipStream_t s) {
    if (T == 0 || L == 0) return CS_OK;
    hipLaunchKernelGGL(q8_row_slot_kernel, dim3((T + 255) / 256), dim3(256), 0, s, d_seq_unit, d_unit_len, T, L, d_row_slot);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_q8_quantize(int src_kind, const void* d_src, uint32_t T, uint32_t K, uint32_t* d_range, const uint32_t* d_row_slot,
                           int8_t* d_xq, Q8RowMeta* d_rmeta, hipStream_t s, const float* d_range_pairs, uint32_t n_pairs) {
    if (K % 32) return fail(CS_ERR_UNSUPPORTED, "dynamic quantisation needs K %% 32 == 0 (K = %u)", K);
    if (T == 0) return CS_OK;
    const uint64_t units = (uint64_t)T * (K / (src_kind == Q8_SRC_F32 ? 4 : 8));
    const uint64_t want = (units + 255) / 256;
    const dim3 grid_mm((uint32_t)(want < 1024 ? want : 1024));  // four blocks per CU: enough loads in flight, few range updates
    const dim3 grid_q((T + Q8_RB - 1) / Q8_RB);
    const bool reduced = d_range_pairs && n_pairs && !d_row_slot;  // the tensor's producer already left per-block ranges
    if (reduced) hipLaunchKernelGGL(q8_range_reduce_kernel, dim3(Q8_RR_BLOCKS), dim3(256), 0, s, d_range_pairs, n_pairs, d_range);
    if (src_kind == Q8_SRC_F32) {
        if (!reduced && d_row_slot) hipLaunchKernelGGL(q8_minmax_units_kernel<Q8_SRC_F32>, dim3((T + 31) / 32), dim3(256), 0, s, d_src, T, K, d_range, d_row_slot);
        else if (!reduced) hipLaunchKernelGGL(q8_minmax_kernel<Q8_SRC_F32>, grid_mm, dim3(256), 0, s, d_src, T, K, d_range);
        hipLaunchKernelGGL(q8_quantize_kernel<Q8_SRC_F32>, grid_q, dim3(256), 0, s, d_src, T, K, d_range, d_row_slot, d_xq, d_rmeta);
    } else {
        if (!reduced && d_row_slot) hipLaunchKernelGGL(q8_minmax_units_kernel<Q8_SRC_SPLIT>, dim3((T + 31) / 32), dim3(256), 0, s, d_src, T, K, d_range, d_row_slot);
        else if (!reduced) hipLaunchKernelGGL(q8_minmax_kernel<Q8_SRC_SPLIT>, grid_mm, dim3(256), 0, s, d_src, T, K, d_range);
        hipLaunchKernelGGL(q8_quantize_kernel<Q8_SRC_SPLIT>, grid_q, dim3(256), 0, s, d_src, T, K, d_range, d_row_slot, d_xq, d_rmeta);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// two blocks per CU (64 KiB of LDS each), a multiple of 8 so that a block's slots stay on its XCD
static uint32_t q8_persistent_grid(uint32_t slots) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const uint32_t want = (uint32_t)(2 * cus + 7) / 8 * 8;
    return slots < want ? slots : want;
}

// The row-block kernel (gemm_q8_rows_kernel) takes K = 384 layers from rows_min_m rows on: below that a row block per CU
// leaves most of the chip idle and the tile-per-block kernel spreads the same work over more CUs.  CS_Q8_ROWS=0: never.
static bool q8_rows_takes(uint32_t M, uint32_t K) {
    static const int min_m = [] { const char* e = cs_lab_env("CS_Q8_ROWS"); return e ? std::atoi(e) : 4096; }();
    return K == 128 * QR_KC && min_m > 0 && M >= (uint32_t)min_m;
}
static int q8_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}
template <int EPI, int SRC = QR_PREQUANT, bool MU = false>
static int32_t launch_rows(const void* d_xq, const Q8RowMeta* d_rmeta, const int8_t* d_wq, const Q8ColMeta* d_cmeta,
                           const float* bias, const float* resid, float* C, _Float16* Cs, uint32_t M, uint32_t N,
                           uint32_t* d_flag, Q8Requant rq, hipStream_t s, const uint32_t* d_in_range = nullptr,
                           const uint32_t* d_row_slot = nullptr) {
    constexpr int LDS = MU ? QR_LDS_MU : (EPI == Q8_EPI_GELU_Q8 ? QR_LDS + QG_LDS : QR_LDS);
    static PerDeviceOnce attr;  // function attributes are per device
    CS_TRY(attr.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_rows_kernel<EPI, SRC, MU>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        return CS_OK;
    }));
    const uint32_t mtiles = (M + 127) / 128, ntiles = N / 128, cus = (uint32_t)q8_cus();
    // a unit = one row block x a range of its n-tiles: whole row blocks when there is one per CU, else cut so every CU has work
    uint32_t parts = mtiles >= cus ? 1u : (cus + mtiles - 1) / mtiles;
    if (parts > ntiles) parts = ntiles;
    const uint32_t units = mtiles * parts;
    (void)bias;  // the row-block kernel takes the bias from the column metadata (folded in at create time)
    hipLaunchKernelGGL((gemm_q8_rows_kernel<EPI, SRC, MU>), dim3(units < cus ? units : cus), dim3(QR_THREADS), LDS, s, d_xq, d_wq, d_rmeta,
                       d_cmeta, resid, C, Cs, M, N, d_flag, rq, parts, units, d_in_range, d_row_slot);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// N = 384, one quantisation unit, from 4,096 rows: the product with its residual add and LayerNorm in one kernel
// (gemm_q8_ln_kernel).  src_kind Q8_SRC_SPLIT (d_src = the split-f16 tensor, d_in_range its range slot) or -1 (d_src = the s8
// tensor, d_rmeta its rows).  X [M][384]: the residual on entry, LayerNorm(product + bias + residual) on return; *out_pairs
// (lo, hi) pairs are left in d_range_pairs for the quantisation that follows — or, with d_out_slot, the waves widen that range
// slot themselves (zeroed by the caller at the start of the forward) and *out_pairs = 0: no reduction launch behind the kernel.
int32_t launch_q8_stage_major(const int8_t* d_wq, uint32_t N, uint32_t K, int8_t* d_out, hipStream_t s) {
    if (K % 64) return fail(CS_ERR_UNSUPPORTED, "stage-major weight copy: K=%u is not a multiple of 64", K);
    if (N == 0 || K == 0) return CS_OK;
    hipLaunchKernelGGL(q8_stage_major_kernel, dim3((N * (K / 16) + 255) / 256), dim3(256), 0, s, d_wq, d_out, N, K);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

bool q8_ln_fused_takes(uint32_t M, uint32_t N, uint32_t K) {
    const char* e = cs_lab_env("CS_Q8_LN_FUSED");  // (read per call: tests and A/B scripts flip it mid-process)
    return !(e && e[0] == '0') && N == (uint32_t)QN_N && (K == 384 || K == 1536) && q8_rows_takes(M, 384);
}
int32_t launch_gemm_q8_ln(int src_kind, const void* d_src, const Q8RowMeta* d_rmeta, const uint32_t* d_in_range, const int8_t* d_wq,
                          const Q8ColMeta* d_cmeta, float* X, const float* ln_g, const float* ln_b, float eps, uint32_t M, uint32_t K,
                          float* d_range_pairs, uint32_t* out_pairs, hipStream_t s, uint32_t* d_out_slot, bool w_stage_major) {
    if (out_pairs) *out_pairs = 0;
    if (M == 0) return CS_OK;
    if (K != 384 && K != 1536) return fail(CS_ERR_UNSUPPORTED, "LayerNorm-fused quantised product: K=%u not built (384, 1536)", K);
    if (src_kind != Q8_SRC_SPLIT && src_kind != QR_PREQUANT) return fail(CS_ERR_BAD_ARG, "LayerNorm-fused quantised product: bad source kind");
    if (src_kind == Q8_SRC_SPLIT && K != 384) return fail(CS_ERR_UNSUPPORTED, "LayerNorm-fused quantised product: quantise-on-load is built for K = 384");
    // CS_Q8_LN_WAVES=4: two blocks of four waves per CU (64 rows, three-stage ring, gamma / beta from global memory) instead of one
    // block of eight (128 rows, five-stage ring).  Measured three times (profiles/r05_q8_ln_waves_ab.log; r06_q8_ln_kernel.log (8); with a double buffer, then with
    // three stages): out-proj 88.3 -> 92.5 us, FFN-down 95.0 -> 103.4 — two independent blocks do not overlap what eight waves in
    // step leave exposed.  Diagnostic library only.
#ifdef CS_DIAGNOSTICS
    const char* we = cs_lab_env("CS_Q8_LN_WAVES");
    const bool four = we && we[0] == '4';
#endif
    static PerDeviceOnce attr;  // function attributes are per device
    CS_TRY(attr.run([&]() -> int32_t {
        auto allow = [](auto kernel, int nst) { return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, qn_lds(nst, nst == 3)); };
#ifdef CS_DIAGNOSTICS
        CS_HIP(allow(gemm_q8_ln_kernel<Q8_SRC_SPLIT, 6, 4, 3>, 3));
        CS_HIP(allow(gemm_q8_ln_kernel<QR_PREQUANT, 6, 4, 3>, 3));
        CS_HIP(allow(gemm_q8_ln_kernel<QR_PREQUANT, 24, 4, 3>, 3));
#endif
        CS_HIP(allow(gemm_q8_ln_kernel<Q8_SRC_SPLIT, 6, 8, 5>, 5));
        CS_HIP(allow(gemm_q8_ln_kernel<QR_PREQUANT, 6, 8, 5>, 5));
        CS_HIP(allow(gemm_q8_ln_kernel<QR_PREQUANT, 24, 8, 5>, 5));
        return CS_OK;
    }));
    auto go = [&](auto kernel, uint32_t nw, int nst) -> int32_t {
        const size_t ldsb = (size_t)qn_lds(nst, nst == 3);
        const uint32_t groups = (M + 16 * nw - 1) / (16 * nw), slots = (uint32_t)q8_cus() * (nw == 4 ? 2u : 1u);
        hipLaunchKernelGGL(kernel, dim3(groups < slots ? groups : slots), dim3(64 * nw), ldsb, s, d_src, d_rmeta, d_in_range, d_wq, d_cmeta, X, ln_g, ln_b,
                           eps, M, d_range_pairs, d_out_slot, w_stage_major ? 1u : 0u);
        CS_HIP(hipGetLastError());
        if (out_pairs) *out_pairs = d_out_slot ? 0 : groups * nw;
        return CS_OK;
    };
#ifdef CS_DIAGNOSTICS
    if (four) {
        if (src_kind == Q8_SRC_SPLIT) return go(gemm_q8_ln_kernel<Q8_SRC_SPLIT, 6, 4, 3>, 4, 3);
        if (K == 384) return go(gemm_q8_ln_kernel<QR_PREQUANT, 6, 4, 3>, 4, 3);
        return go(gemm_q8_ln_kernel<QR_PREQUANT, 24, 4, 3>, 4, 3);
    }
#endif
    if (src_kind == Q8_SRC_SPLIT) return go(gemm_q8_ln_kernel<Q8_SRC_SPLIT, 6, 8, 5>, 8, 5);
    if (K == 384) return go(gemm_q8_ln_kernel<QR_PREQUANT, 6, 8, 5>, 8, 5);
    return go(gemm_q8_ln_kernel<QR_PREQUANT, 24, 8, 5>, 8, 5);
}

int32_t launch_gemm_q8(int epi, const int8_t* d_xq, const Q8RowMeta* d_rmeta, const int8_t* d_wq, const Q8ColMeta* d_cmeta,
                       const float* bias, const float* resid, float* C, _Float16* Cs, uint32_t M, uint32_t N, uint32_t K,
                       uint32_t* d_flag, hipStream_t s, int32_t* d_acc_dbg) {
    if (N % SH_BN || K % 128 || K == 0)
        return fail(CS_ERR_UNSUPPORTED, "quantised GEMM N=%u K=%u must be multiples of 128", N, K);
    if (M == 0) return CS_OK;
    if (q8_rows_takes(M, K) && !d_acc_dbg) {
        const Q8Requant none{nullptr, nullptr, nullptr, 0u};
        if (epi == SH_OUT_F32) return launch_rows<SH_OUT_F32>(d_xq, d_rmeta, d_wq, d_cmeta, bias, resid, C, Cs, M, N, d_flag, none, s);
        if (epi == SH_OUT_F32_RESID) return launch_rows<SH_OUT_F32_RESID>(d_xq, d_rmeta, d_wq, d_cmeta, bias, resid, C, Cs, M, N, d_flag, none, s);
        if (epi == SH_OUT_SPLIT) return launch_rows<SH_OUT_SPLIT>(d_xq, d_rmeta, d_wq, d_cmeta, bias, resid, C, Cs, M, N, d_flag, none, s);
        if (epi == SH_OUT_SPLIT_GELU) return launch_rows<SH_OUT_SPLIT_GELU>(d_xq, d_rmeta, d_wq, d_cmeta, bias, resid, C, Cs, M, N, d_flag, none, s);
        return fail(CS_ERR_BAD_ARG, "quantised GEMM: unknown epilogue %d", epi);
    }
    static PerDeviceOnce attr;  // function attributes are per device
    CS_TRY(attr.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_kernel<SH_OUT_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_kernel<SH_OUT_F32_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_kernel<SH_OUT_SPLIT_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_kernel<SH_OUT_SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
        return CS_OK;
    }));
    const uint32_t slots = sh_grid_blocks((M + SH_BM - 1) / SH_BM, N / SH_BN);
    const dim3 grid(q8_persistent_grid(slots));
#define CS_Q8_LAUNCH(E) hipLaunchKernelGGL(gemm_q8_kernel<E>, grid, dim3(256), SH_LDS_BYTES, s, d_xq, d_wq, d_rmeta, d_cmeta, bias, resid, C, Cs, M, N, K, d_flag, d_acc_dbg, Q8Requant{nullptr, nullptr, nullptr, 0u}, slots)
    if (epi == SH_OUT_F32) CS_Q8_LAUNCH(SH_OUT_F32);
    else if (epi == SH_OUT_F32_RESID) CS_Q8_LAUNCH(SH_OUT_F32_RESID);
    else if (epi == SH_OUT_SPLIT) CS_Q8_LAUNCH(SH_OUT_SPLIT);
    else if (epi == SH_OUT_SPLIT_GELU) CS_Q8_LAUNCH(SH_OUT_SPLIT_GELU);
    else return fail(CS_ERR_BAD_ARG, "quantised GEMM: unknown epilogue %d", epi);
#undef CS_Q8_LAUNCH
    CS_HIP(hipGetLastError());
    return CS_OK;
}

bool q8_rows_from_source(uint32_t M, uint32_t K) {
    static const bool on = [] { const char* e = cs_lab_env("CS_Q8_ROWS_SRC"); return !(e && e[0] == '0'); }();
    return on && q8_rows_takes(M, K);
}

int32_t launch_q8_range(int src_kind, const void* d_src, uint32_t T, uint32_t K, uint32_t* d_range, hipStream_t s,
                        const float* d_range_pairs, uint32_t n_pairs) {
    if (T == 0) return CS_OK;
    if (d_range_pairs && n_pairs) {
        hipLaunchKernelGGL(q8_range_reduce_kernel, dim3(Q8_RR_BLOCKS), dim3(256), 0, s, d_range_pairs, n_pairs, d_range);
    } else {
        const uint64_t units = (uint64_t)T * (K / (src_kind == Q8_SRC_F32 ? 4 : 8));
        const uint64_t want = (units + 255) / 256;
        const dim3 grid_mm((uint32_t)(want < 1024 ? want : 1024));
        if (src_kind == Q8_SRC_F32) hipLaunchKernelGGL(q8_minmax_kernel<Q8_SRC_F32>, grid_mm, dim3(256), 0, s, d_src, T, K, d_range);
        else hipLaunchKernelGGL(q8_minmax_kernel<Q8_SRC_SPLIT>, grid_mm, dim3(256), 0, s, d_src, T, K, d_range);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_q8_range_units(const float* d_range_pairs, uint32_t pairs_per_seq, bool pairs_are_rows, const uint32_t* d_seq_unit,
                              const uint32_t* d_unit_len, uint32_t B, uint32_t units, uint32_t* d_range, hipStream_t s) {
    if (B == 0 || units == 0 || pairs_per_seq == 0) return CS_OK;
    hipLaunchKernelGGL(q8_range_reduce_units_kernel, dim3(units), dim3(256), 0, s, d_range_pairs, pairs_per_seq, pairs_are_rows ? 1u : 0u,
                       d_seq_unit, d_unit_len, B, d_range);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_gemm_q8_from_source(int epi, int src_kind, const void* d_src, const uint32_t* d_in_range, const int8_t* d_wq,
                                   const Q8ColMeta* d_cmeta, const float* bias, const float* resid, float* C, _Float16* Cs,
                                   uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s, const uint32_t* d_row_slot,
                                   const uint32_t* d_cmeta_tiles, int8_t* d_xq_scratch) {
    if (d_cmeta_tiles && !d_row_slot && epi == SH_OUT_SPLIT && src_kind == Q8_SRC_F32 && q8_slab_takes(M, N, K))
        return launch_gemm_q8_slab_split(reinterpret_cast<const float*>(d_src), d_in_range, d_wq, d_cmeta_tiles, Cs, M, N, K, d_flag, s, d_xq_scratch);
    if (!q8_rows_takes(M, K) || N % 128) return fail(CS_ERR_UNSUPPORTED, "quantise-on-load product: M=%u N=%u K=%u not taken by the row-block kernel", M, N, K);
    const Q8Requant none{nullptr, nullptr, nullptr, 0u};
    if (epi == SH_OUT_SPLIT && src_kind == Q8_SRC_F32 && d_row_slot)
        return launch_rows<SH_OUT_SPLIT, Q8_SRC_F32, true>(d_src, nullptr, d_wq, d_cmeta, bias, resid, C, Cs, M, N, d_flag, none, s, d_in_range, d_row_slot);
    if (epi == SH_OUT_F32_RESID && src_kind == Q8_SRC_SPLIT && d_row_slot)
        return launch_rows<SH_OUT_F32_RESID, Q8_SRC_SPLIT, true>(d_src, nullptr, d_wq, d_cmeta, bias, resid, C, Cs, M, N, d_flag, none, s, d_in_range, d_row_slot);
    if (epi == SH_OUT_SPLIT && src_kind == Q8_SRC_F32)
        return launch_rows<SH_OUT_SPLIT, Q8_SRC_F32>(d_src, nullptr, d_wq, d_cmeta, bias, resid, C, Cs, M, N, d_flag, none, s, d_in_range);
    if (epi == SH_OUT_F32_RESID && src_kind == Q8_SRC_SPLIT)
        return launch_rows<SH_OUT_F32_RESID, Q8_SRC_SPLIT>(d_src, nullptr, d_wq, d_cmeta, bias, resid, C, Cs, M, N, d_flag, none, s, d_in_range);
    return fail(CS_ERR_UNSUPPORTED, "quantise-on-load product: epilogue %d from source kind %d is not built", epi, src_kind);
}

uint32_t q8_skinny_max_m() {
    static const uint32_t v = [] { const char* e = cs_lab_env("CS_Q8_SKINNY_MAX_M"); return e ? (uint32_t)std::atoll(e) : 512u; }();
    return v;
}

int32_t launch_gemm_q8_skinny(int epi, int src_kind, const void* d_src, const float* d_range_pairs, uint32_t n_pairs,
                              const int8_t* d_wq, const Q8ColMeta* d_cmeta, const float* resid, float* C, _Float16* Cs, uint32_t M,
                              uint32_t N, uint32_t K, uint32_t* d_flag, float* d_range_out, uint32_t* out_pairs, hipStream_t s) {
    if (out_pairs) *out_pairs = 0;
    if (N % 32 || K % 64 || K == 0) return fail(CS_ERR_UNSUPPORTED, "quantised GEMM N=%u K=%u must be multiples of 32 / 64", N, K);
    if (M == 0) return CS_OK;
    const size_t lds = (size_t)16 * (K + 16);
    const bool f32src = src_kind == Q8_SRC_F32;
    // many row tiles of a wide layer (QKV, FFN-up of a query and its variants): 16 x 64 blocks (CS_Q8_SKINNY_WIDE_MIN_M, laboratory knob; 0 = never)
    static const uint32_t wide_min = [] { const char* e = cs_lab_env("CS_Q8_SKINNY_WIDE_MIN_M"); return e ? (uint32_t)std::atol(e) : 128u; }();
    if (wide_min && M >= wide_min && K == 384 && N >= 1024 && N % 64 == 0 && f32src && (epi == SH_OUT_SPLIT || epi == SH_OUT_SPLIT_GELU)) {
        const dim3 wgrid(N / 64, (M + 15) / 16);
        if (epi == SH_OUT_SPLIT)
            hipLaunchKernelGGL((gemm_q8_skinny_kernel<SH_OUT_SPLIT, Q8_SRC_F32, true>), wgrid, dim3(256), lds, s, d_src, d_range_pairs, n_pairs, d_wq, d_cmeta,
                               resid, C, Cs, M, N, K, d_flag, d_range_out);
        else
            hipLaunchKernelGGL((gemm_q8_skinny_kernel<SH_OUT_SPLIT_GELU, Q8_SRC_F32, true>), wgrid, dim3(256), lds, s, d_src, d_range_pairs, n_pairs, d_wq,
                               d_cmeta, resid, C, Cs, M, N, K, d_flag, d_range_out);
        CS_HIP(hipGetLastError());
        if (out_pairs && d_range_out) *out_pairs = wgrid.x * wgrid.y;
        return CS_OK;
    }
    const dim3 grid(N / 16, (M + 15) / 16);
#define CS_Q8S(E, S) hipLaunchKernelGGL((gemm_q8_skinny_kernel<E, S>), grid, dim3(256), lds, s, d_src, d_range_pairs, n_pairs, d_wq, d_cmeta, \
                                        resid, C, Cs, M, N, K, d_flag, d_range_out)
    if (epi == SH_OUT_SPLIT && f32src) CS_Q8S(SH_OUT_SPLIT, Q8_SRC_F32);
    else if (epi == SH_OUT_SPLIT_GELU && f32src) CS_Q8S(SH_OUT_SPLIT_GELU, Q8_SRC_F32);
    else if (epi == SH_OUT_F32_RESID && !f32src) CS_Q8S(SH_OUT_F32_RESID, Q8_SRC_SPLIT);
    else if (epi == SH_OUT_F32 && f32src) CS_Q8S(SH_OUT_F32, Q8_SRC_F32);
    else if (epi == SH_OUT_F32_RESID && f32src) CS_Q8S(SH_OUT_F32_RESID, Q8_SRC_F32);
    else if (epi == SH_OUT_SPLIT && !f32src) CS_Q8S(SH_OUT_SPLIT, Q8_SRC_SPLIT);
    else return fail(CS_ERR_UNSUPPORTED, "few-rows quantised product: epilogue %d from source kind %d is not built", epi, src_kind);
#undef CS_Q8S
    CS_HIP(hipGetLastError());
    if (out_pairs && d_range_out) *out_pairs = grid.x * grid.y;
    return CS_OK;
}

int32_t launch_gemm_q8_skinny_ln(int epi, const float* d_y, const float* ln_g, const float* ln_b, float eps, float* d_xout,
                                 const int8_t* d_wq, const Q8ColMeta* d_cmeta, _Float16* Cs, uint32_t M, uint32_t N, uint32_t* d_flag,
                                 float* d_range_out, uint32_t* out_pairs, hipStream_t s) {
    if (out_pairs) *out_pairs = 0;
    if (M == 0) return CS_OK;
    if (M > 16 || N % 32 || !d_xout || d_xout == d_y)
        return fail(CS_ERR_UNSUPPORTED, "few-rows quantised product with a LayerNorm prologue: %u rows (<= 16), N = %u", M, N);
    const dim3 grid(N / 16, 1);
    const size_t lds = (size_t)16 * (384 + 16);
    const float* none = nullptr;
    if (epi == SH_OUT_SPLIT)
        hipLaunchKernelGGL((gemm_q8_skinny_kernel<SH_OUT_SPLIT, Q8_SRC_LN>), grid, dim3(256), lds, s, d_y, none, 0u, d_wq, d_cmeta, none, nullptr, Cs, M, N,
                           384u, d_flag, d_range_out, ln_g, ln_b, eps, d_xout);
    else if (epi == SH_OUT_SPLIT_GELU)
        hipLaunchKernelGGL((gemm_q8_skinny_kernel<SH_OUT_SPLIT_GELU, Q8_SRC_LN>), grid, dim3(256), lds, s, d_y, none, 0u, d_wq, d_cmeta, none, nullptr, Cs, M,
                           N, 384u, d_flag, d_range_out, ln_g, ln_b, eps, d_xout);
    else return fail(CS_ERR_UNSUPPORTED, "few-rows quantised product with a LayerNorm prologue: epilogue %d is not built", epi);
    CS_HIP(hipGetLastError());
    if (out_pairs && d_range_out) *out_pairs = grid.x;
    return CS_OK;
}

int32_t launch_gemm_q8_gelu_requant_from_source(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const Q8ColMeta* d_cmeta,
                                                const float* bias, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_range_out,
                                                int8_t* d_out, Q8RowMeta* d_rmeta_out, hipStream_t s, const uint32_t* d_row_slot,
                                                const uint32_t* d_cmeta_tiles, int8_t* d_xq_scratch) {
    if (d_cmeta_tiles && !d_row_slot && q8_slab_takes(M, N, K))
        return launch_gemm_q8_slab_gelu_requant(d_x, d_in_range, d_wq, d_cmeta_tiles, M, N, K, d_range_out, d_out, d_rmeta_out, q8_gelu_table_on(), s,
                                                d_xq_scratch);
    if (!q8_rows_takes(M, K) || N % 128) return fail(CS_ERR_UNSUPPORTED, "quantise-on-load product: M=%u N=%u K=%u not taken by the row-block kernel", M, N, K);
    const Q8Requant rq{d_range_out, d_out, d_rmeta_out, q8_gelu_table_on()};
    if (d_row_slot) {
        CS_TRY((launch_rows<Q8_EPI_GELU_RANGE, Q8_SRC_F32, true>(d_x, nullptr, d_wq, d_cmeta, bias, nullptr, nullptr, nullptr, M, N, nullptr, rq, s, d_in_range, d_row_slot)));
        return launch_rows<Q8_EPI_GELU_Q8, Q8_SRC_F32, true>(d_x, nullptr, d_wq, d_cmeta, bias, nullptr, nullptr, nullptr, M, N, nullptr, rq, s, d_in_range, d_row_slot);
    }
    CS_TRY((launch_rows<Q8_EPI_GELU_RANGE, Q8_SRC_F32>(d_x, nullptr, d_wq, d_cmeta, bias, nullptr, nullptr, nullptr, M, N, nullptr, rq, s, d_in_range)));
    return launch_rows<Q8_EPI_GELU_Q8, Q8_SRC_F32>(d_x, nullptr, d_wq, d_cmeta, bias, nullptr, nullptr, nullptr, M, N, nullptr, rq, s, d_in_range);
}

int32_t launch_gemm_q8_gelu_requant(const int8_t* d_xq, const Q8RowMeta* d_rmeta, const int8_t* d_wq, const Q8ColMeta* d_cmeta,
                                    const float* bias, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_range_out, int8_t* d_out,
                                    Q8RowMeta* d_rmeta_out, hipStream_t s) {
    if (N % SH_BN || K % 128 || K == 0)
        return fail(CS_ERR_UNSUPPORTED, "quantised GEMM N=%u K=%u must be multiples of 128", N, K);
    if (M == 0) return CS_OK;
    if (q8_rows_takes(M, K)) {
        const Q8Requant rq{d_range_out, d_out, d_rmeta_out, q8_gelu_table_on()};
        CS_TRY(launch_rows<Q8_EPI_GELU_RANGE>(d_xq, d_rmeta, d_wq, d_cmeta, bias, nullptr, nullptr, nullptr, M, N, nullptr, rq, s));
        return launch_rows<Q8_EPI_GELU_Q8>(d_xq, d_rmeta, d_wq, d_cmeta, bias, nullptr, nullptr, nullptr, M, N, nullptr, rq, s);
    }
    static PerDeviceOnce attr;  // function attributes are per device
    CS_TRY(attr.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_kernel<Q8_EPI_GELU_RANGE>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_kernel<Q8_EPI_GELU_Q8>), hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
        return CS_OK;
    }));
    const uint32_t slots = sh_grid_blocks((M + SH_BM - 1) / SH_BM, N / SH_BN);
    const dim3 grid(q8_persistent_grid(slots));
    const Q8Requant rq{d_range_out, d_out, d_rmeta_out, 0u};
    hipLaunchKernelGGL(gemm_q8_kernel<Q8_EPI_GELU_RANGE>, grid, dim3(256), SH_LDS_BYTES, s, d_xq, d_wq, d_rmeta, d_cmeta, bias,
                       (const float*)nullptr, (float*)nullptr, (_Float16*)nullptr, M, N, K, (uint32_t*)nullptr, (int32_t*)nullptr, rq, slots);
    hipLaunchKernelGGL(gemm_q8_kernel<Q8_EPI_GELU_Q8>, grid, dim3(256), SH_LDS_BYTES, s, d_xq, d_wq, d_rmeta, d_cmeta, bias,
                       (const float*)nullptr, (float*)nullptr, (_Float16*)nullptr, M, N, K, (uint32_t*)nullptr, (int32_t*)nullptr, rq, slots);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
