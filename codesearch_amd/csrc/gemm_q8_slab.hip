// gemm_q8_slab.hip — the K = 384 products of a dynamically quantised model at indexing batch sizes (SURVEY.md §8a E2 / E5 for the
// registry's *Q models — the reference's DEFAULT model, /root/reference/src/embed/embedder.rs:12-13): QKV and the two passes of
// FFN-up, from the f32 rows the LayerNorm left, ONE quantisation unit.  Same arithmetic as gemm_q8_rows_kernel (gemm_q8.hip), so
// the same bits; another arrangement of the work, built on what that kernel's in-kernel clock stamps said
// (profiles/r05_q8_ln_epilogue_ab.log (4)): its range pass waited 40 % of its tile loop for the next 48-KiB weight tile (a
// 128-row block re-streams all of W: 302 MB of L2 -> LDS per FFN-up pass), and its storing passes spent 4-7x their MFMA time in
// epilogues that cross LDS (byte-wise s8 tile / f32 half tiles) behind three block barriers per tile.
//
//  * a block owns a SLAB of 256 rows (eight waves x 32 rows): every weight tile that lands in LDS is used by twice the rows —
//    151 MB of L2 -> LDS per pass instead of 302;
//  * W is the MFMA's FIRST operand (as in gemm_q8_ln_kernel): a lane then holds, per 16 x 16 tile, four CONSECUTIVE columns of ONE
//    row.  A wave owns its 32 rows over all 128 columns of the n-tile (2 x 8 MFMA tiles, 64 accumulators), so
//      - row quantities (row sum, the row's output sum) are lane constants: no row metadata in LDS, no LDS atomics;
//      - the four re-quantised bytes of a (row, tile) are one dword, and two half-wave exchanges (v_permlane32_swap,
//        v_permlane16_swap: a 4 x 4 transpose over the four lanes of a row) leave every lane 16 consecutive bytes of a row:
//        the s8 tile goes from registers to HBM in 16-byte stores — no byte writes to LDS, no barrier;
//      - the split-f16 planes of QKV leave the same way (one exchange: 16 bytes per lane and plane);
//    ONE barrier per tile (the weight double buffer) instead of four;
//  * the wave's activations are quantised by the lane that will hold them as MFMA operand (16 consecutive k per fragment), straight
//    from global memory: no LDS staging, the row sums by two shuffles;
//  * an n-tile's column metadata rides with its weights as a 2-KiB structure-of-arrays image (ws | -zw | colsum | bias: built once
//    at create time, launch_q8_cmeta_tiles), read as 16-byte vectors for the lane's four columns; za * colsum is the accumulators'
//    initial value, so the zero points cost one v_mad_i32_i24 per output.
// LDS: 2 x 48 KiB weights + 2 x 2 KiB metadata (+ 16 KiB GELU byte table in the store pass): one block per CU.
#include <cstdlib>

#include "encoder.hpp"
#include "gemm_q8.hpp"
#include "gemm_q8_dev.hpp"
#include "split_f16.hpp"

namespace cs {

namespace {

constexpr int QS_KC = 3;                      // K = 384: three 128-byte chunks per row
constexpr int QS_ROWS = 256;
constexpr int QS_THREADS = 512;
constexpr int QS_WTILE = 128 * 128 * QS_KC;   // one n-tile of weights over all of K: 49,152 B
constexpr int QS_CM_BYTES = 4 * 128 * 4;      // ws [128] | -zw [128] | colsum [128] | bias [128]
constexpr int QS_OFF_CM = 2 * QS_WTILE;
constexpr int QS_OFF_TBL = QS_OFF_CM + 2 * QS_CM_BYTES;
constexpr int qs_lds(bool table) { return QS_OFF_TBL + (table ? QG_LDS : 0) + 128; }  // the last 128 B: the waves' extremes

__global__ void __launch_bounds__(256)
q8_cmeta_tiles_kernel(const Q8ColMeta* __restrict__ cm, uint32_t N, uint32_t* __restrict__ out) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const Q8ColMeta m = cm[n];
    uint32_t* t = out + (size_t)(n >> 7) * 512 + (n & 127);
    t[0] = __float_as_uint(m.ws);
    t[128] = (uint32_t)(-m.zw);
    t[256] = (uint32_t)m.colsum;
    t[384] = __float_as_uint(m.bias);
}

// four values -> (hi, lo) planes, the bits of sh_split (split_f16.hpp: same operations as sh_split8, half the width)
__device__ __forceinline__ void qs_split4(sh_f32x4 v, uint32_t (&hw)[2], uint32_t (&lw)[2], uint32_t& mx) {
    asm volatile("" : "+v"(v));  // the f32 values are the only source of hi (see sh_split)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const sh_f32x2 x = {v[2 * p], v[2 * p + 1]};
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

__device__ __forceinline__ void qs_swap32(uint32_t& a, uint32_t& b) {  // a's lanes 32-63 <-> b's lanes 0-31
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void qs_swap16(uint32_t& a, uint32_t& b) {  // a's odd rows of 16 lanes <-> b's even rows
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

typedef uint32_t qs_u32x4 __attribute__((ext_vector_type(4)));

// EPI: SH_OUT_SPLIT (QKV: bias, split-f16 store) | Q8_EPI_GELU_RANGE | Q8_EPI_GELU_Q8 (the two passes of FFN-up, gemm_q8.hip)
template <int EPI>
__global__ void __launch_bounds__(QS_THREADS, 2)
gemm_q8_slab_kernel(const float* __restrict__ X, const int8_t* __restrict__ W, const uint32_t* __restrict__ cmt,
                    _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t* __restrict__ flag, Q8Requant rq, uint32_t parts,
                    uint32_t total_units, const uint32_t* __restrict__ in_range) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr uint32_t K = 128 * QS_KC;
    constexpr bool TABLE = EPI == Q8_EPI_GELU_Q8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int swz = (l15 >> 1) & 7;
    const int f0 = l15 * 128 + (g ^ swz) * 16, f1 = l15 * 128 + ((4 + g) ^ swz) * 16;  // this lane's 16 bytes of a weight row's two k-steps
    const uint32_t ntiles = N / 128, per = (ntiles + parts - 1) / parts;
    float* s_r = reinterpret_cast<float*>(lds + QS_OFF_TBL + (TABLE ? QG_LDS : 0));  // [3][8]
    // this wave's six LDS-DMA instructions of a W tile (of 48: chunk q / 16, rows 8 (q % 16) ..): per-lane source offsets (the image
    // gemm_q8_rows_kernel stages: row r's 16-byte slot c at physical slot c ^ ((r >> 1) & 7))
    uint32_t woff[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int q = wave * 6 + t, c = q >> 4, row = (q & 15) * 8 + (lane >> 3);
        woff[t] = (uint32_t)row * K + c * 128 + (((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    auto issue_w = [&](uint32_t nt, int b) {
        const int8_t* src = W + (size_t)nt * 128 * K;
        char* buf = lds + b * QS_WTILE;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            uint32_t o = woff[t];
            asm volatile("" : "+v"(o));  // (opaque: otherwise the six source addresses are kept — and spilled — as 64-bit loop invariants)
            sh_glds16(src + o, buf + (wave * 6 + t) * 1024);
        }
        if (wave < 2)  // the tile's 2 KiB of column metadata: lane l of wave w moves bytes 1024 w + 16 l ..
            sh_glds16(reinterpret_cast<const char*>(cmt) + (size_t)nt * QS_CM_BYTES + wave * 1024 + lane * 16,
                      lds + QS_OFF_CM + b * QS_CM_BYTES + wave * 1024);
    };
    // DynamicQuantizeLinear's parameters of the input tensor
    float xs, xz;
    q8_params(in_range, xs, xz);
    const float rxs = __fdiv_rn(1.0f, xs);
    const int za = (int)xz - 128, nza = -za;
    // store pass: the output tensor's parameters, the byte by table (q8_build_gelu_table); tb_inv_w == 0: the direct form
    float gs = 1.0f, gz = 0.0f, rgs = 1.0f;
    Q8GeluEntry* gtbl = reinterpret_cast<Q8GeluEntry*>(lds + QS_OFF_TBL);
    float tb_inv_w = 0.0f, tb_c0 = 0.0f;
    if (EPI == Q8_EPI_GELU_Q8) {
        q8_params_gelu(rq.range, gs, gz);
        rgs = __fdiv_rn(1.0f, gs);
        if (rq.use_table && rq.range[2]) {
            uint32_t* ok_bad = reinterpret_cast<uint32_t*>(lds);  // (W buffer 0: nothing has been issued into it yet)
            if (threadIdx.x == 0) *ok_bad = 0u;
            __syncthreads();
            tb_inv_w = q8_build_gelu_table(gtbl, ok_bad, gs, rgs, gz - 128.0f, q8_unkey(rq.range[2]), QS_THREADS);
            tb_c0 = -QG_YL * tb_inv_w;
            __syncthreads();  // (ok_bad has been read by every thread before the first W tile lands on it)
        }
    }
    const float gz128 = gz - 128.0f;
    float ymax = -INFINITY, ya = -INFINITY, yb = INFINITY;
    float ycen = 0.0f, yhw = INFINITY;  // wave-uniform: centre and half-width (padded) of the wave's (a, b) so far
    uint32_t mx = 0;                    // packed maximum of |hi| bit patterns (sh_split_overflowed)

    for (uint32_t unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
        const uint32_t mt = unit / parts, nt0 = (unit % parts) * per;
        const uint32_t nt1 = nt0 + per < ntiles ? nt0 + per : ntiles;
        if (nt0 >= nt1) continue;
        const uint32_t m0 = mt * QS_ROWS + wave * 32;  // this wave's first row
        issue_w(nt0, 0);
        // the wave's 32 rows x 384 k as MFMA operands, quantised on the way in: fragment (c, s, i) = row 16 i + l15, k 128 c + 64 s + 16 g ..
        q8_i32x4 a[QS_KC][2][2];
        int rowsum[2] = {0, 0};
#pragma unroll
        for (int c = 0; c < QS_KC; ++c) {
            sh_f32x4 v[2][2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t row = m0 + i * 16 + l15;
                    const sh_f32x4* p = reinterpret_cast<const sh_f32x4*>(X + (size_t)(row < M ? row : M - 1) * K + c * 128 + s * 64 + g * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[s][i][q] = p[q];
                }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    q8_i32x4 packed;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        uint32_t pw = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // sat_u8(round_half_even(x / x_scale) + x_zp): the quotient by a reciprocal, the true division only where the
                            // two could round apart (|x / x_scale| <= 255: they differ by < 1e-4) — gemm_q8_rows_kernel's arithmetic
                            const float x = v[s][i][w][e];
                            const float t = x * rxs;
                            float rt = rintf(t);
                            if (fabsf(fabsf(t - rt) - 0.5f) < 1.0e-3f) rt = rintf(__fdiv_rn(x, xs));
                            const float q = fminf(fmaxf(__fadd_rn(rt, xz), 0.0f), 255.0f);
                            const int b = (int)q - 128;
                            rowsum[i] += b;
                            pw |= (uint32_t)(b & 0xff) << (8 * e);
                        }
                        packed[w] = (int)pw;
                    }
                    a[c][s][i] = packed;
                }
        }
        int rowsum_c[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rowsum[i] += __shfl_xor(rowsum[i], 16);
            rowsum[i] += __shfl_xor(rowsum[i], 32);
            rowsum_c[i] = rowsum[i] - (int)K * za;
        }
        int osum[2] = {0, 0};  // store pass: the row's sum of stored bytes over this unit's tiles (this lane's columns)
        if (EPI == Q8_EPI_GELU_RANGE && nt0 == 0 && tid < QS_ROWS && mt * QS_ROWS + tid < M) rq.rmeta_out[mt * QS_ROWS + tid].rowsum = 0;
        __syncthreads();  // W tile nt0 has landed (vmcnt(0) precedes the barrier)

        for (uint32_t nt = nt0; nt < nt1; ++nt) {
            const uint32_t n0 = nt * 128;
            const int b = (nt - nt0) & 1;
            const char* cur = lds + b * QS_WTILE;
            const char* cmb = lds + QS_OFF_CM + b * QS_CM_BYTES + g * 16;  // + 64 j: this lane's four columns of tile j; + 512 per array
            if (nt + 1 < nt1) issue_w(nt + 1, b ^ 1);  // lands under this tile's MFMAs and epilogue
            q8_i32x4 acc[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // za * colsum: the accumulators' start
                const q8_i32x4 cs4 = *reinterpret_cast<const q8_i32x4*>(cmb + 1024 + 64 * j);
                q8_i32x4 u;
#pragma unroll
                for (int r = 0; r < 4; ++r) asm("v_mul_i32_i24 %0, %1, %2" : "=v"(u[r]) : "v"(nza), "v"(cs4[r]));
                acc[0][j] = u;
                acc[1][j] = u;
            }
#pragma unroll
            for (int c = 0; c < QS_KC; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const char* wp = cur + c * 16384 + (s ? f1 : f0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const q8_i32x4 w = *reinterpret_cast<const q8_i32x4*>(wp + j * 2048);
                        acc[0][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, a[c][s][0], acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, a[c][s][1], acc[1][j], 0, 0, 0);
                    }
                }
            // The accumulators are read through inline asm below, which the compiler's hazard recogniser does not pad: an MFMA's
            // result must not be read for up to 12 wait states (cdna_hip_programming.md 5.7).  Every later read of acc goes through
            // this statement's outputs, so none is scheduled above it.
            asm volatile("s_nop 15"
                         : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]), "+v"(acc[0][5]),
                           "+v"(acc[0][6]), "+v"(acc[0][7]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]),
                           "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(acc[1][7]));
            // y = float(acc with the zero points back in) * (x_scale * W_scale) + bias for row group i, tile j: four consecutive columns
            auto y4_of = [&](int i, int j) {
                const sh_f32x4 ws4 = *reinterpret_cast<const sh_f32x4*>(cmb + 64 * j);
                const q8_i32x4 nzw4 = *reinterpret_cast<const q8_i32x4*>(cmb + 512 + 64 * j);
                const sh_f32x4 b4 = *reinterpret_cast<const sh_f32x4*>(cmb + 1536 + 64 * j);
                sh_f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int corr;
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(corr) : "v"(nzw4[r]), "v"(rowsum_c[i]), "v"(acc[i][j][r]));
                    y[r] = __fadd_rn(__fmul_rn((float)corr, __fmul_rn(xs, ws4[r])), b4[r]);
                }
                return y;
            };
            if constexpr (EPI == Q8_EPI_GELU_RANGE) {
                // max y per element; the two neighbours a, b of the GELU's minimum (q8_params_gelu) only when this row group holds a
                // value inside the window (a, b) the wave has so far (gemm_q8_rows_kernel's fold)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float ys[32], off = INFINITY;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const sh_f32x4 y = y4_of(i, j);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ys[4 * j + r] = y[r];
                            ymax = fmaxf(ymax, y[r]);
                            off = fminf(off, fabsf(y[r] - ycen));
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
                }
            } else if constexpr (EPI == Q8_EPI_GELU_Q8) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    uint32_t d[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const sh_f32x4 y = y4_of(i, j);
                        uint32_t sel[4];
                        if (tb_inv_w != 0.0f) {  // (block-uniform) the byte by table
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                uint32_t idx = (uint32_t)(int)fmaf(y[r], tb_inv_w, tb_c0);   // y < QG_YL: negative -> the last entry
                                idx = idx < (uint32_t)(QG_NB - 1) ? idx : (uint32_t)(QG_NB - 1);
                                const Q8GeluEntry e = gtbl[idx];
                                sel[r] = y[r] >= e.thr ? e.w >> 8 : e.w;
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) sel[r] = (uint32_t)q8_gelu_byte(y[r], gs, rgs, gz128);
                        }
                        // the four low bytes -> one dword (columns 16 j + 4 g .. + 3 of the row)
                        const uint32_t p01 = __builtin_amdgcn_perm(sel[1], sel[0], 0x0c0c0400u);
                        const uint32_t p23 = __builtin_amdgcn_perm(sel[3], sel[2], 0x0c0c0400u);
                        d[j] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                        osum[i] = __builtin_amdgcn_sdot4((int)d[j], 0x01010101, osum[i], false);
                    }
                    // 4 x 4 transpose over the row's four lanes: lane g ends with tile g's 16 bytes in d[0..3], tile 4 + g's in d[4..7]
                    qs_swap32(d[0], d[2]); qs_swap32(d[1], d[3]); qs_swap32(d[4], d[6]); qs_swap32(d[5], d[7]);
                    qs_swap16(d[0], d[1]); qs_swap16(d[2], d[3]); qs_swap16(d[4], d[5]); qs_swap16(d[6], d[7]);
                    const uint32_t row = m0 + i * 16 + l15;
                    if (row < M) {
                        int8_t* dst = rq.out + (size_t)row * N + n0 + 16 * g;
                        *reinterpret_cast<qs_u32x4*>(dst) = qs_u32x4{d[0], d[1], d[2], d[3]};
                        *reinterpret_cast<qs_u32x4*>(dst + 64) = qs_u32x4{d[4], d[5], d[6], d[7]};
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t row = m0 + i * 16 + l15;
                    _Float16* line = Cs + ((size_t)row * (N / 32) + (n0 >> 5)) * 64;  // + 64 per 32 columns; hi [32] | lo [32]
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {  // a line = tiles 2 jp, 2 jp + 1
                        uint32_t h0[2], l0[2], h1[2], l1[2];
                        qs_split4(y4_of(i, 2 * jp), h0, l0, mx);
                        qs_split4(y4_of(i, 2 * jp + 1), h1, l1, mx);
                        // one exchange between the lane pairs (g, g ^ 1): even g ends with columns 4 g .. 4 g + 7 of the line's first tile,
                        // odd g with columns 4 (g - 1) .. of its second tile — 16 bytes per lane and plane
                        qs_swap16(h0[0], h1[0]); qs_swap16(h0[1], h1[1]);
                        qs_swap16(l0[0], l1[0]); qs_swap16(l0[1], l1[1]);
                        if (row < M) {
                            _Float16* dst = line + 64 * jp + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
                            __builtin_nontemporal_store(qs_u32x4{h0[0], h0[1], h1[0], h1[1]}, reinterpret_cast<qs_u32x4*>(dst));
                            __builtin_nontemporal_store(qs_u32x4{l0[0], l0[1], l1[0], l1[1]}, reinterpret_cast<qs_u32x4*>(dst + 32));
                        }
                    }
                }
            }
            __syncthreads();  // W tile nt + 1 has landed; every wave is done with this tile's weights and metadata
        }
        if constexpr (EPI == Q8_EPI_GELU_Q8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int s = osum[i];
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                const uint32_t row = m0 + i * 16 + l15;
                if (g == 0 && row < M) {
                    atomicAdd(&rq.rmeta_out[row].rowsum, s);
                    if (nt0 == 0) {
                        rq.rmeta_out[row].xs = gs;
                        rq.rmeta_out[row].za = (int)gz - 128;
                    }
                }
            }
        }
    }
    if (EPI == Q8_EPI_GELU_RANGE) {
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
    if (EPI == SH_OUT_SPLIT && flag && sh_split_overflowed(mx)) atomicOr(flag, 1u);
}

int qs_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int EPI>
int32_t launch_slab(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const uint32_t* d_cmt, _Float16* Cs, uint32_t M,
                    uint32_t N, uint32_t* d_flag, Q8Requant rq, hipStream_t s) {
    constexpr int LDS = qs_lds(EPI == Q8_EPI_GELU_Q8);
    static PerDeviceOnce attr;  // function attributes are per device
    CS_TRY(attr.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_slab_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        return CS_OK;
    }));
    const uint32_t slabs = (M + QS_ROWS - 1) / QS_ROWS, ntiles = N / 128, cus = (uint32_t)qs_cus();
    // a unit = one slab x a range of its n-tiles: whole slabs when there is one per CU, else cut so every CU has work
    uint32_t parts = slabs >= cus ? 1u : (cus + slabs - 1) / slabs;
    if (parts > ntiles) parts = ntiles;
    const uint32_t units = slabs * parts;
    hipLaunchKernelGGL(gemm_q8_slab_kernel<EPI>, dim3(units < cus ? units : cus), dim3(QS_THREADS), LDS, s, d_x, d_wq, d_cmt, Cs, M, N, d_flag,
                       rq, parts, units, d_in_range);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace

int32_t launch_q8_cmeta_tiles(const Q8ColMeta* d_cmeta, uint32_t N, uint32_t* d_tiles, hipStream_t s) {
    if (N % 128) return fail(CS_ERR_UNSUPPORTED, "column metadata tiles: N=%u must be a multiple of 128", N);
    if (N == 0) return CS_OK;
    hipLaunchKernelGGL(q8_cmeta_tiles_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_cmeta, N, d_tiles);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

bool q8_slab_takes(uint32_t M, uint32_t N, uint32_t K) {
    // below ~8k rows the slabs x n-tile parts no longer cover the chip and the 128-row blocks spread the same work over more CUs
    static const int min_m = [] { const char* e = std::getenv("CS_Q8_SLAB_MIN_M"); return e ? std::atoi(e) : 8192; }();
    return K == 128 * QS_KC && N % 128 == 0 && N > 0 && min_m > 0 && M >= (uint32_t)min_m;
}

int32_t launch_gemm_q8_slab_split(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const uint32_t* d_cmeta_tiles,
                                  _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s) {
    if (K != 128 * QS_KC || N % 128 || N == 0) return fail(CS_ERR_UNSUPPORTED, "slab product: N=%u K=%u not built (K = 384, N %% 128 == 0)", N, K);
    if (M == 0) return CS_OK;
    return launch_slab<SH_OUT_SPLIT>(d_x, d_in_range, d_wq, d_cmeta_tiles, Cs, M, N, d_flag, Q8Requant{nullptr, nullptr, nullptr, 0u}, s);
}

int32_t launch_gemm_q8_slab_gelu_requant(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const uint32_t* d_cmeta_tiles,
                                         uint32_t M, uint32_t N, uint32_t K, uint32_t* d_range_out, int8_t* d_out, Q8RowMeta* d_rmeta_out,
                                         uint32_t use_table, hipStream_t s) {
    if (K != 128 * QS_KC || N % 128 || N == 0) return fail(CS_ERR_UNSUPPORTED, "slab product: N=%u K=%u not built (K = 384, N %% 128 == 0)", N, K);
    if (M == 0) return CS_OK;
    const Q8Requant rq{d_range_out, d_out, d_rmeta_out, use_table};
    CS_TRY(launch_slab<Q8_EPI_GELU_RANGE>(d_x, d_in_range, d_wq, d_cmeta_tiles, nullptr, M, N, nullptr, rq, s));
    return launch_slab<Q8_EPI_GELU_Q8>(d_x, d_in_range, d_wq, d_cmeta_tiles, nullptr, M, N, nullptr, rq, s);
}

}  // namespace cs
