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
// LDS: 2 x 48 KiB weights + 3 x 2 KiB metadata (+ 16 KiB GELU byte table in the store pass): one block per CU.
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
constexpr int QS_OFF_TBL = QS_OFF_CM + 3 * QS_CM_BYTES;  // (three metadata buffers: the late waves read tile t's while tile t + 2's lands)
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

// -DCS_Q8_STAMPS (benchmarks/build_variant.sh): wave 0 of block 3 reads the shader clock at a unit's start, behind the quantising
// prologue, and per tile behind the MFMAs, behind the epilogue and behind the end barrier; the sums are printed at the kernel's end.
#ifdef CS_Q8_STAMPS
#define QS_STAMP(v)                                                                      \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : : "memory");     \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define QS_STAMP(v) do { } while (0)
#endif
// fragments of W requested ahead of the MFMAs that use them (left to the compiler: two reads, a full wait, four MFMAs)
#ifndef CS_Q8_SLAB_PF
#define CS_Q8_SLAB_PF 4
#endif
#ifndef CS_Q8_SLAB_NT
#define CS_Q8_SLAB_NT 0   // (1: the split-f16 planes leave with the non-temporal policy, for A/B)
#endif

// EPI: SH_OUT_SPLIT (QKV: bias, split-f16 store) | Q8_EPI_GELU_RANGE | Q8_EPI_GELU_Q8 (the two passes of FFN-up, gemm_q8.hip)
// PREQ: X is the s8 tensor [M][384] the range pass left in xq_out (the store pass of FFN-up: a quarter of the f32 rows' bytes —
// the prologue is bound by what a CU can take in, ~10 B per cycle — and no second quantisation)
template <int EPI, bool PREQ = false>
__global__ void __launch_bounds__(QS_THREADS, 2)
gemm_q8_slab_kernel(const void* __restrict__ Xv, const int8_t* __restrict__ W, const uint32_t* __restrict__ cmt,
                    _Float16* __restrict__ Cs, uint32_t M, uint32_t N, uint32_t* __restrict__ flag, Q8Requant rq, uint32_t parts,
                    uint32_t total_units, const uint32_t* __restrict__ in_range, int8_t* __restrict__ xq_out) {
    const float* X = reinterpret_cast<const float*>(Xv);
    const int8_t* Xq = reinterpret_cast<const int8_t*>(Xv);
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
    // (buffer loads: split_f16.hpp, sh_blds16 — the fragment reads' waits stay counted while the next tile is on its way)
    const sh_rsrc w_rsrc = sh_make_rsrc(W, N * K), cm_rsrc = sh_make_rsrc(cmt, ntiles * (uint32_t)QS_CM_BYTES);
    auto issue_w = [&](uint32_t nt, int b, int cb) {
        char* buf = lds + b * QS_WTILE;
#pragma unroll
        for (int t = 0; t < 6; ++t) sh_blds16(w_rsrc, woff[t], nt * 128 * K, buf + (wave * 6 + t) * 1024);
        if (wave < 2)  // the tile's 2 KiB of column metadata: lane l of wave w moves bytes 1024 w + 16 l ..
            sh_blds16(cm_rsrc, (uint32_t)(wave * 1024 + lane * 16), nt * (uint32_t)QS_CM_BYTES, lds + QS_OFF_CM + cb * QS_CM_BYTES + wave * 1024);
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
    float ycen = 0.0f, yhw = INFINITY;  // wave-uniform: centre and half-width (padded) of the (a, b) window in force
    float wwa = -INFINITY, wwb = INFINITY;  // the wave's own window so far
    // The block's waves share their windows through two ordered keys in LDS (0 = none yet): a wave alone sees 4,096 values per tile
    // and its window shrinks like 1 / tiles, so the window update below ran on about half of its tiles (stamps, profiles/
    // r06_q8_slab_ab.log (8)); with eight waves' values behind the window it runs on one tile in sixteen.  Never misses: a value
    // outside ANY wave's (a, b) cannot be the tensor's a or b.
    uint32_t* s_win = reinterpret_cast<uint32_t*>(s_r + 24);
    if (EPI == Q8_EPI_GELU_RANGE) {
        if (tid < 2) s_win[tid] = 0u;
        __syncthreads();
    }
    uint32_t mx = 0;                    // packed maximum of |hi| bit patterns (sh_split_overflowed)
#ifdef CS_Q8_STAMPS
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, c_pro = 0, c_mfma = 0, c_epi = 0, c_bar = 0, c_dma = 0, c_epi_a = 0;
    (void)st2; (void)st3; (void)st4;
    uint32_t c_tiles = 0;
#endif

    // Waves 0-3 and waves 4-7 (a pair per SIMD) walk the tiles half a tile apart: between two barriers the first four run a tile's
    // MFMAs and then its epilogue, the other four the PREVIOUS tile's epilogue and then this tile's MFMAs — the matrix pipe works
    // for one wave of a SIMD while the vector pipe works for the other (run in step, both waited for the pipe together and then
    // left it idle through both epilogues: stamps in profiles/r06_q8_slab_stamps.log).
    const int late = wave >> 2;
    q8_i32x4 acc[2][8];
    for (uint32_t unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
        const uint32_t mt = unit / parts, nt0 = (unit % parts) * per;
        const uint32_t nt1 = nt0 + per < ntiles ? nt0 + per : ntiles;
        if (nt0 >= nt1) continue;
        const uint32_t m0 = mt * QS_ROWS + wave * 32;  // this wave's first row
        QS_STAMP(st0);
        issue_w(nt0, 0, 0);
        // the wave's 32 rows x 384 k as MFMA operands, quantised on the way in: fragment (c, s, i) = row 16 i + l15, k 128 c + 64 s + 16 g ..
        q8_i32x4 a[QS_KC][2][2];
        int rowsum[2] = {0, 0};
        if constexpr (PREQ) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t row = m0 + i * 16 + l15;
                const int8_t* p = Xq + (size_t)(row < M ? row : M - 1) * K + g * 16;
#pragma unroll
                for (int c = 0; c < QS_KC; ++c)
#pragma unroll
                    for (int s = 0; s < 2; ++s) a[c][s][i] = *reinterpret_cast<const q8_i32x4*>(p + c * 128 + s * 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < QS_KC; ++c)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int w = 0; w < 4; ++w) rowsum[i] = __builtin_amdgcn_sdot4(a[c][s][i][w], 0x01010101, rowsum[i], false);
        } else
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
                        // sat_u8(round_half_even(x / x_scale) + x_zp) - 128, four values -> one dword.  The quotient by a reciprocal:
                        // |x / x_scale| <= 255 and two roundings (the reciprocal, the product) put it within 255 * 2^-23 = 3.1e-5 of the
                        // true quotient, so its rint IS the true quotient's unless it lies within 1e-4 of a tie — there (a few dwords in
                        // a hundred, wave-wide) the true division decides.  Same bytes as q8_quantize_kernel.
                        const sh_f32x4 x = v[s][i][w];
                        float rt[4], d[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = x[e] * rxs;
                            rt[e] = rintf(t);
                            d[e] = t - rt[e];
                        }
                        if (fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3]))) > 0.4999f) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (fabsf(d[e]) > 0.4999f) rt[e] = rintf(__fdiv_rn(x[e], xs));
                        }
                        uint32_t pw = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e)  // (the operand is an integer in [0, 255]: the conversion is exact)
                            pw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(__fadd_rn(rt[e], xz), 0.0f, 255.0f), e, pw);
                        pw ^= 0x80808080u;
                        rowsum[i] = __builtin_amdgcn_sdot4((int)pw, 0x01010101, rowsum[i], false);
                        packed[w] = (int)pw;
                    }
                    a[c][s][i] = packed;
                    if (xq_out && nt0 == 0) {  // (range pass) the quantised rows, for the store pass
                        const uint32_t row = m0 + i * 16 + l15;
                        if (row < M) *reinterpret_cast<q8_i32x4*>(xq_out + (size_t)row * K + c * 128 + s * 64 + g * 16) = packed;
                    }
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
        QS_STAMP(st1);
#ifdef CS_Q8_STAMPS
        c_pro += st1 - st0;
#endif

        // y = float(acc with the zero points back in) * (x_scale * W_scale) + bias for tile j, both row groups: four consecutive columns
        // each (one set of metadata reads for the two)
        auto y8_of = [&](const char* cmb, int j, sh_f32x4& y0, sh_f32x4& y1) {
            const sh_f32x4 ws4 = *reinterpret_cast<const sh_f32x4*>(cmb + 64 * j);
            const q8_i32x4 nzw4 = *reinterpret_cast<const q8_i32x4*>(cmb + 512 + 64 * j);
            const sh_f32x4 b4 = *reinterpret_cast<const sh_f32x4*>(cmb + 1536 + 64 * j);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = __fmul_rn(xs, ws4[r]);
                int c0, c1;
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(c0) : "v"(nzw4[r]), "v"(rowsum_c[0]), "v"(acc[0][j][r]));
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(c1) : "v"(nzw4[r]), "v"(rowsum_c[1]), "v"(acc[1][j][r]));
                y0[r] = __fadd_rn(__fmul_rn((float)c0, sc), b4[r]);
                y1[r] = __fadd_rn(__fmul_rn((float)c1, sc), b4[r]);
            }
        };
        const uint32_t T = nt1 - nt0;
        // (the accumulators carry nothing from the previous unit or through the prologue: said so, or the register allocator keeps them live there)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = q8_i32x4{0, 0, 0, 0};
        // sub-step u: the early waves run tile t's MFMAs at u = 2 t and its epilogue at 2 t + 1, the late waves one sub-step behind;
        // a barrier after every odd u (W tile t + 1, issued at u = 2 t, has landed; both halves are done with W tile t)
        for (uint32_t u = 0; u <= 2 * T; ++u) {
            if (!(u & 1) && (u >> 1) + 1 < T) issue_w(nt0 + (u >> 1) + 1, ((u >> 1) + 1) & 1, ((u >> 1) + 1) % 3);
#ifdef CS_Q8_STAMPS
            { unsigned long long td; QS_STAMP(td); c_dma += td - st1; st1 = td; }
#endif
            const uint32_t v = u - (uint32_t)late;  // (late waves at u = 0: nothing yet)
            const uint32_t t = v >> 1;
            if (v <= 2 * T - 1 && !(v & 1)) {
                // ---- tile t's MFMAs
                const char* cur = lds + (t & 1) * QS_WTILE;
                const char* cmb = lds + QS_OFF_CM + (t % 3) * QS_CM_BYTES + g * 16;  // + 64 j: this lane's four columns of tile j; + 512 per array
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // za * colsum: the accumulators' start
                    const q8_i32x4 cs4 = *reinterpret_cast<const q8_i32x4*>(cmb + 1024 + 64 * j);
                    q8_i32x4 uu;
#pragma unroll
                    for (int r = 0; r < 4; ++r) asm("v_mul_i32_i24 %0, %1, %2" : "=v"(uu[r]) : "v"(nza), "v"(cs4[r]));
                    acc[0][j] = uu;
                    acc[1][j] = uu;
                }
                {
                    // fragment f = (chunk f / 16, k-step (f / 8) % 2, tile f % 8): one read feeds two MFMAs; PF reads stay in flight ahead of
                    // the MFMAs that use them (the scheduling groups below pin that order)
                    constexpr int PF = CS_Q8_SLAB_PF;
                    auto wfrag = [&](int f) {
                        return *reinterpret_cast<const q8_i32x4*>(cur + (f >> 4) * 16384 + (((f >> 3) & 1) ? f1 : f0) + (f & 7) * 2048);
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    q8_i32x4 wq[PF];
#pragma unroll
                    for (int f = 0; f < PF; ++f) wq[f] = wfrag(f);
#pragma unroll
                    for (int f = 0; f < 48; ++f) {
                        const int c = f >> 4, s = (f >> 3) & 1, j = f & 7;
                        const q8_i32x4 w = wq[f % PF];
                        if (f + PF < 48) wq[f % PF] = wfrag(f + PF);
                        acc[0][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, a[c][s][0], acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, a[c][s][1], acc[1][j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                    for (int f = 0; f < 48; ++f) {
                        if (f + PF < 48) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // The accumulators are read through inline asm in the epilogue, which the compiler's hazard recogniser does not pad: an
                // MFMA's result must not be read for up to 12 wait states (cdna_hip_programming.md 5.7).  Every later read of acc goes
                // through this statement's outputs, so none is scheduled above it.
                asm volatile("s_nop 15"
                             : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]), "+v"(acc[0][5]),
                               "+v"(acc[0][6]), "+v"(acc[0][7]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]),
                               "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(acc[1][7]));
#ifdef CS_Q8_STAMPS
                QS_STAMP(st2);
                c_mfma += st2 - st1; st1 = st2;
#endif
            } else if (v <= 2 * T - 1) {
                // ---- tile t's epilogue
                const uint32_t n0 = (nt0 + t) * 128;
                const char* cmb = lds + QS_OFF_CM + (t % 3) * QS_CM_BYTES + g * 16;
                if constexpr (EPI == Q8_EPI_GELU_RANGE) {
                    // max y per element; the two neighbours a, b of the GELU's minimum (q8_params_gelu) only when this tile holds a
                    // value inside the window (a, b) the wave has so far (gemm_q8_rows_kernel's fold).  v_max3 / v_min3 by hand: through
                    // fmaxf / fminf the compiler quiets every operand first (a v_max_f32 x, x per value: 330 of the fold's 1,000
                    // instructions); a NaN y is dropped by the hardware min / max all the same.
                    float ys[64], off = INFINITY;
                    {
                        const uint32_t k0 = s_win[0], k1 = s_win[1];  // (the same address in every lane: wave-uniform)
                        const float ba = k0 ? fmaxf(wwa, q8_unkey(k0)) : wwa, bb = k1 ? fminf(wwb, -q8_unkey(k1)) : wwb;
                        ycen = 0.5f * (ba + bb);             // (NaN / inf while a side is still empty: yhw stays inf)
                        yhw = 0.5f * (bb - ba) + 1.0e-5f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        sh_f32x4 y0, y1;
                        y8_of(cmb, j, y0, y1);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ys[8 * j + r] = y0[r];
                            ys[8 * j + 4 + r] = y1[r];
                            const float d0 = y0[r] - ycen, d1 = y1[r] - ycen;
                            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(ymax) : "v"(y0[r]), "v"(y1[r]));
                            asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(off) : "v"(d0), "v"(d1));
                        }
#ifdef CS_Q8_STAMPS
                        if (j == 3) { unsigned long long th; QS_STAMP(th); c_epi_a += th - st1; }
#endif
                    }
                    if (!(yhw < INFINITY) || __any(off < yhw)) {
#pragma unroll
                        for (int e = 0; e < 64; ++e) {
                            ya = ys[e] <= kGeluArgMin ? fmaxf(ya, ys[e]) : ya;
                            yb = ys[e] >= kGeluArgMin ? fminf(yb, ys[e]) : yb;
                        }
                        float wa = ya, wb = yb;  // the wave's window
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            wa = fmaxf(wa, __shfl_xor(wa, o));
                            wb = fminf(wb, __shfl_xor(wb, o));
                        }
                        wwa = wa;
                        wwb = wb;
                        if (lane == 0) {
                            if (wa > -INFINITY) atomicMax(&s_win[0], q8_key(wa));
                            if (wb < INFINITY) atomicMax(&s_win[1], q8_key(-wb));
                        }
                    }
                } else if constexpr (EPI == Q8_EPI_GELU_Q8) {
                    uint32_t d0[8], d1[8];
                    // the four low bytes -> one dword (columns 16 j + 4 g .. + 3 of the row)
                    auto pack4 = [](const uint32_t (&sel)[4]) {
                        const uint32_t p01 = __builtin_amdgcn_perm(sel[1], sel[0], 0x0c0c0400u);
                        const uint32_t p23 = __builtin_amdgcn_perm(sel[3], sel[2], 0x0c0c0400u);
                        return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                    };
                    if (tb_inv_w != 0.0f) {  // (block-uniform) the byte by table
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            sh_f32x4 y0, y1;
                            y8_of(cmb, j, y0, y1);
                            uint32_t s0[4], s1[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                uint32_t i0 = (uint32_t)(int)fmaf(y0[r], tb_inv_w, tb_c0), i1 = (uint32_t)(int)fmaf(y1[r], tb_inv_w, tb_c0);
                                i0 = i0 < (uint32_t)(QG_NB - 1) ? i0 : (uint32_t)(QG_NB - 1);   // y < QG_YL: negative -> the last entry
                                i1 = i1 < (uint32_t)(QG_NB - 1) ? i1 : (uint32_t)(QG_NB - 1);
                                const Q8GeluEntry e0 = gtbl[i0], e1 = gtbl[i1];
                                s0[r] = y0[r] >= e0.thr ? e0.w >> 8 : e0.w;
                                s1[r] = y1[r] >= e1.thr ? e1.w >> 8 : e1.w;
                            }
                            d0[j] = pack4(s0);
                            d1[j] = pack4(s1);
                            if (j & 1) __builtin_amdgcn_sched_barrier(0);  // (two tiles' reads in flight: unrestrained, all eight are hoisted and spill)
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            sh_f32x4 y0, y1;
                            y8_of(cmb, j, y0, y1);
                            uint32_t s0[4], s1[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                s0[r] = (uint32_t)q8_gelu_byte(y0[r], gs, rgs, gz128);
                                s1[r] = (uint32_t)q8_gelu_byte(y1[r], gs, rgs, gz128);
                            }
                            d0[j] = pack4(s0);
                            d1[j] = pack4(s1);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        uint32_t (&d)[8] = i ? d1 : d0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) osum[i] = __builtin_amdgcn_sdot4((int)d[j], 0x01010101, osum[i], false);
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
                    const uint32_t row0 = m0 + l15, row1 = m0 + 16 + l15;
                    // + 64 per 32 columns; hi [32] | lo [32]
                    _Float16* line0 = Cs + ((size_t)row0 * (N / 32) + (n0 >> 5)) * 64 + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
                    _Float16* line1 = line0 + (size_t)16 * (N / 32) * 64;
                    auto put = [&](_Float16* dst, bool live, const uint32_t (&a0)[2], const uint32_t (&a1)[2]) {
                        if (!live) return;
                        // (plain stores: an instruction writes 16 rows x 64 B — half lines — and the non-temporal policy nearly halves the
                        // write rate of that shape, 5.9 -> 3.4 TB/s: profiles/r02_hbm_write_probe.log; QKV 116 -> 89 us)
#if CS_Q8_SLAB_NT
                        __builtin_nontemporal_store(qs_u32x4{a0[0], a0[1], a1[0], a1[1]}, reinterpret_cast<qs_u32x4*>(dst));
#else
                        *reinterpret_cast<qs_u32x4*>(dst) = qs_u32x4{a0[0], a0[1], a1[0], a1[1]};
#endif
                    };
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {  // a line = tiles 2 jp, 2 jp + 1
                        sh_f32x4 ya0, ya1, yb0, yb1;
                        y8_of(cmb, 2 * jp, ya0, ya1);
                        y8_of(cmb, 2 * jp + 1, yb0, yb1);
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            uint32_t h0[2], l0[2], h1[2], l1[2];
                            qs_split4(i ? ya1 : ya0, h0, l0, mx);
                            qs_split4(i ? yb1 : yb0, h1, l1, mx);
                            // one exchange between the lane pairs (g, g ^ 1): even g ends with columns 4 g .. 4 g + 7 of the line's first tile,
                            // odd g with columns 4 (g - 1) .. of its second tile — 16 bytes per lane and plane
                            qs_swap16(h0[0], h1[0]); qs_swap16(h0[1], h1[1]);
                            qs_swap16(l0[0], l1[0]); qs_swap16(l0[1], l1[1]);
                            _Float16* dst = (i ? line1 : line0) + 64 * jp;
                            put(dst, (i ? row1 : row0) < M, h0, h1);
                            put(dst + 32, (i ? row1 : row0) < M, l0, l1);
                        }
                    }
                }
#ifdef CS_Q8_STAMPS
                QS_STAMP(st3);
                c_epi += st3 - st1; st1 = st3; ++c_tiles;
#endif
            }
            if (u & 1) {
                __syncthreads();
#ifdef CS_Q8_STAMPS
                QS_STAMP(st4);
                c_bar += st4 - st1; st1 = st4;
#endif
            }
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
        __syncthreads();  // the late waves' last epilogue has read its metadata: the next unit may stage over it
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
#ifdef CS_Q8_STAMPS
    if (blockIdx.x == 3 && (tid == 0 || tid == 256))
        printf("q8 slab EPI %d N %u wave %d: %u tiles, prologue %llu  dma issue %llu  mfma %llu  epilogue %llu (first half %llu)  end barrier %llu\n", EPI, N, wave, c_tiles, c_pro, c_dma, c_mfma, c_epi, c_epi_a, c_bar);
#endif
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

// X [M][384] f32 -> xq [M][384] s8 with the tensor's parameters: the slab kernel's own quantising arithmetic (reciprocal quotient, the true
// division inside the tie window: the bytes of q8_quantize_kernel), one thread per 16 values.  For calls of few slabs, whose units
// (slab x a range of n-tiles) would otherwise each quantise the slab's 256 rows again.
__global__ void __launch_bounds__(256)
qs_prequant_kernel(const float* __restrict__ X, const uint32_t* __restrict__ in_range, int8_t* __restrict__ xq, uint64_t n16) {
    float xs, xz;
    q8_params(in_range, xs, xz);
    const float rxs = __fdiv_rn(1.0f, xs);
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const sh_f32x4* p = reinterpret_cast<const sh_f32x4*>(X + i * 16);
    q8_i32x4 packed;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const sh_f32x4 x = p[w];
        float rt[4], d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = x[e] * rxs;
            rt[e] = rintf(t);
            d[e] = t - rt[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (fabsf(d[e]) > 0.4999f) rt[e] = rintf(__fdiv_rn(x[e], xs));
        uint32_t pw = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) pw = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(__fadd_rn(rt[e], xz), 0.0f, 255.0f), e, pw);
        packed[w] = (int)(pw ^ 0x80808080u);
    }
    *reinterpret_cast<q8_i32x4*>(xq + i * 16) = packed;
}

// units per slab the launch below cuts (1 when every CU has a slab of its own)
static uint32_t qs_parts(uint32_t M, uint32_t N) {
    const uint32_t slabs = (M + QS_ROWS - 1) / QS_ROWS, ntiles = N / 128, cus = (uint32_t)qs_cus();
    uint32_t parts = slabs >= cus ? 1u : (cus + slabs - 1) / slabs;
    return parts > ntiles ? ntiles : parts;
}

template <int EPI, bool PREQ = false>
int32_t launch_slab(const void* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const uint32_t* d_cmt, _Float16* Cs, uint32_t M,
                    uint32_t N, uint32_t* d_flag, Q8Requant rq, hipStream_t s, int8_t* d_xq_out = nullptr) {
    constexpr int LDS = qs_lds(EPI == Q8_EPI_GELU_Q8);
    static PerDeviceOnce attr;  // function attributes are per device
    CS_TRY(attr.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_q8_slab_kernel<EPI, PREQ>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        return CS_OK;
    }));
    const uint32_t slabs = (M + QS_ROWS - 1) / QS_ROWS, ntiles = N / 128, cus = (uint32_t)qs_cus();
    // a unit = one slab x a range of its n-tiles: whole slabs when there is one per CU, else cut so every CU has work
    uint32_t parts = slabs >= cus ? 1u : (cus + slabs - 1) / slabs;
    if (parts > ntiles) parts = ntiles;
    const uint32_t units = slabs * parts;
    hipLaunchKernelGGL((gemm_q8_slab_kernel<EPI, PREQ>), dim3(units < cus ? units : cus), dim3(QS_THREADS), LDS, s, d_x, d_wq, d_cmt, Cs, M, N, d_flag,
                       rq, parts, units, d_in_range, d_xq_out);
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
    static const int min_m = [] { const char* e = cs_lab_env("CS_Q8_SLAB_MIN_M"); return e ? std::atoi(e) : 8192; }();
    return K == 128 * QS_KC && N % 128 == 0 && N > 0 && min_m > 0 && M >= (uint32_t)min_m;
}

int32_t launch_gemm_q8_slab_split(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const uint32_t* d_cmeta_tiles,
                                  _Float16* Cs, uint32_t M, uint32_t N, uint32_t K, uint32_t* d_flag, hipStream_t s, int8_t* d_xq_scratch) {
    if (K != 128 * QS_KC || N % 128 || N == 0) return fail(CS_ERR_UNSUPPORTED, "slab product: N=%u K=%u not built (K = 384, N %% 128 == 0)", N, K);
    if (M == 0) return CS_OK;
    // few slabs (a call of 32 chunks: the reference's own call shape, src/embed/batch.rs:70): three and more units share a slab and
    // each would quantise its 256 rows again (16 of a unit's ~30 us) — quantise once, the units read the bytes
    if (d_xq_scratch && qs_parts(M, N) >= 3) {
        const uint64_t n16 = (uint64_t)M * K / 16;
        hipLaunchKernelGGL(qs_prequant_kernel, dim3((uint32_t)((n16 + 255) / 256)), dim3(256), 0, s, d_x, d_in_range, d_xq_scratch, n16);
        CS_HIP(hipGetLastError());
        return launch_slab<SH_OUT_SPLIT, true>(d_xq_scratch, d_in_range, d_wq, d_cmeta_tiles, Cs, M, N, d_flag, Q8Requant{nullptr, nullptr, nullptr, 0u}, s);
    }
    return launch_slab<SH_OUT_SPLIT>(d_x, d_in_range, d_wq, d_cmeta_tiles, Cs, M, N, d_flag, Q8Requant{nullptr, nullptr, nullptr, 0u}, s);
}

int32_t launch_gemm_q8_slab_gelu_requant(const float* d_x, const uint32_t* d_in_range, const int8_t* d_wq, const uint32_t* d_cmeta_tiles,
                                         uint32_t M, uint32_t N, uint32_t K, uint32_t* d_range_out, int8_t* d_out, Q8RowMeta* d_rmeta_out,
                                         uint32_t use_table, hipStream_t s, int8_t* d_xq_scratch) {
    if (K != 128 * QS_KC || N % 128 || N == 0) return fail(CS_ERR_UNSUPPORTED, "slab product: N=%u K=%u not built (K = 384, N %% 128 == 0)", N, K);
    if (M == 0) return CS_OK;
    const Q8Requant rq{d_range_out, d_out, d_rmeta_out, use_table};
    // d_xq_scratch [M][384] s8 (optional): the range pass leaves the rows it quantised there and the store pass takes them from it
    if (d_xq_scratch && qs_parts(M, N) >= 3) {  // (few slabs: launch_gemm_q8_slab_split's note) both passes read bytes quantised once
        const uint64_t n16 = (uint64_t)M * K / 16;
        hipLaunchKernelGGL(qs_prequant_kernel, dim3((uint32_t)((n16 + 255) / 256)), dim3(256), 0, s, d_x, d_in_range, d_xq_scratch, n16);
        CS_HIP(hipGetLastError());
        CS_TRY((launch_slab<Q8_EPI_GELU_RANGE, true>(d_xq_scratch, d_in_range, d_wq, d_cmeta_tiles, nullptr, M, N, nullptr, rq, s)));
        return (launch_slab<Q8_EPI_GELU_Q8, true>(d_xq_scratch, d_in_range, d_wq, d_cmeta_tiles, nullptr, M, N, nullptr, rq, s));
    }
    CS_TRY(launch_slab<Q8_EPI_GELU_RANGE>(d_x, d_in_range, d_wq, d_cmeta_tiles, nullptr, M, N, nullptr, rq, s, d_xq_scratch));
    if (d_xq_scratch) return (launch_slab<Q8_EPI_GELU_Q8, true>(d_xq_scratch, d_in_range, d_wq, d_cmeta_tiles, nullptr, M, N, nullptr, rq, s));
    return launch_slab<Q8_EPI_GELU_Q8>(d_x, d_in_range, d_wq, d_cmeta_tiles, nullptr, M, N, nullptr, rq, s);
}

}  // namespace cs
