// small_forward.hip — the forward pass of a FEW SHORT SEQUENCES (the query side of `codesearch search`:
// EmbeddingService::embed_query / embed_queries_batch, /root/reference/src/embed/mod.rs:164-226 — one query and up to eight
// variants of a dozen tokens) as ONE kernel launch.
//
// Why.  On the multi-launch path such a forward is 86 kernels of ~2.6 us behind ~2 us boundaries (one 16-token query:
// 398 us on the device, profiles/r04_query_latency.log; the trace of round 5, r05_small_forward_trace_before.txt): the
// weights of a layer are first touched when its kernel starts, every LayerNorm is a launch of its own, and FFN-down's
// 24 blocks pull 196 KB each through one CU's load path (10 us).  Here 96 resident blocks walk the phases of the forward
// between grid barriers:
//   per layer   QKV (LayerNorm of the previous layer — or the embedding gather — as the PROLOGUE of the product: every
//               block normalises the 16 rows it needs itself) | attention | out-proj + residual | FFN-up (LayerNorm as
//               prologue) + GELU | FFN-down + residual;   last: the final LayerNorm
//   5 phases per layer instead of 7 launches, and each block issues the LDS-DMA of the weight tile its NEXT phase needs
//   before it waits at the barrier (it knows its tile: the assignment is static), so the weights fly while the grid
//   synchronises — the one thing a launch boundary cannot do.
//
// Same arithmetic, same bits as the multi-launch path: a dense-layer tile is gemm_sh_skinny_kernel<EPI, 1, 1>'s (K split
// over the block's four waves, chunks w, w + 4, ... in order, partial tiles summed (w0 + w1) + (w2 + w3)); LayerNorm is
// ln_row_core (encoder_rows.hpp); attention is attention_shx_body<1> (attention_shx_body.hpp).  tests/test_gpu_small_forward.py
// holds the two paths to each other bit for bit.
//
// Hand-off between blocks (cdna_hip_programming.md Guideline 16, R1; MI355X_MICROARCH.md "Valid forms"): every byte one
// phase writes and a later phase reads is stored with sc1 (write-through, 4 / 8 / 16 B per lane) and loaded with sc1
// (buffer_load ... sc1 to registers: L1 is bypassed) — no agent-scope release / acquire, which on this part write back
// and invalidate the L2 the weights sit in (that experiment made a query SLOWER).  A barrier = every
// wave drains its stores (s_waitcnt vmcnt(0)), the block meets, ONE lane adds to a counter and polls it (sc1 loads,
// s_sleep), the block meets again.  Every spin is bounded: a launch that cannot make progress writes a give-up code, every
// block leaves, and the caller reruns the mini-batch on the multi-launch path.
#include "small_forward.hpp"

#include "attention_shx_body.hpp"
#include "encoder_rows.hpp"
#include "gemm_epilogue.hpp"
#include "split_f16.hpp"

#include <type_traits>

namespace cs {

namespace {

typedef unsigned int sf_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int sf_u32x2 __attribute__((ext_vector_type(2)));

constexpr int SF_GRID = 96;                         // resident blocks: the widest phase (FFN-up: 1536 / 16 column tiles)
constexpr int SF_W_BYTES = 16 * 12 * 128;           // one 16-row weight tile of 12 k-chunks: [chunk][16 rows][128 B], swizzled
constexpr int SF_AROW = 384 * 4 + 16;               // a split row of the A image (12 lines) + 16 B: conflict-free ds_read_b128
constexpr int SF_AIMG = 16 * SF_AROW;               // 24,832
constexpr int SF_RED = 4 * 16 * 17 * 4;             // the four waves' partial tiles
constexpr int SF_ATT = 2 * 128 * 128 + 512 * 4 + 16;  // attention_shx_body<1>: K | V images of 128 keys, mask of <= 512 keys
constexpr int SF_OFF_AIMG = SF_W_BYTES, SF_OFF_RED = SF_W_BYTES + SF_AIMG, SF_OFF_ATT = SF_W_BYTES;
constexpr int SF_LDS = SF_W_BYTES + (SF_ATT > SF_AIMG + SF_RED ? SF_ATT : SF_AIMG + SF_RED);  // 59,408
constexpr int SC1 = 16;                             // aux of the raw buffer intrinsics: sc1

// One tensor other blocks write or read inside the launch: every access is a buffer instruction with sc1.
struct Sc1Buf {
    __amdgpu_buffer_rsrc_t r;
    const char* base;
    __device__ __forceinline__ Sc1Buf(const void* p, size_t bytes)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)), base(static_cast<const char*>(p)) {}
    __device__ __forceinline__ uint32_t off(const void* p) const { return (uint32_t)(static_cast<const char*>(p) - base); }
    __device__ __forceinline__ f16x8 ld16(const _Float16* p) const {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off(p), 0, SC1));
    }
    __device__ __forceinline__ float ld4(const float* p) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off(p), 0, SC1));
    }
    __device__ __forceinline__ void st4(void* p, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b32(v, r, off(p), 0, SC1); }
    __device__ __forceinline__ void st4(float* p, float v) const { st4(static_cast<void*>(p), __builtin_bit_cast(uint32_t, v)); }
    __device__ __forceinline__ void st8(float* p, float2 v) const {
        __builtin_amdgcn_raw_buffer_store_b64(sf_u32x2{__builtin_bit_cast(uint32_t, v.x), __builtin_bit_cast(uint32_t, v.y)}, r, off(p), 0, SC1);
    }
    // attention_shx_body's policy: a piece of K / V (8 keys x 128 B) through registers into the lane-linear LDS image
    __device__ __forceinline__ f16x8 stage_load(const _Float16* src, char*) const { return ld16(src); }
    __device__ __forceinline__ void stage_store(char* lds_piece, int lane, f16x8 v) const { *reinterpret_cast<f16x8*>(lds_piece + lane * 16) = v; }
    __device__ __forceinline__ void st8(_Float16* p, f16x4 v) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sf_u32x2, v), r, off(p), 0, SC1);
    }
};

// which column tile (and which of the row tiles) of a dense layer with NT column tiles this block computes
struct SfMap {
    bool active;
    uint32_t nt, mt0, mt_step;
    __device__ __forceinline__ SfMap(uint32_t NT, uint32_t g) {
        const uint32_t groups = SF_GRID / NT;  // NT <= SF_GRID
        active = g / NT < groups;
        nt = g % NT;
        mt0 = g / NT;
        mt_step = groups;
    }
};

// This wave's share of the LDS-DMA that brings W rows [16 nt, 16 nt + 16) (kchunks lines each) into the weight image:
// [chunk][row][128 B], 16-B slot c at c ^ ((row >> 1) & 7) (permutation on the SOURCE address: the DMA destination is
// lane-linear).  Pieces of 8 rows x 128 B = 2 kchunks of them, dealt to waves 1..3 (wave 0's lane 0 polls the grid
// barrier behind this: its queue stays empty).
// c0, nc: the window of k-chunks brought in (image chunk c = row chunk c0 + c); kchunks: chunks per row of W.
__device__ __forceinline__ void sf_w_prefetch(char* lds, const _Float16* __restrict__ W, uint32_t nt, uint32_t kchunks, uint32_t c0,
                                              uint32_t nc, int wave, int lane) {
    if (wave == 0) return;
    const uint32_t pieces = 2 * nc;
    for (uint32_t p = wave - 1; p < pieces; p += 3) {
        const uint32_t c = p >> 1, row = (p & 1) * 8 + (lane >> 3);
        const uint32_t slot = (lane & 7) ^ ((row >> 1) & 7);
        sh_glds16(W + ((size_t)(16 * nt + row) * kchunks + c0 + c) * 64 + slot * 8, lds + p * 1024);
    }
}

// the three MFMAs of one k-chunk of a tile: gemm_sh_skinny_kernel's, with the W fragment read from the LDS image
__device__ __forceinline__ void sf_chunk_mma(const char* lds_w, uint32_t chunk, int l15, int g, f16x8 ah, f16x8 al,
                                             sh_f32x4v& hh, sh_f32x4v& xx) {
    const int swz = (l15 >> 1) & 7;
    const char* line = lds_w + ((size_t)chunk * 16 + l15) * 128;
    const f16x8 wh = *reinterpret_cast<const f16x8*>(line + ((g ^ swz) * 16));
    const f16x8 wl = *reinterpret_cast<const f16x8*>(line + (((4 + g) ^ swz) * 16));
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, hh, 0, 0, 0);
    xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, xx, 0, 0, 0);
    xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, xx, 0, 0, 0);
}

}  // namespace

// ---- the kernel --------------------------------------------------------------------------------------------------------
template <int NPL>
__global__ void __launch_bounds__(256, 1)
small_forward_kernel(SfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr uint32_t H = 64 * NPL, I = 4 * H, KC_H = H / 32, KC_I = I / 32;
    static_assert(NPL == 6, "the weight image and the fragment counts are sized for H = 384, I = 1536");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const uint32_t blk = blockIdx.x;
    const uint32_t T = a.T, MT = (T + 15) / 16;
    char* lds_w = lds;
    char* aimg = lds + SF_OFF_AIMG;
    float (*red)[16][17] = reinterpret_cast<float (*)[16][17]>(lds + SF_OFF_RED);
    __shared__ int s_abort;

    const Sc1Buf bX(a.X, (size_t)T * H * 4), bXA(a.XA, (size_t)T * H * 4), bY(a.Y, (size_t)T * H * 4), bP(a.PARTS, (size_t)4 * T * H * 4),
        bQ(a.QKVS, (size_t)T * 3 * H * 4), bC(a.CTXS, (size_t)T * H * 4), bM(a.MIDS, (size_t)T * I * 4);

    uint32_t target = 0;
    bool ovf = false;
    if (tid == 0) s_abort = 0;
    // diagnostics (a.dbg != null): 100 MHz ticks block `dbg_block` spent computing, draining its stores, and at the grid
    // barriers; written once at the end to a buffer nothing else reads
    uint64_t t_compute = 0, t_drain = 0, t_sync = 0, t_mark = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    uint64_t t_ph[5] = {0, 0, 0, 0, 0};  // compute ticks by phase kind: E2 | E3 | E4 | E5 | E6
    auto stamp = [&](uint64_t& bucket) {
        if (a.dbg) { const uint64_t t = __builtin_amdgcn_s_memrealtime(); bucket += t - t_mark; t_mark = t; }
    };

    // ---- grid barrier + the prefetch of the next phase's weight tile ----
    // Wnext: the next phase's weight ([NTnext * 16][kc_next chunks per row]); kslices != 0: that phase cuts K into four
    // slices by block group (FFN-down), each block needing only its slice of the tile's rows
    auto barrier = [&](const _Float16* Wnext, uint32_t NTnext, uint32_t kc_next, uint32_t code, bool kslices = false) -> bool {
        if (a.dbg) { const uint64_t t = __builtin_amdgcn_s_memrealtime(); t_ph[(code - 1) % 5 < 5 ? (code - 1) % 5 : 0] += t - t_mark; }
        stamp(t_compute);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY wave drains its write-through stores (R1)
        stamp(t_drain);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __syncthreads();                                  // ... and is done with this phase's LDS
        if (Wnext) {
            const SfMap m(NTnext, blk);
            if (m.active) {
                if (kslices) sf_w_prefetch(lds_w, Wnext, m.nt, kc_next, m.mt0 * (kc_next / 4), kc_next / 4, wave, lane);
                else sf_w_prefetch(lds_w, Wnext, m.nt, kc_next, 0, kc_next, wave, lane);
            }
        }
        target += SF_GRID;
        if (tid == 0) {
            __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0;
            while (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 16) || __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(a.sync + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // give up: every block leaves
                    s_abort = 1;
                    break;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the prefetch has landed
        __syncthreads();                                  // ... every wave's; the poll has matched (and s_abort is visible)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        asm volatile("" ::: "memory");
        stamp(t_sync);
        return s_abort == 0;
    };

    // ---- LayerNorm prologue: the 16 rows of m-tile mt -> the A image in LDS (split form); the leader also writes X ----
    // emb != 0: the rows are the embedding gather (word + type + position), else Y's.
    // src 0: Y's rows; 1: (the four FFN-down slabs in order + pbias) + X's row (layernorm_sum_kernel's order); 2: the embedding
    // gather.  The leader writes the normalised rows to `xout` / its accessor (never the buffer src 1 reads its residual from).
    auto ln_prologue = [&](uint32_t mt, const float* gw, const float* bw, int src, const float* pbias, bool leader, float* xout,
                           const Sc1Buf& bout) {
        // this wave's four rows: every load of all four in flight before the first is used (one memory latency, not four)
        float v[4][NPL];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const uint32_t t_raw = 16 * mt + 4 * wave + rr, t = t_raw < T ? t_raw : T - 1;
            if (src == 2) {
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
            } else if (src == 1) {
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    const int c = ln_col(lane, i);
                    float acc = bP.ld4(a.PARTS + (size_t)t * H + c);
#pragma unroll
                    for (uint32_t sl = 1; sl < 4; ++sl) acc += bP.ld4(a.PARTS + ((size_t)sl * T + t) * H + c);
                    v[rr][i] = (acc + pbias[c]) + bX.ld4(a.X + (size_t)t * H + c);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPL; ++i) v[rr][i] = bY.ld4(a.Y + (size_t)t * H + ln_col(lane, i));
            }
        }
        float ov[4][NPL];
        ln_rows_core<NPL, 4>(v, gw, bw, a.eps, lane, ov);  // (the four rows' reduction chains interleaved)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * wave + rr;
            const uint32_t t_raw = 16 * mt + row, t = t_raw < T ? t_raw : T - 1;
            const float (&o)[NPL] = ov[rr];
            char* irow = aimg + row * SF_AROW;
#pragma unroll
            for (int p = 0; p < NPL / 2; ++p) {
                const int c = ln_col(lane, 2 * p);
                f16x2 hi, lo;
                _Float16 x0, x1;
                ovf |= sh_split(o[2 * p], x0, x1); hi[0] = x0; lo[0] = x1;
                ovf |= sh_split(o[2 * p + 1], x0, x1); hi[1] = x0; lo[1] = x1;
                *reinterpret_cast<f16x2*>(irow + (c >> 5) * 128 + (c & 31) * 2) = hi;
                *reinterpret_cast<f16x2*>(irow + (c >> 5) * 128 + 64 + (c & 31) * 2) = lo;
                if (leader && t_raw < T) bout.st8(xout + (size_t)t * H + c, make_float2(o[2 * p], o[2 * p + 1]));
            }
        }
        __syncthreads();
    };

    // ---- one 16 x 16 output tile: gemm_sh_skinny_kernel<EPI, 1, 1> with W from the LDS image ----
    // PRO: A from the LDS image the prologue left; else from `Ag` (split rows, sc1).  U = kchunks / 4.
    // EPI: SH_OUT_SPLIT | SH_OUT_SPLIT_GELU (split rows -> Cs) | SH_OUT_F32_RESID (+ bias + XA's row -> Cf) | SH_OUT_PARTIAL (raw
    // sums of K slice `ks` -> slab ks of Cf: no bias).  A: the LDS image the prologue left (a_from_img) or split rows `Ag` of
    // akc chunks each, read from chunk ac0 on.  Three k-chunks per wave (K = 384, or a quarter of K = 1536).
    auto tile = [&](auto epi_tag, bool a_from_img, const Sc1Buf* bA, const _Float16* Ag, uint32_t akc, uint32_t ac0, uint32_t mt,
                    uint32_t nt, const float* bias, const Sc1Buf* bOut, uint32_t N, float* Cf, _Float16* Cs, uint32_t ks) {
        constexpr int U = 3;
        constexpr int EPI = decltype(epi_tag)::value;
        const uint32_t m0 = 16 * mt, n0 = 16 * nt;
        f16x8 ah[U], al[U];
        if (a_from_img) {
            const char* irow = aimg + l15 * SF_AROW + g * 16;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ah[u] = *reinterpret_cast<const f16x8*>(irow + (wave + 4 * u) * 128);
                al[u] = *reinterpret_cast<const f16x8*>(irow + (wave + 4 * u) * 128 + 64);
            }
        } else {
            const uint32_t r = m0 + l15;
            const _Float16* ap = Ag + ((size_t)(r < T ? r : T - 1) * akc + ac0) * 64 + 8 * g;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ah[u] = bA->ld16(ap + (size_t)(wave + 4 * u) * 64);
                al[u] = bA->ld16(ap + (size_t)(wave + 4 * u) * 64 + 32);
            }
        }
        // the epilogue's operands are requested now, in front of the MFMAs (behind the partial-tile exchange they were two
        // more dependent memory round trips per tile)
        const int m = tid >> 4, n = tid & 15;
        const uint32_t row = m0 + m, col = n0 + n;
        float bias_v = 0.0f, resid_v = 0.0f;
        if constexpr (EPI != SH_OUT_PARTIAL) bias_v = bias[col];
        if constexpr (EPI == SH_OUT_F32_RESID) resid_v = bXA.ld4(a.XA + (size_t)(row < T ? row : T - 1) * N + col);
        sh_f32x4v hh = {0.f, 0.f, 0.f, 0.f}, xx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u) sf_chunk_mma(lds_w, wave + 4 * u, l15, g, ah[u], al[u], hh, xx);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * g + r][l15] = fmaf(xx[r], kShLoInv, hh[r]);
        __syncthreads();
        float v = (red[0][m][n] + red[1][m][n]) + (red[2][m][n] + red[3][m][n]);
        if constexpr (EPI == SH_OUT_PARTIAL) {
            if (row < T) bOut->st4(Cf + ((size_t)ks * T + row) * N + col, v);
        } else if constexpr (EPI == SH_OUT_F32_RESID) {
            if (row < T) bOut->st4(Cf + (size_t)row * N + col, (v + bias_v) + resid_v);
        } else {
            v += bias_v;
            if (EPI == SH_OUT_SPLIT_GELU) v = sh_gelu_erf(v);
            _Float16 hi, lo;
            ovf |= sh_split(v, hi, lo);
            // two neighbouring columns per store: 4-byte write-through stores (R1 takes 4 / 8 / 16 B per lane)
            const uint32_t mine = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
            const uint32_t other = (uint32_t)__shfl_xor((int)mine, 1, 64);
            if (!(n & 1) && row < T) {
                _Float16* dst = Cs + ((size_t)row * (N / 32) + (col >> 5)) * 64 + (col & 31);
                bOut->st4(static_cast<void*>(dst), (mine & 0xffffu) | (other << 16));
                bOut->st4(static_cast<void*>(dst + 32), (mine >> 16) | (other & 0xffff0000u));
            }
        }
        __syncthreads();  // `red` (and the A image) are reused by the block's next tile
    };

    using E_SPLIT = std::integral_constant<int, SH_OUT_SPLIT>;
    using E_GELU = std::integral_constant<int, SH_OUT_SPLIT_GELU>;
    using E_RESID = std::integral_constant<int, SH_OUT_F32_RESID>;
    using E_PART = std::integral_constant<int, SH_OUT_PARTIAL>;

    constexpr uint32_t NT_QKV = 3 * H / 16, NT_H = H / 16, NT_I = I / 16;
    const float scale_log2e = (1.0f / sqrtf(32.0f)) * kLog2e;

    // the first phase's weight tile
    {
        const SfMap m(NT_QKV, blk);
        if (m.active) sf_w_prefetch(lds_w, a.layers[0].wqkv, m.nt, KC_H, 0, KC_H, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    for (uint32_t l = 0; l < a.n_layers; ++l) {
        const SfLayer ly = a.layers[l];
        // ---- E2: the previous layer's last LayerNorm (over its FFN-down slabs + bias + residual) — or the embedding — as
        // prologue, QKV projection -> QKVS; the leader leaves the normalised rows in XA ----
        {
            const SfMap m(NT_QKV, blk);
            if (m.active) {
                const float* gw = l ? a.layers[l - 1].ln2_g : a.emb_g;
                const float* bw = l ? a.layers[l - 1].ln2_b : a.emb_b;
                const float* pb = l ? a.layers[l - 1].bdown : nullptr;
                for (uint32_t mt = m.mt0; mt < MT; mt += m.mt_step) {
                    ln_prologue(mt, gw, bw, l ? 1 : 2, pb, m.nt == 0, a.XA, bXA);
                    tile(E_SPLIT{}, true, nullptr, nullptr, 0, 0, mt, m.nt, ly.bqkv, &bQ, 3 * H, nullptr, a.QKVS, 0);
                }
            }
        }
        if (!barrier(ly.wo, NT_H, KC_H, 5 * l + 1)) return;
        // ---- E3: attention (the out-proj weight tile sits in the weight image meanwhile) ----
        {
            const uint32_t hgroups = a.heads / a.hb, qblocks = (a.L + 127) / 128;
            const uint32_t tasks = hgroups * a.B * qblocks;
            for (uint32_t t = blk; t < tasks; t += SF_GRID) {
                const uint32_t bx = t % hgroups, by = (t / hgroups) % a.B, bz = t / (hgroups * a.B);
                attention_shx_body<1>(lds + SF_OFF_ATT, bQ, bC, a.QKVS, a.mask, a.CTXS, a.flag, a.L, H, scale_log2e, a.hb, nullptr,
                                      nullptr, nullptr, bx, by, bz, hgroups, qblocks);
                __syncthreads();
            }
        }
        if (!barrier(nullptr, 0, 0, 5 * l + 2)) return;
        // ---- E4: out-proj + bias + residual (XA) -> Y ----
        {
            const SfMap m(NT_H, blk);
            if (m.active)
                for (uint32_t mt = m.mt0; mt < MT; mt += m.mt_step)
                    tile(E_RESID{}, false, &bC, a.CTXS, KC_H, 0, mt, m.nt, ly.bo, &bY, H, a.Y, nullptr, 0);
        }
        if (!barrier(ly.wup, NT_I, KC_H, 5 * l + 3)) return;
        // ---- E5: the attention block's LayerNorm as prologue (the leader leaves the rows in X), FFN-up + GELU -> MIDS ----
        {
            const SfMap m(NT_I, blk);
            if (m.active)
                for (uint32_t mt = m.mt0; mt < MT; mt += m.mt_step) {
                    ln_prologue(mt, ly.ln1_g, ly.ln1_b, 0, nullptr, m.nt == 0, a.X, bX);
                    tile(E_GELU{}, true, nullptr, nullptr, 0, 0, mt, m.nt, ly.bup, &bM, I, nullptr, a.MIDS, 0);
                }
        }
        if (!barrier(ly.wdown, NT_H, KC_I, 5 * l + 4, true)) return;
        // ---- E6: FFN-down as four K slices (block group = slice) -> PARTS; summed by the next LayerNorm ----
        {
            const SfMap m(NT_H, blk);  // groups = 4: m.mt0 is the K slice here, every block walks all row tiles
            if (m.active)
                for (uint32_t mt = 0; mt < MT; ++mt)
                    tile(E_PART{}, false, &bM, a.MIDS, KC_I, m.mt0 * (KC_I / 4), mt, m.nt, nullptr, &bP, H, a.PARTS, nullptr, m.mt0);
        }
        const bool last = l + 1 == a.n_layers;
        if (!barrier(last ? nullptr : a.layers[l + 1].wqkv, NT_QKV, KC_H, 5 * l + 5)) return;
    }
    // ---- the last LayerNorm: (slabs + bias) + X -> X, in place (the last hidden state the pooling kernel reads) ----
    {
        const SfLayer ly = a.layers[a.n_layers - 1];
        for (uint32_t t = 4 * blk + wave; t < T; t += 4 * SF_GRID) {
            float v[NPL], o[NPL];
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const int c = ln_col(lane, i);
                float acc = bP.ld4(a.PARTS + (size_t)t * H + c);
#pragma unroll
                for (uint32_t sl = 1; sl < 4; ++sl) acc += bP.ld4(a.PARTS + ((size_t)sl * T + t) * H + c);
                v[i] = (acc + ly.bdown[c]) + bX.ld4(a.X + (size_t)t * H + c);
            }
            ln_row_core<NPL>(v, ly.ln2_g, ly.ln2_b, a.eps, lane, o);
#pragma unroll
            for (int p = 0; p < NPL / 2; ++p)
                *reinterpret_cast<float2*>(a.X + (size_t)t * H + ln_col(lane, 2 * p)) = make_float2(o[2 * p], o[2 * p + 1]);
        }
    }
    if (ovf && a.flag) atomicOr(a.flag, 1u);
    if (a.dbg && tid == 0 && blk < 96) {
        stamp(t_compute);
        a.dbg[3 * blk] = t_compute; a.dbg[3 * blk + 1] = t_drain; a.dbg[3 * blk + 2] = t_sync;
        if (blk == 0) for (int k = 0; k < 5; ++k) a.dbg[3 * 96 + k] = t_ph[k];
    }
}

bool small_forward_supported(uint32_t H, uint32_t I, uint32_t heads, uint32_t T, uint32_t L) {
    return H == 384 && I == 1536 && heads * 32 == H && T >= 1 && T <= SF_MAX_ROWS && L >= 1 && L <= 512;
}

int32_t launch_small_forward(const SfArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_set;  // function attributes are per device
    static int cus = 0;
    CS_TRY(attr_set.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(small_forward_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS));
        int dev = 0;
        CS_HIP(hipGetDevice(&dev));
        CS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        return CS_OK;
    }));
    // every block must be resident (SF_LDS = 59,408 B each; checked here only as one CU per block — another stream, replica or
    // process holding CUs can still leave blocks unscheduled: each barrier then gives up after 2^16 polls, ~0.1 s, and the caller
    // falls back to the launch-per-operator path) or the grid barrier cannot complete
    if (cus < SF_GRID) return fail(CS_ERR_UNSUPPORTED, "the one-launch forward needs %d compute units (device has %d)", SF_GRID, cus);
    hipLaunchKernelGGL(small_forward_kernel<6>, dim3(SF_GRID), dim3(256), SF_LDS, s, a);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
