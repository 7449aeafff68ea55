// attention_split.hip — multi-head self-attention (head_dim 32; 64 in attention_shx_kernel) on the f16 MFMA with split-f16
// operands (split_f16.hpp), SURVEY.md §8a E3.  Replaces the MHA nodes of the ONNX graph that
// /root/reference/src/embed/embedder.rs:286-289 runs through fastembed/ort.
//
// Every wave walks query tiles of 32 over key tiles of 32 staged in LDS (layouts at each kernel):
//   S^T (keys x queries) = K_tile (A: [key][d]) x Q^T (B: [d][query])      6 MFMAs (2 k-steps x hh, hl, lh)
//   softmax in the lane: the query is the lane, its 16 keys of the tile are registers; online
//   max / sum in the exp2 domain (scores pre-multiplied by log2 e), rescale only when a max moves
//   O^T (d x queries) += V_tile^T (A: [d][key]) x P^T (B: [key][query])     6 MFMAs
// P^T is the S^T accumulator itself: its key index sits on (register, lane half) in exactly the
// order the next MFMA's k index wants (cdna_hip_programming.md §3, "An accumulator tile as the next
// MFMA's operand"), so probabilities are split to f16 pairs in registers and never touch LDS.
// Input is the QKV GEMM's split-f16 output (SH_OUT_SPLIT); round 1's kernel that took an f32 qkv and split
// K / V itself in a converting prologue (36 % of a block's life) is no longer built.
#include <cstdlib>
#include "encoder.hpp"
#include "split_f16.hpp"
#include "attention_shx_body.hpp"

// Diagnostic builds (benchmarks/attn_probe.hip) define AT_STAMP to record s_memtime at points of a
// block's life; the product build compiles it to nothing.
#ifndef AT_STAMP
#define AT_STAMP(i)
#endif

namespace cs {

#ifndef CS_ATTN_PIPE_WAVES
#define CS_ATTN_PIPE_WAVES 3   // waves per SIMD the head_dim-32 two-tile kernel is compiled for (168 registers)
#endif

// defaults of CS_ATTN_PIPE per head width (same-box A/B of the four forms: profiles/r05_attention_loop_forms_ab.log)
constexpr int kAttnPipeDefault32 = 2, kAttnPipeDefault64 = 3;

#ifdef CS_DIAGNOSTICS  // (the whole-sequence kernel: CS_ATTN_SHX1=0, A/B only — the product library runs the super-tile kernel below)
// ---- whole sequence of one (sequence, head) resident in LDS (CS_ATTN_SHX1=0; head_dim 32) ----------
// head_dim 32 = one k-chunk, so K and V of a (token, head) are one 128-B line [32 hi | 32 lo] each,
// exactly the layout the matrix pipe wants for K: the prologue is pure LDS-DMA (no conversion, no
// ds_write).  V stays [key][d] in LDS
// and is consumed as the A operand V^T through the transposing read ds_read_b64_tr_b16: per
// 16-lane group it turns a 4-key x 16-d block into "lane d holds its 4 keys" (verified on the
// hardware with benchmarks/tr_probe).  LDS images: K pieces (16 B) at c ^ ((key >> 1) & 7) for
// conflict-free ds_read_b128; V pieces at c ^ (4 * ((key >> 1) & 1)) so the four key rows and
// four pieces a half-wave's transposing read touches cover all 64 banks once.
__global__ void __launch_bounds__(256, 2)
attention_sh2_kernel(const _Float16* __restrict__ qkvs, const int32_t* __restrict__ mask,
                     _Float16* __restrict__ ctxs, uint32_t* __restrict__ flag, uint32_t L, uint32_t H,
                     float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t Lp = (L + 31) & ~31u;
    char* Kt = smem;                            // [Lp][128 B]
    char* Vt = Kt + (size_t)Lp * 128;           // [Lp][128 B]
    float* madd = reinterpret_cast<float*>(Vt + (size_t)Lp * 128);  // [Lp]
    int* last_valid_p = reinterpret_cast<int*>(madd + Lp);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t head = blockIdx.x, b = blockIdx.y;
    const uint32_t nh = H / 32, nch = 3 * nh;  // chunks per token row: Q heads | K heads | V heads
    const _Float16* base = qkvs + (size_t)b * L * nch * 64;
    bool ovf = false;

    AT_STAMP(0);
    if (tid == 0) *last_valid_p = 0;
    for (uint32_t ii = wave; ii < Lp / 8; ii += 4) {  // 8 keys x 128 B per instruction and operand
        const uint32_t row = ii * 8 + (lane >> 3);
        const uint32_t key = row < L ? row : L - 1;   // keys past L are masked; read a valid line
        const uint32_t ck = (lane & 7) ^ ((row >> 1) & 7);
        const uint32_t cv = (lane & 7) ^ (4 * ((row >> 1) & 1));
        sh_glds16(base + ((size_t)key * nch + nh + head) * 64 + ck * 8, Kt + ii * 1024);
        sh_glds16(base + ((size_t)key * nch + 2 * nh + head) * 64 + cv * 8, Vt + ii * 1024);
    }
    __syncthreads();  // (last_valid_p initialised)
    for (uint32_t key = tid; key < Lp; key += 256) {
        const bool ok = key < L && mask[(size_t)b * L + key] != 0;
        madd[key] = ok ? 0.0f : kMaskedLog2;
        if (ok) atomicMax(last_valid_p, (int)key);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t ntiles = (uint32_t)(*last_valid_p) / 32 + 1;
    AT_STAMP(1);

    const int kswz = (l31 >> 1) & 7;
    const char* k_lane = Kt + l31 * 128;
    int k_hi[2], k_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        k_hi[s] = ((2 * s + h) ^ kswz) * 16;
        k_lo[s] = ((4 + 2 * s + h) ^ kswz) * 16;
    }
    // transposing read of V: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of the
    // group's 4-key x 16-d block; group G = lane>>4: d0 = 16 (G&1), keys 4h + q (h = G>>1 = lane>>5)
    const int vq = (lane & 15) >> 2, vp = lane & 3, vg = (lane >> 4) & 1;
    const int vfv = 4 * ((vq >> 1) & 1);
    const int v_hi = (4 * h + vq) * 128 + (((2 * vg + (vp >> 1)) ^ vfv) * 16) + 8 * (vp & 1);
    const int v_lo = v_hi ^ 64;  // logical piece + 4 (the lo half of the line) under the same XOR

    for (uint32_t qb = 0; qb * 128 < L; ++qb) {
        if (qb * 128 + wave * 32 >= L) break;  // wave-uniform: no query of this wave's tile exists
        const uint32_t query = qb * 128 + wave * 32 + l31;
        const uint32_t qsrc = query < L ? query : L - 1;
        // Q^T fragments (B operand): element j of step s = d 16s + 8h + j, straight from the split line
        const _Float16* qp = base + ((size_t)qsrc * nch + head) * 64 + 8 * h;
        f16x8 qh[2], ql[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qh[s] = *reinterpret_cast<const f16x8*>(qp + 16 * s);
            ql[s] = *reinterpret_cast<const f16x8*>(qp + 32 + 16 * s);
        }
        sh_f32x16 ohh, oxx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ohh[r] = 0.0f; oxx[r] = 0.0f; }
        float m = -__builtin_huge_valf(), lsum = 0.0f;

        for (uint32_t kt = 0; kt < ntiles; ++kt) {
            sh_f32x16 hh, xx;
#pragma unroll
            for (int r = 0; r < 16; ++r) { hh[r] = 0.0f; xx[r] = 0.0f; }
            const char* kr = k_lane + (size_t)kt * 32 * 128;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(kr + k_hi[s]);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(kr + k_lo[s]);
                hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], hh, 0, 0, 0);
                xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], xx, 0, 0, 0);
                xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], xx, 0, 0, 0);
            }
            float tmax = -__builtin_huge_valf();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const sh_f32x4 ma = *reinterpret_cast<const sh_f32x4*>(madd + kt * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    hh[r] = fmaf(fmaf(xx[r], kShLoInv, hh[r]), scale_log2e, ma[e]);
                    tmax = fmaxf(tmax, hh[r]);
                }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            if (__any(tmax > m)) {
                const float mnew = fmaxf(m, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                lsum *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { ohh[r] *= alpha; oxx[r] *= alpha; }
                m = mnew;
            }
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hh[r] = __builtin_amdgcn_exp2f(hh[r] - m);
                psum += hh[r];
            }
            lsum += psum;
            Frag8 ph[2], pl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2)
                    split_pair_rtz_ng(hh[8 * s + 2 * w2], hh[8 * s + 2 * w2 + 1], ph[s].u[w2], pl[s].u[w2]);
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
            const char* vr = Vt + (size_t)kt * 32 * 128;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // element j of the V^T fragment = key 16s + 8 (j>>2) + 4h + (j&3): two 4-key blocks
                FragTr vh, vl;
                vh.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vr + s * 16 * 128 + v_hi));
                vh.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vr + s * 16 * 128 + 8 * 128 + v_hi));
                vl.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vr + s * 16 * 128 + v_lo));
                vl.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vr + s * 16 * 128 + 8 * 128 + v_lo));
                ohh = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, ph[s].v, ohh, 0, 0, 0);
                oxx = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh.v, pl[s].v, oxx, 0, 0, 0);
                oxx = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl.v, ph[s].v, oxx, 0, 0, 0);
            }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / lsum;
        if (query < L) {
            _Float16* op = ctxs + (((size_t)b * L + query) * nh + head) * 64 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a, bb;
                    ovf |= sh_split(fmaf(oxx[4 * g + e], kShLoInv, ohh[4 * g + e]) * inv, a, bb);
                    hi[e] = a; lo[e] = bb;
                }
                *reinterpret_cast<f16x4*>(op + 8 * g) = hi;
                *reinterpret_cast<f16x4*>(op + 32 + 8 * g) = lo;
            }
        }
    }
    AT_STAMP(2);
    if (ovf && flag) atomicOr(flag, 1u);
}
#endif  // CS_DIAGNOSTICS

// ---- head_dim 32 * NC (NC = 2: BGE-base / BGE-large / mxbai-large): keys in super-tiles of 128 ------------
// The body lives in attention_shx_body.hpp (shared with small_forward.hip).
template <int NC, int POS = 0, int PIPE = 0>
__global__ void __launch_bounds__(256, NC == 1 ? (PIPE == 1 ? CS_ATTN_PIPE_WAVES : (PIPE >= 2 ? 4 : 2)) : 2)
attention_shx_kernel(const _Float16* __restrict__ qkvs, const int32_t* __restrict__ mask,
                     _Float16* __restrict__ ctxs, uint32_t* __restrict__ flag, uint32_t L, uint32_t H,
                     float scale_log2e, uint32_t HB, float* __restrict__ range_out,
                     const uint32_t* __restrict__ seq_unit, const uint32_t* __restrict__ unit_len,
                     const float* __restrict__ alibi_log2, uint32_t window) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AttnMemPlain mem;
    attention_shx_body<NC, AttnMemPlain, AttnMemPlain, POS, PIPE>(smem, mem, mem, qkvs, mask, ctxs, flag, L, H, scale_log2e, HB, range_out,
                                                            seq_unit, unit_len, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x,
                                                            gridDim.z, alibi_log2, window);
}

namespace {

struct ShxArgs {
    const _Float16* qkv; const int32_t* mask; _Float16* ctx; uint32_t* flag; uint32_t L, H; float scale_log2e; uint32_t hb;
    float* range_out; const uint32_t* seq_unit; const uint32_t* unit_len; const float* alibi_log2; uint32_t window;
};

template <int NC, int POS, int PIPE>
int32_t launch_shx_one(dim3 grid, size_t lds, hipStream_t s, const ShxArgs& a) {
    if (lds > 64 * 1024 - 256) {  // above the default dynamic-LDS limit: the attribute is per function and per device
        static PerDeviceOnce attr;
        CS_TRY(attr.run([&]() -> int32_t {
            CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_shx_kernel<NC, POS, PIPE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
            return CS_OK;
        }));
    }
    hipLaunchKernelGGL((attention_shx_kernel<NC, POS, PIPE>), grid, dim3(256), lds, s, a.qkv, a.mask, a.ctx, a.flag, a.L, a.H,
                       a.scale_log2e, a.hb, a.range_out, a.seq_unit, a.unit_len, a.alibi_log2, a.window);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

template <int NC>
int32_t launch_shx(int pos, int pipe, dim3 grid, size_t lds, hipStream_t s, const ShxArgs& a) {
#ifndef CS_DIAGNOSTICS
    // the product library holds each head width's default form only (the other tile-loop forms are bit-identical and slower:
    // profiles/r05_attention_loop_forms_ab.log; the diagnostic library keeps them for CS_ATTN_PIPE)
    constexpr int P = NC == 1 ? kAttnPipeDefault32 : kAttnPipeDefault64;
    (void)pipe;
    if (pos == 1) return launch_shx_one<NC, 1, P>(grid, lds, s, a);
    if (pos == 2) return launch_shx_one<NC, 2, P>(grid, lds, s, a);
    return launch_shx_one<NC, 0, P>(grid, lds, s, a);
#else
    if (pipe == 3) {
        if (pos == 1) return launch_shx_one<NC, 1, 3>(grid, lds, s, a);
        if (pos == 2) return launch_shx_one<NC, 2, 3>(grid, lds, s, a);
        return launch_shx_one<NC, 0, 3>(grid, lds, s, a);
    }
    if (pipe == 2) {
        if (pos == 1) return launch_shx_one<NC, 1, 2>(grid, lds, s, a);
        if (pos == 2) return launch_shx_one<NC, 2, 2>(grid, lds, s, a);
        return launch_shx_one<NC, 0, 2>(grid, lds, s, a);
    }
    if (pipe == 1) {
        if (pos == 1) return launch_shx_one<NC, 1, 1>(grid, lds, s, a);
        if (pos == 2) return launch_shx_one<NC, 2, 1>(grid, lds, s, a);
        return launch_shx_one<NC, 0, 1>(grid, lds, s, a);
    }
    if (pos == 1) return launch_shx_one<NC, 1, 0>(grid, lds, s, a);
    if (pos == 2) return launch_shx_one<NC, 2, 0>(grid, lds, s, a);
    return launch_shx_one<NC, 0, 0>(grid, lds, s, a);
#endif
}

}  // namespace

int32_t launch_attention_sh2(const _Float16* qkv_split, const int32_t* mask, void* ctx_split, uint32_t* flag,
                             uint32_t B, uint32_t L, uint32_t H, uint32_t heads, hipStream_t s, float* range_out,
                             uint32_t* range_pairs, const uint32_t* seq_unit, const uint32_t* unit_len, const float* alibi,
                             uint32_t window) {
    if (alibi && window) return fail(CS_ERR_BAD_ARG, "attention: ALiBi and a local window together are not built");
    // alibi (CS_ARCH_JINA*): [2][heads] — the slopes, then the slopes times log2 e (what these kernels add in the exp2 domain)
    const float* alibi_log2 = alibi ? alibi + heads : nullptr;
    const int pos = alibi_log2 ? 1 : (window ? 2 : 0);
    if (range_pairs) *range_pairs = 0;
    const uint32_t dh = heads ? H / heads : 0;
    if ((dh != 32 && dh != 64) || H % heads)
        return fail(CS_ERR_UNSUPPORTED, "head_dim %u not supported (32 or 64)", dh);
    const size_t Lp = (L + 31) & ~31u;
    // CS_ATTN_PIPE (attention_shx_body.hpp, PIPE): 0 the rolled tile loop, 1 two key tiles in flight per wave, 2 the super-tile's
    // four tiles written out (immediate LDS offsets, cross-half max on the VALU)
    static const int pipe_env = [] { const char* e = cs_lab_env("CS_ATTN_PIPE"); return e && e[0] >= '0' && e[0] <= '3' ? e[0] - '0' : -1; }();
    ShxArgs a{qkv_split, mask, static_cast<_Float16*>(ctx_split), flag, L, H, 0.0f, 1u, range_out, seq_unit, unit_len, alibi_log2, window};
    if (dh == 64) {  // two 128-B lines per (token, head): keys staged 128 at a time
        const size_t lds = 2 * 2 * 128 * 128 + Lp * sizeof(float) + 16;
        a.scale_log2e = (1.0f / sqrtf(64.0f)) * kLog2e;
        const int pipe = pipe_env < 0 ? kAttnPipeDefault64 : pipe_env;
        CS_TRY(launch_shx<2>(pos, pipe, dim3(heads, B, (L + 127) / 128), lds, s, a));
        if (range_pairs) *range_pairs = heads * B * ((L + 127) / 128) * 4;
        return CS_OK;
    }
    // head_dim 32 also runs on the 128-key super-tile kernel by default: 32 KiB of LDS per block instead of
    // 64 (four blocks per CU, one per 128 queries, so K/V staging of one block hides behind the softmax of
    // the others) — 256 x 256 tokens 12.75 -> 12.54 ms per forward although K/V are staged once per query
    // block.  CS_ATTN_SHX1=0 selects the whole-sequence kernel (attention_sh2_kernel) for A/B.
    static const bool shx1 = [] { const char* e = cs_lab_env("CS_ATTN_SHX1"); return !(e && e[0] == '0'); }();
    if (shx1 || pos) {
        // CS_ATTN_LDS_PAD (diagnostics): extra dynamic LDS per block, i.e. fewer co-resident blocks per CU — how the kernel's
        // time moves with occupancy says whether a tile's dependent chain (latency) or issue slots bound it
        static const size_t lds_pad = [] { const char* e = cs_lab_env("CS_ATTN_LDS_PAD"); return e ? (size_t)std::atol(e) : (size_t)0; }();
        const size_t lds1 = 2 * 1 * 128 * 128 + Lp * sizeof(float) + 16 + lds_pad;
        static const bool pack_heads = [] { const char* e = cs_lab_env("CS_ATTN_PACK_HEADS"); return !(e && e[0] == '0'); }();
        uint32_t hb = !pack_heads ? 1u : (Lp <= 32 ? 4u : (Lp <= 64 ? 2u : 1u));  // heads per block (kernel comment)
        while (heads % hb) hb >>= 1;
        a.scale_log2e = (1.0f / sqrtf(32.0f)) * kLog2e;
        a.hb = hb;
        const int pipe = pipe_env < 0 ? kAttnPipeDefault32 : pipe_env;
        CS_TRY(launch_shx<1>(pos, pipe, dim3(heads / hb, B, (L + 127) / 128), lds1, s, a));
        if (range_pairs) *range_pairs = (heads / hb) * B * ((L + 127) / 128) * 4;
        return CS_OK;
    }
#ifndef CS_DIAGNOSTICS
    return fail(CS_ERR_UNSUPPORTED, "attention: no kernel for head_dim %u", dh);  // (not reached: shx1 is constant here)
#else
    const size_t lds = 2 * Lp * 128 + Lp * sizeof(float) + 16;
    if (lds > 160 * 1024 - 64) return fail(CS_ERR_UNSUPPORTED, "sequence length %u exceeds the LDS-resident K/V limit", L);
    static PerDeviceOnce attr_set;  // function attributes are per device
    CS_TRY(attr_set.run([&]() -> int32_t {
        CS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_sh2_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        return CS_OK;
    }));
    hipLaunchKernelGGL(attention_sh2_kernel, dim3(heads, B), dim3(256), lds, s, qkv_split, mask,
                       static_cast<_Float16*>(ctx_split), flag, L, H, (1.0f / sqrtf(32.0f)) * kLog2e);
    CS_HIP(hipGetLastError());
    return CS_OK;
#endif
}

}  // namespace cs
