// nomic.hip — what the NomicBert encoder (cs_bert_config.arch == CS_ARCH_NOMIC: the reference registry's nomic-embed-text
// entries, /root/reference/src/embed/embedder.rs:30-35, :64-66) does that BERT does not: the rotary position map on Q and K
// behind the QKV projection, and the gate of its feed-forward,  fc2( fc11(x) * silu(fc12(x)) ).  Both are element-wise passes
// between the dense layers of gemm_wide.hip / gemm_split.hip (the up projection is ONE GEMM over the [2I, H] weight of
// fc11's and fc12's rows interleaved in groups of 16; at indexing batch sizes the gate is that GEMM's epilogue,
// GW_OUT_SWIGLU, and the stand-alone kernel below serves the smaller launches); each exists for the split-f16 tensors of the default path (split_f16.hpp: a 32-column chunk
// of a row is one 128-B line, 32 hi then 32 lo) and for the plain f32 tensors of the exact path.  HBM-bound: 16-byte
// accesses, one thread per eight neighbouring elements.
//
// Rotary map (non-interleaved, the model's rotary_emb_interleaved = false): a head's d_h columns are two halves x1 | x2 and
//   (x1_i, x2_i) -> (x1_i cos a - x2_i sin a,  x2_i cos a + x1_i sin a),   a = pos * base^(-2i / d_h),  i < d_h / 2
// with cos / sin from a table the host fills in f32 exactly as the module builds its cache (cs_embedder_create).  With
// d_h = 64 the two halves are two neighbouring lines of the split tensor, with d_h = 32 the two halves of one line.
//
// CS_ARCH_JINA / CS_ARCH_JINA_QKNORM (JinaBert: the registry's jina-embeddings-v2-base-code, embedder.rs:40-41) share the gate
// kernels with the erf-GELU in the silu's place ( down( value * gelu(gate) ) ), and _QKNORM adds the LayerNorm its attention
// applies to the whole query row and the whole key row before they are cut into heads: one wave per (token, Q | K).
#include "encoder.hpp"
#include "split_f16.hpp"

namespace cs {

namespace {

__device__ __forceinline__ void unsplit8(const f16x8 hi, const f16x8 lo, sh_f32x4& a, sh_f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = fmaf((float)lo[e], kShLoInv, (float)hi[e]);          // exact: hi + lo / 2048 has at most 24 significant bits
        b[e] = fmaf((float)lo[4 + e], kShLoInv, (float)hi[4 + e]);
    }
}

// rope [L][half] float2 (cos, sin).  One thread: eight pair indices i0 .. i0 + 7 of one (token, Q | K, head).
template <int DH>
__global__ void __launch_bounds__(256)
rope_split_kernel(_Float16* __restrict__ qkvs, const float2* __restrict__ rope, uint32_t T, uint32_t L, uint32_t H,
                  uint32_t* __restrict__ flag) {
    constexpr int HALF = DH / 2, TPH = HALF / 8;  // threads per head: 4 | 2
    const uint32_t heads = H / DH;
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t total = (uint64_t)T * 2 * heads * TPH;
    if (gid >= total) return;
    const uint32_t part = (uint32_t)(gid % TPH);
    const uint32_t head = (uint32_t)((gid / TPH) % heads);
    const uint32_t which = (uint32_t)((gid / TPH / heads) & 1);  // 0 = Q, 1 = K
    const uint64_t t = gid / TPH / heads / 2;
    const uint32_t col1 = which * H + head * DH + part * 8;  // first column of x1's eight values; x2's: + HALF
    const uint32_t col2 = col1 + HALF;
    _Float16* row = qkvs + t * (uint64_t)(3 * H) * 2;
    _Float16* p1 = row + (col1 >> 5) * 64 + (col1 & 31);
    _Float16* p2 = row + (col2 >> 5) * 64 + (col2 & 31);
    const f16x8 h1 = *reinterpret_cast<const f16x8*>(p1), l1 = *reinterpret_cast<const f16x8*>(p1 + 32);
    const f16x8 h2 = *reinterpret_cast<const f16x8*>(p2), l2 = *reinterpret_cast<const f16x8*>(p2 + 32);
    sh_f32x4 a1, b1, a2, b2;
    unsplit8(h1, l1, a1, b1);
    unsplit8(h2, l2, a2, b2);
    const float2* cs = rope + (size_t)(t % L) * HALF + part * 8;
    sh_f32x4 o1a, o1b, o2a, o2b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float2 ca = cs[e], cb = cs[4 + e];
        o1a[e] = a1[e] * ca.x - a2[e] * ca.y;
        o2a[e] = a2[e] * ca.x + a1[e] * ca.y;
        o1b[e] = b1[e] * cb.x - b2[e] * cb.y;
        o2b[e] = b2[e] * cb.x + b1[e] * cb.y;
    }
    f16x8 oh, ol;
    uint32_t mx = 0;
    sh_split8(o1a, o1b, oh, ol, mx);
    *reinterpret_cast<f16x8*>(p1) = oh;
    *reinterpret_cast<f16x8*>(p1 + 32) = ol;
    sh_split8(o2a, o2b, oh, ol, mx);
    *reinterpret_cast<f16x8*>(p2) = oh;
    *reinterpret_cast<f16x8*>(p2 + 32) = ol;
    if (sh_split_overflowed(mx) && flag) atomicOr(flag, 1u);
}

// the same on f32 rows [T][3H]: one thread, four pair indices
__global__ void __launch_bounds__(256)
rope_f32_kernel(float* __restrict__ qkv, const float2* __restrict__ rope, uint32_t T, uint32_t L, uint32_t H, uint32_t DH) {
    const uint32_t half = DH / 2, tph = half / 4, heads = H / DH;
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (uint64_t)T * 2 * heads * tph) return;
    const uint32_t part = (uint32_t)(gid % tph);
    const uint32_t head = (uint32_t)((gid / tph) % heads);
    const uint32_t which = (uint32_t)((gid / tph / heads) & 1);
    const uint64_t t = gid / tph / heads / 2;
    float* p1 = qkv + t * (uint64_t)(3 * H) + which * H + head * DH + part * 4;
    float* p2 = p1 + half;
    const sh_f32x4 x1 = *reinterpret_cast<const sh_f32x4*>(p1), x2 = *reinterpret_cast<const sh_f32x4*>(p2);
    const float2* cs = rope + (size_t)(t % L) * half + part * 4;
    sh_f32x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float2 c = cs[e];
        o1[e] = x1[e] * c.x - x2[e] * c.y;
        o2[e] = x2[e] * c.x + x1[e] * c.y;
    }
    *reinterpret_cast<sh_f32x4*>(p1) = o1;
    *reinterpret_cast<sh_f32x4*>(p2) = o2;
}

__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
template <bool GELU> __device__ __forceinline__ float gate_act(float v) { return GELU ? gelu_exact(v) : silu(v); }

// up2 [T][2I/32][64], columns interleaved in groups of 16 (GW_OUT_SWIGLU's weight order, encoder.hpp): line u of a row holds
// values 16 u .. 16 u + 15 and their gates — hi: [16 v | 16 g], lo: [16 v | 16 g] -> out [T][I/32][64].  One thread: eight
// gated columns.
template <bool GELU>
__global__ void __launch_bounds__(256)
swiglu_split_kernel(const _Float16* __restrict__ up2, _Float16* __restrict__ out, uint32_t T, uint32_t I,
                    uint32_t* __restrict__ flag) {
    const uint32_t per_row = I / 8;
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (uint64_t)T * per_row) return;
    const uint64_t t = gid / per_row;
    const uint32_t col = (uint32_t)(gid % per_row) * 8;  // gated column: line col / 16 of the raw row, offset col % 16
    const _Float16* pv = up2 + t * (uint64_t)(2 * I) * 2 + (size_t)(col >> 4) * 64 + (col & 15);
    const _Float16* pg = pv + 16;
    sh_f32x4 va, vb, ga, gb;
    unsplit8(*reinterpret_cast<const f16x8*>(pv), *reinterpret_cast<const f16x8*>(pv + 32), va, vb);
    unsplit8(*reinterpret_cast<const f16x8*>(pg), *reinterpret_cast<const f16x8*>(pg + 32), ga, gb);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        va[e] = va[e] * gate_act<GELU>(ga[e]);
        vb[e] = vb[e] * gate_act<GELU>(gb[e]);
    }
    f16x8 oh, ol;
    uint32_t mx = 0;
    sh_split8(va, vb, oh, ol, mx);
    _Float16* po = out + t * (uint64_t)I * 2 + (col >> 5) * 64 + (col & 31);
    *reinterpret_cast<f16x8*>(po) = oh;
    *reinterpret_cast<f16x8*>(po + 32) = ol;
    if (sh_split_overflowed(mx) && flag) atomicOr(flag, 1u);
}

// value [T][I] *= silu(gate [T][I])
template <bool GELU>
__global__ void __launch_bounds__(256)
swiglu_f32_kernel(float* __restrict__ value, const float* __restrict__ gate, uint64_t n4) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    sh_f32x4 v = reinterpret_cast<const sh_f32x4*>(value)[gid];
    const sh_f32x4 g = reinterpret_cast<const sh_f32x4*>(gate)[gid];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * gate_act<GELU>(g[e]);
    reinterpret_cast<sh_f32x4*>(value)[gid] = v;
}

// LayerNorm over the H query (which = 0) or key (which = 1) columns of one token row of the QKV tensor, in place.  One wave
// per (token, which); a lane holds eight neighbouring columns per round (H <= 1024: two rounds).  Two-pass statistics as
// the row kernels of encoder.hip (mean, then the variance of the deviations).  ln = gamma_q | beta_q | gamma_k | beta_k.
template <bool SPLIT>
__global__ void __launch_bounds__(256)
qk_layernorm_kernel(void* __restrict__ qkv, const float* __restrict__ ln, float eps, uint32_t T, uint32_t H, uint32_t* __restrict__ flag) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t job = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= (uint64_t)T * 2) return;
    const uint64_t t = job >> 1;
    const uint32_t which = (uint32_t)(job & 1);
    const float* gamma = ln + (size_t)which * 2 * H;
    const float* beta = gamma + H;
    sh_f32x4 va[2], vb[2];
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t c = (lane + 64 * r) * 8;  // first of eight columns inside the Q (K) part
        va[r] = sh_f32x4{0.f, 0.f, 0.f, 0.f}; vb[r] = va[r];
        if (c < H) {
            const uint32_t col = which * H + c;
            if (SPLIT) {
                const _Float16* p = static_cast<const _Float16*>(qkv) + t * (uint64_t)(3 * H) * 2 + (size_t)(col >> 5) * 64 + (col & 31);
                unsplit8(*reinterpret_cast<const f16x8*>(p), *reinterpret_cast<const f16x8*>(p + 32), va[r], vb[r]);
            } else {
                const float* p = static_cast<const float*>(qkv) + t * (uint64_t)(3 * H) + col;
                va[r] = *reinterpret_cast<const sh_f32x4*>(p);
                vb[r] = *reinterpret_cast<const sh_f32x4*>(p + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += va[r][e] + vb[r][e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)H;
    float var = 0.0f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if ((lane + 64 * r) * 8 < H) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float da = va[r][e] - mean, db = vb[r][e] - mean;
                var += da * da + db * db;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float inv = 1.0f / sqrtf(var / (float)H + eps);
    uint32_t mx = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t c = (lane + 64 * r) * 8;
        if (c < H) {
            const sh_f32x4 g0 = *reinterpret_cast<const sh_f32x4*>(gamma + c), g1 = *reinterpret_cast<const sh_f32x4*>(gamma + c + 4);
            const sh_f32x4 b0 = *reinterpret_cast<const sh_f32x4*>(beta + c), b1 = *reinterpret_cast<const sh_f32x4*>(beta + c + 4);
            sh_f32x4 oa, ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                oa[e] = (va[r][e] - mean) * inv * g0[e] + b0[e];
                ob[e] = (vb[r][e] - mean) * inv * g1[e] + b1[e];
            }
            const uint32_t col = which * H + c;
            if (SPLIT) {
                _Float16* p = static_cast<_Float16*>(qkv) + t * (uint64_t)(3 * H) * 2 + (size_t)(col >> 5) * 64 + (col & 31);
                f16x8 oh, ol;
                sh_split8(oa, ob, oh, ol, mx);
                *reinterpret_cast<f16x8*>(p) = oh;
                *reinterpret_cast<f16x8*>(p + 32) = ol;
            } else {
                float* p = static_cast<float*>(qkv) + t * (uint64_t)(3 * H) + col;
                *reinterpret_cast<sh_f32x4*>(p) = oa;
                *reinterpret_cast<sh_f32x4*>(p + 4) = ob;
            }
        }
    }
    if (SPLIT && sh_split_overflowed(mx) && flag) atomicOr(flag, 1u);
}

}  // namespace

int32_t launch_rope_split(_Float16* qkvs, const float2* rope, uint32_t T, uint32_t L, uint32_t H, uint32_t heads,
                          uint32_t* flag, hipStream_t s) {
    if (!T) return CS_OK;
    const uint32_t dh = heads ? H / heads : 0;
    if (dh != 32 && dh != 64) return fail(CS_ERR_UNSUPPORTED, "rotary map: head_dim %u not supported (32 or 64)", dh);
    const uint64_t threads = (uint64_t)T * 2 * heads * (dh / 16);
    const uint32_t blocks = (uint32_t)((threads + 255) / 256);
    if (dh == 64) hipLaunchKernelGGL(rope_split_kernel<64>, dim3(blocks), dim3(256), 0, s, qkvs, rope, T, L, H, flag);
    else hipLaunchKernelGGL(rope_split_kernel<32>, dim3(blocks), dim3(256), 0, s, qkvs, rope, T, L, H, flag);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_rope_f32(float* qkv, const float2* rope, uint32_t T, uint32_t L, uint32_t H, uint32_t heads, hipStream_t s) {
    if (!T) return CS_OK;
    const uint32_t dh = heads ? H / heads : 0;
    if (dh == 0 || dh % 8) return fail(CS_ERR_UNSUPPORTED, "rotary map: head_dim %u not supported", dh);
    const uint64_t threads = (uint64_t)T * 2 * heads * (dh / 8);
    hipLaunchKernelGGL(rope_f32_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, qkv, rope, T, L, H, dh);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_swiglu_split(const _Float16* up2, _Float16* out, uint32_t T, uint32_t I, uint32_t* flag, hipStream_t s, bool gelu_gate) {
    if (!T) return CS_OK;
    if (I % 32) return fail(CS_ERR_UNSUPPORTED, "gated feed-forward: intermediate size %u is not a multiple of 32", I);
    const uint64_t threads = (uint64_t)T * (I / 8);
    const dim3 grid((uint32_t)((threads + 255) / 256));
    if (gelu_gate) hipLaunchKernelGGL(swiglu_split_kernel<true>, grid, dim3(256), 0, s, up2, out, T, I, flag);
    else hipLaunchKernelGGL(swiglu_split_kernel<false>, grid, dim3(256), 0, s, up2, out, T, I, flag);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_swiglu_f32(float* value, const float* gate, uint32_t T, uint32_t I, hipStream_t s, bool gelu_gate) {
    if (!T) return CS_OK;
    const uint64_t n4 = (uint64_t)T * I / 4;
    const dim3 grid((uint32_t)((n4 + 255) / 256));
    if (gelu_gate) hipLaunchKernelGGL(swiglu_f32_kernel<true>, grid, dim3(256), 0, s, value, gate, n4);
    else hipLaunchKernelGGL(swiglu_f32_kernel<false>, grid, dim3(256), 0, s, value, gate, n4);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_qk_layernorm_split(_Float16* qkvs, const float* ln, float eps, uint32_t T, uint32_t H, uint32_t* flag, hipStream_t s) {
    if (!T) return CS_OK;
    if (H % 32 || H > 1024) return fail(CS_ERR_UNSUPPORTED, "query / key LayerNorm: hidden size %u not supported (multiples of 32 up to 1024)", H);
    hipLaunchKernelGGL(qk_layernorm_kernel<true>, dim3((uint32_t)(((uint64_t)T * 2 + 3) / 4)), dim3(256), 0, s, (void*)qkvs, ln, eps, T, H, flag);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_qk_layernorm_f32(float* qkv, const float* ln, float eps, uint32_t T, uint32_t H, hipStream_t s) {
    if (!T) return CS_OK;
    if (H % 8 || H > 1024) return fail(CS_ERR_UNSUPPORTED, "query / key LayerNorm: hidden size %u not supported (multiples of 8 up to 1024)", H);
    hipLaunchKernelGGL(qk_layernorm_kernel<false>, dim3((uint32_t)(((uint64_t)T * 2 + 3) / 4)), dim3(256), 0, s, (void*)qkv, ln, eps, T, H, (uint32_t*)nullptr);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
