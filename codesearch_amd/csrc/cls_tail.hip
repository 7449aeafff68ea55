// cls_tail.hip — the LAST encoder layer of a CLS-pooled model (BGE family) restricted to what the embedding reads.
//
// fastembed pools a BGE model by taking the hidden state of token 0 (the reference calls it at
// /root/reference/src/embed/embedder.rs:286-289).  In the last layer nothing but that row is ever looked at again: its
// attention needs K and V of every token (so the K/V projection stays whole) but only ONE query per sequence, and the
// output projection, both LayerNorms and the feed-forward block are needed for B rows instead of B x L.  The result is
// the same embedding — not an approximation — for 1/12 less work at 12 layers (measured: DESIGN.md §3.6b).
// Mean-pooled models (MiniLM family) read every row and keep the full layer.
//
// Even the query projection shrinks: the last layer's big GEMM computes K and V only (N = 2H); the B CLS queries come from
// a small GEMM over the gathered rows.
//   attention_cls_kernel : one wave per (sequence, head): scores of the CLS query against all keys, softmax, P V —
//                          plain f32 arithmetic on the split operands (hi + lo / 2048 is exact in f32), 16-byte loads:
//                          a pass covers 8 keys (head_dim 32) or 4 (head_dim 64), lane = (key slot, 16-byte chunk).
//   gather_cls_kernel    : row b*L of the split residual stream -> compact f32 + split rows [B, H].
// The dense layers and LayerNorms of the B compact rows run on the small-batch kernels the query path already uses.
#include "encoder.hpp"
#include "split_f16.hpp"

namespace cs {

namespace {

constexpr float kClsMasked = -3.0e38f;

template <int NC>  // head_dim = 32 * NC
__global__ void __launch_bounds__(64)
attention_cls_kernel(const _Float16* __restrict__ q_cls, const _Float16* __restrict__ kvs, const int32_t* __restrict__ mask,
                     _Float16* __restrict__ ctxs_cls, uint32_t* __restrict__ flag, uint32_t L, uint32_t H, float scale_log2e) {
    __shared__ float q_s[64];
    __shared__ float p_s[512];
    constexpr int CH = 8 * NC;        // 16-byte chunks of one key (hi and lo of every 8 dims)
    constexpr int KP = 64 / CH;       // keys per pass
    const int lane = threadIdx.x;
    const uint32_t head = blockIdx.x, b = blockIdx.y;
    // kvs: [T][2H/32][64] — per token the K lines of every head, then the V lines (the last layer projects K and V only;
    // the CLS queries come from a GEMM over the B compact rows: q_cls [B][H/32][64])
    const uint32_t nh = H / (32 * NC), nch = 2 * nh * NC;
    const _Float16* base = kvs + (size_t)b * L * nch * 64;
    // the CLS query (token 0 of the sequence) as f32, pre-multiplied by log2(e) / sqrt(d)
    if (lane < 32 * NC) {
        const _Float16* qp = q_cls + ((size_t)b * (H / 32) + head * NC + lane / 32) * 64;
        q_s[lane] = fmaf((float)qp[32 + lane % 32], kShLoInv, (float)qp[lane % 32]) * scale_log2e;
    }
    __syncthreads();
    const int c8 = lane % CH, ks = lane / CH;
    const int line = c8 / 8, slot = c8 % 8;
    const bool lo_slot = slot >= 4;
    const int d0 = line * 32 + (slot & 3) * 8;  // first of this lane's 8 dims
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = q_s[d0 + e] * (lo_slot ? kShLoInv : 1.0f);
    // ---- scores ---- (four passes per iteration: four independent 16-byte loads in flight per lane)
    float mx = kClsMasked;
    constexpr int U = 4;
    for (uint32_t j0 = 0; j0 < L; j0 += KP * U) {
        f16x8 kv[U];
        uint32_t jx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            jx[u] = j0 + u * KP + ks;
            const uint32_t jj = jx[u] < L ? jx[u] : L - 1;
            kv[u] = *reinterpret_cast<const f16x8*>(base + ((size_t)jj * nch + head * NC + line) * 64 + slot * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float part = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) part = fmaf(qv[e], (float)kv[u][e], part);
#pragma unroll
            for (int m = 1; m < CH; m <<= 1) part += __shfl_xor(part, m, 64);
            const uint32_t j = jx[u];
            const bool ok = j < L && mask[(size_t)b * L + (j < L ? j : L - 1)] != 0;
            const float sc = ok ? part : kClsMasked;
            if (c8 == 0 && j < L) p_s[j] = sc;
            mx = fmaxf(mx, sc);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    __syncthreads();
    float lsum = 0.0f;
    for (uint32_t j = lane; j < L; j += 64) {
        const float p = __builtin_amdgcn_exp2f(p_s[j] - mx);  // a masked key: exp2(-3e38 - mx) = 0 (mx is a live key's score)
        p_s[j] = p;
        lsum += p;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lsum += __shfl_xor(lsum, m, 64);
    __syncthreads();
    // ---- P V ----
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.0f;
    for (uint32_t j0 = 0; j0 < L; j0 += KP * U) {
        f16x8 vv[U];
        float pu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = j0 + u * KP + ks;
            const uint32_t jj = j < L ? j : L - 1;
            vv[u] = *reinterpret_cast<const f16x8*>(base + ((size_t)jj * nch + (nh + head) * NC + line) * 64 + slot * 8);
            pu[u] = j < L ? p_s[jj] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pu[u], (float)vv[u][e], o[e]);
    }
    // sum over the key slots (lanes with the same chunk), then hi + lo / 2048 (lanes slot and slot + 4)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int m = CH; m < 64; m <<= 1) o[e] += __shfl_xor(o[e], m, 64);
        const float other = __shfl_xor(o[e], 4, 64);  // the lo-slot partner (or, seen from it, the hi slot)
        o[e] = lo_slot ? 0.0f : fmaf(other, kShLoInv, o[e]);
    }
    if (ks == 0 && !lo_slot) {
        const float inv = 1.0f / lsum;
        sh_f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = o[e] * inv; v1[e] = o[4 + e] * inv; }
        f16x8 hi, lo;
        uint32_t mxh = 0;
        sh_split8(v0, v1, hi, lo, mxh);
        _Float16* op = ctxs_cls + ((size_t)b * (H / 32) + head * NC + line) * 64 + slot * 8;
        *reinterpret_cast<f16x8*>(op) = hi;
        *reinterpret_cast<f16x8*>(op + 32) = lo;
        if (flag && sh_split_overflowed(mxh)) atomicOr(flag, 1u);
    }
}

// row b * L of the split stream -> x_cls[b] (f32: hi + lo / 2048, exact) and xs_cls[b] (the same line bytes)
__global__ void __launch_bounds__(256)
gather_cls_kernel(const _Float16* __restrict__ xs, float* __restrict__ x_cls, _Float16* __restrict__ xs_cls, uint32_t B,
                  uint32_t L, uint32_t H) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 8 columns
    const uint32_t per_row = H / 8;
    if (i >= B * per_row) return;
    const uint32_t b = i / per_row, c = (i % per_row) * 8;
    const _Float16* src = xs + ((size_t)b * L * (H / 32) + (c >> 5)) * 64 + (c & 31);
    const f16x8 hi = *reinterpret_cast<const f16x8*>(src), lo = *reinterpret_cast<const f16x8*>(src + 32);
    _Float16* dst = xs_cls + ((size_t)b * (H / 32) + (c >> 5)) * 64 + (c & 31);
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + 32) = lo;
    sh_f32x4 a, d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = fmaf((float)lo[e], kShLoInv, (float)hi[e]);
        d[e] = fmaf((float)lo[4 + e], kShLoInv, (float)hi[4 + e]);
    }
    *reinterpret_cast<sh_f32x4*>(x_cls + (size_t)b * H + c) = a;
    *reinterpret_cast<sh_f32x4*>(x_cls + (size_t)b * H + c + 4) = d;
}

}  // namespace

int32_t launch_attention_cls(const _Float16* q_cls, const _Float16* kv_split, const int32_t* mask, _Float16* ctxs_cls,
                             uint32_t* flag, uint32_t B, uint32_t L, uint32_t H, uint32_t heads, hipStream_t s) {
    const uint32_t dh = heads ? H / heads : 0;
    if ((dh != 32 && dh != 64) || H % heads || L == 0 || L > 512)
        return fail(CS_ERR_UNSUPPORTED, "CLS attention: head_dim %u / length %u not supported", dh, L);
    constexpr float kLog2e = 1.4426950408889634f;
    if (dh == 32)
        hipLaunchKernelGGL(attention_cls_kernel<1>, dim3(heads, B), dim3(64), 0, s, q_cls, kv_split, mask, ctxs_cls, flag, L, H,
                           (1.0f / sqrtf(32.0f)) * kLog2e);
    else
        hipLaunchKernelGGL(attention_cls_kernel<2>, dim3(heads, B), dim3(64), 0, s, q_cls, kv_split, mask, ctxs_cls, flag, L, H,
                           (1.0f / sqrtf(64.0f)) * kLog2e);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int32_t launch_gather_cls(const _Float16* xs, float* x_cls, _Float16* xs_cls, uint32_t B, uint32_t L, uint32_t H, hipStream_t s) {
    const uint32_t n = B * (H / 8);
    hipLaunchKernelGGL(gather_cls_kernel, dim3((n + 255) / 256), dim3(256), 0, s, xs, x_cls, xs_cls, B, L, H);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
