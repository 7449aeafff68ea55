// small_forward.hpp — launch interface of the one-launch forward of short queries (small_forward.hip).
#pragma once

#include "encoder.hpp"

namespace cs {

// device pointers of one encoder layer (weights in split-f16 form, [N][K/32][64])
struct SfLayer {
    const _Float16 *wqkv, *wo, *wup, *wdown;
    const float *bqkv, *bo, *bup, *bdown, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};

struct SfArgs {
    const int32_t* ids;
    const int32_t* mask;
    const float *word, *pos, *type0, *emb_g, *emb_b;
    const SfLayer* layers;  // device array [n_layers]
    uint32_t n_layers;
    float eps;
    uint32_t T, L, B, vocab, heads, hb;
    float* X;          // [T, H] f32: the residual stream behind the attention block's LayerNorm (and, at the end, the last hidden state)
    float* XA;         // [T, H] f32: the residual stream behind a layer's last LayerNorm (the next layer's input)
    float* Y;          // [T, H] f32: out-proj + bias + residual, in front of the attention block's LayerNorm
    float* PARTS;      // [4][T][H] f32: FFN-down's four K slices (summed, with bias and residual, by the LayerNorm that follows)
    _Float16* QKVS;    // [T][3H/32][64]
    _Float16* CTXS;    // [T][H/32][64]
    _Float16* MIDS;    // [T][I/32][64]
    uint32_t* flag;    // split-f16 range flag (as every split kernel)
    uint32_t* sync;    // [4] zeroed before every launch: [0] arrivals, [1] give-up code (0 = none)
    uint64_t* dbg;     // diagnostics or null: [96][3] ticks of 10 ns each block spent computing / draining stores / at barriers
};

constexpr uint32_t SF_MAX_ROWS = 512;  // token rows one launch takes
// H = 384, I = 1536, head_dim 32 (the 384-d BERT family: BGE-small, MiniLM-L6 / L12, multilingual-e5-small), L <= 512
bool small_forward_supported(uint32_t H, uint32_t I, uint32_t heads, uint32_t T, uint32_t L);
// Everything of the forward up to the last LayerNorm (X = the last hidden state, f32), as ONE kernel on `grid` resident blocks.
// sync must be zeroed on the stream first (hipMemsetAsync of 16 bytes); after the stream has drained, sync[1] != 0 says the
// launch gave up at a grid barrier (the caller reruns the mini-batch on the multi-launch path).
int32_t launch_small_forward(const SfArgs& a, hipStream_t s);

}  // namespace cs
