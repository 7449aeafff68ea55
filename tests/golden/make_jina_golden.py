"""Generates tests/golden/jina_golden.npz — known answers for the CS_ARCH_JINA / CS_ARCH_JINA_QKNORM encoder (JinaBert:
the registry's jina-embeddings-v2-base-code entry, /root/reference/src/embed/embedder.rs:40-41, :69, :92, :112).

Run in the build container:  python tests/golden/make_jina_golden.py

Source of truth: a float64 torch statement of the published JinaBert forward (post-norm BERT without a position table,
the symmetric ALiBi bias -slope_h |i - j| on the scores, a GELU-gated feed-forward over the halves of one bias-free
[2I, H] projection, optionally LayerNorm on the whole query / key rows, mean pooling) assembled from library pieces this
repo did not write: the head slopes come out of transformers' own `build_mpt_alibi_tensor` (MPT's ALiBi: the same
geometric sequence and the same "closest power of two, then every second slope of the doubled set" rule), the rest is
torch's scaled_dot_product_attention with an additive float mask, layer_norm, gelu and linear.  `transformers` has no
JinaBert class (the model ships its code in its repository, not reachable from here), so this is NOT the reference and not
the model's own code: encoder parity stays "unpinned against the reference" (DESIGN.md); these vectors pin the oracle
and the HIP path to an independent implementation of the same operators.  Weights: the integer generator of
include/cs_bert_params.h, so only seeds and outputs are stored.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from codesearch_amd.bert_params import (ARCH_JINA, ARCH_JINA_QKNORM, POOL_MEAN, BertConfig, synth_params,  # noqa: E402
                                        synth_token_batch, to_state_dict)


def library_slopes(heads: int) -> torch.Tensor:
    from transformers.models.mpt.modeling_mpt import build_mpt_alibi_tensor

    a = build_mpt_alibi_tensor(heads, 4, alibi_bias_max=8)  # [heads, 1, 4]: slope_h * (j - 3)
    return (-a[:, 0, 2]).to(torch.float32)                  # position -1 -> the slope itself


def jina_forward(cfg: BertConfig, flat, ids, mask):
    sd = {k: torch.from_numpy(v.astype(np.float64)) for k, v in to_state_dict(cfg, flat).items()}
    B, L = ids.shape
    H, NH = cfg.hidden, cfg.heads
    DH = H // NH
    x = sd["embeddings.word_embeddings.weight"][torch.from_numpy(ids.astype(np.int64))] \
        + sd["embeddings.token_type_embeddings.weight"][0]
    x = F.layer_norm(x, (H,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], cfg.layer_norm_eps)
    hs = [x.numpy()]
    pos = torch.arange(L)
    dist = (pos[None, :] - pos[:, None]).abs().double()                   # |i - j|
    bias = -library_slopes(NH).double()[:, None, None] * dist[None]       # [NH, L, L]
    pad = torch.where(torch.from_numpy(mask.astype(bool)), 0.0, torch.finfo(torch.float32).min).double()  # HF's additive mask
    add = bias[None] + pad[:, None, None, :]                              # [B, NH, L, L]
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."

        def lin(t, name):
            return F.linear(t, sd[p + name + ".weight"], sd[p + name + ".bias"])

        def heads(t):
            return t.view(B, L, NH, DH).transpose(1, 2)

        q, k, v = lin(x, "attention.self.query"), lin(x, "attention.self.key"), lin(x, "attention.self.value")
        if cfg.arch == ARCH_JINA_QKNORM:
            q = F.layer_norm(q, (H,), sd[p + "attention.self.layer_norm_q.weight"], sd[p + "attention.self.layer_norm_q.bias"], cfg.layer_norm_eps)
            k = F.layer_norm(k, (H,), sd[p + "attention.self.layer_norm_k.weight"], sd[p + "attention.self.layer_norm_k.bias"], cfg.layer_norm_eps)
        ctx = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), attn_mask=add)
        ctx = ctx.transpose(1, 2).reshape(B, L, H)
        x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (H,), sd[p + "attention.output.LayerNorm.weight"],
                         sd[p + "attention.output.LayerNorm.bias"], cfg.layer_norm_eps)
        y = lin(x, "intermediate.dense") * F.gelu(lin(x, "intermediate.gate"))
        x = F.layer_norm(lin(y, "output.dense") + x, (H,), sd[p + "output.LayerNorm.weight"],
                         sd[p + "output.LayerNorm.bias"], cfg.layer_norm_eps)
        hs.append(x.numpy())
    last = hs[-1]
    m = mask.astype(np.float64)[:, :, None]
    mean = (last * m).sum(1) / np.maximum(m.sum(1), 1e-9)
    return hs, mean / (np.linalg.norm(mean, axis=1, keepdims=True) + 1e-12)


CASES = [
    # name, arch, config, weight seed, B, L, ragged
    ("dh32_L7", ARCH_JINA, dict(vocab_size=512, hidden=384, layers=2, heads=12, intermediate=1536, max_position=512), 401, 4, 7, True),
    ("dh32_qkn_L64", ARCH_JINA_QKNORM, dict(vocab_size=512, hidden=384, layers=2, heads=12, intermediate=1536, max_position=512), 401, 4, 64, True),
    ("dh64_L48", ARCH_JINA, dict(vocab_size=512, hidden=768, layers=2, heads=12, intermediate=3072, max_position=512), 402, 3, 48, True),
    ("dh64_qkn_L200", ARCH_JINA_QKNORM, dict(vocab_size=512, hidden=768, layers=2, heads=12, intermediate=3072, max_position=512), 402, 2, 200, True),
    ("heads16_qkn_full_mask", ARCH_JINA_QKNORM, dict(vocab_size=512, hidden=1024, layers=1, heads=16, intermediate=4096, max_position=512), 404, 2, 32, False),
    # jina-embeddings-v2-base-code's own shape (12 x 768, 12 heads of 64, intermediate 3072, vocab 61056)
    ("jina_code_shape", ARCH_JINA_QKNORM, dict(vocab_size=61056, hidden=768, layers=12, heads=12, intermediate=3072, max_position=512), 403, 4, 128, True),
]


def case_config(arch, kw) -> BertConfig:
    return BertConfig(pooling=POOL_MEAN, arch=arch, **kw)


def main():
    torch.set_num_threads(8)
    out, names = {}, []
    for heads in (8, 12, 16):
        out[f"slopes/{heads}"] = library_slopes(heads).numpy()
    for name, arch, kw, wseed, B, L, ragged in CASES:
        cfg = case_config(arch, kw)
        flat = synth_params(cfg, wseed)
        ids, mask = synth_token_batch(cfg, wseed + 50, B, L, ragged)
        with torch.no_grad():
            hs, mean = jina_forward(cfg, flat, ids, mask)
        names.append(name)
        out[name + "/meta"] = np.array([kw["vocab_size"], kw["hidden"], kw["layers"], kw["heads"], kw["intermediate"],
                                        kw["max_position"], wseed, wseed + 50, B, L, int(ragged), arch], np.int64)
        out[name + "/mean"] = mean
        valid = mask.astype(bool)
        out[name + "/layer_absmean"] = np.array([np.abs(h[valid]).mean() for h in hs])
        H = kw["hidden"]
        out[name + "/layer_probe"] = np.array([[h[0, 0, 0], h[B - 1, 1, 7], h[0, mask[0].sum() - 1, H - 1]] for h in hs])
        out[name + "/last_row0"] = hs[-1][0, 0, :]
        off = (mean @ mean.T)[~np.eye(B, dtype=bool)]
        print(name, "mean[0,:3]", mean[0, :3], "max off-diagonal cosine", float(off.max()))
    out["names"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jina_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
