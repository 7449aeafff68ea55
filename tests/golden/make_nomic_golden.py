"""Generates tests/golden/nomic_golden.npz — known answers for the CS_ARCH_NOMIC encoder (NomicBert: the registry's
nomic-embed-text-v1 / v1.5 / v1.5-Q entries, /root/reference/src/embed/embedder.rs:30-35).

Run in the build container:  python tests/golden/make_nomic_golden.py

Source of truth: a float64 torch statement of the published NomicBert forward (post-norm, rotary on Q / K, swiglu
feed-forward, mean pooling) assembled from library pieces this repo did not write: transformers' own `rotate_half` /
`apply_rotary_pos_emb` (the non-interleaved map NomicBert uses with rotary_emb_interleaved = false), torch's
scaled_dot_product_attention, layer_norm, silu and linear.  `transformers` has no NomicBert class (the model ships its code
in its repository, which is not reachable from here), so this is NOT the reference and not the model's own code: encoder
parity stays "unpinned against the reference" (DESIGN.md §4); these vectors pin the oracle and the HIP path to an
independent implementation of the same operators.  Angles are formed in f32 as the module builds its cos / sin cache
(inv_freq and pos * inv_freq in f32), everything behind them in f64.  Weights: the integer generator of
include/cs_bert_params.h, so only seeds and outputs are stored.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from codesearch_amd.bert_params import (ARCH_NOMIC, POOL_MEAN, BertConfig, synth_params, synth_token_batch,  # noqa: E402
                                        to_state_dict)


def nomic_forward(cfg: BertConfig, flat, ids, mask):
    from transformers.models.llama.modeling_llama import apply_rotary_pos_emb

    sd = {k: torch.from_numpy(v.astype(np.float64)) for k, v in to_state_dict(cfg, flat).items()}
    B, L = ids.shape
    H, NH = cfg.hidden, cfg.heads
    DH = H // NH
    x = sd["embeddings.word_embeddings.weight"][torch.from_numpy(ids.astype(np.int64))] \
        + sd["embeddings.token_type_embeddings.weight"][0]
    x = F.layer_norm(x, (H,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], cfg.layer_norm_eps)
    hs = [x.numpy()]
    inv_freq = 1.0 / (torch.tensor(cfg.rotary_base, dtype=torch.float32)
                      ** (torch.arange(0, DH, 2, dtype=torch.float32) / DH))
    ang = torch.outer(torch.arange(L, dtype=torch.float32), inv_freq)  # f32, as the module's cache
    ang = torch.cat([ang, ang], dim=-1).double()[None]                 # [1, L, DH]
    cos, sin = torch.cos(ang), torch.sin(ang)
    keep = torch.from_numpy(mask.astype(bool))[:, None, None, :]         # [B, 1, 1, L]
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."

        def lin(t, name):
            return F.linear(t, sd[p + name + ".weight"], sd[p + name + ".bias"])

        def heads(t):
            return t.view(B, L, NH, DH).transpose(1, 2)

        q, k, v = heads(lin(x, "attention.self.query")), heads(lin(x, "attention.self.key")), heads(lin(x, "attention.self.value"))
        q, k = apply_rotary_pos_emb(q, k, cos, sin)
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=keep)
        ctx = ctx.transpose(1, 2).reshape(B, L, H)
        x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (H,), sd[p + "attention.output.LayerNorm.weight"],
                         sd[p + "attention.output.LayerNorm.bias"], cfg.layer_norm_eps)
        y = lin(x, "intermediate.dense") * F.silu(lin(x, "intermediate.gate"))
        x = F.layer_norm(lin(y, "output.dense") + x, (H,), sd[p + "output.LayerNorm.weight"],
                         sd[p + "output.LayerNorm.bias"], cfg.layer_norm_eps)
        hs.append(x.numpy())
    last = hs[-1]
    m = mask.astype(np.float64)[:, :, None]
    mean = (last * m).sum(1) / np.maximum(m.sum(1), 1e-9)
    return hs, mean / (np.linalg.norm(mean, axis=1, keepdims=True) + 1e-12)


CASES = [
    # name, config, weight seed, B, L, ragged
    ("dh32_L7", dict(vocab_size=512, hidden=384, layers=2, heads=12, intermediate=1536, max_position=512), 301, 4, 7, True),
    ("dh32_L64", dict(vocab_size=512, hidden=384, layers=2, heads=12, intermediate=1536, max_position=512), 301, 4, 64, True),
    ("dh64_L48", dict(vocab_size=512, hidden=768, layers=2, heads=12, intermediate=3072, max_position=512), 302, 3, 48, True),
    ("dh64_full_mask", dict(vocab_size=512, hidden=768, layers=2, heads=12, intermediate=3072, max_position=512), 302, 2, 32, False),
    # nomic-embed-text-v1.5's own shape (12 x 768, 12 heads of 64, n_inner 3072, vocab 30528, rotary base 1000)
    ("nomic_shape", dict(vocab_size=30528, hidden=768, layers=12, heads=12, intermediate=3072, max_position=512), 303, 4, 128, True),
]


def case_config(kw) -> BertConfig:
    return BertConfig(pooling=POOL_MEAN, arch=ARCH_NOMIC, rotary_base=1000.0, **kw)


def main():
    torch.set_num_threads(8)
    out, names = {}, []
    for name, kw, wseed, B, L, ragged in CASES:
        cfg = case_config(kw)
        flat = synth_params(cfg, wseed)
        ids, mask = synth_token_batch(cfg, wseed + 50, B, L, ragged)
        with torch.no_grad():
            hs, mean = nomic_forward(cfg, flat, ids, mask)
        names.append(name)
        out[name + "/meta"] = np.array([kw["vocab_size"], kw["hidden"], kw["layers"], kw["heads"], kw["intermediate"],
                                        kw["max_position"], wseed, wseed + 50, B, L, int(ragged)], np.int64)
        out[name + "/mean"] = mean
        valid = mask.astype(bool)
        out[name + "/layer_absmean"] = np.array([np.abs(h[valid]).mean() for h in hs])
        H = kw["hidden"]
        out[name + "/layer_probe"] = np.array([[h[0, 0, 0], h[B - 1, 1, 7], h[0, mask[0].sum() - 1, H - 1]] for h in hs])
        out[name + "/last_row0"] = hs[-1][0, 0, :]
        off = (mean @ mean.T)[~np.eye(B, dtype=bool)]
        print(name, "mean[0,:3]", mean[0, :3], "max off-diagonal cosine", float(off.max()))
    out["names"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nomic_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
