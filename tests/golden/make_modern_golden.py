"""Generates tests/golden/modern_golden.npz — known answers for the CS_ARCH_MODERN encoder (ModernBERT: the registry's
modernbert-embed-large entry, /root/reference/src/embed/embedder.rs:47, :72).

Run in the build container:  python tests/golden/make_modern_golden.py

Source of truth: HF transformers' OWN `ModernBertModel` (modeling_modernbert.py of the installed library), in float64, eager
attention, fed the synthetic weights of include/cs_bert_params.h through its state dict — pre-norm layers, rotary positions
with a global and a local base, the sliding-window mask of the local layers, the GELU-gated feed-forward, the final
LayerNorm — followed by mean pooling and L2 normalisation (fastembed's pooling for the model).  The synthetic block carries
biases on every Linear and LayerNorm, so the HF config switches attention_bias / mlp_bias / norm_bias ON (the published
checkpoints have none: zero slots).  transformers is a third-party library, not the reference: these vectors pin the oracle
and the HIP path to the model family's published modelling code; parity against the reference itself stays unpinned
(DESIGN.md).  Only seeds and outputs are stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from codesearch_amd.bert_params import (ARCH_MODERN, POOL_MEAN, BertConfig, synth_params, to_state_dict,  # noqa: E402
                                        token_batch_with_lens)


def hf_state_dict(cfg: BertConfig, flat):
    ours = to_state_dict(cfg, flat)
    sd = {"embeddings.tok_embeddings.weight": ours["embeddings.word_embeddings.weight"],
          "embeddings.norm.weight": ours["embeddings.LayerNorm.weight"], "embeddings.norm.bias": ours["embeddings.LayerNorm.bias"],
          "final_norm.weight": ours["final_norm.weight"], "final_norm.bias": ours["final_norm.bias"]}
    for l in range(cfg.layers):
        a, b = f"encoder.layer.{l}.", f"layers.{l}."
        if l:
            sd[b + "attn_norm.weight"], sd[b + "attn_norm.bias"] = ours[a + "attention.output.LayerNorm.weight"], ours[a + "attention.output.LayerNorm.bias"]
        sd[b + "attn.Wqkv.weight"] = np.concatenate([ours[a + f"attention.self.{r}.weight"] for r in ("query", "key", "value")])
        sd[b + "attn.Wqkv.bias"] = np.concatenate([ours[a + f"attention.self.{r}.bias"] for r in ("query", "key", "value")])
        sd[b + "attn.Wo.weight"], sd[b + "attn.Wo.bias"] = ours[a + "attention.output.dense.weight"], ours[a + "attention.output.dense.bias"]
        sd[b + "mlp_norm.weight"], sd[b + "mlp_norm.bias"] = ours[a + "output.LayerNorm.weight"], ours[a + "output.LayerNorm.bias"]
        # Wi rows [0, I) go through the activation (our `gate`), rows [I, 2I) multiply it (our `dense`)
        sd[b + "mlp.Wi.weight"] = np.concatenate([ours[a + "intermediate.gate.weight"], ours[a + "intermediate.dense.weight"]])
        sd[b + "mlp.Wi.bias"] = np.concatenate([ours[a + "intermediate.gate.bias"], ours[a + "intermediate.dense.bias"]])
        sd[b + "mlp.Wo.weight"], sd[b + "mlp.Wo.bias"] = ours[a + "output.dense.weight"], ours[a + "output.dense.bias"]
    return sd


def hf_forward(cfg: BertConfig, flat, ids, mask):
    from transformers import ModernBertConfig, ModernBertModel

    hc = ModernBertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                          num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_position,
                          norm_eps=cfg.layer_norm_eps, norm_bias=True, attention_bias=True, mlp_bias=True,
                          local_attention=2 * cfg.local_window, global_attn_every_n_layers=cfg.global_every,
                          global_rope_theta=cfg.rotary_base, local_rope_theta=cfg.rotary_base_local, pad_token_id=0,
                          bos_token_id=1, eos_token_id=2, cls_token_id=1, sep_token_id=2)
    hc._attn_implementation = "eager"
    model = ModernBertModel(hc).double().eval()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v).astype(np.float64)) for k, v in hf_state_dict(cfg, flat).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("inv_freq" in m for m in missing), (missing, unexpected)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(mask.astype(np.int64)),
                    output_hidden_states=True)
    last = out.last_hidden_state.numpy()
    hs = [h.numpy() for h in out.hidden_states] if out.hidden_states is not None else []
    m = mask.astype(np.float64)[:, :, None]
    mean = (last * m).sum(1) / np.maximum(m.sum(1), 1e-9)
    return hs, last, mean / (np.linalg.norm(mean, axis=1, keepdims=True) + 1e-12)


# Row lengths: a padded QUERY position whose whole window is padding has no key to attend to — HF's eager softmax returns NaN
# there (and 0 x NaN then reaches the valid rows one layer on); real batches are only ever as ragged as their texts, but the
# cases here keep every padding position within `local_window` of its row's last token so that the library's own answer is
# finite everywhere.  (The HIP path masks with a finite value: its padded rows are finite don't-cares, its valid rows never
# read them.)
CASES = [
    # name, config, weight seed, row lengths, L   (small windows at these lengths: local layers really are local)
    ("dh32_L7", dict(vocab_size=512, hidden=384, layers=4, heads=12, intermediate=1536, max_position=512, local_window=16), 501, [7, 3, 5, 2], 7),
    ("dh32_L100", dict(vocab_size=512, hidden=384, layers=4, heads=12, intermediate=1536, max_position=512, local_window=16), 501, [100, 84, 91], 100),
    ("dh64_L48", dict(vocab_size=512, hidden=768, layers=4, heads=12, intermediate=3072, max_position=512, local_window=8), 502, [48, 40, 45], 48),
    ("dh64_full_mask", dict(vocab_size=512, hidden=1024, layers=2, heads=16, intermediate=2688, max_position=512, local_window=64), 504, [160, 160], 160),
    # modernbert-embed-large's own shape: 28 x 1024, 16 heads of 64, intermediate 2624 (padded to 2688 = 21 x 128: zero rows /
    # columns, the same function), vocab 50368, window 64, every third layer global
    ("modern_large_shape", dict(vocab_size=50368, hidden=1024, layers=28, heads=16, intermediate=2688, max_position=512, local_window=64), 503, [200, 137], 200),
]


def case_config(kw) -> BertConfig:
    return BertConfig(pooling=POOL_MEAN, arch=ARCH_MODERN, layer_norm_eps=1e-5, rotary_base=160000.0, rotary_base_local=10000.0,
                      global_every=3, **kw)


def main():
    torch.set_num_threads(8)
    out, names = {}, []
    for name, kw, wseed, lens, L in CASES:
        cfg = case_config(kw)
        flat = synth_params(cfg, wseed)
        B = len(lens)
        ids, mask = token_batch_with_lens(cfg, wseed + 50, lens, L)
        hs, last, mean = hf_forward(cfg, flat, ids, mask)
        names.append(name)
        out[name + "/meta"] = np.array([kw["vocab_size"], kw["hidden"], kw["layers"], kw["heads"], kw["intermediate"],
                                        kw["max_position"], wseed, wseed + 50, B, L, 0, kw["local_window"]], np.int64)
        out[name + "/lens"] = np.array(lens, np.int64)
        assert np.isfinite(last).all(), name
        out[name + "/mean"] = mean
        valid = mask.astype(bool)
        out[name + "/last_absmean"] = np.array(np.abs(last[valid]).mean())
        out[name + "/last_row0"] = last[0, 0, :]
        out[name + "/last_probe"] = np.array([last[0, 0, 0], last[B - 1, 1, 7], last[0, mask[0].sum() - 1, kw["hidden"] - 1]])
        off = (mean @ mean.T)[~np.eye(B, dtype=bool)]
        print(name, "mean[0,:3]", mean[0, :3], "max off-diagonal cosine", float(off.max()), "hidden states", len(hs))
    out["names"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "modern_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
