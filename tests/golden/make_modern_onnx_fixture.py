#!/usr/bin/env python3
"""Writes tests/golden/modern_tiny_export.onnx + modern_tiny_export_state.npz: transformers' OWN ModernBertModel (hidden 64, 2
heads of 32, 3 layers — global, local, local — feed-forward 80, vocabulary 48, local window 8 either side, no bias anywhere:
the published configuration's attention_bias / mlp_bias / norm_bias = false) exported by torch.onnx's TorchScript exporter,
the state dict it was exported from, and the model's own output on a small padded batch.

This is the shape of file fastembed caches for the registry's modernbert-embed-large entry
(/root/reference/src/embed/embedder.rs:47, :72 -> onnx/model.onnx): bias-free Linear weights arrive as anonymous transposed
`onnx::MatMul_N` initialisers in module order (Wqkv, attn.Wo, mlp.Wi, mlp.Wo per layer), LayerNorm weights keep their
state-dict names, the GELU gate is a Split whose first half reaches an Erf.  cs_bert_params_from_onnx's ModernBERT branch
(codesearch_amd/csrc/onnx_reader.cpp) has to find its way through it; the GPU test then runs the loaded model against the
stored output of the exporting model itself.  transformers is a third-party library, not the reference (parity
against the reference itself stays unpinned).  Every parameter is seeded noise.
Run: python tests/golden/make_modern_onnx_fixture.py"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
H, NH, LAYERS, INNER, VOCAB = 64, 2, 3, 80, 48


def write(out_dir, stem, dims=None, local_attention=16, opset=14):
    """dims = (hidden, heads, layers, intermediate, vocabulary): the GPU tests export at a width the kernels run (384)"""
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    from transformers import ModernBertConfig, ModernBertModel

    H, NH, LAYERS, INNER, VOCAB = dims or (globals()["H"], globals()["NH"], globals()["LAYERS"], globals()["INNER"], globals()["VOCAB"])
    onnx_proto_utils._add_onnxscript_fn = lambda proto, *a, **k: proto
    hc = ModernBertConfig(vocab_size=VOCAB, hidden_size=H, intermediate_size=INNER, num_hidden_layers=LAYERS, num_attention_heads=NH,
                          max_position_embeddings=64, norm_eps=1e-5, norm_bias=False, attention_bias=False, mlp_bias=False,
                          local_attention=local_attention, global_attn_every_n_layers=3, global_rope_theta=160000.0, local_rope_theta=10000.0,
                          pad_token_id=0, bos_token_id=1, eos_token_id=2, cls_token_id=1, sep_token_id=2)
    hc._attn_implementation = "eager"
    torch.manual_seed(20261005)
    model = ModernBertModel(hc).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.randn_like(p) * 0.08 + (1.0 if name.endswith("norm.weight") else 0.0))
    ids = torch.randint(3, VOCAB, (2, 12))
    mask = torch.ones(2, 12, dtype=torch.long)
    out = os.path.join(out_dir, stem + ".onnx")
    axes = {n: {0: "batch", 1: "seq"} for n in ("input_ids", "attention_mask")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model, (ids, mask), out, input_names=list(axes), output_names=["last_hidden_state"], dynamic_axes=axes,
                          opset_version=opset, dynamo=False)
    # the exporting model's own answer: 3 sequences of 40 tokens (longer than the local window), the last two padded
    g = torch.Generator().manual_seed(7)
    qids = torch.randint(3, VOCAB, (3, 40), generator=g)
    lens = [40, 29, 18]
    qmask = torch.zeros(3, 40, dtype=torch.long)
    for b, n in enumerate(lens):
        qmask[b, :n] = 1
        qids[b, n:] = 0
    with torch.no_grad():
        hidden = model(input_ids=qids, attention_mask=qmask).last_hidden_state
    w = qmask[:, :, None].to(hidden.dtype)
    pooled = (hidden * w).sum(1) / w.sum(1)
    pooled = pooled / pooled.norm(dim=1, keepdim=True)
    state = {k: v.numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(out_dir, stem + "_state.npz"), query_ids=qids.numpy().astype(np.int32),
                        query_mask=qmask.numpy().astype(np.int32), query_pooled=pooled.numpy().astype(np.float32), **state)
    print("wrote", out, os.path.getsize(out), "bytes")
    return out


def main():
    write(HERE, "modern_tiny_export")


if __name__ == "__main__":
    sys.exit(main())
