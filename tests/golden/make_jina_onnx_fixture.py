#!/usr/bin/env python3
"""Writes tests/golden/jina_tiny_export.onnx + jina_tiny_export_state.npz: a JinaBert-shaped encoder (hidden 64, 2 heads of 32,
2 layers, intermediate 128, vocabulary 48; BERT's attention block without a position table, the symmetric ALiBi bias on the
scores, LayerNorm on the whole query / key rows, a GELU-gated feed-forward over the halves of one bias-free [2I, H] projection
— the module and parameter names of jinaai/jina-bert-v2-qk-post-norm's modelling file, the one jina-embeddings-v2-base-code's
config points at: attention.self.{query, key, value, layer_norm_q, layer_norm_k}, attention.output.{dense, LayerNorm},
mlp.{up_gated_layer, down_layer, layernorm}) exported by torch.onnx's TorchScript exporter, the state dict it was exported
from, and the module's own output on a padded batch.

The MODULE below is this repo's restatement of that structure (the modelling file is not reachable from here; the ALiBi
slopes come out of transformers' own build_mpt_alibi_tensor); the FILE is a real exporter's output: Linear weights arrive as
anonymous transposed `onnx::MatMul_N` initialisers, found through the Add of their named bias — except the bias-free
up_gated_layer, which only its shape and position identify.  cs_bert_params_from_onnx's JinaBert branch
(codesearch_amd/csrc/onnx_reader.cpp) has to find its way through it (fastembed caches such an export for the registry's
jina-embeddings-v2-base-code entry, /root/reference/src/embed/embedder.rs:40-41, :112).  Every parameter is seeded noise.
Run: python tests/golden/make_jina_onnx_fixture.py"""
import math
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
H, NH, LAYERS, INNER, VOCAB = 64, 2, 2, 128, 48


def library_slopes(heads: int) -> torch.Tensor:
    from transformers.models.mpt.modeling_mpt import build_mpt_alibi_tensor

    a = build_mpt_alibi_tensor(heads, 4, alibi_bias_max=8)  # [heads, 1, 4]: slope_h * (j - 3)
    return (-a[:, 0, 2]).to(torch.float32)                  # position -1 -> the slope itself


# FIRST_FILE: jinaai/jina-bert-implementation's arrangement instead (no query / key LayerNorm; mlp.gated_layers whose FIRST half
# goes through the GELU, mlp.wo) — tests export that variant into a temporary directory
FIRST_FILE = False


class SelfAttention(nn.Module):
    def __init__(self):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        if not FIRST_FILE:
            self.layer_norm_q, self.layer_norm_k = nn.LayerNorm(H, eps=1e-12), nn.LayerNorm(H, eps=1e-12)

    def forward(self, x, bias):
        B, L, _ = x.shape
        dh = H // NH
        q, k, v = self.query(x), self.key(x), self.value(x)
        if not FIRST_FILE:
            q, k = self.layer_norm_q(q), self.layer_norm_k(k)

        def heads(t):
            return t.view(B, L, NH, dh).transpose(1, 2)

        s = torch.matmul(heads(q), heads(k).transpose(-1, -2)) / math.sqrt(dh) + bias
        return torch.matmul(torch.softmax(s, dim=-1), heads(v)).transpose(1, 2).reshape(B, L, H)


class SelfOutput(nn.Module):
    def __init__(self):
        super().__init__()
        self.dense = nn.Linear(H, H)
        self.LayerNorm = nn.LayerNorm(H, eps=1e-12)

    def forward(self, ctx, x):
        return self.LayerNorm(self.dense(ctx) + x)


class Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.self = SelfAttention()
        self.output = SelfOutput()

    def forward(self, x, bias):
        return self.output(self.self(x, bias), x)


class GLUMLP(nn.Module):
    def __init__(self):
        super().__init__()
        if FIRST_FILE:
            self.gated_layers = nn.Linear(H, 2 * INNER, bias=False)
            self.wo = nn.Linear(INNER, H)
        else:
            self.up_gated_layer = nn.Linear(H, 2 * INNER, bias=False)
            self.down_layer = nn.Linear(INNER, H)
        self.layernorm = nn.LayerNorm(H, eps=1e-12)

    def forward(self, x):
        if FIRST_FILE:
            both = self.gated_layers(x)
            gated, non_gated = both[:, :, :INNER], both[:, :, INNER:]
            return self.layernorm(self.wo(F.gelu(gated) * non_gated) + x)
        up_gated = self.up_gated_layer(x)
        up, gated = up_gated[:, :, :INNER], up_gated[:, :, INNER:]
        return self.layernorm(self.down_layer(up * F.gelu(gated)) + x)


class Layer(nn.Module):
    def __init__(self):
        super().__init__()
        self.attention, self.mlp = Attention(), GLUMLP()

    def forward(self, x, bias):
        return self.mlp(self.attention(x, bias))


class Embeddings(nn.Module):
    def __init__(self):
        super().__init__()
        self.word_embeddings = nn.Embedding(VOCAB, H)
        self.token_type_embeddings = nn.Embedding(2, H)
        self.LayerNorm = nn.LayerNorm(H, eps=1e-12)

    def forward(self, ids, tt):
        return self.LayerNorm(self.word_embeddings(ids) + self.token_type_embeddings(tt))


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer = nn.ModuleList([Layer() for _ in range(LAYERS)])


class JinaTiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.embeddings = Embeddings()
        self.encoder = Encoder()
        self.register_buffer("slopes", library_slopes(NH), persistent=False)

    def forward(self, input_ids, attention_mask, token_type_ids):
        x = self.embeddings(input_ids, token_type_ids)
        L = input_ids.shape[1]
        pos = torch.arange(L, dtype=torch.float32)
        dist = (pos[None, :] - pos[:, None]).abs()
        bias = -self.slopes[None, :, None, None] * dist[None, None] + (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
        for layer in self.encoder.layer:
            x = layer(x, bias)
        return x


def write(out_dir, stem, first_file=False, dims=None, opset=14):
    """dims = (hidden, heads, layers, intermediate, vocabulary): the GPU tests export at a width the kernels run (384)"""
    global FIRST_FILE, H, NH, LAYERS, INNER, VOCAB
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

    FIRST_FILE = first_file
    if dims:
        H, NH, LAYERS, INNER, VOCAB = dims
    onnx_proto_utils._add_onnxscript_fn = lambda proto, *a, **k: proto
    torch.manual_seed(20261006)
    model = JinaTiny().eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            ln_gain = ("LayerNorm.weight" in name) or ("layernorm.weight" in name) or ("layer_norm_q.weight" in name) or ("layer_norm_k.weight" in name)
            p.copy_(torch.randn_like(p) * 0.06 + (1.0 if ln_gain else 0.0))
    ids = torch.randint(0, VOCAB, (2, 8))
    mask = torch.ones(2, 8, dtype=torch.long)
    tt = torch.zeros(2, 8, dtype=torch.long)
    out = os.path.join(out_dir, stem + ".onnx")
    axes = {n: {0: "batch", 1: "seq"} for n in ("input_ids", "attention_mask", "token_type_ids")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model, (ids, mask, tt), out, input_names=list(axes), output_names=["last_hidden_state"], dynamic_axes=axes,
                          opset_version=opset, dynamo=False)
    g = torch.Generator().manual_seed(11)
    qids = torch.randint(0, VOCAB, (3, 24), generator=g)
    lens = [24, 17, 9]
    qmask = torch.zeros(3, 24, dtype=torch.long)
    for b, n in enumerate(lens):
        qmask[b, :n] = 1
    with torch.no_grad():
        hidden = model(qids, qmask, torch.zeros_like(qids))
    w = qmask[:, :, None].to(hidden.dtype)
    pooled = (hidden * w).sum(1) / w.sum(1)
    pooled = pooled / pooled.norm(dim=1, keepdim=True)
    np.savez_compressed(os.path.join(out_dir, stem + "_state.npz"), query_ids=qids.numpy().astype(np.int32),
                        query_mask=qmask.numpy().astype(np.int32), query_pooled=pooled.numpy().astype(np.float32),
                        **{k: v.numpy() for k, v in model.state_dict().items()})
    print("wrote", out, os.path.getsize(out), "bytes")
    return out


def main():
    write(HERE, "jina_tiny_export")


if __name__ == "__main__":
    sys.exit(main())
