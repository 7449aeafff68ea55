#!/usr/bin/env python3
"""Writes tests/golden/nomic_tiny_export.onnx + nomic_tiny_export_state.npz: a NomicBert-shaped encoder (hidden 64, 2 heads
of 32, 2 layers, n_inner 128, vocabulary 48; fused bias-free Wqkv, rotary positions on Q / K, swiglu feed-forward
fc2(fc11(x) * silu(fc12(x))), post-norm — the module and parameter names of the model repository's modelling file: emb_ln,
encoder.layers.N.attn.Wqkv / out_proj, norm1, mlp.fc11 / fc12 / fc2, norm2) exported by torch.onnx's TorchScript exporter,
and the state dict it was exported from.  The MODULE below is this repo's restatement of that structure (the modelling
file is not reachable from here); the FILE is a real exporter's output: bias-free Linear weights arrive as anonymous
transposed `onnx::MatMul_N` initialisers, silu as Sigmoid + Mul, the attention products as MatMuls between activations —
the shape cs_bert_params_from_onnx's NomicBert branch (codesearch_amd/csrc/onnx_reader.cpp) has to find its way through
(fastembed caches such an export for the registry's Nomic entries, /root/reference/src/embed/embedder.rs:36-37, :218-245).
Every parameter is seeded noise (no two tensors alike: the exporter de-duplicates identical initialisers).
Run: python tests/golden/make_nomic_onnx_fixture.py"""
import math
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
H, NH, LAYERS, INNER, VOCAB, BASE = 64, 2, 2, 128, 48, 1000.0


class Attn(nn.Module):
    def __init__(self):
        super().__init__()
        self.Wqkv = nn.Linear(H, 3 * H, bias=False)
        self.out_proj = nn.Linear(H, H, bias=False)

    def forward(self, x, bias, cos, sin):
        B, L, _ = x.shape
        dh = H // NH
        qkv = self.Wqkv(x).view(B, L, 3, NH, dh)
        q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)

        def rot(t):
            t1, t2 = t[..., : dh // 2], t[..., dh // 2:]
            return torch.cat([t1 * cos - t2 * sin, t2 * cos + t1 * sin], dim=-1)

        s = torch.matmul(rot(q), rot(k).transpose(-1, -2)) / math.sqrt(dh) + bias
        ctx = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, L, H)
        return self.out_proj(ctx)


class Mlp(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc11 = nn.Linear(H, INNER, bias=False)
        self.fc12 = nn.Linear(H, INNER, bias=False)
        self.fc2 = nn.Linear(INNER, H, bias=False)

    def forward(self, x):
        y = self.fc11(x)
        gate = self.fc12(x)
        return self.fc2(y * F.silu(gate))


class Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.attn, self.mlp = Attn(), Mlp()
        self.norm1, self.norm2 = nn.LayerNorm(H, eps=1e-12), nn.LayerNorm(H, eps=1e-12)

    def forward(self, x, bias, cos, sin):
        x = self.norm1(self.attn(x, bias, cos, sin) + x)
        return self.norm2(self.mlp(x) + x)


class Embeddings(nn.Module):
    def __init__(self):
        super().__init__()
        self.word_embeddings = nn.Embedding(VOCAB, H)
        self.token_type_embeddings = nn.Embedding(2, H)

    def forward(self, ids, tt):
        return self.word_embeddings(ids) + self.token_type_embeddings(tt)


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList([Block() for _ in range(LAYERS)])


class NomicTiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.embeddings = Embeddings()
        self.emb_ln = nn.LayerNorm(H, eps=1e-12)
        self.encoder = Encoder()

    def forward(self, input_ids, attention_mask, token_type_ids):
        x = self.emb_ln(self.embeddings(input_ids, token_type_ids))
        L = input_ids.shape[1]
        dh = H // NH
        inv_freq = 1.0 / (BASE ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
        ang = torch.outer(torch.arange(L, dtype=torch.float32), inv_freq)
        cos, sin = torch.cos(ang)[None, None], torch.sin(ang)[None, None]
        bias = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
        for layer in self.encoder.layers:
            x = layer(x, bias, cos, sin)
        return x


def main():
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

    onnx_proto_utils._add_onnxscript_fn = lambda proto, *a, **k: proto
    torch.manual_seed(20261005)
    model = NomicTiny().eval()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    ids = torch.randint(0, VOCAB, (2, 8))
    mask = torch.ones(2, 8, dtype=torch.long)
    tt = torch.zeros(2, 8, dtype=torch.long)
    out = os.path.join(HERE, "nomic_tiny_export.onnx")
    axes = {n: {0: "batch", 1: "seq"} for n in ("input_ids", "attention_mask", "token_type_ids")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model, (ids, mask, tt), out, input_names=list(axes), output_names=["last_hidden_state"],
                          dynamic_axes=axes, opset_version=14, dynamo=False)
    np.savez_compressed(os.path.join(HERE, "nomic_tiny_export_state.npz"), **{k: v.numpy() for k, v in model.state_dict().items()})
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
