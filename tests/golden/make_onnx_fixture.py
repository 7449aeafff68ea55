#!/usr/bin/env python3
"""Writes tests/golden/bert_tiny_export.onnx + bert_tiny_export_state.npz: a BERT encoder (hidden 64, 2 heads,
2 layers, FFN 128, vocabulary 48, 16 positions — 72k parameters) exported by torch.onnx's TorchScript exporter from
`transformers.BertModel`, and the state dict it was exported from.  This is a file produced by a real
exporter, not by this repo: parameters consumed directly keep their state-dict names, Linear weights arrive
as transposed anonymous `onnx::MatMul_N` initialisers — the shape cs_bert_params_from_onnx
(codesearch_amd/csrc/onnx_reader.cpp) has to understand (fastembed's cache holds such an export,
/root/reference/src/embed/embedder.rs:218-245).  Every parameter is seeded noise so that the exporter's
initialiser de-duplication (identical all-zero biases collapse into one tensor) cannot alias two of them.
The image has no `onnx` package; the exporter only needs it to attach onnxscript functions, which this
model has none of, so that one hook is bypassed.  Run: python tests/golden/make_onnx_fixture.py"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    from transformers import BertConfig, BertModel

    onnx_proto_utils._add_onnxscript_fn = lambda proto, *a, **k: proto
    cfg = BertConfig(vocab_size=48, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=128, max_position_embeddings=16)
    cfg._attn_implementation = "eager"
    torch.manual_seed(20260410)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn_like(p) * 0.05)

    class Wrapped(torch.nn.Module):  # positional ONNX inputs -> keyword arguments
        def __init__(self, m):
            super().__init__()
            self.bert = m

        def forward(self, input_ids, attention_mask, token_type_ids):
            return self.bert(input_ids=input_ids, attention_mask=attention_mask,
                             token_type_ids=token_type_ids).last_hidden_state

    ids = torch.randint(0, 48, (2, 8))
    mask = torch.ones(2, 8, dtype=torch.long)
    tt = torch.zeros(2, 8, dtype=torch.long)
    out = os.path.join(HERE, "bert_tiny_export.onnx")
    axes = {n: {0: "batch", 1: "seq"} for n in ("input_ids", "attention_mask", "token_type_ids")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(Wrapped(model), (ids, mask, tt), out, input_names=list(axes), output_names=["last_hidden_state"],
                          dynamic_axes=axes, opset_version=14, dynamo=False)
    np.savez_compressed(os.path.join(HERE, "bert_tiny_export_state.npz"),
                        **{k: v.numpy() for k, v in model.state_dict().items()})
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
