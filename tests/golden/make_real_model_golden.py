#!/usr/bin/env python3
"""golden.npz for tests/test_gpu_real_model.py: embeddings of a fixed text list by a CPU runtime, written beside the model.

    python tests/golden/make_real_model_golden.py /path/to/model_dir [--pooling cls|mean]

Runs where `onnxruntime` (the reference's own runtime: fastembed 5.8.1 -> ort 2.0.0-rc.11) or `transformers` + `torch`
is importable and a model directory exists — the build container has transformers but no weights, the GPU box has
neither: the file travels with the model directory, never with this repo.  What it restates of the reference:
tokenise with the directory's tokenizer.json (truncation 512, batch-longest padding: fastembed's settings), run the
graph in f32, take last_hidden_state, pool (CLS for the BGE family, mean otherwise: fastembed's per-model choice),
L2-normalise with v / max(|v|, 1e-12)."""
import argparse
import json
import os
import sys

import numpy as np

TEXTS = [
    "Hello, world!",
    "Rust is awesome",
    "Code search with AI",
    "The quick brown fox jumps over the lazy dog",
    "A fast auburn fox leaps over a sleepy canine",
    "Python is a programming language",
    "fn cosine_similarity(a: &[f32], b: &[f32]) -> f32 { let dot: f32 = a.iter().zip(b.iter()).map(|(x, y)| x * y).sum(); dot }",
    "File: src/vectordb/store.rs\nFunction: search\nSignature: pub fn search(&self, query_embedding: &[f32], limit: usize)\n"
    "Code:\nlet results = reader.nns(limit).by_vector(&rtxn, query_embedding)?;",
    "",
    "naïve café ÅNGSTRÖM ﬁnal — unicode: accents, ligatures, dashes, 漢字",
]


def run_onnx(model_dir, ids, mask):
    import onnxruntime as ort

    # (the *Q registry entries cache onnx/model_quantized.onnx: onnxruntime then runs the dynamically quantised graph, which
    # is what the library's CS_GEMM_Q8_DYNAMIC mode restates; the transformers fallback below cannot stand in for it)
    for rel in ("onnx/model.onnx", "model.onnx", "model_optimized.onnx", "onnx/model_quantized.onnx", "model_quantized.onnx"):
        p = os.path.join(model_dir, rel)
        if os.path.exists(p):
            break
    else:
        raise FileNotFoundError("no ONNX file in the model directory")
    sess = ort.InferenceSession(p, providers=["CPUExecutionProvider"])
    names = {i.name for i in sess.get_inputs()}
    feed = {"input_ids": ids.astype(np.int64), "attention_mask": mask.astype(np.int64)}
    if "token_type_ids" in names:
        feed["token_type_ids"] = np.zeros_like(feed["input_ids"])
    return sess.run(None, feed)[0], f"onnxruntime {ort.__version__} ({rel})"


def run_transformers(model_dir, ids, mask):
    import torch
    import transformers

    model = transformers.AutoModel.from_pretrained(model_dir, add_pooling_layer=False).eval().float()
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(mask.astype(np.int64)))
    return out.last_hidden_state.numpy(), f"transformers {transformers.__version__} (f32, CPU)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model_dir")
    ap.add_argument("--pooling", choices=("cls", "mean"), default=None)
    args = ap.parse_args()
    from tokenizers import Tokenizer

    tok = Tokenizer.from_file(os.path.join(args.model_dir, "tokenizer.json"))
    tok.enable_truncation(max_length=512)
    tok.enable_padding()
    enc = tok.encode_batch(TEXTS)
    ids = np.array([e.ids for e in enc], np.int32)
    mask = np.array([e.attention_mask for e in enc], np.int32)
    try:
        hidden, how = run_onnx(args.model_dir, ids, mask)
    except Exception as e:  # no onnxruntime, or no ONNX file
        print(f"onnxruntime route unavailable ({type(e).__name__}: {e}); using transformers", file=sys.stderr)
        hidden, how = run_transformers(args.model_dir, ids, mask)
    pooling = args.pooling
    if pooling is None:
        pc = os.path.join(args.model_dir, "1_Pooling", "config.json")
        pooling = "mean" if os.path.exists(pc) and json.load(open(pc)).get("pooling_mode_mean_tokens") else "cls"
    if pooling == "cls":
        v = hidden[:, 0, :].astype(np.float64)
    else:
        m = mask[:, :, None].astype(np.float64)
        v = (hidden.astype(np.float64) * m).sum(axis=1) / np.maximum(m.sum(axis=1), 1e-9)
    v = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)
    out = os.path.join(args.model_dir, "golden.npz")
    np.savez(out, texts=np.array(TEXTS), embeddings=v.astype(np.float32), input_ids=ids, lengths=mask.sum(axis=1).astype(np.int32),
             pooling=np.array(pooling), runtime=np.array(how))
    print(f"wrote {out}: {len(TEXTS)} texts, pooling {pooling}, {how}")


if __name__ == "__main__":
    main()
