"""The reference's own embedding checks, for whoever holds the weights ("encoder parity unpinned" is
structural — the reference holds no embedding vector and no model file exists offline).

Skipped unless CS_REAL_MODEL_DIR names a model directory as hf-hub leaves it in fastembed's cache (config.json; the
weights as onnx/model.onnx | model.onnx | model_optimized.onnx | onnx/model_quantized.onnx | model.safetensors;
tokenizer.json or vocab.txt), e.g.
the snapshot of Xenova/bge-small-en-v1.5 or BAAI/bge-small-en-v1.5:

    CS_REAL_MODEL_DIR=/path/to/snapshot python -m pytest tests/test_gpu_real_model.py -m gpu -q

Loads through cs_embedder_create_from_dir + cs_tokenizer_create_from_dir (FastEmbedder.from_dir) and runs, literally,
  * /root/reference/src/embed/embedder.rs:453-463  test_embed_single_text: 384 values, |v| within 0.1 of 1;
  * embedder.rs:466-483                            test_embed_batch: three texts, three vectors of the model's width;
  * embedder.rs:485-506                            test_semantic_similarity: sim(fox, fox') > sim(fox, python) and > 0.7;
and, when the directory also holds golden.npz written by tests/golden/make_real_model_golden.py (onnxruntime or
transformers on a CPU, in a container that has them: never on the GPU box), the embeddings of its texts within 1e-4
(the north star's bound on cosine scores) and every pairwise cosine within 1e-4."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODEL_DIR = os.environ.get("CS_REAL_MODEL_DIR")
needs_model = pytest.mark.skipif(not MODEL_DIR, reason="CS_REAL_MODEL_DIR is not set (no model weights offline)")


@pytest.fixture(scope="module")
def embedder():
    from codesearch_amd import FastEmbedder

    emb = FastEmbedder.from_dir(MODEL_DIR)
    assert emb.tokenizer is not None, "the directory holds neither tokenizer.json nor vocab.txt"
    yield emb
    emb.close()


def _cos(a, b):
    # embedder.rs:508-513 cosine_similarity
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))


@needs_model
def test_embed_single_text(embedder):
    v = embedder.embed_one("Hello, world!")
    assert len(v) == embedder.dimensions()
    if os.environ.get("CS_REAL_MODEL_IS_DEFAULT", "1") == "1" and embedder.dimensions() != 384:
        pytest.skip("not the reference's default model (384-d): the width check of embedder.rs:460 does not apply")
    assert abs(float(np.linalg.norm(v)) - 1.0) < 0.1


@needs_model
def test_embed_batch(embedder):
    out = embedder.embed_batch(["Hello, world!", "Rust is awesome", "Code search with AI"])
    assert len(out) == 3 and all(len(e) == embedder.dimensions() for e in out)


@needs_model
def test_semantic_similarity(embedder):
    e1 = embedder.embed_one("The quick brown fox jumps over the lazy dog")
    e2 = embedder.embed_one("A fast auburn fox leaps over a sleepy canine")
    e3 = embedder.embed_one("Python is a programming language")
    s12, s13 = _cos(e1, e2), _cos(e1, e3)
    assert s12 > s13, (s12, s13)
    assert s12 > 0.7, s12


@needs_model
def test_embeddings_match_the_cpu_runtime_golden(embedder):
    path = os.path.join(MODEL_DIR, "golden.npz")
    if not os.path.exists(path):
        pytest.skip("no golden.npz beside the model: run tests/golden/make_real_model_golden.py where onnxruntime or "
                    "transformers is importable")
    g = np.load(path, allow_pickle=False)
    texts = [t for t in g["texts"].tolist()]
    want = g["embeddings"].astype(np.float64)
    got = np.stack(embedder.embed_batch(texts)).astype(np.float64)
    assert got.shape == want.shape
    # token ids first: a tokenizer difference would otherwise read as a numerics difference
    ids, mask = embedder.tokenizer.encode_batch(texts)
    for i, t in enumerate(texts):
        n = int(mask[i].sum())
        assert ids[i][:n].tolist() == g["input_ids"][i][: int(g["lengths"][i])].tolist(), f"tokenisation differs: {t!r}"
    # A dynamically quantised model (a *Q directory: model_quantized.onnx, run as CS_GEMM_Q8_DYNAMIC) has an 8-bit rounding
    # behind every Linear: two f32-class evaluations of the same graph flip a few activation bytes (ORT on
    # two CPUs differs from itself the same way), so the bound is the flip noise, not f32 rounding.  The golden must then
    # have been made with the SAME call units (make_real_model_golden.py embeds its texts in one call, as this does).
    tol = 3e-3 if embedder.gemm_mode() == "q8" else 1e-4
    assert np.abs(got - want).max() < tol, np.abs(got - want).max()
    gn = got / np.linalg.norm(got, axis=1, keepdims=True)
    wn = want / np.linalg.norm(want, axis=1, keepdims=True)
    assert np.abs(gn @ gn.T - wn @ wn.T).max() < tol
