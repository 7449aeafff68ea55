"""Generates tests/golden/encoder_golden.npz — known answers for the encoder.

Run in the build container:  python tests/golden/make_encoder_golden.py

Source of truth: HF `transformers.BertModel` (add_pooling_layer=False, eager attention) in
float64 on CPU — a third-party implementation of the same published architecture the
reference runs through fastembed/ONNX Runtime (BAAI/bge-small-en-v1.5).  It is NOT the
reference (which is Rust and ships no embedding vectors), so encoder parity stays
"unpinned against the reference" (DESIGN.md); these vectors pin the oracle and the HIP path
to an independent implementation.  Weights come from the integer generator of
include/cs_bert_params.h (numpy mirror codesearch_amd/bert_params.py), so only seeds and
outputs are stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from codesearch_amd.bert_params import BertConfig, synth_params, synth_token_batch, to_state_dict  # noqa: E402


def hf_forward(cfg: BertConfig, flat, ids, mask):
    from transformers import BertConfig as HFConfig
    from transformers import BertModel

    hf = HFConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                  num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
                  max_position_embeddings=cfg.max_position, type_vocab_size=cfg.type_vocab_size,
                  layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu", hidden_dropout_prob=0.0,
                  attention_probs_dropout_prob=0.0, attn_implementation="eager")
    model = BertModel(hf, add_pooling_layer=False).double().eval()
    sd = {k: torch.from_numpy(v.astype(np.float64)) for k, v in to_state_dict(cfg, flat).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "position_ids" not in m], missing
    assert not unexpected, unexpected
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids.astype(np.int64)),
                    attention_mask=torch.from_numpy(mask.astype(np.int64)),
                    output_hidden_states=True)
    hs = [h.numpy() for h in out.hidden_states]  # embeddings + each layer
    last = out.last_hidden_state.numpy()
    m = mask.astype(np.float64)[:, :, None]
    cls = last[:, 0, :]
    mean = (last * m).sum(1) / np.maximum(m.sum(1), 1e-9)

    def norm(v):
        return v / (np.linalg.norm(v, axis=1, keepdims=True) + 1e-12)

    return hs, norm(cls), norm(mean)


def main():
    out = {}
    cases = []
    tiny = dict(vocab_size=512, hidden=384, layers=2, heads=12, intermediate=1536, max_position=512)
    for L in (7, 16, 64):
        cases.append(("tiny_L%d" % L, tiny, 101, 4, L, True))
    cases.append(("tiny_full_mask", tiny, 101, 3, 32, False))
    full = dict(vocab_size=30522, hidden=384, layers=12, heads=12, intermediate=1536, max_position=512)
    cases.append(("full_ragged", full, 202, 8, 256, True))
    cases.append(("full_dense", full, 202, 4, 128, False))
    names = []
    for name, kw, wseed, B, L, ragged in cases:
        cfg = BertConfig(**kw)
        flat = synth_params(cfg, wseed)
        ids, mask = synth_token_batch(cfg, wseed + 50, B, L, ragged)
        hs, cls, mean = hf_forward(cfg, flat, ids, mask)
        names.append(name)
        out[name + "/meta"] = np.array([kw["vocab_size"], kw["hidden"], kw["layers"], kw["heads"],
                                        kw["intermediate"], kw["max_position"], wseed, wseed + 50, B, L,
                                        int(ragged)], np.int64)
        out[name + "/cls"] = cls
        out[name + "/mean"] = mean
        # per-layer fingerprints of the hidden states (valid tokens only): mean |h| and three probes
        valid = mask.astype(bool)
        out[name + "/layer_absmean"] = np.array([np.abs(h[valid]).mean() for h in hs])
        out[name + "/layer_probe"] = np.array([[h[0, 0, 0], h[B - 1, 1, 7], h[0, mask[0].sum() - 1, 383]] for h in hs])
        out[name + "/last_row0"] = hs[-1][0, 0, :]
        print(name, "cls[0,:3]", cls[0, :3], "cos(cls,mean)[0]", float((cls[0] * mean[0]).sum()))
    out["names"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "encoder_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
