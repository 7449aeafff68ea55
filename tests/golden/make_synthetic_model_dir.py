#!/usr/bin/env python3
"""A stand-in model directory for REHEARSING tests/test_gpu_real_model.py where no real weights exist: a 4-layer
BGE-small-shaped BERT with seeded random weights, written exactly as a HF snapshot (config.json, model.safetensors,
tokenizer.json, tokenizer_config.json).  `make_real_model_golden.py <dir>` then writes golden.npz beside it with
transformers on the CPU, and CS_REAL_MODEL_DIR=<dir> runs the GPU test against it (the semantic-similarity test needs
trained weights and is expected to fail on this directory; the others must pass).

    python tests/golden/make_synthetic_model_dir.py build/synth_model"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "synth_model")
    os.makedirs(out, exist_ok=True)
    from safetensors.numpy import save_file
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    from codesearch_amd.bert_params import POOL_CLS, BertConfig, synth_params, to_state_dict
    from codesearch_amd.pipeline import synth_vocab

    vocab = synth_vocab(2048)
    # real words of the golden texts, so that not everything is [UNK] / single characters
    extra = ("hello world rust is awesome code search with ai the quick brown fox jumps over lazy dog a fast auburn leaps "
             "sleepy canine python programming language fn let file function signature pub self query limit results").split()
    for w in extra:
        if w not in vocab:
            vocab[w] = len(vocab)
    cfg = BertConfig(vocab_size=len(vocab), layers=4, max_position=512, pooling=POOL_CLS)
    flat = synth_params(cfg, 4242)
    config = {"model_type": "bert", "architectures": ["BertModel"], "vocab_size": cfg.vocab_size, "hidden_size": 384,
              "num_hidden_layers": cfg.layers, "num_attention_heads": 12, "intermediate_size": 1536,
              "max_position_embeddings": 512, "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu",
              "pad_token_id": vocab["[PAD]"]}
    json.dump(config, open(os.path.join(out, "config.json"), "w"))
    save_file({k: np.ascontiguousarray(v) for k, v in to_state_dict(cfg, flat).items()}, os.path.join(out, "model.safetensors"))
    tk = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
    tk.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=True)
    tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                      special_tokens=[("[CLS]", vocab["[CLS]"]), ("[SEP]", vocab["[SEP]"])])
    tk.add_special_tokens(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
    tk.save(os.path.join(out, "tokenizer.json"))
    json.dump({"do_lower_case": True, "model_max_length": 512}, open(os.path.join(out, "tokenizer_config.json"), "w"))
    print(f"wrote {out}: {cfg.layers} layers, vocab {cfg.vocab_size}")


if __name__ == "__main__":
    main()
