#!/usr/bin/env python3
"""Encoder throughput at BASELINE.json configs[2]: BGE-small shape, batch 256, seq 256.
Prints one JSON line: sequences ("chunks") embedded per second, ms per batch, achieved
TFLOP/s against the fp32 MFMA peak (157.3 TF, MI355X_MICROARCH.md)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--stages", action="store_true", help="also print per-kernel-class microseconds per layer (one stream)")
    ap.add_argument("--model", default="bge-small", help="registry short name: bge-small, bge-base, bge-large, minilm-l6, ...")
    ap.add_argument("--quant", default="", help="u8 | s8 | u8c | s8c: quantise the Linear weights as onnxruntime's quantize_dynamic "
                    "does (per tensor / per channel) and run the dynamic-quantisation mode (the *Q models)")
    args = ap.parse_args()
    import numpy as np

    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import synth_token_batch

    mt = ModelType.parse(args.model)
    if mt is None:
        raise SystemExit(f"unknown model {args.model!r}")
    cfg = mt.bert_config()
    if args.quant:
        from codesearch_amd.bert_params import quantize_linear_weights, synth_params

        params, wscale = quantize_linear_weights(cfg, synth_params(cfg, 202), per_channel=args.quant.endswith("c"),
                                                 unsigned=args.quant.startswith("u"))
        emb = FastEmbedder(mt, config=cfg, params=params, wscale=wscale)
    else:
        emb = FastEmbedder(mt, config=cfg, seed=202)
    ids, mask = synth_token_batch(cfg, 999, args.batch, args.seq, args.ragged)
    emb.embed_ids(ids, mask)  # warm-up (allocates the workspace)
    emb.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = emb.embed_ids(ids, mask)
    wall = (time.perf_counter() - t0) / args.iters
    ms, n = emb.profile_read()
    ms /= max(args.iters, 1)  # device time per BATCH (the reference's mini-batch policy cuts 768-d models at 128, 1024-d at 64)
    stages = None
    if args.stages:
        emb.profile_stages(True)
        emb.embed_ids(ids, mask)
        emb.profile_stages_read(reset=True)
        emb.profile_read(reset=True)
        for _ in range(3):
            emb.embed_ids(ids, mask)
        st, nf = emb.profile_stages_read()
        ms1, n1 = emb.profile_read()
        emb.profile_stages(False)
        stages = {k: round(v / (cfg.layers if k not in ("embed_ln", "pool_normalize") else 1), 1) for k, v in st.items()}
        stages["one_stream_ms_per_forward"] = round(ms1 / max(n1, 1), 3)
    L, H, I, layers = args.seq, cfg.hidden, cfg.intermediate, cfg.layers
    up = 3 if getattr(cfg, "arch", 0) in (1, 2, 3, 4) else 2  # gated feed-forwards (NomicBert, JinaBert, ModernBERT): value and gate projections in front of the down projection
    flops_tok = layers * (2 * (4 * H * H + up * H * I) + 4 * L * H)
    flops = flops_tok * args.batch * args.seq
    print(json.dumps({
        "workload": f"{mt.name_str()} shape ({cfg.layers} x hidden {cfg.hidden}, {cfg.heads} heads), batch {args.batch} x seq {args.seq}, {'ragged' if args.ragged else 'full'} mask, {'dynamic int8 Linears (' + args.quant + ')' if args.quant else 'fp32'}",
        "device_ms_per_batch": ms, "wall_ms_per_batch_incl_pcie": wall * 1e3,
        "chunks_per_s_device": args.batch / (ms * 1e-3), "tokens_per_s_device": args.batch * args.seq / (ms * 1e-3),
        "algorithmic_tflop_per_batch": flops / 1e12, "achieved_tflops": flops / (ms * 1e-3) / 1e12,
        "peak_tflops_f32_mfma": 157.3, "frac": flops / (ms * 1e-3) / 1e12 / 157.3,
        "norm0": float(np.linalg.norm(out[0])), "stages_us_per_layer": stages,
    }))


if __name__ == "__main__":
    main()
