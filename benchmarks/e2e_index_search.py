#!/usr/bin/env python3
"""BASELINE.json configs[3]: end-to-end index + search on one MI355X — 100k synthetic code
chunks (token ids [100000, 256]) embedded on the GPU with the BGE-small-shaped encoder, kept
resident as the corpus, then 64 batched queries, top-10.  One JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=100_000)
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--texts", action="store_true",
                    help="start from strings: synthetic code-like chunks through cs_embedder_embed_texts "
                         "(WordPiece on host threads, overlapped with the device)")
    args = ap.parse_args()
    import numpy as np

    from codesearch_amd import BertConfig, FastEmbedder, ModelType, VectorStore
    from codesearch_amd.bert_params import synth_token_batch
    from codesearch_amd.pipeline import index_token_chunks, search_token_queries

    cfg = BertConfig.bge_small()
    if args.texts:
        return main_texts(args, cfg)
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202)
    ids, mask = synth_token_batch(cfg, 31337, args.chunks, args.seq, args.ragged)
    # queries = lightly edited copies of known chunks, so the right answer is known
    targets = [(i * 7919) % args.chunks for i in range(args.queries)]
    q_ids, q_mask = ids[targets].copy(), mask[targets].copy()
    q_ids[:, 5] = (q_ids[:, 5] + 1) % cfg.vocab_size
    emb.embed_ids(ids[:256], mask[:256])  # warm-up: allocate workspace
    store = VectorStore(None, cfg.hidden, capacity=args.chunks)
    t0 = time.perf_counter()
    t_index = index_token_chunks(emb, store, ids, mask)
    cos, rid, counts, t_search = search_token_queries(emb, store, q_ids, q_mask, args.k)
    wall = time.perf_counter() - t0
    hit = float(np.mean([rid[i][0] == targets[i] for i in range(args.queries)]))
    fwd_ms, fwd_n = emb.profile_read()
    print(json.dumps({
        "workload": f"index {args.chunks} chunks x {args.seq} tokens (BGE-small shape, fp32) + {args.queries} batched queries top-{args.k}",
        "chunks_per_s_end_to_end": args.chunks / wall, "wall_s": wall,
        "embed_s": t_index["embed_s"], "insert_build_s": t_index["insert_build_s"],
        "embed_queries_s": t_search["embed_queries_s"], "search_s": t_search["search_s"],
        "encoder_device_ms_per_batch": fwd_ms / max(fwd_n, 1), "encoder_batches": fwd_n,
        "top1_is_edited_source_chunk": hit, "mean_top1_cos": float(cos[:, 0].mean()),
    }))


def main_texts(args, cfg):
    import numpy as np

    from codesearch_amd import FastEmbedder, ModelType, VectorStore
    from codesearch_amd.pipeline import (index_text_chunks, search_text_queries, synth_code_texts,
                                         synth_vocab)
    from codesearch_amd.tokenizer import WordPieceTokenizer

    vocab = synth_vocab(cfg.vocab_size)
    tok = WordPieceTokenizer(vocab, max_length=cfg.max_position)
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202, tokenizer=tok)
    t0 = time.perf_counter()
    texts = synth_code_texts(vocab, args.chunks, 31337)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    ids, mask = tok.encode_batch(texts[:4096])
    tok_rate = float(mask.sum()) / (time.perf_counter() - t0)
    targets = [(i * 7919) % args.chunks for i in range(args.queries)]
    queries = [texts[t][: len(texts[t]) * 3 // 4] for t in targets]  # a prefix of a known chunk
    emb.embed_batch(texts[:256])  # warm-up: allocate workspace
    store = VectorStore(None, cfg.hidden, capacity=args.chunks)
    t0 = time.perf_counter()
    t_index = index_text_chunks(emb, store, texts)
    cos, rid, counts, t_search = search_text_queries(emb, store, queries, args.k)
    wall = time.perf_counter() - t0
    hit = float(np.mean([rid[i][0] == targets[i] for i in range(args.queries)]))
    fwd_ms, fwd_n = emb.profile_read()
    print(json.dumps({
        "workload": f"index {args.chunks} code-like text chunks (mean {float(mask.sum(1).mean()):.0f} tokens, max "
                    f"{int(mask.sum(1).max())}; mini-batches grouped by token count inside 4096-text windows unless "
                    f"CS_EMBED_LENGTH_SORT=0; BGE-small shape) from strings + {args.queries} text queries top-{args.k}",
        "chunks_per_s_end_to_end": args.chunks / wall, "wall_s": wall,
        "embed_s": t_index["embed_s"], "insert_build_s": t_index["insert_build_s"],
        "embed_queries_s": t_search["embed_queries_s"], "search_s": t_search["search_s"],
        "encoder_device_ms_per_batch": fwd_ms / max(fwd_n, 1), "encoder_batches": fwd_n,
        "tokenizer_tokens_per_s_standalone": tok_rate, "text_generation_s": gen_s,
        "top1_is_source_chunk_of_prefix_query": hit, "mean_top1_cos": float(cos[:, 0].mean()),
    }))


if __name__ == "__main__":
    main()
