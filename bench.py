#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json): brute-force cosine
top-10 over a 10M x 384 fp32 corpus resident in HBM, one MI355X per rank.

A "step" = one pass of the hot path over one batch of synthetic input: the query batch
(already in HBM) is scored against every row of the rank's shard by the fused scan +
top-k kernel, block partials are merged, and for N > 1 the per-shard top-k lists are
all-gathered over RCCL and merged.  `value` = corpus rows ("chunks") searched per second
over all ranks = N * rows_per_gpu * nq * K / t.

  python bench.py                      # N=1, 10M x 384, Q=1, k=10
  python bench.py --gpus N             # ONE process drives N GPUs through cs_shards_* (what a Rust VectorStore
                                       # inside `codesearch search` would call): shard g = rows [g*10M, (g+1)*10M)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
                                       # one rank per GPU, RCCL broadcast + all-gather (codesearch_amd/sharded.py)

Adds to the JSON line:
  roofline     — the scan kernel against the 8 TB/s HBM peak: algorithmic bytes per launch
                 (rows*dim*4) / HIP-event duration of that kernel, measured live.
  cpu_baseline — the CPU oracle's tuned port timed on this box's host cores over a bounded
                 sample (reported baseline, not the target).
  encoder      — BASELINE configs[2] (BGE-small shape, 256 x 256 tokens) with its own roofline
                 (f16 MFMA), per-kernel microseconds and a CPU baseline (oracle/bert_oracle.c at the
                 reference's effective batch of 32); the reference's 32-chunk call shape beside it.
  embed_search / embedded_and_searched_chunks_per_s — the literal wording of BASELINE's metric.
  e2e_index_search — BASELINE configs[3]: 100k chunks embedded on the GPU, then 64 queries top-10.
`--only-scan` runs the timed loop alone (for rocprofv3: one kernel population per CSV row).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
# benchmarks/mfma_rate_probe.hip (profiles/r03_mfma_rate_probe.log): what a loop of nothing but independent MFMAs sustains
# chip-wide when every instruction gets different pseudo-random operands (constant operands reach 2,040 / 4,810): the chip is
# power-limited under toggling data.  Reported beside the data-sheet peaks, never instead of them.
MFMA_F16_16x16x32_SUSTAINED_TFLOPS = 1490.0
MFMA_I8_32x32x32_SUSTAINED_TOPS = 3430.0
MFMA_I8_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md MFMA table: I8 32x32x32 / 16x16x64 = 2x the BF16 rate per clock
SEED = 0xC0DE5EA


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows per GPU")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--nq", type=int, default=1, help="queries per step")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encoder", action="store_true", help="skip the encoder / embed+search legs (N=1)")
    ap.add_argument("--no-1m", action="store_true", help="skip the configs[1] leg (1M rows)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the configs[3] leg (100k chunks indexed, 64 queries)")
    ap.add_argument("--e2e-chunks", type=int, default=100_000)
    ap.add_argument("--route", choices=("stream", "cost", "filter"), default="stream",
                    help="one query per step: stream = the f32 streaming scan (the north-star kernel `value` is quoted on; "
                         "default), cost = the library's default route (int8 filter + exact refine from 32,768 rows on at k < 48, 300,000 above), filter")
    ap.add_argument("--no-rccl-child", action="store_true",
                    help="--gpus N without a launcher: do not run the one-rank-per-GPU RCCL form as child processes first")
    ap.add_argument("--no-config5", action="store_true", help="--gpus N without a launcher: skip the 1,000-query leg")
    ap.add_argument("--rccl-child-timeout", type=float, default=200.0,
                    help="--gpus N without a launcher: seconds the RCCL child run may take before its process group is ended "
                         "(well under the driver's own limit: the one-process line must still be printed)")
    ap.add_argument("--only-scan", action="store_true",
                    help="timed loop only: no CPU baseline, no 1M / filter / encoder / e2e legs, no recall sample")
    a = ap.parse_args()
    if a.only_scan:
        a.no_cpu_baseline = a.no_encoder = a.no_1m = a.no_e2e = True
    return a


def _encoder_flops(cfg, B, L):
    H, I, layers = cfg.hidden, cfg.intermediate, cfg.layers
    gemm = layers * 2 * (4 * H * H + 2 * H * I) * B * L   # E2 + E4 + E5 + E6
    attn = layers * 4 * L * H * B * L                     # E3: QK^T and PV
    return gemm, attn


def _encoder_flops_executed(cfg, B, L):
    """What the device runs: a CLS-pooled model's LAST layer computes only what the embedding reads (csrc/cls_tail.hip,
    from 4,096 tokens per mini-batch): K and V for every token, then one query per sequence and B rows through the rest."""
    H, I, layers = cfg.hidden, cfg.intermediate, cfg.layers
    gemm, attn = _encoder_flops(cfg, B, L)
    tail = (cfg.pooling == 0 and B * L >= int(os.environ.get("CS_ENCODER_CLS_TAIL_MIN_TOKENS", "4096")) and L >= 16
            and os.environ.get("CS_ENCODER_CLS_TAIL", "1")[0] != "0")
    if not tail:
        return gemm, attn, False
    per_layer_gemm, per_layer_attn = gemm // layers, attn // layers
    last_gemm = 2 * (2 * H * H) * B * L + 2 * (2 * H * H + 2 * H * I) * B   # K, V for all tokens; Q, out-proj, FFN for B rows
    last_attn = 4 * L * H * B                                                 # one query per sequence
    return gemm - per_layer_gemm + last_gemm, attn - per_layer_attn + last_attn, True


def encoder_cpu_baseline(cfg, seed):
    """the C restatement of the encoder (oracle/bert_oracle.c, OpenMP) at the reference's
    effective batch — BatchEmbedder hands FastEmbedder 32 chunks at a time (src/embed/batch.rs:70,94) —
    on this job's host cores.  Bounded: two forwards of 32 x 256 tokens."""
    from codesearch_amd.bert_params import synth_token_batch
    from tests.oracle_lib import load_oracle

    oracle = load_oracle()
    threads = host_cpu_share(oracle.num_threads())
    os.environ["OMP_NUM_THREADS"] = str(threads)
    try:
        oracle.lib.omp_set_num_threads(threads)
    except AttributeError:
        pass
    params = oracle.bert_synth_params(cfg, seed)
    B, L = 32, 256
    ids, mask = synth_token_batch(cfg, 999, B, L, False)
    reps, total = 0, 0.0
    while reps < 2 or (total < 6.0 and reps < 6):
        t0 = time.perf_counter()
        oracle.bert_forward(cfg, params, ids, mask)
        total += time.perf_counter() - t0
        reps += 1
    return {"value": B * reps / total, "unit": "chunks/s", "cores": threads, "kind": "port",
            "sample": f"{reps} forwards of {B} x {L} tokens (the reference's 32-chunk embed_chunks slices) through "
                      f"oracle/bert_oracle.c cs_oracle_bert_forward, f32 scalar + OpenMP, {threads} threads = this job's "
                      f"CPU share (the host shows {oracle.num_threads()}); the reference publishes 19.6 chunks/s for "
                      "bge-small on its own (unstated) CPU",
            "seconds_per_forward": total / reps}


def encoder_legs(shard, k, device, with_cpu=True):
    """BASELINE.json's metric also names embedding: (i) the BGE-small-shaped encoder alone at configs[2] as BASELINE
    words it (batch 256 x seq 256, mean-pool + L2-norm; synthetic weights) with its own roofline and per-kernel times,
    and the CLS-pooled variant the reference's BGE model actually uses (fastembed pools BGE by CLS: the last layer then
    computes only what the embedding reads); (ii) the reference's real call shape (32 chunks per embed call,
    src/embed/batch.rs:70); (iii) embed a batch of 256 chunks then search them against the resident corpus ("chunks
    embedded+searched/sec"), serial and pipelined; (iv) the same encoder at the tokenizer's truncation length (128 x 512
    tokens: chunks run to 2,000 characters, src/chunker/semantic.rs:22-29).
    Reported next to `value`, which stays the scan (north-star target)."""
    import dataclasses

    import torch

    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import POOL_MEAN, synth_token_batch

    cfg = BertConfig.bge_small()
    B, L = 256, 256
    ids, mask = synth_token_batch(cfg, 999, B, L, False)
    dev = f"cuda:{device}"
    d_q = [torch.empty((B, cfg.hidden), dtype=torch.float32, device=dev) for _ in range(2)]
    iters = 5
    side = torch.cuda.Stream(device=dev)
    from codesearch_amd import _lib as _cslib

    _l = _cslib.load()
    pk = [torch.zeros(B * k, dtype=torch.int64, device=dev) for _ in range(2)]

    def search_nowait(buf, keys):
        """cs_index_search_device on torch's current stream, never waiting for the device: ShardedVectorStore's search of
        more than 16 queries asks cs_index_search_status right away (one stream synchronisation); the pipelined loop asks
        once, after the loop (the status word is sticky)."""
        _cslib.check(_l.cs_index_search_device(shard.store.handle, C.c_void_p(buf.data_ptr()), B, cfg.hidden, k,
                                               C.c_void_p(keys.data_ptr()), None, None, None,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def measure(emb):
        """device ms per forward alone, per-kernel stage times, and the embed+search loops"""
        emb.embed_ids_to_device(ids, mask, d_q[0].data_ptr())  # warm-up, allocates the workspace
        shard.search_device(d_q[0], B, k)
        torch.cuda.synchronize()
        emb.profile_read(reset=True)
        for _ in range(iters):
            emb.embed_ids_to_device(ids, mask, d_q[0].data_ptr())
        torch.cuda.synchronize()
        ms, n = emb.profile_read()
        ms /= max(n, 1)
        # serial: embed, search, wait — what a caller that needs each batch's results before the next one sees
        t0 = time.perf_counter()
        for _ in range(iters):
            emb.embed_ids_to_device(ids, mask, d_q[0].data_ptr())
            shard.search_device(d_q[0], B, k)
            torch.cuda.synchronize()
        serial = (time.perf_counter() - t0) / iters
        # pipelined: the search of batch i runs on a second stream (its own query buffer) under the forward of batch
        # i + 1; the loop never waits for a search until the end
        emb.profile_read(reset=True)
        t0 = time.perf_counter()
        for i in range(2 * iters):
            buf = d_q[i & 1]
            emb.embed_ids_to_device(ids, mask, buf.data_ptr())   # returns when the forward is complete
            with torch.cuda.stream(side):
                search_nowait(buf, pk[i & 1])
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t0) / (2 * iters)
        assert not shard.store.search_status(side.cuda_stream), "a pipelined search overflowed a candidate buffer"
        # ... and they return what the blocking call returns
        ref = shard.search_device(d_q[1], B, k)
        torch.cuda.synchronize()
        assert torch.equal(ref["keys"], pk[1]), "pipelined search differs from the blocking one"
        ms_overlapped, n2 = emb.profile_read()
        ms_overlapped /= max(n2, 1)
        emb.profile_stages(True)
        emb.embed_ids_to_device(ids, mask, d_q[0].data_ptr())
        emb.profile_stages_read(reset=True)
        emb.profile_read(reset=True)
        for _ in range(3):
            emb.embed_ids_to_device(ids, mask, d_q[0].data_ptr())
        stages, sf = emb.profile_stages_read()
        ms1, n1 = emb.profile_read()
        emb.profile_stages(False)
        return {"ms": ms, "serial": serial, "piped": piped, "ms_overlapped": ms_overlapped, "stages": stages, "sf": sf,
                "ms1": ms1 / max(n1, 1)}

    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202, device=device)  # CLS pooling (the BGE family)
    cls = measure(emb)
    # search alone, same 256 queries
    t0 = time.perf_counter()
    for _ in range(iters):
        shard.search_device(d_q[0], B, k)
    torch.cuda.synchronize()
    search_ms = (time.perf_counter() - t0) / iters * 1e3
    # the reference's call shape: 32 chunks per embed_chunks slice
    ids32, mask32 = ids[:32], mask[:32]
    d32 = torch.empty((32, cfg.hidden), dtype=torch.float32, device=dev)
    emb.embed_ids_to_device(ids32, mask32, d32.data_ptr())
    emb.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(10):
        emb.embed_ids_to_device(ids32, mask32, d32.data_ptr())
    wall32 = (time.perf_counter() - t0) / 10
    ms32, n32 = emb.profile_read()
    ms32 /= max(n32, 1)
    # the same call shape through the submission queue: eight slices of 32 submitted, then collected — the library
    # packs them into one 256-row device batch (cs_embedder_submit_ids / cs_embedder_wait)
    def queued_round():
        t0 = time.perf_counter()
        ts = [emb.submit_ids(ids[lo:lo + 32], mask[lo:lo + 32]) for lo in range(0, B, 32)]
        t1 = time.perf_counter()
        r = [emb.wait(ts[0])]
        t2 = time.perf_counter()
        r += [emb.wait(t) for t in ts[1:]]
        if os.environ.get("CS_BENCH_QUEUE_DEBUG"):
            print(f"queued: submit {1e3*(t1-t0):.2f} first wait {1e3*(t2-t1):.2f} rest {1e3*(time.perf_counter()-t2):.2f} ms", file=sys.stderr)
        return r

    queued_round()
    emb.profile_read(reset=True)
    q_reps = 7
    walls = []
    for _ in range(q_reps):
        t0 = time.perf_counter()
        queued_round()
        walls.append(time.perf_counter() - t0)
    # median: the wall clock of a host loop this short takes an occasional tens-of-ms hit from the interpreter (a
    # collection pass over the process's large arrays, seen as one 38 ms submit in seven rounds); the mean is beside it
    wall_q = sorted(walls)[q_reps // 2]
    wall_q_mean = sum(walls) / q_reps
    ms_q, n_q = emb.profile_read()
    gemm_flops, attn_flops = _encoder_flops(cfg, B, L)
    gemm_exec, attn_exec, cls_tail = _encoder_flops_executed(cfg, B, L)
    split, f32n, fb = emb.debug_counters()
    emb.close()
    # BASELINE.json words configs[2] as "mean-pool + L2-norm": the same weights and shape with mask-weighted mean pooling
    # (every one of the 12 layers runs whole) is the headline of this record
    emb_mean = FastEmbedder(ModelType.BGESmallENV15, config=dataclasses.replace(cfg, pooling=POOL_MEAN), seed=202, device=device)
    mean = measure(emb_mean)
    # (iv) the tokenizer's truncation length: 128 x 512 tokens (the same 65,536 token rows; attention's share doubles)
    B5, L5 = 128, 512
    ids5, mask5 = synth_token_batch(cfg, 998, B5, L5, False)
    d5 = torch.empty((B5, cfg.hidden), dtype=torch.float32, device=dev)
    emb_mean.embed_ids_to_device(ids5, mask5, d5.data_ptr())
    emb_mean.profile_read(reset=True)
    for _ in range(iters):
        emb_mean.embed_ids_to_device(ids5, mask5, d5.data_ptr())
    torch.cuda.synchronize()
    ms5, n5 = emb_mean.profile_read()
    ms5 /= max(n5, 1)
    emb_mean.profile_stages(True)
    emb_mean.embed_ids_to_device(ids5, mask5, d5.data_ptr())
    emb_mean.profile_stages_read(reset=True)
    for _ in range(3):
        emb_mean.embed_ids_to_device(ids5, mask5, d5.data_ptr())
    stages5, _ = emb_mean.profile_stages_read()
    emb_mean.profile_stages(False)
    query_embed = query_embed_latency(emb_mean, ids, mask)  # the query side on the BGE-small shape (benchmarks/query_latency.py's first line)
    emb_mean.close()
    g5, a5 = _encoder_flops(cfg, B5, L5)
    quantized_default = quantized_default_model_leg(ids, mask, d_q[0], device, iters)
    nomic = nomic_model_leg(device, iters)
    layers = cfg.layers
    stage_names = ("qkv_gemm", "attention", "out_proj_gemm", "layernorm_attn", "ffn_up_gemm", "ffn_down_gemm", "layernorm_ffn")
    Hh, Ii, BL = cfg.hidden, cfg.intermediate, B * L

    def per_kernel(stages, tail):
        full = layers - 1 if tail else layers   # layers that run whole
        fl = {
            "qkv_gemm": full * 2 * 3 * Hh * Hh * BL + (2 * 2 * Hh * Hh * BL + 2 * Hh * Hh * B if tail else 0),
            "out_proj_gemm": full * 2 * Hh * Hh * BL + (2 * Hh * Hh * B if tail else 0),
            "ffn_up_gemm": full * 2 * Hh * Ii * BL + (2 * Hh * Ii * B if tail else 0),
            "ffn_down_gemm": full * 2 * Hh * Ii * BL + (2 * Hh * Ii * B if tail else 0),
            "attention": full * 4 * L * Hh * BL + (4 * L * Hh * B if tail else 0),
        }
        return ({kn: stages[kn] / layers for kn in stage_names},
                {kn: 3 * f / (stages[kn] * 1e-6) / 1e12 for kn, f in fl.items()})

    def roofline(m, executed, tail):
        sec = m["ms"] * 1e-3
        per_layer, per_tf = per_kernel(m["stages"], tail)
        return {
            "bound": "mfma", "pipe": "mfma_f16", "unit": "TFLOP/s", "peak": MFMA_F16_PEAK_TFLOPS,
            "achieved": executed / sec / 1e12, "frac": executed / sec / 1e12 / MFMA_F16_PEAK_TFLOPS,
            "executed_flops_per_batch": executed, "algorithmic_flops_per_batch": gemm_flops + attn_flops,
            "algorithmic_tflops": (gemm_flops + attn_flops) / sec / 1e12,
            "frac_algorithmic_of_f32_mfma_peak": (gemm_flops + attn_flops) / sec / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "sustained_mfma_stream_tflops": MFMA_F16_16x16x32_SUSTAINED_TFLOPS,
            "frac_of_sustained_mfma_stream": executed / sec / 1e12 / MFMA_F16_16x16x32_SUSTAINED_TFLOPS,
            "sustained_note": "benchmarks/mfma_rate_probe.hip: a loop of independent v_mfma_f32_16x16x32_f16 alone, one wave per "
                              "SIMD on every CU, sustains 1,490 TFLOP/s on pseudo-random operands (2,040 on constant ones): "
                              "the ceiling a kernel with real data can approach on this power-limited part",
            "traffic": None, "cls_tail": tail,
            "note": "achieved = executed f16-MFMA flops (3 per f32 product: hi*hi + the two cross terms) / device time of "
                    "the whole forward (HIP events on the encoder's stream; one stream at this shape, whose tile rounds are whole — "
                    "ragged shapes run as two half-batches on two streams)",
            "per_kernel_us_per_layer": per_layer,
            "per_kernel_us_per_forward": {"embed_ln": m["stages"]["embed_ln"], "pool_normalize": m["stages"]["pool_normalize"]},
            "per_kernel_note": f"one stream, a HIP event after every kernel ({m['sf']} forwards, {m['ms1']:.3f} ms each in that mode)",
            "per_kernel_executed_tflops": per_tf,
        }

    def embed_search(m):
        return {"serial_ms_per_batch": m["serial"] * 1e3, "serial_chunks_per_s": B / m["serial"],
                "pipelined_ms_per_batch": m["piped"] * 1e3, "pipelined_chunks_per_s": B / m["piped"],
                "embed_ms_alone": m["ms"], "search_ms_alone": search_ms,
                "embed_ms_with_previous_search_in_flight": m["ms_overlapped"]}

    enc = {
        "workload": f"BGE-small-en-v1.5 shape (12 x [MHA, GELU FFN, LN], hidden 384), batch {B} x seq {L}, "
                    "synthetic weights, mean-pool + L2 normalise (BASELINE.json configs[2] as worded)",
        "ms_per_batch": mean["ms"], "chunks_per_s": B / (mean["ms"] * 1e-3),
        "algorithmic_tflops": (gemm_flops + attn_flops) / (mean["ms"] * 1e-3) / 1e12,
        "dense_layers": "split-f16 operands on v_mfma_f32_16x16x32_f16, 3 MFMAs per f32 product block",
        "split_forwards": split, "f32_fallbacks": fb,
        "roofline": roofline(mean, 3 * (gemm_flops + attn_flops), False),
        "cls_pool_variant": {
            "ms_per_batch": cls["ms"], "chunks_per_s": B / (cls["ms"] * 1e-3),
            "note": "the pooling fastembed applies to the BGE family (SURVEY.md §0 #4, third-party recollection): the LAST "
                    "layer computes K and V for every token, then ONE query per sequence and B rows through the dense "
                    "layers (csrc/cls_tail.hip) — the same embedding as running the layer whole "
                    "(test_cls_tail_equals_the_full_last_layer); executed flops count what runs",
            "roofline": roofline(cls, 3 * (gemm_exec + attn_exec), cls_tail),
        },
        "encoder_l512": {
            "workload": f"the same encoder (mean pooling) at the tokenizer's truncation length: batch {B5} x seq {L5} "
                        "(src/chunker/semantic.rs:22-29: chunks run to 2,000 characters)",
            "ms_per_batch": ms5, "chunks_per_s": B5 / (ms5 * 1e-3), "tokens_per_s": B5 * L5 / (ms5 * 1e-3),
            "executed_tflops": 3 * (g5 + a5) / (ms5 * 1e-3) / 1e12,
            "frac_of_f16_mfma_peak": 3 * (g5 + a5) / (ms5 * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
            "per_kernel_us_per_layer": {kn: stages5[kn] / layers for kn in stage_names},
            "attention_executed_tflops": 3 * layers * 4 * L5 * Hh * B5 * L5 / (stages5["attention"] * 1e-6) / 1e12,
        },
        "quantized_default_model": quantized_default,
        "query_embed": query_embed,
        "nomic_model": nomic,
        "reference_call_shape": {
            "workload": "32 chunks x 256 tokens per call (BatchEmbedder slices by 32, src/embed/batch.rs:70,94), CLS pooling",
            "device_ms_per_call": ms32, "wall_ms_per_call_incl_h2d": wall32 * 1e3, "chunks_per_s": 32 / (ms32 * 1e-3),
            "queued": {
                "workload": "eight such calls submitted (cs_embedder_submit_ids), then collected (cs_embedder_wait, host "
                            "buffers): the library embeds the 256 queued rows as one device batch",
                "wall_ms_per_8_calls": wall_q * 1e3, "chunks_per_s": B / wall_q, "wall_is": f"median of {q_reps} rounds",
                "wall_ms_per_8_calls_mean": wall_q_mean * 1e3,
                "device_ms_per_8_calls": ms_q / max(q_reps, 1), "device_batches_per_8_calls": n_q / max(q_reps, 1),
            },
        },
    }
    if with_cpu:
        enc["cpu_baseline"] = encoder_cpu_baseline(cfg, 202)
    es = embed_search(mean)
    es["workload"] = (f"embed {B} chunks (seq {L}, mean pooling: BASELINE.json's wording) on the GPU, then one batched "
                      f"top-{k} search of the {B} embeddings over the resident corpus")
    best = "pipelined" if es["pipelined_chunks_per_s"] >= es["serial_chunks_per_s"] else "serial"
    es["ms_per_batch"] = es[best + "_ms_per_batch"]
    es["chunks_embedded_and_searched_per_s"] = es[best + "_chunks_per_s"]
    es["headline_loop"] = best
    es["pipelined_note"] = ("search of batch i on a second stream under the forward of batch i + 1 (two query buffers, no "
                            "host wait until the end); serial = embed, search, wait per batch.  Both kernels want every "
                            "CU (persistent one-block-per-CU grids): the overlap is worth what the forward's tails leave "
                            "free, and the forward itself runs slower beside a search (embed_ms_with_previous_search_in_flight)")
    es["cls_pool_variant"] = embed_search(cls)
    return {"encoder": enc, "embed_search": es}


def nomic_model_leg(device, iters):
    """The registry's Nomic entries (ModelType::NomicEmbedTextV1 / V15 / V15Q, /root/reference/src/embed/embedder.rs:30-35) are
    NomicBert encoders: 12 x 768, 12 heads of 64, rotary positions on Q / K, fc2(fc11(x) * silu(fc12(x))) with n_inner 3072
    (cs_bert_config.arch = CS_ARCH_NOMIC, csrc/nomic.hip).  The reference's mini-batch for a 768-d model (128 sequences,
    embedder.rs:251-261) at 256 tokens, synthetic weights, mean pooling; flops = 16 H^2 multiply-adds per token and layer in
    the dense layers (3 + 1 + 8 + 4) against BERT's 12."""
    import torch

    from codesearch_amd import FastEmbedder, ModelType
    from codesearch_amd.bert_params import synth_token_batch

    mt = ModelType.NomicEmbedTextV15
    cfg = mt.bert_config()
    Bn, Ln = 128, 256
    ids, mask = synth_token_batch(cfg, 997, Bn, Ln, False)
    emb = FastEmbedder(mt, config=cfg, seed=203, device=device)
    d_out = torch.empty((Bn, cfg.hidden), dtype=torch.float32, device=f"cuda:{device}")
    emb.embed_ids_to_device(ids, mask, d_out.data_ptr())
    torch.cuda.synchronize()
    emb.profile_read(reset=True)
    for _ in range(iters):
        emb.embed_ids_to_device(ids, mask, d_out.data_ptr())
    torch.cuda.synchronize()
    ms, n = emb.profile_read()
    ms /= max(n, 1)
    emb.profile_stages(True)
    emb.embed_ids_to_device(ids, mask, d_out.data_ptr())
    emb.profile_stages_read(reset=True)
    for _ in range(3):
        emb.embed_ids_to_device(ids, mask, d_out.data_ptr())
    stages, _ = emb.profile_stages_read()
    emb.profile_stages(False)
    split_fw, f32_fw, _ = emb.debug_counters()
    emb.close()
    H, I, T = cfg.hidden, cfg.intermediate, Bn * Ln
    dense = 2.0 * T * (3 * H * H + H * H + 2 * I * H + H * I) * cfg.layers
    attn = 4.0 * Bn * Ln * Ln * H * cfg.layers
    stage_names = ("qkv_gemm", "attention", "out_proj_gemm", "layernorm_attn", "ffn_up_gemm", "ffn_down_gemm", "layernorm_ffn")
    return {
        "workload": f"{mt.name_str()} shape ({cfg.layers} x hidden {H}, {cfg.heads} heads of {H // cfg.heads}, n_inner {I}, rotary base "
                    f"{cfg.rotary_base:g}), batch {Bn} x seq {Ln}, synthetic weights, mean pooling",
        "ms_per_batch": ms, "chunks_per_s": Bn / (ms * 1e-3), "tokens_per_s": T / (ms * 1e-3),
        "algorithmic_tflops": (dense + attn) / (ms * 1e-3) / 1e12,
        "executed_tflops": 3 * (dense + attn) / (ms * 1e-3) / 1e12,
        "frac_of_f16_mfma_peak": 3 * (dense + attn) / (ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
        "split_forwards": split_fw, "f32_fallbacks": f32_fw,
        "per_kernel_us_per_layer": {kn: stages[kn] / cfg.layers for kn in stage_names},
        "per_kernel_note": "qkv_gemm includes the rotary map on Q and K (nomic.hip rope_split_kernel); ffn_up_gemm is ONE product over "
                           "fc11 | fc12 whose epilogue stores value * silu(gate) (gemm_wide.hip GW_OUT_SWIGLU)",
    }


def query_embed_latency(emb, ids, mask, reps=60):
    """One short query (1 x 16 tokens) through `emb`, and a query with its eight variants (9 x 16: what the reference embeds per search,
    src/search/mod.rs:508-611): device time of the forward (HIP events) and wall time of the host call."""
    out = {}
    for B, key in ((1, ""), (9, "variants_")):
        q_ids, q_mask = ids[:B, :16].copy(), mask[:B, :16].copy()
        q_mask[:] = 1
        for _ in range(10):
            emb.embed_ids(q_ids, q_mask)
        emb.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            emb.embed_ids(q_ids, q_mask)
        wall = (time.perf_counter() - t0) / reps
        ms, n = emb.profile_read()
        emb.profile_read(reset=True)
        out[key + "tokens"] = f"{B} x 16"
        out[key + "device_us"] = ms / max(n, 1) * 1e3
        out[key + "host_call_us"] = wall * 1e6
    return out


def quantized_default_model_leg(ids, mask, d_out, device, iters):
    """The reference's DEFAULT model is a dynamically quantised one (ModelType::AllMiniLML6V2Q,
    /root/reference/src/embed/embedder.rs:12-13): 6 layers, hidden 384, Linear weights as onnxruntime's quantize_dynamic
    stores them, activations re-quantised to 8 bits per call.  Same batch (256 x 256 tokens, synthetic weights quantised
    per tensor to uint8): the dynamic-quantisation mode (int8 MFMA, csrc/gemm_q8.hip) beside the f32 graph of the same
    weights (split-f16 kernels)."""
    import torch

    from codesearch_amd import FastEmbedder, ModelType
    from codesearch_amd.bert_params import quantize_linear_weights, synth_params

    mt = ModelType.AllMiniLML6V2Q
    cfg = mt.bert_config()
    params, wscale = quantize_linear_weights(cfg, synth_params(cfg, 202), per_channel=False, unsigned=True)
    emb = FastEmbedder(mt, config=cfg, params=params, wscale=wscale, device=device)
    stage_names = ("qkv_gemm", "attention", "out_proj_gemm", "layernorm_attn", "ffn_up_gemm", "ffn_down_gemm", "layernorm_ffn")
    out = {"workload": f"{mt.name_str()} shape ({cfg.layers} x hidden {cfg.hidden}), batch {ids.shape[0]} x seq {ids.shape[1]}, "
                       "mean pooling, uint8 per-tensor Linear weights"}
    for mode, key in (("q8", "dynamic_quantisation"), ("split", "f32_graph_of_the_quantised_weights")):
        emb.set_gemm_mode(mode)
        emb.embed_ids_to_device(ids, mask, d_out.data_ptr())
        torch.cuda.synchronize()
        emb.profile_read(reset=True)
        for _ in range(iters):
            emb.embed_ids_to_device(ids, mask, d_out.data_ptr())
        torch.cuda.synchronize()
        ms, n = emb.profile_read()
        ms /= max(n, 1)
        emb.profile_stages(True)
        emb.embed_ids_to_device(ids, mask, d_out.data_ptr())
        emb.profile_stages_read(reset=True)
        for _ in range(3):
            emb.embed_ids_to_device(ids, mask, d_out.data_ptr())
        stages, _ = emb.profile_stages_read()
        emb.profile_stages(False)
        out[key] = {"ms_per_batch": ms, "chunks_per_s": ids.shape[0] / (ms * 1e-3),
                    "per_kernel_us_per_layer": {kn: stages[kn] / cfg.layers for kn in stage_names}}
    out["per_kernel_note"] = ("dynamic_quantisation: each Linear's time includes the range reduction (and, below 4,096 rows, the "
                              "quantising pass) over its input; ffn_up_gemm is the two-pass product that leaves already "
                              "re-quantised for ffn_down")
    # the reference's call shape on its default model: 32 chunks per embed call (src/embed/batch.rs:70,94), one call at a
    # time and eight of them through the submission queue — each stays its own quantisation unit inside the shared batch
    emb.set_gemm_mode("q8")
    out["query_embed"] = query_embed_latency(emb, ids, mask)  # `codesearch search`: one short query (src/embed/mod.rs:164-181)
    B = ids.shape[0]
    emb.embed_ids(ids[:32], mask[:32])
    t0 = time.perf_counter()
    for _ in range(5):
        for lo in range(0, B, 32):
            emb.embed_ids(ids[lo:lo + 32], mask[lo:lo + 32])
    one = (time.perf_counter() - t0) / 5

    def queued_round():
        ts = [emb.submit_ids(ids[lo:lo + 32], mask[lo:lo + 32]) for lo in range(0, B, 32)]
        return [emb.wait(t) for t in ts]

    queued_round()
    emb.profile_read(reset=True)
    walls = []
    for _ in range(7):
        t0 = time.perf_counter()
        queued_round()
        walls.append(time.perf_counter() - t0)
    ms_q, n_q = emb.profile_read()
    out["reference_call_shape"] = {
        "workload": f"{B // 32} calls of 32 chunks x {ids.shape[1]} tokens, dynamic quantisation",
        "one_call_at_a_time_chunks_per_s": B / one,
        "queued_chunks_per_s": B / sorted(walls)[3], "queued_wall_is": "median of 7 rounds",
        "queued_device_ms_per_round": ms_q / 7, "queued_device_batches_per_round": n_q / 7,
        "note": "queued: one device batch, one quantisation unit per call (its own range per tensor; rows beyond a call's own "
                "padded length stay out of it)",
    }
    emb.close()
    return out


def e2e_leg(chunks, k, device):
    """BASELINE.json configs[3]: `chunks` synthetic code chunks (token ids [chunks, 256], BGE-small shape)
    embedded on the GPU, appended to a device-resident index without leaving HBM, then 64 batched queries
    (lightly edited copies of known chunks, so the right answer is known) top-k."""
    import numpy as np

    from codesearch_amd import BertConfig, FastEmbedder, ModelType, VectorStore
    from codesearch_amd.bert_params import synth_token_batch
    from codesearch_amd.pipeline import index_token_chunks, search_token_queries

    cfg = BertConfig.bge_small()
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202, device=device)
    seq, nq = 256, 64
    ids, mask = synth_token_batch(cfg, 31337, chunks, seq, False)
    targets = [(i * 7919) % chunks for i in range(nq)]
    q_ids, q_mask = ids[targets].copy(), mask[targets].copy()
    q_ids[:, 5] = (q_ids[:, 5] + 1) % cfg.vocab_size
    emb.embed_ids(ids[:256], mask[:256])  # warm-up: allocate workspace
    emb.profile_read(reset=True)
    store = VectorStore(None, cfg.hidden, device=device, capacity=chunks)
    t0 = time.perf_counter()
    t_index = index_token_chunks(emb, store, ids, mask)
    cos, rid, counts, t_search = search_token_queries(emb, store, q_ids, q_mask, k)
    wall = time.perf_counter() - t0
    hit = float(np.mean([rid[i][0] == targets[i] for i in range(nq)]))
    fwd_ms, fwd_n = emb.profile_read()
    emb.close()
    store.close()
    return {
        "workload": f"index {chunks} chunks x {seq} tokens (BGE-small shape, fp32) on the GPU + {nq} batched queries "
                    f"top-{k} (BASELINE.json configs[3])",
        "chunks_per_s_end_to_end": chunks / wall, "wall_s": wall,
        "embed_s": t_index["embed_s"], "insert_build_s": t_index["insert_build_s"],
        "embed_queries_s": t_search["embed_queries_s"], "search_s": t_search["search_s"],
        "encoder_device_ms_per_batch": fwd_ms / max(fwd_n, 1), "encoder_batches": fwd_n,
        "top1_is_edited_source_chunk": hit, "mean_top1_cos": float(cos[:, 0].mean()),
    }


def scan_1m_leg(dim, k, device, store_cls):
    """BASELINE.json configs[1] for information: 1 query over 1,000,000 x 384 fp32 rows (1.536 GB per launch:
    launch ramp and tail weigh ~10x more than at 10M rows).  Kernel time from the library's HIP events."""
    import ctypes as C

    import torch

    from codesearch_amd import _lib
    from codesearch_amd.synth import synth_rows

    rows = 1_000_000
    st = store_cls(None, dim, device=device, capacity=rows)
    st.insert_synthetic(rows, SEED, 0)
    st.build_index()
    st.set_single_query_route(st.ROUTE_STREAM)  # configs[1] names the scan kernel; the default route is timed below
    lib = _lib.load()
    dev = f"cuda:{device}"
    d_q = torch.from_numpy(synth_rows(SEED + 1, 0, 1, dim)).to(dev)
    keys = torch.zeros(k, dtype=torch.int64, device=dev)
    cos = torch.zeros(k, dtype=torch.float32, device=dev)
    ids = torch.zeros(k, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda x: C.c_void_p(x.data_ptr())

    def search():
        _lib.check(lib.cs_index_search_device(st.handle, vp(d_q), 1, dim, k, vp(keys), vp(cos), vp(ids), vp(cnt), stream))

    for _ in range(20):
        search()
    torch.cuda.synchronize()
    st.profile(True)
    st.profile_read(reset=True)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        search()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    scan_ms, launches, _ = st.profile_read(reset=True)
    st.profile(False)
    us = scan_ms * 1e3 / max(launches, 1)
    gbps = rows * dim * 4 / (us * 1e-6) / 1e9
    ref_keys = keys.clone()
    st.set_single_query_route(st.ROUTE_COST)
    for _ in range(20):
        search()
    torch.cuda.synchronize()
    same = bool(torch.equal(keys, ref_keys))
    t0 = time.perf_counter()
    for _ in range(reps):
        search()
    torch.cuda.synchronize()
    wall_def = (time.perf_counter() - t0) / reps
    out = {"workload": f"brute-force cosine top-{k}, 1 query over {rows} x {dim} fp32 rows (BASELINE.json configs[1])",
           "ms_per_search": wall * 1e3, "chunks_per_s": rows / wall, "scan_kernel_us": us,
           "hbm_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
           "default_routing": {"ms_per_search": wall_def * 1e3, "chunks_per_s": rows / wall_def,
                               "bit_identical_to_streaming_scan": same,
                               "note": "CS_ROUTE_COST: int8 filter + exact refine from 32,768 rows on (k < 48; 300,000 from k = 48); the figures above "
                                       "select CS_ROUTE_STREAM (the HIP scan kernel configs[1] names)"}}
    del st
    return out


def hbm_reference(device, nbytes=2 << 30, reps=10):
    """measured device bandwidths beside the 8 TB/s spec figure.  read_probe_GBps is this repo's own read-only
    probe (benchmarks/hbm_read_probe.hip: the scan's access shape — 16 B per lane, non-temporal, 12-KiB tiles per wave — and
    none of its arithmetic; built by __graft_entry__.build(), run as a child process over 15.36 GB), the number to read the
    scan against; copy / fill are torch's copy_ and zero_ kernels, for orientation only."""
    import torch

    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    out = {}
    for name, fn, moved in (("copy", lambda: dst.copy_(src), 2 * nbytes), ("fill", lambda: dst.zero_(), nbytes)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_GBps"] = moved / (e0.elapsed_time(e1) * 1e-3 / reps) / 1e9
    del src, dst
    torch.cuda.empty_cache()
    probe = os.path.join(ROOT, "benchmarks", "hbm_read_probe")
    out["read_probe_GBps"] = None
    if os.path.exists(probe):
        try:
            r = subprocess.run([probe, "14.3"], capture_output=True, text=True, timeout=120,
                               env=dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(torch.device(device).index or 0))))
            rates = [float(x) for x in re.findall(r"nt ([0-9.]+) TB/s", r.stdout)]
            if rates:
                out["read_probe_GBps"] = max(rates) * 1e3
                out["read_probe"] = "benchmarks/hbm_read_probe 14.3: best of its (loads per lane x blocks per CU) grid, non-temporal 16-B loads"
        except Exception as e:  # the probe is orientation, never a reason to lose the record
            out["read_probe_error"] = repr(e)[:200]
    out["note"] = (f"copy / fill: torch copy_ (read + write bytes counted) and zero_ over {nbytes >> 20} MiB; "
                   "MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy")
    return out


def host_cpu_share(visible):
    """Threads the CPU baseline may really use: the cgroup CPU quota of this container (a GPU box hands a
    one-GPU job a share of its host cores) and the affinity mask, not the number of cores the host shows."""
    import math

    n = visible
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, math.ceil(quota / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(oracle, sample_rows, dim, k, store_cls, timed_route="stream"):
    """Time the oracle's tuned CPU port (and the literal scalar loop on a smaller slice) on
    a bounded sample of the same workload; also returns recall@k of the HIP path against
    the CPU result on that sample (BASELINE.json configs[1]: 1 query over 1M x 384)."""
    import numpy as np

    from codesearch_amd.synth import synth_rows

    corpus = oracle.synth_rows(SEED, 0, sample_rows, dim)
    q = synth_rows(SEED + 1, 0, 1, dim)[0]
    threads = host_cpu_share(oracle.num_threads())
    oracle.scan_topk(corpus[: min(sample_rows, 50_000)], q, k, mode="omp", threads=threads)  # warm threads
    reps, t_total, res = 0, 0.0, None
    while t_total < 8.0 and reps < 64:
        t0 = time.perf_counter()
        res = oracle.scan_topk(corpus, q, k, mode="omp", threads=threads)
        t_total += time.perf_counter() - t0
        reps += 1
    omp_rate = sample_rows * reps / t_total
    lit_rows = min(sample_rows, 200_000)
    t0 = time.perf_counter()
    oracle.scan_topk(corpus[:lit_rows], q, k, mode="literal")
    lit_rate = lit_rows / (time.perf_counter() - t0)
    # recall@k of the HIP path vs the CPU result on the same sample — on the TIMED route (the f32 streaming scan, the kernel
    # `value` and `roofline` are quoted on) and on the library's default route (int8 filter + exact refine), each named
    st = store_cls(None, dim, device=0)
    st.insert_synthetic(sample_rows, SEED, 0)
    st.build_index()
    checks = {}
    for name, route in (("stream", st.ROUTE_STREAM), ("cost", st.ROUTE_COST)):
        st.set_single_query_route(route)
        b0, _ = st.debug_counters()
        cos, ids, _ = st.search_raw(q, k)
        b1, _ = st.debug_counters()
        checks[name] = {"route": {"stream": "CS_ROUTE_STREAM: cs::scan_topk_kernel (the timed kernel)",
                                  "cost": "CS_ROUTE_COST (default): int8 filter + exact f32 refine"}[name],
                        "took_filter_path": b1 > b0,
                        "recall_at_k": len(set(ids[0].tolist()) & set(res[1].tolist())) / float(k),
                        "ids_equal_cpu_in_order": ids[0].tolist() == res[1].tolist(),
                        "max_abs_cos_err_vs_cpu": float(np.abs(cos[0] - res[0]).max()), "rows": sample_rows}
    st.close()
    recall = checks[timed_route]["recall_at_k"]
    max_err = checks[timed_route]["max_abs_cos_err_vs_cpu"]
    return {
        "value": omp_rate, "unit": "chunks/s", "cores": threads, "kind": "port",
        "sample": f"{sample_rows} x {dim} fp32 rows of the same synthetic corpus, 1 query, top-{k}, "
                  f"{reps} passes of oracle/scan_oracle.c cs_oracle_scan_topk_omp ({threads} threads = this job's CPU "
                  f"share; the host shows {oracle.num_threads()})",
        "literal_1thread_chunks_per_s": lit_rate,
        "literal_sample_rows": lit_rows,
    }, recall, max_err, checks


def rccl_child_record(args, nproc, force_dist):
    """The one-rank-per-GPU form of this benchmark (torch.distributed.run, RCCL broadcast + all-gather:
    codesearch_amd/sharded.py) run as CHILD processes, BEFORE this process makes its first GPU call (a process that has
    touched the GPU is never replaced or forked from).  Returns the child's line reduced to what answers "did RCCL see
    N ranks, and what did the exchange cost": world size, value, ms per step, the per-shard checks."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.pop("CS_BENCH_SHARD_DEVICES", None)
    if force_dist:
        env["CS_BENCH_FORCE_DIST"] = "1"  # a rehearsal on fewer GPUs than shards: the exchange at world 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", str(min(args.steps, 50)), "--warmup", str(min(args.warmup, 10)),
           "--rows", str(args.rows), "--dim", str(args.dim), "--nq", str(args.nq), "--k", str(args.k), "--only-scan"]
    t0 = time.perf_counter()
    budget = float(args.rccl_child_timeout)
    try:
        # own session: a child that hangs in RCCL initialisation is ended as a whole process GROUP (the launcher and every
        # rank), by the ids this call created — never by name
        import signal

        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT,
                             start_new_session=True)
        try:
            out_s, err_s = p.communicate(timeout=budget)
        except subprocess.TimeoutExpired:
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(p.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    p.communicate(timeout=10)
                    break
                except subprocess.TimeoutExpired:
                    continue
            return {"error": f"the RCCL child processes did not finish within {budget:.0f} s and were ended (process group "
                             f"{p.pid}); the one-process line below does not depend on them",
                    "wall_s": time.perf_counter() - t0}

        class _R:  # the fields of subprocess.CompletedProcess used below
            stdout, stderr, returncode = out_s, err_s, p.returncode
        r = _R
    except Exception as e:  # a missing launcher: recorded, never fatal for the one-process line
        return {"error": f"{type(e).__name__}: {e}"}
    rec = None
    for ln in reversed(r.stdout.splitlines()):
        if ln.startswith("{"):
            try:
                rec = json.loads(ln)
                break
            except ValueError:
                continue
    if rec is None:
        return {"error": f"child exited {r.returncode} without a JSON line", "stderr_tail": r.stderr[-600:]}
    return {"launch": f"torch.distributed.run --nproc-per-node {nproc} bench.py --only-scan (child processes)",
            "collective": rec.get("collective"), "rccl_world_size": rec.get("rccl_world_size"),
            "n_gpus": rec.get("n_gpus"), "value": rec.get("value"), "ms_per_step": rec.get("ms_per_step"),
            "steps": rec.get("steps"), "multi_gpu_checks": rec.get("multi_gpu_checks"),
            "roofline_frac": (rec.get("roofline") or {}).get("frac"), "error": rec.get("error"),
            "wall_s_incl_fill_and_build": time.perf_counter() - t0}


def main_one_process(args):
    """`python bench.py --gpus N` without a launcher: this process owns all N GPUs through the C ABI's row-sharded
    store (cs_shards_*, csrc/shards.hip) — the form the reference's single-process `VectorStore::search`
    (src/vectordb/store.rs:431-486, called at src/search/mod.rs:508-511) maps to.  Weak scaling: shard g holds rows
    [g * rows, (g + 1) * rows).  A step = cs_shards_search_device: queries in HBM of GPU 0, fetched by every shard
    over xGMI, per-shard scan + top-k, nq*k*8 bytes per shard sent back, one merge on GPU 0; nothing waits for the
    host inside the timed region.  No collective library is involved in this form (peer copies / peer stores); the
    RCCL form runs first, as child processes, and is reported under `rccl`."""
    N = args.gpus
    rehearsal = bool(os.environ.get("CS_BENCH_SHARD_DEVICES"))
    devices = [int(x) for x in os.environ["CS_BENCH_SHARD_DEVICES"].split(",")] if rehearsal else list(range(N))
    rccl = None
    phases = {}  # wall seconds per phase of this run, carried in the line
    t_ph = time.perf_counter()

    def phase(name):
        nonlocal t_ph
        now = time.perf_counter()
        phases[name] = round(now - t_ph, 3)
        t_ph = now

    if not args.no_rccl_child:
        distinct = len(set(devices))
        rccl = rccl_child_record(args, distinct, force_dist=(distinct == 1))
    phase("rccl_child_s")

    import numpy as np
    import torch

    from codesearch_amd import VectorStore, _lib
    from codesearch_amd.synth import synth_planted, synth_rows

    lib = _lib.load()
    ndev = int(lib.cs_device_count())
    # CS_BENCH_SHARD_DEVICES=0,0: rehearse the N-shard code path on a box with fewer GPUs (not a scaling measurement)
    if len(devices) != N or max(devices) >= ndev:
        raise SystemExit(f"--gpus {N}: only {ndev} HIP device(s) visible")
    dim, rows, nq, k = args.dim, args.rows, args.nq, args.k
    phase("import_s")
    st = VectorStore(None, dim, devices=devices, rows_per_stripe=rows, capacity=N * rows)
    st.insert_synthetic(N * rows, SEED, 0)
    for d in sorted(set(devices)):
        torch.cuda.synchronize(d)
    phase("fill_s")
    st.build_index()
    for d in sorted(set(devices)):
        torch.cuda.synchronize(d)
    phase("build_s")
    assert st.shard_lens() == [rows] * N
    # `value` is quoted on the f32 streaming scan (the north-star kernel) when a step is one query; the default route of
    # such a search (int8 filter + exact refine, same bits) is timed beside it
    if nq == 1:
        st.set_single_query_route(st.ROUTE_STREAM)
    root = st.root_device()
    torch.cuda.set_device(root)
    dev = f"cuda:{root}"
    q_host = synth_rows(SEED + 1, 0, nq, dim)
    d_q = torch.from_numpy(q_host).to(dev)
    keys = torch.zeros(nq * k, dtype=torch.int64, device=dev)
    cos = torch.zeros(nq * k, dtype=torch.float32, device=dev)
    ids = torch.zeros(nq * k, dtype=torch.int32, device=dev)
    cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(q=d_q, m=nq, kk=keys, cc=cos, ii=ids, nn=cnt):
        st.search_device(q.data_ptr(), m, k, kk.data_ptr(), cc.data_ptr(), ii.data_ptr(), nn.data_ptr(), stream)

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    def timed(fn, steps):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync_all()
        return (time.perf_counter() - t0) / steps

    # (1) every shard must answer for its own rows: one planted query per shard (a noisy copy of a row living there)
    planted_rows = [g * rows + (4242 + 1013 * g) % rows for g in range(N)]
    planted = synth_planted(SEED, SEED + 7, planted_rows, dim)
    planted_ok = []
    for g in range(N):
        pq = torch.from_numpy(planted[g:g + 1].copy()).to(dev)
        step(pq, 1)
        sync_all()
        planted_ok.append(int(ids[0].item()) & 0xFFFFFFFF == planted_rows[g])

    # (2) the merged answer of the timed queries against a HOST merge of what each shard's own index returns
    # (cs_index_search on the shard handles; local row -> global id as merge_topk_kernel remaps it; order: cosine
    # descending, id ascending): ids and cosine bits must be equal
    def host_merge(qh, m):
        per_cos, per_ids = [], []
        for g in range(N):
            c = np.zeros((m, k), np.float32)
            i = np.zeros((m, k), np.uint32)
            n = np.zeros(m, np.uint32)
            _lib.check(lib.cs_index_search(st.shard_handle(g), qh.ctypes.data_as(_lib.f32p), m, dim, k,
                                           c.ctypes.data_as(_lib.f32p), i.ctypes.data_as(_lib.u32p), n.ctypes.data_as(_lib.u32p)))
            gid = ((i.astype(np.uint64) // rows) * N + g) * rows + i.astype(np.uint64) % rows
            for r in range(m):
                c[r, n[r]:] = -np.inf
            per_cos.append(c)
            per_ids.append(gid)
        ac, ai = np.concatenate(per_cos, axis=1), np.concatenate(per_ids, axis=1)
        oc, oi = np.zeros((m, k), np.float32), np.zeros((m, k), np.uint64)
        for r in range(m):
            order = np.lexsort((ai[r], -ac[r].astype(np.float64)))[:k]
            oc[r], oi[r] = ac[r][order], ai[r][order]
        return oc, oi

    step()
    sync_all()
    ref_keys = keys.clone()
    got_ids = ids.cpu().numpy().astype(np.uint32).reshape(nq, k)
    got_cos = cos.cpu().numpy().reshape(nq, k)
    exp_cos, exp_ids = host_merge(q_host, nq)
    merged_ok = bool(np.array_equal(got_ids.astype(np.uint64), exp_ids) and got_cos.tobytes() == exp_cos.tobytes())

    phase("checks_s")
    for _ in range(args.warmup):
        step()
    sync_all()
    sh0 = st.shard_handle(0)
    _lib.check(lib.cs_index_profile(sh0, 1))
    s_ms, m_ms, n_l = C.c_double(), C.c_double(), C.c_uint64()
    _lib.check(lib.cs_index_profile_read(sh0, C.byref(s_ms), C.byref(n_l), C.byref(m_ms), 1))
    elapsed = timed(step, args.steps) * args.steps
    _lib.check(lib.cs_index_profile_read(sh0, C.byref(s_ms), C.byref(n_l), C.byref(m_ms), 1))
    _lib.check(lib.cs_index_profile(sh0, 0))
    timed_same = bool(torch.equal(keys, ref_keys))
    phase("warmup_and_timed_s")
    scan_us = s_ms.value * 1e3 / max(n_l.value, 1)
    alg_bytes = rows * dim * 4
    achieved = alg_bytes / (scan_us * 1e-6) / 1e9 if scan_us else 0.0
    total_rows = rows * N
    direct = bool(lib.cs_shards_direct_gather(st.handle))
    line = {
        "metric": "chunks embedded+searched/sec over 10M\u00d7384 corpus; recall@10 vs CPU ref",
        "value_is": f"chunks searched/sec: brute-force cosine top-{k}, {nq} query/step over {rows} x {dim} fp32 rows "
                    f"per GPU, {N} GPUs driven by one process",
        "value": total_rows * nq * args.steps / elapsed,
        "unit": "chunks/s", "n_gpus": len(set(devices)), "shards": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"brute-force cosine top-{k}, {nq} query/step over {rows} x {dim} fp32 rows per GPU "
                        f"(BASELINE.json north-star target / configs[4] layout; rows generated in HBM by "
                        f"include/cs_synth.h, seed {SEED:#x})",
            "rows_per_gpu": rows, "dim": dim, "queries_per_step": nq, "k": k,
            "parallelism": f"one process, cs_shards over {N} GPUs: queries fetched from GPU {root} over xGMI, per-shard "
                           f"scan + top-k, {nq * k * 8} B per shard gathered on GPU {root} "
                           f"({'written by the search kernel' if direct else 'one peer copy per shard'}), "
                           "one merge; no host synchronisation inside the timed region",
        },
        "collective": "none — peer copies over xGMI (hipMemcpyPeerAsync / peer stores); RCCL runs in the `rccl` record",
        "roofline": {
            "kernel": "cs::scan_topk_kernel (shard 0's launches; every shard runs the same kernel on its own rows)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_us": scan_us, "launches_timed": int(n_l.value),
            "merge_avg_us": m_ms.value * 1e3 / max(n_l.value, 1),
            "note": "per GPU: HIP-event span of the prime pass + scan kernel on shard 0's stream; bytes = that shard's rows",
        },
        "top1": {"id": int(ids[0].item()) & 0xFFFFFFFF, "cos": float(cos[0].item())},
        "multi_gpu_checks": {
            "planted_query_per_shard_returns_its_row": planted_ok,
            "merged_topk_equals_host_merge_of_per_shard_searches": merged_ok,
            "timed_searches_equal_the_first": timed_same,
        },
        "rccl": rccl,
        "phases_wall_s": phases,
    }
    failed = []
    if not all(planted_ok):
        failed.append("a shard did not return its planted row")
    if not merged_ok:
        failed.append("the merged top-k differs from the host merge of the per-shard searches")
    if not timed_same:
        failed.append("timed searches disagree with the first one")
    if failed:
        line["error"] = "; ".join(failed)
    # The record is on stdout BEFORE any optional leg runs (the opt-in gather mode, the default route, configs[4]'s
    # 1,000-query step): a leg that hangs or is ended from outside cannot take the measured line with it.  The complete
    # line — the same keys plus the legs' — follows as the LAST line when they finish.
    pending = ["opt_in_gather_mode"] + (["default_routing"] if nq == 1 else []) + ([] if args.no_config5 else ["config_5"])
    print(json.dumps(dict(line, legs_pending=pending)), flush=True)
    # the gather form NOT in use, for the record (opt-in CS_SHARDS_DIRECT): a disagreement is reported below, it does
    # not fail the default path's line
    other = "0" if bool(lib.cs_shards_direct_gather(st.handle)) else "1"
    os.environ["CS_SHARDS_DIRECT"], prev = other, os.environ.get("CS_SHARDS_DIRECT")
    alt_same, alt_direct, alt_error = None, None, None
    try:
        st2 = VectorStore(None, dim, devices=devices, rows_per_stripe=1 << 16, capacity=N << 16)
        st2.insert_synthetic(N << 16, SEED, 0)
        st2.build_index()
        alt_direct = bool(lib.cs_shards_direct_gather(st2.handle))
        c_alt, i_alt, _ = st2.search_raw(q_host, k)
        st2.close()
    except Exception as e:
        alt_error = f"{type(e).__name__}: {e}"
    finally:
        if prev is None:
            del os.environ["CS_SHARDS_DIRECT"]
        else:
            os.environ["CS_SHARDS_DIRECT"] = prev
    if alt_error is None:
        st3 = VectorStore(None, dim, devices=devices, rows_per_stripe=1 << 16, capacity=N << 16)
        st3.insert_synthetic(N << 16, SEED, 0)
        st3.build_index()
        c_def, i_def, _ = st3.search_raw(q_host, k)
        alt_same = bool(np.array_equal(i_alt, i_def) and c_alt.tobytes() == c_def.tobytes())
        st3.close()

    line["multi_gpu_checks"].update({"opt_in_gather_mode_agrees": alt_same,
                                     "opt_in_gather_mode": "direct" if alt_direct else "copy",
                                     "opt_in_gather_mode_error": alt_error})
    phase("opt_in_gather_s")
    # the default route of the same search (CS_ROUTE_COST): one query per shard through the int8 filter + exact refine
    if nq == 1:
        st.set_single_query_route(st.ROUTE_COST)
        step()
        sync_all()
        same_bits = bool(torch.equal(keys, ref_keys))
        ms_def = timed(step, max(20, args.steps // 4)) * 1e3
        line["default_routing"] = {"ms_per_step": ms_def, "chunks_per_s": total_rows / (ms_def * 1e-3),
                                   "bit_identical_to_streaming_scan": same_bits,
                                   "note": "cs_index_set_single_query_route(CS_ROUTE_COST): what a caller gets without asking; "
                                           "`value` selects CS_ROUTE_STREAM"}
        st.set_single_query_route(st.ROUTE_STREAM)
        phase("default_routing_s")
    # BASELINE.json configs[4] as worded: 1,000 batched queries per step over the resident shards (int8 MFMA filter +
    # exact f32 refine per shard, one merge of N lists per query)
    if not args.no_config5:
        q5 = 1000
        q5_host = synth_rows(SEED + 5, 0, q5, dim)
        d_q5 = torch.from_numpy(q5_host).to(dev)
        k5, c5 = torch.zeros(q5 * k, dtype=torch.int64, device=dev), torch.zeros(q5 * k, dtype=torch.float32, device=dev)
        i5, n5 = torch.zeros(q5 * k, dtype=torch.int32, device=dev), torch.zeros(q5, dtype=torch.int32, device=dev)
        f5 = lambda: step(d_q5, q5, k5, c5, i5, n5)
        f5()
        sync_all()
        sample = [0, 333, 999]
        e_cos, e_ids = host_merge(q5_host[sample], len(sample))
        g_ids = i5.cpu().numpy().astype(np.uint32).reshape(q5, k)[sample].astype(np.uint64)
        g_cos = c5.cpu().numpy().reshape(q5, k)[sample]
        ok5 = bool(np.array_equal(g_ids, e_ids) and g_cos.tobytes() == e_cos.tobytes())
        overflow5 = st.search_status(stream)
        ms5 = timed(f5, 10) * 1e3
        per, tiles = 256, (q5 + 255) // 256  # scan_filter.hip: 256 resident queries per block at dim 384
        exe = 2.0 * rows * tiles * per * dim  # int8 ops per GPU per step
        line["config_5"] = {
            "workload": f"{q5} batched queries per step, top-{k}, over {rows} x {dim} rows per GPU on {N} shards "
                        "(BASELINE.json configs[4]); per shard: int8 MFMA filter + exact f32 refine; one merge",
            "ms_per_step": ms5, "value": total_rows * q5 / (ms5 * 1e-3), "unit": "chunks/s",
            "sampled_queries_equal_host_merge": ok5, "candidate_overflow_reported": bool(overflow5),
            "roofline": {"bound": "mfma", "pipe": "mfma_i8", "unit": "TOP/s", "peak": MFMA_I8_PEAK_TOPS,
                         "achieved": exe / (ms5 * 1e-3) / 1e12, "frac": exe / (ms5 * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                         "executed_i8_ops_per_gpu_per_step": exe,
                         "note": "per GPU, over the whole step (filter phases, re-score, select, gather, merge): the filter "
                                 "kernel alone is reported by the N = 1 line's --nq 1000 run"},
        }
        phase("config_5_s")
    print(json.dumps(line), flush=True)
    st.close()
    if failed:  # the DEFAULT path only: the opt-in gather mode and the RCCL child are reported, never fatal
        raise SystemExit(3)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return main_one_process(args)
    import torch

    from codesearch_amd import VectorStore
    from codesearch_amd.sharded import ShardedVectorStore
    from codesearch_amd.synth import synth_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world  # under a launcher the world decides
    torch.cuda.set_device(local_rank)
    dist = None
    # CS_BENCH_FORCE_DIST=1 (under torch.distributed.run) exercises the RCCL exchange at world 1
    force_dist = os.environ.get("CS_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist

        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device(f"cuda:{local_rank}"))

    def barrier():
        if dist is not None:
            dist.barrier()

    shard = ShardedVectorStore(args.dim, args.rows, rank, world, local_rank, force_exchange=force_dist)
    shard.fill_synthetic(SEED)
    # `value` is quoted on the f32 streaming scan (the north-star kernel) when a step is one query: selected explicitly —
    # by default such a search takes the int8 filter + exact refine (same bits), timed below as `default_routing`
    if args.nq == 1:
        shard.store.set_single_query_route({"stream": shard.store.ROUTE_STREAM, "cost": shard.store.ROUTE_COST,
                                            "filter": shard.store.ROUTE_FILTER}[args.route])
    # the queries arrive on rank 0 (the process a caller of VectorStore::search talks to); for N > 1 every
    # step broadcasts them to the other shards inside the timed region
    q_host = synth_rows(SEED + 1, 0, args.nq, args.dim)
    dev = f"cuda:{local_rank}"
    d_q = torch.from_numpy(q_host).to(dev) if rank == 0 else torch.zeros((args.nq, args.dim), dtype=torch.float32, device=dev)
    bsrc = 0 if dist is not None else None
    torch.cuda.synchronize()

    # N > 1: every shard must answer for its own rows before anything is timed — one planted query per shard (a noisy copy
    # of a row living there; known on rank 0 only, broadcast like every query): the merged top-1 must be that row's global id
    planted_ok = None
    if dist is not None and world > 1:
        from codesearch_amd.synth import synth_planted

        planted_rows = [g * args.rows + (4242 + 1013 * g) % args.rows for g in range(world)]
        planted = synth_planted(SEED, SEED + 7, planted_rows, args.dim)
        planted_ok = []
        for g in range(world):
            pq = torch.from_numpy(planted[g:g + 1].copy()).to(dev) if rank == 0 else torch.zeros((1, args.dim), dtype=torch.float32, device=dev)
            r = shard.search_device(pq, 1, args.k, broadcast_src=0)
            torch.cuda.synchronize()
            planted_ok.append(int(r["ids"][0].item()) & 0xFFFFFFFF == planted_rows[g])
    for _ in range(args.warmup):
        shard.search_device(d_q, args.nq, args.k, broadcast_src=bsrc)
    torch.cuda.synchronize()
    if dist is not None:  # every rank now holds rank 0's queries
        assert torch.equal(d_q.cpu(), torch.from_numpy(q_host)), "query broadcast failed"
    shard.store.profile(True)
    shard.store.profile_read(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = shard.search_device(d_q, args.nq, args.k, broadcast_src=bsrc)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    scan_ms, launches, merge_ms = shard.store.profile_read(reset=True)
    shard.store.profile(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ids0 = out["ids"].cpu().numpy().astype("uint32").reshape(args.nq, args.k)
    cos0 = out["cos"].cpu().numpy().reshape(args.nq, args.k)
    # N > 1: the device merge of the last timed step against a HOST merge of the keys the all-gather delivered (every
    # rank holds all of them): the exchange and the merge kernel checked on the real data, beside the planted rows
    merged_ok = None
    if dist is not None and (world > 1 or force_dist):
        import numpy as np

        from codesearch_amd.sharded import merge_keys_host

        gathered = out["gathered"].cpu().numpy().view(np.uint64).reshape(world, args.nq, args.k)
        merged_ok = bool(np.array_equal(merge_keys_host(gathered, args.k),
                                        out["keys"].cpu().numpy().view(np.uint64).reshape(args.nq, args.k)))

    if rank == 0:
        total_rows = args.rows * world
        value = total_rows * args.nq * args.steps / elapsed
        scan_us = scan_ms * 1e3 / max(launches, 1)
        alg_bytes = args.rows * args.dim * 4  # per launch: every row of the shard read once
        achieved = alg_bytes / (scan_us * 1e-6) / 1e9
        # the scan is HBM-bound below ~39 queries per pass and fp32-MFMA-bound above
        wants_filter = (args.nq >= int(os.environ.get("CS_FILTER_MIN_Q", "2"))  # one query: --route (default: streaming scan)
                        or (args.nq == 1 and args.route != "stream" and (args.route == "filter" or args.rows >= (32_768 if args.k < 48 else 300_000))))
        filter_path = wants_filter and args.dim in (384, 768, 1024) and os.environ.get("CS_INDEX_SPLIT", "1")[0] != "0"
        alg_flops = 2.0 * args.rows * args.nq * args.dim
        int8_path = filter_path and os.environ.get("CS_FILTER_INT8", "1")[0] != "0"
        if int8_path:
            # scan_filter.hip "int8 filter copy": the filter streams the int8 copy of the unit rows (rows*dim B) once per
            # query tile through the i8 MFMA (exact integer products, proven band); candidates are re-scored exactly in f32.
            per = 32 if args.nq <= 32 else 64 if (args.nq <= 64 or args.dim == 1024) else \
                256 if (args.dim == 384 and args.nq > 128) else 128
            tiles = (args.nq + per - 1) // per
            i8_bytes = args.rows * args.dim
            exe = 2.0 * args.rows * tiles * per * args.dim
            rq8 = args.dim == 384 and args.nq > 128 and os.environ.get("CS_FILTER_INT8_RQ", "1")[0] != "0"
            kern = (f"cs::score_filter_rq8_kernel<8,3> (eight waves per block, corpus fragments through registers)" if rq8 else
                    f"cs::score_filter_rw8_kernel<{per // 32},{args.dim // 128}>") + \
                " (+ rescore_keys_kernel / select_candidates_kernel between phases)"
            hbm = {"kernel": kern, "bound": "hbm", "achieved": i8_bytes / (scan_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS,
                   "unit": "GB/s", "frac": i8_bytes / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                   "algorithmic_bytes_per_launch": i8_bytes,
                   "f32_bytes_equivalent_frac": alg_bytes / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                   "note": "bytes = the int8 filter copy (a quarter of the f32 matrix), read once when all queries fit "
                           "one query tile; avg_launch_us spans every phase's filter, re-score and select kernels"}
            mfma = {"kernel": kern, "bound": "mfma", "achieved": exe / (scan_us * 1e-6) / 1e12, "peak": MFMA_I8_PEAK_TOPS,
                    "unit": "TOP/s", "frac": exe / (scan_us * 1e-6) / 1e12 / MFMA_I8_PEAK_TOPS,
                    "traffic": None, "executed_i8_ops_per_launch": exe,
                    "algorithmic_flops_per_launch": alg_flops,
                    "l2_GBps_for_information": i8_bytes * tiles / (scan_us * 1e-6) / 1e9,
                    "frac_of_sustained_mfma_stream": exe / (scan_us * 1e-6) / 1e12 / MFMA_I8_32x32x32_SUSTAINED_TOPS,
                    "note": "every query tile re-reads the int8 copy through its XCD's L2; the kernel is paced by that "
                            "stream and by LDS reads, not by the matrix pipe (DESIGN.md §3.1c)"}
            roof = hbm if tiles == 1 else mfma
        elif filter_path:
            # scan_filter.hip: the filter streams the f16 unit-vector copy of the corpus (rows*dim*2 B)
            # once per 128-query tile through the f16 MFMA; candidates are re-scored exactly in f32.
            tiles = (args.nq + 127) // 128
            f16_bytes = args.rows * args.dim * 2
            exe = 2.0 * args.rows * tiles * 128 * args.dim
            rw_max_q = {384: 64, 768: 64, 1024: 32}[args.dim]  # scan_filter.hip scan_split_impl: resident-query kernel
            hbm = {"kernel": ("cs::score_filter_rw_kernel" if args.nq <= rw_max_q else "cs::score_filter_kernel")
                             + " (+ rescore_keys_kernel / select_candidates_kernel between phases)",
                   "bound": "hbm", "achieved": f16_bytes / (scan_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS,
                   "unit": "GB/s", "frac": f16_bytes / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                   "algorithmic_bytes_per_launch": f16_bytes,
                   "note": "bytes = the f16 filter copy, read once when all queries fit one 128-query tile"}
            mfma = {"kernel": "cs::score_filter256p_kernel (+ rescore_keys_kernel / select_candidates_kernel between phases)",
                    "bound": "mfma", "achieved": exe / (scan_us * 1e-6) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": exe / (scan_us * 1e-6) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                    "traffic": None, "executed_f16_flops_per_launch": exe,
                    "algorithmic_flops_per_launch": alg_flops,
                    "hbm_GBps_for_information": f16_bytes * tiles / (scan_us * 1e-6) / 1e9}
            roof = hbm if tiles == 1 else mfma
        elif args.nq >= 40:
            roof = {"kernel": "cs::score_append_kernel (+ select_candidates_kernel between phases)",
                    "bound": "mfma", "achieved": alg_flops / (scan_us * 1e-6) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": alg_flops / (scan_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                    "traffic": None, "algorithmic_flops_per_launch": alg_flops,
                    "hbm_GBps_for_information": achieved}
        else:
            deep = args.k <= 64 or (args.k <= 128 and args.rows >= 4_000_000)  # scan.hip scan_deep()
            roof = {"kernel": ("cs::scan_topk_kernel<3,8,1,true,false>" if deep else "cs::scan_topk_kernel<3,4,1,true,false>")
                    if args.dim == 384 and args.nq == 1
                    else ("cs::score_append_kernel" if args.nq >= 5 else "cs::scan_topk_kernel"),
                    "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "note": "avg_launch_us is the HIP-event span of the prime pass (scan_topk_kernel<..,false,true> "
                            "over the first n/256 <= 16384 rows, ~10 us) plus the scan kernel; bytes count the scan only"}
        roof.update({"avg_launch_us": scan_us, "launches_timed": launches,
                     "merge_avg_us": merge_ms * 1e3 / max(launches, 1)})
        line = {
            # BASELINE.json's metric, verbatim.  `value` = its search half at the north-star target (chunks searched/s by
            # the exact scan); the embedding half and the literal embedded+searched figure are the `encoder`,
            # `embed_search` and `embedded_and_searched_chunks_per_s` fields below.
            "metric": "chunks embedded+searched/sec over 10M\u00d7384 corpus; recall@10 vs CPU ref",
            "value_is": f"chunks searched/sec: brute-force cosine top-{args.k}, {args.nq} query/step over "
                        f"{args.rows} x {args.dim} fp32 rows per GPU",
            "value": value,
            "unit": "chunks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"brute-force cosine top-{args.k}, {args.nq} query/step over "
                            f"{args.rows} x {args.dim} fp32 rows per GPU (BASELINE.json north-star target; "
                            f"rows generated in HBM by include/cs_synth.h, seed {SEED:#x})",
                "rows_per_gpu": args.rows, "dim": args.dim, "queries_per_step": args.nq, "k": args.k,
                "parallelism": f"row-sharded x{world}, all-gather of per-shard top-k" if world > 1 else "single GPU",
            },
            "roofline": roof,
            "top1": {"id": int(ids0[0][0]), "cos": float(cos0[0][0])},
            "collective": ("RCCL (torch.distributed nccl backend): broadcast of the queries, all-gather of per-shard top-k keys"
                           if dist is not None else "none (one GPU)"),
            "rccl_world_size": (dist.get_world_size() if dist is not None else None),
        }
        if planted_ok is not None or merged_ok is not None:
            line["multi_gpu_checks"] = {"planted_query_per_shard_returns_its_row": planted_ok,
                                        "device_merge_equals_host_merge_of_gathered_keys": merged_ok}
            if planted_ok is not None and not all(planted_ok):
                line["error"] = "a shard did not return its planted row"
            if merged_ok is False:
                line["error"] = (line.get("error", "") + "; " if line.get("error") else "") + \
                    "the device merge differs from the host merge of the gathered keys"
        traffic_file = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                if tr.get("rows") == args.rows and tr.get("dim") == args.dim and args.nq == 1:
                    line["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
                    line["roofline"]["traffic_source"] = tr.get("source")
                    # PMC counters cannot be read inside this run (rocprofv3 passes of their own): the figure is the
                    # committed measurement, stamped with the commit of the kernel it was taken on
                    line["roofline"]["traffic_measured_at"] = tr.get("measured_at_commit")
                    line["roofline"]["traffic_is"] = "static: profiles/scan_traffic.json (benchmarks/derive_scan_traffic.py)"
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            from tests.oracle_lib import load_oracle

            timed_route = args.route if args.nq == 1 and args.route in ("stream", "cost") else "cost"
            base, recall, err, checks = cpu_baseline(load_oracle(), args.cpu_sample_rows, args.dim, args.k, VectorStore, timed_route)
            line["cpu_baseline"] = base
            line["recall_at_10"] = recall
            line["max_abs_cos_err_vs_cpu"] = err
            line["recall_checks"] = dict(checks, quoted=f"recall_at_10 and max_abs_cos_err_vs_cpu are the '{timed_route}' entry "
                                                         f"(the route `value` was timed on); the 10M-row results of both routes are "
                                                         f"held to the oracle by tests/test_gpu_scan.py")
        if world == 1 and args.nq == 1 and args.route == "stream" and not args.only_scan:
            # The DEFAULT route of the same search (CS_ROUTE_COST, index.hip run_search): one query over >= 32,768 rows (k < 48; 300,000 from k = 48) goes
            # through the MFMA filter over the int8 copy (a quarter of the f32 bytes) + exact f32 re-score — the shape of the
            # reference's MCP and HTTP searches (src/mcp/mod.rs:252, src/server/mod.rs:547).  Same bits as the streaming scan.
            def timed_single(kk, reps=50):
                ks = torch.zeros(kk, dtype=torch.int64, device=dev)
                for _ in range(3):
                    shard.search_device(d_q, 1, kk)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    r = shard.search_device(d_q, 1, kk)
                torch.cuda.synchronize()
                ms_ = (time.perf_counter() - t1) * 1e3 / reps
                del ks
                return ms_, r["ids"].cpu().numpy().astype("uint32").reshape(-1), r["cos"].cpu().numpy().reshape(-1)

            route = {}
            for kk in sorted({args.k, 25, 75}):
                shard.store.set_single_query_route(shard.store.ROUTE_STREAM)
                s_ms, s_ids, s_cos = timed_single(kk, reps=20)
                b0, _ = shard.store.debug_counters()
                shard.store.set_single_query_route(shard.store.ROUTE_COST)
                d_ms, d_ids, d_cos = timed_single(kk)
                b1, _ = shard.store.debug_counters()
                route[f"k{kk}"] = {"default_ms_per_search": d_ms, "streaming_ms_per_search": s_ms,
                                   "default_took_filter_path": b1 > b0,
                                   "bit_identical": bool((d_ids == s_ids).all() and d_cos.tobytes() == s_cos.tobytes())}
            shard.store.set_single_query_route(shard.store.ROUTE_STREAM)
            copy, spread, reruns = shard.store.filter_state()
            has8, has16, fbytes = shard.store.filter_copies()
            line["default_routing_ms_per_search"] = route[f"k{args.k}"]["default_ms_per_search"]
            line["default_routing"] = {
                "route": "CS_ROUTE_COST (default): int8 filter + exact f32 refine for one query over >= 32,768 rows at k < 48, >= 300,000 rows from k = 48",
                "per_k": route,
                "chunks_per_s": args.rows / (route[f"k{args.k}"]["default_ms_per_search"] * 1e-3),
                "filter_copy": {2: "int8", 1: "f16", 0: "none"}[copy], "copies_in_hbm": {"int8": has8, "f16": has16},
                "filter_copy_bytes": fbytes, "f32_bytes": args.rows * args.dim * 4,
                "note": "`value` is measured with cs_index_set_single_query_route(CS_ROUTE_STREAM), the f32 streaming scan "
                        "BASELINE's roofline target is quoted on; this is what a caller gets without asking",
            }
        if world == 1 and args.nq == 1 and args.dim == 384 and not args.no_1m:
            line["config_1m"] = scan_1m_leg(args.dim, args.k, local_rank, VectorStore)
        if world == 1 and not args.only_scan:
            ref = hbm_reference(f"cuda:{local_rank}")
            line["hbm_reference"] = ref
            if line["roofline"]["bound"] == "hbm" and ref.get("read_probe_GBps"):
                line["roofline"]["frac_of_read_probe"] = line["roofline"]["achieved"] / ref["read_probe_GBps"]
        if world == 1 and not args.no_encoder:
            line.update(encoder_legs(shard, args.k, local_rank, with_cpu=not args.no_cpu_baseline))
            # (flat copies of the two encoder figures the reviews track, so the driver's `parsed` record shows them)
            line["encoder_ms_per_batch"] = line["encoder"].get("ms_per_batch")  # BGE-small shape, 256 x 256 tokens, mean pooling (configs[2])
            qd = (line["encoder"].get("quantized_default_model") or {}).get("dynamic_quantisation") or {}
            line["q8_default_model_ms_per_batch"] = qd.get("ms_per_batch")      # AllMiniLML6V2Q shape, 256 x 256 tokens (the reference's default model)
            line["query_embed_device_us"] = (line["encoder"].get("query_embed") or {}).get("device_us")  # one 16-token query, BGE-small shape
            line["query_embed_default_model_device_us"] = ((line["encoder"].get("quantized_default_model") or {}).get("query_embed") or {}).get("device_us")
            line["embedded_and_searched_chunks_per_s"] = line["embed_search"]["chunks_embedded_and_searched_per_s"]
            # BASELINE's metric as worded ("chunks embedded+searched/sec over 10M x 384"): the literal figure beside
            # `value`, which is its search half alone at the north-star target
            line["value_literal_metric"] = line["embedded_and_searched_chunks_per_s"]
            line["value_literal_metric_unit"] = "chunks embedded AND searched per second (256 x 256-token chunks per step)"
            line["embedded_and_searched_config"] = (
                f"256 chunks x 256 tokens embedded by the BGE-small-shaped HIP encoder, then one batched top-{args.k} "
                f"search of the 256 embeddings over the resident {args.rows} x {args.dim} fp32 corpus (all on one GPU)")
        if world == 1 and not args.no_e2e:
            del shard
            torch.cuda.empty_cache()
            line["e2e_index_search"] = e2e_leg(args.e2e_chunks, args.k, local_rank)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
