#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json): brute-force cosine
top-10 over a 10M x 384 fp32 corpus resident in HBM, one MI355X per rank.

A "step" = one pass of the hot path over one batch of synthetic input: the query batch
(already in HBM) is scored against every row of the rank's shard by the fused scan +
top-k kernel, block partials are merged, and for N > 1 the per-shard top-k lists are
all-gathered over RCCL and merged.  `value` = corpus rows ("chunks") searched per second
over all ranks = N * rows_per_gpu * nq * K / t.

  python bench.py                      # N=1, 10M x 384, Q=1, k=10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N

Adds to the JSON line:
  roofline     — the scan kernel against the 8 TB/s HBM peak: algorithmic bytes per launch
                 (rows*dim*4) / HIP-event duration of that kernel, measured live.
  cpu_baseline — the CPU oracle's tuned port timed on this box's host cores over a bounded
                 sample (reported baseline, not the target).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
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
    return ap.parse_args()


def encoder_legs(shard, k, device):
    """BASELINE.json's metric also names embedding: (i) the BGE-small encoder alone at
    configs[2] (batch 256 x seq 256, synthetic weights) and (ii) embed a batch of 256 query chunks
    then search them against the resident corpus ("chunks embedded+searched/sec").  Reported
    next to `value`, which stays the scan (north-star target)."""
    import torch

    from codesearch_amd import BertConfig, FastEmbedder, ModelType
    from codesearch_amd.bert_params import synth_token_batch

    cfg = BertConfig.bge_small()
    emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202, device=device)
    B, L = 256, 256
    ids, mask = synth_token_batch(cfg, 999, B, L, False)
    d_q = torch.empty((B, cfg.hidden), dtype=torch.float32, device=f"cuda:{device}")
    emb.embed_ids_to_device(ids, mask, d_q.data_ptr())  # warm-up, allocates the workspace
    shard.search_device(d_q, B, k)
    torch.cuda.synchronize()
    emb.profile_read(reset=True)
    iters = 5
    t0 = time.perf_counter()
    for _ in range(iters):
        emb.embed_ids_to_device(ids, mask, d_q.data_ptr())
        shard.search_device(d_q, B, k)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    ms, n = emb.profile_read()
    ms /= max(n, 1)
    H, I, layers = cfg.hidden, cfg.intermediate, cfg.layers
    gemm_flops = layers * 2 * (4 * H * H + 2 * H * I) * B * L
    attn_flops = layers * 4 * L * H * B * L
    split, f32n, fb = emb.debug_counters()
    emb.close()
    return {
        "encoder": {
            "workload": f"BGE-small-en-v1.5 shape (12 x [MHA, GELU FFN, LN], hidden 384), batch {B} x seq {L}, "
                        "synthetic weights, CLS pool + L2 normalise",
            "ms_per_batch": ms, "chunks_per_s": B / (ms * 1e-3),
            "algorithmic_tflops": (gemm_flops + attn_flops) / (ms * 1e-3) / 1e12,
            "dense_layers": "split-f16 operands on v_mfma_f32_16x16x32_f16, 3 MFMAs per f32 product block",
            "executed_f16_mfma_tflops": 3 * gemm_flops / (ms * 1e-3) / 1e12,
            "f16_mfma_peak_tflops": MFMA_F16_PEAK_TFLOPS,
            "split_forwards": split, "f32_fallbacks": fb,
        },
        "embed_search": {
            "workload": f"embed {B} query chunks (seq {L}) on the GPU, then one batched top-{k} search of them "
                        f"over the resident corpus",
            "ms_per_batch": wall * 1e3, "chunks_embedded_and_searched_per_s": B / wall,
        },
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
    out = {"workload": f"brute-force cosine top-{k}, 1 query over {rows} x {dim} fp32 rows (BASELINE.json configs[1])",
           "ms_per_search": wall * 1e3, "chunks_per_s": rows / wall, "scan_kernel_us": us,
           "hbm_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS}
    del st
    return out


def hbm_reference(device, nbytes=2 << 30, reps=10):
    """a measured device copy / fill bandwidth beside the 8 TB/s spec figure, so the
    scan's fraction can be read against both.  torch's copy_ and zero_ kernels on `nbytes`."""
    import torch

    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    out = {}
    for name, fn, moved in (("copy", lambda: dst.copy_(src), 2 * nbytes), ("fill", lambda: dst.zero_(), nbytes),
                            ("read", lambda: torch.sum(src), nbytes)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_GBps"] = moved / (e0.elapsed_time(e1) * 1e-3 / reps) / 1e9
    out["note"] = (f"torch copy_ (read + write bytes counted), zero_ and sum (a read-only pass, the scan's own traffic "
                   f"shape) over {nbytes >> 20} MiB on this device; "
                   "MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy")
    del src, dst
    torch.cuda.empty_cache()
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


def cpu_baseline(oracle, sample_rows, dim, k, store_cls):
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
    # recall@k of the HIP path vs the CPU result on the same sample
    st = store_cls(None, dim, device=0)
    st.insert_synthetic(sample_rows, SEED, 0)
    st.build_index()
    cos, ids, _ = st.search_raw(q, k)
    st.close()
    recall = len(set(ids[0].tolist()) & set(res[1].tolist())) / float(k)
    max_err = float(np.abs(cos[0] - res[0]).max())
    return {
        "value": omp_rate, "unit": "chunks/s", "cores": threads, "kind": "port",
        "sample": f"{sample_rows} x {dim} fp32 rows of the same synthetic corpus, 1 query, top-{k}, "
                  f"{reps} passes of oracle/scan_oracle.c cs_oracle_scan_topk_omp ({threads} threads = this job's CPU "
                  f"share; the host shows {oracle.num_threads()})",
        "literal_1thread_chunks_per_s": lit_rate,
        "literal_sample_rows": lit_rows,
    }, recall, max_err


def main():
    args = parse_args()
    import torch

    from codesearch_amd import VectorStore
    from codesearch_amd.sharded import ShardedVectorStore
    from codesearch_amd.synth import synth_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
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
    q_host = synth_rows(SEED + 1, 0, args.nq, args.dim)  # same queries on every rank
    d_q = torch.from_numpy(q_host).to(f"cuda:{local_rank}")
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        shard.search_device(d_q, args.nq, args.k)
    torch.cuda.synchronize()
    shard.store.profile(True)
    shard.store.profile_read(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = shard.search_device(d_q, args.nq, args.k)
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

    if rank == 0:
        total_rows = args.rows * world
        value = total_rows * args.nq * args.steps / elapsed
        scan_us = scan_ms * 1e3 / max(launches, 1)
        alg_bytes = args.rows * args.dim * 4  # per launch: every row of the shard read once
        achieved = alg_bytes / (scan_us * 1e-6) / 1e9
        # the scan is HBM-bound below ~39 queries per pass and fp32-MFMA-bound above
        single_min_k = int(os.environ.get("CS_FILTER_SINGLE_MIN_K", "100"))  # index.hip run_search: one query, long list
        wants_filter = (args.nq >= int(os.environ.get("CS_FILTER_MIN_Q", "2"))
                        or (args.nq == 1 and single_min_k and args.k >= single_min_k and args.rows >= 2_000_000))
        filter_path = wants_filter and args.dim in (384, 768, 1024) and os.environ.get("CS_INDEX_SPLIT", "1")[0] != "0"
        alg_flops = 2.0 * args.rows * args.nq * args.dim
        if filter_path:
            # scan_filter.hip: the filter streams the f16 unit-vector copy of the corpus (rows*dim*2 B)
            # once per 128-query tile through the f16 MFMA; candidates are re-scored exactly in f32.
            tiles = (args.nq + 127) // 128
            f16_bytes = args.rows * args.dim * 2
            exe = 2.0 * args.rows * tiles * 128 * args.dim
            hbm = {"kernel": ("cs::score_filter_rw_kernel" if args.nq <= 64 and args.dim == 384 else "cs::score_filter_kernel")
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
            "metric": "chunks searched/sec, brute-force cosine top-10 over 10M x 384 fp32 corpus per GPU",
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
        }
        traffic_file = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                if tr.get("rows") == args.rows and tr.get("dim") == args.dim and args.nq == 1:
                    line["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
                    line["roofline"]["traffic_source"] = tr.get("source")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            from tests.oracle_lib import load_oracle

            base, recall, err = cpu_baseline(load_oracle(), args.cpu_sample_rows, args.dim, args.k, VectorStore)
            line["cpu_baseline"] = base
            line["recall_at_10"] = recall
            line["max_abs_cos_err_vs_cpu"] = err
        if world == 1 and args.nq == 1:
            # for information: the same exact search routed through the f16 filter + f32 refine path
            # (bit-identical result; reads the half-size filter copy instead of the f32 matrix)
            shard.store.set_filter_min_queries(1)
            for _ in range(3):
                shard.search_device(d_q, 1, args.k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 50
            for _ in range(reps):
                alt = shard.search_device(d_q, 1, args.k)
            torch.cuda.synchronize()
            alt_ms = (time.perf_counter() - t1) * 1e3 / reps
            same = bool((alt["ids"].cpu().numpy().astype("uint32").reshape(-1) == ids0.reshape(-1)).all()
                        and (alt["cos"].cpu().numpy().reshape(-1) == cos0.reshape(-1)).all())
            shard.store.set_filter_min_queries(2)
            line["single_query_via_filter"] = {
                "ms_per_search": alt_ms, "chunks_per_s": args.rows / (alt_ms * 1e-3),
                "bit_identical_to_streaming_scan": same,
                "note": "cs_index_set_filter_min_queries(1): f16 MFMA filter over the 7.68 GB unit-vector copy, "
                        "then exact f32 re-score of the candidates; not used for `value`",
            }
        if world == 1 and args.nq == 1 and args.dim == 384:
            line["config_1m"] = scan_1m_leg(args.dim, args.k, local_rank, VectorStore)
        if world == 1:
            ref = hbm_reference(f"cuda:{local_rank}")
            line["hbm_reference"] = ref
            if line["roofline"]["bound"] == "hbm":
                line["roofline"]["frac_of_measured_copy"] = line["roofline"]["achieved"] / ref["copy_GBps"]
        if world == 1 and not args.no_encoder:
            line.update(encoder_legs(shard, args.k, local_rank))
        print(json.dumps(line), flush=True)

    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
