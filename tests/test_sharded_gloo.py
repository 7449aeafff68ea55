"""N>1 data path on CPU: two processes, gloo backend.  Each rank scores its own row shard
(the CPU oracle stands in for the per-shard GPU scan, which needs a device), packs (cos, id)
into the 64-bit keys the HIP kernels emit, all-gathers them in the [world, nq, k] layout that
cs_merge_topk_device consumes, merges, and every rank must end with exactly the single-store
result.  Covers shard ranges, global ids (id_base), key order and the exchange layout."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_rows, dim, nq, k, seed, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from codesearch_amd.sharded import key_pack, key_unpack, merge_keys_host, shard_range
    from codesearch_amd.synth import synth_rows
    from tests.oracle_lib import load_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = load_oracle()
    lo, hi = shard_range(rank, world, total_rows)
    rows = oracle.synth_rows(seed, lo, hi - lo, dim)  # shard regenerates its own range
    queries = synth_rows(seed + 1, 0, nq, dim)
    local = np.zeros((nq, k), np.uint64)
    for i in range(nq):
        cos, ids = oracle.scan_topk(rows, queries[i], k, id_base=lo, mode="omp", threads=2)
        local[i, : len(ids)] = key_pack(cos, ids)
    mine = torch.from_numpy(local.view(np.int64).reshape(-1).copy())
    gathered = torch.zeros(world * nq * k, dtype=torch.int64)
    dist.all_gather_into_tensor(gathered, mine)
    merged = merge_keys_host(gathered.numpy().view(np.uint64).reshape(world, nq, k), k)
    cos, ids = key_unpack(merged)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cos=cos, ids=ids, lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total_rows", [(2, 20_001), (3, 9_000)])
def test_allgather_merge_equals_single_store(tmp_path, oracle, world, total_rows):
    import torch.multiprocessing as mp

    from codesearch_amd.synth import synth_rows

    dim, nq, k, seed = 384, 5, 10, 4242
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total_rows, dim, nq, k, seed, str(tmp_path)), nprocs=world, join=True)
    corpus = oracle.synth_rows(seed, 0, total_rows, dim)
    queries = synth_rows(seed + 1, 0, nq, dim)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert outs[0]["lo"] == 0 and outs[-1]["hi"] == total_rows
    for r in range(1, world):
        assert outs[r]["lo"] == outs[r - 1]["hi"]
    for i in range(nq):
        ecos, eids = oracle.scan_topk(corpus, queries[i], k, mode="omp")
        for o in outs:  # every rank holds the identical, exact result
            assert o["ids"][i].tolist() == eids.tolist()
            assert np.array_equal(o["cos"][i], ecos)


def test_shard_range_partitions():
    from codesearch_amd.sharded import shard_range

    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 80_000_000):
            spans = [shard_range(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
