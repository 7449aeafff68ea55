"""N>1 data path on CPU: two and three processes, gloo backend, running the PRODUCT class
codesearch_amd.sharded.ShardedVectorStore.  Only its backend is swapped: the CPU oracle stands in for the
per-shard GPU scan and merge_keys_host for cs_merge_topk_device (both need a device); the query broadcast,
the buffers, the all-gather in the [world, nq, k] layout and the call order are the class's own.  Every
rank must end with exactly the single-store result.  Covers global ids (id_base), key order, the broadcast
and the exchange layout."""
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


class OracleShardBackend:
    """Stand-in for HipShardBackend on a machine without a GPU: the CPU oracle scores this rank's shard and
    merge_keys_host merges — everything else (buffers, broadcast, all-gather layout, call order) is the
    product class ShardedVectorStore itself."""

    tensor_device = "cpu"

    def __init__(self, oracle, dim, rank, rows_per_shard):
        self.oracle, self.dim, self.lo = oracle, dim, rank * rows_per_shard
        self.rows = None

    def fill_synthetic(self, n, seed, first_row):
        assert first_row == self.lo
        self.rows = self.oracle.synth_rows(seed, first_row, n, self.dim)  # the shard regenerates its own range

    def search_local(self, d_queries, nq, k, keys, cos=None, ids=None, counts=None):
        from codesearch_amd.sharded import key_pack, key_unpack

        q = d_queries.numpy()
        local = np.zeros((nq, k), np.uint64)
        for i in range(nq):
            c, ii = self.oracle.scan_topk(self.rows, q[i], k, id_base=self.lo, mode="omp", threads=2)
            local[i, : len(ii)] = key_pack(c, ii)
        keys.copy_(__import__("torch").from_numpy(local.view(np.int64).reshape(-1)))
        if cos is not None:
            self._decode(local, cos, ids, counts, nq, k)

    def merge(self, gathered, world, nq, k, keys, cos, ids, counts):
        import torch

        from codesearch_amd.sharded import merge_keys_host

        merged = merge_keys_host(gathered.numpy().view(np.uint64).reshape(world, nq, k), k)
        keys.copy_(torch.from_numpy(merged.view(np.int64).reshape(-1)))
        self._decode(merged, cos, ids, counts, nq, k)

    @staticmethod
    def _decode(merged, cos, ids, counts, nq, k):
        import torch

        from codesearch_amd.sharded import key_unpack

        c, i = key_unpack(merged)
        cos.copy_(torch.from_numpy(np.where(merged != 0, c, 0).astype(np.float32).reshape(-1)))
        ids.copy_(torch.from_numpy(i.astype(np.int64).astype(np.int32, casting="unsafe").reshape(-1)))
        counts.copy_(torch.from_numpy((merged != 0).sum(1).astype(np.int32)))


def _worker(rank, world, port, rows_per_shard, dim, nq, k, seed, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from codesearch_amd.sharded import ShardedVectorStore
    from codesearch_amd.synth import synth_rows
    from tests.oracle_lib import load_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = load_oracle()
    sh = ShardedVectorStore(dim, rows_per_shard, rank, world, device=0,
                            backend=OracleShardBackend(oracle, dim, rank, rows_per_shard))
    sh.fill_synthetic(seed)
    # the queries exist on rank 0 only; every other rank starts from zeros and receives the broadcast
    q = torch.from_numpy(synth_rows(seed + 1, 0, nq, dim)) if rank == 0 else torch.zeros((nq, dim), dtype=torch.float32)
    out = sh.search_device(q, nq, k, broadcast_src=0)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cos=out["cos"].numpy().reshape(nq, k),
             ids=out["ids"].numpy().astype(np.int64).astype(np.uint32).reshape(nq, k), counts=out["counts"].numpy(),
             keys=out["keys"].numpy().view(np.uint64).reshape(nq, k), q=q.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows_per_shard", [(2, 10_001), (3, 3_000)])
def test_allgather_merge_equals_single_store(tmp_path, oracle, world, rows_per_shard):
    """The product class ShardedVectorStore at world 2 and 3 over gloo: broadcast of rank 0's queries, local
    search, ONE all_gather_into_tensor in the [world, nq, k] layout, merge — every rank must end with the
    exact single-store answer for the whole corpus."""
    import torch.multiprocessing as mp

    from codesearch_amd.synth import synth_rows

    dim, nq, k, seed = 384, 5, 10, 4242
    port = _free_port()
    mp.spawn(_worker, args=(world, port, rows_per_shard, dim, nq, k, seed, str(tmp_path)), nprocs=world, join=True)
    total_rows = world * rows_per_shard
    corpus = oracle.synth_rows(seed, 0, total_rows, dim)
    queries = synth_rows(seed + 1, 0, nq, dim)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        assert np.array_equal(o["q"], queries)  # the broadcast delivered rank 0's queries
    for i in range(nq):
        ecos, eids = oracle.scan_topk(corpus, queries[i], k, mode="omp")
        for o in outs:  # every rank holds the identical, exact result
            assert o["counts"][i] == k
            assert o["ids"][i].tolist() == eids.tolist()
            assert np.array_equal(o["cos"][i], ecos)
    assert all(np.array_equal(o["keys"], outs[0]["keys"]) for o in outs)


def test_shard_range_partitions():
    from codesearch_amd.sharded import shard_range

    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 80_000_000):
            spans = [shard_range(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
