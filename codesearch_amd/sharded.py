"""Row-sharded store: one process per GPU, one exchange step.

Shard g owns the contiguous id range [g*rows_per_shard, (g+1)*rows_per_shard): ids are
global (`cs_index_create(..., id_base)`), so no translation happens after the exchange.
A search is: every rank scans its shard -> [nq, k] packed keys -> ONE all-gather of
nq*k*8 bytes per rank (RCCL over xGMI; `nccl` backend) -> every rank merges the
world_size lists with the same HIP merge kernel.  top-k of a union = top-k of the
per-shard top-ks, so the result equals a single-GPU scan of the whole corpus.

The key helpers below are pure numpy so the N>1 data path can be exercised on CPU with
the gloo backend (tests/test_sharded_gloo.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np


def shard_range(rank: int, world: int, total_rows: int) -> Tuple[int, int]:
    """Contiguous, balanced row range of a shard (first rows get the remainder)."""
    base, rem = divmod(total_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def key_pack(cos: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """cs_key_pack: order-preserving f32 image in the high word, ~id in the low word."""
    c = (np.asarray(cos, np.float32) + np.float32(0.0)).view(np.uint32).astype(np.uint64)
    o = np.where(c & np.uint64(0x80000000), (~c) & np.uint64(0xFFFFFFFF), c | np.uint64(0x80000000))
    return (o << np.uint64(32)) | ((~np.asarray(ids, np.uint32)).astype(np.uint64))


def key_unpack(keys: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """cs_key_cos / cs_key_id."""
    keys = np.asarray(keys, np.uint64)
    o = (keys >> np.uint64(32)).astype(np.uint32)
    u = np.where(o & np.uint32(0x80000000), o & np.uint32(0x7FFFFFFF), ~o)
    return u.astype(np.uint32).view(np.float32), ~(keys & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def merge_keys_host(gathered: np.ndarray, k: int) -> np.ndarray:
    """Host statement of the shard merge: gathered [world, nq, k] keys (0 = empty) ->
    [nq, k] best-first.  Used by the gloo tests; the product path is cs_merge_topk_device."""
    world, nq, kk = gathered.shape
    flat = np.transpose(gathered, (1, 0, 2)).reshape(nq, world * kk)
    return np.sort(flat, axis=1)[:, ::-1][:, :k].copy()


class HipShardBackend:
    """The product backend of ShardedVectorStore: this rank's shard is a cs_index in HBM, buffers are torch
    CUDA tensors (plumbing), kernels are libcsgpu's, launched on torch's current stream so RCCL orders
    after them."""

    def __init__(self, dim: int, rows_per_shard: int, rank: int, device: int):
        import torch

        from . import _lib
        from .vector_store import VectorStore

        self.torch, self._lib, self._check = torch, _lib.load(), _lib.check
        self.dim, self.device = dim, device
        self.tensor_device = f"cuda:{device}"
        self.store = VectorStore(None, dim, device=device, capacity=rows_per_shard, id_base=rank * rows_per_shard)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def fill_synthetic(self, n: int, seed: int, first_row: int) -> None:
        self.store.insert_synthetic(n, seed, first_row)
        self.store.build_index()

    GATED_MAX_Q = 16  # csrc/scan.hpp kGatedMaxQ: up to this many queries a device search repairs itself

    def search_local(self, d_queries, nq: int, k: int, keys, cos=None, ids=None, counts=None) -> None:
        """Exact on return-to-stream: searches of more than 16 queries report a candidate-buffer overflow (adversarial
        row orders) through cs_index_search_status instead of rerunning on the device; it is asked here, BEFORE the
        keys enter the all-gather, and an overflowed search is redone in slices of 16 queries, which always are exact.
        The status call waits for this rank's search — one stream synchronisation per > 16-query search."""
        vp = lambda x, off=0: None if x is None else C.c_void_p(x.data_ptr() + off)
        self._check(self._lib.cs_index_search_device(self.store.handle, vp(d_queries), nq, self.dim, k, vp(keys),
                                                     vp(cos), vp(ids), vp(counts), self._stream()))
        if nq <= self.GATED_MAX_Q:
            return
        ov = C.c_uint32(0)
        self._check(self._lib.cs_index_search_status(self.store.handle, self._stream(), C.byref(ov)))
        if not ov.value:
            return
        self.overflow_reruns = getattr(self, "overflow_reruns", 0) + 1
        for q0 in range(0, nq, self.GATED_MAX_Q):
            m = min(self.GATED_MAX_Q, nq - q0)
            self._check(self._lib.cs_index_search_device(
                self.store.handle, vp(d_queries, q0 * self.dim * 4), m, self.dim, k, vp(keys, q0 * k * 8),
                vp(cos, q0 * k * 4), vp(ids, q0 * k * 4), vp(counts, q0 * 4), self._stream()))

    def merge(self, gathered, world: int, nq: int, k: int, keys, cos, ids, counts) -> None:
        vp = lambda x: C.c_void_p(x.data_ptr())
        self._check(self._lib.cs_merge_topk_device(self.device, vp(gathered), world, nq, k, vp(keys), vp(cos), vp(ids),
                                                   vp(counts), self._stream()))


class ShardedVectorStore:
    """One rank's shard plus the exchange: (optional) query broadcast -> local search -> ONE all-gather of
    nq*k*8 bytes per rank -> merge on every rank.  `backend` supplies the local search and the merge:
    HipShardBackend (default; needs a GPU) or, in the CPU tests, a stand-in built on the oracle — the
    exchange sequence below is the same code either way."""

    def __init__(self, dim: int, rows_per_shard: int, rank: int, world: int, device: int, group=None,
                 force_exchange: bool = False, backend=None):
        import torch

        self.torch = torch
        self.dim, self.rank, self.world, self.device = dim, rank, world, device
        self.rows_per_shard = rows_per_shard
        self.group = group
        self.force_exchange = force_exchange  # run the all-gather + merge even when world == 1
        self.backend = backend if backend is not None else HipShardBackend(dim, rows_per_shard, rank, device)
        self.store = getattr(self.backend, "store", None)
        self._bufs = {}

    def fill_synthetic(self, seed: int) -> None:
        self.backend.fill_synthetic(self.rows_per_shard, seed, self.rank * self.rows_per_shard)

    def _buffers(self, nq: int, k: int):
        key = (nq, k)
        if key not in self._bufs:
            t, dev = self.torch, self.backend.tensor_device
            self._bufs[key] = dict(
                local=t.zeros(nq * k, dtype=t.int64, device=dev),
                gathered=t.zeros(self.world * nq * k, dtype=t.int64, device=dev),
                keys=t.zeros(nq * k, dtype=t.int64, device=dev),
                cos=t.zeros(nq * k, dtype=t.float32, device=dev),
                ids=t.zeros(nq * k, dtype=t.int32, device=dev),
                counts=t.zeros(nq, dtype=t.int32, device=dev),
            )
        return self._bufs[key]

    def search_device(self, d_queries, nq: int, k: int, broadcast_src: Optional[int] = None):
        """d_queries: torch f32 tensor [nq, dim] on this rank's device.  With broadcast_src = r the queries
        are those of rank r, sent to every shard first (RCCL broadcast of nq*dim*4 bytes: the caller of
        VectorStore::search lives in one process); otherwise every rank must already
        hold the same queries.  Asynchronous; returns the dict of device result tensors
        (cos/ids/counts/keys)."""
        t = self.torch
        b = self._buffers(nq, k)
        exchange = self.world > 1 or self.force_exchange
        if broadcast_src is not None and exchange:
            t.distributed.broadcast(d_queries, src=broadcast_src, group=self.group)
        if not exchange:
            self.backend.search_local(d_queries, nq, k, b["keys"], b["cos"], b["ids"], b["counts"])
            return b
        self.backend.search_local(d_queries, nq, k, b["local"])
        t.distributed.all_gather_into_tensor(b["gathered"], b["local"], group=self.group)
        self.backend.merge(b["gathered"], self.world, nq, k, b["keys"], b["cos"], b["ids"], b["counts"])
        return b
