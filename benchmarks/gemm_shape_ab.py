#!/usr/bin/env python3
"""A/B of the wide GEMM's block shapes (128 x 384, one block per CU | 128 x 192, two per CU) on the encoder's dense-layer
shapes: variants interleaved over rounds in ONE process (cdna_hip_programming.md rule 24), median and min per variant.
(Round 4 also ran DMA schedules, a store-grace wait, a CU-paired stagger and a role-split kernel through this script:
commit "Wide GEMM experiments", logs profiles/r04_gemm_*_ab.log.)"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from codesearch_amd import _lib

    lib = _lib.load_diag()  # cs_debug_*: libcsgpu_diag.so (include/codesearch_gpu_diag.h)
    M = int(os.environ.get("M", 65536))
    rounds = int(os.environ.get("ROUNDS", 5))
    iters = int(os.environ.get("ITERS", 30))
    # codes of cs_debug_gemm_time's `ablation`: 384 / 192 = that block shape of the product kernel
    scheds = [int(x) for x in os.environ.get("CODES", "384,192").split(",")]

    def t(epi, N, K, code):
        ms = C.c_double()
        _lib.check_diag(lib.cs_debug_gemm_time(0, 2, epi, M, N, K, iters, code, C.byref(ms)))
        return ms.value * 1e3

    shapes = [("qkv      N=1152 K=384  bias->split", 4, 1152, 384), ("ffn_up   N=1536 K=384  GELU->split", 1, 1536, 384),
              ("out+LN   N=384  K=384  LayerNorm  ", 3, 384, 384), ("down+LN  N=384  K=1536 LayerNorm  ", 3, 384, 1536)]
    only = os.environ.get("SHAPES")
    for name, epi, N, K in shapes:
        if only and name.split()[0] not in only.split(","):
            continue
        res = {s: [] for s in scheds}
        for _ in range(rounds):
            for s in scheds:
                res[s].append(t(epi, N, K, s))
        line = "  ".join(f"{s}: {statistics.median(v):6.1f} (min {min(v):6.1f})" for s, v in res.items())
        print(f"{name}: {line}", flush=True)


if __name__ == "__main__":
    main()
