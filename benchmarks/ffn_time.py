#!/usr/bin/env python3
"""Device time of the encoder's feed-forward block (E5 + E6) at the BGE-small shape, 65,536 token rows: the fused
kernel (ffn_fused.hip) next to the two wide kernels it replaces, plus the fused kernel's ablation builds.
cs_debug_ffn_time: synthetic operands resident in HBM, HIP events around `iters` back-to-back launches."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codesearch_amd import _lib

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
I = int(sys.argv[2]) if len(sys.argv) > 2 else 1536


def t(fused, abl=0, iters=30):
    ms = C.c_double()
    _lib.check(lib.cs_debug_ffn_time(0, fused, M, I, iters, abl, C.byref(ms)))
    return ms.value * 1e3


for rep in range(3):
    print(f"M={M} I={I}: two kernels {t(0):7.1f} us | fused {t(1):7.1f} us | fused, no GELU arithmetic {t(1, 1):7.1f} us | "
          f"fused, no LDS-DMA {t(1, 2):7.1f} us", flush=True)
