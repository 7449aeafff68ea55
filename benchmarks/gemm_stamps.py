#!/usr/bin/env python3
"""In-kernel clock and the k-loop / epilogue split of the wide GEMM's QKV shape from the stamped build (ablation 7):
the library prints the figures on stderr."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codesearch_amd import _lib

lib = _lib.load_diag()  # cs_debug_*: libcsgpu_diag.so (include/codesearch_gpu_diag.h)
ms = C.c_double()
for iters in (200, int(os.environ.get("ITERS", 12000))):
    _lib.check_diag(lib.cs_debug_gemm_time(0, 2, 4, 65536, 1152, 384, iters, 7, C.byref(ms)))
    print(f"stamped build, {iters} launches: {ms.value * 1e3:.1f} us per launch", flush=True)
