#!/usr/bin/env python3
"""Device time of the encoder's dense layers, one kernel at a time, through cs_debug_gemm_time (operands resident
in HBM): the 128 x 128 split kernels vs the persistent 128 x 384 one-accumulator kernel, and the wide kernel's
ablation builds (no LDS-DMA / no MFMA / DMAs at the step start)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from codesearch_amd import _lib

    lib = _lib.load_diag()  # cs_debug_*: libcsgpu_diag.so (include/codesearch_gpu_diag.h)
    M = int(os.environ.get("M", 65536))

    def t(mode, epi, N, K, abl=0, iters=20):
        ms = C.c_double()
        _lib.check_diag(lib.cs_debug_gemm_time(0, mode, epi, M, N, K, iters, abl, C.byref(ms)))
        return ms.value * 1e3

    shapes = [("qkv      (N=1152, K=384,  bias -> split)", 4, 1152, 384), ("ffn_up   (N=1536, K=384,  GELU -> split)", 1, 1536, 384),
              ("out_proj (N=384,  K=384,  + resid f32)", 2, 384, 384), ("ffn_down (N=384,  K=1536, + resid f32)", 2, 384, 1536)]
    shape = os.environ.get("CS_GEMM_WIDE_SHAPE", "384; QKV 192")
    for name, epi, N, K in shapes:
        tf = 3 * 2.0 * M * N * K / 1e12
        a, b = t(1, epi, N, K), t(2, epi, N, K)
        print(f"{name}: 128x128 {a:7.1f} us ({tf / a * 1e6:6.0f} TF/s executed)   wide(128x{shape}) {b:7.1f} us ({tf / b * 1e6:6.0f} TF/s)", flush=True)
    if os.environ.get("GEMM_TIME_SHAPES_ONLY"):
        return
    for K in (384, 1536):
        print(f"N=384 K={K} LayerNorm-fused wide: {t(2, 3, 384, K):7.1f} us", flush=True)
    for name, abl in (("full", 0), ("no LDS-DMA", 1), ("no MFMA", 2), ("DMAs at step start", 3), ("no DMA, no barrier", 4),
                      ("no DMA/barrier/LDS reads", 5), ("no DMA, no epilogue", 6), ("full, no global stores", 8),
                      ("full, default-policy stores", 9), ("full, line-ordered stores (timing only)", 10)):
        print(f"wide qkv ablation {name:20s}: {t(2, 4, 1152, 384, abl):7.1f} us", flush=True)
    # stamped build (s_memtime / s_memrealtime per block) after > 2 s of back-to-back launches: the clock the chip
    # holds under this kernel (the library prints it on stderr)
    print(f"wide qkv stamped build, 12000 launches: {t(2, 4, 1152, 384, 7, iters=12000):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
