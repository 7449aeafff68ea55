import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from codesearch_amd import _lib
lib = _lib.load_diag()  # cs_debug_*: libcsgpu_diag.so (include/codesearch_gpu_diag.h)
def t(mode, epi, N, K, abl=0, iters=20, M=65536):
    ms = C.c_double()
    _lib.check_diag(lib.cs_debug_gemm_time(0, mode, epi, M, N, K, iters, abl, C.byref(ms)))
    return ms.value * 1e3
for name, N, K in (("ffn_down shape N=384 K=1536", 384, 1536), ("out_proj shape N=384 K=384", 384, 384), ("ffn_up shape N=1536 K=384 (split epilogue, no GELU)", 1536, 384)):
    print(name, {a: round(t(2, 4, N, K, a), 1) for a in (0, 1, 2, 5, 6)}, flush=True)
