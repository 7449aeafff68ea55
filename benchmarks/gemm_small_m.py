import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from codesearch_amd import _lib
lib = _lib.load_diag()  # cs_debug_*: libcsgpu_diag.so (include/codesearch_gpu_diag.h)
def t(mode, epi, M, N, K, abl=0, iters=50):
    ms = C.c_double()
    _lib.check_diag(lib.cs_debug_gemm_time(0, mode, epi, M, N, K, iters, abl, C.byref(ms)))
    return ms.value * 1e3
for M in (2048, 4096, 8192, 16384):
    print(f"M={M}: out-proj 128x128+resid {t(1,2,M,384,384):6.1f} us | wide LN-fused {t(2,3,M,384,384):6.1f} us || ffn-down 128x128+resid {t(1,2,M,384,1536):6.1f} | wide LN-fused {t(2,3,M,384,1536):6.1f} || qkv 128x128 {t(1,4,M,1152,384):6.1f} wide384 {t(2,4,M,1152,384):6.1f} || ffn-up 128x128 {t(1,1,M,1536,384):6.1f} wide {t(2,1,M,1536,384):6.1f}", flush=True)
