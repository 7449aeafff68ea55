#!/bin/bash
# The 256 x 192 block of the wide split-f16 GEMM (GwGeom<2, 4>: 0.875x / 0.70x the operand bytes per flop of the 128 x 384 / two
# 128 x 192 blocks) against the default shapes, diagnostic library, same box: (1) parity of the GEMM tests under the shape,
# (2) per-layer-shape kernel times, (3) the encoder forward with per-stage times, alternating.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
D=$R/codesearch_amd/libcsgpu_diag.so
echo "# (1) parity with CS_GEMM_WIDE_SHAPE=256"
CS_GEMM_WIDE_SHAPE=256 timeout -k 10 600 python3 -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu 2>&1 | tail -3
echo "# (2) kernel times (us): shape 0 = default"
python3 - <<'PY'
import ctypes as C
from codesearch_amd import _lib
lib = _lib.load_diag()
def t(epi, M, N, K, abl):
    ms = C.c_double()
    _lib.check_diag(lib.cs_debug_gemm_time(0, 2, epi, M, N, K, 20, abl, C.byref(ms)))
    return ms.value * 1e3
for name, epi, N, K in (("QKV", 4, 1152, 384), ("FFN-up", 1, 1536, 384), ("out f32+resid", 2, 384, 384), ("FFN-down f32+resid", 2, 384, 1536)):
    for rep in range(2):
        print(name, {s: round(t(epi, 65536, N, K, s), 1) for s in (0, 192, 384, 256)})
PY
echo "# (3) encoder forward, BGE-small shape 256 x 256, per-stage us per layer"
for rep in 1 2 3; do for sh in 0 256; do CS_LIBCSGPU=$D CS_GEMM_WIDE_SHAPE=$sh python3 benchmarks/encoder_bench.py --iters 10 --stages 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_us_per_layer']; print('shape $sh', round(d['device_ms_per_batch'],3), s['qkv_gemm'], s['ffn_up_gemm'], s['attention'], s['out_proj_gemm'], s['ffn_down_gemm'])"; done; done
