#!/bin/bash
# Round 4, wide GEMM, second batch: two blocks per CU with the CU's second block late (paired by HW_ID), parity of the
# index changes (lazy f16 copy), encoder/index tests.
set -e -o pipefail
out=gpurun_out/r04
mkdir -p $out
SHAPES=qkv,ffn_up CODES=100,1000,1400,1800,2200,2600,3000 python benchmarks/gemm_sched_ab.py 2>/dev/null | tee $out/stagger_by_cu_ab.log
python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_filter_int8.py tests/test_gpu_indexing.py tests/test_gpu_scan.py -x -q -m gpu > $out/tests_gemm_index.log 2>&1 || { tail -40 $out/tests_gemm_index.log; exit 1; }
tail -2 $out/tests_gemm_index.log
