#!/bin/bash
# Round 4: the role-split wide GEMM (loader waves / storing waves): parity, then A/B against the product kernel.
set -e -o pipefail
out=gpurun_out/r04
mkdir -p $out
CS_GEMM_WIDE_ROLES=1 python -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu > $out/gemm_tests_roles.log 2>&1 || { tail -40 $out/gemm_tests_roles.log; exit 1; }
tail -1 $out/gemm_tests_roles.log
SHAPES=qkv,ffn_up CODES=100,120,1000 ROUNDS=7 python benchmarks/gemm_sched_ab.py 2>/dev/null | tee $out/roles_ab.log
for rep in 1 2 3; do for r in 0 1; do CS_GEMM_WIDE_ROLES=$r python3 benchmarks/encoder_bench.py --iters 10 --stages 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_us_per_layer']; print('roles $r', round(d['device_ms_per_batch'],3), s['qkv_gemm'], s['ffn_up_gemm'], s['attention'], s['out_proj_gemm'], s['ffn_down_gemm'])"; done; done | tee $out/roles_encoder_ab.log
CS_GEMM_WIDE_ROLES=1 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu > $out/encoder_tests_roles.log 2>&1 || { tail -40 $out/encoder_tests_roles.log; exit 1; }
tail -1 $out/encoder_tests_roles.log
