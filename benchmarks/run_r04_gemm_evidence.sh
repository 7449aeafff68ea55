#!/bin/bash
# Round 4 evidence for the wide GEMM as shipped: parity, shape A/B, zero-vs-random operands (DVFS), ablations.
set -e -o pipefail
out=gpurun_out/r04
mkdir -p $out
python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_encoder.py -x -q -m gpu > $out/gemm_encoder_tests.log 2>&1 || { tail -40 $out/gemm_encoder_tests.log; exit 1; }
tail -1 $out/gemm_encoder_tests.log
SHAPES=qkv,ffn_up ROUNDS=7 python benchmarks/gemm_shape_ab.py 2>/dev/null | tee $out/gemm_shape_ab.log
bash benchmarks/gemm_zero_vs_random.sh | tee $out/gemm_zero_vs_random.log
python3 benchmarks/encoder_bench.py --iters 10 --stages 2>/dev/null | tail -1 | tee $out/encoder_stages.json
