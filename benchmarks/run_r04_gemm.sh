#!/bin/bash
# Round 4, wide GEMM: parity of the new epilogue parameter path and of the DMA schedules, schedule A/B, encoder A/B
# against the round-3 library (codesearch_amd/variants/libcsgpu_r03.so, built from the r03 commit).
set -e -o pipefail
out=gpurun_out/r04
mkdir -p $out
python -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu > $out/gemm_tests_default.log 2>&1 || { tail -30 $out/gemm_tests_default.log; exit 1; }
tail -2 $out/gemm_tests_default.log
for s in 2 12; do
  CS_GEMM_WIDE_SCH=$s python -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu -k "wide" > $out/gemm_tests_sch$s.log 2>&1 || { tail -30 $out/gemm_tests_sch$s.log; exit 1; }
  tail -1 $out/gemm_tests_sch$s.log
done
SCHEDS=${SCHEDS:-0,1,2,3,4,10,12} python benchmarks/gemm_sched_ab.py 2>/dev/null | tee $out/sched_ab.log
python benchmarks/gemm_time.py 2>&1 | grep -v amdgpu.ids | tee $out/gemm_time.log
for rep in 1 2; do
  for v in codesearch_amd/variants/libcsgpu_r03.so codesearch_amd/libcsgpu.so; do
    for pool in "" "--model minilm-l6"; do
      CS_LIBCSGPU=$(realpath $v) python3 benchmarks/encoder_bench.py --iters 10 --stages $pool 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$pool', round(d['device_ms_per_batch'],3), d.get('stages_us_per_layer', d.get('stages')))"
    done
  done
done | tee $out/encoder_ab.log
