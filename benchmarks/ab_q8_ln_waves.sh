#!/bin/bash
# gemm_q8_ln_kernel as two blocks of four waves per CU (64 rows, double-buffered weight stages; CS_Q8_LN_WAVES=4) against one block
# of eight (128 rows, four-stage ring; =8), same box, alternating; then the quantised parity tests under both.
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2 3; do
for t in 4 8; do
  echo "== CS_Q8_LN_WAVES=$t ($rep)"
  CS_Q8_LN_WAVES=$t python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 10 --stages 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['device_ms_per_batch'],3), d['stages_us_per_layer'])"
done
done
echo "== 12-layer BGE-small-Q shape"
for t in 4 8; do
  CS_Q8_LN_WAVES=$t python3 $R/benchmarks/encoder_bench.py --model bge-small-q --quant u8 --iters 10 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves $t', round(d['device_ms_per_batch'],3))"
done
for t in 4 8; do
  echo "== tests, CS_Q8_LN_WAVES=$t"
  CS_Q8_LN_WAVES=$t python3 -m pytest $R/tests/test_gpu_quantized.py -q -x 2>&1 | tail -3
done
