#!/bin/bash
# FFN-up store pass of the quantised forward: the output byte by table lookup (default) against the direct form
# (CS_Q8_GELU_TABLE=0), same box, alternating; then the parity tests of the quantised path under both.
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2 3; do
for t in 1 0; do
  echo "== CS_Q8_GELU_TABLE=$t ($rep)"
  CS_Q8_GELU_TABLE=$t python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 10 --stages 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['device_ms_per_batch'],3), d['stages_us_per_layer'])"
done
done
for t in 1 0; do
  echo "== tests, CS_Q8_GELU_TABLE=$t"
  CS_Q8_GELU_TABLE=$t python3 -m pytest $R/tests/test_gpu_quantized.py -q -x 2>&1 | tail -3
done
