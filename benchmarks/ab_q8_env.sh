#!/bin/bash
# Same-box A/B of one environment knob on the quantised default model (MiniLM-L6-Q shape, 256 x 256 tokens): per-stage microseconds
# per layer and the forward, three alternating pairs; then the 12-layer shape and the quantised parity tests under both values.
#   usage: ab_q8_env.sh NAME "v0 v1"
R=${GRAFT_REPO_ROOT:-$PWD}
name=$1; vals=${2:-"0 1"}
for rep in 1 2 3; do
for t in $vals; do
  echo "== $name=$t ($rep)"
  env $name=$t python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 10 --stages 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['device_ms_per_batch'],3), d['stages_us_per_layer'])"
done
done
echo "== 12-layer BGE-small-Q shape"
for t in $vals; do
  env $name=$t python3 $R/benchmarks/encoder_bench.py --model bge-small-q --quant u8 --iters 10 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name=$t', round(d['device_ms_per_batch'],3))"
done
for t in $vals; do
  echo "== tests, $name=$t"
  env $name=$t python3 -m pytest $R/tests/test_gpu_quantized.py -q -x 2>&1 | tail -3
done
