#!/bin/bash
# Same-box A/B of several library variants (codesearch_amd/variants/libcsgpu_<name>.so, benchmarks/build_variant.sh; "base" = the
# in-tree library) on the quantised default model (MiniLM-L6-Q shape, 256 x 256 tokens): per-stage microseconds per layer and the
# forward, REPS alternating rounds.   usage: ab_q8_libs.sh "base pf4 pf8" [reps]
R=${GRAFT_REPO_ROOT:-$PWD}
reps=${2:-2}
for rep in $(seq 1 $reps); do
for name in $1; do
  v=$R/codesearch_amd/variants/libcsgpu_$name.so
  [ "$name" == "base" ] && v=$R/codesearch_amd/libcsgpu.so
  echo "== $name ($rep)"
  CS_LIBCSGPU=$v python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 10 --stages 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['device_ms_per_batch'],3), d['stages_us_per_layer'])"
done
done
