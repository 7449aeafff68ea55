#!/bin/bash
# The quantised MiniLM-L6 forward (256 x 256, one unit) with per-stage times for library variants on ONE box:
#   ab_q8_lib.sh name...      ("base" = the in-tree library; others: codesearch_amd/variants/libcsgpu_<name>.so)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/abq8
for rep in 1 2; do
for v in "$@"; do
  lib=$R/codesearch_amd/libcsgpu.so
  [ "$v" != base ] && lib=$R/codesearch_amd/variants/libcsgpu_$v.so
  CS_LIBCSGPU=$lib python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 10 --stages > $R/gpurun_out/abq8/$v.$rep.log 2>&1
  echo "== $v ($rep)"; tail -1 $R/gpurun_out/abq8/$v.$rep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['device_ms_per_batch'],3), d['stages_us_per_layer'])"
done
done
