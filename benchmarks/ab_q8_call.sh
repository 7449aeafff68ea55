#!/bin/bash
# The reference's call shape on the quantised default model for library variants on ONE box: 32 x 256 tokens one call
# at a time and eight calls queued (benchmarks/q8_queue_rate.py), then the 256 x 256 one-unit forward.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/abq8
for rep in 1 2; do
for v in "$@"; do
  lib=$R/codesearch_amd/libcsgpu.so
  [ "$v" != base ] && lib=$R/codesearch_amd/variants/libcsgpu_$v.so
  CS_LIBCSGPU=$lib python3 $R/benchmarks/q8_queue_rate.py > $R/gpurun_out/abq8/call_$v.$rep.log 2>&1
  CS_LIBCSGPU=$lib python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 10 > $R/gpurun_out/abq8/fwd_$v.$rep.log 2>&1
  echo "== $v ($rep)"; tail -1 $R/gpurun_out/abq8/call_$v.$rep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['q8']; print({k: round(v, 1) for k, v in d.items()})"
  tail -1 $R/gpurun_out/abq8/fwd_$v.$rep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('256 x 256 device ms', round(d['device_ms_per_batch'],3))"
done
done
