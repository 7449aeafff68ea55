#!/bin/bash
# Same-box A/B of two libraries on the quantised default model (MiniLM-L6-Q shape, 256 x 256 tokens): per-stage microseconds per layer
# and the forward, three alternating pairs; then the quantised parity tests on the in-tree library.
#   usage: ab_q8_lib2.sh <variant name under codesearch_amd/variants>
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2 3; do
for v in $R/codesearch_amd/variants/libcsgpu_$1.so $R/codesearch_amd/libcsgpu.so; do
  echo "== $(basename $v) ($rep)"
  CS_LIBCSGPU=$v python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 10 --stages 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['device_ms_per_batch'],3), d['stages_us_per_layer'])"
done
done
echo "== 12-layer BGE-small-Q shape"
for v in $R/codesearch_amd/variants/libcsgpu_$1.so $R/codesearch_amd/libcsgpu.so; do
  CS_LIBCSGPU=$v python3 $R/benchmarks/encoder_bench.py --model bge-small-q --quant u8 --iters 10 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $v)', round(d['device_ms_per_batch'],3))"
done
python3 -m pytest $R/tests/test_gpu_quantized.py -q -x 2>&1 | tail -3
