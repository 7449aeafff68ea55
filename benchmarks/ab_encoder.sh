#!/bin/bash
# A/B of two builds of libcsgpu.so on the encoder forward: usage ab_encoder.sh A.so B.so [encoder_bench args]
# (the build under test is selected with CS_LIBCSGPU; the in-tree library is never overwritten)
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  for v in $a $b; do
    CS_LIBCSGPU=$(realpath $v) python3 benchmarks/encoder_bench.py --iters 10 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['device_ms_per_batch'],3))"
  done
done
