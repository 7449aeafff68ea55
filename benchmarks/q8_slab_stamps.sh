#!/bin/bash
# In-kernel clock stamps of gemm_q8_slab_kernel (variants built with -DCS_Q8_STAMPS): one forward, the last layer's lines.
#   usage: q8_slab_stamps.sh "stamps stamps_pf4"
R=${GRAFT_REPO_ROOT:-$PWD}
for name in $1; do
  echo "== $name"
  CS_LIBCSGPU=$R/codesearch_amd/variants/libcsgpu_$name.so CS_ENCODER_STREAMS=1 python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 1 2>&1 | grep "q8 slab\|q8 ln" | tail -12
done
