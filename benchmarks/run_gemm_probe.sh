#!/bin/bash
# A/B of the split-f16 GEMM main loops on the encoder's four shapes (benchmarks/gemm_probe.hip).
#   usage: benchmarks/run_gemm_probe.sh "128 5 256"
cd "$(dirname "$0")/.." || exit 1
for t in ${1:-128 5}; do
  for shape in "65536 1152 384 3" "65536 384 384 1" "65536 1536 384 2" "65536 384 1536 1"; do
    echo "== CS_GEMM_TILE=$t shape(M N K epi)=$shape"
    CS_GEMM_TILE=$t timeout -k 10 120 ./benchmarks/gemm_probe $shape || exit 1
  done
done
