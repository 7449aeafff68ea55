#!/bin/bash
# Block-life stamps of the 128 x 128 split-f16 GEMM on the encoder's four shapes (benchmarks/gemm_probe.hip).
cd "$(dirname "$0")/.." || exit 1
for shape in "65536 1152 384 3" "65536 384 384 1" "65536 1536 384 2" "65536 384 1536 1"; do
  echo "== shape(M N K epi)=$shape"
  timeout -k 10 120 ./benchmarks/gemm_probe $shape || exit 1
done
