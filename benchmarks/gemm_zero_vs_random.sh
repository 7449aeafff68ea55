#!/bin/bash
# The wide GEMM's four shapes on random and on all-zero operands, alternating: the same instruction stream and the same
# cycles; the difference is the clock the chip holds (MI355X_MICROARCH.md, DVFS give-back item 1).
for rep in 1 2 3; do
  for z in 0 1; do
    if [ $z == 1 ]; then export CS_DEBUG_GEMM_ZERO=1; else unset CS_DEBUG_GEMM_ZERO; fi
    echo "zero_operands=$z"; GEMM_TIME_SHAPES_ONLY=1 python benchmarks/gemm_time.py 2>/dev/null | sed 's/128x128.*wide/wide/'
  done
done
