#!/bin/bash
# A/B of two builds of libcsgpu.so on one box: usage ab_filter.sh A.so B.so "nq k" ...
# (the build under test is selected with CS_LIBCSGPU; the in-tree library is never overwritten)
a=$1; b=$2; shift 2
for cfg in "$@"; do
  set -- $cfg
  for rep in 1 2; do
    for v in $a $b; do
      ms=$(CS_LIBCSGPU=$(realpath $v) python3 bench.py --nq $1 --k $2 --steps 40 --warmup 5 --only-scan 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")
      echo "nq=$1 k=$2 $v $ms"
    done
  done
done
