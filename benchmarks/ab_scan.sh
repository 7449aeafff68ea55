#!/bin/bash
# A/B of two builds of libcsgpu.so on the single-query scan: usage ab_scan.sh A.so B.so "k" ...
# (the build under test is selected with CS_LIBCSGPU; the in-tree library is never overwritten)
a=$1; b=$2; shift 2
for k in "$@"; do
  for rep in 1 2; do
    for v in $a $b; do
      CS_LIBCSGPU=$(realpath $v) python3 bench.py --k $k --steps 60 --warmup 8 --no-cpu-baseline --no-encoder --no-e2e 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('k=$k $v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['config_1m']['scan_kernel_us'],1))"
    done
  done
done
