#!/bin/bash
# primed vs unprimed streaming scan at 1M rows (bench.py's config_1m leg) and query latency
set -e
out=gpurun_out/prime_small.log
: > $out
for cfg in "0 16384" "1 16384" "1 8192" "1 4096"; do
  set -- $cfg
  for k in 10 200; do
    echo "== k=$k prime_min_k=$1 rows=$2" >> $out
    CS_SCAN_PRIME_MIN_K=$1 CS_SCAN_PRIME_ROWS=$2 CS_SCAN_PRIME_MIN_ROWS=100000 \
      python3 bench.py --k $k --rows 2000000 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d.get('config_1m'))" >> $out
  done
done
cat $out
