#!/bin/bash
# k sweep of the streaming scan with and without the primed threshold (one box, same corpus size)
set -e
out=gpurun_out/prime_sweep.log
: > $out
for k in 10 64 200 256; do
  for cfg in "0 32768" "48 32768" "48 16384" "48 65536" "48 131072"; do
    set -- $cfg
    if [ "$1" = "48" ] && [ $k -lt 48 ]; then mk=1; else mk=$1; fi
    echo "== k=$k prime_min_k=$mk rows=$2" >> $out
    CS_SCAN_PRIME_MIN_K=$mk CS_SCAN_PRIME_ROWS=$2 \
      python3 bench.py --k $k --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'])" >> $out
  done
done
cat $out
