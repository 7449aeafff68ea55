#!/bin/bash
# One query, top-k: the f32 streaming scan vs the filter + refine route by corpus size (ms per search, device API).
run() { python3 bench.py --only-scan --rows $1 --k $2 --route $3 --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
mkdir -p gpurun_out/r04
for rows in 50000 100000 200000 400000 1000000 2000000 4000000; do
  for k in 10 25; do
    echo "rows=$rows k=$k  stream $(run $rows $k stream) $(run $rows $k stream) us   filter $(run $rows $k filter) $(run $rows $k filter) us"
  done
done
