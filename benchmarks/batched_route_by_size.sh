#!/bin/bash
# Batched searches (the reference's hybrid search: 9 variants x retrieval limit 200; and 8 x 10) by corpus size: the filter
# path (default) against the exact-f32 MFMA batched path (CS_INDEX_SPLIT=0), us per search, device API.
run() { python3 bench.py --only-scan --rows $1 --nq $2 --k $3 --steps 100 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
for rows in 2000 5000 20000 100000 400000 1000000; do
  for cfg in "9 200" "8 10" "2 25"; do
    set -- $cfg
    echo "rows=$rows nq=$1 k=$2  filter $(run $rows $1 $2) $(run $rows $1 $2)   exact-f32-mfma $(CS_INDEX_SPLIT=0 run $rows $1 $2) $(CS_INDEX_SPLIT=0 run $rows $1 $2)"
  done
done
