#!/bin/bash
# Phase 0 of a few-query filter search: 96 blocks per query (one round of 32 rows per block, the default up to ten queries)
# against the refine's 32 (CS_FILTER_PHASE0_BLOCKS=32 = before); us per search, device API, alternating.
# Arguments (optional): "rows nq k" triples.
run() { CS_FILTER_PHASE0_BLOCKS=$4 python3 bench.py --only-scan --rows $1 --nq $2 --k $3 --route filter --steps 300 --warmup 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
cfgs=("$@")
[ ${#cfgs[@]} -eq 0 ] && cfgs=("100000 1 10" "400000 1 10" "1000000 1 10" "1000000 1 25" "10000000 1 10" "1000000 8 10" "10000000 8 10" "100000 9 200")
for cfg in "${cfgs[@]}"; do
  set -- $cfg
  echo "rows=$1 nq=$2 k=$3 :  blocks32 $(run $1 $2 $3 32) $(run $1 $2 $3 32)   blocks96 $(run $1 $2 $3 96) $(run $1 $2 $3 96)   blocks32 $(run $1 $2 $3 32)   blocks96 $(run $1 $2 $3 96)"
done
