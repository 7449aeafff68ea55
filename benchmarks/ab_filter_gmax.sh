#!/bin/bash
# Largest ratio between consecutive phase boundaries of the geometric plan (CS_FILTER_GMAX) vs ms per search.
run() { python3 bench.py --route cost --rows $3 --nq $1 --k $2 --steps 60 --warmup 5 --only-scan 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
for cfg in "1 10 1000000" "1 25 1000000" "1 10 10000000" "8 10 10000000" "1 25 10000000" "32 10 10000000" "1 40 10000000"; do
  set -- $cfg
  line="nq=$1 k=$2 rows=$3 :"
  for g in 15 30 60 120; do line="$line  gmax$g $(CS_FILTER_GMAX=$g run $1 $2 $3) $(CS_FILTER_GMAX=$g run $1 $2 $3)"; done
  echo "$line"
done
for cfg in "1 200 10000000" "9 200 10000000" "1 100 1000000"; do
  set -- $cfg
  line="nq=$1 k=$2 rows=$3 :"
  for g in 5.5 8 12; do line="$line  gmax$g $(CS_FILTER_GMAX=$g run $1 $2 $3) $(CS_FILTER_GMAX=$g run $1 $2 $3)"; done
  echo "$line"
done
