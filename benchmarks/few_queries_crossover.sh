#!/bin/bash
# Two to four queries over a small corpus: the streaming scan (CS_FILTER_FEW_MIN_ROWS=1000000000) against the filter route
# (=0); us per search, device API.
run() { CS_FILTER_FEW_MIN_ROWS=$4 python3 bench.py --only-scan --rows $1 --nq $2 --k $3 --steps 300 --warmup 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
for rows in 5000 10000 20000 35000 50000; do
  for cfg in "2 10" "2 25" "4 10" "4 25"; do
    set -- $cfg
    echo "rows=$rows nq=$1 k=$2 :  stream $(run $rows $1 $2 1000000000) $(run $rows $1 $2 1000000000)   filter $(run $rows $1 $2 0) $(run $rows $1 $2 0)"
  done
done
