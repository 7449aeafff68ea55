#!/bin/bash
# One query, k above the short lists: the f32 streaming scan vs the filter + refine route by corpus size (us per search,
# device API).  Arguments: rows list, k list (defaults below).
run() { python3 bench.py --only-scan --rows $1 --k $2 --route $3 --steps 300 --warmup 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
ROWS=${1:-"35000 50000 75000 100000 150000"}
KS=${2:-"25 50 75"}
for rows in $ROWS; do
  for k in $KS; do
    echo "rows=$rows k=$k :  stream $(run $rows $k stream) $(run $rows $k stream)   filter $(run $rows $k filter) $(run $rows $k filter)"
  done
done
