#!/bin/bash
# One query over a mid-size corpus: the streaming scan, the filter route with the round plan capped at 24 (CS_FILTER_G1MAX=1 =
# before) and with ONE round up to 60 x 3,072 rows (default); us per search, device API.
run() { CS_FILTER_G1MAX=$4 python3 bench.py --only-scan --rows $1 --k $2 --route $3 --steps 300 --warmup 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
for rows in 20000 35000 50000 75000 100000 150000 184000 200000; do
  for k in 10 16; do
    echo "rows=$rows k=$k :  stream $(run $rows $k stream 60) $(run $rows $k stream 60)   filter/two rounds $(run $rows $k filter 1) $(run $rows $k filter 1)   filter/one round $(run $rows $k filter 60) $(run $rows $k filter 60)"
  done
done
