#!/bin/bash
# Round 4: the geometric phase plan with a 3,072-row phase 0 (default) against round 3's fixed growth
# (CS_FILTER_GROWTH1=16 / CS_FILTER_GROWTH: the old rule, same library), ms per search over 10M x 384, device API.
run() { python3 bench.py --route cost --nq $1 --k $2 --steps 40 --warmup 5 --only-scan 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for cfg in "1 10" "1 25" "1 200" "2 10" "8 10" "9 200" "32 10" "64 10" "64 200" "128 10" "1000 10"; do
  set -- $cfg
  if [ $2 -ge 48 ]; then old="CS_FILTER_GROWTH=4"; else old="CS_FILTER_GROWTH1=16"; fi
  echo "nq=$1 k=$2  geometric $(run $1 $2) $(run $1 $2)  r03-growth $(export $old; run $1 $2) $(export $old; run $1 $2)"
done
