run() { CS_FILTER_FEW_MIN_ROWS=$4 python3 bench.py --only-scan --rows $1 --nq $2 --k $3 --steps 300 --warmup 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
for cfg in "4000 3 10" "4000 4 10" "5000 3 10" "5000 3 25" "10000 3 10" "10000 3 25" "20000 3 25" "4000 2 10" "4000 2 25"; do
  set -- $cfg
  echo "rows=$1 nq=$2 k=$3 :  stream(forced for two; nq>=3 by CS_FILTER_MIN_Q) $(CS_FILTER_MIN_Q=5 run $1 $2 $3 1000000000) $(CS_FILTER_MIN_Q=5 run $1 $2 $3 1000000000)   filter $(run $1 $2 $3 0) $(run $1 $2 $3 0)"
done
