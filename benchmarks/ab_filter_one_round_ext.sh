run() { CS_FILTER_G1_MAXQ=$4 CS_FILTER_G1_CAND=$5 python3 bench.py --only-scan --rows $1 --nq $2 --k $3 --route filter --steps 300 --warmup 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))"; }
for cfg in "100000 8 10" "184000 8 10" "50000 8 10" "100000 9 10" "100000 6 16" "100000 1 25" "110000 1 25" "100000 1 20" "100000 4 25" "150000 1 20"; do
  set -- $cfg
  echo "rows=$1 nq=$2 k=$3 :  base(q<=4,k<=10@60) $(run $1 $2 $3 4 600) $(run $1 $2 $3 4 600)   q<=10 $(run $1 $2 $3 10 600) $(run $1 $2 $3 10 600)   q<=10,cand970 $(run $1 $2 $3 10 970) $(run $1 $2 $3 10 970)"
done
