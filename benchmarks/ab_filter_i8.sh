#!/bin/bash
# int8 filter A/B on one box: ring depth variants of the 33..128-query kernels (CS_LIBCSGPU)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --nq $1 --k $2 --steps 40 --warmup 5 --only-scan 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for cfg in "$@"; do
  set -- $cfg
  for rep in 1 2; do
    echo "nq=$1 k=$2 int8 $(run $1 $2)"
    for v in q3 q4 q6; do echo "nq=$1 k=$2 $v $(CS_LIBCSGPU=$R/codesearch_amd/variants/libcsgpu_$v.so run $1 $2)"; done
  done
done
