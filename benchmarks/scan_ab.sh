#!/bin/bash
# Two builds of libcsgpu.so on the streamed single-query search, same box, alternating: configs[1] (1M x 384 rows) and the
# north-star search (10M rows, bench.py --only-scan).   usage: scan_ab.sh A.so B.so [reps]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
one_m() {
  CS_LIBCSGPU=$1 python3 -c "
import json, bench
from codesearch_amd import VectorStore
d = bench.scan_1m_leg(384, 10, 0, VectorStore)
print(json.dumps({k: round(d[k], 4) for k in ('ms_per_search', 'scan_kernel_us', 'frac_of_hbm_peak')}))
" 2>&1 | tail -1
}
ten_m() {
  CS_LIBCSGPU=$1 python3 bench.py --only-scan --steps 100 --warmup 10 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['roofline'].get('avg_launch_us'),1))"
}
for rep in $(seq 1 ${3:-3}); do
  for v in $1 $2; do
    echo "$v 1M: $(one_m $(realpath $v))"
    echo "$v 10M: $(ten_m $(realpath $v))"
  done
done
