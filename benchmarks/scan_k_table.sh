#!/bin/bash
# One query over 10M x 384 rows at several k (and what kernel the search selected): ms per search, span of the
# scan-side kernels, fraction of the 8 TB/s HBM peak the f32 matrix was read at, merge time.
for k in "$@"; do
  python3 bench.py --k $k --steps 30 --warmup 5 --no-cpu-baseline --no-encoder --no-e2e --no-1m 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('k=$k ms_per_search %.4f kernel %s span_us %.1f frac %.4f merge_us %s' % (d['ms_per_step'], r.get('kernel'), r['avg_launch_us'], r['frac'], r.get('merge_avg_us')))"
done
