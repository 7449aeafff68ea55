#!/bin/bash
# end-of-session evidence set (profiles/r01e_*): run from the repo root on the GPU box
set -e
R=$PWD
O=$R/gpurun_out/r01e
mkdir -p $O
python bench.py > $O/bench_full_run.json 2> $O/bench_full_run.err
echo "bench done" 
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/scan_q1 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-encoder > $O/scan_q1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/filter_q8 -- python3 $R/bench.py --nq 8 --steps 50 --warmup 5 --no-cpu-baseline --no-encoder > $O/filter_q8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/filter_q9_k200 -- python3 $R/bench.py --nq 9 --k 200 --steps 50 --warmup 5 --no-cpu-baseline --no-encoder > $O/filter_q9_k200.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/scan_q1_k200 -- python3 $R/bench.py --k 200 --steps 50 --warmup 5 --no-cpu-baseline --no-encoder > $O/scan_q1_k200.log 2>&1
echo "profiles done"
cd $R
python3 benchmarks/phase_timeline.py $O/filter_q9_k200 22 > $O/filter_q9_k200_timeline.log
python3 benchmarks/phase_timeline.py $O/scan_q1_k200 4 > $O/scan_q1_k200_timeline.log
: > $O/search_nq_k_table.log
for k in 10 200; do for nq in 1 2 8 9 32 64 128 256 1000; do
  ms=$(python3 bench.py --nq $nq --k $k --steps 20 --warmup 3 --no-cpu-baseline --no-encoder 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")
  echo "queries=$nq k=$k ms_per_search=$ms" >> $O/search_nq_k_table.log
done; done
cat $O/search_nq_k_table.log
