#!/bin/bash
# Round-3 evidence for the int8 filter copy (profiles/r03_int8_*): run from the repo root on the GPU box; results land
# in gpurun_out/r03i8/.  A/B table of the filter copies, kernel stats of three query shapes, FETCH_SIZE of the 8-query case.
set -e
R=$PWD
O=$R/gpurun_out/r03i8
mkdir -p $O
run() { python3 $R/bench.py --nq $1 --k $2 --steps 40 --warmup 5 --only-scan 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
{
  echo "# ms per search over 10M x 384 (bench.py --only-scan, device API): int8 copy (default) | f16 copy (CS_FILTER_INT8=0)"
  for cfg in "1 200" "2 10" "8 10" "9 200" "32 10" "64 10" "64 200" "128 10" "256 10" "1000 10"; do
    set -- $cfg
    echo "nq=$1 k=$2  int8 $(run $1 $2) $(run $1 $2)  f16 $(CS_FILTER_INT8=0 run $1 $2) $(CS_FILTER_INT8=0 run $1 $2)"
  done
} > $O/filter_copy_ab.txt 2>&1
echo "ab done"
python3 $R/bench.py --nq 8 --k 10 --steps 100 --warmup 10 --only-scan > $O/bench_q8_k10.json 2>/dev/null
python3 $R/bench.py --nq 9 --k 200 --steps 100 --warmup 10 --only-scan > $O/bench_q9_k200.json 2>/dev/null
python3 $R/bench.py --nq 1000 --k 10 --steps 20 --warmup 3 --only-scan > $O/bench_q1000_k10.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
stats() {  # name, then the command after `--`
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  cp "$(find $O/$name -name '*kernel_stats.csv' | head -1)" $O/${name}_kernel_stats.csv
  python3 $R/benchmarks/phase_timeline.py $O/$name 24 > $O/${name}_timeline.txt 2>&1 || true
  rm -rf $O/$name
}
stats filter_q8_k10 python3 $R/bench.py --only-scan --nq 8 --k 10 --steps 50 --warmup 5
stats filter_q9_k200 python3 $R/bench.py --only-scan --nq 9 --k 200 --steps 50 --warmup 5
stats filter_q1000_k10 python3 $R/bench.py --only-scan --nq 1000 --k 10 --steps 20 --warmup 3
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --only-scan --nq 8 --k 10 --steps 5 --warmup 1 > $O/pmc_fetch.log 2>&1
cp "$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)" $O/pmc_fetch_q8_counter_collection.csv
rm -rf $O/pmc_fetch
ls -la $O
