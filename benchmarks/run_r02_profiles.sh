#!/bin/bash
# Round-2 evidence set (profiles/r02_*): run from the repo root on the GPU box; results land in gpurun_out/r02/.
# Every rocprofv3 pass profiles ONE kernel population: bench.py --only-scan runs the timed loop alone (no 1M leg, no
# filter leg, no encoder), so AverageNs of the scan row is the 10M-row figure.
set -e
R=$PWD
O=$R/gpurun_out/r02
mkdir -p $O
python bench.py > $O/bench_full_run.json 2> $O/bench_full_run.err
echo "bench done"
python benchmarks/gemm_time.py > $O/gemm_time.log 2>&1
cd /tmp && export TMPDIR=/tmp
stats() {  # name, then the command after `--`
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/${name}_kernel_stats.csv
}
stats scan_q1_only python3 $R/bench.py --only-scan --steps 100 --warmup 10
stats scan_q1_k200_only python3 $R/bench.py --only-scan --k 200 --steps 50 --warmup 5
stats filter_q9_k200_only python3 $R/bench.py --only-scan --nq 9 --k 200 --steps 50 --warmup 5
CS_ENCODER_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encoder_1stream -- python3 $R/benchmarks/encoder_bench.py --iters 10 > $O/encoder_1stream.log 2>&1
cp "$(find $O/encoder_1stream -name '*kernel_stats.csv' | head -1)" $O/encoder_1stream_kernel_stats.csv
echo "kernel traces done"
# counters in their own runs (no other trace domain)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --only-scan --steps 5 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --only-scan --steps 5 --warmup 1 > $O/pmc_write.log 2>&1
CS_ENCODER_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_enc -- python3 $R/benchmarks/encoder_bench.py --iters 2 > $O/pmc_enc.log 2>&1
cp "$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)" $O/pmc_fetch_counter_collection.csv
cp "$(find $O/pmc_write -name '*counter_collection.csv' | head -1)" $O/pmc_write_counter_collection.csv
cp "$(find $O/pmc_enc -name '*counter_collection.csv' | head -1)" $O/pmc_enc_counter_collection.csv
echo "pmc done"
cd $R
python3 benchmarks/derive_scan_traffic.py $O/pmc_fetch_counter_collection.csv $O/pmc_write_counter_collection.csv r02
cp profiles/scan_traffic.json $O/scan_traffic.json
cp profiles/r02_scan_pmc_fetch.csv profiles/r02_scan_pmc_write.csv $O/ 2>/dev/null || true
python3 profiles/summarize_mfma_pmc.py $O/pmc_enc_counter_collection.csv > $O/encoder_mfma_utilisation.csv 2>/dev/null || true
# keep the merge-back small: drop the raw rocprof trees (the per-kernel summaries were copied out above)
rm -rf $O/scan_q1_only $O/scan_q1_k200_only $O/filter_q9_k200_only $O/encoder_1stream $O/pmc_fetch $O/pmc_write $O/pmc_enc
ls -la $O
