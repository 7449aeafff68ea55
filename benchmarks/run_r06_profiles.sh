#!/bin/bash
# Round-6 evidence set (profiles/r06_*): run from the repo root on the GPU box; results land in gpurun_out/r06p/.
# Every rocprofv3 pass profiles ONE kernel population (bench.py --only-scan = the timed loop alone); counters are
# collected in their own passes with --kernel-trace only.
set -e
R=$PWD
O=$R/gpurun_out/r06p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # name, then the command after `--`
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/${name}_kernel_stats.csv
  python3 $R/benchmarks/phase_timeline.py $O/$name 24 > $O/${name}_timeline.txt 2>&1 || true
}
stats scan_q1_only python3 $R/bench.py --only-scan --steps 100 --warmup 10
stats default_route_q1_k10 python3 $R/bench.py --only-scan --route cost --steps 100 --warmup 10
stats filter_q8_k10 python3 $R/bench.py --only-scan --nq 8 --k 10 --steps 50 --warmup 5
stats filter_q9_k200 python3 $R/bench.py --only-scan --nq 9 --k 200 --steps 50 --warmup 5
stats filter_q1000_k10 python3 $R/bench.py --only-scan --nq 1000 --steps 20 --warmup 3
CS_ENCODER_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encoder_1stream -- python3 $R/benchmarks/encoder_bench.py --iters 10 > $O/encoder_1stream.log 2>&1
cp "$(find $O/encoder_1stream -name '*kernel_stats.csv' | head -1)" $O/encoder_1stream_kernel_stats.csv
CS_ENCODER_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encoder_l512 -- python3 $R/benchmarks/encoder_bench.py --iters 10 --batch 128 --seq 512 > $O/encoder_l512.log 2>&1 || true
cp "$(find $O/encoder_l512 -name '*kernel_stats.csv' | head -1)" $O/encoder_l512_kernel_stats.csv 2>/dev/null || true
echo "kernel traces done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --only-scan --steps 5 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --only-scan --steps 5 --warmup 1 > $O/pmc_write.log 2>&1
CS_ENCODER_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_enc -- python3 $R/benchmarks/encoder_bench.py --iters 2 > $O/pmc_enc.log 2>&1
cp "$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)" $O/pmc_fetch_counter_collection.csv
cp "$(find $O/pmc_write -name '*counter_collection.csv' | head -1)" $O/pmc_write_counter_collection.csv
cp "$(find $O/pmc_enc -name '*counter_collection.csv' | head -1)" $O/pmc_enc_counter_collection.csv
echo "pmc done"
cd $R
python3 benchmarks/derive_scan_traffic.py $O/pmc_fetch_counter_collection.csv $O/pmc_write_counter_collection.csv r06 || true
cp profiles/scan_traffic.json $O/scan_traffic.json || true
cp profiles/r06_scan_pmc_fetch.csv profiles/r06_scan_pmc_write.csv $O/ 2>/dev/null || true
python3 profiles/summarize_mfma_pmc.py $O/pmc_enc_counter_collection.csv > $O/encoder_mfma_utilisation.csv 2>/dev/null || true
rm -rf $O/scan_q1_only $O/default_route_q1_k10 $O/filter_q8_k10 $O/filter_q9_k200 $O/filter_q1000_k10 $O/encoder_1stream $O/encoder_l512 $O/pmc_fetch $O/pmc_write $O/pmc_enc
ls -la $O
