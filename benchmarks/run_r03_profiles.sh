#!/bin/bash
# (round 3 script, kept for the record: the fused FFN kernel and benchmarks/ffn_time.py it profiles were removed in round 4)
# Round-3 evidence set (profiles/r03_*): run from the repo root on the GPU box; results land in gpurun_out/r03/.
# Every rocprofv3 pass profiles ONE kernel population (bench.py --only-scan = the timed loop alone); counters are
# collected in their own passes with --kernel-trace only.
set -e
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
python bench.py > $O/bench_full_run.json 2> $O/bench_full_run.err
echo "bench done"
# the one-process multi-GPU path rehearsed on this box's single GPU (two shards on device 0: NOT a scaling figure)
CS_BENCH_SHARD_DEVICES=0,0 python bench.py --gpus 2 --rows 5000000 --steps 100 --warmup 10 > $O/bench_one_process_2shards_on_1gpu.json 2> $O/bench_one_process.err
python benchmarks/ffn_time.py > $O/ffn_fused_time_and_ablations.log 2>&1
bash benchmarks/ab_attention_p.sh > $O/attention_p_single_f16_experiment.log 2>&1 || true
for f in 1 0; do CS_FFN_FUSED=$f python3 benchmarks/encoder_bench.py --iters 10 --stages 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('CS_FFN_FUSED=$f', round(d['device_ms_per_batch'],3), d.get('stages_us_per_layer'))"; done > $O/encoder_ffn_fused_ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
stats() {  # name, then the command after `--`
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/${name}_kernel_stats.csv
}
stats scan_q1_only python3 $R/bench.py --only-scan --steps 100 --warmup 10
stats filter_q1000_only python3 $R/bench.py --only-scan --nq 1000 --steps 20 --warmup 3
stats filter_q9_k200_only python3 $R/bench.py --only-scan --nq 9 --k 200 --steps 50 --warmup 5
CS_ENCODER_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encoder_1stream -- python3 $R/benchmarks/encoder_bench.py --iters 10 > $O/encoder_1stream.log 2>&1
cp "$(find $O/encoder_1stream -name '*kernel_stats.csv' | head -1)" $O/encoder_1stream_kernel_stats.csv
CS_FFN_FUSED=1 CS_ENCODER_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encoder_ffn_fused -- python3 $R/benchmarks/encoder_bench.py --iters 10 > $O/encoder_ffn_fused.log 2>&1
cp "$(find $O/encoder_ffn_fused -name '*kernel_stats.csv' | head -1)" $O/encoder_ffn_fused_kernel_stats.csv
echo "kernel traces done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --only-scan --steps 5 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --only-scan --steps 5 --warmup 1 > $O/pmc_write.log 2>&1
CS_ENCODER_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_enc -- python3 $R/benchmarks/encoder_bench.py --iters 2 > $O/pmc_enc.log 2>&1
cp "$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)" $O/pmc_fetch_counter_collection.csv
cp "$(find $O/pmc_write -name '*counter_collection.csv' | head -1)" $O/pmc_write_counter_collection.csv
cp "$(find $O/pmc_enc -name '*counter_collection.csv' | head -1)" $O/pmc_enc_counter_collection.csv
echo "pmc done"
cd $R
python3 benchmarks/derive_scan_traffic.py $O/pmc_fetch_counter_collection.csv $O/pmc_write_counter_collection.csv r03
cp profiles/scan_traffic.json $O/scan_traffic.json
cp profiles/r03_scan_pmc_fetch.csv profiles/r03_scan_pmc_write.csv $O/ 2>/dev/null || true
python3 profiles/summarize_mfma_pmc.py $O/pmc_enc_counter_collection.csv > $O/encoder_mfma_utilisation.csv 2>/dev/null || true
rm -rf $O/scan_q1_only $O/filter_q1000_only $O/filter_q9_k200_only $O/encoder_1stream $O/encoder_ffn_fused $O/pmc_fetch $O/pmc_write $O/pmc_enc
ls -la $O
