#!/bin/bash
# rocprofv3 evidence for the NomicBert forward (profiles/r04_nomic_*): run from the repo root on the GPU box; results in
# gpurun_out/r04n/.  One kernel population per pass; the MFMA-busy counters in their own pass with --kernel-trace only.
set -e
R=$PWD
O=$R/gpurun_out/r04n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CS_ENCODER_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nomic_1stream -- python3 $R/benchmarks/encoder_bench.py --model nomic-v1.5 --batch 128 --iters 10 > $O/nomic_1stream.log 2>&1
cp "$(find $O/nomic_1stream -name '*kernel_stats.csv' | head -1)" $O/nomic_1stream_kernel_stats.csv
CS_ENCODER_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_nomic -- python3 $R/benchmarks/encoder_bench.py --model nomic-v1.5 --batch 128 --iters 2 > $O/pmc_nomic.log 2>&1
cp "$(find $O/pmc_nomic -name '*counter_collection.csv' | head -1)" $O/pmc_nomic_counter_collection.csv
cd $R
python3 profiles/summarize_mfma_pmc.py $O/pmc_nomic_counter_collection.csv > $O/nomic_mfma_utilisation.csv 2>/dev/null || true
rm -rf $O/nomic_1stream $O/pmc_nomic
ls -la $O
