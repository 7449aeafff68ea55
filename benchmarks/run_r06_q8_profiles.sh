#!/bin/bash
# Round-6 evidence for the quantised default model (MiniLM-L6-Q shape, 256 x 256 tokens): kernel stats and the matrix pipe's busy
# share, each in its own rocprofv3 pass.
set -e
R=$PWD
O=$R/gpurun_out/r06q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CS_ENCODER_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q8_stats -- python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 3 > $O/q8_stats.log 2>&1
cp "$(find $O/q8_stats -name '*kernel_stats.csv' | head -1)" $O/q8_minilm_l6_kernel_stats.csv
CS_ENCODER_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/q8_pmc -- python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 2 > $O/q8_pmc.log 2>&1
cp "$(find $O/q8_pmc -name '*counter_collection.csv' | head -1)" $O/q8_pmc_mfma_busy_raw.csv
cd $R
python3 profiles/summarize_mfma_pmc.py $O/q8_pmc_mfma_busy_raw.csv > $O/q8_mfma_utilisation.csv 2>/dev/null || true
rm -rf $O/q8_stats $O/q8_pmc
head -12 $O/q8_minilm_l6_kernel_stats.csv | cut -c1-160; cat $O/q8_mfma_utilisation.csv | cut -c1-200
