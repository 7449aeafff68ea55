#!/bin/bash
# Per-kernel times of the queued multi-unit q8 forward (8 units x 32 x 256 tokens in one device batch) for library
# variants (benchmarks/build_variant.sh), and of the one-unit forward of the same size on the same box:
#   ab_q8_units.sh name...      ("base" = the in-tree library)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/abmu
rm -rf /tmp/abmu_one
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abmu_one -- python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 5 > $R/gpurun_out/abmu/one_unit.log 2>&1
f=$(find /tmp/abmu_one -name "*kernel_stats.csv" | head -1)
echo "== one unit (in-tree library)"; grep "gemm_q8_rows_kernel" $f | awk -F, '{print substr($1,40,24), $2, $4}'
for v in "$@"; do
  lib=$R/codesearch_amd/libcsgpu.so
  [ "$v" != base ] && lib=$R/codesearch_amd/variants/libcsgpu_$v.so
  rm -rf /tmp/abmu_$v
  CS_LIBCSGPU=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abmu_$v -- python3 $R/benchmarks/q8_queue_rate.py > $R/gpurun_out/abmu/$v.log 2>&1
  f=$(find /tmp/abmu_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep "gemm_q8_rows_kernelILi[0-9]*ELi[01]ELb1" $f | awk -F, '{print substr($1,40,24), $2, $4}'
done
