#!/bin/bash
# LDS-side SQ counters of the attention kernel inside the encoder forward (one --pmc pass each, --kernel-trace only).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r05p
mkdir -p $O
pass() {  # name, counters...
  name=$1; shift
  CS_ENCODER_STREAMS=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/attnpmc_$name -- python3 $R/benchmarks/encoder_bench.py --iters 2 > $O/attn_pmc_$name.log 2>&1
  f=$(find /tmp/attnpmc_$name -name '*counter_collection.csv' | head -1)
  [ -s "$f" ] || { echo "pass $name: no counter_collection.csv (too many counters for one pass? at most four per pass)"; tail -5 $O/attn_pmc_$name.log; exit 1; }
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    key = "attention" if "attention_shx" in k else ("gemm_gelu" if "gemm_wide_kernelILi2E" in k else None)
    if key is None: continue
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key, d in acc.items():
    print(key, {c: round(v / n[(key, c)] / 1e6, 3) for c, v in d.items()}, "(millions per dispatch)")
PY
}
# (at most four counters per pass: a pass that asks for more than the SQ's counter slots comes back empty)
pass a1 SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pass a2 SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
pass b1 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA
pass b2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pass c1 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU
pass c2 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_CVT
