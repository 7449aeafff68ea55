#!/bin/bash
# SQ counters of the attention kernel inside the encoder forward (one pass, --kernel-trace only): where its wave-cycles go.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
CS_ENCODER_STREAMS=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/attnpmc -- python3 $R/benchmarks/encoder_bench.py --iters 2 > $O/attn_pmc.log 2>&1
f=$(find /tmp/attnpmc -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    key = "attention" if "attention_shx" in k else ("gemm_ln" if "gemm_wide_kernelILi16" in k else ("gemm_gelu" if "gemm_wide_kernelILi2E" in k else ("gemm_qkv" if "gemm_wide_kernelILi3E" in k else None)))
    if key is None: continue
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key, d in acc.items():
    print(key, {c: round(v / n[(key, c)] / 1e6, 2) for c, v in d.items()}, "(millions per dispatch)")
PY
