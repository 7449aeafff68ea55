#!/bin/bash
# SQ counters of the dynamic-int8 dense-layer kernels inside the quantised MiniLM-L6 forward (one pass, --kernel-trace
# only): where their wave-cycles go (VALU vs LDS vs matrix pipe vs waiting).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
CS_ENCODER_STREAMS=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/q8pmc -- python3 $R/benchmarks/encoder_bench.py --model minilm-l6-q --quant u8 --iters 2 > $O/q8_pmc.log 2>&1
f=$(find /tmp/q8pmc -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY' | tee $O/q8_pmc_summary.txt
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    m = re.search(r"(gemm_q8_rows_kernel|gemm_q8_kernel|gemm_q8_skinny_kernel)ILi(\d+)E(?:Li(-?\d+)E)?", k)
    if not m: continue
    key = f"{m.group(1)}<{m.group(2)},{m.group(3)}>"
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key, d in sorted(acc.items()):
    v = {c: x / n[(key, c)] for c, x in d.items()}
    wc = v.get("SQ_WAVE_CYCLES", 1.0)
    print(key, {c: round(x / 1e6, 2) for c, x in v.items()}, "(millions per dispatch)")
    print("   fractions of wave-cycles:", {c: round(v[c] / wc, 3) for c in v if c not in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")})
PY
