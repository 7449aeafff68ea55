#!/bin/bash
# Round 5: the two encoder families built this round at the reference's mini-batch for their width (768-d: 128, 1024-d: 64
# sequences, embedder.rs:256-260), 256 and 512 tokens, with per-kernel-class times; then their parity tests.
R=${GRAFT_REPO_ROOT:-$PWD}
for spec in "jina-code 128 256" "jina-code 64 512" "modernbert-large 64 256" "modernbert-large 32 512" "nomic-v1.5 128 256" "bge-large 64 256"; do
  set -- $spec
  echo "== $1 batch $2 x seq $3"
  python3 $R/benchmarks/encoder_bench.py --model $1 --batch $2 --seq $3 --iters 5 --stages 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['device_ms_per_batch'],3), 'ms,', round(d['chunks_per_s_device']), 'chunks/s,', round(d['achieved_tflops'],1), 'TFLOP/s algorithmic'); print(d['stages_us_per_layer'])"
done
python3 -m pytest $R/tests/test_gpu_modern.py $R/tests/test_gpu_jina.py -q 2>&1 | tail -2
