#!/bin/bash
# Same-box A/B of the attention kernel's arithmetic (CS_ATTN_ARITH, attention_shx_body.hpp): the in-tree library (2: one score
# accumulator, unscaled residual of the probabilities) against codesearch_amd/variants/libcsgpu_${AB_VARIANT:-arith1}.so (build_variant.sh arith1
# "-DCS_ATTN_ARITH=1" attention_split.hip small_forward.hip: the split product's two accumulators on both sides, rounds 1-4).
# Error against the fp32 oracle first, then attention microseconds per layer and the forward, then the parity tests in-tree.
set -e
for v in codesearch_amd/variants/libcsgpu_${AB_VARIANT:-arith1}.so codesearch_amd/libcsgpu.so; do
  echo "== $v"
  CS_LIBCSGPU=$(realpath $v) python3 tests/encoder_error_vs_oracle.py 2>&1 | tail -1
done
for rep in 1 2; do
  for v in codesearch_amd/variants/libcsgpu_${AB_VARIANT:-arith1}.so codesearch_amd/libcsgpu.so; do
    for shape in "bge-small 256 256" "bge-small 128 512" "bge-base 128 256" "bge-base 64 512"; do
      m=${shape%% *}; r=${shape#* }; b=${r% *}; l=${r#* }
      CS_LIBCSGPU=$(realpath $v) python3 benchmarks/encoder_bench.py --model $m --batch $b --seq $l --iters 8 --stages 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_us_per_layer']; print('$(basename $v)  $m $b x $l  attention us/layer', s['attention'], ' forward ms', round(d['device_ms_per_batch'],3))"
    done
  done
done
python3 -m pytest tests/test_gpu_encoder.py tests/test_gpu_jina.py tests/test_gpu_modern.py tests/test_gpu_nomic.py tests/test_gpu_small_forward.py tests/test_gpu_quantized.py -m gpu -q -x 2>&1 | tail -3
