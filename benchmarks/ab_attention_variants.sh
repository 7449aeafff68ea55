#!/bin/bash
# attention variants (benchmarks/build_variant.sh ... attention_split.hip) x CS_ATTN_PIPE, BGE-small 256 x 256 and 128 x 512:
# attention microseconds per layer on one box.  Usage: ab_attention_variants.sh name1 name2 ...  ("product" = in-tree library)
for rep in 1 2; do
for v in "$@"; do
  lib=codesearch_amd/variants/libcsgpu_$v.so; [ "$v" == product ] && lib=codesearch_amd/libcsgpu.so
  for pipe in 0 1; do
    for shape in "bge-small 256 256" "bge-small 128 512"; do
      set -- $shape
      CS_LIBCSGPU=$(realpath $lib) CS_ATTN_PIPE=$pipe python3 benchmarks/encoder_bench.py --model $1 --batch $2 --seq $3 --iters 6 --stages 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_us_per_layer']; print('$v pipe $pipe  $1 $2 x $3  attention us/layer', s['attention'], ' forward ms', round(d['device_ms_per_batch'],3))"
    done
  done
done
done
