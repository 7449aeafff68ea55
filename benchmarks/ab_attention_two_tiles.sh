#!/bin/bash
# Same-box A/B of the attention kernel's tile-loop forms (CS_ATTN_PIPE, attention_shx_body.hpp): 0 the rolled loop, 1 two key tiles
# in flight per wave, 2 the super-tile written out over pinned LDS addresses + cross-half max on the VALU, 3 = 2 + a tile's K
# fragments requested together and its V fragments before the exponentials.  Attention microseconds per layer and the forward:
# BGE-small (head_dim 32) at 256 x 256 and 128 x 512, BGE-base (head_dim 64) at 128 x 256 and 64 x 512; then the encoder parity
# tests under each form.   usage: ab_attention_two_tiles.sh "0 1 2 3"
set -e
forms=${1:-"0 1 2 3"}
for rep in 1 2; do
  for pipe in $forms; do
    for shape in "bge-small 256 256" "bge-small 128 512" "bge-base 128 256" "bge-base 64 512"; do
      m=${shape%% *}; r=${shape#* }; b=${r% *}; l=${r#* }
      CS_ATTN_PIPE=$pipe python3 benchmarks/encoder_bench.py --model $m --batch $b --seq $l --iters 8 --stages 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_us_per_layer']; print('pipe $pipe  $m $b x $l  attention us/layer', s['attention'], ' forward ms', round(d['device_ms_per_batch'],3))"
    done
  done
done
for pipe in $forms; do
  [ "$pipe" == 0 ] && continue
  echo "parity tests under CS_ATTN_PIPE=$pipe"
  CS_ATTN_PIPE=$pipe python3 -m pytest tests/test_gpu_encoder.py tests/test_gpu_jina.py tests/test_gpu_modern.py tests/test_gpu_nomic.py -m gpu -q -x 2>&1 | tail -2
done
