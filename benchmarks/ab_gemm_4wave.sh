#!/bin/bash
# the 128 x 384 block as FOUR waves of 128 x 96 wave tiles (GwGeom<8>: 192 accumulators, one wave per SIMD,
# 28 KiB of fragment reads per 144 MFMAs instead of 2 x 20 KiB) against the eight-wave kernel: parity, the four layer
# shapes interleaved in one process, then the encoder forward.
CS_GEMM_WIDE_SHAPE=1384 python -m pytest tests/test_gpu_gemm_split.py -q -m gpu -k "wide" 2>&1 | tail -2
CODES=384,192,1384 ROUNDS=5 python benchmarks/gemm_shape_ab.py 2>/dev/null
for rep in 1 2; do for sh in 0 1384; do
  if [ $sh == 0 ]; then unset CS_GEMM_WIDE_SHAPE; else export CS_GEMM_WIDE_SHAPE=$sh; fi
  python3 benchmarks/encoder_bench.py --iters 10 --stages 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_us_per_layer']; print('shape $sh', round(d['device_ms_per_batch'],3), s['qkv_gemm'], s['ffn_up_gemm'], s['attention'], s['out_proj_gemm'], s['ffn_down_gemm'])"
done; done
unset CS_GEMM_WIDE_SHAPE
CS_GEMM_WIDE_SHAPE=1384 python -m pytest tests/test_gpu_encoder.py -q -m gpu 2>&1 | tail -2
