#!/bin/bash
# A/B of the wide GEMM's MFMA shape: gemm_wide.hip (16x16x32) against gemm_wide32.hip (32x32x16), same block
# shapes, same epilogues.  (1) parity of the new kernels through the existing wide-GEMM and encoder tests with
# CS_GEMM_WIDE_MFMA=32; (2) per-layer-shape kernel times, variants interleaved in one process; (3) the encoder forward,
# libraries alternating.  Output: gpurun_out/r05_gemm_mfma_shape_ab.log
set -o pipefail
out=gpurun_out/r05_gemm_mfma_shape_ab.log
mkdir -p gpurun_out
{
echo "# bash benchmarks/ab_gemm_mfma_shape.sh  ($(git rev-parse --short HEAD 2>/dev/null || echo tree))"
echo "# (1) parity with CS_GEMM_WIDE_MFMA=32"
CS_GEMM_WIDE_MFMA=32 timeout -k 10 900 python3 -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_encoder.py -x -q -m gpu 2>&1 | tail -3
echo "# (2) us per launch, 65,536 rows, codes: 384 / 192 = 16x16x32 at that block shape, 3384 / 3192 = 32x32x16 (median, min)"
CODES=1384,3384,1192,3192 ROUNDS=7 ITERS=30 timeout -k 10 600 python3 benchmarks/gemm_shape_ab.py
echo "# (3) encoder forward (256 x 256, BGE-small shape), ms | per layer us: qkv ffn_up attention out_proj+LN ffn_down+LN"
for rep in 1 2 3; do for mf in 16 32; do CS_GEMM_WIDE_MFMA=$mf python3 benchmarks/encoder_bench.py --iters 10 --stages 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_us_per_layer']; print('mfma $mf', round(d['device_ms_per_batch'],3), s['qkv_gemm'], s['ffn_up_gemm'], s['attention'], s['out_proj_gemm'], s['ffn_down_gemm'])"; done; done
} 2>&1 | tee $out
