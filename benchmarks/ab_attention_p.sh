for v in codesearch_amd/libcsgpu.so codesearch_amd/variants/libcsgpu_phi.so; do
  echo "== $v"
  CS_LIBCSGPU=$(realpath $v) python3 tests/encoder_error_vs_oracle.py 2>&1 | tail -1
  CS_LIBCSGPU=$(realpath $v) python3 benchmarks/encoder_bench.py --iters 10 --stages 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['device_ms_per_batch'],3), d.get('stages_us_per_layer'))"
  CS_LIBCSGPU=$(realpath $v) python3 -m pytest tests/test_gpu_encoder.py -m gpu -q -x -k "golden or full_bge" 2>&1 | tail -2
done
