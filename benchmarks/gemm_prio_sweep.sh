export GEMM_TIME_SHAPES_ONLY=1
for shape in 384 192; do for stag in 0 1200; do for prio in 0 1; do
  if [ $shape = 384 ] && [ $stag != 0 ]; then continue; fi
  echo "== shape $shape stagger $stag prio $prio"
  CS_GEMM_WIDE_SHAPE=$shape CS_GEMM_WIDE_STAGGER=$stag CS_GEMM_WIDE_PRIO=$prio python benchmarks/gemm_time.py 2>/dev/null | sed 's/128x128.*wide/wide/'
done; done; done
