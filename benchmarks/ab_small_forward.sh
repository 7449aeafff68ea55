#!/bin/bash
# A/B of the one-launch forward of short queries (csrc/small_forward.hip) against the kernel-by-kernel path: host-API latency
# of benchmarks/query_latency.py, alternating, same box.  Output: gpurun_out/r05_small_forward_ab.log
out=gpurun_out/r05_small_forward_ab.log
mkdir -p gpurun_out
{
echo "# bash benchmarks/ab_small_forward.sh"
for rep in 1 2 3; do for sf in 0 1; do echo "## CS_SMALL_FORWARD=$sf"; CS_SMALL_FORWARD=$sf python3 benchmarks/query_latency.py 2>/dev/null | head -4; done; done
} 2>&1 | tee $out
