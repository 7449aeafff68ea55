#!/bin/bash
# configs[1] (1 query over 1,000,000 x 384 rows, streaming route) under a few knob settings, same box:
#   prime pass from k = 1 (default) | never at k = 10 (CS_SCAN_PRIME_MIN_K=17) ; blocks per CU 1 (default) | 2 | 4 (laboratory knob: diagnostic library)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() {  # label, env...
  label=$1; shift
  env "$@" python3 -c "
import json, bench
from codesearch_amd import VectorStore
d = bench.scan_1m_leg(384, 10, 0, VectorStore)
print('$label', json.dumps({k: d[k] for k in d if not isinstance(d[k], (dict, list, str))}))
" 2>&1 | tail -1
}
for rep in 1 2; do
run "default" A=1
run "prime_min_k=17" CS_SCAN_PRIME_MIN_K=17
run "diag default" CS_LIBCSGPU=$R/codesearch_amd/libcsgpu_diag.so
run "diag blocks/CU=2" CS_LIBCSGPU=$R/codesearch_amd/libcsgpu_diag.so CS_SCAN_BLOCKS_PER_CU=2
run "diag blocks/CU=2 no prime" CS_LIBCSGPU=$R/codesearch_amd/libcsgpu_diag.so CS_SCAN_BLOCKS_PER_CU=2 CS_SCAN_PRIME_MIN_K=17
run "diag blocks/CU=4" CS_LIBCSGPU=$R/codesearch_amd/libcsgpu_diag.so CS_SCAN_BLOCKS_PER_CU=4
done
