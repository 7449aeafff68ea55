#!/bin/bash
# Builds codesearch_amd/variants/libcsgpu_<name>.so: the in-tree objects, with the listed sources recompiled under extra
# flags / defines.  For A/B runs through CS_LIBCSGPU (benchmarks/ab_*.sh); the in-tree library is never touched.
#   usage: build_variant.sh <name> "<extra flags>" file1.hip [file2.hip ...]
set -e
name=$1; extra=$2; shift 2
cd "$(dirname "$0")/../codesearch_amd/csrc"
make -s
mkdir -p ../variants/obj_$name
objs=""
for o in $(make -s objs); do
  src=${o%.o}.hip
  use=$o
  for f in "$@"; do
    if [ "$f" == "$src" ]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-fast-math -fvisibility=hidden -fvisibility-inlines-hidden $extra -c $src -o ../variants/obj_$name/$o
      use=../variants/obj_$name/$o
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script=exports.map -o ../variants/libcsgpu_$name.so $objs
echo built codesearch_amd/variants/libcsgpu_$name.so
