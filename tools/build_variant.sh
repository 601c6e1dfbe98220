#!/bin/bash
# build a variant of the library for timing / ablation runs: tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]
# compiles the named translation units with the extra flags and links them with the product's other objects -> build_abl/lib<NAME>.so
# (build_abl/ is git-ignored but travels with gpurun; load with DIFFPHORE_LIB=build_abl/lib<NAME>.so, ablation builds also need
# DIFFPHORE_ALLOW_ABLATION=1)
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../diffphore_amd/csrc"
mkdir -p ../../build_abl/obj_$name
objs=""
for f in *.hip; do
  o=${f%.hip}.o
  for v in "$@"; do
    if [ "$v" == "$f" ]; then
      o=../../build_abl/obj_$name/${f%.hip}.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result $flags -c $f -o $o &
    fi
  done
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../build_abl/lib$name.so
rm -rf ../../build_abl/obj_$name
echo built build_abl/lib$name.so
