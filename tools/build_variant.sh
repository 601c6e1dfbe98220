#!/bin/bash
# build a variant of the library for timing / ablation runs: tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]
# compiles the named translation units with the extra flags and links them with the product's other objects -> build_abl/lib<NAME>.so
# (build_abl/ is git-ignored but travels with gpurun; load with DIFFPHORE_LIB=build_abl/lib<NAME>.so, ablation builds also need
# DIFFPHORE_ALLOW_ABLATION=1).  dp_host.hip is ALWAYS recompiled with the same flags and the source hash "variant:<NAME>": dp_build_info then
# reports the variant's switches (DP_NABL / DP_BABL / DPF_ABL / DP_ABLATE / DP_PROF / DP_SPLIT_LEAN), so _lib.lib() refuses an ablation build
# unless asked and bench.py's config.library never shows the product's hash for a variant.  The two-limb fp16 kernels get -fno-slp-vectorize
# as in the product build (diffphore_amd/csrc/build.py).
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../diffphore_amd/csrc"
mkdir -p ../../build_abl/obj_$name
objs=""
for f in *.hip; do
  o=${f%.hip}.o
  hit=0
  [ "$f" == "dp_host.hip" ] && hit=1
  for v in "$@"; do [ "$v" == "$f" ] && hit=1; done
  if [ $hit == 1 ]; then
    o=../../build_abl/obj_$name/${f%.hip}.o
    extra=""
    case "$f" in dp_conv_node_f16_k*) extra="-fno-slp-vectorize";; esac
    [ "$f" == "dp_host.hip" ] && extra="-DDP_SOURCE_HASH=\"variant:$name\""
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result $flags $extra -c $f -o $o &
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $objs -o ../../build_abl/lib$name.so
rm -rf ../../build_abl/obj_$name
echo built build_abl/lib$name.so
