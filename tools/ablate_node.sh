#!/bin/bash
# build ablation variants of the node-side conv (timing only) -> build_abl/libabl_node<k>.so
cd "$(dirname "$0")/../diffphore_amd/csrc" || exit 1
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result -DDP_NABL=$k -c dp_conv_node.hip -o /tmp/dp_conv_node_abl$k.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC dp_conv.o /tmp/dp_conv_node_abl$k.o dp_graph.o dp_host.o dp_update.o -o ../../build_abl/libabl_node$k.so || exit 1
done
