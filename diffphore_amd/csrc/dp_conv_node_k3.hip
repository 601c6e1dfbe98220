// dp_conv_node_k3.hip - node-side conv kernels of layer kind 3 (see dp_conv_node_impl.h); one translation unit per kind so that the
// four compile in parallel
#include "dp_conv_node_impl.h"

int dpn_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpn::launch<3>(fa, xmode, st); }
