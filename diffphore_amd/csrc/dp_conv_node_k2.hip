// dp_conv_node_k2.hip - node-side conv kernels of layer kind 2 (see dp_conv_node_impl.h); one translation unit per kind so that the
// four compile in parallel
#include "dp_conv_node_impl.h"

int dpn_launch_k2(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpn::launch<2>(fa, xmode, st); }
