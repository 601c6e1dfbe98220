// dp_conv_node_k0.hip - node-side conv kernels of layer kind 0 (see dp_conv_node_impl.h); one translation unit per kind so that the
// four compile in parallel
#include "dp_conv_node_impl.h"

int dpn_launch_k0(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpn::launch<0>(fa, xmode, st); }
