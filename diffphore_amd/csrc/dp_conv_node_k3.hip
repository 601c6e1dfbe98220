// dp_conv_node_k3.hip - node-side conv kernels of layer kind 3 (see dp_conv_node_impl.h); one translation unit per kind so that the
// four compile in parallel
#include "dp_conv_node_impl.h"

int dpn_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpn::launch<3>(fa, xmode, st); }

#ifdef DP_PROF
extern "C" int dp_debug_prof_node(unsigned long long* host16, int reset) {
  if (host16 && hipMemcpyFromSymbol(host16, HIP_SYMBOL(dpn_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(dpn_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
