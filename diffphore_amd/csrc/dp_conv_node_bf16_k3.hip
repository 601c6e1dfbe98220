// dp_conv_node_bf16_k3.hip - the bf16-operand node-side conv (dp_conv_node_bf16_impl.h) instantiated for layer kind 3: one translation unit per kind,
// so that the four compile side by side
#include "dp_conv_node_bf16_impl.h"

int dpb_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpb::launch_bf16<3>(fa, xmode, st); }

#ifdef DP_PROF
extern "C" int dp_debug_prof_node(unsigned long long* host16, int reset) {
  if (host16 && hipMemcpyFromSymbol(host16, HIP_SYMBOL(dpb_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(dpb_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
