// dp_conv_k0.hip - the edge-side fused conv (dp_conv_impl.h) instantiated for layer kind 0: one translation unit per kind, so that the six
// compile side by side
#include "dp_conv_impl.h"

#ifndef DP_CONV_ONE_TU
int dp_conv_launch_k0(const DpConvArgs& a, hipStream_t st) { return dp::launch<0>(a, st); }
#endif
