// dp_conv_k2.hip - the edge-side fused conv (dp_conv_impl.h) instantiated for layer kind 2: one translation unit per kind, so that the six
// compile side by side
#include "dp_conv_impl.h"

#ifndef DP_CONV_ONE_TU
int dp_conv_launch_k2(const DpConvArgs& a, hipStream_t st) { return dp::launch<2>(a, st); }
#endif
