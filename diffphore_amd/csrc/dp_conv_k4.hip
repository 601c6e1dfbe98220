// dp_conv_k4.hip - the edge-side fused conv (dp_conv_impl.h) instantiated for layer kind 4: one translation unit per kind, so that the six
// compile side by side
#include "dp_conv_impl.h"

#ifndef DP_CONV_ONE_TU
int dp_conv_launch_k4(const DpConvArgs& a, hipStream_t st) { return dp::launch<4>(a, st); }
#endif
