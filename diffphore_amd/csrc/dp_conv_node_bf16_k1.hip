// dp_conv_node_bf16_k1.hip - the bf16-operand node-side conv (dp_conv_node_bf16_impl.h) instantiated for layer kind 1: one translation unit per kind,
// so that the four compile side by side
#include "dp_conv_node_bf16_impl.h"

int dpb_launch_k1(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpb::launch_bf16<1>(fa, xmode, st); }
