// dp_conv_node_bf16_k2.hip - the bf16-operand node-side conv (dp_conv_node_bf16_impl.h) instantiated for layer kind 2: one translation unit per kind,
// so that the four compile side by side
#include "dp_conv_node_bf16_impl.h"

int dpb_launch_k2(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpb::launch_bf16<2>(fa, xmode, st); }
