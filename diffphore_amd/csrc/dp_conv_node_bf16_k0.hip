// dp_conv_node_bf16_k0.hip - the bf16-operand node-side conv (dp_conv_node_bf16_impl.h) instantiated for layer kind 0: one translation unit per kind,
// so that the four compile side by side
#include "dp_conv_node_bf16_impl.h"

int dpb_launch_k0(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpb::launch_bf16<0>(fa, xmode, st); }
