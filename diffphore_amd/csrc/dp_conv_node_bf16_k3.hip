// dp_conv_node_bf16_k3.hip - the bf16-operand node-side conv (dp_conv_node_bf16_impl.h) instantiated for layer kind 3: one translation unit per kind,
// so that the four compile side by side
#include "dp_conv_node_bf16_impl.h"

int dpb_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpb::launch_bf16<3>(fa, xmode, st); }
