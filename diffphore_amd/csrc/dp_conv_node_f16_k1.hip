// dp_conv_node_f16_k1.hip - the fp32-grade two-limb fp16 node-side conv (dp_conv_node_f16_impl.h) instantiated for layer kind 1: one translation
// unit per kind, so that the four compile side by side
#include "dp_conv_node_f16_impl.h"

int dpf_launch_k1(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpf::launch_f16<1>(fa, xmode, st); }
