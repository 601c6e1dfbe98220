// dp_conv_node_f16_k3.hip - the fp32-grade two-limb fp16 node-side conv (dp_conv_node_f16_impl.h) instantiated for layer kind 3: one translation
// unit per kind, so that the four compile side by side
#include "dp_conv_node_f16_impl.h"

int dpf_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st) { return dpf::launch_f16<3>(fa, xmode, st); }
