// dp_conv_node_conv.h - every pass of ONE conv of a (possibly fused) node-side launch; included by dp_conv_node_wg.h once per edge mode.
// Expects in scope: KIND, HALFOK, SIDE, P0; constexpr MODE; `const DpConvArgs& a` (this conv); grp; tid / lane / q / g / c; sT, s_pair, s_T, s_S,
// s_inv; acc; prof_prev.
{
  const Wg r = wg_of<MODE>(a, grp.pose);
  __syncthreads();                                 // the previous conv's GEMM1 is done with s_inv
  if (tid < NB) {
    // scatter MEAN over the node's edges (torch_scatter: / max(count, 1)); the dense cross sets have En edges per node
    const int node = grp.n_lo + (tid < grp.n_loc ? tid : 0);
    const int deg = MODE == LL ? a.rowend[node] - a.rowptr[node] : r.En;
    s_inv[tid] = 1.f / (float)max(deg, 1);
  }
  const int n_et = (r.En + 31) >> 5;
#pragma unroll 1
  for (int et = 0; et < n_et; ++et) {
    // a last pass of <= 16 edge slots runs with one 16-edge tile per node (half of GEMM1 and GEMM-A); which variant a pass takes
    // depends on the pose's own edge count only, so a pose's result does not depend on the rest of the batch
    if constexpr (HALFOK && DP_NODE_HALF_PASS) {
      if (r.En - 32 * et <= 16) {
        constexpr bool HALF = true;
#include "dp_conv_node_pass.h"
      } else {
        constexpr bool HALF = false;
#include "dp_conv_node_pass.h"
      }
    } else {        // lean instantiation (FusedArgs.lean: the host knows that no pose ends on a short pass) - or a short pass run as a full one
      constexpr bool HALF = false;
#include "dp_conv_node_pass.h"
    }
  }
}
