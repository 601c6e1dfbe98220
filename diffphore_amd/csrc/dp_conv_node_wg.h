// dp_conv_node_wg.h - the body of a node-side conv workgroup (dp_conv_node_impl.h), textually included
//   * directly into tp_conv_node_kernel<KIND, HAS_LL, XMODE>            (SIDE = -1: one workgroup per node group computes every path)
//   * into the two branches of tp_conv_node_split_kernel                 (SIDE = 0 / 1: the paths of one half of the out slots)
// Expects in scope: template parameters KIND, HAS_LL, XMODE; `constexpr int SIDE`; `FusedArgs fa`; `int bid`; `cx_only` (conv split: the only
// conv this workgroup runs, -1 = all of them); the __shared__ arrays
// s_T, s_S, s_inv.  `return` leaves the kernel.
  using S = Spec<KIND>;
  static_assert(4 * NB * S::D_OUT <= 4 * TSTRIDE, "reduction scratch fits in the T buffers");
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, g = lane >> 4, c = lane & 15;
  const Grp grp = grp_of(fa.conv[0], bid);
  if (grp.n_loc <= 0) return;
  float* sT = s_T + q * TSTRIDE;
  int* const s_pair = reinterpret_cast<int*>(s_T + TSTRIDE - 2 * NB * 32);     // LL mode: [2][NB][32] edge slots of the pass's pairs
  (void)s_pair;
  constexpr int P0 = next_owned<KIND, SIDE>(-1);                               // first path this workgroup computes

  f32x4 acc[n_acc4<KIND>()];
#pragma unroll
  for (int i = 0; i < n_acc4<KIND>(); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  NProfT prof_prev{};
  (void)prof_prev;
  NPROF(-1);

  // ---- the convs of the launch, back to back into the same accumulators ------------------------------------------------------------
  if constexpr (HAS_LL) {
    if (cx_only <= 0) {
      constexpr int MODE = LL;
      const DpConvArgs& a = fa.conv[0];
      NPROF_CONV(0);
#include "dp_conv_node_conv.h"
    }
  }
#pragma unroll 1
  for (int cx = 0; cx < fa.n_x; ++cx) {
    if (cx_only >= 0 && cx_only != (HAS_LL ? 1 : 0) + cx) continue;
    constexpr int MODE = XMODE;
    const DpConvArgs& a = fa.conv[(HAS_LL ? 1 : 0) + cx];
    NPROF_CONV((HAS_LL ? 1 : 0) + cx);
#include "dp_conv_node_conv.h"
  }

  // ---- sum the four k-partials, BatchNorm shift, residual stream ------------------------------------------------------------------
  // (mean and BatchNorm scale are already inside the operands.)  Epilogue operands (shifts, residual / previous output): fixed trip
  // count, unrolled and requested BEFORE the partial sums go through LDS, so the rounds' loads travel together and behind the reduction
  {
    const DpConvArgs& a = fa.conv[0];
    const int n_conv = (HAS_LL ? 1 : 0) + fa.n_x;
    constexpr int EP_ROUNDS = (NB * S::D_OUT + 255) / 256;
    float er[EP_ROUNDS], eb[EP_ROUNDS];
#pragma unroll
    for (int it = 0; it < EP_ROUNDS; ++it) {
      const int i = tid + 256 * it;
      const bool live = i < grp.n_loc * S::D_OUT;
      const int n = i / S::D_OUT, ch = i - n * S::D_OUT;
      const bool mine = live && (SIDE < 0 || kChanOwner<KIND>.v[live ? ch : 0] == SIDE);
      const int node = grp.n_lo + (live ? n : 0);
      float sh = 0.f;
      if (cx_only < 0)
        for (int k = 0; k < n_conv; ++k) sh += fa.conv[k].bn_shift[live ? ch : 0];
      eb[it] = sh;
      const float* src = a.accumulate ? a.out + (size_t)node * a.out_stride + (live ? ch : 0)
                                      : (a.resid != nullptr && live && ch < a.resid_width ? a.resid + (size_t)node * a.resid_stride + ch : nullptr);
      er[it] = (cx_only < 0 && mine && src != nullptr) ? *src : 0.f;
    }
    __syncthreads();
    AccStore<KIND, 0>::run(acc, s_T + q * (NB * S::D_OUT), lane);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EP_ROUNDS; ++it) {
      const int i = tid + 256 * it;
      const bool live = i < grp.n_loc * S::D_OUT;
      const int n = i / S::D_OUT, ch = i - n * S::D_OUT;
      if (!live || (SIDE >= 0 && kChanOwner<KIND>.v[ch] != SIDE)) continue;
      const float part = ((s_T[i] + s_T[NB * S::D_OUT + i]) + s_T[2 * NB * S::D_OUT + i]) + s_T[3 * NB * S::D_OUT + i];
      if (cx_only < 0) a.out[(size_t)(grp.n_lo + n) * a.out_stride + ch] = er[it] + (part + eb[it]);
      else fa.part[(size_t)cx_only * fa.part_stride + (size_t)(grp.n_lo - fa.node0 + n) * a.out_stride + ch] = part;   // summed by dp_conv_node_combine
    }
  }
  NPROF(12);
#ifdef DP_PROF
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&dpn_prof[i], prof_prev.acc[i]);
#endif
