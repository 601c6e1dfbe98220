// dp_conv_node_pass.h - one pass (32 edge slots per node; HALF: 16) of one conv of the node-side kernel, textually included by
// dp_conv_node_conv.h once per variant.  Why includes and not functions: the same text wrapped in a (force-inlined) device function or
// lambda compiles to measurably worse code with this toolchain (hipcc 7.2: 183 instead of 55 spilled registers for layer 3, +6 % launch
// time) although nothing is left of the call after inlining.
// Expects in scope: template parameters KIND; constexpr MODE, SIDE, HALF, P0; `const DpConvArgs& a` (this conv), `const Grp grp`,
// `const Wg r`, `int et`; tid / lane / q / g / c; sT, s_pair, s_T, s_S, s_inv; acc; prof_prev.
{
    constexpr int NT2 = HALF ? 1 : 2;                // 16-edge tiles per node in this pass
    constexpr int NSTEP = NB * NT2;                  // GEMM1 steps = (node, tile) pairs
    __syncthreads();                               // readers of the previous pass's factors are done
    // operands of the pass's node tiles (see GEMM1 below) and the packed first layer: requested now, consumed after the staging
    float w1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) w1[s] = a.w1n[(q * 16 + s) * 64 + lane];
    constexpr int SBLK = MODE == DP_EDGES_L2P ? 2 : 1, GBLK = MODE == DP_EDGES_L2P ? 1 : 2;      // attr block of the scatter / gather node
    f4u nt4[3];
    float nt1[3];
    {
      const float* s_tab = MODE == DP_EDGES_L2P ? a.attr2_tab : a.attr1_tab;
      const float* g_tab = MODE == DP_EDGES_L2P ? a.attr1_tab : a.attr2_tab;
      const int s_stride = MODE == DP_EDGES_L2P ? a.attr2_stride : a.attr1_stride, g_stride = MODE == DP_EDGES_L2P ? a.attr1_stride : a.attr2_stride;
#pragma unroll
      for (int t2 = 0; t2 < NT2; ++t2) {
        const int e0 = 32 * et + 16 * t2 + c;
        const int grow = (MODE == DP_EDGES_P2L ? r.p0 : r.l0) + (e0 < r.En ? e0 : 0);
        const float* rowp = g_tab + (size_t)grow * g_stride + 5 * g;
        nt4[t2] = *reinterpret_cast<const f4u*>(rowp);
        nt1[t2] = rowp[4];
      }
      const int srow = grp.n_lo + (c < grp.n_loc ? c : 0);
      const float* rowp = s_tab + (size_t)srow * s_stride + 5 * g;
      nt4[2] = *reinterpret_cast<const f4u*>(rowp);
      nt1[2] = rowp[4];
    }
    const float bone = g == 0 ? 1.f : 0.f;
    // GEMM1 step -> (node, 16-edge tile)
    auto load_av = [&](int step, f4u& v4, float& v1, float& msk, int which) {
      int nn_o = HALF ? step : step >> 1;            // opaque copy: keeps the 16 nodes' row pointers from being hoisted out of the pass loop (spills)
      asm volatile("" : "+s"(nn_o));
      const int t2s = HALF ? 0 : step & 1;
      const int node = grp.n_lo + (nn_o < grp.n_loc ? nn_o : 0);
      const int e0 = 32 * et + 16 * t2s + c;         // GEMM1 view: lane (g, c) feeds emb[edge c][group g]
      bool ok = e0 < r.En;
      const int e = ok ? e0 : 0;
      int frow;
      if constexpr (MODE == LL) {
        frow = s_pair[which * (NB * 32) + nn_o * 32 + 16 * t2s + c];        // -1: no such node / pair / edge of this kind
        ok = frow >= 0;
        frow = ok ? frow : 0;
      } else {
        frow = feat_row<MODE>(r, node, e);
      }
      const float* pe = a.edge_emb + (size_t)frow * 20 + 5 * g;
      v4 = *reinterpret_cast<const f4u*>(pe);
      v1 = pe[4];
      msk = (!ok && g == 0) ? -1e30f : 0.f;
    };
    constexpr int PF = DP_NODE_PF;                             // steps the edge operands are requested ahead (ring of PF + 1 register sets)
    f4u e4[PF + 1];
    float e1[PF + 1], em[PF + 1];
    if constexpr (MODE != LL) {                                // dense modes: edge rows are arithmetic, the first requests can leave now
#pragma unroll
      for (int s0 = 0; s0 < PF; ++s0) load_av(s0, e4[s0], e1[s0], em[s0], 0);
    }
    constexpr int SLOTS = HALF ? 16 : 32;                      // edge slots per node staged in this pass
#pragma unroll
    for (int idx = tid; idx < NB * SLOTS; idx += 256) {       // (two rounds, unrolled: both rounds' row loads travel together)
      const int nn = idx / SLOTS, el = idx % SLOTS, e = 32 * et + el;
      bool ok = nn < grp.n_loc && e < r.En;
      int frow = 0;
      if constexpr (MODE == LL) {
        // the pass's pair -> edge-slot entries are parked in LDS (tail of wave 0's idle T buffer) for GEMM1: its row gathers then hang
        // on an LDS read instead of on a second dependent global load
        int sa = -1, sb = -1;
        if (ok) { sa = pair_slot<0>(a, r, grp.n_lo + nn, e); sb = pair_slot<1>(a, r, grp.n_lo + nn, e); }
        s_pair[nn * 32 + el] = sa;
        s_pair[NB * 32 + nn * 32 + el] = sb;
        frow = sa >= 0 ? sa : sb;
        ok = frow >= 0;
      } else if (ok) {
        frow = feat_row<MODE>(r, grp.n_lo + nn, e);
      }
      float v[SCOMP];
#pragma unroll
      for (int i = 0; i < SCOMP; ++i) v[i] = 0.f;
      if (ok) {
        const float* ps = a.edge_sh + (size_t)frow * a.sh_stride;
        const float cc = 0.57735026918962576451f * ps[6];
        v[0] = ps[1]; v[1] = ps[2]; v[2] = ps[3];
        v[3] = -cc - ps[8]; v[4] = ps[5]; v[5] = ps[4]; v[6] = 2.f * cc; v[7] = ps[7]; v[8] = -cc + ps[8];
      }
#pragma unroll
      for (int i = 0; i < SCOMP; ++i) s_S[((nn * 2 + (el >> 4)) * SCOMP + i) * 16 + (el & 15)] = v[i];
    }
    __syncthreads();
    NPROF(0);

    // ---- GEMM1: hidden rows 16q..16q+15 of the pass's edges of every node of the batch -----------------------------------------
    // attr = [edge_emb | attr1[:20] | attr2[:20]] and two of the three blocks belong to NODES, not edges: the scatter node's block is the
    // same for all of its edges and the gather node's block the same for every scatter node.  So the first layer is split
    //     pre[e] = W1_emb emb[e]  +  u_s[scatter(e)]  +  u_g[gather(e)],     u_s = W1_s attr_s + b,   u_g = W1_g attr_g
    // with u_s / u_g evaluated once per pass as three 16-row MFMA tiles (rows = nodes) and the per-edge part shrinking from 16 K-steps
    // to 5 (+ 1 mask step).  u_g comes out of the MFMA in exactly the D-fragment layout of the edge tiles (rows = the pass's gather
    // nodes = its edge slots), u_s is broadcast over rows through 1 KiB of the (idle) T buffer.
    float hid[NB][8];
    auto node_tile = [&](const f4u& v4, float v1, int blk, bool bias) {
      f32x4 D = {0.f, 0.f, 0.f, 0.f};
      D = mfma16(v4.x, w1[5 * blk + 0], D);
      D = mfma16(v4.y, w1[5 * blk + 1], D);
      D = mfma16(v4.z, w1[5 * blk + 2], D);
      D = mfma16(v4.w, w1[5 * blk + 3], D);
      D = mfma16(v1, w1[5 * blk + 4], D);
      if (bias) D = mfma16(g == 0 ? 1.f : 0.f, w1[15], D);
      return D;
    };
    f32x4 ug[2];                                               // u_g of the pass's edge slots: register rr <-> slot 16 t2 + 4 g + rr
    // edge part: five consecutive floats of edge_emb per lane group; the loads of step i + PF are issued before the MFMAs of step i
    // (a step is only 6 MFMAs, so the requests have to run several steps ahead of an L2 / HBM round trip).  An edge that does not exist
    // (e >= En, or no radius / bond edge for the pair) gets -1e30 added to every pre-activation by the mask step (A = -1e30 in lane
    // group 0, B = ones): its hidden vector - constant-one unit included - is exactly 0 and it drops out of T.
    auto gemm1 = [&](const f4u& v4, float v1, float msk, float usn, const f32x4& ugt) {
      f32x4 D = {usn + ugt[0], usn + ugt[1], usn + ugt[2], usn + ugt[3]};
#if DP_NABL != 6
      D = mfma16(v4.x, w1[0], D);
      D = mfma16(v4.y, w1[1], D);
      D = mfma16(v4.z, w1[2], D);
      D = mfma16(v4.w, w1[3], D);
      D = mfma16(v1, w1[4], D);
      D = mfma16(msk, bone, D);
#else
      D[0] += v4.x + msk; D[1] += v4.y; D[2] += v4.z; D[3] += v4.w + v1;
#endif
      return D;
    };
    if constexpr (MODE == LL) {                                // (the dense modes' first requests left before the staging)
#pragma unroll
      for (int s0 = 0; s0 < PF; ++s0) load_av(s0, e4[s0], e1[s0], em[s0], 0);
    }
    // the node tiles, while those requests travel
    ug[0] = node_tile(nt4[0], nt1[0], GBLK, false);
    if constexpr (!HALF) ug[1] = node_tile(nt4[1], nt1[1], GBLK, false);
    else ug[1] = ug[0];
    {
      const f32x4 us = node_tile(nt4[2], nt1[2], SBLK, true);                              // + fc.0 bias (and the constant-one unit)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) sT[(4 * g + rr) * 16 + c] = us[rr];                   // [node][k]: read back as a broadcast over rows
    }
    __builtin_amdgcn_sched_barrier(0);
    float xr0[3][8];
    float us_cur = sT[c], us_nxt = 0.f;
    // the scatter MEAN (torch_scatter: sum / max(count, 1)) is applied to the hidden vectors: T and the output are linear in them, so
    // the convs of a fused launch - whose degrees differ - add up in the same accumulators
    float inv_cur = s_inv[0], inv_nxt = 0.f;
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int nn = HALF ? step : step >> 1, t2 = HALF ? 0 : step & 1;
      const bool last_of_node = HALF || t2 == 1;
      if (step + PF < NSTEP) load_av(step + PF, e4[(step + PF) % (PF + 1)], e1[(step + PF) % (PF + 1)], em[(step + PF) % (PF + 1)], 0);
      if (t2 == 0 && nn + 1 < NB) { us_nxt = sT[(nn + 1) * 16 + c]; inv_nxt = s_inv[nn + 1]; }
      if (step == NSTEP - DP_NODE_XR0_BACK) load_x<KIND, MODE, HALF, P0>(a, r, et, lane, 0, xr0);   // first column tile's operands: behind the last emb request
      const f32x4 D = gemm1(e4[step % (PF + 1)], e1[step % (PF + 1)], em[step % (PF + 1)], us_cur, ug[t2]);
      if constexpr (MODE == LL) {                 // (scaled after the bond sweep below)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hid[nn][4 * t2 + rr] = fmaxf(D[rr], 0.f);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hid[nn][4 * t2 + rr] = fmaxf(D[rr], 0.f) * inv_cur;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (last_of_node) { us_cur = us_nxt; inv_cur = inv_nxt; }
    }
    if constexpr (MODE == LL) {
      // second GEMM1 over the bond edges of the same pairs (1-4 per atom): skipped per 16-pair tile when none of them is bonded;
      // same operand ring
#pragma unroll
      for (int s0 = 0; s0 < PF; ++s0) load_av(s0, e4[s0], e1[s0], em[s0], 1);
      us_cur = sT[c];
      inv_cur = s_inv[0];
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        const int nn = HALF ? step : step >> 1, t2 = HALF ? 0 : step & 1;
        const bool last_of_node = HALF || t2 == 1;
        if (step + PF < NSTEP) load_av(step + PF, e4[(step + PF) % (PF + 1)], e1[(step + PF) % (PF + 1)], em[(step + PF) % (PF + 1)], 1);
        if (t2 == 0 && nn + 1 < NB) { us_nxt = sT[(nn + 1) * 16 + c]; inv_nxt = s_inv[nn + 1]; }
        // lane group 0 carries the validity of the tile's 16 pairs (msk == 0 <=> bonded)
        if (__builtin_amdgcn_ballot_w64(g == 0 && em[step % (PF + 1)] == 0.f) != 0) {
          const f32x4 D = gemm1(e4[step % (PF + 1)], e1[step % (PF + 1)], em[step % (PF + 1)], us_cur, ug[t2]);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) hid[nn][4 * t2 + rr] += fmaxf(D[rr], 0.f);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hid[nn][4 * t2 + rr] *= inv_cur;
        __builtin_amdgcn_sched_barrier(0);
        if (last_of_node) { us_cur = us_nxt; inv_cur = inv_nxt; }
      }
    }
    if constexpr (MODE == LL) __syncthreads();     // s_pair lives in wave 0's T buffer: every wave is done reading it before T tiles land there
    NPROF(1);
    PathStage<KIND, MODE, SIDE, HALF, P0>::run(a, r, et, q, lane, hid, sT, s_S, acc, xr0, prof_prev);
}
