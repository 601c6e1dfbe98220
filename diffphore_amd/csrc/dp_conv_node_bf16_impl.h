// dp_conv_node_bf16_impl.h - the node-side formulation of the fused tensor-product convolution (dp_conv_node_impl.h) with bf16
// OPERANDS on the bf16 matrix pipe of gfx950 (v_mfma_f32_16x16x32_bf16: 16x the fp32-input MFMA rate), fp32 accumulate.
//
// Replaces TensorProductConvLayer.forward (/root/reference/src/models/score_model_phore.py:134-149) for BASELINE.json configs[4]
// ("bf16 mixed precision": bf16 radial MLP / weights, fp32 accumulate, fp32 geometry - SURVEY.md 8d).  What is bf16 here: the radial
// MLP's two layers (W1, W2 and their activations), the edge embedding and node scalars fed to layer 1, the tensor-product operand
// z = (gathered features x spherical-harmonic factors) - computed in fp32, rounded once - and T = hid^T z between the two GEMMs.
// What stays fp32: positions, distances, spherical harmonics, edge embeddings' own MLPs, every accumulation, BatchNorm shift, residual.
//
// Same association as the fp32 kernel:
//     T_p[n][k][u,m]  = sum_{e in n} hid[e][k] z_p[e][u,m]            GEMM-A: ONE 16x16x32 MFMA per (node, 15-column tile, 32-edge pass)
//     out[n][w,m]    += sum_{k,u} W2_p[(u,w)][k] T_p[n][k][u,m]        GEMM-B: K = 32 = two in-channels x the wave's 16 hidden rows
// and the same mapping: a workgroup = 4 waves = one group of <= 16 scatter nodes of one pose, wave q owns hidden rows 16q..16q+15, so
// GEMM1 -> GEMM-A -> LDS transpose -> GEMM-B is wave-private; the four k-partials meet in LDS once at the end.
//
// What changes with the 16x pipe - the kernel is no longer bound by the matrix pipe but by the VALU that builds z and by the LDS:
//   * GEMM1 is one MFMA per (node, 16-edge tile): A = a 64-byte bf16 row of the edge embedding (written by the edge-embedding kernels
//     beside the fp32 one; feature 20 is the "edge does not exist" flag whose W1 row is -1e30), B = the wave's 16 columns of W1.
//     The node terms u_s / u_g are two more MFMAs per pass, as before.  hid leaves the accumulator as 8 bf16 per lane = GEMM-A's A
//     operand (64 registers for the 16 nodes instead of 128).
//   * column tiles are processed in PAIRS (a "supertile" = 10 in-channels: 2 x (5 in-channels x 3 m), or the 10 in-channels of a
//     scalar-output tile): both tiles of a pair take the same SH factors per lane, so those are read from LDS once per (node, pair),
//     and GEMM-B's K = 32 is exactly two in-channels (5 K-chunks per supertile).
//   * T crosses from "columns on lanes" to "nodes on lanes" as bf16 through a wave-private LDS image [column][k quad][node ^ column][4]:
//     one ds_write_b64 per (node, tile) - the XOR spreads the 16 lanes of a store over all banks - and two ds_read_b64 per MFMA.
//   * W2 is packed offline as bf16 A fragments (16 bytes per lane and MFMA), BatchNorm scale folded in.
#pragma once
#include "dp_conv_node_impl.h"

namespace dpb {
using namespace dpn;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(4)));
typedef u32x2 __attribute__((address_space(3))) lds_u32x2;
__device__ __forceinline__ lds_u32x2* lds_ptr(unsigned a) { return (lds_u32x2*)(__UINTPTR_TYPE__)a; }      // LDS byte address -> pointer

constexpr int SUPER_NU = 10;          // in-channels per supertile
constexpr int SUPER_CH = 5;           // GEMM-B K chunks (two in-channels x 16 hidden rows) per supertile
constexpr int EMB_ROW = 32;           // bf16 per row of the bf16 edge-embedding table (20 features | flag | padding) = 64 bytes
constexpr int EMB_FLAG = 20;          // feature slot of the "edge does not exist" flag
constexpr int COLB = 512;             // bytes per T column: 4 k quads x 16 nodes x 4 bf16

template <int KIND> constexpr int st_count(int p) { return Spec<KIND>::P[p].mul_in / SUPER_NU; }
template <int KIND> constexpr int w2b_path_off(int p) {             // in 16-byte quads
  int o = 0;
  for (int i = 0; i < p; ++i) o += st_count<KIND>(i) * SUPER_CH * path_mt<KIND>(i) * 4 * 64;
  return o;
}
template <int KIND> constexpr int w2b_quads() { return w2b_path_off<KIND>(Spec<KIND>::NPATH); }
constexpr int W1B_QUADS = 3 * 4 * 64;                                // three images (edge embedding, attr1 block, attr2 block) x 4 waves
template <int KIND> constexpr bool supertiles_ok() {
  for (int i = 0; i < Spec<KIND>::NPATH; ++i)
    if (Spec<KIND>::P[i].mul_in % SUPER_NU != 0) return false;
  return true;
}
static_assert(supertiles_ok<0>() && supertiles_ok<1>() && supertiles_ok<2>() && supertiles_ok<3>(), "every path has a multiple of 10 in-channels");

__device__ __forceinline__ unsigned pk2(float lo, float hi) {       // v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// column of in-channel `ul` (0..9 inside the supertile), component m, in the wave's T image; `xkey` = the lane (column of its tile) that wrote it
template <bool VEC> constexpr int t_col(int ul, int m) { return VEC ? (ul / 5) * TCOLS + m * 5 + (ul % 5) : ul; }
template <bool VEC> constexpr int t_key(int ul, int m) { return VEC ? m * 5 + (ul % 5) : ul; }

// gathered operands of supertile (path P, st) for this lane's 8 (HALF: 4) edges, both tiles; masking and sign are applied when the
// supertile starts, so that the loads can travel during the previous supertile's GEMM-B
template <int KIND, int MODE, bool HALF, int P>
__device__ __forceinline__ void load_x2(const DpConvArgs& a, const Wg& r, int et, int lane_l, int st, float (&xr)[2][3][8]) {
  using S = Spec<KIND>;
  if constexpr (P < S::NPATH) {
    constexpr PathC pc = S::P[P];
    constexpr int ZT = pc.ztype;
    constexpr bool VEC = d_out_of(ZT) == 3;
    constexpr int DIN = d_in_of(ZT), NTERM = n_terms<ZT>(), NTILE = VEC ? 2 : 1;
    const int g = lane_l >> 4, c = lane_l & 15;
    const int m = VEC ? (c >= 10 ? 2 : (c >= 5 ? 1 : 0)) : 0;
    const int ul = VEC ? c - 5 * m : c;
    const bool colv = VEC ? c < 15 : c < 10;
    const float* __restrict__ base = a.gather_tab;
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) {
      const int u = colv ? SUPER_NU * st + 5 * tl + ul : 0;
      unsigned coff[NTERM];
#pragma unroll
      for (int t = 0; t < NTERM; ++t) coff[t] = (unsigned)(pc.in_off + u * DIN + term_of<ZT>(t, m).xcomp);
#pragma unroll
      for (int j = 0; j < (HALF ? 4 : 8); ++j) {
        const int e = 32 * et + 16 * (j >> 2) + 4 * g + (j & 3);
        const unsigned roff = (unsigned)gather_row<MODE>(r, e < r.En ? e : 0) * (unsigned)a.gather_stride;
#pragma unroll
        for (int t = 0; t < NTERM; ++t) xr[tl][t][j] = base[roff + coff[t]];
      }
    }
  }
}

template <int KIND, int MODE, bool HALF, int P>
struct StageB {
  static __device__ __forceinline__ void run(const DpConvArgs& a, const Wg& r, int et, int q, int lane, const u32x4 (&hid)[NB],
                                             unsigned tl_base, unsigned dummy_base, const float* sS, f32x4 (&acc)[n_acc4<KIND>()],
                                             float (&xr)[2][3][8]) {
    using S = Spec<KIND>;
    if constexpr (P < S::NPATH) {
      constexpr PathC pc = S::P[P];
      constexpr int ZT = pc.ztype;
      constexpr bool VEC = d_out_of(ZT) == 3;
      constexpr int NG = VEC ? 3 : 1, NTERM = n_terms<ZT>(), MT = path_mt<KIND>(P), AB = acc_base<KIND>(P), NTILE = VEC ? 2 : 1;
      constexpr int NJ = HALF ? 4 : 8, NT2 = HALF ? 1 : 2;
#pragma unroll 1
      for (int st = 0; st < st_count<KIND>(P); ++st) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));          // lane-derived constants are rebuilt per supertile (keeps LICM from hoisting and spilling them)
        const int g = lane_l >> 4, c = lane_l & 15;
        const int m = VEC ? (c >= 10 ? 2 : (c >= 5 ? 1 : 0)) : 0;
        const bool colv = VEC ? c < 15 : c < 10;
        // ---- gathered operand of the lane's column in both tiles (the same rows for every node of the group) --------------------
        float xa[NTILE][NTERM][NJ];
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl)
#pragma unroll
          for (int t = 0; t < NTERM; ++t) {
            const Term tm = term_of<ZT>(t, m);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              const int e = 32 * et + 16 * (j >> 2) + 4 * g + (j & 3);
              xa[tl][t][j] = (colv && e < r.En) ? tm.sign * xr[tl][t][j] : 0.f;
            }
          }
        // T image: byte address of (column, k quad g, node nn) = column * 512 + g * 128 + ((nn ^ key) << 3), key = the writing lane's c
        unsigned twr[NTILE];
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl)
          twr[tl] = (colv ? tl_base + (unsigned)((tl * TCOLS + c) * COLB) : dummy_base) + (unsigned)(g * 128 + c * 8);
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM-A: one MFMA per (node, tile); T leaves as bf16 ----------------------------------------------------------------
        if constexpr (needs_s<ZT>()) {
          const char* sbase = reinterpret_cast<const char*>(sS);
          int soff[NTERM];
#pragma unroll
          for (int t = 0; t < NTERM; ++t) soff[t] = (term_of<ZT>(t, m).scomp * 16 + 4 * g) * 4;
          auto loadS = [&](int nn, f32x4 (&sv)[NTERM][NT2]) {
#pragma unroll
            for (int t = 0; t < NTERM; ++t)
#pragma unroll
              for (int t2 = 0; t2 < NT2; ++t2)
                sv[t][t2] = *reinterpret_cast<const f32x4*>(sbase + (2 * nn + t2) * (SCOMP * 64) + soff[t]);
          };
          f32x4 sv[2][NTERM][NT2];
          loadS(0, sv[0]);
#pragma unroll
          for (int nn = 0; nn < NB; ++nn) {
            if (nn + 1 < NB) loadS(nn + 1, sv[(nn + 1) & 1]);
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) {
              float z[NJ];
#pragma unroll
              for (int j = 0; j < NJ; ++j) {
                float zz = xa[tl][0][j] * sv[nn & 1][0][j >> 2][j & 3];
#pragma unroll
                for (int t = 1; t < NTERM; ++t) zz = fmaf(xa[tl][t][j], sv[nn & 1][t][j >> 2][j & 3], zz);
                z[j] = zz;
              }
              u32x4 zb = {pk2(z[0], z[1]), pk2(z[2], z[3]), 0u, 0u};
              if constexpr (!HALF) { zb[2] = pk2(z[NJ - 4], z[NJ - 3]); zb[3] = pk2(z[NJ - 2], z[NJ - 1]); }
              const f32x4 D = mfma_bf(as_bf(hid[nn]), as_bf(zb), f32x4{0.f, 0.f, 0.f, 0.f});
              *lds_ptr(twr[tl] ^ (unsigned)(nn << 3)) = u32x2{pk2(D[0], D[1]), pk2(D[2], D[3])};
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // z is the gathered input itself (sh_0 = 1): the B operand is the same for every node
          u32x4 zb[NTILE];
#pragma unroll
          for (int tl = 0; tl < NTILE; ++tl)
          {
            zb[tl] = u32x4{pk2(xa[tl][0][0], xa[tl][0][1]), pk2(xa[tl][0][2], xa[tl][0][3]), 0u, 0u};
            if constexpr (!HALF) { zb[tl][2] = pk2(xa[tl][0][NJ - 4], xa[tl][0][NJ - 3]); zb[tl][3] = pk2(xa[tl][0][NJ - 2], xa[tl][0][NJ - 1]); }
          }
#pragma unroll
          for (int nn = 0; nn < NB; ++nn) {
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) {
              const f32x4 D = mfma_bf(as_bf(hid[nn]), as_bf(zb[tl]), f32x4{0.f, 0.f, 0.f, 0.f});
              *lds_ptr(twr[tl] ^ (unsigned)(nn << 3)) = u32x2{pk2(D[0], D[1]), pk2(D[2], D[3])};
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // ---- GEMM-B: out[w][node] += W2[w][(u,k)] T[(u,k)][node], K = (two in-channels) x (the wave's 16 hidden rows) -----------------
        {
          constexpr int PN = P + 1;
          const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.w2n) + w2b_path_off<KIND>(P) + (size_t)(st * SUPER_CH * MT) * 256 + q * 64 + lane_l;
          const unsigned rbase = tl_base + (unsigned)(g * 128 + c * 8);          // GEMM-B view: node c, k quad g
          u32x4 wf[MT], wn[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) wf[mt] = wsrc[mt * 256];
#pragma unroll
          for (int ch = 0; ch < SUPER_CH; ++ch) {
            if (ch + 1 < SUPER_CH) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) wn[mt] = wsrc[((ch + 1) * MT + mt) * 256];
            }
            if (ch == SUPER_CH - 2) {
              // the next supertile's gathered operands travel behind the last W2 request of this one
              if (st + 1 < st_count<KIND>(P)) load_x2<KIND, MODE, HALF, P>(a, r, et, lane_l, st + 1, xr);
              else load_x2<KIND, MODE, HALF, PN>(a, r, et, lane_l, 0, xr);
            }
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
              const u32x2 lo = *lds_ptr((rbase ^ (unsigned)(t_key<VEC>(2 * ch, grp) << 3)) + (unsigned)(t_col<VEC>(2 * ch, grp) * COLB));
              const u32x2 hi = *lds_ptr((rbase ^ (unsigned)(t_key<VEC>(2 * ch + 1, grp) << 3)) + (unsigned)(t_col<VEC>(2 * ch + 1, grp) * COLB));
              const u32x4 tb = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) acc[AB + grp * MT + mt] = mfma_bf(as_bf(wf[mt]), as_bf(tb), acc[AB + grp * MT + mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ch + 1 < SUPER_CH) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) wf[mt] = wn[mt];
            }
          }
        }
      }
      StageB<KIND, MODE, HALF, P + 1>::run(a, r, et, q, lane, hid, tl_base, dummy_base, sS, acc, xr);
    }
  }
};

// one pass (32 edge slots per node; HALF: 16) of one conv: stage the SH factors, GEMM1 -> hid (bf16 A fragments), then the path stages
template <int KIND, int MODE, bool HALF>
__device__ __forceinline__ void pass_bf16(const DpConvArgs& a, const Grp& grp, const Wg& r, int et, int tid, float* s_T, float* s_S,
                                          const float* s_inv, f32x4 (&acc)[n_acc4<KIND>()]) {
  const int lane = tid & 63, q = tid >> 6, g = lane >> 4, c = lane & 15;
  constexpr int NT2 = HALF ? 1 : 2;
  constexpr int NSTEP = NB * NT2;
  float* sT = s_T + q * TSTRIDE;
  int* const s_pair = reinterpret_cast<int*>(s_T + TSTRIDE - 2 * NB * 32);     // LL mode: [2][NB][32] edge slots of the pass's pairs
  (void)s_pair;
  __syncthreads();                               // readers of the previous pass's factors (and of s_pair / the T images) are done
  // ---- operands that do not depend on the staging: the wave's three W1 images and the pass's node rows ---------------------------
  const u32x4* w1 = reinterpret_cast<const u32x4*>(a.w1n);
  const u32x4 Wemb = w1[(0 * 4 + q) * 64 + lane], Wa1 = w1[(1 * 4 + q) * 64 + lane], Wa2 = w1[(2 * 4 + q) * 64 + lane];
  const u32x4 Wsc = MODE == DP_EDGES_L2P ? Wa2 : Wa1, Wga = MODE == DP_EDGES_L2P ? Wa1 : Wa2;      // attr block of the scatter / gather node
  // a node tile's A fragment: lane (row c, group g) holds features 8g..8g+7 of the row's first 20 (bf16), feature 20 = 1 carries the bias
  auto attr_frag = [&](const float* rowp, bool bias) {
    const int o = g < 2 ? 8 * g : 16;
    f32x4 lo = *reinterpret_cast<const f4u*>(rowp + o);
    f32x4 hi = *reinterpret_cast<const f4u*>(rowp + (g < 2 ? o + 4 : 16));
    if (g >= 2) hi = f32x4{(g == 2 && bias) ? 1.f : 0.f, 0.f, 0.f, 0.f};
    if (g == 3) lo = f32x4{0.f, 0.f, 0.f, 0.f};
    return u32x4{pk2(lo[0], lo[1]), pk2(lo[2], lo[3]), pk2(hi[0], hi[1]), pk2(hi[2], hi[3])};
  };
  u32x4 ag[NT2], as;
  {
    const float* s_tab = MODE == DP_EDGES_L2P ? a.attr2_tab : a.attr1_tab;
    const float* g_tab = MODE == DP_EDGES_L2P ? a.attr1_tab : a.attr2_tab;
    const int s_stride = MODE == DP_EDGES_L2P ? a.attr2_stride : a.attr1_stride, g_stride = MODE == DP_EDGES_L2P ? a.attr1_stride : a.attr2_stride;
#pragma unroll
    for (int t2 = 0; t2 < NT2; ++t2) {
      const int e0 = 32 * et + 16 * t2 + c;
      const int grow = (MODE == DP_EDGES_P2L ? r.p0 : r.l0) + (e0 < r.En ? e0 : 0);
      ag[t2] = attr_frag(g_tab + (size_t)grow * g_stride, false);
    }
    const int srow = grp.n_lo + (c < grp.n_loc ? c : 0);
    as = attr_frag(s_tab + (size_t)srow * s_stride, true);
  }
  // ---- stage the SH-derived factors of the pass (fp32) and, in the pair mode, the pass's pair -> edge-slot entries ----------------
  constexpr int SLOTS = HALF ? 16 : 32;
#pragma unroll
  for (int idx = tid; idx < NB * SLOTS; idx += 256) {
    const int nn = idx / SLOTS, el = idx % SLOTS, e = 32 * et + el;
    bool ok = nn < grp.n_loc && e < r.En;
    int frow = 0;
    if constexpr (MODE == LL) {
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

  // ---- GEMM1: pre[e][k] = W1_emb emb[e] + u_s[scatter node] + u_g[gather node]; hid = relu(pre) / degree, as bf16 A fragments -----------
  f32x4 ug[NT2];
#pragma unroll
  for (int t2 = 0; t2 < NT2; ++t2) ug[t2] = mfma_bf(as_bf(ag[t2]), as_bf(Wga), f32x4{0.f, 0.f, 0.f, 0.f});     // rows = the pass's edge slots
  {
    const f32x4 us = mfma_bf(as_bf(as), as_bf(Wsc), f32x4{0.f, 0.f, 0.f, 0.f});                                   // rows = the group's nodes, + bias
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sT[(4 * g + rr) * 16 + c] = us[rr];             // [node][k]: read back as a broadcast over rows
  }
  const unsigned short* embb = reinterpret_cast<const unsigned short*>(a.edge_emb_bf);
  // A fragment of (node nn, 16-edge tile t2): 16 bytes of the edge's bf16 embedding row; the flag row for an edge that does not exist
  auto load_a = [&](int step, int which) {
    int nn_o = HALF ? step : step >> 1;
    asm volatile("" : "+s"(nn_o));                // opaque: keeps the 16 nodes' row arithmetic from being hoisted out of the pass (spills)
    const int t2s = HALF ? 0 : step & 1;
    const int e0 = 32 * et + 16 * t2s + c;
    int frow;
    if constexpr (MODE == LL) {
      frow = s_pair[which * (NB * 32) + nn_o * 32 + 16 * t2s + c];
    } else {
      frow = (nn_o < grp.n_loc && e0 < r.En) ? feat_row<MODE>(r, grp.n_lo + nn_o, e0) : -1;
    }
    if (frow < 0) frow = a.emb_flag_row;
    return *reinterpret_cast<const u4u*>(embb + (size_t)frow * EMB_ROW + 8 * g);
  };
  constexpr int PF = 4;
  u32x4 ea[PF + 1], eb[PF + 1];
  (void)eb;
#pragma unroll
  for (int s0 = 0; s0 < PF; ++s0) {
    ea[s0] = load_a(s0, 0);
    if constexpr (MODE == LL) eb[s0] = load_a(s0, 1);
  }
  u32x4 hid[NB];
  float xr0[2][3][8];
  float us_cur = sT[c], us_nxt = 0.f, inv_cur = s_inv[0], inv_nxt = 0.f;
  float hf[8];
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int nn = HALF ? step : step >> 1, t2 = HALF ? 0 : step & 1;
    const bool last_of_node = HALF || t2 == 1;
    if (step + PF < NSTEP) {
      ea[(step + PF) % (PF + 1)] = load_a(step + PF, 0);
      if constexpr (MODE == LL) eb[(step + PF) % (PF + 1)] = load_a(step + PF, 1);
    }
    if (t2 == 0 && nn + 1 < NB) { us_nxt = sT[(nn + 1) * 16 + c]; inv_nxt = s_inv[nn + 1]; }
    if (step == NSTEP - 3) load_x2<KIND, MODE, HALF, 0>(a, r, et, lane, 0, xr0);          // first supertile's operands: behind the last emb request
    const f32x4 init = {us_cur + ug[t2][0], us_cur + ug[t2][1], us_cur + ug[t2][2], us_cur + ug[t2][3]};
    const f32x4 D = mfma_bf(as_bf(ea[step % (PF + 1)]), as_bf(Wemb), init);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) hf[4 * t2 + rr] = fmaxf(D[rr], 0.f);
    if constexpr (MODE == LL) {
      // the bond edges of the same pairs (1-4 per atom): second layer-1 evaluation over the same u_s + u_g, skipped per 16-pair tile when
      // none of them is bonded; the pair's two hidden vectors add (same geometry, same gathered row)
      const u32x4 b4 = eb[step % (PF + 1)];
      // an absent bond edge reads the flag row: its slot 20 (lane group 2, third dword, low half) is 1.0
      if (__builtin_amdgcn_ballot_w64(g == 2 && (b4[2] & 0xFFFFu) == 0u) != 0) {
        const f32x4 D2 = mfma_bf(as_bf(b4), as_bf(Wemb), init);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hf[4 * t2 + rr] += fmaxf(D2[rr], 0.f);
      }
    }
    if (last_of_node) {
      hid[nn] = u32x4{pk2(hf[0] * inv_cur, hf[1] * inv_cur), pk2(hf[2] * inv_cur, hf[3] * inv_cur), HALF ? 0u : pk2(hf[4] * inv_cur, hf[5] * inv_cur),
                      HALF ? 0u : pk2(hf[6] * inv_cur, hf[7] * inv_cur)};
      us_cur = us_nxt; inv_cur = inv_nxt;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (MODE == LL) __syncthreads();      // s_pair lives in wave 0's T image: every wave is done reading it before T tiles land there
  const unsigned tl_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)(s_T) + (unsigned)(q * TSTRIDE * 4);
  const unsigned dummy_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)(s_T) + (unsigned)(4 * TSTRIDE * 4);
  StageB<KIND, MODE, HALF, 0>::run(a, r, et, q, lane, hid, tl_base, dummy_base, s_S, acc, xr0);
}

// every pass of one conv of the (possibly fused) launch
template <int KIND, int MODE>
__device__ __forceinline__ void conv_bf16(const DpConvArgs& a, const Grp& grp, int tid, float* s_T, float* s_S, float* s_inv,
                                          f32x4 (&acc)[n_acc4<KIND>()]) {
  const Wg r = wg_of<MODE>(a, grp.pose);
  __syncthreads();                                 // the previous conv's GEMM1 is done with s_inv
  if (tid < NB) {
    const int node = grp.n_lo + (tid < grp.n_loc ? tid : 0);
    const int deg = MODE == LL ? a.rowend[node] - a.rowptr[node] : r.En;
    s_inv[tid] = 1.f / (float)max(deg, 1);         // scatter MEAN (torch_scatter: / max(count, 1)), applied to the hidden vectors
  }
  const int n_et = (r.En + 31) >> 5;
#pragma unroll 1
  for (int et = 0; et < n_et; ++et) {
    if (r.En - 32 * et <= 16) pass_bf16<KIND, MODE, true>(a, grp, r, et, tid, s_T, s_S, s_inv, acc);
    else pass_bf16<KIND, MODE, false>(a, grp, r, et, tid, s_T, s_S, s_inv, acc);
  }
}

template <int KIND, bool HAS_LL, int XMODE>
__global__ __launch_bounds__(256, 2) void tp_conv_node_bf16_kernel(FusedArgs fa) {
  __shared__ __attribute__((aligned(128))) float s_T[4 * TSTRIDE + TDUMMY];   // wave-private T images (bf16) + the dummy column; reused for the reduction
  __shared__ __attribute__((aligned(16))) float s_S[NB * SCOMP * 32];         // per-edge sh factors of the current pass (fp32)
  __shared__ float s_inv[NB];
  using S = Spec<KIND>;
  static_assert(4 * NB * S::D_OUT <= 4 * TSTRIDE, "reduction scratch fits in the T images");
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const Grp grp = grp_of(fa.conv[0], (int)blockIdx.x);
  if (grp.n_loc <= 0) return;
  f32x4 acc[n_acc4<KIND>()];
#pragma unroll
  for (int i = 0; i < n_acc4<KIND>(); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (HAS_LL) conv_bf16<KIND, LL>(fa.conv[0], grp, tid, s_T, s_S, s_inv, acc);
#pragma unroll 1
  for (int cx = 0; cx < fa.n_x; ++cx) conv_bf16<KIND, XMODE>(fa.conv[(HAS_LL ? 1 : 0) + cx], grp, tid, s_T, s_S, s_inv, acc);
  // ---- sum the four k-partials, BatchNorm shift, residual stream (mean and BatchNorm scale are inside the operands) -----------------
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
      const int node = grp.n_lo + (live ? n : 0);
      float sh = 0.f;
      for (int k = 0; k < n_conv; ++k) sh += fa.conv[k].bn_shift[live ? ch : 0];
      eb[it] = sh;
      const float* src = a.accumulate ? a.out + (size_t)node * a.out_stride + (live ? ch : 0)
                                      : (a.resid != nullptr && live && ch < a.resid_width ? a.resid + (size_t)node * a.resid_stride + ch : nullptr);
      er[it] = (live && src != nullptr) ? *src : 0.f;
    }
    __syncthreads();
    AccStore<KIND, 0>::run(acc, s_T + q * (NB * S::D_OUT), lane);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EP_ROUNDS; ++it) {
      const int i = tid + 256 * it;
      if (i >= grp.n_loc * S::D_OUT) continue;
      const int n = i / S::D_OUT, ch = i - n * S::D_OUT;
      const float part = ((s_T[i] + s_T[NB * S::D_OUT + i]) + s_T[2 * NB * S::D_OUT + i]) + s_T[3 * NB * S::D_OUT + i];
      a.out[(size_t)(grp.n_lo + n) * a.out_stride + ch] = er[it] + (part + eb[it]);
    }
  }
}

template <int KIND> int launch_bf16(const FusedArgs& fa, int xmode, hipStream_t st) {
  const DpConvArgs& a = fa.conv[0];
  const dim3 grid((unsigned)(a.B * a.slices)), block(256);
  if (fa.n_ll) hipLaunchKernelGGL((tp_conv_node_bf16_kernel<KIND, true, DP_EDGES_P2L>), grid, block, 0, st, fa);
  else if (xmode == DP_EDGES_P2L) hipLaunchKernelGGL((tp_conv_node_bf16_kernel<KIND, false, DP_EDGES_P2L>), grid, block, 0, st, fa);
  else hipLaunchKernelGGL((tp_conv_node_bf16_kernel<KIND, false, DP_EDGES_L2P>), grid, block, 0, st, fa);
  return (int)hipGetLastError();
}

}  // namespace dpb

int dpb_launch_k0(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpb_launch_k1(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpb_launch_k2(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpb_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
