// dp_conv_node_impl.h - device side of the node-side formulation of the fused tensor-product convolution for gfx950.
//
// Same operator as dp_conv.hip (TensorProductConvLayer.forward, score_model_phore.py:134-149) with the two contractions
// re-associated.  With hid[e][k] the radial MLP's hidden layer (k = 60 units + a constant-one unit that carries fc.3's bias),
// z_p[e][u,m] the Clebsch-Gordan-contracted (gathered input x spherical harmonics) of path p and W2_p[(u,w)][k] the second layer:
//
//   edge side (dp_conv.hip):   msg[e][w,m]  = sum_u ( sum_k W2_p[(u,w)][k] hid[e][k] ) z_p[e][u,m]         2*61*numel flop / edge
//   node side (this file):     T_p[n][k][u,m] = sum_{e in n} hid[e][k] z_p[e][u,m]                        GEMM-A, edges are K
//                              out[n][w,m]   += sum_{k,u} W2_p[(u,w)][k] T_p[n][k][u,m]                   GEMM-B, batched over 16 nodes
//
// GEMM-A costs 2*64*(sum_p mul_in*d_out) flop per edge (280 columns at layer 2 instead of numel = 1600), GEMM-B is per NODE, not per
// edge: ~3.5x fewer fp32 MFMA cycles for 32-edge nodes.  The per-edge weight tensor never exists, not even in registers.
//
// Mapping to CDNA4 (v_mfma_f32_16x16x4_f32 everywhere, 32 cycles each)
//   * a workgroup = 4 waves = one group of <= 16 scatter nodes of one pose; wave q owns hidden rows 16q..16q+15 (k-split), so each
//     wave's whole pipeline GEMM1 -> GEMM-A -> LDS transpose -> GEMM-B is wave-private: no workgroup barrier inside a batch.
//   * GEMM1  hid^T[16 edges x 16 k]: the radial MLP's first layer is split into per-node terms (the scatter node's and the gather node's
//     attr blocks, three 16-row node tiles per pass) and a per-edge part over edge_emb (5 K-steps + 1 mask step instead of 16); its D
//     registers ARE GEMM-A's A operand (edge order of GEMM-A's K steps is chosen to be the D-fragment row order), kept in registers
//     for the 16 nodes of the batch (128 VGPRs).  The scatter MEAN (1 / degree) is applied to hid: T and the output are linear in it.
//   * GEMM-A per (column tile, node): 8 MFMAs; B operand z generated on the VALU from the gathered input (registers, loaded once per
//     tile: in the dense modes a lane's gathered rows are the same for every node) and sh-derived factors staged in LDS per batch.
//   * the 16x16 T tile goes through a wave-private, XOR-swizzled LDS buffer (one ds_write_b128 per node, ds_read_b128 back):
//     GEMM-A leaves columns on lanes, GEMM-B needs nodes on lanes.
//   * GEMM-B per column tile: A = packed W2 slice with the BatchNorm scale folded in (dwordx4 loads from L2, shared by the three m
//     components), B = T, N = 16 nodes, accumulating into <= 40 persistent registers; the four waves' k-partial outputs are summed
//     through LDS at the end.
//   Nodes with more than 32 edges run several 32-edge passes that accumulate into the same GEMM-B accumulators (GEMM-B is linear); a
//   last pass of <= 16 edges runs as a HALF pass (one 16-edge tile per node: half of GEMM1 / GEMM-A).
//   CONV FUSION: one launch runs up to three convs that scatter onto the same nodes - ligand-ligand (pair-dense) + phore->ligand +
//   phore->ligand(normals), or ligand->phore + ligand->phore(normals) - back to back into the SAME accumulators (mean and BatchNorm
//   scale are folded into the operands, so the convs' outputs simply add): one prologue, one k-partial reduction, one epilogue and
//   one write of `out` instead of three read-modify-writes (score_model_phore.py:703-710 sums exactly these outputs).
//   Batches too small to fill the chip launch two workgroups per node group, each owning half of the output irrep slots (path split).
// The workgroup body is dp_conv_node_wg.h (included into the unsplit kernel and into both branches of the split kernel); one pass of one
// conv is dp_conv_node_pass.h (included per mode and per full / half variant).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffphore_amd.h"
#include "dp_spec.h"

#ifdef DP_PROF   // cycle-stamp breakdown (timing builds only: tools/build_variant.sh prof "-DDP_PROF" dp_conv_node_k3.hip): wave 0 of every workgroup
                 // adds its segment times; one counter array per translation unit (no relocatable device code), read through dp_debug_prof_node.
                 // Bins: 4 * conv + {0 staging, 1 GEMM1, 2 GEMM-A, 3 GEMM-B} for the convs of a fused launch (conv 0 = ligand-ligand), 12 = k-partial
                 // reduction + epilogue, 13 = workgroup prologue (the bf16 kernel uses bins 0..4 only)
static __device__ unsigned long long dpn_prof[16];
struct NProf { unsigned long long prev, acc[16]; int base; };
__device__ __forceinline__ void nprof_mark(int seg, NProf& p) {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long now = __builtin_readcyclecounter();
  if (seg >= 0) p.acc[seg < 4 ? p.base + seg : seg] += now - p.prev;
  p.prev = now;
  __builtin_amdgcn_sched_barrier(0);
}
#define NPROF(seg) nprof_mark(seg, prof_prev)
#define NPROF_CONV(c) prof_prev.base = 4 * (c)
typedef NProf NProfT;
#else
#define NPROF(seg)
#define NPROF_CONV(c)
typedef int NProfT;
#endif

#ifndef DP_NABL    // ablation builds (tools/ablate_node.sh): 1 no sh factors / z arithmetic, 2 no T transpose through LDS, 3 no GEMM-B MFMAs,
#define DP_NABL 0  // 4 no GEMM-B at all, 5 no GEMM-A MFMAs, 6 no GEMM1 MFMAs, 8 GEMM-B emulated on the bf16 pipe (consumer-side operand split +
                   // 16x16x32 bf16 MFMAs: profiles/r3_experiments.md).  Results are wrong by construction; timing only.
#endif
#ifndef DP_NODE_PF
#define DP_NODE_PF 4          // GEMM1: steps the edge operands are requested ahead
#endif
#ifndef DP_NODE_XR0_BACK
#define DP_NODE_XR0_BACK 3    // GEMM1: the first column tile's gathered operands are requested this many steps before its end
#endif
#ifndef DP_NODE_HALF_PASS
#define DP_NODE_HALF_PASS 1   // 0: every pass is a full 32-slot pass (A/B builds)
#endif

namespace dpn {
using namespace dp;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte load from a 4-byte aligned address

constexpr int NB = 16;        // nodes per workgroup = N of the GEMM-B MFMAs
constexpr int TCOLS = 15;     // columns of a T tile that are stored (3 m-groups x 5 in-channels, or 10 in-channels of a scalar-output path)
constexpr int TSTRIDE = TCOLS * NB * 16;        // floats per wave: 15 column rows of 1 KiB
constexpr int TDUMMY = NB * 16;                 // one more KiB shared by all waves: where the lanes of the unused 16th column store
constexpr int SCOMP = 9;      // per-edge factors staged in LDS: sh1[0..2], M00, M01, M02, M11, M12, M22
constexpr int HIDN = 60;      // hidden units of the radial MLP (kinds 0-3); unit 60 is the constant one

template <int KIND> constexpr bool vec_out(int p) { return d_out_of(Spec<KIND>::P[p].ztype) == 3; }
template <int KIND> constexpr int tile_nu(int p) { return vec_out<KIND>(p) ? 5 : 10; }                 // in-channels per column tile
template <int KIND> constexpr int path_tiles(int p) { return Spec<KIND>::P[p].mul_in / tile_nu<KIND>(p); }
template <int KIND> constexpr int path_mt(int p) { return (Spec<KIND>::P[p].mul_out + 15) / 16; }        // 16-row M tiles of GEMM-B
template <int KIND> constexpr int path_steps(int p) { return tile_nu<KIND>(p) * 4; }                      // GEMM-B K steps per m-group
template <int KIND> constexpr int w2n_tile_floats(int p) { return 4 * 64 * path_steps<KIND>(p); }         // per (tile, M tile): [q][lane][step]
template <int KIND> constexpr int w2n_path_off(int p) {
  int o = 0;
  for (int i = 0; i < p; ++i) o += path_tiles<KIND>(i) * path_mt<KIND>(i) * w2n_tile_floats<KIND>(i);
  return o;
}
template <int KIND> constexpr int w2n_floats() { return w2n_path_off<KIND>(Spec<KIND>::NPATH); }
constexpr int W1N_FLOATS = 4 * 16 * 64;

// accumulators: one f32x4 per (out slot, m, M tile); slots are identified by the distinct out_off values
template <int KIND> constexpr int slot_rep(int p) {
  for (int j = 0; j < p; ++j)
    if (Spec<KIND>::P[j].out_off == Spec<KIND>::P[p].out_off) return j;
  return p;
}
template <int KIND> constexpr int acc_base(int p) {
  const int rep = slot_rep<KIND>(p);
  int o = 0;
  for (int i = 0; i < rep; ++i)
    if (slot_rep<KIND>(i) == i) o += d_out_of(Spec<KIND>::P[i].ztype) * path_mt<KIND>(i);
  return o;
}
template <int KIND> constexpr int n_acc4() {
  int o = 0;
  for (int i = 0; i < Spec<KIND>::NPATH; ++i)
    if (slot_rep<KIND>(i) == i) o += d_out_of(Spec<KIND>::P[i].ztype) * path_mt<KIND>(i);
  return o;
}

// path split (small batches): the out slots are dealt to two workgroups per node group, heaviest first onto the lighter side; a path
// belongs to the workgroup that owns its out slot, so every output channel is still summed by exactly one workgroup in the usual order
template <int KIND> constexpr int slot_tiles(int rep) {
  int t = 0;
  for (int i = 0; i < Spec<KIND>::NPATH; ++i)
    if (slot_rep<KIND>(i) == rep) t += path_tiles<KIND>(i) * (4 + path_mt<KIND>(i));     // ~ GEMM-A + GEMM-B work of the path's tiles
  return t;
}
template <int KIND> constexpr int split_owner(int p) {
  constexpr int NP = Spec<KIND>::NPATH;
  int owner[NP] = {}, load[2] = {0, 0};
  bool done[NP] = {};
  for (int it = 0; it < NP; ++it) {
    int best = -1;
    for (int i = 0; i < NP; ++i)
      if (slot_rep<KIND>(i) == i && !done[i] && (best < 0 || slot_tiles<KIND>(i) > slot_tiles<KIND>(best))) best = i;
    if (best < 0) break;
    done[best] = true;
    const int side = load[1] < load[0] ? 1 : 0;
    owner[best] = side;
    load[side] += slot_tiles<KIND>(best);
  }
  return owner[slot_rep<KIND>(p)];
}
// SIDE < 0: no split (every path is owned)
template <int KIND, int SIDE> constexpr bool owns(int p) { return SIDE < 0 || p >= Spec<KIND>::NPATH || split_owner<KIND>(p) == SIDE; }
template <int KIND, int SIDE> constexpr int next_owned(int p) {          // first owned path > p (NPATH if none)
  int n = p + 1;
  while (n < Spec<KIND>::NPATH && !owns<KIND, SIDE>(n)) ++n;
  return n;
}
template <int KIND> constexpr bool split_is_sound() {                   // both sides own something; paths of one out slot stay together
  if (next_owned<KIND, 0>(-1) >= Spec<KIND>::NPATH || next_owned<KIND, 1>(-1) >= Spec<KIND>::NPATH) return false;
  for (int i = 0; i < Spec<KIND>::NPATH; ++i)
    for (int j = 0; j < Spec<KIND>::NPATH; ++j)
      if (Spec<KIND>::P[i].out_off == Spec<KIND>::P[j].out_off && split_owner<KIND>(i) != split_owner<KIND>(j)) return false;
  return true;
}
static_assert(split_is_sound<0>() && split_is_sound<1>() && split_is_sound<2>() && split_is_sound<3>(), "path split partitions the out slots");
template <int KIND> constexpr int chan_owner(int ch) {
  for (int i = 0; i < Spec<KIND>::NPATH; ++i) {
    const PathC& pc = Spec<KIND>::P[i];
    if (ch >= pc.out_off && ch < pc.out_off + pc.mul_out * d_out_of(pc.ztype)) return split_owner<KIND>(i);
  }
  return 0;
}
template <int KIND> struct ChanOwner {
  unsigned char v[Spec<KIND>::D_OUT];
  constexpr ChanOwner() : v{} {
    for (int c = 0; c < Spec<KIND>::D_OUT; ++c) v[c] = (unsigned char)chan_owner<KIND>(c);
  }
};
template <int KIND> __device__ constexpr ChanOwner<KIND> kChanOwner{};

// attr index consumed by lane group g at GEMM1 K-step s (s < 15): five consecutive floats of each 20-float attr block
constexpr int attr_index(int s, int g) { return (s / 5) * 20 + 5 * g + (s % 5); }

// -------------------------------------------------------------------------------------------------------------------------
// device
// -------------------------------------------------------------------------------------------------------------------------
constexpr int LL = DP_EDGES_LLPAIR;

struct Grp { int pose, n_lo, n_loc; };      // the workgroup's node group: the same for every conv of a fused launch
struct Wg { int l0, p0, NL, NP, x0, En, pair0; };

// `a.slices` workgroup slots per pose are launched (enough for the batch's largest pose); a pose uses only the ceil(nodes / 16) groups
// it needs, spread evenly, and its other workgroups leave at once: a ragged batch (virtual screening: 15..45 atoms per ligand) does not
// pay 16-node tiles for the node groups only its largest pose has, and a pose's grouping depends on nothing but the pose itself
__device__ __forceinline__ Grp grp_of(const DpConvArgs& a, int bid) {
  Grp r;
  r.pose = bid / a.slices;
  const int slice = bid % a.slices;
  const int sc0 = a.scat_ptr[r.pose];
  const int nsc = a.scat_ptr[r.pose + 1] - sc0;
  const int ng = (nsc + NB - 1) / NB;
  if (slice >= ng) { r.n_lo = sc0; r.n_loc = 0; return r; }
  const int lo = (nsc * slice) / ng, hi = (nsc * (slice + 1)) / ng;
  r.n_lo = sc0 + lo;
  r.n_loc = hi - lo;
  return r;
}

template <int MODE> __device__ __forceinline__ Wg wg_of(const DpConvArgs& a, int pose) {
  Wg r;
  r.l0 = a.lig_ptr[pose];
  r.NL = a.lig_ptr[pose + 1] - r.l0;
  r.p0 = r.NP = r.x0 = r.pair0 = 0;
  if constexpr (MODE == LL) {
    r.pair0 = a.pair_ptr[pose];
  } else {
    r.p0 = a.ph_ptr[pose];
    r.NP = a.ph_ptr[pose + 1] - r.p0;
    r.x0 = a.x_ptr[pose];
  }
  r.En = MODE == DP_EDGES_P2L ? r.NP : r.NL;
  return r;
}

// edge `e` (0..En-1) of scatter node `node` (global id): feature row, gather row, attr rows
template <int MODE> __device__ __forceinline__ int feat_row(const Wg& r, int node, int e) {
  return MODE == DP_EDGES_P2L ? r.x0 + (node - r.l0) * r.NP + e : r.x0 + e * r.NP + (node - r.p0);
}
template <int MODE> __device__ __forceinline__ int gather_row(const Wg& r, int e) { return MODE == DP_EDGES_P2L ? r.p0 + e : r.l0 + e; }
// LL mode (ligand-ligand, pair-dense): slot e of scatter atom `node` is the atom pair (node, l0 + e); it carries up to two edges of
// the reference's multigraph - the radius edge and the bond edge (build_lig_conv_graph concatenates both, :721) - whose edge slots
// in the CSR-ordered feature arrays come from two pair maps (-1 = no such edge).  Both edges share the geometry (same sh, same
// gathered row), so T = sum_e hid[e] z[e] only needs the SUM of their hidden vectors: two GEMM1 evaluations, one GEMM-A.
template <int WHICH> __device__ __forceinline__ int pair_slot(const DpConvArgs& a, const Wg& r, int node, int e) {
  return (WHICH == 0 ? a.pair_a : a.pair_b)[r.pair0 + (node - r.l0) * r.NL + e];
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// z = sum_t xa[t] * s[t]: which gathered component and which staged factor each term uses, per (ztype, m)
struct Term { int xcomp; int scomp; float sign; };
template <int ZT> constexpr int n_terms() { return ZT == Z000 || ZT == Z101 || ZT == Z011 ? 1 : (ZT == Z111 ? 2 : 3); }
template <int ZT> constexpr bool needs_s() { return !(ZT == Z000 || ZT == Z101); }
constexpr bool needs_s_rt(int zt) { return !(zt == Z000 || zt == Z101); }
template <int ZT> __device__ __forceinline__ Term term_of(int t, int m) {
  // xcomp: offset inside the in-channel's d_in components; scomp: row of s_S
  if constexpr (ZT == Z000) return Term{0, 0, 1.f};
  else if constexpr (ZT == Z011) return Term{0, m, 1.f};
  else if constexpr (ZT == Z101) return Term{m, 0, 1.f};
  else if constexpr (ZT == Z110) return Term{t, t, 1.f};
  else if constexpr (ZT == Z111) {
    const int m1 = m == 2 ? 0 : m + 1, m2 = m == 0 ? 2 : m - 1;       // (m+1)%3, (m+2)%3
    return t == 0 ? Term{m1, m2, 1.f} : Term{m2, m1, -1.f};
  } else {
    const int a = t < m ? t : m, b = t < m ? m : t;
    return Term{t, 3 + (a == 0 ? b : (a == 1 ? 2 + b : 5)), 1.f};
  }
}

// raw gathered operands of column tile (path P, in-channel tile ut) for this lane's 8 (HALF: 4) edges; masking and sign are applied when
// the tile starts, so the loads can be issued a whole GEMM-B ahead
template <int KIND, int MODE, bool HALF, int P>
__device__ __forceinline__ void load_x(const DpConvArgs& a, const Wg& r, int et, int lane_l, int ut, float (&xr)[3][8]) {
  using S = Spec<KIND>;
  if constexpr (P < S::NPATH) {
    constexpr PathC pc = S::P[P];
    constexpr int ZT = pc.ztype;
    constexpr bool VEC = d_out_of(ZT) == 3;
    constexpr int NU = VEC ? 5 : 10, DIN = d_in_of(ZT), NTERM = n_terms<ZT>();
    const int g = lane_l >> 4, c = lane_l & 15;
    const int m = VEC ? (c >= 10 ? 2 : (c >= 5 ? 1 : 0)) : 0;
    const int ul = VEC ? c - 5 * m : c;
    const bool colv = VEC ? c < 15 : c < 10;
    const int u = colv ? NU * ut + ul : 0;
    // uniform base + 32-bit lane offset (saddr addressing): no 64-bit address arithmetic per load
    const float* __restrict__ base = a.gather_tab;
    unsigned coff[NTERM];
#pragma unroll
    for (int t = 0; t < NTERM; ++t) coff[t] = (unsigned)(pc.in_off + u * DIN + term_of<ZT>(t, m).xcomp);
#pragma unroll
    for (int j = 0; j < (HALF ? 4 : 8); ++j) {
      const int e = 32 * et + 16 * (j >> 2) + 4 * g + (j & 3);
      const unsigned roff = (unsigned)gather_row<MODE>(r, e < r.En ? e : 0) * (unsigned)a.gather_stride;
#pragma unroll
      for (int t = 0; t < NTERM; ++t) xr[t][j] = base[roff + coff[t]];
    }
  }
}

template <int KIND, int MODE, int SIDE, bool HALF, int P>
struct PathStage {
  // `xr`: raw gathered operands of this path's first tile, requested by the previous stage (or the pass) a GEMM-B earlier
  // SIDE >= 0 (path split): only the paths whose out slot this workgroup owns are instantiated; the operand prefetch chain - every stage
  // requests its successor's first tile - links the owned paths directly
  // HALF: the pass holds <= 16 edges per node: one quad step (4 K-steps) per node instead of two
  static __device__ __forceinline__ void run(const DpConvArgs& a, const Wg& r, int et, int q, int lane, const float (&hid)[NB][8],
                                             float* sT, const float* sS, f32x4 (&acc)[n_acc4<KIND>()], float (&xr)[3][8],
                                             NProfT& prof_prev) {
    using S = Spec<KIND>;
    if constexpr (P < S::NPATH) {
      static_assert(owns<KIND, SIDE>(P), "stages are chained over owned paths only");
      constexpr int PN = next_owned<KIND, SIDE>(P);
      constexpr PathC pc = S::P[P];
      constexpr int ZT = pc.ztype;
      constexpr bool VEC = d_out_of(ZT) == 3;
      constexpr int NU = VEC ? 5 : 10, NG = VEC ? 3 : 1, DIN = d_in_of(ZT), NTERM = n_terms<ZT>(), MT = path_mt<KIND>(P);
      constexpr int STEPS = NU * 4, AB = acc_base<KIND>(P);
      constexpr int NJ = HALF ? 4 : 8;                         // K-steps (edges of this lane group) per node
      constexpr int QSN = HALF ? NB : 2 * NB;                  // quad steps of the batch
      (void)DIN;
#pragma unroll 1
      for (int ut = 0; ut < pc.mul_in / NU; ++ut) {
        // lane-derived constants are recomputed from a laundered lane id in every tile: otherwise LICM hoists the per-path address
        // arithmetic of all paths out of the pass loop and spills it
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int g = lane_l >> 4, c = lane_l & 15;
        const int m = VEC ? (c >= 10 ? 2 : (c >= 5 ? 1 : 0)) : 0;
        const bool colv = VEC ? c < 15 : c < 10;
        const int lane_slot = (c << 2) | g;                       // GEMM-B view: node c, K-row group g
        // ---- gathered operand of this tile's column, for the lane's edges (same rows for every node of the batch) -----------
        float xa[NTERM][NJ];
        int soff[NTERM];
#pragma unroll
        for (int t = 0; t < NTERM; ++t) {
          const Term tm = term_of<ZT>(t, m);
          soff[t] = (tm.scomp * 16 + 4 * g) * 4;        // s_S[(nn, t2)][comp][4 g + rr]: a step's 9 comps are 576 contiguous bytes, so the
                                                        // <= 3 comps a lane group mixes sit in different quarters of the 64 banks
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int e = 32 * et + 16 * (j >> 2) + 4 * g + (j & 3);
            xa[t][j] = (colv && e < r.En) ? tm.sign * xr[t][j] : 0.f;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM-A: T[16 k x 16 cols] per node, through LDS ----------------------------------------------------------------
        // Software pipeline over the "quad steps" (full pass: node nn = qs / 2, edge quad t2 = qs % 2; half pass: nn = qs) of the batch:
        //     step qs:  request the sh factors of step qs + 2  |  4 MFMAs of step qs (B operand z_cur)
        //               ... with the VALU that turns the factors of step qs + 1 (requested one step earlier) into z_nxt in their shadows
        // so neither the LDS latency nor the z arithmetic sits between two MFMA groups of the same wave.  sched_group_barrier pins
        // the MFMA / VALU interleave inside a step, sched_barrier keeps steps apart (hoisting all LDS reads together spills).
        // Lanes of the unused 16th column store into a per-wave dummy slot instead of being masked off (no exec juggling).
        // store address of node nn's tile: [col][((nn << 2) | g) ^ (c & 7)][4] = two lane constants (even / odd nn) + an immediate nn * 64 B
        // (the swizzle flips bit 0 of nn for the lanes with c & 4), so a store costs no VALU; lanes c == 15 aim at the shared dummy row
        const int tb = (c >> 2) & 1;
        float* const tw_base = c < TCOLS ? sT + c * (NB * 16) + ((g ^ (c & 3)) << 2) : sT + (4 - q) * TSTRIDE + (g << 2);
        float* const tw_even = c < TCOLS ? tw_base + tb * 16 : tw_base;
        float* const tw_odd = c < TCOLS ? tw_base - tb * 16 : tw_base;
        auto tslot = [&](int nn) { return ((nn & 1) ? tw_odd : tw_even) + nn * 16; };
        if constexpr (needs_s<ZT>() && DP_NABL != 1) {
          const char* sbase = reinterpret_cast<const char*>(sS);
          // factors of quad step qs: rows (2 nn + t2) of s_S; a half pass uses the t2 = 0 rows only
          auto loadS = [&](int qs, f32x4 (&sv)[NTERM]) {
            const int row = HALF ? 2 * qs : qs;
#pragma unroll
            for (int t = 0; t < NTERM; ++t) sv[t] = *reinterpret_cast<const f32x4*>(sbase + row * (SCOMP * 64) + soff[t]);
          };
          // explicit ring buffers, no register copies between steps: the factors of step qs live in sv[qs % RD], its z in z[qs & 1]
          // (with a two-name rotation the allocator reused the registers a step's VALU still reads for that step's prefetch, which
          // forbids "reads first" and collapsed every second step's interleave into MFMA x4 followed by all of the VALU)
          auto makeZ1 = [&](int t2, int rr, const f32x4 (&sv)[NTERM]) {
            float zz = xa[0][4 * t2 + rr] * sv[0][rr];
#pragma unroll
            for (int t = 1; t < NTERM; ++t) zz = fmaf(xa[t][4 * t2 + rr], sv[t][rr], zz);
            return zz;
          };
          // ring depth / prefetch distance of the factor reads: two steps ahead, one for the three-term paths (12 fewer live registers
          // where the pressure peaks: fewer accumulator spills across path turns, ~1 % on the layer-3 launch)
          constexpr int RD = NTERM == 3 ? 2 : 3, PD = RD - 1;
          f32x4 sv[RD][NTERM];
          float z[2][4];
          loadS(0, sv[0]);
          if (PD > 1) loadS(1, sv[1]);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) z[0][rr] = makeZ1(0, rr, sv[0]);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 D[2];
#pragma unroll
          for (int qs = 0; qs < QSN; ++qs) {
            const int nn = HALF ? qs : qs >> 1, t2 = HALF ? 0 : qs & 1;
            const int t2n = HALF ? 0 : (qs + 1) & 1;                              // edge quad of the NEXT step
            if (qs + PD < QSN) loadS(qs + PD, sv[(qs + PD) % RD]);
            if (t2 == 0) D[nn & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
            // source order = issue order wanted: MFMA rr, then the z value of the next step's K-step rr (one mul / fma chain)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
#if DP_NABL == 5
              asm volatile("" ::"v"(z[qs & 1][rr]), "v"(hid[nn][4 * t2 + rr]));
#else
              D[nn & 1] = mfma16(hid[nn][4 * t2 + rr], z[qs & 1][rr], D[nn & 1]);
#endif
              if (qs + 1 < QSN) z[(qs + 1) & 1][rr] = makeZ1(t2n, rr, sv[(qs + 1) % RD]);
            }
            // the finished tile of the PREVIOUS node is stored one step late, behind this step's first MFMA: its last MFMA has long retired
#if DP_NABL == 2
            if (t2 == 0 && nn > 0) asm volatile("" ::"v"(D[(nn - 1) & 1]));
#else
            if (t2 == 0 && nn > 0) *reinterpret_cast<f32x4*>(tslot(nn - 1)) = D[(nn - 1) & 1];
#endif
            if (qs + PD < QSN) __builtin_amdgcn_sched_group_barrier(0x100, NTERM, 0);      // the prefetch reads go first
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              // one MFMA ...
              if (rr == 0 && t2 == 0 && nn > 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // (the previous node's tile store)
              if (qs + 1 < QSN) __builtin_amdgcn_sched_group_barrier(0x002, NTERM, 0);     // ... then one z value's mul / fma chain
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#if DP_NABL == 2
          asm volatile("" ::"v"(D[(NB - 1) & 1]));
#else
          *reinterpret_cast<f32x4*>(tslot(NB - 1)) = D[(NB - 1) & 1];
#endif
          __builtin_amdgcn_sched_barrier(0);
        } else {
          // z is the gathered input itself (sh_0 = 1): nothing to fetch or compute per node
#pragma unroll
          for (int nn = 0; nn < NB; ++nn) {
            f32x4 D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#if DP_NABL == 5
              asm volatile("" ::"v"(xa[0][j]), "v"(hid[nn][j]));
#else
              D = mfma16(hid[nn][j], xa[0][j], D);
#endif
            }
#if DP_NABL == 2
            asm volatile("" ::"v"(D));
#else
            *reinterpret_cast<f32x4*>(tslot(nn)) = D;
#endif
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        NPROF(2);
        __builtin_amdgcn_s_setprio(1);      // a wave in its (VALU-free) GEMM-B goes first: -3.4 % measured against no priority
        // ---- GEMM-B: out[w][node] += W2[w][(u,k)] T[(u,k)][node] -------------------------------------------------------------
#if DP_NABL == 4
        if (ut + 1 < pc.mul_in / NU) load_x<KIND, MODE, HALF, P>(a, r, et, lane_l, ut + 1, xr);
        else load_x<KIND, MODE, HALF, PN>(a, r, et, lane_l, 0, xr);
        if (false)
#endif
        {
          const float* wsrc = a.w2n + w2n_path_off<KIND>(P) + (size_t)(ut * MT) * w2n_tile_floats<KIND>(P) + (size_t)(q * 64 + lane_l) * STEPS;
          f32x4 wf[MT], wn[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) wf[mt] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)mt * w2n_tile_floats<KIND>(P));
          // i-outer / m-group-inner: one W2 fragment (per M tile) is live at a time and serves the three m components; the next
          // in-channel's fragment (L2) and the next step's T fragment (LDS) are requested before the current ones are consumed
          auto readT = [&](int st) {
            const int col = (VEC ? 5 * (st % NG) : 0) + st / NG;
#if DP_NABL == 2
            return f32x4{(float)col, xr[0][0], xr[0][1], xr[0][2]};
#else
            return *reinterpret_cast<const f32x4*>(sT + (size_t)col * (NB * 16) + ((lane_slot ^ (col & 7)) << 2));
#endif
          };
          f32x4 tv = readT(0), tn;
#pragma unroll
          for (int st = 0; st < NU * NG; ++st) {
            const int i = st / NG, grp = st % NG;
            if (st + 1 < NU * NG) tn = readT(st + 1);
            if (grp == 0 && i + 1 < NU) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) wn[mt] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)mt * w2n_tile_floats<KIND>(P) + 4 * (i + 1));
            }
            if (grp == 0 && i == NU - 2) {
              // the NEXT tile's gathered operands (this path's next in-channel tile, or the next path's first) travel during the tail of
              // this GEMM-B - behind the last W2 fragment request: vector-memory returns are in order, an earlier burst would delay it
              if (ut + 1 < pc.mul_in / NU) load_x<KIND, MODE, HALF, P>(a, r, et, lane_l, ut + 1, xr);
              else load_x<KIND, MODE, HALF, PN>(a, r, et, lane_l, 0, xr);
            }
#if DP_NABL == 8
            // TIMING EMULATION of GEMM-B on the bf16 pipe (results are wrong by construction): the step's four T values are split into three
            // bf16 parts each on the VALU (truncate, exact residual, truncate, exact residual, pack: what a consumer-side split costs), and the
            // step issues 3 MT 16x16x32 bf16 MFMAs (6 MT per two in-channels) instead of 4 MT fp32 ones
            {
              typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              unsigned hb[4], mb[4], lb[4];
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                const unsigned b0 = __builtin_bit_cast(unsigned, tv[rr]) & 0xFFFF0000u;
                const float r1 = tv[rr] - __builtin_bit_cast(float, b0);
                const unsigned b1 = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
                const float r2 = r1 - __builtin_bit_cast(float, b1);
                hb[rr] = b0; mb[rr] = b1; lb[rr] = __builtin_bit_cast(unsigned, r2);
              }
              const unsigned h01 = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), h23 = __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u);
              const unsigned m01 = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), m23 = __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u);
              const unsigned l01 = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), l23 = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
              const bf16x8_t B0 = __builtin_bit_cast(bf16x8_t, (u32x4){h01, h23, m01, m23});
              const bf16x8_t B1 = __builtin_bit_cast(bf16x8_t, (u32x4){m01, m23, l01, l23});
              const bf16x8_t B2 = __builtin_bit_cast(bf16x8_t, (u32x4){h01, h23, l01, l23});
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                const bf16x8_t A = __builtin_bit_cast(bf16x8_t, wf[mt]);
                f32x4& d = acc[AB + grp * MT + mt];
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B2, d, 0, 0, 0);
              }
            }
#else
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
#if DP_NABL == 3
                asm volatile("" ::"v"(wf[mt][rr]), "v"(tv[rr]));
#else
                acc[AB + grp * MT + mt] = mfma16(wf[mt][rr], tv[rr], acc[AB + grp * MT + mt]);
#endif
              }
#endif
            __builtin_amdgcn_sched_barrier(0);
            tv = tn;
            if (grp == NG - 1) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) wf[mt] = wn[mt];
            }
          }
        }
        __builtin_amdgcn_s_setprio(0);
        NPROF(3);
      }
      PathStage<KIND, MODE, SIDE, HALF, PN>::run(a, r, et, q, lane, hid, sT, sS, acc, xr, prof_prev);
    }
  }
};

// acc -> LDS partials [q][node][channel]
template <int KIND, int P>
struct AccStore {
  static __device__ __forceinline__ void run(const f32x4 (&acc)[n_acc4<KIND>()], float* red, int lane) {
    using S = Spec<KIND>;
    if constexpr (P < S::NPATH) {
      if constexpr (slot_rep<KIND>(P) == P) {
        constexpr PathC pc = S::P[P];
        constexpr int D = d_out_of(pc.ztype), MT = path_mt<KIND>(P), AB = acc_base<KIND>(P);
        const int g = lane >> 4, nn = lane & 15;
#pragma unroll
        for (int m = 0; m < D; ++m)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int w = 16 * mt + 4 * g + rr;
              if (w < pc.mul_out) red[nn * S::D_OUT + pc.out_off + w * D + m] = acc[AB + m * MT + mt][rr];
            }
      }
      AccStore<KIND, P + 1>::run(acc, red, lane);
    }
  }
};

// A fused launch: `n_ll` (0 / 1) ligand-ligand convs (slot 0) followed by `n_x` (0..2) dense cross convs of mode XMODE.  Every conv
// scatters onto the same node groups; conv[0] names the output / residual.  HAS_LL instantiates the pair-dense pass as well.
struct FusedArgs {
  int32_t n_ll, n_x;
  int32_t lean;              // every conv of the launch carries DpConvArgs.no_tail: the instantiation without the half-pass variant
  // conv split (small batches, with path_split): every conv of the launch gets its OWN workgroups (grid x n convs) and leaves the sum of its
  // four k-partials in part[conv][node - node0][out_stride] - no shifts, no residual; dp_conv_node_combine adds the convs up in a fixed
  // order.  One workgroup then runs one conv's passes instead of all convs' back to back: a latency-bound launch gets n x shorter.
  int32_t conv_split;
  float* part;
  int32_t part_stride;       // floats per conv in `part`
  int32_t node0;             // first scatter node of the launch (scat_ptr[0])
  DpConvArgs conv[3];
};

template <int KIND, bool HAS_LL, int XMODE, bool HALFOK>
__global__ __launch_bounds__(256, 2) void tp_conv_node_kernel(FusedArgs fa) {
  __shared__ __attribute__((aligned(16))) float s_T[4 * TSTRIDE + TDUMMY];     // wave-private T tiles + the shared dummy row; reused for the final reduction
  __shared__ __attribute__((aligned(16))) float s_S[NB * SCOMP * 32];         // per-edge sh factors of the current pass
  __shared__ float s_inv[NB];                                                  // 1 / degree of the group's nodes (current conv)
#if DP_NABL == 7                                                                // one workgroup per CU (a single wave per SIMD)
  __shared__ float s_pad[2048];
  if (threadIdx.x == 1023) s_pad[threadIdx.x] = 0.f;
#endif
  constexpr int SIDE = -1;
  constexpr int cx_only = -1;
  const int bid = (int)blockIdx.x;
#include "dp_conv_node_wg.h"
}

// DpConvArgs.path_split: two workgroups per node group, each owning half of the out slots (split_owner).  Both run GEMM1 (4 % of the MFMAs
// since the first-layer split) and their own paths' GEMM-A / GEMM-B: a conv's latency halves when the batch cannot fill the chip with one
// workgroup per node group; every output channel is computed exactly as in the unsplit launch (bit-identical).
template <int KIND, bool HAS_LL, int XMODE, bool HALFOK>
__global__ __launch_bounds__(256, 2) void tp_conv_node_split_kernel(FusedArgs fa) {
  __shared__ __attribute__((aligned(16))) float s_T[4 * TSTRIDE + TDUMMY];
  __shared__ __attribute__((aligned(16))) float s_S[NB * SCOMP * 32];
  __shared__ float s_inv[NB];
  const int groups = fa.conv[0].B * fa.conv[0].slices;
  const int bid_all = (int)(blockIdx.x >> 1);
  const int bid = fa.conv_split ? bid_all % groups : bid_all;
  const int cx_only = fa.conv_split ? bid_all / groups : -1;          // conv split: the one conv this workgroup computes
  if (blockIdx.x & 1u) {
    constexpr int SIDE = 1;
#include "dp_conv_node_wg.h"
  } else {
    constexpr int SIDE = 0;
#include "dp_conv_node_wg.h"
  }
}

// the lean instantiation exists for the unsplit kernel only: split launches (small batches) are latency-bound, not pipe-bound
template <int KIND, bool HAS_LL, int XMODE> int launch_seq(const FusedArgs& fa, hipStream_t st) {
  const DpConvArgs& a = fa.conv[0];
  if (a.path_split) {
    const int n_conv = fa.conv_split ? fa.n_ll + fa.n_x : 1;
#ifdef DP_SPLIT_LEAN      // experiment (profiles/r4_experiments.md): the path-split kernel without the half-pass variant, for batches with no short pass
    if (fa.lean && DP_NODE_HALF_PASS) {
      hipLaunchKernelGGL((tp_conv_node_split_kernel<KIND, HAS_LL, XMODE, false>), dim3((unsigned)(a.B * a.slices * 2 * n_conv)), dim3(256), 0, st, fa);
      return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((tp_conv_node_split_kernel<KIND, HAS_LL, XMODE, true>), dim3((unsigned)(a.B * a.slices * 2 * n_conv)), dim3(256), 0, st, fa);
  } else if (fa.lean && DP_NODE_HALF_PASS) {
    hipLaunchKernelGGL((tp_conv_node_kernel<KIND, HAS_LL, XMODE, false>), dim3((unsigned)(a.B * a.slices)), dim3(256), 0, st, fa);
  } else {
    hipLaunchKernelGGL((tp_conv_node_kernel<KIND, HAS_LL, XMODE, true>), dim3((unsigned)(a.B * a.slices)), dim3(256), 0, st, fa);
  }
  return (int)hipGetLastError();
}

// xmode: DP_EDGES_P2L / DP_EDGES_L2P of the cross convs (ignored when n_x == 0)
template <int KIND> int launch(const FusedArgs& fa, int xmode, hipStream_t st) {
  if (fa.n_ll) return launch_seq<KIND, true, DP_EDGES_P2L>(fa, st);            // ligand-side sequence: LL [+ P2L [+ P2L]]
  if (xmode == DP_EDGES_P2L) return launch_seq<KIND, false, DP_EDGES_P2L>(fa, st);
  return launch_seq<KIND, false, DP_EDGES_L2P>(fa, st);
}

}  // namespace dpn

int dpn_launch_k0(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpn_launch_k1(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpn_launch_k2(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpn_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
