// dp_conv_node.hip - node-side formulation of the fused tensor-product convolution for gfx950 (dense cross edge sets).
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
//   * GEMM1  hid^T[16 edges x 16 k] = attr[16 x 64] W1^T: its D registers ARE GEMM-A's A operand (edge order of GEMM-A's K steps is
//     chosen to be the D-fragment row order), kept in registers for the 16 nodes of the batch (128 VGPRs).
//   * GEMM-A per (column tile, node): 8 MFMAs; B operand z generated on the VALU from the gathered input (registers, loaded once per
//     tile: in the dense modes a lane's gathered rows are the same for every node) and sh-derived factors staged in LDS per batch.
//   * the 16x16 T tile goes through a wave-private, XOR-swizzled LDS buffer (one ds_write_b128 per node, ds_read_b128 back):
//     GEMM-A leaves columns on lanes, GEMM-B needs nodes on lanes.
//   * GEMM-B per column tile: A = packed W2 slice (dwordx4 loads from L2, shared by the three m components), B = T, N = 16 nodes,
//     accumulating into <= 40 persistent registers; the four waves' k-partial outputs are summed through LDS at the end.
//   Nodes with more than 32 edges run several 32-edge passes that accumulate into the same GEMM-B accumulators (GEMM-B is linear).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffphore_amd.h"
#include "dp_spec.h"

namespace dpn {
using namespace dp;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte load from a 4-byte aligned address

constexpr int NB = 16;        // nodes per workgroup = N of the GEMM-B MFMAs
constexpr int NPAIR = 2;      // nodes whose GEMM-A chains are interleaved
constexpr int TCOLS = 15;     // columns of a T tile that are stored (3 m-groups x 5 in-channels, or 10 in-channels of a scalar-output path)
constexpr int TSTRIDE = TCOLS * NB * 16 + 16;   // floats per wave: 15 column rows of 1 KiB + the dummy 16-byte slots of lanes c == 15
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

// attr index consumed by lane group g at GEMM1 K-step s (s < 15): five consecutive floats of each 20-float attr block
constexpr int attr_index(int s, int g) { return (s / 5) * 20 + 5 * g + (s % 5); }

// -------------------------------------------------------------------------------------------------------------------------
// host-side packing
// -------------------------------------------------------------------------------------------------------------------------
template <int KIND> void pack_w1n(const float* fc0_w, const float* fc0_b, float* w1n) {
  using S = Spec<KIND>;
  static_assert(S::HID == HIDN && S::N_ATTR == 60, "node-side kernel: layer convs only");
  for (int q = 0; q < 4; ++q)
    for (int s = 0; s < 16; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, k = 16 * q + (lane & 15);
        float v = 0.f;
        if (s < 15) {
          if (k < HIDN) v = fc0_w[k * 60 + attr_index(s, g)];
        } else if (g == 0) {                       // bias step: the attr operand is 1.0 in lane group 0
          v = k < HIDN ? fc0_b[k] : (k == HIDN ? 1.f : 0.f);
        }
        w1n[(q * 16 + s) * 64 + lane] = v;
      }
}

template <int KIND> void pack_w2n(const float* fc3_w, const float* fc3_b, float* w2n) {
  using S = Spec<KIND>;
  for (int p = 0; p < S::NPATH; ++p) {
    const PathC& pc = S::P[p];
    const double scale = __builtin_sqrt((double)(2 * (d_out_of(pc.ztype) == 3 ? 1 : 0) + 1) / (double)pc.fan_in) * cg_const(pc.ztype);
    const int nu = tile_nu<KIND>(p), steps = path_steps<KIND>(p), MT = path_mt<KIND>(p);
    for (int ut = 0; ut < path_tiles<KIND>(p); ++ut)
      for (int mt = 0; mt < MT; ++mt) {
        float* dst = w2n + w2n_path_off<KIND>(p) + (size_t)(ut * MT + mt) * w2n_tile_floats<KIND>(p);
        for (int q = 0; q < 4; ++q)
          for (int lane = 0; lane < 64; ++lane)
            for (int s = 0; s < steps; ++s) {
              const int g = lane >> 4, w = 16 * mt + (lane & 15);
              const int u = nu * ut + (s >> 2), k = 16 * q + 4 * g + (s & 3);
              float v = 0.f;
              if (w < pc.mul_out && k <= HIDN) {
                const int widx = pc.w_off + u * pc.mul_out + w;
                v = (float)(scale * (double)(k < HIDN ? fc3_w[widx * HIDN + k] : fc3_b[widx]));
              }
              dst[(size_t)(q * 64 + lane) * steps + s] = v;
            }
      }
  }
}

// -------------------------------------------------------------------------------------------------------------------------
// device
// -------------------------------------------------------------------------------------------------------------------------
struct Wg {
  int pose, n_lo, n_loc, l0, p0, NL, NP, x0, En;
};

template <int MODE> __device__ __forceinline__ Wg wg_of(const DpConvArgs& a) {
  Wg r;
  r.pose = blockIdx.x / a.slices;
  const int slice = blockIdx.x % a.slices;
  const int sc0 = a.scat_ptr[r.pose];
  const int nsc = a.scat_ptr[r.pose + 1] - sc0;
  const int lo = (int)(((long long)nsc * slice) / a.slices), hi = (int)(((long long)nsc * (slice + 1)) / a.slices);
  r.n_lo = sc0 + lo;
  r.n_loc = hi - lo;
  r.l0 = a.lig_ptr[r.pose];
  r.NL = a.lig_ptr[r.pose + 1] - r.l0;
  r.p0 = a.ph_ptr[r.pose];
  r.NP = a.ph_ptr[r.pose + 1] - r.p0;
  r.x0 = a.x_ptr[r.pose];
  r.En = MODE == DP_EDGES_P2L ? r.NP : r.NL;
  return r;
}

// edge `e` (0..En-1) of scatter node `node` (global id): feature row, gather row, attr rows
template <int MODE> __device__ __forceinline__ int feat_row(const Wg& r, int node, int e) {
  return MODE == DP_EDGES_P2L ? r.x0 + (node - r.l0) * r.NP + e : r.x0 + e * r.NP + (node - r.p0);
}
template <int MODE> __device__ __forceinline__ int gather_row(const Wg& r, int e) { return MODE == DP_EDGES_P2L ? r.p0 + e : r.l0 + e; }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// z = sum_t xa[t] * s[t]: which gathered component and which staged factor each term uses, per (ztype, m)
struct Term { int xcomp; int scomp; float sign; };
template <int ZT> constexpr int n_terms() { return ZT == Z000 || ZT == Z101 || ZT == Z011 ? 1 : (ZT == Z111 ? 2 : 3); }
template <int ZT> constexpr bool needs_s() { return !(ZT == Z000 || ZT == Z101); }
constexpr int sym_idx(int i, int m) {
  const int a = i < m ? i : m, b = i < m ? m : i;
  return a == 0 ? b : (a == 1 ? 2 + b : 5);
}
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

template <int KIND, int MODE, int P>
struct PathStage {
  static __device__ __forceinline__ void run(const DpConvArgs& a, const Wg& r, int et, int q, int lane, const float (&hid)[NB][8],
                                             float* sT, const float* sS, f32x4 (&acc)[n_acc4<KIND>()]) {
    using S = Spec<KIND>;
    if constexpr (P < S::NPATH) {
      constexpr PathC pc = S::P[P];
      constexpr int ZT = pc.ztype;
      constexpr bool VEC = d_out_of(ZT) == 3;
      constexpr int NU = VEC ? 5 : 10, NG = VEC ? 3 : 1, DIN = d_in_of(ZT), NTERM = n_terms<ZT>(), MT = path_mt<KIND>(P);
      constexpr int STEPS = NU * 4, AB = acc_base<KIND>(P);
#pragma unroll 1
      for (int ut = 0; ut < pc.mul_in / NU; ++ut) {
        // lane-derived constants are recomputed from a laundered lane id in every tile: otherwise LICM hoists the per-path address
        // arithmetic of all paths out of the pass loop and spills it
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int g = lane_l >> 4, c = lane_l & 15;
        const int m = VEC ? (c >= 10 ? 2 : (c >= 5 ? 1 : 0)) : 0;
        const int ul = VEC ? c - 5 * m : c;
        const bool colv = VEC ? c < 15 : c < 10;
        const int lane_slot = (c << 2) | g;                       // GEMM-B view: node c, K-row group g
        // ---- gathered operand of this tile's column, for the lane's 8 edges (same rows for every node of the batch) -----------
        float xa[NTERM][8];
        int soff[NTERM];
        {
          const int u = NU * ut + ul;
#pragma unroll
          for (int t = 0; t < NTERM; ++t) {
            const Term tm = term_of<ZT>(t, m);
            soff[t] = (tm.scomp * 32 + 4 * g) * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int e = 32 * et + 16 * (j >> 2) + 4 * g + (j & 3);
              const bool ok = colv && e < r.En;
              const float v = a.gather_tab[(size_t)gather_row<MODE>(r, ok ? e : 0) * a.gather_stride + pc.in_off + u * DIN + tm.xcomp];
              xa[t][j] = ok ? tm.sign * v : 0.f;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM-A: T[16 k x 16 cols] per node, through LDS ----------------------------------------------------------------
        // NPAIR nodes at a time: their MFMA chains are independent, so they issue back to back (a lone 8-deep chain on one
        // accumulator pays the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 per step instead of the 32-cycle issue rate).
        // Lanes of the unused 16th column store into a per-wave dummy slot instead of being masked off (no exec juggling).
        const int twr = c < TCOLS ? c * (NB * 16) : TCOLS * (NB * 16);
#pragma unroll
        for (int nn0 = 0; nn0 < NB; nn0 += NPAIR) {
          f32x4 D[NPAIR];
#pragma unroll
          for (int pi = 0; pi < NPAIR; ++pi) D[pi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            float z[NPAIR][4];
#pragma unroll
            for (int pi = 0; pi < NPAIR; ++pi) {
              if constexpr (needs_s<ZT>()) {
                f32x4 sv[NTERM];
#pragma unroll
                for (int t = 0; t < NTERM; ++t)
                  sv[t] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(sS) + (nn0 + pi) * (SCOMP * 32 * 4) + t2 * 64 + soff[t]);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  float zz = xa[0][4 * t2 + rr] * sv[0][rr];
#pragma unroll
                  for (int t = 1; t < NTERM; ++t) zz = fmaf(xa[t][4 * t2 + rr], sv[t][rr], zz);
                  z[pi][rr] = zz;
                }
              } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) z[pi][rr] = xa[0][4 * t2 + rr];
              }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
              for (int pi = 0; pi < NPAIR; ++pi) D[pi] = mfma16(hid[nn0 + pi][4 * t2 + rr], z[pi][rr], D[pi]);
          }
#pragma unroll
          for (int pi = 0; pi < NPAIR; ++pi)
            *reinterpret_cast<f32x4*>(sT + twr + (c < TCOLS ? (((((nn0 + pi) << 2) | g) ^ (c & 7)) << 2) : (g << 2))) = D[pi];
          __builtin_amdgcn_sched_barrier(0);       // keep the node bodies apart: hoisting all their LDS reads together spills
        }
        // ---- GEMM-B: out[w][node] += W2[w][(u,k)] T[(u,k)][node] -------------------------------------------------------------
        {
          const float* wsrc = a.w2n + w2n_path_off<KIND>(P) + (size_t)(ut * MT) * w2n_tile_floats<KIND>(P) + (size_t)(q * 64 + lane_l) * STEPS;
          // i-outer / m-group-inner: one W2 fragment (per M tile) is live at a time and serves the three m components; the next
          // in-channel's fragment is requested before the current one is consumed
          f32x4 wf[MT], wn[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) wf[mt] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)mt * w2n_tile_floats<KIND>(P));
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            if (i + 1 < NU) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) wn[mt] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)mt * w2n_tile_floats<KIND>(P) + 4 * (i + 1));
            }
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
              const int col = (VEC ? 5 * grp : 0) + i;
              const f32x4 tv = *reinterpret_cast<const f32x4*>(sT + (size_t)col * (NB * 16) + ((lane_slot ^ (col & 7)) << 2));
#pragma unroll
              for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[AB + grp * MT + mt] = mfma16(wf[mt][rr], tv[rr], acc[AB + grp * MT + mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) wf[mt] = wn[mt];
          }
        }
      }
      PathStage<KIND, MODE, P + 1>::run(a, r, et, q, lane, hid, sT, sS, acc);
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

template <int KIND, int MODE>
__global__ __launch_bounds__(256, 2) void tp_conv_node_kernel(DpConvArgs a) {
  using S = Spec<KIND>;
  __shared__ __attribute__((aligned(16))) float s_T[4 * TSTRIDE];              // wave-private T tiles (+ a dummy slot); reused for the final reduction
  __shared__ __attribute__((aligned(16))) float s_S[NB * SCOMP * 32];         // per-edge sh factors of the current pass
  static_assert(4 * NB * S::D_OUT <= 4 * TSTRIDE, "reduction scratch fits in the T buffers");
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, g = lane >> 4, c = lane & 15;
  const Wg r = wg_of<MODE>(a);
  if (r.n_loc <= 0) return;
  float* sT = s_T + q * TSTRIDE;

  f32x4 acc[n_acc4<KIND>()];
#pragma unroll
  for (int i = 0; i < n_acc4<KIND>(); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int n_et = (r.En + 31) >> 5;
#pragma unroll 1
  for (int et = 0; et < n_et; ++et) {
    __syncthreads();                               // readers of the previous pass's factors are done
    for (int idx = tid; idx < NB * 32; idx += 256) {
      const int nn = idx >> 5, el = idx & 31, e = 32 * et + el;
      const bool ok = nn < r.n_loc && e < r.En;
      float v[SCOMP];
#pragma unroll
      for (int i = 0; i < SCOMP; ++i) v[i] = 0.f;
      if (ok) {
        const float* ps = a.edge_sh + (size_t)feat_row<MODE>(r, r.n_lo + nn, e) * a.sh_stride;
        const float cc = 0.57735026918962576451f * ps[6];
        v[0] = ps[1]; v[1] = ps[2]; v[2] = ps[3];
        v[3] = -cc - ps[8]; v[4] = ps[5]; v[5] = ps[4]; v[6] = 2.f * cc; v[7] = ps[7]; v[8] = -cc + ps[8];
      }
#pragma unroll
      for (int i = 0; i < SCOMP; ++i) s_S[(nn * SCOMP + i) * 32 + el] = v[i];
    }
    __syncthreads();

    // ---- GEMM1: hidden rows 16q..16q+15 of the 32 edges of every node of the batch ---------------------------------------------
    float hid[NB][8];
    float w1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) w1[s] = a.w1n[(q * 16 + s) * 64 + lane];
#pragma unroll
    for (int nn = 0; nn < NB; ++nn) {
      const int node = r.n_lo + (nn < r.n_loc ? nn : 0);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        int e = 32 * et + 16 * t2 + c;               // GEMM1 view: lane (g, c) feeds attr[edge c][group g]
        e = e < r.En ? e : 0;
        const float* pe = a.edge_emb + (size_t)feat_row<MODE>(r, node, e) * 20 + 5 * g;
        const int r1 = MODE == DP_EDGES_P2L ? node : r.l0 + e, r2 = MODE == DP_EDGES_P2L ? r.p0 + e : node;
        const float* p1 = a.attr1_tab + (size_t)r1 * a.attr1_stride + 5 * g;
        const float* p2 = a.attr2_tab + (size_t)r2 * a.attr2_stride + 5 * g;
        float av[15];
        {
          const f4u v0 = *reinterpret_cast<const f4u*>(pe), v1 = *reinterpret_cast<const f4u*>(p1), v2 = *reinterpret_cast<const f4u*>(p2);
          av[0] = v0.x; av[1] = v0.y; av[2] = v0.z; av[3] = v0.w; av[4] = pe[4];
          av[5] = v1.x; av[6] = v1.y; av[7] = v1.z; av[8] = v1.w; av[9] = p1[4];
          av[10] = v2.x; av[11] = v2.y; av[12] = v2.z; av[13] = v2.w; av[14] = p2[4];
        }
        f32x4 D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 15; ++s) D = mfma16(av[s], w1[s], D);
        D = mfma16(g == 0 ? 1.f : 0.f, w1[15], D);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hid[nn][4 * t2 + rr] = fmaxf(D[rr], 0.f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    PathStage<KIND, MODE, 0>::run(a, r, et, q, lane, hid, sT, s_S, acc);
  }

  // ---- sum the four k-partials, mean, BatchNorm affine, residual stream ------------------------------------------------------------
  __syncthreads();
  AccStore<KIND, 0>::run(acc, s_T + q * (NB * S::D_OUT), lane);
  __syncthreads();
  const float inv = 1.f / (float)max(r.En, 1);
  for (int i = tid; i < r.n_loc * S::D_OUT; i += 256) {
    const int n = i / S::D_OUT, ch = i - n * S::D_OUT;
    float v = ((s_T[i] + s_T[NB * S::D_OUT + i]) + s_T[2 * NB * S::D_OUT + i]) + s_T[3 * NB * S::D_OUT + i];
    v = v * inv;
    v = v * a.bn_scale[ch] + a.bn_shift[ch];
    const int node = r.n_lo + n;
    float* o = a.out + (size_t)node * a.out_stride + ch;
    if (a.accumulate) {
      *o += v;
    } else {
      if (a.resid != nullptr && ch < a.resid_width) v += a.resid[(size_t)node * a.resid_stride + ch];
      *o = v;
    }
  }
}

template <int KIND> int launch(const DpConvArgs& a, hipStream_t st) {
  dim3 grid((unsigned)(a.B * a.slices));
  if (a.mode == DP_EDGES_P2L) hipLaunchKernelGGL((tp_conv_node_kernel<KIND, DP_EDGES_P2L>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((tp_conv_node_kernel<KIND, DP_EDGES_L2P>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

}  // namespace dpn

extern "C" {

int dp_conv_node_max_nodes(void) { return dpn::NB; }
int dp_conv_w1n_floats(int32_t kind) { return kind >= 0 && kind <= 3 ? dpn::W1N_FLOATS : DP_E_BADARG; }
int dp_conv_w2n_floats(int32_t kind) {
  switch (kind) {
    case 0: return dpn::w2n_floats<0>(); case 1: return dpn::w2n_floats<1>();
    case 2: return dpn::w2n_floats<2>(); case 3: return dpn::w2n_floats<3>();
    default: return DP_E_BADARG;
  }
}

int dp_pack_conv_weights_node(int32_t kind, const float* fc0_w, const float* fc0_b, const float* fc3_w, const float* fc3_b, float* w1n,
                              float* w2n) {
  if (!fc0_w || !fc0_b || !fc3_w || !fc3_b || !w1n || !w2n) return DP_E_BADARG;
  switch (kind) {
#define DP_CASE(K) case K: dpn::pack_w1n<K>(fc0_w, fc0_b, w1n); dpn::pack_w2n<K>(fc3_w, fc3_b, w2n); return DP_OK;
    DP_CASE(0) DP_CASE(1) DP_CASE(2) DP_CASE(3)
#undef DP_CASE
    default: return DP_E_UNSUPPORTED;
  }
}

// node-side launch of one layer conv (kinds 0-3, dense cross edge sets only); called by dp_tp_conv when DpConvArgs.algo asks for it
int dp_tp_conv_node(const DpConvArgs* a, void* stream) {
  if (!a || a->B <= 0 || a->slices <= 0 || !a->scat_ptr || !a->lig_ptr || !a->ph_ptr || !a->x_ptr || !a->gather_tab || !a->attr1_tab ||
      !a->attr2_tab || !a->edge_emb || !a->edge_sh || !a->w1n || !a->w2n || !a->bn_scale || !a->bn_shift || !a->out)
    return DP_E_BADARG;
  if (a->mode != DP_EDGES_P2L && a->mode != DP_EDGES_L2P) return DP_E_UNSUPPORTED;
  if (a->kind < 0 || a->kind > 3 || a->sh_stride < 9) return DP_E_UNSUPPORTED;
  if (a->max_scatter <= 0) return DP_E_BADARG;
  if ((a->max_scatter + a->slices - 1) / a->slices > dpn::NB) return DP_E_TOOLARGE;
  hipStream_t st = (hipStream_t)stream;
  switch (a->kind) {
    case 0: return dpn::launch<0>(*a, st);
    case 1: return dpn::launch<1>(*a, st);
    case 2: return dpn::launch<2>(*a, st);
    default: return dpn::launch<3>(*a, st);
  }
}

}  // extern "C"
