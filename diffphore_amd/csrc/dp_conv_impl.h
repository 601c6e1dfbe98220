// dp_conv_impl.h - fused tensor-product convolution for gfx950 (MI355X), edge-side formulation: kernels, launchers and host packers as templates
// over the layer kind.  Instantiated by dp_conv_k0..k5.hip (one translation unit per kind, so that they compile side by side: the single-file build
// took 5.4 minutes) and used by dp_conv.hip (C entry points, packers).
//
// One launch = one TensorProductConvLayer.forward of the reference (score_model_phore.py:134-149):
//     w   = fc(edge_attr_)                      Linear(N_ATTR->HID), ReLU, Linear(HID->numel)      [never materialised]
//     msg = tp(node_attr[edge_dst], edge_sh, w) e3nn FullyConnectedTensorProduct, per-edge weights
//     out = scatter_mean(msg, edge_src) ; out = BatchNorm_eval(out) ; (+ residual stream)
//
// Mapping to CDNA4
//   * a wave owns 32 edges (the N dimension of v_mfma_f32_32x32x2_f32); lanes l and l+32 share edge l&31.
//   * GEMM1  hidden[64 x 32 edges] = W1p[64 x K1] * attr[K1 x 32]          (2 x k1 MFMAs)
//     its D registers ARE the B operand of GEMM2 (K index permuted at pack time so no lane movement is needed).
//   * GEMM2  per 32-row weight tile: Wt[32 x 32 edges] = W2p[32 x K2] * hidden[K2 x 32]   (K2 MFMAs of 64 cycles)
//     rows of a tile are 16 "triples" (path, in-channel pair q, out channel w) x 2 lane halves; half h holds
//     in-channel u = h*mul_in/2 + q.  The per-edge weight therefore lives in a register for exactly one use.
//   * contraction on the VALU: acc[out_off + w*d + k] += Wt * z(path, u, k), z = CG-contracted gathered input x sh
//     (path coefficient and CG norm folded into W2p).  Fully unrolled: every register index is static.
//   * halves are summed with one ds_swizzle/shfl, messages staged in LDS, a per-workgroup ordered segmented sum
//     reduces edges (sorted by scatter node) into LDS node accumulators -> mean -> BN affine -> global.
//   The [E, numel] weight tensor of the reference (2.4-8.8 KB per edge) never exists in memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "../../include/diffphore_amd.h"
#include "dp_spec.h"
#include "dp_f16.h"

#ifndef DP_ABLATE
#define DP_ABLATE 0
#endif
#ifndef DP_NODE_AUTO_MIN_POSES          // DP_ALGO_AUTO picks the node-side kernel from this batch size on
#define DP_NODE_AUTO_MIN_POSES 1         // profiles/r3_batch_sweep.csv: with path split + conv split the node-side kernel wins at every batch size
                                         // (1 pose: 0.86 vs 1.06 ms/step; round 2, before the conv split: from 20 poses on)
#endif
#if defined(DP_PROF) && !defined(DP_CONV_ONE_TU)
#error "-DDP_PROF needs -DDP_CONV_ONE_TU: one device-side counter array, every kind instantiated in dp_conv.hip"
#endif
#ifdef DP_PROF   // cycle-stamp breakdown of the bf16 kernel (ablation builds only; tools/bench_conv.py --prof)
__device__ unsigned long long dp_prof[16];
__shared__ unsigned long long s_prof_prev;
__device__ __forceinline__ void prof_mark(int seg) {
  if (threadIdx.x == 0) {
    const unsigned long long now = clock64();
    if (seg >= 0) atomicAdd(&dp_prof[seg], now - s_prof_prev);
    s_prof_prev = now;
  }
}
#define PROF_MARK(seg) prof_mark(seg)
extern "C" int dp_debug_prof(unsigned long long* host16, int reset) {
  if (host16 && hipMemcpyFromSymbol(host16, HIP_SYMBOL(dp_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(dp_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#else
#define PROF_MARK(seg)
#endif

namespace dp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVES = 4;            // waves per workgroup
constexpr int CHUNK = WAVES * 32;   // edges per workgroup iteration
constexpr int MAX_NODES = 48;       // scatter nodes per workgroup (host picks `slices` accordingly)

template <int KIND> struct Geo {
  using S = Spec<KIND>;
  static constexpr int K1 = k1_steps<KIND>();
  static constexpr int K1P = k1_steps_pad<KIND>();
  static constexpr int K2S = k2_steps<KIND>();       // fp32 K-steps actually executed per tile
  static constexpr int K2 = k2_steps_pad<KIND>();    // padded to whole dwordx4 operand loads
  static constexpr int NT = n_tiles<KIND>();
  static constexpr int NTRI = n_triples<KIND>();
  static constexpr int ABLK = S::N_ATTR / 20;        // attr blocks of 20 (edge emb, attr1, attr2)
  static constexpr int W1P_FLOATS = 2 * K1P * 64;
  static constexpr int W2P_FLOATS = NT * K2 * 64;
  static constexpr int MSG_LD = S::D_OUT + 1;        // +1: conflict-free column reads in the segmented sum
};

// ---------------------------------------------------------------------------------------------------------------
// host-side packing (single source of truth for the operand layouts the kernel assumes)
// ---------------------------------------------------------------------------------------------------------------
// attr index k consumed by lane half `half` at GEMM1 K-step s:   block = s/10 ; k = block*20 + half*10 + s%10
// (each lane reads 10 contiguous floats per 20-float attr block).  Last real step: bias (half 0) / zero (half 1).
template <int KIND> void pack_w1(const float* fc0_w, const float* fc0_b, float* w1p) {
  using G = Geo<KIND>;
  using S = Spec<KIND>;
  for (int i = 0; i < G::W1P_FLOATS; ++i) w1p[i] = 0.f;
  for (int t = 0; t < 2; ++t)
    for (int s = 0; s < G::K1; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int row = t * 32 + (lane & 31), half = lane >> 5;
        const int unit = unit_of_row<KIND>(row);
        float v = 0.f;
        if (s < S::N_ATTR / 2) {
          int k = (s / 10) * 20 + half * 10 + (s % 10);
          if (unit >= 0 && unit < S::HID) v = fc0_w[unit * S::N_ATTR + k];
        } else if (half == 0) {                       // bias step: attr value 1.0
          if (unit >= 0 && unit < S::HID) v = fc0_b[unit];
          else if (unit == S::HID) v = 1.f;           // constant-one hidden unit that carries fc.3's bias through GEMM2
        }
        // layout [t][s/4][lane][s%4]  -> one dwordx4 per lane per 4 MFMAs
        w1p[((t * (G::K1P / 4) + s / 4) * 64 + lane) * 4 + (s & 3)] = v;
      }
}

template <int KIND> void pack_w2(const float* fc3_w, const float* fc3_b, float* w2p) {
  using G = Geo<KIND>;
  using S = Spec<KIND>;
  for (int i = 0; i < G::W2P_FLOATS; ++i) w2p[i] = 0.f;
  for (int tile = 0; tile < G::NT; ++tile)
    for (int sg = 0; sg < G::K2; ++sg)
      for (int lane = 0; lane < 64; ++lane) {
        int c = lane & 31, half = lane >> 5;           // c = D row of the tile this lane's A operand feeds
        int r = (c & 3) + 4 * (c >> 3), chalf = (c >> 2) & 1;
        int n = tile * 16 + r;
        Triple tr = triple_at<KIND>(n);
        float v = 0.f;
        if (tr.valid) {
          const PathC& p = S::P[tr.p];
          int u = chalf * (p.mul_in / 2) + tr.q;
          int widx = p.w_off + u * p.mul_out + tr.w;
          double scale = __builtin_sqrt((double)(2 * (d_out_of(p.ztype) == 3 ? 1 : 0) + 1) / (double)p.fan_in) * cg_const(p.ztype);
          const int unit = unit_of_row<KIND>(hid_row(sg, half));
          if (unit >= 0 && unit < S::HID) v = (float)(scale * (double)fc3_w[widx * S::HID + unit]);
          else if (unit == S::HID) v = (float)(scale * (double)fc3_b[widx]);
        }
        w2p[((tile * (G::K2 / 4) + sg / 4) * 64 + lane) * 4 + (sg & 3)] = v;
      }
}

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
struct EdgeRef {
  int scat_local;   // scatter node index within the workgroup's node range, -1 = padding slot
  int gather;       // row in gather_tab
  int a1, a2;       // rows in attr1_tab / attr2_tab
  int feat;         // row in edge_emb / edge_sh
};

typedef float vf4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) vf4* gptr4;   // global (not flat) pointer to packed weights

template <int KIND> constexpr int NACC = n_acc<KIND>() > 0 ? n_acc<KIND>() : 1;

template <int KIND, int N>
__device__ __forceinline__ void contract_one(float wv, const float (&xr)[Spec<KIND>::D_IN / 2], const float (&sh)[Spec<KIND>::SH_DIM],
                                             const float (&M)[9], float (&acc)[NACC<KIND>]) {
  constexpr Triple T = triple_at<KIND>(N);
  if constexpr (T.valid) {
    constexpr PathC p = Spec<KIND>::P[T.p];
    constexpr int din = d_in_of(p.ztype);
    constexpr int xo = p.in_off / 2 + T.q * din;      // this half's slice of the in-slot starts at in_off/2
    constexpr int co = compact_off<KIND>(T.p);        // accumulator base of the out slot within its phase
    if constexpr (p.ztype == Z000) {
      acc[co + T.w] += wv * (xr[xo] * sh[p.sh_off]);
    } else if constexpr (p.ztype == Z011) {
      const float t = wv * xr[xo];
      acc[co + 3 * T.w + 0] += t * sh[p.sh_off + 0];
      acc[co + 3 * T.w + 1] += t * sh[p.sh_off + 1];
      acc[co + 3 * T.w + 2] += t * sh[p.sh_off + 2];
    } else if constexpr (p.ztype == Z101) {
      const float t = wv * sh[p.sh_off];
      acc[co + 3 * T.w + 0] += t * xr[xo + 0];
      acc[co + 3 * T.w + 1] += t * xr[xo + 1];
      acc[co + 3 * T.w + 2] += t * xr[xo + 2];
    } else if constexpr (p.ztype == Z110) {
      const float d = xr[xo] * sh[p.sh_off] + xr[xo + 1] * sh[p.sh_off + 1] + xr[xo + 2] * sh[p.sh_off + 2];
      acc[co + T.w] += wv * d;
    } else if constexpr (p.ztype == Z111) {
      const float a0 = xr[xo], a1 = xr[xo + 1], a2 = xr[xo + 2];
      const float b0 = sh[p.sh_off], b1 = sh[p.sh_off + 1], b2 = sh[p.sh_off + 2];
      acc[co + 3 * T.w + 0] += wv * (a1 * b2 - a2 * b1);
      acc[co + 3 * T.w + 1] += wv * (a2 * b0 - a0 * b2);
      acc[co + 3 * T.w + 2] += wv * (a0 * b1 - a1 * b0);
    } else {  // Z121: z[k] = sum_i x[i] M[i][k], M symmetric
      const float a0 = xr[xo], a1 = xr[xo + 1], a2 = xr[xo + 2];
      acc[co + 3 * T.w + 0] += wv * (a0 * M[0] + a1 * M[3] + a2 * M[6]);
      acc[co + 3 * T.w + 1] += wv * (a0 * M[1] + a1 * M[4] + a2 * M[7]);
      acc[co + 3 * T.w + 2] += wv * (a0 * M[2] + a1 * M[5] + a2 * M[8]);
    }
    // Pin the update in program order.  Without this the DAG scheduler sinks every tile's contraction below all the
    // MFMA chains of the phase (the accumulators are only read at the end) and spills each 16-register D tile.
    if constexpr (d_out_of(p.ztype) == 1) {
      asm volatile("" : "+v"(acc[co + T.w]));
    } else {
      asm volatile("" : "+v"(acc[co + 3 * T.w + 0]), "+v"(acc[co + 3 * T.w + 1]), "+v"(acc[co + 3 * T.w + 2]));
    }
  }
}

template <int KIND, int TILE, int... R>
__device__ __forceinline__ void contract_tile(const f32x16& D, const float (&xr)[Spec<KIND>::D_IN / 2],
                                              const float (&sh)[Spec<KIND>::SH_DIM], const float (&M)[9],
                                              float (&acc)[NACC<KIND>], std::integer_sequence<int, R...>) {
  (contract_one<KIND, TILE * 16 + R>(D[R], xr, sh, M, acc), ...);
}

// registers [R0, R0 + sizeof...(R)) of a tile's D fragment only (the software-pipelined one-product stage contracts a tile in pieces, between
// the MFMAs of the next one)
template <int KIND, int TILE, int R0, int... R>
__device__ __forceinline__ void contract_part(const f32x16& D, const float (&xr)[Spec<KIND>::D_IN / 2],
                                              const float (&sh)[Spec<KIND>::SH_DIM], const float (&M)[9],
                                              float (&acc)[NACC<KIND>], std::integer_sequence<int, R...>) {
  (contract_one<KIND, TILE * 16 + R0 + R>(D[R0 + R], xr, sh, M, acc), ...);
}

// One pipeline stage: prefetch W2p of tile T+1, run the MFMA chain of tile T, contract tile T-1 on the VALU.
// The MFMA chain and the contraction are independent, so the VALU work can sit in the 64-cycle MFMA issue shadows;
// the sched_barrier keeps stages from smearing into each other (register pressure).
template <int KIND> struct WTile { vf4 v[Geo<KIND>::K2 / 4]; };

// `wp` is a wave-uniform pointer advanced tile by tile and laundered through an empty asm, so no per-tile address is
// hoisted out of the chunk loop (hipcc otherwise materialises NT*K2/4 64-bit addresses and spills them).
template <int KIND>
__device__ __forceinline__ void load_wtile(gptr4 wp, int lane, WTile<KIND>& w) {
  using G = Geo<KIND>;
#pragma unroll
  for (int g = 0; g < G::K2 / 4; ++g) w.v[g] = wp[g * 64 + lane];
}

// MFMA chain of one tile; as each dwordx4 of W operands is consumed, the same registers are refilled with the next
// tile's operands (single W buffer: the refill has a whole tile, ~2k cycles, to land).
template <int KIND, bool PREFETCH>
__device__ __forceinline__ f32x16 mfma_tile(WTile<KIND>& w, gptr4 wp_next, int lane, const float (&hid)[32]) {
  using G = Geo<KIND>;
  f32x16 D = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < G::K2 / 4; ++g) {
    if (4 * g + 0 < G::K2S) D = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v[g].x, hid[4 * g + 0], D, 0, 0, 0);
    if (4 * g + 1 < G::K2S) D = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v[g].y, hid[4 * g + 1], D, 0, 0, 0);
    if (4 * g + 2 < G::K2S) D = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v[g].z, hid[4 * g + 2], D, 0, 0, 0);
    if (4 * g + 3 < G::K2S) D = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v[g].w, hid[4 * g + 3], D, 0, 0, 0);
    if constexpr (PREFETCH) w.v[g] = wp_next[g * 64 + lane];
  }
  return D;
}

// tiles [TILE, TEND) of one phase.  No intra-wave software pipelining: the kernel runs two waves per SIMD, so one
// wave's VALU contraction overlaps the other wave's 64-cycle MFMAs, and each D tile is consumed as soon as it lands.
template <int KIND, int TILE, int TEND>
struct Stage {
  static __device__ __forceinline__ void run(gptr4 wp, int lane, WTile<KIND>& w, const float (&hid)[32],
                                             const float (&xr)[Spec<KIND>::D_IN / 2], const float (&sh)[Spec<KIND>::SH_DIM],
                                             const float (&M)[9], float (&acc)[NACC<KIND>]) {
    using G = Geo<KIND>;
    if constexpr (TILE < TEND) {
      wp += (G::K2 / 4) * 64;
      asm volatile("" : "+s"(wp));
      const f32x16 D = mfma_tile<KIND, (TILE + 1 < TEND) && DP_ABLATE != 13>(w, wp, lane, hid);
#if DP_ABLATE == 11
      for (int r_ = 0; r_ < 16; ++r_) asm volatile("" ::"v"(D[r_]));
#else
      contract_tile<KIND, TILE>(D, xr, sh, M, acc, std::make_integer_sequence<int, 16>{});
#endif
      __builtin_amdgcn_sched_barrier(0);
      Stage<KIND, TILE + 1, TEND>::run(wp, lane, w, hid, xr, sh, M, acc);
    }
  }
};

// phase-local accumulator index -> output channel
template <int KIND, int PH> struct ChanLut {
  static constexpr int N = n_acc_phase<KIND>(PH) > 0 ? n_acc_phase<KIND>(PH) : 1;
  int v[N];
  constexpr ChanLut() : v{} {
    for (int i = 0; i < n_acc_phase<KIND>(PH); ++i) v[i] = channel_of<KIND>(PH, i);
  }
};
template <int KIND, int PH> __device__ constexpr ChanLut<KIND, PH> kChanLut{};

// Cross-half sum of one phase's accumulators + staging into the LDS message row (column COL0 + phase-local index).
// The two lane halves hold partial sums over disjoint in-channels of the SAME edge.  v_permlane32_swap exchanges the upper half of
// one register with the lower half of another, so for an accumulator pair (a, b) one swap + one add leaves total(a) in the lower
// half and total(b) in the upper half; each half then stores its own column: NA/2 swaps, adds and ds_write_b32, no LDS permutes.
template <int KIND, int PH, int COL0, int... PI>
__device__ __forceinline__ void stage_phase(float (&acc)[NACC<KIND>], float* row, int half, std::integer_sequence<int, PI...>) {
  static_assert(n_acc_phase<KIND>(PH) % 2 == 0, "accumulators are staged in pairs");
  float* dst = row + COL0 + half;
  (
      [&] {
        float x = acc[2 * PI], y = acc[2 * PI + 1];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));   // x = [x.lo, y.lo], y = [x.hi, y.hi]
        dst[2 * PI] = x + y;
      }(),
      ...);
}

// ---------------------------------------------------------------------------------------------------------------
// pieces shared by the fp32-MFMA kernel (PREC 0) and the split-bf16 kernels (PREC 2 / 3)
// ---------------------------------------------------------------------------------------------------------------
struct WgRange {
  int pose, sc0, n_lo, n_hi, n_loc, l0, p0, NL, NP, x0, k_begin, k_end;
};

__device__ __forceinline__ WgRange wg_range(const DpConvArgs& a, int bid = -1) {
  WgRange r;
  if (bid < 0) bid = blockIdx.x;
  if (a.mode == DP_EDGES_LIST) {        // workgroup = 128 consecutive edges of the list, whatever nodes they belong to (messages are written out)
    r.pose = r.sc0 = r.n_lo = r.l0 = r.p0 = r.NL = r.NP = r.x0 = 0;
    r.k_begin = bid * 128;
    r.k_end = r.k_begin + 128 < a.n_list ? r.k_begin + 128 : a.n_list;
    r.n_hi = r.n_loc = r.k_begin < a.n_list ? 1 : 0;
    return r;
  }
  r.pose = bid / a.slices;
  const int slice = bid % a.slices;
  r.sc0 = a.scat_ptr[r.pose];
  const int nsc = a.scat_ptr[r.pose + 1] - r.sc0;
  r.n_lo = r.sc0 + (int)(((long long)nsc * slice) / a.slices);
  r.n_hi = r.sc0 + (int)(((long long)nsc * (slice + 1)) / a.slices);
  r.n_loc = r.n_hi - r.n_lo;
  r.l0 = r.p0 = r.NL = r.NP = r.x0 = 0;
  r.k_begin = r.k_end = 0;
  if (r.n_loc <= 0) return r;
  if (a.mode != DP_EDGES_CSR) {
    r.l0 = a.lig_ptr[r.pose];
    r.NL = a.lig_ptr[r.pose + 1] - r.l0;
    if (a.mode != DP_EDGES_CENTER) {
      r.p0 = a.ph_ptr[r.pose];
      r.NP = a.ph_ptr[r.pose + 1] - r.p0;
      r.x0 = a.x_ptr[r.pose];
    }
  }
  if (a.mode == DP_EDGES_CSR) {
    r.k_begin = a.rowptr[r.n_lo];
    r.k_end = a.rowend[r.n_hi - 1];
  } else if (a.mode == DP_EDGES_P2L) {
    r.k_begin = (r.n_lo - r.sc0) * r.NP;
    r.k_end = (r.n_hi - r.sc0) * r.NP;
  } else if (a.mode == DP_EDGES_L2P) {
    r.k_begin = (r.n_lo - r.sc0) * r.NL;
    r.k_end = (r.n_hi - r.sc0) * r.NL;
  } else {
    r.k_begin = 0;
    r.k_end = r.NL;
  }
  return r;
}

// sorted edge slot -> (scatter, gather, attr rows, feature row)
__device__ __forceinline__ EdgeRef decode_edge(const DpConvArgs& a, const WgRange& r, int k) {
  const bool valid = k < r.k_end;
  EdgeRef e;
  if (a.mode == DP_EDGES_CSR) {
    const int kk = valid ? k : r.k_begin;
    e.feat = kk;
    e.gather = a.nbr[kk];
    const int sabs = a.edge_scat[kk];
    const bool in_rng = sabs >= r.n_lo && sabs < r.n_hi;
    e.scat_local = (valid && in_rng) ? sabs - r.n_lo : -1;
    const int sg = in_rng ? sabs : r.n_lo;
    e.a1 = a.attr_swap ? e.gather : sg;
    e.a2 = a.attr_swap ? sg : e.gather;
  } else if (a.mode == DP_EDGES_LIST) {
    const int kk = valid ? k : r.k_begin;
    e.feat = a.feat_idx ? a.feat_idx[kk] : kk;
    e.gather = a.nbr[kk];
    const int sabs = a.edge_scat[kk];
    e.scat_local = valid ? 0 : -1;
    e.a1 = a.attr_swap ? e.gather : sabs;
    e.a2 = a.attr_swap ? sabs : e.gather;
  } else if (a.mode == DP_EDGES_P2L) {
    const int kk = valid ? k : r.k_begin;
    const int i = kk / r.NP, j = kk - i * r.NP;
    e.feat = r.x0 + kk;
    e.gather = r.p0 + j;
    e.a1 = r.l0 + i;
    e.a2 = r.p0 + j;
    e.scat_local = valid ? (i - (r.n_lo - r.sc0)) : -1;
  } else if (a.mode == DP_EDGES_L2P) {
    const int kk = valid ? k : r.k_begin;
    const int j = kk / r.NL, i = kk - j * r.NL;
    e.feat = r.x0 + i * r.NP + j;
    e.gather = r.l0 + i;
    e.a1 = r.l0 + i;
    e.a2 = r.p0 + j;
    e.scat_local = valid ? (j - (r.n_lo - r.sc0)) : -1;
  } else {
    const int kk = valid ? k : 0;
    e.feat = r.l0 + kk;
    e.gather = r.l0 + kk;
    e.a1 = r.l0 + kk;
    e.a2 = 0;
    e.scat_local = valid ? 0 : -1;
  }
  return e;
}

// gathered TP input (this lane half's in-channels), spherical harmonics, l=2 contraction matrix
template <int KIND>
__device__ __forceinline__ void load_tp_operands(const DpConvArgs& a, const EdgeRef& e, int half, float (&xr)[Spec<KIND>::D_IN / 2],
                                                 float (&sh)[Spec<KIND>::SH_DIM], float (&M)[9]) {
  using S = Spec<KIND>;
  const float* px = a.gather_tab + (size_t)e.gather * a.gather_stride;
#pragma unroll
  for (int i = 0; i < 10; ++i) xr[i] = px[half * 10 + i];                                   // 20x0e
  if constexpr (S::D_IN >= 50) {
#pragma unroll
    for (int i = 0; i < 15; ++i) xr[10 + i] = px[20 + half * 15 + i];                       // 10x1o
  }
  if constexpr (S::D_IN >= 80) {
#pragma unroll
    for (int i = 0; i < 15; ++i) xr[25 + i] = px[50 + half * 15 + i];                       // 10x1e
  }
  if constexpr (S::D_IN >= 100) {
#pragma unroll
    for (int i = 0; i < 10; ++i) xr[40 + i] = px[80 + half * 10 + i];                       // 20x0o
  }
  const float* ps = a.edge_sh + (size_t)e.feat * a.sh_stride;
#pragma unroll
  for (int i = 0; i < S::SH_DIM; ++i) sh[i] = ps[i];
  if (a.mode == DP_EDGES_LIST && a.sh0_zero) sh[0] = 0.f;      // the l = 0 part of the conv is computed elsewhere (DpConvArgs.s0_only)
#pragma unroll
  for (int i = 0; i < 9; ++i) M[i] = 0.f;
  if constexpr (S::SH_DIM == 9) {              // sqrt(10) * sum_j sh2[j] C(1,2,1)[i,j,k]  (1/sqrt10 folded into W2p)
    const float c = 0.57735026918962576451f * sh[6];
    M[0] = -c - sh[8]; M[4] = 2.f * c; M[8] = -c + sh[8];
    M[1] = M[3] = sh[5]; M[2] = M[6] = sh[4]; M[5] = M[7] = sh[7];
  }
}

// GEMM1 (fp32 MFMA): hidden = relu(W1 [edge_emb | attr1[:20] | attr2[:20]] + b1); result = this lane's 32 D registers
template <int KIND>
__device__ __forceinline__ void gemm1_hidden(const DpConvArgs& a, const EdgeRef& e, int lane, float (&hid)[32]) {
  using S = Spec<KIND>;
  using G = Geo<KIND>;
  const int half = lane >> 5;
  float av[G::ABLK * 10];                      // this half's 10 floats of every 20-float attr block
  {
    const float* pe = a.edge_emb + (size_t)e.feat * 20 + half * 10;
#pragma unroll
    for (int i = 0; i < 10; i += 2) { const float2 v = *reinterpret_cast<const float2*>(pe + i); av[i] = v.x; av[i + 1] = v.y; }
    const float* p1 = a.attr1_tab + (size_t)e.a1 * a.attr1_stride + half * 10;
#pragma unroll
    for (int i = 0; i < 10; i += 2) { const float2 v = *reinterpret_cast<const float2*>(p1 + i); av[10 + i] = v.x; av[11 + i] = v.y; }
    if constexpr (G::ABLK == 3) {
      const float* p2 = a.attr2_tab + (size_t)e.a2 * a.attr2_stride + half * 10;
#pragma unroll
      for (int i = 0; i < 10; i += 2) { const float2 v = *reinterpret_cast<const float2*>(p2 + i); av[20 + i] = v.x; av[21 + i] = v.y; }
    }
  }
  gptr4 w1p_lane = (gptr4)a.w1p + lane;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x16 H = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G::K1P / 4; ++g) {
      const vf4 w = w1p_lane[(t * (G::K1P / 4) + g) * 64];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int s = 4 * g + c;
        float b;
        if (s < S::N_ATTR / 2) b = av[s];
        else if (s == S::N_ATTR / 2) b = half ? 0.f : 1.f;
        else b = 0.f;
        const float wa = c == 0 ? w.x : c == 1 ? w.y : c == 2 ? w.z : w.w;
        if (s < G::K1) H = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, b, H, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[t * 16 + r] = fmaxf(H[r], 0.f);
  }
}

// Deterministic segmented sum of staged messages into the LDS node accumulators, one wave per 32-edge tile.
//   Every wave reduces the 32 message rows IT staged (so no workgroup barrier sits between staging and reduction): lane l owns
//   message column l (+64 per pass) and walks the 32 edges; the scatter node of edge ei lives in lane ei's register and is
//   broadcast with v_readlane, so segment boundaries are scalar branches.  A node that starts strictly inside the tile is
//   added to s_acc directly (exactly one wave can do that for a given node); the tile's FIRST segment may continue a node begun by
//   an earlier wave, so its sum is parked in the wave's own (now dead) row 0 and merged by wave 0 after a barrier, in wave
//   order (merge_wave_partials).  The summation tree depends only on the pose's own edge list: reruns are bit-identical.
template <int D_OUT, int NCOL, typename ChanFn>
__device__ __forceinline__ void wave_segsum(float* rows, int ld, int lane, int scat_local, float* s_acc, int* s_pnode_w, ChanFn chan,
                                            int col_lo = 0, int col_hi = NCOL) {
  const int first_node = __builtin_amdgcn_readlane(scat_local, 0);
  if (lane == 0) *s_pnode_w = first_node;
#pragma unroll
  for (int p = 0; p < (NCOL + 63) / 64; ++p) {
    const int col = lane + 64 * p;
    const bool active = col >= col_lo && col < col_hi;
    const int ch = active ? chan(col) : 0;
    float v[32];
#pragma unroll
    for (int ei = 0; ei < 32; ++ei) v[ei] = active ? rows[ei * ld + col] : 0.f;
    int cur = first_node;
    bool first = true;
    float run = 0.f;
#pragma unroll
    for (int ei = 0; ei < 32; ++ei) {
      const int n = __builtin_amdgcn_readlane(scat_local, ei);
      if (n != cur) {                                    // uniform
        if (cur >= 0 && active) {
          if (first) rows[col] = run;
          else s_acc[cur * D_OUT + ch] += run;
        }
        first = false;
        run = 0.f;
        cur = n;
      }
      if (n >= 0) run += v[ei];                          // padding edges (-1) only trail the pose's edge list
    }
    if (cur >= 0 && active) {
      if (first) rows[col] = run;
      else s_acc[cur * D_OUT + ch] += run;
    }
  }
}

// second half of the above: wave 0 folds the NW parked first-segment sums into s_acc in wave order (call after a barrier)
template <int D_OUT, int NCOL, int NW, typename ChanFn>
__device__ __forceinline__ void merge_wave_partials(const float* s_msg, int ld, int lane, const int* s_pnode, float* s_acc, ChanFn chan,
                                                    int col_lo = 0, int col_hi = NCOL) {
#pragma unroll
  for (int p = 0; p < (NCOL + 63) / 64; ++p) {
    const int col = lane + 64 * p;
    if (col >= col_lo && col < col_hi) {
      const int ch = chan(col);
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const int node = s_pnode[q];
        if (node >= 0) s_acc[node * D_OUT + ch] += s_msg[(size_t)q * 32 * ld + col];
      }
    }
  }
}

// DP_EDGES_LIST: the wave's staged message rows go out as they are - dst = row of the wave's first edge in msg_out [n_list, D_OUT], n_rows of them exist
template <int D_OUT, int NCOL, typename ChanFn>
__device__ __forceinline__ void wave_emit(const float* rows, int ld, int lane, float* __restrict__ dst, int n_rows, ChanFn chan) {
#pragma unroll
  for (int p = 0; p < (NCOL + 63) / 64; ++p) {
    const int col = lane + 64 * p;
    if (col < NCOL) {
      const int ch = chan(col);
      for (int ei = 0; ei < n_rows; ++ei) dst[(size_t)ei * D_OUT + ch] = rows[ei * ld + col];
    }
  }
}

// mean, BatchNorm affine, residual stream
template <int KIND, int NTHREADS>
__device__ __forceinline__ void conv_epilogue(const DpConvArgs& a, const WgRange& r, const float* s_acc) {
  using S = Spec<KIND>;
  for (int i = threadIdx.x; i < r.n_loc * S::D_OUT; i += NTHREADS) {
    const int n = i / S::D_OUT, c = i - n * S::D_OUT;
    const int node = r.n_lo + n;
    int deg;
    if (a.mode == DP_EDGES_CSR) deg = a.rowend[node] - a.rowptr[node];
    else if (a.mode == DP_EDGES_P2L) deg = r.NP;
    else deg = r.NL;
    float v = s_acc[i] / (float)max(deg, 1);
    v = v * a.bn_scale[c] + a.bn_shift[c];
    float* o = a.out + (size_t)node * a.out_stride + c;
    if (a.accumulate) {
      *o += v;
    } else {
      if (a.resid != nullptr && c < a.resid_width) v += a.resid[(size_t)node * a.resid_stride + c];
      *o = v;
    }
  }
}

// the same for a window of message columns (phase-split launches: a workgroup owns one phase's output channels)
template <int KIND, int NTHREADS, typename ChanFn>
__device__ __forceinline__ void conv_epilogue_cols(const DpConvArgs& a, const WgRange& r, const float* s_acc, ChanFn chan, int col_lo, int col_hi) {
  using S = Spec<KIND>;
  const int nc = col_hi - col_lo;
  for (int i = threadIdx.x; i < r.n_loc * nc; i += NTHREADS) {
    const int n = i / nc, c = chan(col_lo + (i - n * nc));
    const int node = r.n_lo + n;
    int deg;
    if (a.mode == DP_EDGES_CSR) deg = a.rowend[node] - a.rowptr[node];
    else if (a.mode == DP_EDGES_P2L) deg = r.NP;
    else deg = r.NL;
    float v = s_acc[n * S::D_OUT + c] / (float)max(deg, 1);
    v = v * a.bn_scale[c] + a.bn_shift[c];
    float* o = a.out + (size_t)node * a.out_stride + c;
    if (a.accumulate) {
      *o += v;
    } else {
      if (a.resid != nullptr && c < a.resid_width) v += a.resid[(size_t)node * a.resid_stride + c];
      *o = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// PREC 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 4 waves / workgroup, W2p streamed from L2 per lane
// ---------------------------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(WAVES * 64, 2) void tp_conv_kernel(DpConvArgs a) {
  using S = Spec<KIND>;
  using G = Geo<KIND>;
  __shared__ float s_msg[CHUNK * G::MSG_LD];
  __shared__ float s_acc[MAX_NODES * S::D_OUT];
  __shared__ int s_pnode[WAVES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ecol = lane & 31, half = lane >> 5;
  // phase_split: two workgroups per (pose, slice), one per accumulation phase = disjoint output channels.  Small batches cannot
  // fill the chip with one workgroup per pose slice and a conv's latency is its longest workgroup; the split halves it at the
  // price of doing the gathers and GEMM1 twice.  Per-channel sums are unchanged, so results are bit-identical to the unsplit launch.
  const int only_ph = a.phase_split ? (int)(blockIdx.x & 1u) : -1;
  const WgRange r = wg_range(a, a.phase_split ? (int)(blockIdx.x >> 1) : (int)blockIdx.x);
  if (r.n_loc <= 0) return;
  for (int i = tid; i < r.n_loc * S::D_OUT; i += WAVES * 64) s_acc[i] = 0.f;
  constexpr int NA0_ = n_acc_phase<KIND>(0);
  const int col_lo = only_ph == 1 ? NA0_ : 0, col_hi = only_ph == 0 ? NA0_ : S::D_OUT;
  const auto chan = [](int c) { return c < NA0_ ? kChanLut<KIND, 0>.v[c < NA0_ ? c : 0] : kChanLut<KIND, 1>.v[c < NA0_ ? 0 : c - NA0_]; };

  PROF_MARK(-1);
  for (int kc = r.k_begin; kc < r.k_end; kc += CHUNK) {
    const EdgeRef e = decode_edge(a, r, kc + wave * 32 + ecol);
    float xr[S::D_IN / 2], sh[S::SH_DIM], M[9], hid[32];
    load_tp_operands<KIND>(a, e, half, xr, sh, M);
    gemm1_hidden<KIND>(a, e, lane, hid);

    // ---- GEMM2 tiles + CG contraction, phase 0 (scalar outputs) then phase 1 (vector outputs) ----------------------
    PROF_MARK(0);
    __syncthreads();                              // previous chunk's partial sums have been merged (first pass: s_acc zeroed)
    PROF_MARK(1);
    float* row = s_msg + (size_t)(wave * 32 + ecol) * G::MSG_LD;
    {
      constexpr int NT0 = n_tiles_phase<KIND>(0), NT1 = n_tiles_phase<KIND>(1);
      constexpr int NA0 = n_acc_phase<KIND>(0), NA1 = n_acc_phase<KIND>(1);
      float acc[NACC<KIND>];
      if constexpr (NT0 > 0) if (only_ph != 1) {
#pragma unroll
        for (int i = 0; i < NACC<KIND>; ++i) acc[i] = 0.f;
        gptr4 wp = (gptr4)a.w2p;
        WTile<KIND> w0;
        load_wtile<KIND>(wp, lane, w0);
        __builtin_amdgcn_sched_barrier(0);
        Stage<KIND, 0, NT0>::run(wp, lane, w0, hid, xr, sh, M, acc);
        PROF_MARK(3);
        stage_phase<KIND, 0, 0>(acc, row, half, std::make_integer_sequence<int, NA0 / 2>{});
        PROF_MARK(4);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (NT1 > 0) if (only_ph != 0) {
#pragma unroll
        for (int i = 0; i < NACC<KIND>; ++i) acc[i] = 0.f;
        gptr4 wp = (gptr4)a.w2p + (size_t)NT0 * (G::K2 / 4) * 64;
        asm volatile("" : "+s"(wp));
        WTile<KIND> w0;
        load_wtile<KIND>(wp, lane, w0);
        __builtin_amdgcn_sched_barrier(0);
        Stage<KIND, NT0, NT0 + NT1>::run(wp, lane, w0, hid, xr, sh, M, acc);
        PROF_MARK(5);
        stage_phase<KIND, 1, NA0>(acc, row, half, std::make_integer_sequence<int, NA1 / 2>{});
      }
    }
#if DP_ABLATE != 16
    wave_segsum<S::D_OUT, S::D_OUT>(s_msg + (size_t)wave * 32 * G::MSG_LD, G::MSG_LD, lane, e.scat_local, s_acc, &s_pnode[wave], chan,
                                    col_lo, col_hi);
    __syncthreads();
    if (wave == 0) merge_wave_partials<S::D_OUT, S::D_OUT, WAVES>(s_msg, G::MSG_LD, lane, s_pnode, s_acc, chan, col_lo, col_hi);
#endif
    PROF_MARK(6);
  }
  __syncthreads();
  if (only_ph < 0) conv_epilogue<KIND, WAVES * 64>(a, r, s_acc);
  else conv_epilogue_cols<KIND, WAVES * 64>(a, r, s_acc, chan, col_lo, col_hi);
  PROF_MARK(7);
}

// ---------------------------------------------------------------------------------------------------------------
// PREC 2 / 3: radial-MLP layer 2 on the bf16 matrix pipe with operand splitting (Ozaki-style error-free transformation)
//   x = x_hi + x_mid (+ x_lo), each part a bf16;  w.h ~= sum of the NS*(NS+1)/2 largest cross products, fp32 accumulate.
//   NS = 3 (6 products): relative error ~2^-24, i.e. fp32-grade;  NS = 2 (3 products): ~2^-17.
//   v_mfma_f32_32x32x16_bf16 runs 16x the fp32 MFMA rate, so a tile costs 24 (or 12) x 32 cycles instead of 32 x 64.
//   At that rate per-lane weight streaming from L2 would saturate the CU's L1/L2 path, so the 4 waves of a workgroup share
//   each packed weight tile through a double-buffered LDS stage (register-staged, written after the MFMA phase).  Two
//   workgroups run per CU; inside one the waves are barrier-coupled per tile, across the two they are not, which is what lets
//   one wave's VALU / LDS / memory phases overlap the other's MFMA chain on the same SIMD.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int vu4 __attribute__((ext_vector_type(4)));
constexpr int WAVES_B = 4;        // two such workgroups share a CU: their waves are not barrier-coupled, so one hides the other
constexpr int CHUNK_B = WAVES_B * 32;

template <int KIND> constexpr int NKB = (Spec<KIND>::HID == 60 ? 62 : Spec<KIND>::HID + 1 + 15) / 16 + (Spec<KIND>::HID == 60 ? 1 : 0);   // 16-row K blocks covering every non-padding row
template <int KIND, int NS> constexpr int TILE_QUADS = (NKB<KIND> * NS * 64 + 255) / 256 * 256;   // uint4 per packed tile, padded to whole 4-wave DMA rounds
// hidden row held by lane half `half` at position j (0..7) of K block b: the D-fragment row of GEMM1 register 8*(b&1)+j, tile b>>1
constexpr int hid_row_b(int b, int half, int j) {
  const int r = 8 * (b & 1) + j;
  return 32 * (b >> 1) + (r & 3) + 8 * (r >> 2) + 4 * half;
}

__host__ __device__ inline unsigned short f32_to_bf16_rne(float x) {
  unsigned int u = __builtin_bit_cast(unsigned int, x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned int)h << 16); }

template <int KIND, int NS> void pack_w2_bf16(const float* fc3_w, const float* fc3_b, unsigned short* out) {
  using G = Geo<KIND>;
  using S = Spec<KIND>;
  const size_t total = (size_t)G::NT * TILE_QUADS<KIND, NS> * 8;
  for (size_t i = 0; i < total; ++i) out[i] = 0;
  for (int tile = 0; tile < G::NT; ++tile)
    for (int b = 0; b < NKB<KIND>; ++b)
      for (int lane = 0; lane < 64; ++lane) {
        const int c = lane & 31, half = lane >> 5;
        const int r = (c & 3) + 4 * (c >> 3), chalf = (c >> 2) & 1;
        const Triple tr = triple_at<KIND>(tile * 16 + r);
        if (!tr.valid) continue;
        const PathC& p = S::P[tr.p];
        const int u = chalf * (p.mul_in / 2) + tr.q;
        const int widx = p.w_off + u * p.mul_out + tr.w;
        const double scale = __builtin_sqrt((double)(2 * (d_out_of(p.ztype) == 3 ? 1 : 0) + 1) / (double)p.fan_in) * cg_const(p.ztype);
        for (int j = 0; j < 8; ++j) {
          const int unit = unit_of_row<KIND>(hid_row_b(b, half, j));
          float v = 0.f;
          if (unit >= 0 && unit < S::HID) v = (float)(scale * (double)fc3_w[widx * S::HID + unit]);
          else if (unit == S::HID) v = (float)(scale * (double)fc3_b[widx]);
          float rem = v;
          for (int s = 0; s < NS; ++s) {
            const unsigned short h = f32_to_bf16_rne(rem);
            rem -= bf16_to_f32(h);
            out[((size_t)tile * TILE_QUADS<KIND, NS> + ((size_t)b * NS + s) * 64 + lane) * 8 + j] = h;
          }
        }
      }
}

// DP_PREC_F16X3 on this kernel: the tiles of pack_w2_bf16<KIND, 2> with fp16 parts of W2 * 2^w (w = the largest exponent that keeps every
// element <= 2048; exact) and one trailing quad whose float[0] is w.  The kernel never reads the trailer: the host multiplies the conv's BatchNorm
// scale by 2^-w (the contraction, the segmented mean and the scale are linear in W2), diffphore_amd/packing.py.
template <int KIND> void pack_w2_f16(const float* fc3_w, const float* fc3_b, unsigned short* out) {
  using G = Geo<KIND>;
  using S = Spec<KIND>;
  constexpr int NS = 2;
  const size_t total = (size_t)G::NT * TILE_QUADS<KIND, NS> * 8 + 8;
  for (size_t i = 0; i < total; ++i) out[i] = 0;
  auto value = [&](int tile, int b, int lane, int j) -> double {
    const int c = lane & 31, half = lane >> 5;
    const int r = (c & 3) + 4 * (c >> 3), chalf = (c >> 2) & 1;
    const Triple tr = triple_at<KIND>(tile * 16 + r);
    if (!tr.valid) return 0.0;
    const PathC& p = S::P[tr.p];
    const int u = chalf * (p.mul_in / 2) + tr.q;
    const int widx = p.w_off + u * p.mul_out + tr.w;
    const double scale = __builtin_sqrt((double)(2 * (d_out_of(p.ztype) == 3 ? 1 : 0) + 1) / (double)p.fan_in) * cg_const(p.ztype);
    const int unit = unit_of_row<KIND>(hid_row_b(b, half, j));
    if (unit >= 0 && unit < S::HID) return scale * (double)fc3_w[widx * S::HID + unit];
    if (unit == S::HID) return scale * (double)fc3_b[widx];
    return 0.0;
  };
  double mx = 0.0;
  for (int tile = 0; tile < G::NT; ++tile)
    for (int b = 0; b < NKB<KIND>; ++b)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) { const double v = __builtin_fabs(value(tile, b, lane, j)); if (v > mx) mx = v; }
  const int wexp = f16_image_exponent(mx);
  const double sc = __builtin_ldexp(1.0, wexp);
  for (int tile = 0; tile < G::NT; ++tile)
    for (int b = 0; b < NKB<KIND>; ++b)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          unsigned short hi, lo;
          f16_split((float)(sc * value(tile, b, lane, j)), hi, lo);
          out[((size_t)tile * TILE_QUADS<KIND, NS> + ((size_t)b * NS + 0) * 64 + lane) * 8 + j] = hi;
          out[((size_t)tile * TILE_QUADS<KIND, NS> + ((size_t)b * NS + 1) * 64 + lane) * 8 + j] = lo;
        }
  float* trailer = reinterpret_cast<float*>(out + (size_t)G::NT * TILE_QUADS<KIND, NS> * 8);
  trailer[0] = (float)wexp;
}

template <int KIND, int NS> struct HidB { bf16x8 v[NKB<KIND>][NS]; };

// F16 (DP_PREC_F16X3 on this kernel, round 5): the parts are fp16 (11 significand bits each: two parts = 22 bits, fp32-grade with 3 cross
// products instead of the 6 of the 3-way bf16 split).  The hidden activations are relu outputs of O(1): no scale needed on this side; the packed
// weights carry 2^w (pack_w2_f16), which the host folds into the conv's BatchNorm scale (DpConvW.bn_scale_x)
template <int KIND, int NS, bool F16 = false>
__device__ __forceinline__ void split_hidden(const float (&hid)[32], HidB<KIND, NS>& hb) {
#pragma unroll
  for (int b = 0; b < NKB<KIND>; ++b) {
    unsigned short q[NS][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float rem = hid[8 * b + j];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if constexpr (F16) {
          const _Float16 h = (_Float16)rem;               // round to nearest even
          rem -= (float)h;
          q[s][j] = __builtin_bit_cast(unsigned short, h);
        } else {
          const unsigned short h = f32_to_bf16_rne(rem);
          rem -= bf16_to_f32(h);
          q[s][j] = h;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      vu4 pk;
      pk.x = q[s][0] | ((unsigned)q[s][1] << 16);
      pk.y = q[s][2] | ((unsigned)q[s][3] << 16);
      pk.z = q[s][4] | ((unsigned)q[s][5] << 16);
      pk.w = q[s][6] | ((unsigned)q[s][7] << 16);
      hb.v[b][s] = __builtin_bit_cast(bf16x8, pk);
    }
  }
}

typedef const __attribute__((address_space(1))) vu4* gptrq;

// tiles staged per LDS buffer: the workgroup barrier is taken once per KT tiles, so inside a group the two waves of a SIMD
// drift apart and one wave's VALU contraction overlaps the other's MFMA chain (a barrier per tile keeps them in lock-step).
template <int NS> constexpr int KT = 1;
template <int KIND> constexpr int MSG_LD_B = (n_acc_phase<KIND>(0) > n_acc_phase<KIND>(1) ? n_acc_phase<KIND>(0) : n_acc_phase<KIND>(1)) + 1;

typedef __attribute__((address_space(3))) vu4* lptrq;
constexpr int MAX_NODES_B = 32;                                   // LDS budget: two workgroups per CU
template <int NS> constexpr int RING = NS == 3 ? 3 : 4;           // weight-tile ring depth (tiles in flight = RING-1)
template <int KIND, int NS> constexpr int PIECES = TILE_QUADS<KIND, NS> / (WAVES_B * 64);   // 1-KiB DMA pieces per wave per tile

// asynchronous L2 -> LDS copy of one packed weight tile (buffer_load...lds / global_load_lds_dwordx4: no VGPRs, counted on vmcnt)
template <int KIND, int NS>
__device__ __forceinline__ void dma_tile(gptrq src_tile, vu4* slot, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < PIECES<KIND, NS>; ++i)
    __builtin_amdgcn_global_load_lds(src_tile + i * WAVES_B * 64 + wave * 64 + lane, (lptrq)(slot + i * WAVES_B * 64 + wave * 64), 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One tile's operand stream: NKB K-blocks x NS weight splits, each a 1-KiB lane-contiguous ds_read_b128.  Split s of block b+1
// is requested as soon as the last MFMA that consumes split s of block b has been issued, so NS reads stay in flight behind the
// matrix pipe with only NS operand registers.  The reads are inline asm (see StageB) and therefore waited for by hand:
// LDS returns in order, so "at most NS-1 outstanding" == "the oldest request has landed".
__device__ __forceinline__ void lds_read_q(bf16x8& dst, unsigned addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
}
template <int N> __device__ __forceinline__ void lds_landed(bf16x8& w) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w) : "n"(N)); }

template <int NS> constexpr int WREGS = NS == 1 ? 2 : NS;      // operand registers of the pipe (NS = 1: two, so that one read stays in flight)
typedef _Float16 f16x8_e __attribute__((ext_vector_type(8)));
template <bool F16> __device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) {      // the same registers read as bf16 or as fp16 operands
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_e, a), __builtin_bit_cast(f16x8_e, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int KIND, int NS, int B, bool FULL = false, bool F16 = false>      // FULL (NS == 3 only): all 9 cross products = exact fp32 products
struct OperandPipe {
  static __device__ __forceinline__ void run(unsigned la, bf16x8 (&W)[WREGS<NS>], const HidB<KIND, NS>& hb, f32x16& D) {
    constexpr int NB = NKB<KIND>;
    if constexpr (NS == 1) {
      // DP_PREC_BF16: plain bf16 operands, one product per K block; blocks alternate between two registers, one read ahead
      if constexpr (B < NB) {
        if constexpr (B == 0) {
          lds_read_q(W[0], la, 0);
          if constexpr (NB > 1) lds_read_q(W[1], la, 1024);
        }
        lds_landed<(B + 1 < NB) ? 1 : 0>(W[B & 1]);
        D = mma32<F16>(W[B & 1], hb.v[B][0], D);
        if constexpr (B + 2 < NB) lds_read_q(W[B & 1], la, (B + 2) * 1024);
        OperandPipe<KIND, NS, B + 1, FULL, F16>::run(la, W, hb, D);
      }
    } else if constexpr (B < NB) {
      constexpr bool more = B + 1 < NB;
      if constexpr (B == 0) {
#pragma unroll
        for (int s = NS - 1; s >= 0; --s) lds_read_q(W[s], la, s * 1024);      // request order = consumption order: last split first
      }
      // smallest cross terms first; a split's register is re-requested right after its last use
      if constexpr (NS == 3) {
        lds_landed<NS - 1>(W[2]);
        if constexpr (FULL) {
          D = mma32<F16>(W[2], hb.v[B][2], D);
          D = mma32<F16>(W[2], hb.v[B][1], D);
        }
        D = mma32<F16>(W[2], hb.v[B][0], D);
        bf16x8 n2;
        if constexpr (more) lds_read_q(n2, la, ((B + 1) * NS + 2) * 1024);
        lds_landed<more ? NS - 1 : 1>(W[1]);
        if constexpr (FULL) D = mma32<F16>(W[1], hb.v[B][2], D);
        D = mma32<F16>(W[1], hb.v[B][1], D);
        D = mma32<F16>(W[1], hb.v[B][0], D);
        bf16x8 n1;
        if constexpr (more) lds_read_q(n1, la, ((B + 1) * NS + 1) * 1024);
        lds_landed<more ? NS - 1 : 0>(W[0]);
        D = mma32<F16>(W[0], hb.v[B][2], D);
        D = mma32<F16>(W[0], hb.v[B][1], D);
        D = mma32<F16>(W[0], hb.v[B][0], D);
        if constexpr (more) {
          lds_read_q(W[0], la, ((B + 1) * NS + 0) * 1024);
          W[2] = n2;
          W[1] = n1;
        }
      } else {
        lds_landed<NS - 1>(W[1]);
        D = mma32<F16>(W[1], hb.v[B][0], D);
        bf16x8 n1;
        if constexpr (more) lds_read_q(n1, la, ((B + 1) * NS + 1) * 1024);
        lds_landed<more ? NS - 1 : 0>(W[0]);
        D = mma32<F16>(W[0], hb.v[B][1], D);
        D = mma32<F16>(W[0], hb.v[B][0], D);
        if constexpr (more) {
          lds_read_q(W[0], la, ((B + 1) * NS + 0) * 1024);
          W[1] = n1;
        }
      }
      OperandPipe<KIND, NS, B + 1, FULL, F16>::run(la, W, hb, D);
    }
  }
};

// Stage TILE of the split-bf16 pipeline.  The packed weights stream L2 -> LDS through a RING-deep ring filled by LDS-DMA
// RING-1 tiles ahead (an L2 round trip under load is ~2k cycles, longer than one bf16 tile), so a tile costs:
//   issue DMA(tile+RING-1) | ds_read operands | 12 or 24 bf16 MFMAs | VALU contraction | counted vmcnt | s_barrier.
// a tile whose 16 weight rows all belong to paths through the l = 0 harmonic (Z000 / Z101): a DP_EDGES_LIST launch with DpConvArgs.sh0_zero reads
// that component as 0, so the tile's products are multiplied by zero - its MFMAs and contraction are skipped (SKIPL0; the ring still carries it)
template <int KIND, int TILE> constexpr bool tile_l0_only() {
  for (int r = 0; r < 16; ++r) {
    const Triple t = triple_at<KIND>(TILE * 16 + r);
    if (!t.valid) continue;
    const int z = Spec<KIND>::P[t.p].ztype;
    if (z != Z000 && z != Z101) return false;
  }
  return true;
}

// EMIT (DP_EDGES_LIST): the staged messages are written to `emit` (the wave's first row of msg_out; emit_rows rows exist) instead of being summed
// PH (EMIT only): -1 = every tile; 0 / 1 = this workgroup runs the tiles of accumulation phase 0 / 1 only (DpConvArgs.phase_split of a list launch:
// two workgroups per 128-edge chunk, each emits its phase's output channels - the phases share nothing but the operands)
template <int KIND, int PH> constexpr int tile_begin() { return PH == 1 ? n_tiles_phase<KIND>(0) : 0; }
template <int KIND, int PH> constexpr int tile_end() { return PH == 0 ? n_tiles_phase<KIND>(0) : Geo<KIND>::NT; }
template <int KIND, int NS, int TILE, bool FULL = false, bool F16 = false, bool EMIT = false, bool SKIPL0 = false, int PH = -1>
struct StageB {
  static __device__ __forceinline__ void run(gptrq w2p, int wave, int lane, vu4* s_w, const HidB<KIND, NS>& hb,
                                             const float (&xr)[Spec<KIND>::D_IN / 2], const float (&sh)[Spec<KIND>::SH_DIM],
                                             const float (&M)[9], float (&acc)[NACC<KIND>], float* row, int half,
                                             float* s_msg, int* s_pnode, float* s_acc, int scat_local, float* emit = nullptr, int emit_rows = 0) {
    using G = Geo<KIND>;
    constexpr int NT0 = n_tiles_phase<KIND>(0);
    constexpr int NQ = TILE_QUADS<KIND, NS>;
    constexpr int R = RING<NS>;
    constexpr int TE = tile_end<KIND, PH>();
    static_assert(PH < 0 || EMIT, "the phase split exists for list launches only");
    if constexpr (TILE < TE) {
      if constexpr (TILE == NT0 && NT0 > 0 && PH < 0) {      // phase boundary: scalar outputs done -> stage, reduce, restart accumulators
        PROF_MARK(3);
        stage_phase<KIND, 0, 0>(acc, row, half, std::make_integer_sequence<int, n_acc_phase<KIND>(0) / 2>{});
        const auto chan0 = [](int c) { return kChanLut<KIND, 0>.v[c]; };
        if constexpr (EMIT) {
          // (the stores count on vmcnt like the ring's DMA pieces: the counted waits below then ask for no less than before - loads land in order)
          wave_emit<Spec<KIND>::D_OUT, n_acc_phase<KIND>(0)>(s_msg + (size_t)wave * 32 * MSG_LD_B<KIND>, MSG_LD_B<KIND>, lane, emit, emit_rows, chan0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are read before phase 1 restages them (each wave its own)
        } else {
        wave_segsum<Spec<KIND>::D_OUT, n_acc_phase<KIND>(0)>(s_msg + (size_t)wave * 32 * MSG_LD_B<KIND>, MSG_LD_B<KIND>, lane, scat_local,
                                                             s_acc, &s_pnode[wave], chan0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // LDS only: the weight ring's DMA stays in flight across this barrier
        __builtin_amdgcn_s_barrier();                          // (phase-1 tiles end in barriers: rows are not restaged before wave 0 is done)
        if (wave == 0) merge_wave_partials<Spec<KIND>::D_OUT, n_acc_phase<KIND>(0), WAVES_B>(s_msg, MSG_LD_B<KIND>, lane, s_pnode, s_acc, chan0);
        }
#pragma unroll
        for (int i = 0; i < NACC<KIND>; ++i) acc[i] = 0.f;
        PROF_MARK(4);
      }
      if constexpr (TILE + R - 1 < TE) {        // slot (TILE-1)%R was released by the barrier that ended stage TILE-1
#if DP_ABLATE == 5
        gptrq src = w2p;                                        // every tile from the same 8-12 KB: L1/L2-hot source
#else
        gptrq src = w2p + (size_t)(TILE + R - 1) * NQ;
#endif
        asm volatile("" : "+s"(src));
        dma_tile<KIND, NS>(src, s_w + ((TILE + R - 1) % R) * NQ, wave, lane);
      }
      const unsigned lds_w = (unsigned)(__UINTPTR_TYPE__)(lptrq)s_w + lane * 16;   // LDS byte address of this lane's quad
      if constexpr (!(SKIPL0 && tile_l0_only<KIND, TILE>())) {
      f32x16 D = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#if DP_ABLATE == 2
      for (int r_ = 0; r_ < 16; ++r_) D[r_] = hb.v[0][0][0] == (__bf16)123.f ? 1.f : 0.5f;
#else
#if DP_ABLATE == 3
#pragma unroll
      for (int b = 0; b < NKB<KIND>; ++b) {
        if constexpr (NS == 3) {
          D = mma32<F16>(hb.v[b][2], hb.v[b][0], D);
          D = mma32<F16>(hb.v[b][1], hb.v[b][1], D);
          D = mma32<F16>(hb.v[b][0], hb.v[b][2], D);
        }
        D = mma32<F16>(hb.v[b][1], hb.v[b][0], D);
        D = mma32<F16>(hb.v[b][0], hb.v[b][1], D);
        D = mma32<F16>(hb.v[b][0], hb.v[b][0], D);
      }
#else
      {
        // A-operand fetches as raw ds_read_b128 + hand-counted lgkmcnt.  With ordinary LDS loads the compiler cannot prove that
        // they do not alias the LDS-DMA destinations and puts an s_waitcnt vmcnt(0) in front of the first read of every tile,
        // i.e. a full L2 round trip for the tile that was JUST requested (measured: 0.88 -> see profiles/r1_conv_ablation.md).
        bf16x8 W[WREGS<NS>];
        const unsigned la = lds_w + (TILE % R) * NQ * 16;
        OperandPipe<KIND, NS, 0, FULL, F16>::run(la, W, hb, D);
      }
#endif
#endif
#if DP_ABLATE == 1
      for (int r_ = 0; r_ < 16; ++r_) asm volatile("" ::"v"(D[r_]));
#else
      contract_tile<KIND, TILE>(D, xr, sh, M, acc, std::make_integer_sequence<int, 16>{});
#endif
      }
      if constexpr (TILE + 1 < TE) {
        // tile TILE+1 must have landed; tiles TILE+2 .. min(TILE+R-1, NT-1) may stay in flight across the barrier
        constexpr int last = (TILE + R - 1 < TE - 1) ? (TILE + R - 1) : (TE - 1);
        wait_vmcnt<(last - (TILE + 1)) * PIECES<KIND, NS>>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's ds_reads of the slot about to be recycled are done
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
      StageB<KIND, NS, TILE + 1, FULL, F16, EMIT, SKIPL0, PH>::run(w2p, wave, lane, s_w, hb, xr, sh, M, acc, row, half, s_msg, s_pnode, s_acc, scat_local, emit, emit_rows);
    } else {
      constexpr int PHL = PH == 0 ? 0 : (n_tiles_phase<KIND>(1) > 0 ? 1 : 0);
      PROF_MARK(5);
      if constexpr (PHL == 1 && n_tiles_phase<KIND>(1) < 2 && n_tiles_phase<KIND>(0) > 0) __syncthreads();   // wave 0 may still be merging phase 0
      stage_phase<KIND, PHL, 0>(acc, row, half, std::make_integer_sequence<int, n_acc_phase<KIND>(PHL) / 2>{});
      const auto chanl = [](int c) { return kChanLut<KIND, PHL>.v[c]; };
      if constexpr (EMIT) {
        wave_emit<Spec<KIND>::D_OUT, n_acc_phase<KIND>(PHL)>(s_msg + (size_t)wave * 32 * MSG_LD_B<KIND>, MSG_LD_B<KIND>, lane, emit, emit_rows, chanl);
        return;
      }
      wave_segsum<Spec<KIND>::D_OUT, n_acc_phase<KIND>(PHL)>(s_msg + (size_t)wave * 32 * MSG_LD_B<KIND>, MSG_LD_B<KIND>, lane, scat_local,
                                                             s_acc, &s_pnode[wave], chanl);
      __syncthreads();
      if (wave == 0) merge_wave_partials<Spec<KIND>::D_OUT, n_acc_phase<KIND>(PHL), WAVES_B>(s_msg, MSG_LD_B<KIND>, lane, s_pnode, s_acc, chanl);
      PROF_MARK(6);
    }
  }
};

// DP_PREC_BF16 (NS == 1, one product per K block): the same tile stream, software-pipelined inside the wave.  A one-product tile is four
// 32-cycle MFMAs behind an LDS round trip, followed by a contraction of about the same length that has to wait for them: with one D fragment per
// wave the matrix pipe idles through every contraction and the VALU through every operand fetch, and the second wave of the SIMD only hides part
// of it (profiles/r4_experiments.md 7: 0.09 ms MFMA / operand stream + 0.06 ms contraction + 0.09 ms per-chunk work of 0.24 ms).  Here stage TILE
// holds TWO fragments: Dcur = W(TILE) hid, finished during stage TILE-1, is contracted in NKB pieces, and between the pieces the K blocks of tile
// TILE+1 are fetched and multiplied into Dnext - the MFMAs run under the contraction's VALU instructions, the ds_reads under both.  Same
// arithmetic in the same order per accumulator: bit-identical to StageB<KIND, 1>.
//   ring: tile t lives in slot t % R.  Tile TILE+1 is read during stage TILE, so the barrier that ends stage TILE-1 releases slot TILE % R and
//   stage TILE starts by requesting tile TILE+R into it (R tiles requested in the prologue instead of R-1); the stage ends when tile TILE+2 has
//   landed in every wave.
// NB + 1 pieces of the 16 D registers: piece 0 (4 registers) covers the LDS round trip of the first two K blocks, pieces 1 .. NB-1 sit between the
// MFMAs, piece NB (4 registers) behind the last MFMA - so that the next stage, which starts by reading Dn, does not start inside that MFMA's latency
template <int KIND> constexpr int piped_cut(int i) {
  constexpr int NB = NKB<KIND>;
  return i <= 0 ? 0 : i > NB ? 16 : NB == 1 ? 8 : 4 + 8 * (i - 1) / (NB - 1);
}
template <int KIND, int TILE, int B, bool F16>
struct PipedTile {      // K blocks B.. of tile TILE+1 into Dn, pieces B.. of the contraction of D (tile TILE) around them
  static __device__ __forceinline__ void run(unsigned la, bf16x8 (&W)[2], const HidB<KIND, 1>& hb, f32x16& Dn, const f32x16& D,
                                             const float (&xr)[Spec<KIND>::D_IN / 2], const float (&sh)[Spec<KIND>::SH_DIM],
                                             const float (&M)[9], float (&acc)[NACC<KIND>]) {
    constexpr int NB = NKB<KIND>;
    if constexpr (B < NB) {
      if constexpr (B == 0) {
        lds_read_q(W[0], la, 0);
        if constexpr (NB > 1) lds_read_q(W[1], la, 1024);
      }
      constexpr int r0 = piped_cut<KIND>(B), r1 = piped_cut<KIND>(B + 1);
      contract_part<KIND, TILE, r0>(D, xr, sh, M, acc, std::make_integer_sequence<int, r1 - r0>{});
      lds_landed<(B + 1 < NB) ? 1 : 0>(W[B & 1]);
      Dn = mma32<F16>(W[B & 1], hb.v[B][0], Dn);
      if constexpr (B + 2 < NB) lds_read_q(W[B & 1], la, (B + 2) * 1024);
      PipedTile<KIND, TILE, B + 1, F16>::run(la, W, hb, Dn, D, xr, sh, M, acc);
    } else {
      constexpr int r0 = piped_cut<KIND>(NB);
      contract_part<KIND, TILE, r0>(D, xr, sh, M, acc, std::make_integer_sequence<int, 16 - r0>{});
    }
  }
};

template <int KIND, int TILE, bool F16 = false>
struct StageP {
  static __device__ __forceinline__ void run(gptrq w2p, int wave, int lane, vu4* s_w, const HidB<KIND, 1>& hb,
                                             const float (&xr)[Spec<KIND>::D_IN / 2], const float (&sh)[Spec<KIND>::SH_DIM],
                                             const float (&M)[9], float (&acc)[NACC<KIND>], float* row, int half,
                                             float* s_msg, int* s_pnode, float* s_acc, int scat_local, const f32x16& D) {
    using G = Geo<KIND>;
    constexpr int NT0 = n_tiles_phase<KIND>(0);
    constexpr int NQ = TILE_QUADS<KIND, 1>;
    constexpr int R = RING<1>;
    if constexpr (TILE < G::NT) {
      if constexpr (TILE == NT0 && NT0 > 0) {      // phase boundary (as in StageB); D = first tile of phase 1, already computed
        stage_phase<KIND, 0, 0>(acc, row, half, std::make_integer_sequence<int, n_acc_phase<KIND>(0) / 2>{});
        const auto chan0 = [](int c) { return kChanLut<KIND, 0>.v[c]; };
        wave_segsum<Spec<KIND>::D_OUT, n_acc_phase<KIND>(0)>(s_msg + (size_t)wave * 32 * MSG_LD_B<KIND>, MSG_LD_B<KIND>, lane, scat_local,
                                                             s_acc, &s_pnode[wave], chan0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0) merge_wave_partials<Spec<KIND>::D_OUT, n_acc_phase<KIND>(0), WAVES_B>(s_msg, MSG_LD_B<KIND>, lane, s_pnode, s_acc, chan0);
#pragma unroll
        for (int i = 0; i < NACC<KIND>; ++i) acc[i] = 0.f;
      }
      if constexpr (TILE + R < G::NT) {            // slot TILE % R: its reads ended with stage TILE-1
        gptrq src = w2p + (size_t)(TILE + R) * NQ;
        asm volatile("" : "+s"(src));
        dma_tile<KIND, 1>(src, s_w + (TILE % R) * NQ, wave, lane);
      }
      f32x16 Dn = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (TILE + 1 < G::NT) {
        const unsigned la = (unsigned)(__UINTPTR_TYPE__)(lptrq)s_w + lane * 16 + ((TILE + 1) % R) * NQ * 16;
        bf16x8 W[2];
        PipedTile<KIND, TILE, 0, F16>::run(la, W, hb, Dn, D, xr, sh, M, acc);
      } else {
        contract_tile<KIND, TILE>(D, xr, sh, M, acc, std::make_integer_sequence<int, 16>{});
      }
      if constexpr (TILE + 2 < G::NT) {
        // tile TILE+2 must have landed; tiles TILE+3 .. min(TILE+R, NT-1) may stay in flight across the barrier
        constexpr int last = (TILE + R < G::NT - 1) ? (TILE + R) : (G::NT - 1);
        wait_vmcnt<(last - (TILE + 2)) * PIECES<KIND, 1>>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's ds_reads of the slot about to be recycled are done
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
      StageP<KIND, TILE + 1, F16>::run(w2p, wave, lane, s_w, hb, xr, sh, M, acc, row, half, s_msg, s_pnode, s_acc, scat_local, Dn);
    } else {
      constexpr int PHL = n_tiles_phase<KIND>(1) > 0 ? 1 : 0;
      // wave 0 may still be merging phase 0 unless a stage of phase 1 ended in a barrier (stages NT0 .. NT-3 do)
      if constexpr (PHL == 1 && n_tiles_phase<KIND>(1) < 3 && n_tiles_phase<KIND>(0) > 0) __syncthreads();
      stage_phase<KIND, PHL, 0>(acc, row, half, std::make_integer_sequence<int, n_acc_phase<KIND>(PHL) / 2>{});
      const auto chanl = [](int c) { return kChanLut<KIND, PHL>.v[c]; };
      wave_segsum<Spec<KIND>::D_OUT, n_acc_phase<KIND>(PHL)>(s_msg + (size_t)wave * 32 * MSG_LD_B<KIND>, MSG_LD_B<KIND>, lane, scat_local,
                                                             s_acc, &s_pnode[wave], chanl);
      __syncthreads();
      if (wave == 0) merge_wave_partials<Spec<KIND>::D_OUT, n_acc_phase<KIND>(PHL), WAVES_B>(s_msg, MSG_LD_B<KIND>, lane, s_pnode, s_acc, chanl);
    }
  }
};

#ifndef DP_PIPE_NS1
#define DP_PIPE_NS1 1      // 0: the one-product kernel on StageB like the split kernels (A/B builds)
#endif

template <int KIND, int NS, bool FULL = false, bool F16 = false, bool EMIT = false, bool SKIPL0 = false, bool PSPLIT = false>
__global__ __launch_bounds__(WAVES_B * 64, 2) void tp_conv_bf16_kernel(DpConvArgs a) {
  using S = Spec<KIND>;
  using G = Geo<KIND>;
  constexpr int NQ = TILE_QUADS<KIND, NS>;
  constexpr int R = RING<NS>;
  static_assert(NQ % (WAVES_B * 64) == 0, "a packed tile must be a whole number of 1-KiB DMA pieces per wave");
  __shared__ __attribute__((aligned(16))) vu4 s_w[R * NQ];       // first: the DMA destination must be 16-byte aligned
  __shared__ float s_msg[CHUNK_B * MSG_LD_B<KIND>];
  __shared__ float s_acc[MAX_NODES_B * S::D_OUT];
  __shared__ int s_pnode[WAVES_B];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ecol = lane & 31, half = lane >> 5;
  static_assert(!PSPLIT || (EMIT && NS == 2), "phase split: list launches of the two-limb kernels");
  const WgRange r = wg_range(a, PSPLIT ? (int)(blockIdx.x >> 1) : -1);
  if (r.n_loc <= 0) return;
  for (int i = tid; i < r.n_loc * S::D_OUT; i += WAVES_B * 64) s_acc[i] = 0.f;
  gptrq w2p = (gptrq)a.w2p;

  PROF_MARK(-1);
  auto chunks = [&]<int PH>() {
  constexpr int TB = tile_begin<KIND, PH>(), TE = tile_end<KIND, PH>();
  for (int kc = r.k_begin; kc < r.k_end; kc += CHUNK_B) {
    const EdgeRef e = decode_edge(a, r, kc + wave * 32 + ecol);
    float xr[S::D_IN / 2], sh[S::SH_DIM], M[9];
    HidB<KIND, NS> hb;
    {
      float hid[32];
      gemm1_hidden<KIND>(a, e, lane, hid);
      split_hidden<KIND, NS, F16>(hid, hb);
    }
    PROF_MARK(0);
    load_tp_operands<KIND>(a, e, half, xr, sh, M);
    // keep the gathered operands in registers NOW: no ordinary global load may be pending once the DMA ring is running
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                              // previous chunk: partial sums merged, s_msg rows and the ring are free
    PROF_MARK(1);
    constexpr bool PIPED = NS == 1 && DP_PIPE_NS1 != 0 && DP_ABLATE == 0;     // StageP: tile t+1's MFMAs under tile t's contraction
    constexpr int NPRO = (R - (PIPED ? 0 : 1) < TE - TB) ? (R - (PIPED ? 0 : 1)) : TE - TB;
#pragma unroll
    for (int t = 0; t < NPRO; ++t) dma_tile<KIND, NS>(w2p + (size_t)(TB + t) * NQ, s_w + ((TB + t) % R) * NQ, wave, lane);
    float* row = s_msg + (size_t)(wave * 32 + ecol) * MSG_LD_B<KIND>;
    float acc[NACC<KIND>];
#pragma unroll
    for (int i = 0; i < NACC<KIND>; ++i) acc[i] = 0.f;
    wait_vmcnt<(NPRO - 1) * PIECES<KIND, NS>>();   // tile 0 landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PROF_MARK(2);
    if constexpr (PIPED) {
      // tile 0's fragment before the first stage; then tile 1 landed in every wave and slot 0 released
      f32x16 D0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      {
        bf16x8 W[WREGS<NS>];
        OperandPipe<KIND, NS, 0, FULL, F16>::run((unsigned)(__UINTPTR_TYPE__)(lptrq)s_w + lane * 16, W, hb, D0);
      }
      if constexpr (G::NT > 1) {
        wait_vmcnt<(NPRO - 2) * PIECES<KIND, NS>>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NS == 1)
        StageP<KIND, 0, F16>::run(w2p, wave, lane, s_w, hb, xr, sh, M, acc, row, half, s_msg, s_pnode, s_acc, e.scat_local, D0);
    } else if constexpr (EMIT) {
      const int k0 = kc + wave * 32, left = r.k_end - k0;
      StageB<KIND, NS, TB, FULL, F16, true, SKIPL0, PH>::run(w2p, wave, lane, s_w, hb, xr, sh, M, acc, row, half, s_msg, s_pnode, s_acc, e.scat_local,
                                                             a.msg_out + (size_t)k0 * S::D_OUT, left < 0 ? 0 : (left > 32 ? 32 : left));
    } else {
      StageB<KIND, NS, 0, FULL, F16>::run(w2p, wave, lane, s_w, hb, xr, sh, M, acc, row, half, s_msg, s_pnode, s_acc, e.scat_local);
    }
  }
  };
  if constexpr (PSPLIT) {
    if (blockIdx.x & 1) chunks.template operator()<1>();
    else chunks.template operator()<0>();
  } else {
    chunks.template operator()<-1>();
  }
  if constexpr (EMIT) return;
  __syncthreads();
  conv_epilogue<KIND, WAVES_B * 64>(a, r, s_acc);
  PROF_MARK(7);
}

template <int KIND> int launch(const DpConvArgs& a_in, hipStream_t st) {
  DpConvArgs a = a_in;
  if (a.mode == DP_EDGES_LIST) {                 // explicit edge list, messages written out: the two-limb fp16 form of the layer convs only
    if constexpr (KIND <= 3) {
      if (a.precision != DP_PREC_F16X3) return DP_E_UNSUPPORTED;
      if (a.n_list <= 0) return DP_OK;
      const dim3 grid_l((unsigned)((a.n_list + CHUNK_B - 1) / CHUNK_B));
      // phase_split (whole convs only): two workgroups per chunk, the tiles of one accumulation phase each - half the latency of a launch that cannot
      // fill the chip anyway (a small batch's phore-phore conv sits on the critical path of its layer)
      if (a.phase_split && !a.sh0_zero && n_tiles_phase<KIND>(0) > 0 && n_tiles_phase<KIND>(1) > 0) {
        hipLaunchKernelGGL((tp_conv_bf16_kernel<KIND, 2, false, true, true, false, true>), dim3(grid_l.x * 2), dim3(WAVES_B * 64), 0, st, a);
        return (int)hipGetLastError();
      }
      // sh0_zero: the tiles whose rows all belong to l = 0 paths multiply by a zero and are skipped (their own instantiation)
      if (a.sh0_zero) hipLaunchKernelGGL((tp_conv_bf16_kernel<KIND, 2, false, true, true, true>), grid_l, dim3(WAVES_B * 64), 0, st, a);
      else hipLaunchKernelGGL((tp_conv_bf16_kernel<KIND, 2, false, true, true, false>), grid_l, dim3(WAVES_B * 64), 0, st, a);
      return (int)hipGetLastError();
    } else {
      return DP_E_UNSUPPORTED;
    }
  }
  // phase-split needs both phases to exist and is implemented by the fp32-MFMA kernel only
  if (a.precision != DP_PREC_F32 || n_tiles_phase<KIND>(0) == 0 || n_tiles_phase<KIND>(1) == 0) a.phase_split = 0;
  dim3 grid((unsigned)(a.B * a.slices * (a.phase_split ? 2 : 1)));
  if (a.precision == DP_PREC_F32) hipLaunchKernelGGL(tp_conv_kernel<KIND>, grid, dim3(WAVES * 64), 0, st, a);
  else if (a.precision == DP_PREC_BF16X6) hipLaunchKernelGGL((tp_conv_bf16_kernel<KIND, 3>), grid, dim3(WAVES_B * 64), 0, st, a);
  else if (a.precision == DP_PREC_BF16X3) hipLaunchKernelGGL((tp_conv_bf16_kernel<KIND, 2>), grid, dim3(WAVES_B * 64), 0, st, a);
  else if (a.precision == DP_PREC_BF16X9) hipLaunchKernelGGL((tp_conv_bf16_kernel<KIND, 3, true>), grid, dim3(WAVES_B * 64), 0, st, a);
  else if (a.precision == DP_PREC_BF16) hipLaunchKernelGGL((tp_conv_bf16_kernel<KIND, 1>), grid, dim3(WAVES_B * 64), 0, st, a);
  else if (a.precision == DP_PREC_F16X3) hipLaunchKernelGGL((tp_conv_bf16_kernel<KIND, 2, false, true>), grid, dim3(WAVES_B * 64), 0, st, a);
  else return DP_E_BADARG;
  return (int)hipGetLastError();
}

template <int KIND> int w2p_floats(int prec) {
  if (prec == DP_PREC_F32) return Geo<KIND>::W2P_FLOATS;
  if (prec == DP_PREC_BF16X6 || prec == DP_PREC_BF16X9) return Geo<KIND>::NT * TILE_QUADS<KIND, 3> * 4;
  if (prec == DP_PREC_BF16X3) return Geo<KIND>::NT * TILE_QUADS<KIND, 2> * 4;
  if (prec == DP_PREC_BF16) return Geo<KIND>::NT * TILE_QUADS<KIND, 1> * 4;
  if (prec == DP_PREC_F16X3) return Geo<KIND>::NT * TILE_QUADS<KIND, 2> * 4 + 4;      // + one trailing quad: float[0] = the image's exponent w
  return DP_E_BADARG;
}

}  // namespace dp

// launcher of one layer kind, defined in dp_conv_k<kind>.hip
int dp_conv_launch_k0(const DpConvArgs& a, hipStream_t st);
int dp_conv_launch_k1(const DpConvArgs& a, hipStream_t st);
int dp_conv_launch_k2(const DpConvArgs& a, hipStream_t st);
int dp_conv_launch_k3(const DpConvArgs& a, hipStream_t st);
int dp_conv_launch_k4(const DpConvArgs& a, hipStream_t st);
int dp_conv_launch_k5(const DpConvArgs& a, hipStream_t st);
