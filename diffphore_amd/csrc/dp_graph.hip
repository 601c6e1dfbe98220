// dp_graph.hip - per-step graph construction, embeddings and score heads of the DiffPhore score model for gfx950.
//
// Everything here is small, latency/launch-bound work around the fused TP convs (dp_conv.hip): one workgroup per
// pose, atoms / edges of the pose spread over the threads, per-pose constants staged in LDS, MLP weights read through
// wave-uniform (scalar) loads.  No host synchronisation anywhere.  Reference line numbers: score_model_phore.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "dp_dev.h"

#ifndef DP_NODE_LL_MAX_ATOMS
// The ligand-ligand conv runs node-side (pair-dense: N x N slots) while every ligand of the batch fits ONE 32-slot pass; beyond, the
// pair slots are < 50 % full (radius graph capped at 32 neighbours) and the edge-side CSR kernel in its fp32-grade 16-bit form is faster.
// Round 3 (fp32 node-side kernel vs edge-side bf16x6): 256 poses x 80 atoms 8.93 vs 9.53 ms/step, the 15-45-atom virtual-screen batch
// 18.80 vs 19.14, 512 x 30 atoms 5.74 vs 5.50 (node-side stays).  Round 5, both forms on two fp16 limbs (node-side 0.48 ns per pair SLOT,
// edge-side 0.78 ns per EDGE at layer 2: the node form needs > 60 % of its slots filled): 256 x 80 atoms 48.9k (edge-side) vs 41.6k
// pose-steps/s (three pair passes), 15-45 atoms 59.9k vs 58.2k, 512 x 30 atoms 140.3k vs 146.8k (node-side stays); against the slower
// bf16x6 edge-side form the three-pass node form had still won (40.3k vs 41.6k), which is why the cap briefly stood at 96.
// DpWork.ll_node_max_atoms overrides it.  tools/gpu_ll_policy.sh
#define DP_NODE_LL_MAX_ATOMS 32
#endif

namespace dp {

constexpr int LIG_DIMS[16] = {119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2};   // process_mols.py:162-179

// ---------------------------------------------------------------------------------------------------------------------
// time parameters per pose: sinusoidal sigma embedding, sigmas, SO(3)/torus score norms
//   diffusion_utils.py:16-20,82-93 ; so3.py:92-96 ; torus.py:82-86 ; score_model_phore.py:333,352,376
// ---------------------------------------------------------------------------------------------------------------------
// 32 lanes per pose, one output each (the 20 double-precision sin / cos of the embedding had been one thread's serial chain at the very start of a step)
// t_row (DpWork.t_row): one row of times for every pose - read instead of t[p], and written there (lanes 24..26)
__global__ __launch_bounds__(256) void time_params_kernel(int B, float* __restrict__ t, const float* __restrict__ t_row, DpWeights w, float* __restrict__ tp) {
  const int p = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;
  if (p >= B) return;
  const float* ts = t_row ? t_row : t + (size_t)p * 3;
  const float t_tr = ts[0], t_rot = ts[1], t_tor = ts[2];
  if (t_row && j >= 24 && j < 27) t[(size_t)p * 3 + (j - 24)] = ts[j - 24];
  float* o = tp + (size_t)p * TP_STRIDE;
  if (j < 20) {
    const float st = __fmul_rn(w.emb_scale, t_tr);                 // embedding_scale * x      (float32)
    const float arg = __fmul_rn(st, w.emb_freq[j < 10 ? j : j - 10]);      // timesteps[:,None] * emb[None,:]
    o[j] = j < 10 ? (float)sin((double)arg) : (float)cos((double)arg);
  } else if (j == 20) {
    o[20] = powf(w.tr_smin, 1.f - t_tr) * powf(w.tr_smax, t_tr);
  } else if (j == 21) {
    const float s_rot = powf(w.rot_smin, 1.f - t_rot) * powf(w.rot_smax, t_rot);
    o[21] = s_rot;
    // so3.score_norm: nearest index on a log10 grid of 1000 eps in [0.01, 2]
    const double x = (log10((double)s_rot) - log10(0.01)) / (log10(2.0) - log10(0.01)) * 1000.0;
    int idx = (int)rint(x);
    idx = idx < 0 ? 0 : (idx > 999 ? 999 : idx);
    o[23] = w.so3_table[idx];
  } else if (j == 22) {
    const float s_tor = powf(w.tor_smin, 1.f - t_tor) * powf(w.tor_smax, t_tor);
    o[22] = s_tor;
    // torus.score_norm: index on a log grid of 5001 sigma/pi in [3e-3, 2]; clip BEFORE round (torus.py:85)
    double x = (log((double)s_tor / M_PI) - log(3e-3)) / (log(2.0) - log(3e-3)) * 5000.0;
    x = x < 0.0 ? 0.0 : (x > 5000.0 ? 5000.0 : x);
    const int idx = (int)rint(x);
    o[24] = sqrtf(w.torus_table[idx]);
  }
}

// the time parameters of one row of times (DpWork.t_row), computed before (DpWork.tp_row), for every pose; the times themselves into DpBatch.t
__global__ __launch_bounds__(256) void tp_bcast_kernel(int B, float* __restrict__ t, const float* __restrict__ t_row, const float* __restrict__ tp_row,
                                                       float* __restrict__ tp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * TP_STRIDE) return;
  const int p = i / TP_STRIDE, j = i - p * TP_STRIDE;
  if (j < 25) tp[i] = tp_row[j];
  else if (j < 28) t[(size_t)p * 3 + (j - 25)] = t_row[j - 25];
}

// ---------------------------------------------------------------------------------------------------------------------
// one-time static tables
// ---------------------------------------------------------------------------------------------------------------------
__global__ void static_nodes_kernel(DpBatch b, DpWeights w, DpWork k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b.n_lig) {
    // AtomEncoder categorical part (:64-69): sum of 16 embedding rows
    float acc[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) acc[c] = 0.f;
    int off = 0;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const float* row = w.lig_emb + (size_t)(off + b.lig_x[i * 16 + f]) * NS;
#pragma unroll
      for (int c = 0; c < NS; ++c) acc[c] += row[c];
      off += LIG_DIMS[f];
    }
    store_row<NS>(k.lig_x_emb + (size_t)i * NS, acc);
    // static parts of the cross edge MLPs per (atom, phore type):  P = [A | onehot(type) | phorefp[atom]]  (:790-796)
    float fp[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) fp[t] = b.phorefp[i * NPT + t];
    for (int ty = 0; ty < NPT; ++ty) {
      const float a = ty == EX ? 0.f : fp[ty];
      float pre[NS];
#pragma unroll
      for (int o = 0; o < NS; ++o) {
        const float* W = w.x_edge.w0 + o * 73 + 40;
        float s = w.x_edge.b0[o] + a * W[ty] + W[11 + ty];
#pragma unroll
        for (int t = 0; t < NPT; ++t) s = fmaf(W[22 + t], fp[t], s);
        pre[o] = s;
      }
      store_row<NS>(k.pair_pre + ((size_t)i * NPT + ty) * NS, pre);
      // phoretype_match_transition: Linear(33,11), ReLU, Linear(11,1), Softplus  (:571-573, :807)
      float z = w.pmt.b3[0];
#pragma unroll
      for (int o = 0; o < NPT; ++o) {
        const float* W = w.pmt.w0 + o * 33;
        float s = w.pmt.b0[o] + a * W[ty] + W[11 + ty];
#pragma unroll
        for (int t = 0; t < NPT; ++t) s = fmaf(W[22 + t], fp[t], s);
        z = fmaf(w.pmt.w3[o], fmaxf(s, 0.f), z);
      }
      k.pair_fm[(size_t)i * NPT + ty] = softplus_f(z);
    }
  }
  if (i < b.n_ph) {
    // phore AtomEncoder: 3 embeddings + Linear on [alpha, weight | sigma emb]; the sigma part is added per step (:64-73, :662)
    float acc[NS];
    const float al = b.ph_scalar[i * 2 + 0], wt = b.ph_scalar[i * 2 + 1];
    const float* r0 = w.ph_emb + (size_t)b.ph_x[i * 3 + 0] * NS;
    const float* r1 = w.ph_emb + (size_t)(11 + b.ph_x[i * 3 + 1]) * NS;
    const float* r2 = w.ph_emb + (size_t)(13 + b.ph_x[i * 3 + 2]) * NS;
#pragma unroll
    for (int c = 0; c < NS; ++c)
      acc[c] = r0[c] + r1[c] + r2[c] + (w.ph_lin_b[c] + w.ph_lin_w[c * 22 + 0] * al + w.ph_lin_w[c * 22 + 1] * wt);
    store_row<NS>(k.ph_x_emb + (size_t)i * NS, acc);
  }
}

__global__ void static_pp_kernel(DpBatch b, DpWeights w, DpWork k) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= b.n_pp) return;
  const int s = b.pp_src[e], d = b.pp_dst[e];
  const float vx = b.ph_pos[d * 3 + 0] - b.ph_pos[s * 3 + 0];
  const float vy = b.ph_pos[d * 3 + 1] - b.ph_pos[s * 3 + 1];
  const float vz = b.ph_pos[d * 3 + 2] - b.ph_pos[s * 3 + 2];
  float sh[9];
  sh9(vx, vy, vz, sh);
  store_row<9>(k.pp_sh + (size_t)e * 9, sh);
  float g[NS], pre[NS];
  gauss20(sqrtf(vx * vx + vy * vy + vz * vz), w.ph_gauss_mu, w.ph_gauss_coeff, g);
#pragma unroll
  for (int o = 0; o < NS; ++o) pre[o] = w.ph_edge.b0[o];
  matvec_acc<NS, NS>(w.ph_edge.w0, 40, 20, g, pre);
  store_row<NS>(k.pp_pre + (size_t)e * NS, pre);
}

// the flag row ("this edge does not exist": feature 20 = 1.0) behind the table's last row
__device__ __forceinline__ void store_flag_row_bf16(uint16_t* __restrict__ tab, int n_rows, int tid) {
  if (tid < 32) tab[(size_t)n_rows * 32 + tid] = tid == 20 ? (uint16_t)0x3F80 : (uint16_t)0;
}
__device__ __forceinline__ void store_flag_row_f16x2(uint16_t* __restrict__ tab, int n_rows, int tid) {
  if (tid < 64) tab[(size_t)n_rows * 64 + tid] = tid == 20 ? (uint16_t)0x3C00 : (uint16_t)0;
}
// The edge kernels hold ONE ROW PER THREAD in registers, rows of consecutive threads consecutive in memory.  A thread that stores its own 36- / 80- /
// 128-byte row makes every store instruction of a wave touch 64 different cache lines (the 128-byte two-limb image had doubled both edge
// kernels, 50 -> 102 us; staged: cross_edges_kernel 77 us, lig_edges_kernel 92 us - that one is bound by its dependent gathers).  rows_store stages the workgroup's rows in LDS ([256][ROW | 1] dwords: an odd stride, conflict-free) and writes them back
// as dwords in memory order - 256 contiguous bytes per wave and instruction.  n_rows = live rows of the workgroup (a prefix of its threads),
// dst = the row of thread 0.  Every thread of the workgroup must call it (two workgroup barriers).
constexpr int ROWBUF_DWORDS = 256 * 33;
template <int ROW>
__device__ __forceinline__ void rows_store(uint32_t* __restrict__ sbuf, int tid, const uint32_t (&v)[ROW], int n_rows, uint32_t* __restrict__ dst) {
  constexpr int RP = ROW | 1;
  static_assert(256 * RP <= ROWBUF_DWORDS, "row buffer");
#pragma unroll
  for (int i = 0; i < ROW; ++i) sbuf[tid * RP + i] = v[i];
  __syncthreads();
  const int total = n_rows * ROW;
#pragma unroll
  for (int it = 0; it < ROW; ++it) {
    const int j = it * 256 + tid;
    if (j < total) {
      const int r = j / ROW, c = j - r * ROW;
      dst[j] = sbuf[r * RP + c];
    }
  }
  __syncthreads();
}
template <int N>
__device__ __forceinline__ void rows_store_f32(uint32_t* __restrict__ sbuf, int tid, const float (&v)[N], int n_rows, float* __restrict__ dst) {
  uint32_t u[N];
#pragma unroll
  for (int i = 0; i < N; ++i) u[i] = __builtin_bit_cast(uint32_t, v[i]);
  rows_store<N>(sbuf, tid, u, n_rows, reinterpret_cast<uint32_t*>(dst));
}
// the 16-bit image of an embedded edge row as dwords: two fp16 limbs (32 dwords = 128 bytes: hi limbs of features 0..19 | zeros to 32 | lo limbs |
// zeros; hi = rne16(v), lo = rne16(v - hi): the same split as dp_conv_node_f16_impl.h split2) or bf16 (16 dwords = 64 bytes)
__device__ __forceinline__ void row_image_f16x2(const float* v, uint32_t (&u)[32]) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < 32; ++i) u[i] = 0u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const h2 h = __builtin_convertvector((f2){v[2 * i], v[2 * i + 1]}, h2);
    const h2 l = __builtin_convertvector((f2){v[2 * i] - (float)h[0], v[2 * i + 1] - (float)h[1]}, h2);
    u[i] = __builtin_bit_cast(uint32_t, h);
    u[16 + i] = __builtin_bit_cast(uint32_t, l);
  }
}
__device__ __forceinline__ void row_image_bf16(const float* v, uint32_t (&u)[16]) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < 10; ++i) u[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){v[2 * i], v[2 * i + 1]}, bf2));
#pragma unroll
  for (int i = 10; i < 16; ++i) u[i] = 0u;
}
// the workgroup's rows of the 16-bit image (DpWork.node_f16, as dp_forward sets it in the kernels' copy of the workspace: which image the
// node-side convs of this forward read); row0 = the row of thread 0
__device__ __forceinline__ void rows_store_16(uint32_t* __restrict__ sbuf, int tid, const float* v, int n_rows, uint16_t* __restrict__ tab, size_t row0, int f16) {
  if (f16) {
    uint32_t u[32];
    row_image_f16x2(v, u);
    rows_store<32>(sbuf, tid, u, n_rows, reinterpret_cast<uint32_t*>(tab + row0 * 64));
  } else {
    uint32_t u[16];
    row_image_bf16(v, u);
    rows_store<16>(sbuf, tid, u, n_rows, reinterpret_cast<uint32_t*>(tab + row0 * 32));
  }
}
__device__ __forceinline__ void store_flag_row_16(uint16_t* __restrict__ tab, int n_rows, int tid, int f16) {
  if (f16) store_flag_row_f16x2(tab, n_rows, tid);
  else store_flag_row_bf16(tab, n_rows, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// ligand graph per pose: radius graph (torch_cluster CUDA semantics) + bond edges, CSR by scatter node, edge embedding, SH
//   build_lig_conv_graph :715-739 ; lig_edge_embedding :657
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lig_graph_kernel(DpBatch b, DpWork k) {
  constexpr int BCAP = 2048;                             // directed bonds of the pose staged in LDS (a 256-atom ligand has ~550); beyond: read in place
  __shared__ float s_pos[MAXN_L * 3];
  __shared__ uint32_t s_adj[MAXN_L * (MAXN_L / 32)];     // s_adj[centre][nbr/32]: bit set <=> edge (nbr -> centre) exists
  __shared__ int s_deg[MAXN_L], s_row[MAXN_L + 1];
  __shared__ int s_bond[BCAP];                           // src (local) | dst (local) << 8 | type << 16
  const int p = blockIdx.x, tid = threadIdx.x;
  const int l0 = b.lig_ptr[p], N = b.lig_ptr[p + 1] - l0;
  const int bo0 = b.bond_ptr[p], nbond = b.bond_ptr[p + 1] - bo0;
  const int base = b.ll_ptr[p];
  const int words = (N + 31) / 32;
  for (int i = tid; i < N * 3; i += 256) s_pos[i] = b.pos[(size_t)l0 * 3 + i];
  for (int i = tid; i < N * words; i += 256) s_adj[i] = 0u;
  for (int e = tid; e < nbond && e < BCAP; e += 256)
    s_bond[e] = (b.bond_src[bo0 + e] - l0) | ((b.bond_dst[bo0 + e] - l0) << 8) | (b.bond_type[bo0 + e] << 16);
  auto bond = [&](int e) {
    return e < BCAP ? s_bond[e] : ((b.bond_src[bo0 + e] - l0) | ((b.bond_dst[bo0 + e] - l0) << 8) | (b.bond_type[bo0 + e] << 16));
  };
  const int pr0 = b.llp_ptr ? b.llp_ptr[p] : 0;
  const bool pairs = b.llp_ptr != nullptr && k.ll_pair_rad != nullptr && k.ll_pair_bond != nullptr;
  if (k.ll_emb_bf && p == 0) store_flag_row_16(k.ll_emb_bf, b.ll_cap, tid, k.node_f16);
  if (pairs)
    for (int i = tid; i < N * N; i += 256) { k.ll_pair_rad[pr0 + i] = -1; k.ll_pair_bond[pr0 + i] = -1; }
  __syncthreads();
  if (pairs) {
    // every edge must own its entry of the pair tables (lig_edges_kernel fills them): radius edges are unique per ordered pair by
    // construction, so the only way to lose one is a ligand graph that lists the same ordered bond twice -> flag the pose
    for (int e = tid; e < nbond; e += 256) {
      const int me = bond(e) & 0xFFFF;
      bool dup = false;
      for (int f = 0; f < e; ++f) dup = dup || (bond(f) & 0xFFFF) == me;
      if (dup) atomicOr(&k.status[p], 2);
    }
  }
  // radius search: for centre y scan x ascending, keep d^2 < 25, stop after 33 hits (incl. self), drop self.
  for (int y = tid; y < N; y += 256) {
    const float yx = s_pos[y * 3], yy = s_pos[y * 3 + 1], yz = s_pos[y * 3 + 2];
    int cnt = 0;
    for (int x = 0; x < N && cnt < 33; ++x) {
      const float dx = s_pos[x * 3] - yx, dy = s_pos[x * 3 + 1] - yy, dz = s_pos[x * 3 + 2] - yz;
      const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (d2 < 25.0f) {
        ++cnt;
        if (x != y) s_adj[y * words + (x >> 5)] |= 1u << (x & 31);     // own row only: no race
      }
    }
  }
  __syncthreads();
  // degree of scatter node a = #bond edges with src a + #centres y whose list contains a
  for (int a = tid; a < N; a += 256) {
    int deg = 0;
    for (int e = 0; e < nbond; ++e) deg += ((bond(e) & 0xFF) == a) ? 1 : 0;
    for (int y = 0; y < N; ++y) deg += (s_adj[y * words + (a >> 5)] >> (a & 31)) & 1u;
    s_deg[a] = deg;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int a = 0; a < N; ++a) { s_row[a] = run; run += s_deg[a]; }
    s_row[N] = run;
  }
  __syncthreads();
  for (int a = tid; a < N; a += 256) {
    k.ll_rowptr[l0 + a] = base + s_row[a];
    k.ll_rowend[l0 + a] = base + s_row[a + 1];
    int o = base + s_row[a];
    for (int e = 0; e < nbond; ++e) {
      const int be = bond(e);
      if ((be & 0xFF) == a) {
        k.ll_nbr[o] = l0 + ((be >> 8) & 0xFF);
        k.ll_src[o] = (l0 + a) | ((be >> 16) << 28);      // the bond type rides in the top bits until lig_edges_kernel has read it
        ++o;
      }
    }
    for (int y = 0; y < N; ++y)
      if ((s_adj[y * words + (a >> 5)] >> (a & 31)) & 1u) {
        k.ll_nbr[o] = l0 + y;
        k.ll_src[o] = (l0 + a) | (7 << 28);
        ++o;
      }
  }
}

// features of the ligand-ligand edges lig_graph_kernel listed: spherical harmonics, edge embedding (lig_edge_embedding :657), the pair tables
// of the node-side conv.  One edge per thread: `slices` workgroup slots per pose (enough for the batch's largest ligand at its full radius
// degree; a pose with more edges loops), a pose's unused slots leave at once.  (Inside lig_graph_kernel this pass was three dependent rounds
// on the pose's one workgroup.)
__global__ __launch_bounds__(256) void lig_edges_kernel(DpBatch b, DpWeights w, DpWork k, int slices) {
  __shared__ float s_c0[NS];
  __shared__ uint32_t s_rows[ROWBUF_DWORDS];
  const int p = blockIdx.x / slices, slice = blockIdx.x % slices, tid = threadIdx.x;
  const int l0 = b.lig_ptr[p], N = b.lig_ptr[p + 1] - l0;
  if (N <= 0) return;
  const int base = b.ll_ptr[p];
  const int E = k.ll_rowend[l0 + N - 1] - base;
  if (slice * 256 >= E) return;
  const int pr0 = b.llp_ptr ? b.llp_ptr[p] : 0;
  const bool pairs = b.llp_ptr != nullptr && k.ll_pair_rad != nullptr && k.ll_pair_bond != nullptr;
  if (tid < NS) {   // layer-0 bias + sigma-embedding part of the edge MLP (same for every edge of the pose)
    const float* se = k.tparams + (size_t)p * TP_STRIDE;
    float a = w.lig_edge.b0[tid];
    for (int i = 0; i < NS; ++i) a = fmaf(w.lig_edge.w0[tid * 44 + 4 + i], se[i], a);
    s_c0[tid] = a;
  }
  __syncthreads();
  // (every thread runs every round - the row stores are workgroup-wide - a thread past the pose's last edge recomputes that edge and stores nothing)
  for (int e0 = slice * 256; e0 < E; e0 += slices * 256) {
    const bool live = e0 + tid < E;
    const int n_rows = E - e0 < 256 ? E - e0 : 256;
    const int o = base + (live ? e0 + tid : E - 1);
    const size_t o0 = (size_t)(base + e0);
    // (a thread past the last edge re-reads slot E - 1 while that slot's own thread rewrites it below with the type bits stripped: either value
    //  gives a valid (src, dst) - only `bt` may differ - and such a thread stores nothing: a benign race)
    const int packed = k.ll_src[o];
    const int bt = (packed >> 28) & 7, src = packed & 0x0fffffff, dst = k.ll_nbr[o];
    if (live) {
      k.ll_src[o] = src;
      if (pairs) (bt < 4 ? k.ll_pair_bond : k.ll_pair_rad)[pr0 + (src - l0) * N + (dst - l0)] = o;    // pair tables of the node-side lig-lig conv
    }
    const float vx = b.pos[(size_t)dst * 3] - b.pos[(size_t)src * 3];
    const float vy = b.pos[(size_t)dst * 3 + 1] - b.pos[(size_t)src * 3 + 1];
    const float vz = b.pos[(size_t)dst * 3 + 2] - b.pos[(size_t)src * 3 + 2];
    float sh[9];
    sh9(vx, vy, vz, sh);
    rows_store_f32<9>(s_rows, tid, sh, n_rows, k.ll_sh + o0 * 9);
    float g[NS], h[NS], out[NS];
    gauss20(sqrtf(vx * vx + vy * vy + vz * vz), w.lig_gauss_mu, w.lig_gauss_coeff, g);
#pragma unroll
    for (int c = 0; c < NS; ++c) h[c] = s_c0[c] + (bt < 4 ? w.lig_edge.w0[c * 44 + bt] : 0.f);
    matvec_acc<NS, NS>(w.lig_edge.w0, 44, 24, g, h);
    mlp_tail20(w.lig_edge, h, out);
    rows_store_f32<NS>(s_rows, tid, out, n_rows, k.ll_emb + o0 * NS);
    if (k.ll_emb_bf) rows_store_16(s_rows, tid, out, n_rows, k.ll_emb_bf, o0, k.node_f16);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// node embeddings per pose (:649-655, :662, boarder_analyze :898-935) and per-step phore edge embedding (:663)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void node_embed_kernel(DpBatch b, DpWeights w, DpWork k) {
  __shared__ float s_ls[NS], s_ps[NS], s_pc[NS];
  __shared__ float s_pp[MAXN_P * 3];            // the pose's pharmacophore positions and types (boarder_analyze: distance to the nearest
  __shared__ int s_ty[MAXN_P];                  // exclusion sphere): read by every atom, so staged once instead of N_p global loads per atom
  const int p = blockIdx.x, tid = threadIdx.x;
  const int l0 = b.lig_ptr[p], NL = b.lig_ptr[p + 1] - l0, p0 = b.ph_ptr[p], NP = b.ph_ptr[p + 1] - p0;
  const float* se = k.tparams + (size_t)p * TP_STRIDE;
  if (tid < NS) {
    float a = w.lig_lin_b[tid], c = 0.f, d = 0.f;
    for (int i = 0; i < NS; ++i) {
      a = fmaf(w.lig_lin_w[tid * NS + i], se[i], a);
      c = fmaf(w.ph_lin_w[tid * 22 + 2 + i], se[i], c);
      d = fmaf(w.ph_edge.w0[tid * 40 + i], se[i], d);
    }
    s_ls[tid] = a; s_ps[tid] = c; s_pc[tid] = d;
  }
  for (int j = tid; j < NP; j += 256) s_ty[j] = b.ph_type[p0 + j];
  for (int j = tid; j < NP * 3; j += 256) s_pp[j] = b.ph_pos[(size_t)p0 * 3 + j];
  __syncthreads();
  for (int i = tid; i < NL; i += 256) {
    const float x = b.pos[(l0 + i) * 3], y = b.pos[(l0 + i) * 3 + 1], z = b.pos[(l0 + i) * 3 + 2];
    float dmin2 = -1.f;
    for (int j = 0; j < NP; ++j)
      if (s_ty[j] == EX) {
        const float dx = x - s_pp[j * 3], dy = y - s_pp[j * 3 + 1], dz = z - s_pp[j * 3 + 2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        dmin2 = (dmin2 < 0.f || d2 < dmin2) ? d2 : dmin2;
      }
    const float dmin = dmin2 < 0.f ? 1e9f : sqrtf(dmin2);
    float* o = k.h_l[0] + (size_t)(l0 + i) * 100;
    const float* xe = k.lig_x_emb + (size_t)(l0 + i) * NS;
    const kptr bl_b = K(w.boarder_lin_b), bl_w = K(w.boarder_lin_w);
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      float v = xe[c] + s_ls[c];
      float bd = bl_b[c] + bl_w[c] * dmin;
#pragma unroll
      for (int q = 0; q < 5; ++q) bd += w.boarder_emb[(q * 2 + (dmin <= (float)(q + 1) ? 1 : 0)) * NS + c];
      o[c] = v + bd;
    }
  }
  for (int j = tid; j < NP; j += 256) {
    float* o = k.h_p[0] + (size_t)(p0 + j) * 100;
    const float* xe = k.ph_x_emb + (size_t)(p0 + j) * NS;
#pragma unroll
    for (int c = 0; c < NS; ++c) o[c] = xe[c] + s_ps[c];
  }
  const int e0 = b.pp_ptr[p], e1 = b.pp_ptr[p + 1];
  for (int e = e0 + tid; e < e1; e += 256) {
    float h[NS], out[NS];
    const float* pre = k.pp_pre + (size_t)e * NS;
#pragma unroll
    for (int c = 0; c < NS; ++c) h[c] = pre[c] + s_pc[c];
    mlp_tail20(w.ph_edge, h, out);
    store_row<NS>(k.pp_emb + (size_t)e * NS, out);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dense ligand x pharmacophore cross edges per pose (_build_phoretype_cross_conv_graph :759-895, cross_edge_embedding :673)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_edges_kernel(DpBatch b, DpWeights w, DpWork k, int slices) {
  __shared__ float s_cs[NS];
  __shared__ float s_ew[256];                   // exp(total_weight) of the workgroup's edges, in edge order
  __shared__ float s_den[256];
  __shared__ uint32_t s_rows[ROWBUF_DWORDS];
  const int p = blockIdx.x / slices, slice = blockIdx.x % slices, tid = threadIdx.x;
  const int l0g = b.lig_ptr[p], NLp = b.lig_ptr[p + 1] - l0g, p0 = b.ph_ptr[p], NP = b.ph_ptr[p + 1] - p0;
  if (k.x_emb_bf && blockIdx.x == 0) store_flag_row_16(k.x_emb_bf, b.n_x, tid, k.node_f16);
  // ONE EDGE PER THREAD.  `slices` workgroup slots per pose are launched (enough for the batch's largest pose); a pose deals its atoms -
  // whole atoms: the softmax denominator is per atom - to ceil(N_l / floor(256 / N_p)) of them, at most 256 edges each, and its other slots
  // leave at once.  (One workgroup per pose ran 4 dependent rounds of ~3 000 instructions on two waves per SIMD, and evaluated the distance
  // embedding and the total-weight MLP twice: once for the softmax numerator, once for the rest.)
  const int apw = NP > 0 ? (256 / NP > 0 ? 256 / NP : 1) : 256;
  const int ng = NLp > 0 ? (NLp + apw - 1) / apw : 0;
  if (slice >= ng || NP <= 0) return;
  const int a_lo = (NLp * slice) / ng, a_hi = (NLp * (slice + 1)) / ng;
  const int l0 = l0g + a_lo, NL = a_hi - a_lo;               // this workgroup's atoms
  const int x0 = b.x_ptr[p] + a_lo * NP, E = NL * NP;        // E <= 256 (DpBatch limits: N_p <= 256)
  if (tid < NS) {
    const float* se = k.tparams + (size_t)p * TP_STRIDE;
    float a = 0.f;
    for (int i = 0; i < NS; ++i) a = fmaf(w.x_edge.w0[tid * 73 + i], se[i], a);
    s_cs[tid] = a;
  }
  const int e = tid;
  const bool live = e < E;
  const int i = live ? e / NP : 0, j = live ? e - i * NP : 0;
  const int li = l0 + i, pj = p0 + j, ty = b.ph_type[pj];
  float vx = b.ph_pos[pj * 3] - b.pos[li * 3];
  float vy = b.ph_pos[pj * 3 + 1] - b.pos[li * 3 + 1];
  float vz = b.ph_pos[pj * 3 + 2] - b.pos[li * 3 + 2];
  float g[NS];
  gauss20(sqrtf(vx * vx + vy * vy + vz * vz), w.x_gauss_mu, w.x_gauss_coeff, g);
  // total_weight and its exponential (:801-811, :839)
  float tw, ew;
  {
    const kptr c_b0 = K(w.cdt.b0), c_w0 = K(w.cdt.w0), c_w3 = K(w.cdt.w3);
    float z = K(w.cdt.b3)[0];
#pragma unroll
    for (int o = 0; o < 10; ++o) {
      float s = c_b0[o];
#pragma unroll
      for (int q = 0; q < NS; ++q) s = fmaf(c_w0[o * NS + q], g[q], s);
      z = fmaf(c_w3[o], fmaxf(s, 0.f), z);
    }
    const float dist = softplus_f(z);
    const float fm = k.pair_fm[(size_t)li * NPT + ty];
    tw = fm * (dist * 1.0f) * w.scaler;
    ew = expf(tw);
  }
  if (live) {
    s_ew[e] = ew;
    k.x_ew[x0 + e] = ew;
  }
  __syncthreads();
  if (tid < NL) {     // scatter(..., reduce='sum') over the atom's edges, in edge order
    float s = 0.f;
    for (int jj = 0; jj < NP; ++jj) s += s_ew[tid * NP + jj];
    s_den[tid] = s;
  }
  __syncthreads();
  // (no early exit: the row stores below are workgroup-wide - a thread without an edge recomputes edge 0 and stores nothing)
  const size_t r0 = (size_t)x0;
  // edge embedding: Linear(73,20) on [sigma emb | gauss | P], ReLU, Linear(20,20)
  {
    float h[NS], out[NS];
    const float* pre = k.pair_pre + ((size_t)li * NPT + ty) * NS;
#pragma unroll
    for (int c = 0; c < NS; ++c) h[c] = pre[c] + s_cs[c];
    matvec_acc<NS, NS>(w.x_edge.w0, 73, 20, g, h);
    mlp_tail20(w.x_edge, h, out);
    rows_store_f32<NS>(s_rows, tid, out, E, k.x_emb + r0 * NS);
    if (k.x_emb_bf) rows_store_16(s_rows, tid, out, E, k.x_emb_bf, r0, k.node_f16);
  }
  // sign rule (:813-816) and softmax weight (:839,:845-848)
  {
    const kptr p_b0 = K(w.pdt.b0), p_w0 = K(w.pdt.w0), p_w3 = K(w.pdt.w3);
    float dz = K(w.pdt.b3)[0];
#pragma unroll
    for (int o = 0; o < NPT; ++o) dz = fmaf(p_w3[o], leaky_f(fmaf(p_w0[o], tw, p_b0[o])), dz);
    const float dir = leaky_f(dz) < 0.f ? -1.f : 1.f;
    const float aw = ew / s_den[i];
    vx = vx * dir * aw; vy = vy * dir * aw; vz = vz * dir * aw;
    float sh[9];
    sh9(vx, vy, vz, sh);
    rows_store_f32<9>(s_rows, tid, sh, E, k.x_sh + r0 * 9);
  }
  // normal-vector rule (:874-889); only the phore's own type can agree (phoretype is one-hot)
  {
    const float a = ty == EX ? 0.f : b.phorefp[li * NPT + ty];
    const float* ln = b.norm + (size_t)li * 33 + ty * 3;
    const float lx = a * ln[0], ly = a * ln[1], lz = a * ln[2];
    const float px = b.ph_norm[pj * 3], py = b.ph_norm[pj * 3 + 1], pz = b.ph_norm[pj * 3 + 2];
    float rx = fmaxf(ly * pz - lz * py, 1e-12f) * a;       // torch.clip(cross, 1e-12) * sum(agreement)  (quirk N4)
    float ry = fmaxf(lz * px - lx * pz, 1e-12f) * a;
    float rz = fmaxf(lx * py - ly * px, 1e-12f) * a;
    const float rn = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
    rx /= rn; ry /= rn; rz /= rn;
    const float lnm = sqrtf(lx * lx + ly * ly + lz * lz), pnm = sqrtf(px * px + py * py + pz * pz);
    const float mx = lx * pnm - lnm * px, my = ly * pnm - lnm * py, mz = lz * pnm - lnm * pz;
    const float qx = lx * pnm + lnm * px, qy = ly * pnm + lnm * py, qz = lz * pnm + lnm * pz;
    const float ang = 2.f * atan2f(sqrtf(mx * mx + my * my + mz * mz), sqrtf(qx * qx + qy * qy + qz * qz));
    const float d1 = ang - a * b.norm_angle1[li * NPT + ty], d2 = ang - a * b.norm_angle2[li * NPT + ty];
    const float dl = fabsf(d2) < fabsf(d1) ? d2 : d1;
    float sh[9];
    sh9(rx * dl, ry * dl, rz * dl, sh);
    rows_store_f32<9>(s_rows, tid, sh, E, k.x_nsh + r0 * 9);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// centre edges (build_center_conv_graph :381-406, center_edge_embedding :249-254,336)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void center_edges_kernel(DpBatch b, DpWeights w, DpWork k) {
  __shared__ float s_c[3];
  __shared__ float s_cs[NS];
  __shared__ float s_pos[MAXN_L * 3];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int l0 = b.lig_ptr[p], N = b.lig_ptr[p + 1] - l0;
  for (int i = tid; i < N * 3; i += 256) s_pos[i] = b.pos[(size_t)l0 * 3 + i];
  __syncthreads();
  if (tid == 0) {     // index_add_ over the atoms in order, then / count
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = 0; i < N; ++i) { sx += s_pos[i * 3]; sy += s_pos[i * 3 + 1]; sz += s_pos[i * 3 + 2]; }
    s_c[0] = sx / (float)N; s_c[1] = sy / (float)N; s_c[2] = sz / (float)N;
  }
  if (tid >= 64 && tid < 64 + NS) {
    const int c = tid - 64;
    const float* se = k.tparams + (size_t)p * TP_STRIDE;
    float a = w.center_edge.b0[c];
    for (int i = 0; i < NS; ++i) a = fmaf(w.center_edge.w0[c * 40 + 20 + i], se[i], a);
    s_cs[c] = a;
  }
  __syncthreads();
  for (int i = tid; i < N; i += 256) {
    const float vx = s_pos[i * 3] - s_c[0], vy = s_pos[i * 3 + 1] - s_c[1], vz = s_pos[i * 3 + 2] - s_c[2];
    float sh[9], g[NS], h[NS], out[NS];
    sh9(vx, vy, vz, sh);
    store_row<9>(k.c_sh + (size_t)(l0 + i) * 9, sh);
    gauss20(sqrtf(vx * vx + vy * vy + vz * vz), w.center_gauss_mu, w.center_gauss_coeff, g);
#pragma unroll
    for (int c = 0; c < NS; ++c) h[c] = s_cs[c];
    matvec_acc<NS, NS>(w.center_edge.w0, 40, 0, g, h);
    mlp_tail20(w.center_edge, h, out);
    store_row<NS>(k.c_emb + (size_t)(l0 + i) * NS, out);
  }
}

// tr / rot heads (get_trtheta_score :340-352)
// one wave per pose: lanes 0..19 hold the hidden units of the translation head, lanes 32..51 those of the rotation head; lanes 0 / 32 add them up in
// the reference's order (the same arithmetic as one thread per pose, a 20th of its dependent chain)
__global__ __launch_bounds__(256) void trrot_head_kernel(int B, DpWeights w, DpWork k, float* __restrict__ tr, float* __restrict__ rot) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, hsel = lane >> 5, o = lane & 31;
  if (p >= B) return;
  const float* g = k.gpred + (size_t)p * 12;
  const float* tp = k.tparams + (size_t)p * TP_STRIDE;
  float v[3];
  for (int c = 0; c < 3; ++c) v[c] = hsel == 0 ? g[c] + g[6 + c] : g[3 + c] + g[9 + c];
  const DpMlp& m = hsel == 0 ? w.tr_final : w.rot_final;
  const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  float r = 0.f;
  if (o < NS) {
    float s = fmaf(m.w0[o * 21], nrm, m.b0[o]);
    for (int i = 0; i < NS; ++i) s = fmaf(m.w0[o * 21 + 1 + i], tp[i], s);
    r = fmaxf(s, 0.f);
  }
  float out = m.b3[0];
#pragma unroll
  for (int j = 0; j < NS; ++j) out = fmaf(m.w3[j], __shfl(r, (lane & 32) + j, 64), out);
  if (o == 0) {
    float* dst = (hsel == 0 ? tr : rot) + (size_t)p * 3;
    for (int c = 0; c < 3; ++c) {
      float x = v[c] / nrm * out;
      x = hsel == 0 ? x / tp[20] : x * tp[23];
      dst[c] = x;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// torsion graph (build_bond_conv_graph :409-437, FullTensorProduct(sh, 2e) :276,365-366, edge attrs :368-369)
// ---------------------------------------------------------------------------------------------------------------------
// Geometry only (positions of this step): dp_forward runs it beside the last interaction layer, whose output only tor_hb_kernel below needs.
__global__ __launch_bounds__(256) void tor_graph_kernel(DpBatch b, DpWeights w, DpWork k) {
  __shared__ float s_pos[MAXN_L * 3];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int l0 = b.lig_ptr[p], N = b.lig_ptr[p + 1] - l0;
  const int t0 = b.tor_ptr[p], T = b.tor_ptr[p + 1] - t0;
  if (T == 0) return;
  for (int i = tid; i < N * 3; i += 256) s_pos[i] = b.pos[(size_t)l0 * 3 + i];
  __syncthreads();
  // one thread per bond builds the capped neighbour list (torch_cluster.radius, max 32, index order)
  for (int q = tid; q < T; q += 256) {
    const int u = b.tor_u[t0 + q] - l0, v = b.tor_v[t0 + q] - l0;
    const float cx = (s_pos[u * 3] + s_pos[v * 3]) / 2.f, cy = (s_pos[u * 3 + 1] + s_pos[v * 3 + 1]) / 2.f,
                cz = (s_pos[u * 3 + 2] + s_pos[v * 3 + 2]) / 2.f;
    const int slot0 = (t0 + q) * TOR_SLOTS;
    int cnt = 0;
    for (int x = 0; x < N && cnt < TOR_SLOTS; ++x) {
      const float dx = s_pos[x * 3] - cx, dy = s_pos[x * 3 + 1] - cy, dz = s_pos[x * 3 + 2] - cz;
      const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (d2 < 25.0f) {
        k.t_nbr[slot0 + cnt] = l0 + x;
        k.t_src[slot0 + cnt] = t0 + q;
        ++cnt;
      }
    }
    k.t_rowptr[t0 + q] = slot0;
    k.t_rowend[t0 + q] = slot0 + cnt;
    for (int c = cnt; c < TOR_SLOTS; ++c) { k.t_nbr[slot0 + c] = l0; k.t_src[slot0 + c] = -1; }
  }
  __syncthreads();
  for (int e = tid; e < T * TOR_SLOTS; e += 256) {
    const int q = e / TOR_SLOTS, slot = (t0 + q) * TOR_SLOTS + (e - q * TOR_SLOTS);
    if (k.t_src[slot] < 0) continue;
    const int u = b.tor_u[t0 + q] - l0, v = b.tor_v[t0 + q] - l0, x = k.t_nbr[slot] - l0;
    const float cx = (s_pos[u * 3] + s_pos[v * 3]) / 2.f, cy = (s_pos[u * 3 + 1] + s_pos[v * 3 + 1]) / 2.f,
                cz = (s_pos[u * 3 + 2] + s_pos[v * 3 + 2]) / 2.f;
    const float vx = s_pos[x * 3] - cx, vy = s_pos[x * 3 + 1] - cy, vz = s_pos[x * 3 + 2] - cz;
    float sh[9], bsh[9];
    sh9(vx, vy, vz, sh);
    sh9(s_pos[v * 3] - s_pos[u * 3], s_pos[v * 3 + 1] - s_pos[u * 3 + 1], s_pos[v * 3 + 2] - s_pos[u * 3 + 2], bsh);
    const float* Y1 = sh + 1;
    const float* Y2 = sh + 4;
    const float* bb = bsh + 4;                 // Y_2e(bond vector)
    float o7[7];
    // FullTensorProduct 'uvuv', coefficient sqrt(2 l_out + 1):
    //   0e <- 2e x 2e : sqrt(1) * (Y2 . b) / sqrt(5)
    o7[0] = (Y2[0] * bb[0] + Y2[1] * bb[1] + Y2[2] * bb[2] + Y2[3] * bb[3] + Y2[4] * bb[4]) * 0.4472135954999579f;
    //   1o <- 1o x 2e : sqrt(3) * sum_ij C(1,2,1)[i,j,k] Y1_i b_j = sqrt(3/10) * (M(b) Y1)_k
    {
      const float c = 0.57735026918962576451f * bb[2];
      const float m00 = -c - bb[4], m11 = 2.f * c, m22 = -c + bb[4], m01 = bb[1], m02 = bb[0], m12 = bb[3];
      const float f = 0.5477225575051661f;
      o7[1] = f * (m00 * Y1[0] + m01 * Y1[1] + m02 * Y1[2]);
      o7[2] = f * (m01 * Y1[0] + m11 * Y1[1] + m12 * Y1[2]);
      o7[3] = f * (m02 * Y1[0] + m12 * Y1[1] + m22 * Y1[2]);
    }
    //   1e <- 2e x 2e : sqrt(3) * sum_ij C(2,2,1)[i,j,k] Y2_i b_j   (C(2,2,1) * sqrt(10): 1/sqrt3, 1, 2/sqrt3 entries)
    {
      const float a = 0.57735026918962576451f, c2 = 1.1547005383792517f, f = 0.5477225575051661f;
      const float k0 = -a * Y2[0] * bb[1] + a * Y2[1] * bb[0] + Y2[2] * bb[3] - Y2[3] * bb[2] + a * Y2[3] * bb[4] - a * Y2[4] * bb[3];
      const float k1 = -c2 * Y2[0] * bb[4] - a * Y2[1] * bb[3] + a * Y2[3] * bb[1] + c2 * Y2[4] * bb[0];
      const float k2 = a * Y2[0] * bb[3] + Y2[1] * bb[2] + a * Y2[1] * bb[4] - Y2[2] * bb[1] - a * Y2[3] * bb[0] - a * Y2[4] * bb[1];
      o7[4] = f * k0; o7[5] = f * k1; o7[6] = f * k2;
    }
    store_row<7>(k.t_sh + (size_t)slot * 7, o7);
    float g[NS], h[NS], out[NS];
    gauss20(sqrtf(vx * vx + vy * vy + vz * vz), w.lig_gauss_mu, w.lig_gauss_coeff, g);
#pragma unroll
    for (int c = 0; c < NS; ++c) h[c] = w.final_edge.b0[c];
    matvec_acc<NS, NS>(w.final_edge.w0, 20, 0, g, h);
    mlp_tail20(w.final_edge, h, out);
    store_row<NS>(k.t_emb + (size_t)slot * NS, out);
  }
}

// the bonds' node attribute of the torsion conv: (h_l[u] + h_l[v])[:20] of the final node features (:368)
__global__ __launch_bounds__(256) void tor_hb_kernel(DpBatch b, DpWork k, const float* __restrict__ h_l) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.n_tor * NS) return;
  const int q = i / NS, c = i - q * NS;
  k.t_hb[i] = h_l[(size_t)b.tor_u[q] * 100 + c] + h_l[(size_t)b.tor_v[q] * 100 + c];
}

// tor_final_layer: Linear(40,20,no bias), Tanh, Linear(20,1,no bias); * sqrt(torus.score_norm(sigma))  (:286-291,372-377)
// 32 lanes per bond: lane o < 20 holds hidden unit o (its 40-term dot product in the reference's order, tanh), lane 0 adds the 20 terms up in order -
// the same arithmetic as one thread per bond, a 20th of its dependent chain (the launch ends the torsion side of a step: 24 -> 8 us at 320 bonds)
__global__ __launch_bounds__(256) void tor_head_kernel(DpBatch b, DpWeights w, DpWork k, float* __restrict__ tor) {
  const int q = blockIdx.x * 8 + (threadIdx.x >> 5), o = threadIdx.x & 31;
  const bool live = q < b.n_tor;
  float t = 0.f;
  if (live && o < NS) {
    const float* x = k.t_q + (size_t)q * 40;
    const float* w0 = w.tor_w0 + o * 40;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 40; ++i) s = fmaf(w0[i], x[i], s);
    t = tanhf(s);
  }
  float out = 0.f;
#pragma unroll
  for (int j = 0; j < NS; ++j) out = fmaf(w.tor_w3[j], __shfl(t, (threadIdx.x & 32) + j, 64), out);
  if (live && o == 0) {
    // pose of this bond: binary search in tor_ptr
    int lo = 0, hi = b.B;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b.tor_ptr[mid] <= q) lo = mid; else hi = mid; }
    tor[q] = out * k.tparams[(size_t)lo * TP_STRIDE + 24];
  }
}

}  // namespace dp

// ---------------------------------------------------------------------------------------------------------------------
// C ABI: orchestration
// ---------------------------------------------------------------------------------------------------------------------
extern "C" {

// node-side convs: two workgroups per node group (DpConvArgs.path_split) while one per group leaves half of the 256 CUs x 2 slots idle

static int dp_check_batch(const DpBatch* b) {
  if (!b || b->B <= 0) return DP_E_BADARG;
  if (b->max_nl > dp::MAXN_L || b->max_np > dp::MAXN_P || b->max_t > dp::MAXT) return DP_E_TOOLARGE;
  return DP_OK;
}

// library-internal (dp_sample): the time parameters of S rows of times [S,3] into table [S,32] - time_params_kernel with a schedule's rows as its "poses"
int dpg_time_table(const float* t, int32_t S, const DpWeights* w, float* table, void* stream) {
  if (!t || S <= 0 || !w || !table) return DP_E_BADARG;
  hipLaunchKernelGGL(dp::time_params_kernel, dim3((S + 7) / 8), dim3(256), 0, (hipStream_t)stream, S, const_cast<float*>(t), (const float*)nullptr, *w, table);
  return (int)hipGetLastError();
}

int dp_prepare(const DpBatch* b, const DpWeights* w, DpWork* k, void* stream) {
  int rc = dp_check_batch(b);
  if (rc) return rc;
  if (!w || !k) return DP_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int n = b->n_lig > b->n_ph ? b->n_lig : b->n_ph;
  hipLaunchKernelGGL(dp::static_nodes_kernel, dim3((n + 127) / 128), dim3(128), 0, st, *b, *w, *k);
  if (b->n_pp > 0) hipLaunchKernelGGL(dp::static_pp_kernel, dim3((b->n_pp + 127) / 128), dim3(128), 0, st, *b, *w, *k);
  return (int)hipGetLastError();
}

// Small batches: launches that cannot fill the chip (256 CUs x 2 workgroup slots).
//  * fp32 / bf16 node-side kernels (16-node groups): path split (fp32 only: two workgroups per group, half of the paths each) + conv split (one
//    workgroup per conv of the fused launch, dp_conv_node_combine adds them up) while B x groups <= 256;
//  * the two-limb fp16 kernel (8-node groups: twice the workgroups to begin with, ~1.5x shorter passes): conv split while the FUSED launch would
//    have at most DP_F16_CSPLIT_MAX_WG workgroups; with it the fp32 split kernels are no longer the default of an fp32 model at any batch size.
//    ms per step, 30-atom ligands / 32 phore nodes, best driver (profiles/r5_batch_sweep.csv; before = fp32 split kernels up to 24 poses, unsplit
//    f16x3 launches beyond): 1 pose 0.570 -> 0.468, 8: 0.622 -> 0.490, 20: 0.698 -> 0.527, 40: 0.697 -> 0.651, 64: 0.838 -> 0.793, 96: 1.036 -> 0.991,
//    128 (512 workgroups per fused launch): 1.11 either way.  The two sides of a layer run side by side: the gain is large while BOTH split
//    launches fit the chip at once (<= 25 poses of this shape) and small beyond.
// (f16_avail: only reached by builds with DP_F16_CSPLIT_MAX_WG <= 0 - the shipped library gives an fp32 model's small batches the f16x3 conv split
//  instead and never takes the fp32 path split by itself; the bf16 model calls this with f16_avail = false)
static int node_path_split_auto(int B, int slices, bool f16_avail) { return (long long)B * slices <= (f16_avail ? 48 : 256) ? 1 : 0; }
#ifndef DP_F16_CSPLIT_MAX_WG
#define DP_F16_CSPLIT_MAX_WG 384
#endif
#ifndef DP_SYM_BARRIER
#define DP_SYM_BARRIER 1
#endif
#ifndef DP_HEADS_EARLY
#define DP_HEADS_EARLY 512           // centre edges / torsion-bond graph on the second stream beside the last interaction layer while that has at most
                                     // this many 16-atom ligand node groups (B x slices_pl; 0: always in the tail)
#endif
#ifndef DP_PP_PSPLIT_MAX_CHUNKS
#define DP_PP_PSPLIT_MAX_CHUNKS 64   // phore-phore conv as a list of at most this many 128-edge chunks: phase split (DpConvArgs.phase_split of a list launch)
#endif
#ifndef DP_PP_BESIDE_MAX_WG
#define DP_PP_BESIDE_MAX_WG 640      // third stream for the phore-phore conv while the layer's launches together have at most this many workgroups
#endif
static int node_f16_conv_split_auto(int B, int slices16) { return (long long)B * slices16 * 2 <= DP_F16_CSPLIT_MAX_WG ? 1 : 0; }
// which formulation the convs of a forward take (shared by dp_forward and dp_forward_launch_of)
static bool plan_node_side(const DpBatch* b, const DpWeights* w, const DpWork* k) {
  return dp_resolve_conv_algo(k->conv_algo, w->precision, b->B) == DP_ALGO_NODE && w->p2l_conv[0].w2n != nullptr;
}
static bool plan_node_ll(const DpBatch* b, const DpWeights* w, const DpWork* k) {
  return plan_node_side(b, w, k) && w->lig_conv[0].w2n != nullptr && b->llp_ptr != nullptr && k->ll_pair_rad != nullptr &&
         k->ll_pair_bond != nullptr && b->max_nl <= (k->ll_node_max_atoms > 0 ? k->ll_node_max_atoms : DP_NODE_LL_MAX_ATOMS);
}

// dp_conv_node.hip (library-internal): the conv-split launch whose combine kernel first waits for an event of another stream
extern "C" int dpn_fused_split_after(const DpConvArgs* convs, int32_t n, int32_t n_nodes, float* part, int64_t part_floats, void* stream, void* before_combine);

// which node-side launches run the two-limb fp16 kernel, and whether the normals convs split into an s0-only launch + their listed pairs
struct F16Plan { bool f16_ok, f16_lig, f16_ph, xs_lig, xs_ph; };
static F16Plan plan_f16(const DpBatch* b, const DpWeights* w, const DpWork* k) {
  F16Plan p;
  const bool node_side = plan_node_side(b, w, k), node_ll = plan_node_ll(b, w, k);
  const int nb_ = dp_conv_node_max_nodes();
  // node-side convs of an fp32 model at fp32 grade on the 16-bit matrix pipe (DP_PREC_F16X3, two fp16 limbs per operand) when the blob carries
  // those images; DpWork.path_split = 1 (A/B runs) keeps the fp32 path-split kernels
  p.f16_ok = node_side && w->precision == DP_PREC_F32 && k->node_f16 != 0 && w->p2l_conv[0].w2h != nullptr && k->x_emb_bf != nullptr;
  auto split_of = [&](int slices_) {
    return k->path_split == 1 ? 1 : (k->path_split == 2 ? 0 : ((p.f16_ok && DP_F16_CSPLIT_MAX_WG > 0) ? 0 : node_path_split_auto(b->B, slices_, p.f16_ok)));
  };
  p.f16_lig = p.f16_ok && !split_of((b->max_nl + nb_ - 1) / nb_) && (!node_ll || (w->lig_conv[0].w2h != nullptr && k->ll_emb_bf != nullptr));
  p.f16_ph = p.f16_ok && !split_of((b->max_np + nb_ - 1) / nb_);
  // The two "normals" convs of a layer (phore_to_lig_norm / lig_to_phore_norm, :686-689,698-700).  Their spherical harmonics are e0 = (1, 0, ..., 0) on
  // every cross pair without type agreement (:874-894: the vector they are taken of is cross(...) * sum(agreement)), and the pairs WITH agreement are a
  // static, short list (DpBatch.xs_* / xt_*: 3 % of the pairs of the synthetic workloads, 5 feature nodes of 79 in the shipped example).  The conv is
  // bilinear in (features, harmonics), so it is evaluated as
  //     [paths through the l = 0 component, on every pair]   node-side kernel, s0-only instantiation: no per-node operand build, a third of the columns
  //   + [the other paths, on the listed pairs only]          edge-side kernel over the list (DP_EDGES_LIST, component 0 read as 0), folded into the
  //                                                          s0-only launch's epilogue
  // - the same sum term by term, minus the terms that multiply by an exact zero.  Taken when the f16x3 node-side kernels run, the batch carries the
  // lists and they are short (a long list costs more edge-side than the node-side conv it replaces); DpWork.xn_sparse = 2 keeps the dense convs.
#ifdef DP_S0_TIMING      // timing build (wrong results): the l = 0 part only
  p.xs_lig = p.f16_lig; p.xs_ph = p.f16_ph;
#else
  const bool xs_have = b->xs_src && b->xs_nbr && b->xs_feat && b->xs_rowptr && b->xt_src && b->xt_nbr && b->xt_feat && b->xt_rowptr && b->n_xs >= 0 &&
                       k->xs_msg != nullptr && w->p2l_norm_conv[0].w2x && w->p2l_norm_conv[0].bn_scale_x && w->l2p_norm_conv[0].w2x && w->l2p_norm_conv[0].bn_scale_x;
  // ... and the launches fill the chip: a batch small enough for the conv split (<= DP_F16_CSPLIT_MAX_WG workgroups per fused launch) is bound by the
  // LATENCY of its chain of launches, which the split makes two launches longer per side and layer (ms per step, 30 atoms / 32 nodes, graph + second
  // stream, dense vs split: 40 poses 0.65 / 0.77, 96: 0.99 / 1.00, 128: 1.17 / 1.13, 256: 1.87 / 1.70, 512: 3.50 / 3.09 - profiles/r6_batch_sweep.csv)
  const bool short_list = (long long)b->n_xs * 4 <= (long long)b->n_x;
  auto big = [&](int max_nodes) { return k->path_split != 0 || !node_f16_conv_split_auto(b->B, (max_nodes + nb_ - 1) / nb_); };
  p.xs_lig = p.f16_lig && xs_have && (k->xn_sparse == 1 || (k->xn_sparse == 0 && short_list && big(b->max_nl)));
  p.xs_ph = p.f16_ph && xs_have && (k->xn_sparse == 1 || (k->xn_sparse == 0 && short_list && big(b->max_np)));
  // a DP_PREC_BF16 model: the same split on its node-side kernels (their s0-only instantiation; the listed pairs run the two-limb fp16 list kernel,
  // 3 % of the pairs at more than the model's precision), under that kernel's small-batch rule (conv split while B x groups <= 256)
  if (node_side && w->precision == DP_PREC_BF16 && k->x_emb_bf != nullptr && xs_have) {
    auto big_bf = [&](int max_nodes) { return k->path_split == 2 || (k->path_split == 0 && !node_path_split_auto(b->B, (max_nodes + nb_ - 1) / nb_, false)); };
    p.xs_lig = k->xn_sparse == 1 || (k->xn_sparse == 0 && short_list && big_bf(b->max_nl));
    p.xs_ph = k->xn_sparse == 1 || (k->xn_sparse == 0 && short_list && big_bf(b->max_np));
  }
#endif
  return p;
}

int dp_forward_launch_of(const DpBatch* b, const DpWeights* w, const DpWork* k, int32_t layer, int32_t conv, int32_t* first, int32_t* count,
                         int32_t* node_side_out) {
  if (!b || !w || !k || !first || !count || layer < 0 || layer > 3 || conv < 0 || conv > 5 || (layer == 3 && conv > 2)) return DP_E_BADARG;
  const bool node_side = plan_node_side(b, w, k), node_ll = plan_node_ll(b, w, k), fuse = k->conv_fuse != 1;
  const F16Plan pl = plan_f16(b, w, k);
  int f = conv, n = 1;
  bool ns = false, s0 = false;
  if (conv <= 2) {                                  // ligand side: lig_conv, phore_to_lig, phore_to_lig_norm
    if (pl.xs_lig && conv == 2) { s0 = true; }      // the normals conv: its own s0-only launch (+ the listed pairs)
    else if (node_side && fuse && node_ll && k->conv_fuse != 2) { f = 0; n = pl.xs_lig ? 2 : 3; }
    else if (node_side && fuse && conv >= 1 && !pl.xs_lig) { f = 1; n = 2; }
    ns = conv == 0 ? node_ll : node_side;
  } else if (conv >= 4) {                           // pharmacophore side: lig_to_phore, lig_to_phore_norm (conv 3 = phore_conv, edge-side)
    if (pl.xs_ph) s0 = conv == 5;
    else if (node_side && fuse) { f = 4; n = 2; }
    ns = node_side;
  }
  *first = f; *count = n;
  if (node_side_out) *node_side_out = ns ? (s0 ? 2 : 1) : 0;
  return DP_OK;
}

int dp_forward(const DpBatch* b, const DpWeights* w, DpWork* k, float* tr, float* rot, float* tor, void* stream) {
  int rc = dp_check_batch(b);
  if (rc) return rc;
  if (!w || !k || !tr || !rot || (b->n_tor > 0 && !tor)) return DP_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int B = b->B;
  // dense cross convs: node-side formulation when asked for (or chosen by AUTO) and the weights were packed for it
  const bool node_side = plan_node_side(b, w, k);
  // lig-lig conv on the node-side kernel too (pair-dense form): pays while a pose's atoms fit a few 32-slot passes
  const bool node_ll = plan_node_ll(b, w, k);
  // optional second stream: phore-side convs of a layer (and the torsion head) run beside the ligand-side ones
  hipStream_t aux = (k->aux_stream && k->ev_fork && k->ev_join) ? (hipStream_t)k->aux_stream : nullptr;
  int ev_rc = 0;                                   // first failing event / stream call (reported at the end: launches are asynchronous anyway)
  auto note = [&](hipError_t e) { if (e != hipSuccess && ev_rc == 0) ev_rc = (int)e; };
  bool aux_forked = false;
  auto fork = [&]() {
    if (aux) { note(hipEventRecord((hipEvent_t)k->ev_fork, st)); note(hipStreamWaitEvent(aux, (hipEvent_t)k->ev_fork, 0)); aux_forked = true; }
  };
  auto join = [&]() {
    if (aux) { note(hipEventRecord((hipEvent_t)k->ev_join, aux)); note(hipStreamWaitEvent(st, (hipEvent_t)k->ev_join, 0)); }
    aux_forked = false;
  };
  // error exit: never leave the aux stream forked (eager: later work on `st` must stay ordered after it; capture: an unjoined stream
  // turns the caller's hipStreamEndCapture into an error that masks `rc`)
  // Between two stretches that both use the two streams (head -> layer 0, layer l -> l + 1) each stream needs what the OTHER one produced.  join() + fork()
  // would chain the two waits (the second stream resumes one cross-queue latency - ~19 us eager - after the launch stream, which itself resumed one
  // latency after the later of the two finished); recorded first and waited for crosswise, both resume one latency after the later one.  In a
  // captured graph the dependencies are the same either way.
  bool aux_synced = false, want_x2 = false;
  bool capturing = false;
  {
    hipStreamCaptureStatus cs_ = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs_) == hipSuccess) capturing = cs_ != hipStreamCaptureStatusNone;
    else (void)hipGetLastError();
  }
  auto barrier2 = [&]() {
    // (while the launch stream is being captured: the chained form.  The dependencies are the same, yet the replay of the graph captured from the
    //  crosswise form is 6-9 % slower at 1-8 poses - 7.92 vs 7.25 ms per 20 steps at one pose - and equal from 40 poses on; eager it is the other way
    //  round: 8.15 vs 9.07 ms at one pose, 11.4 vs 12.3 at 40, equal at 512.  DP_SYM_BARRIER = 0: A/B builds)
    if (capturing || DP_SYM_BARRIER == 0) {
      join();
      return;
    }
    note(hipEventRecord((hipEvent_t)k->ev_join, aux));
    if (want_x2) note(hipEventRecord((hipEvent_t)k->ev_x2, aux));      // (the same position once more, for a third stream: see pp_beside)
    note(hipEventRecord((hipEvent_t)k->ev_fork, st));
    note(hipStreamWaitEvent(st, (hipEvent_t)k->ev_join, 0));
    note(hipStreamWaitEvent(aux, (hipEvent_t)k->ev_fork, 0));
    aux_forked = false; aux_synced = true;
  };
  auto fail = [&](int rc_) { if (aux_forked) join(); return rc_; };
  if (k->t_row && k->tp_row)
    hipLaunchKernelGGL(dp::tp_bcast_kernel, dim3((B * dp::TP_STRIDE + 255) / 256), dim3(256), 0, st, B, const_cast<float*>(b->t), k->t_row, k->tp_row, k->tparams);
  else
    hipLaunchKernelGGL(dp::time_params_kernel, dim3((B + 7) / 8), dim3(256), 0, st, B, const_cast<float*>(b->t), k->t_row, *w, k->tparams);
  // the three graph / embedding kernels only share their inputs (positions, time parameters): the ligand radius graph runs on the second
  // stream beside the node embeddings and the cross edges (each is one workgroup per pose and latency-bound on its own)
  // (starting the radius graph - positions only - before the time parameters: nothing eager, 10-15 % SLOWER as a captured graph, whose first node of
  //  a step then fans out and whose lig_edges node gets a second predecessor: 1 pose 9.03 vs 7.88 ms per 20 steps, 40 poses 12.0 vs 10.7)
  if (aux) fork();
  const F16Plan pl = plan_f16(b, w, k);
  const bool f16_ok = pl.f16_ok, f16_lig = pl.f16_lig, f16_ph = pl.f16_ph;
  // the 16-bit images of the edge embeddings are written only for the convs that read them (DP_PREC_BF16 or DP_PREC_F16X3, node-side)
  const bool bf_node = node_side && w->precision == DP_PREC_BF16;
  if (bf_node && (!k->x_emb_bf || (node_ll && !k->ll_emb_bf))) return fail(DP_E_BADARG);
  DpWork kx = *k;
  kx.node_f16 = (f16_lig || f16_ph) ? 1 : 0;
  if (!bf_node && !kx.node_f16) kx.x_emb_bf = kx.ll_emb_bf = nullptr;
  if (!node_ll || !(bf_node || f16_lig)) kx.ll_emb_bf = nullptr;
  {
    // the N x N pair tables are filled (and verified: status bit 1) only when the node-side ligand-ligand conv will read them
    DpWork kg = kx;
    if (!node_ll) kg.ll_pair_rad = kg.ll_pair_bond = nullptr;
    hipLaunchKernelGGL(dp::lig_graph_kernel, dim3(B), dim3(256), 0, aux ? aux : st, *b, kg);
    // edge features: one edge per thread, workgroup slots for the largest ligand at its full radius degree (+ ~3 directed bonds per atom)
    const int deg = b->max_nl - 1 < 33 ? (b->max_nl > 1 ? b->max_nl - 1 : 1) : 33;
    const int e_slices = (b->max_nl * (deg + 3) + 255) / 256;
    hipLaunchKernelGGL(dp::lig_edges_kernel, dim3(B * e_slices), dim3(256), 0, aux ? aux : st, *b, *w, kg, e_slices);
  }
  hipLaunchKernelGGL(dp::node_embed_kernel, dim3(B), dim3(256), 0, st, *b, *w, *k);
  {
    // workgroup slots per pose of the cross-edge kernel: whole atoms, at most 256 edges each (see the kernel)
    const int apw = b->max_np > 0 && 256 / b->max_np > 0 ? 256 / b->max_np : 1;
    const int x_slices = b->max_nl > 0 ? (b->max_nl + apw - 1) / apw : 1;
    hipLaunchKernelGGL(dp::cross_edges_kernel, dim3(B * x_slices), dim3(256), 0, st, *b, *w, kx, x_slices);
  }
  // workgroups per pose: keep <= 48 scatter nodes per workgroup and >= ~2 workgroups per CU in flight
  const int prec_edge = w->precision;        // (DP_PREC_BF16: the convs that stay edge-side run the one-product form of the split-bf16 kernel)
  int slices = 1;
  {
    const int mx = b->max_nl > b->max_np ? b->max_nl : b->max_np;
    while ((mx + slices - 1) / slices > dp_conv_max_nodes(prec_edge)) ++slices;
    while (B * slices < 512 && slices < 8) ++slices;
  }
  const bool fuse = k->conv_fuse != 1;        // node-side convs that scatter onto the same nodes share one launch
  // the normals convs as s0-only launch + listed pairs (plan_f16)
  const bool xs_lig = pl.xs_lig, xs_ph = pl.xs_ph;
#ifdef DP_S0_TIMING
  const bool xs_list = false;
#else
  const bool xs_list = true;
#endif
  // the listed pairs of one normals conv: their messages through the edge-side kernel BEFORE the conv's s0-only launch, whose epilogue folds them in
  // (scatter mean over the full degree, BatchNorm scale: DpConvArgs.list_*)
  auto xs_bind = [&](int l_, bool lig_side, DpConvArgs& s0conv) {
    if (!xs_list || b->n_xs <= 0) return;
    const DpConvW& cw = lig_side ? w->p2l_norm_conv[l_] : w->l2p_norm_conv[l_];
    s0conv.list_msg = k->xs_msg + (lig_side ? (size_t)0 : (size_t)b->n_xs * 100);
    s0conv.list_rowptr = lig_side ? b->xs_rowptr : b->xt_rowptr;
    s0conv.list_scale = cw.bn_scale_x;
  };
  auto xs_launch = [&](int l_, bool lig_side, const float* hl_, const float* hp_, hipStream_t s_) -> int {
    if (!xs_list || b->n_xs <= 0) return DP_OK;
    const DpConvW& cw = lig_side ? w->p2l_norm_conv[l_] : w->l2p_norm_conv[l_];
    float* msg = k->xs_msg + (lig_side ? (size_t)0 : (size_t)b->n_xs * 100);
    DpConvArgs c;
    memset(&c, 0, sizeof(c));
    c.kind = l_; c.mode = DP_EDGES_LIST; c.B = B; c.slices = 1; c.precision = DP_PREC_F16X3; c.max_scatter = 1;
    c.n_list = b->n_xs; c.sh0_zero = 1; c.msg_out = msg;
    c.edge_scat = lig_side ? b->xs_src : b->xt_src; c.nbr = lig_side ? b->xs_nbr : b->xt_nbr; c.feat_idx = lig_side ? b->xs_feat : b->xt_feat;
    // attr1 = the ligand atom, attr2 = the pharmacophore node in both directions (:682-700): scatter / gather for phore->ligand, swapped for ligand->phore
    c.attr_swap = lig_side ? 0 : 1;
    c.gather_tab = lig_side ? hp_ : hl_; c.attr1_tab = hl_; c.attr2_tab = hp_;
    c.gather_stride = c.attr1_stride = c.attr2_stride = 100;
    c.edge_emb = k->x_emb; c.edge_sh = k->x_nsh; c.sh_stride = 9;
    c.w1p = cw.w1p; c.w2p = cw.w2x;
    return dp_tp_conv(&c, s_);
  };
  // Where the list launches of a layer run (DpWork.xs_place).  They are small grids (one chunk's latency each, ~120 workgroups): in front of their s0-only
  // launch they sit on that stream's chain; CROSSED, each opens the OTHER side's stream at the start of the layer (the ligand side's list launch on the
  // pharmacophore side's stream and vice versa; layer 3, which has no pharmacophore side, uses the idle second stream) - its consumer waits for an event
  // that has long fired.  2 = both on the third stream (measured slower: one more concurrent launch on a full chip).
  const bool xs_ev = k->ev_join2 && k->ev_x2 && xs_list && b->n_xs > 0 && aux != nullptr;
  const int xs_place = (!xs_ev || k->xs_place == 0 || k->xs_place == 4) ? 3 : (k->xs_place == 2 && !k->aux2_stream ? 3 : k->xs_place);
  const bool xs_async = xs_place == 2;
  bool xs_early[2] = {false, false};
  auto node_path_split = [&](int B_, int slices_) {
    return k->path_split == 1 ? 1 : (k->path_split == 2 ? 0 : ((f16_ok && DP_F16_CSPLIT_MAX_WG > 0) ? 0 : node_path_split_auto(B_, slices_, f16_ok)));
  };
  // f16x3 launches: DpConvArgs.path_split only asks for the conv split (the kernel has no path split)
  auto f16_csplit = [&](int slices_) { return k->path_split == 0 ? node_f16_conv_split_auto(B, slices_) : 0; };
  // the convs that stay edge-side in an fp32 model (phore-phore, torsion bonds) take the two-limb fp16 form of their radial contraction when the
  // blob carries it (DpConvW.w2x / bn_scale_x; until round 4: the 3-way bf16 split): same tolerances, 16x the matrix rate per product
  const bool edge_x6 = w->precision == DP_PREC_F32 && k->edge_bf16x6 != 0;
  int slices_x6 = 1;
  {
    const int mx = b->max_nl > b->max_np ? b->max_nl : b->max_np;
    while ((mx + slices_x6 - 1) / slices_x6 > dp_conv_max_nodes(DP_PREC_BF16X6)) ++slices_x6;
    while (B * slices_x6 < 512 && slices_x6 < 8) ++slices_x6;
  }
  const int nb = dp_conv_node_max_nodes();
  const int slices_pl = (b->max_nl + nb - 1) / nb, slices_lp = (b->max_np + nb - 1) / nb;
  int cur = 0;
  static const int D[5] = {20, 50, 80, 100, 100};
  auto ev_all = [&](int layer, int which, int side) -> hipEvent_t {      // DpWork.prof_events[layer][conv][start / stop]
    return k->prof_events ? (hipEvent_t)k->prof_events[(layer * 6 + which) * 2 + side] : nullptr;
  };
  auto timed = [&](const DpConvArgs& c, int layer, int which, hipStream_t s_) -> int {
    const bool prof = k->prof_start && k->prof_stop && k->prof_layer == layer && k->prof_conv == which;
    if (prof) note(hipEventRecord((hipEvent_t)k->prof_start, s_));
    if (ev_all(layer, which, 0)) note(hipEventRecord(ev_all(layer, which, 0), s_));
    const int r = dp_tp_conv(&c, s_);
    if (ev_all(layer, which, 1)) note(hipEventRecord(ev_all(layer, which, 1), s_));
    if (prof) note(hipEventRecord((hipEvent_t)k->prof_stop, s_));
    return r;
  };
  // a normals conv as two launches - its listed pairs (messages), then its s0-only launch, which folds them in - inside ONE pair of timing events
  auto timed_xs = [&](DpConvArgs& c, int layer, int which, bool lig_side, const float* hl_, const float* hp_, hipStream_t s_) -> int {
    const bool prof = k->prof_start && k->prof_stop && k->prof_layer == layer && k->prof_conv == which;
    if (prof) note(hipEventRecord((hipEvent_t)k->prof_start, s_));
    if (ev_all(layer, which, 0)) note(hipEventRecord(ev_all(layer, which, 0), s_));
    int r = DP_OK;
    // the messages reach the output through a dp_list_reduce launch behind the s0-only launch (library's choice: 171.1 k vs 169.9 k pose-steps/s folded into
    // that launch's epilogue, DpConvArgs.list_*, xs_place = 3: the fold's dependent loads stand at the end of a short kernel)
    const bool own_reduce = k->xs_place != 3 && xs_list && b->n_xs > 0;
    if (!own_reduce) xs_bind(layer, lig_side, c);
    if (xs_early[lig_side ? 0 : 1]) note(hipStreamWaitEvent(s_, (hipEvent_t)(lig_side ? k->ev_join2 : k->ev_x2), 0));      // its list launch ran elsewhere
    else r = xs_launch(layer, lig_side, hl_, hp_, s_);
    if (!r) r = dp_tp_conv(&c, s_);
    if (!r && own_reduce) {
      static const int DOUT_[4] = {50, 80, 100, 100};
      const DpConvW& cw = lig_side ? w->p2l_norm_conv[layer] : w->l2p_norm_conv[layer];
      r = dp_list_reduce(k->xs_msg + (lig_side ? (size_t)0 : (size_t)b->n_xs * 100), lig_side ? b->xs_rowptr : b->xt_rowptr, B, lig_side ? b->lig_ptr : b->ph_ptr,
                         lig_side ? b->max_nl : b->max_np, lig_side ? b->ph_ptr : b->lig_ptr, DOUT_[layer], cw.bn_scale_x, c.out, 100, s_);
    }
    if (ev_all(layer, which, 1)) note(hipEventRecord(ev_all(layer, which, 1), s_));
    if (prof) note(hipEventRecord((hipEvent_t)k->prof_stop, s_));
    return r;
  };
  // (while the layer-3 launch leaves workgroup slots free: beside a launch that fills the chip they take its slots.  ms per step with / without, 30 atoms /
  //  32 nodes, graph + second stream: 64 poses 0.666 / 0.673, 128: 1.034 / 1.062, 192: 1.359 / 1.371, 256: 1.621 / 1.622, 512: 3.015 / 2.954)
  // third stream for the phore-phore conv (see the pharmacophore side of the layer loop): the batch's answer, the same for layers 0-2
  bool pp_beside_ok = false;
  {
    const bool pp_list_ = edge_x6 && w->ph_conv[0].w2x && w->ph_conv[0].bn_scale_x && k->pp_msg != nullptr && dp_pp_as_list(b);
    const long long wg_pp = pp_list_ ? (b->n_pp + 127) / 128 : (long long)B * slices_x6;
    const long long wg_layer = wg_pp + (long long)B * ((b->max_np + nb - 1) / nb) * 2 * 2 + (long long)B * ((b->max_nl + nb - 1) / nb) * 2 * 3;
    pp_beside_ok = aux != nullptr && k->aux2_stream && k->ev_join2 && k->pp_tmp && node_side && fuse && !xs_ph && !xs_lig && f16_ph && f16_csplit(slices_lp) &&
                   k->conv_part != nullptr && k->conv_fuse != 3 && (int64_t)2 * b->n_ph * 100 <= DP_CONV_PART_FLOATS / 2 && wg_layer <= DP_PP_BESIDE_MAX_WG;
  }
  const bool heads_early = aux != nullptr && k->aux_mode != 1 && (long long)B * slices_pl <= DP_HEADS_EARLY;
  // end of the head (launched above; everything in between is host arithmetic): both streams need all of it
  want_x2 = pp_beside_ok && !capturing && k->ev_x2 != nullptr;
  if (aux) { if (k->aux_mode != 1) barrier2(); else join(); }      // (layer 0 has two sides unless aux_mode = 1)
  for (int l = 0; l < 4; ++l) {
    const float* hl = k->h_l[cur];
    const float* hp = k->h_p[cur];
    float* hl2 = k->h_l[cur ^ 1];
    float* hp2 = k->h_p[cur ^ 1];
    const int din = D[l];
    DpConvArgs a;
    // ligand side: lig_conv (+ residual stream), phore_to_lig, phore_to_lig_norm      (:678-689, :703-706)
    memset(&a, 0, sizeof(a));
    a.kind = l; a.B = B; a.slices = slices; a.precision = prec_edge; a.lig_ptr = b->lig_ptr; a.ph_ptr = b->ph_ptr; a.x_ptr = b->x_ptr;
    a.phase_split = (B * slices < 512) ? 1 : 0;      // under-filled chip: one workgroup per accumulation phase (see DpConvArgs)
    a.sh_stride = 9; a.out_stride = 100; a.resid_stride = 100;
    a.gather_stride = a.attr1_stride = a.attr2_stride = 100;
    const bool two = aux != nullptr && l != 3 && k->aux_mode != 1;
    hipStream_t sp_ = two ? aux : st;           // stream of the phore-side convs
    const bool layer_synced = two && aux_synced;
    if (two) {
      if (aux_synced) aux_forked = true;          // (the barrier that ended the previous stretch has made the second stream wait already)
      else fork();
      aux_synced = false;
    }
    // the last layer has no pharmacophore side: the second stream builds the graphs of the two score heads meanwhile - centre edges and torsion-bond
    // neighbourhoods are functions of this step's positions and times only (35 us off the tail of a step, at any batch size)
    if (heads_early && l == 3) {
      fork();
      hipLaunchKernelGGL(dp::center_edges_kernel, dim3(B), dim3(256), 0, aux, *b, *w, *k);
      if (b->n_tor > 0) hipLaunchKernelGGL(dp::tor_graph_kernel, dim3(B), dim3(256), 0, aux, *b, *w, *k);
    }
    xs_early[0] = xs_early[1] = false;
    if ((xs_async || xs_place == 1) && (xs_lig || (xs_ph && l != 3))) {
      // the layer's inputs are complete on st here (ev_fork: recorded by fork() when the layer has two sides)
      // (a third stream: a fresh record - after barrier2 the launch stream has everything only behind its own wait for the second stream)
      if (!two || xs_async) note(hipEventRecord((hipEvent_t)k->ev_fork, st));
      hipStream_t s_lig = xs_async ? (hipStream_t)k->aux2_stream : aux;       // crossed: the ligand side's list on the second stream ...
      hipStream_t s_ph = xs_async ? (hipStream_t)k->aux2_stream : st;         // ... the pharmacophore side's on the launch stream
      if (xs_async || !two) note(hipStreamWaitEvent(s_lig, (hipEvent_t)k->ev_fork, 0));      // (with two sides the second stream has waited in fork())
      if (xs_lig) {
        if ((rc = xs_launch(l, true, hl, hp, s_lig))) return fail(rc);
        note(hipEventRecord((hipEvent_t)k->ev_join2, s_lig));
        xs_early[0] = true;
      }
      if (xs_ph && l != 3) {
        if ((rc = xs_launch(l, false, hl, hp, s_ph))) return fail(rc);
        note(hipEventRecord((hipEvent_t)k->ev_x2, s_ph));
        xs_early[1] = true;
      }
    }
    // timing hook around the launch that contains conv `which` (a fused launch answers for each of its convs)
    auto timed_fused = [&](const DpConvArgs* cs, int n, int which_lo, int which_hi, hipStream_t s_, void* before_combine = nullptr) -> int {
      const bool prof = k->prof_start && k->prof_stop && k->prof_layer == l && k->prof_conv >= which_lo && k->prof_conv <= which_hi;
      if (prof) note(hipEventRecord((hipEvent_t)k->prof_start, s_));
      if (ev_all(l, which_lo, 0)) note(hipEventRecord(ev_all(l, which_lo, 0), s_));
      // small batches (path split on): one conv per workgroup instead of all convs back to back in one (conv split); the two sides of a
      // layer may run on two streams, so each has its own half of the scratch
      const bool lig_side = which_lo < 3;
      const int n_nodes = lig_side ? b->n_lig : b->n_ph;
      const int64_t half = DP_CONV_PART_FLOATS / 2;
      const bool csplit = n >= 2 && cs[0].path_split && k->conv_part != nullptr && k->conv_fuse != 3 && (int64_t)n * n_nodes * cs[0].out_stride <= half;
      if (before_combine && !csplit) note(hipStreamWaitEvent(s_, (hipEvent_t)before_combine, 0));      // (no combine launch: wait in front of the conv)
      const int r = csplit ? (before_combine ? dpn_fused_split_after(cs, n, n_nodes, k->conv_part + (lig_side ? 0 : half), half, s_, before_combine)
                                             : dp_tp_conv_node_fused_split(cs, n, n_nodes, k->conv_part + (lig_side ? 0 : half), half, s_))
                           : dp_tp_conv_node_fused(cs, n, s_);
      if (ev_all(l, which_lo, 1)) note(hipEventRecord(ev_all(l, which_lo, 1), s_));
      if (prof) note(hipEventRecord((hipEvent_t)k->prof_stop, s_));
      return r;
    };
    DpConvArgs ligc[3];                          // ligand side: lig_conv, phore_to_lig, phore_to_lig_norm
    {
      DpConvArgs& c = ligc[0];
      c = a;
      c.mode = DP_EDGES_CSR; c.scat_ptr = b->lig_ptr; c.max_scatter = b->max_nl; c.rowptr = k->ll_rowptr; c.rowend = k->ll_rowend; c.nbr = k->ll_nbr;
      c.edge_scat = k->ll_src; c.gather_tab = hl; c.attr1_tab = hl; c.attr2_tab = hl; c.edge_emb = k->ll_emb; c.edge_sh = k->ll_sh;
      c.w1p = w->lig_conv[l].w1p; c.w2p = w->lig_conv[l].w2p; c.bn_scale = w->lig_conv[l].bn_scale; c.bn_shift = w->lig_conv[l].bn_shift;
      c.out = hl2; c.accumulate = 0; c.resid = hl; c.resid_width = din;
      if (!node_ll && edge_x6 && w->lig_conv[l].w2x && w->lig_conv[l].bn_scale_x) { c.precision = DP_PREC_F16X3; c.w2p = w->lig_conv[l].w2x; c.bn_scale = w->lig_conv[l].bn_scale_x; c.slices = slices_x6; c.phase_split = 0; }
      if (node_ll) {
        c.algo = DP_ALGO_NODE; c.mode = DP_EDGES_LLPAIR; c.w1n = w->lig_conv[l].w1n; c.w2n = w->lig_conv[l].w2n; c.slices = slices_pl; c.phase_split = 0;
        c.precision = w->precision; c.edge_emb_bf = k->ll_emb_bf; c.emb_flag_row = b->ll_cap;
        c.no_tail = b->no_tail_nl;
        c.path_split = node_path_split(B, slices_pl);
        c.pair_a = k->ll_pair_rad; c.pair_b = k->ll_pair_bond; c.pair_ptr = b->llp_ptr;
        if (f16_lig) { c.precision = DP_PREC_F16X3; c.w1n = w->lig_conv[l].w1h; c.w2n = w->lig_conv[l].w2h; c.path_split = f16_csplit(slices_pl); }
      }
    }
    for (int nv = 0; nv < 2; ++nv) {
      DpConvArgs& c = ligc[1 + nv];
      c = a;
      const DpConvW& cw = nv ? w->p2l_norm_conv[l] : w->p2l_conv[l];
      c.mode = DP_EDGES_P2L; c.scat_ptr = b->lig_ptr; c.max_scatter = b->max_nl; c.gather_tab = hp; c.attr1_tab = hl; c.attr2_tab = hp;
      c.edge_emb = k->x_emb; c.edge_sh = nv ? k->x_nsh : k->x_sh;
      c.w1p = cw.w1p; c.w2p = cw.w2p; c.bn_scale = cw.bn_scale; c.bn_shift = cw.bn_shift;
      c.out = hl2; c.accumulate = 1;
      if (node_side) { c.algo = DP_ALGO_NODE; c.w1n = cw.w1n; c.w2n = cw.w2n; c.slices = slices_pl; c.phase_split = 0; c.path_split = node_path_split(B, slices_pl); c.no_tail = b->no_tail_np;
                       c.precision = w->precision; c.edge_emb_bf = k->x_emb_bf; c.emb_flag_row = b->n_x;
                       if (f16_lig) { c.precision = DP_PREC_F16X3; c.w1n = cw.w1h; c.w2n = cw.w2h; c.path_split = f16_csplit(slices_pl); }
      }
    }
    if (xs_lig) ligc[2].s0_only = 1;            // the normals conv as its l = 0 part (own launch) + the listed pairs (below)
    if (node_side && fuse && node_ll && k->conv_fuse != 2) {
      // one launch: the convs scatter onto the same atoms and their outputs add up (:703-706)
      ligc[1].accumulate = 0;
      if (!xs_lig) {
        ligc[2].accumulate = 0;
        if ((rc = timed_fused(ligc, 3, 0, 2, st))) return fail(rc);
      } else {
        if ((rc = timed_fused(ligc, 2, 0, 1, st))) return fail(rc);
        // (a batch that fills the chip - heads_early off - with the head graphs on the second stream HERE, behind the fused launch and beside the normals
        //  conv's short list launch: no difference, 173.8 vs 173.7 k pose-steps/s at 512 poses; not kept)
        if ((rc = timed_xs(ligc[2], l, 2, true, hl, hp, st))) return fail(rc);
      }
    } else {
      if ((rc = timed(ligc[0], l, 0, st))) return fail(rc);
      if (node_side && fuse && !xs_lig) {
        if ((rc = timed_fused(ligc + 1, 2, 1, 2, st))) return fail(rc);
      } else {
        for (int nv = 0; nv < 2; ++nv)
          if ((rc = (xs_lig && nv) ? timed_xs(ligc[2], l, 2, true, hl, hp, st) : timed(ligc[1 + nv], l, 1 + nv, st))) return fail(rc);
      }
    }
    if (l != 3) {   // phore side (:691-700, :708-710) - reads the PRE-update ligand features
      // third stream (DpWork.aux2_stream): when the ligand->phore launch takes the conv split (a batch that cannot fill the chip: the step is a chain
      // of short dependent launches), the phore-phore conv runs BESIDE it instead of in front of it: it writes pad(h_p) + its output to pp_tmp, which
      // the split launch's combine kernel (behind ev_join2) adds to the other convs' parts
      // ... while the three launches of a layer together leave workgroup slots free (256 CUs x 2): ms per step with / without, 30 atoms / 32 nodes,
      // graph + streams: 1 pose 0.439 / 0.473, 8: 0.476 / 0.499, 20: 0.570 / 0.535, 40: 0.678 / 0.652 - from ~18 poses on the chip is full of (long)
      // workgroups and one more concurrent launch only stretches the others
      const bool pp_beside = two && pp_beside_ok;
      hipStream_t s3 = pp_beside ? (hipStream_t)k->aux2_stream : sp_;
      if (pp_beside) {
        // the layer's inputs are what BOTH streams had produced when it began.  After fork() all of it is behind ev_fork; after barrier2 behind one
        // event per stream (the second stream's position under a name of its own, ev_x2).  While the launch stream is being CAPTURED the third stream
        // waits for a fresh record on the launch stream instead: joining a capture through the events of two different streams crashes inside the
        // HIP runtime of ROCm 7.2 (hipStreamEndCapture), and a graph has no launch-stream latency to save anyway - the dependencies are the same.
        if (layer_synced && want_x2) {      // barrier2: one event per stream, waited for side by side
          note(hipStreamWaitEvent(s3, (hipEvent_t)k->ev_fork, 0));
          note(hipStreamWaitEvent(s3, (hipEvent_t)k->ev_x2, 0));
        } else {
          if (layer_synced) note(hipEventRecord((hipEvent_t)k->ev_fork, st));
          note(hipStreamWaitEvent(s3, (hipEvent_t)k->ev_fork, 0));
        }
      }
      {
        DpConvArgs c = a;
        c.mode = DP_EDGES_CSR; c.scat_ptr = b->ph_ptr; c.max_scatter = b->max_np; c.rowptr = b->pp_rowptr; c.rowend = b->pp_rowend; c.nbr = b->pp_dst;
        c.edge_scat = b->pp_src; c.gather_tab = hp; c.attr1_tab = hp; c.attr2_tab = hp; c.edge_emb = k->pp_emb; c.edge_sh = k->pp_sh;
        c.w1p = w->ph_conv[l].w1p; c.w2p = w->ph_conv[l].w2p; c.bn_scale = w->ph_conv[l].bn_scale; c.bn_shift = w->ph_conv[l].bn_shift;
        c.out = hp2; c.accumulate = 0; c.resid = hp; c.resid_width = din;
        const bool x6 = edge_x6 && w->ph_conv[l].w2x && w->ph_conv[l].bn_scale_x;
        if (x6) { c.precision = DP_PREC_F16X3; c.w2p = w->ph_conv[l].w2x; c.bn_scale = w->ph_conv[l].bn_scale_x; c.slices = slices_x6; c.phase_split = 0; }
        if (pp_beside) { c.out = k->pp_tmp; c.out_stride = 100; }
        // one pharmacophore for the whole batch at one diffusion time (DpBatch.ph_shared, DpWork.t_uniform): the layer-0 conv reads h_p^0 (type / feature
        // embeddings + the time embedding), e_pp and sh_pp only (:691-693 at l = 0) - the same numbers for every pose.  Pose 0's edges go through the list
        // kernel, dp_list_finish hands node j's mean to node j of every pose (the residual pad(h_p) is read per pose: equal too, no special case).
        const bool ph_once = l == 0 && b->ph_shared != 0 && (k->t_uniform != 0 || k->t_row != nullptr) && k->ph_share != 2 && B > 1 && x6 && k->pp_msg != nullptr &&
                             b->n_pp % B == 0 && b->n_ph % B == 0;
        if (x6 && k->pp_msg != nullptr && (ph_once || dp_pp_as_list(b))) {
          // small pharmacophores: the batch's phore-phore edges as ONE list in full 128-edge chunks (dp_pp_as_list), then mean / BatchNorm / residual
          DpConvArgs cl = c;
          cl.mode = DP_EDGES_LIST; cl.n_list = ph_once ? b->n_pp / B : b->n_pp; cl.sh0_zero = 0; cl.msg_out = k->pp_msg; cl.feat_idx = nullptr;
          // a list that cannot fill the chip: two workgroups per chunk, one accumulation phase each (the launch heads the pharmacophore side's chain)
          cl.phase_split = (cl.n_list + 127) / 128 <= DP_PP_PSPLIT_MAX_CHUNKS ? 1 : 0;
          const bool prof = k->prof_start && k->prof_stop && k->prof_layer == l && k->prof_conv == 3;
          if (prof) note(hipEventRecord((hipEvent_t)k->prof_start, s3));
          if (ev_all(l, 3, 0)) note(hipEventRecord(ev_all(l, 3, 0), s3));
          rc = dp_tp_conv(&cl, s3);
          // (its mean / BatchNorm affine as a launch of its own: folded into the epilogue of the layer's s0-only lig_to_phore_norm launch - one launch
          //  and one pass over hp2 less - the step was 3 % SLOWER, 164.4 k vs 169.3 k: the dependent loads stand at the end of a short kernel)
          if (!rc) rc = dp_list_finish(k->pp_msg, b->pp_rowptr, b->pp_rowend, b->n_ph, D[l + 1], c.bn_scale, c.bn_shift, c.resid, c.resid_stride, c.resid_width,
                                       c.out, c.out_stride, 0, ph_once ? b->n_ph / B : 0, s3);
          if (ev_all(l, 3, 1)) note(hipEventRecord(ev_all(l, 3, 1), s3));
          if (prof) note(hipEventRecord((hipEvent_t)k->prof_stop, s3));
          if (rc) return fail(rc);
        } else if ((rc = timed(c, l, 3, s3))) return fail(rc);
        if (pp_beside) note(hipEventRecord((hipEvent_t)k->ev_join2, s3));
      }
      DpConvArgs phc[2];
      for (int nv = 0; nv < 2; ++nv) {
        DpConvArgs& c = phc[nv];
        c = a;
        const DpConvW& cw = nv ? w->l2p_norm_conv[l] : w->l2p_conv[l];
        c.mode = DP_EDGES_L2P; c.scat_ptr = b->ph_ptr; c.max_scatter = b->max_np; c.gather_tab = hl; c.attr1_tab = hl; c.attr2_tab = hp;
        c.edge_emb = k->x_emb; c.edge_sh = nv ? k->x_nsh : k->x_sh;
        c.w1p = cw.w1p; c.w2p = cw.w2p; c.bn_scale = cw.bn_scale; c.bn_shift = cw.bn_shift;
        c.out = hp2; c.accumulate = 1;
        if (node_side) { c.algo = DP_ALGO_NODE; c.w1n = cw.w1n; c.w2n = cw.w2n; c.slices = slices_lp; c.phase_split = 0; c.path_split = node_path_split(B, slices_lp); c.no_tail = b->no_tail_nl;
                         c.precision = w->precision; c.edge_emb_bf = k->x_emb_bf; c.emb_flag_row = b->n_x;
                         if (f16_ph) { c.precision = DP_PREC_F16X3; c.w1n = cw.w1h; c.w2n = cw.w2h; c.path_split = f16_csplit(slices_lp); }
      }
      }
      if (xs_ph) phc[1].s0_only = 1;
      if (pp_beside) {      // the combine launch starts from pp_tmp = pad(h_p) + phore-phore output instead of accumulating onto hp2
        phc[0].accumulate = 0; phc[0].resid = k->pp_tmp; phc[0].resid_stride = 100; phc[0].resid_width = 100;
      }
      if (node_side && fuse && !xs_ph) {
        if ((rc = timed_fused(phc, 2, 4, 5, sp_, pp_beside ? k->ev_join2 : nullptr))) return fail(rc);
      } else {
        for (int nv = 0; nv < 2; ++nv)
          if ((rc = (xs_ph && nv) ? timed_xs(phc[1], l, 5, false, hl, hp, sp_) : timed(phc[nv], l, 4 + nv, sp_))) return fail(rc);
      }
    }
    if (two && l + 1 < 3) barrier2();           // the next layer has two sides as well
    else if (two || (heads_early && l == 3)) join();
    cur ^= 1;
  }
  const float* hl = k->h_l[cur];
  // the two score heads side by side when a second stream was given: the LONGER chain (torsion bonds: conv over up to 32 edge slots per bond) stays on
  // the launch stream, where the caller's next kernel (the conformer update) follows it without a cross-queue wait (~10 us); the translation /
  // rotation head, done long before, goes to the second stream
  const bool tor_aux = aux != nullptr && b->n_tor > 0 && k->aux_mode != 1;
  hipStream_t s_tr = tor_aux ? aux : st;
  if (tor_aux) fork();
  if (!heads_early) hipLaunchKernelGGL(dp::center_edges_kernel, dim3(B), dim3(256), 0, s_tr, *b, *w, *k);
  {
    DpConvArgs c;
    memset(&c, 0, sizeof(c));
    c.kind = DP_CONV_FINAL; c.mode = DP_EDGES_CENTER; c.B = B; c.slices = 1; c.precision = prec_edge; c.max_scatter = 1; c.scat_ptr = b->pose_ptr; c.lig_ptr = b->lig_ptr;
    c.gather_tab = hl; c.gather_stride = 100; c.attr1_tab = hl; c.attr1_stride = 100;
    c.edge_emb = k->c_emb; c.edge_sh = k->c_sh; c.sh_stride = 9;
    c.w1p = w->final_conv.w1p; c.w2p = w->final_conv.w2p; c.bn_scale = w->final_conv.bn_scale; c.bn_shift = w->final_conv.bn_shift;
    c.out = k->gpred; c.out_stride = 12;
    if ((rc = dp_tp_conv(&c, s_tr))) return fail(rc);
  }
  hipLaunchKernelGGL(dp::trrot_head_kernel, dim3((B + 3) / 4), dim3(256), 0, s_tr, B, *w, *k, tr, rot);
  // torsion head
  if (b->n_tor > 0) {
    hipStream_t s2 = st;
    if (!heads_early) hipLaunchKernelGGL(dp::tor_graph_kernel, dim3(B), dim3(256), 0, s2, *b, *w, *k);
    hipLaunchKernelGGL(dp::tor_hb_kernel, dim3((b->n_tor * dp::NS + 255) / 256), dim3(256), 0, s2, *b, *k, hl);
    DpConvArgs c;
    memset(&c, 0, sizeof(c));
    c.kind = DP_CONV_TOR; c.mode = DP_EDGES_CSR; c.B = B; c.scat_ptr = b->tor_ptr; c.precision = prec_edge; c.max_scatter = b->max_t > 0 ? b->max_t : 1;
    c.slices = 1;
    while ((b->max_t + c.slices - 1) / c.slices > dp_conv_max_nodes(prec_edge)) ++c.slices;
    c.rowptr = k->t_rowptr; c.rowend = k->t_rowend; c.nbr = k->t_nbr; c.edge_scat = k->t_src;
    c.gather_tab = hl; c.gather_stride = 100; c.attr1_tab = hl; c.attr1_stride = 100; c.attr2_tab = k->t_hb; c.attr2_stride = 20;
    c.attr_swap = 1; c.edge_emb = k->t_emb; c.edge_sh = k->t_sh; c.sh_stride = 7;
    c.w1p = w->tor_conv.w1p; c.w2p = w->tor_conv.w2p; c.bn_scale = w->tor_conv.bn_scale; c.bn_shift = w->tor_conv.bn_shift;
    c.out = k->t_q; c.out_stride = 40;
    if (edge_x6 && w->tor_conv.w2x && w->tor_conv.bn_scale_x) {
      c.precision = DP_PREC_F16X3; c.w2p = w->tor_conv.w2x; c.bn_scale = w->tor_conv.bn_scale_x;
      c.slices = 1;
      while ((b->max_t + c.slices - 1) / c.slices > dp_conv_max_nodes(DP_PREC_BF16X6)) ++c.slices;
      // a batch that cannot fill the chip: fewer torsion bonds per workgroup (a bond has up to 32 edge slots; one 128-edge chunk per workgroup instead
      // of one after the other - the launch sits on the step's critical path between tor_graph and tor_head)
      while (B * c.slices < 512 && c.slices < 8 && c.slices < b->max_t) ++c.slices;
    }
    if ((rc = dp_tp_conv(&c, s2))) return fail(rc);
    hipLaunchKernelGGL(dp::tor_head_kernel, dim3((b->n_tor + 7) / 8), dim3(256), 0, s2, *b, *w, *k, tor);
    if (tor_aux) join();
  }
  return ev_rc ? ev_rc : (int)hipGetLastError();
}

}  // extern "C"
