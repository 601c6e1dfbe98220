// dp_dev.h - small device helpers shared by the graph / head / update kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffphore_amd.h"

namespace dp {

constexpr int NS = 20;           // scalar channels / sigma-embedding width / distance-embedding width
constexpr int NPT = 11;          // pharmacophore types, 10 = exclusion sphere
constexpr int EX = 10;
constexpr int TP_STRIDE = 32;    // floats per pose in DpWork::tparams
constexpr int MAXN_L = 256, MAXN_P = 256, MAXT = 128, TOR_SLOTS = 32;

// Parameters are read through the CONSTANT address space: a wave-uniform address then becomes a scalar load (s_load_dwordxN into SGPRs, one
// per wave) whatever stores surround it.  As plain global pointers the compiler must assume that the kernel's own stores may alias them and
// falls back to one vector load per weight and lane group - the graph kernels then wait on ~400 dependent global loads per edge.
// The pointer is made opaque at every use: otherwise the loads of a whole 20 x 73 matrix are hoisted out of the per-edge loops, which takes
// ~1 500 SGPRs - parked in VGPR lanes and fetched back with one v_readlane per weight and iteration.
typedef const float __attribute__((address_space(4)))* kptr;
__device__ __forceinline__ kptr K(const float* p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  return (kptr)v;
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }     // F.softplus, beta=1, threshold=20
__device__ __forceinline__ float leaky_f(float x) { return x >= 0.f ? x : 0.01f * x; }

// e3nn spherical harmonics l<=2, normalize=True (v / max(|v|,1e-12)), 'component' normalisation.
__device__ __forceinline__ void sh9(float x, float y, float z, float* o) {
  const float n = sqrtf(x * x + y * y + z * z);
  const float inv = 1.f / fmaxf(n, 1e-12f);
  x *= inv; y *= inv; z *= inv;
  const float s3 = 1.7320508075688772f, s5 = 2.23606797749979f, s15 = 3.872983346207417f;
  o[0] = 1.f;
  o[1] = s3 * x; o[2] = s3 * y; o[3] = s3 * z;
  o[4] = s15 * x * z;
  o[5] = s15 * x * y;
  o[6] = s5 * (y * y - 0.5f * (x * x + z * z));
  o[7] = s15 * y * z;
  o[8] = 0.5f * s15 * (z * z - x * x);
}

// GaussianSmearing.forward: exp(coeff * (d - offset[k])^2), offset = the module's registered buffer (linspace(0, stop, 20))
__device__ __forceinline__ void gauss20(float d, const float* __restrict__ mu_, float coeff, float* g) {
  const kptr mu = K(mu_);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const float t = d - mu[k];
    g[k] = expf(coeff * (t * t));
  }
}

// out[o] = b[o] + sum_i W[o*ld + off + i] * x[i]   (uniform weight addresses -> scalar loads)
template <int NOUT, int NIN>
__device__ __forceinline__ void matvec_acc(const float* __restrict__ W_, int ld, int off, const float* x, float* out) {
  const kptr W = K(W_);
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    float a = out[o];
#pragma unroll
    for (int i = 0; i < NIN; ++i) a = fmaf(W[o * ld + off + i], x[i], a);
    out[o] = a;
  }
}

template <int N>
__device__ __forceinline__ void store_row(float* __restrict__ dst, const float* v) {
#pragma unroll
  for (int i = 0; i < N; ++i) dst[i] = v[i];
}

// second Linear of a 20->20 MLP with ReLU in between:  out = b3 + W3 relu(h)
__device__ __forceinline__ void mlp_tail20(const DpMlp& m, const float* h, float* out) {
  const kptr b3 = K(m.b3), w3 = K(m.w3);
#pragma unroll
  for (int o = 0; o < NS; ++o) {
    float a = b3[o];
#pragma unroll
    for (int i = 0; i < NS; ++i) a = fmaf(w3[o * NS + i], fmaxf(h[i], 0.f), a);
    out[o] = a;
  }
}

}  // namespace dp

// Whether the phore-phore convs of a batch run as ONE DP_EDGES_LIST launch over all its edges (+ dp_list_finish) instead of the CSR launch with its
// workgroups per (pose, slice): the CSR kernel works in 128-edge chunks per workgroup, and a small pharmacophore leaves them mostly empty (configs[1]:
// 150 edges per pose = 2 chunks, 59 % full; a 40-pose call: 8 slices of 19 edges).  The list packs the batch's edges into full chunks.  Estimated from
// the batch's averages (host arithmetic; dp_work_bind sizes DpWork.pp_msg by it, dp_forward takes the list when that array is there).
static inline bool dp_pp_as_list(const DpBatch* b) {
  if (b->n_pp <= 0 || b->B <= 0) return false;
  const int mx = b->max_nl > b->max_np ? b->max_nl : b->max_np;
  int slices = 1;
  while ((mx + slices - 1) / slices > 32) ++slices;                 // (dp_conv_max_nodes of the split-bf16 kernels, as dp_forward deals the slices)
  while (b->B * slices < 512 && slices < 8) ++slices;
  const double per_slice = (double)b->n_pp / b->B / slices;
  const double chunks = (double)b->B * slices * (double)(long long)((per_slice + 127.0) / 128.0);
  return (double)b->n_pp < 0.7 * chunks * 128.0;
}

