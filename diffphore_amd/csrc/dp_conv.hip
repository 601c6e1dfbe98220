// dp_conv.hip - the edge-side fused tensor-product convolution's C entry points: sizes, host packers, dp_tp_conv (dispatch by layer kind to the
// kernels of dp_conv_impl.h, instantiated per kind in dp_conv_k0..k5.hip).
#include "dp_conv_impl.h"

#ifdef DP_CONV_ONE_TU      // cycle-stamp builds (-DDP_PROF: one device-side counter array): every kind in this translation unit, dp_conv_k*.hip empty
int dp_conv_launch_k0(const DpConvArgs& a, hipStream_t st) { return dp::launch<0>(a, st); }
int dp_conv_launch_k1(const DpConvArgs& a, hipStream_t st) { return dp::launch<1>(a, st); }
int dp_conv_launch_k2(const DpConvArgs& a, hipStream_t st) { return dp::launch<2>(a, st); }
int dp_conv_launch_k3(const DpConvArgs& a, hipStream_t st) { return dp::launch<3>(a, st); }
int dp_conv_launch_k4(const DpConvArgs& a, hipStream_t st) { return dp::launch<4>(a, st); }
int dp_conv_launch_k5(const DpConvArgs& a, hipStream_t st) { return dp::launch<5>(a, st); }
#endif

namespace dp {
// finishes a DP_EDGES_LIST conv: the scatter MEAN over the node's FULL degree (the listed edges are a subset of a dense conv's edges: the degree
// is the pose's number of gather nodes), the BatchNorm scale, and the sum onto `out`.  blockIdx.x = pose, blockIdx.y = a block of 256 (node, channel)
// items of it; a node's messages are added in list order by one thread per channel: deterministic.
__global__ __launch_bounds__(256) void list_reduce_kernel(const float* __restrict__ msg, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ scat_ptr,
                                                          const int32_t* __restrict__ deg_ptr, int d_out, const float* __restrict__ bn_scale,
                                                          float* __restrict__ out, int out_stride) {
  const int p = blockIdx.x;
  const int n0 = scat_ptr[p], n1 = scat_ptr[p + 1];
  const int i = blockIdx.y * 256 + threadIdx.x;
  if (i >= (n1 - n0) * d_out) return;
  const int n = n0 + i / d_out, c = i % d_out;
  const int k0 = rowptr[n], k1 = rowptr[n + 1];
  if (k0 == k1) return;
  const int deg = deg_ptr[p + 1] - deg_ptr[p];
  float s = 0.f;
  for (int k = k0; k < k1; ++k) s += msg[(size_t)k * d_out + c];
  out[(size_t)n * out_stride + c] += (s / (float)(deg > 1 ? deg : 1)) * bn_scale[c];
}

// finishes a DP_EDGES_LIST launch that carried a WHOLE conv (every edge of every node listed, grouped by scatter node): scatter mean over the node's
// own edges, BatchNorm affine, residual / accumulate - what conv_epilogue does for the CSR kernel.  One thread per (node, channel).  The edge rows of
// the first pose's nodes start at 0 (rowptr is global), so "node n mod bcast_nodes" addresses them directly.
__global__ __launch_bounds__(256) void list_finish_kernel(const float* __restrict__ msg, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
                                                          int n_nodes, int d_out, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                          const float* __restrict__ resid, int resid_stride, int resid_width, float* __restrict__ out,
                                                          int out_stride, int accumulate, int bcast_nodes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_nodes * d_out) return;
  const int n = i / d_out, c = i - n * d_out;
  const int nm = bcast_nodes > 0 ? n % bcast_nodes : n;      // (one pharmacophore's conv for every pose that carries it: DpBatch.ph_shared)
  const int k0 = rowptr[nm], k1 = rowend[nm];
  float s = 0.f;
  for (int k = k0; k < k1; ++k) s += msg[(size_t)k * d_out + c];
  float v = s / (float)(k1 - k0 > 1 ? k1 - k0 : 1);
  v = v * bn_scale[c] + bn_shift[c];
  float* o = out + (size_t)n * out_stride + c;
  if (accumulate) *o += v;
  else *o = v + ((resid != nullptr && c < resid_width) ? resid[(size_t)n * resid_stride + c] : 0.f);
}
}  // namespace dp

extern "C" {

int dp_abi_version(void) { return DP_ABI_VERSION; }

int dp_list_finish(const float* msg, const int32_t* rowptr, const int32_t* rowend, int32_t n_nodes, int32_t d_out, const float* bn_scale, const float* bn_shift,
                   const float* resid, int32_t resid_stride, int32_t resid_width, float* out, int32_t out_stride, int32_t accumulate, int32_t bcast_nodes,
                   void* stream) {
  if (!msg || !rowptr || !rowend || n_nodes <= 0 || d_out <= 0 || !bn_scale || !bn_shift || !out || out_stride < d_out || bcast_nodes < 0) return DP_E_BADARG;
  const long long total = (long long)n_nodes * d_out;
  hipLaunchKernelGGL(dp::list_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, msg, rowptr, rowend, n_nodes, d_out, bn_scale,
                     bn_shift, resid, resid_stride, resid_width, out, out_stride, accumulate, bcast_nodes);
  return (int)hipGetLastError();
}

int dp_list_reduce(const float* msg, const int32_t* rowptr, int32_t B, const int32_t* scat_ptr, int32_t max_scatter, const int32_t* deg_ptr, int32_t d_out,
                   const float* bn_scale, float* out, int32_t out_stride, void* stream) {
  if (!msg || !rowptr || B <= 0 || !scat_ptr || max_scatter <= 0 || !deg_ptr || d_out <= 0 || !bn_scale || !out || out_stride < d_out) return DP_E_BADARG;
  const unsigned blocks = (unsigned)((max_scatter * d_out + 255) / 256);
  hipLaunchKernelGGL(dp::list_reduce_kernel, dim3((unsigned)B, blocks), dim3(256), 0, (hipStream_t)stream, msg, rowptr, scat_ptr, deg_ptr, d_out, bn_scale, out, out_stride);
  return (int)hipGetLastError();
}

int dp_conv_max_nodes(int32_t precision) { return precision == DP_PREC_F32 ? dp::MAX_NODES : dp::MAX_NODES_B; }

int dp_conv_dims(int32_t kind, int32_t* d_in, int32_t* d_out, int32_t* numel, int32_t* n_attr, int32_t* hidden) {
#define DP_CASE(K)                                                                        \
  case K:                                                                                 \
    if (d_in) *d_in = dp::Spec<K>::D_IN;                                                  \
    if (d_out) *d_out = dp::Spec<K>::D_OUT;                                               \
    if (numel) *numel = dp::Spec<K>::NUMEL;                                               \
    if (n_attr) *n_attr = dp::Spec<K>::N_ATTR;                                            \
    if (hidden) *hidden = dp::Spec<K>::HID;                                               \
    return DP_OK;
  switch (kind) {
    DP_CASE(0) DP_CASE(1) DP_CASE(2) DP_CASE(3) DP_CASE(4) DP_CASE(5)
    default: return DP_E_BADARG;
  }
#undef DP_CASE
}

int dp_conv_w1p_floats(int32_t kind) {
  switch (kind) {
    case 0: return dp::Geo<0>::W1P_FLOATS; case 1: return dp::Geo<1>::W1P_FLOATS; case 2: return dp::Geo<2>::W1P_FLOATS;
    case 3: return dp::Geo<3>::W1P_FLOATS; case 4: return dp::Geo<4>::W1P_FLOATS; case 5: return dp::Geo<5>::W1P_FLOATS;
    default: return DP_E_BADARG;
  }
}
int dp_conv_w2p_floats(int32_t kind, int32_t precision) {
  switch (kind) {
    case 0: return dp::w2p_floats<0>(precision); case 1: return dp::w2p_floats<1>(precision); case 2: return dp::w2p_floats<2>(precision);
    case 3: return dp::w2p_floats<3>(precision); case 4: return dp::w2p_floats<4>(precision); case 5: return dp::w2p_floats<5>(precision);
    default: return DP_E_BADARG;
  }
}

// HOST pointers in, HOST pointers out: re-orders one TensorProductConvLayer's `fc` parameters (reference layout:
// fc.0.weight [HID, N_ATTR], fc.0.bias [HID], fc.3.weight [numel, HID], fc.3.bias [numel]) into MFMA operand order.
int dp_pack_conv_weights(int32_t kind, int32_t precision, const float* fc0_w, const float* fc0_b, const float* fc3_w,
                         const float* fc3_b, float* w1p, float* w2p) {
  if (!fc0_w || !fc0_b || !fc3_w || !fc3_b || !w1p || !w2p) return DP_E_BADARG;
  if (precision != DP_PREC_F32 && precision != DP_PREC_BF16X6 && precision != DP_PREC_BF16X3 && precision != DP_PREC_BF16X9 && precision != DP_PREC_BF16 && precision != DP_PREC_F16X3)
    return DP_E_BADARG;
  switch (kind) {
#define DP_CASE(K)                                                                                            \
  case K:                                                                                                     \
    dp::pack_w1<K>(fc0_w, fc0_b, w1p);                                                                        \
    if (precision == DP_PREC_F32) dp::pack_w2<K>(fc3_w, fc3_b, w2p);                                          \
    else if (precision == DP_PREC_BF16X6 || precision == DP_PREC_BF16X9) dp::pack_w2_bf16<K, 3>(fc3_w, fc3_b, (unsigned short*)w2p);         \
    else if (precision == DP_PREC_BF16) dp::pack_w2_bf16<K, 1>(fc3_w, fc3_b, (unsigned short*)w2p);           \
    else if (precision == DP_PREC_F16X3) dp::pack_w2_f16<K>(fc3_w, fc3_b, (unsigned short*)w2p);              \
    else dp::pack_w2_bf16<K, 2>(fc3_w, fc3_b, (unsigned short*)w2p);                                          \
    return DP_OK;
    DP_CASE(0) DP_CASE(1) DP_CASE(2) DP_CASE(3) DP_CASE(4) DP_CASE(5)
#undef DP_CASE
    default: return DP_E_BADARG;
  }
}

int dp_resolve_conv_algo(int32_t algo, int32_t precision, int32_t B) {
  // node-side: exact fp32 (dp_conv_node_impl.h) or plain bf16 operands (dp_conv_node_bf16_impl.h); the split-bf16 precisions exist
  // edge-side only
  const bool has_node = precision == DP_PREC_F32 || precision == DP_PREC_BF16;
  if (algo == DP_ALGO_NODE) return has_node ? DP_ALGO_NODE : DP_ALGO_EDGE;
  if (algo == DP_ALGO_EDGE) return DP_ALGO_EDGE;
  return (has_node && B >= DP_NODE_AUTO_MIN_POSES) ? DP_ALGO_NODE : DP_ALGO_EDGE;
}

int dp_tp_conv(const DpConvArgs* a, void* stream) {
  if (a && a->algo == DP_ALGO_NODE) return dp_tp_conv_node(a, stream);
  if (a && a->mode == DP_EDGES_LIST) {           // explicit edge list -> per-edge messages (kinds 0-3, DP_PREC_F16X3; finished by dp_list_reduce)
    if (a->n_list < 0 || (a->n_list > 0 && (!a->nbr || !a->edge_scat || !a->msg_out)) || !a->gather_tab || !a->attr1_tab || !a->attr2_tab ||
        !a->edge_emb || !a->edge_sh || !a->w1p || !a->w2p)
      return DP_E_BADARG;
    if (a->kind < 0 || a->kind > 3) return DP_E_UNSUPPORTED;
    hipStream_t st_ = (hipStream_t)stream;
    switch (a->kind) {
      case 0: return dp_conv_launch_k0(*a, st_);
      case 1: return dp_conv_launch_k1(*a, st_);
      case 2: return dp_conv_launch_k2(*a, st_);
      default: return dp_conv_launch_k3(*a, st_);
    }
  }
  if (!a || a->B <= 0 || a->slices <= 0 || !a->scat_ptr || !a->gather_tab || !a->attr1_tab || !a->edge_emb || !a->edge_sh ||
      !a->w1p || !a->w2p || !a->bn_scale || !a->bn_shift || !a->out)
    return DP_E_BADARG;
  if (a->mode == DP_EDGES_CSR && (!a->rowptr || !a->rowend || !a->nbr || !a->edge_scat)) return DP_E_BADARG;
  if (a->mode != DP_EDGES_CSR && !a->lig_ptr) return DP_E_BADARG;
  if ((a->mode == DP_EDGES_P2L || a->mode == DP_EDGES_L2P) && (!a->ph_ptr || !a->x_ptr)) return DP_E_BADARG;
  if (a->kind != DP_CONV_FINAL && !a->attr2_tab) return DP_E_BADARG;
  if (a->max_scatter <= 0) return DP_E_BADARG;
  if ((a->max_scatter + a->slices - 1) / a->slices > dp_conv_max_nodes(a->precision)) return DP_E_TOOLARGE;
  hipStream_t st = (hipStream_t)stream;
  switch (a->kind) {
    case 0: return dp_conv_launch_k0(*a, st);
    case 1: return dp_conv_launch_k1(*a, st);
    case 2: return dp_conv_launch_k2(*a, st);
    case 3: return dp_conv_launch_k3(*a, st);
    case 4: return dp_conv_launch_k4(*a, st);
    case 5: return dp_conv_launch_k5(*a, st);
    default: return DP_E_BADARG;
  }
}

}  // extern "C"
