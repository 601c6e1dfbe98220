// dp_conv_node.hip - host half and C entry points of the node-side formulation of the fused tensor-product convolution
// (TensorProductConvLayer.forward, score_model_phore.py:134-149).  The device code is dp_conv_node_impl.h, instantiated per layer kind
// in dp_conv_node_k0.hip .. dp_conv_node_k3.hip (four translation units: they compile side by side).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "dp_conv_node_f16_impl.h"
#include "dp_f16.h"


namespace dpn {

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

// W2 (+ bias row) per (path, in-channel tile, M tile) in GEMM-B's A-fragment order, with the path coefficient, the CG norm and - when
// `bn_scale` is given - the e3nn BatchNorm(eval) scale of the output channel folded in (BN is a per-channel affine map after the
// scatter mean, SURVEY.md N8: scale * (W2 T) = (scale * W2) T)
template <int KIND> void pack_w2n(const float* fc3_w, const float* fc3_b, const float* bn_scale, float* w2n) {
  using S = Spec<KIND>;
  for (int p = 0; p < S::NPATH; ++p) {
    const PathC& pc = S::P[p];
    const int dout = d_out_of(pc.ztype);
    const double scale = __builtin_sqrt((double)(2 * (dout == 3 ? 1 : 0) + 1) / (double)pc.fan_in) * cg_const(pc.ztype);
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
                const double bn = bn_scale ? (double)bn_scale[pc.out_off + w * dout] : 1.0;     // one scale per multiplicity copy
                v = (float)(scale * bn * (double)(k < HIDN ? fc3_w[widx * HIDN + k] : fc3_b[widx]));
              }
              dst[(size_t)(q * 64 + lane) * steps + s] = v;
            }
      }
  }
}

// ---- DP_PREC_BF16 (dp_conv_node_bf16_impl.h): bf16 MFMA fragments ------------------------------------------------------------------
static inline unsigned short bf16_rne(float x) {
  unsigned int u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// three B-operand images of the first layer per wave q: lane (k = 16 q + lane % 16, group g) holds features 8g..8g+7 of
//   image 0: the edge-embedding block (fc.0 columns 0..19); feature 20 = -1e30: the row of the "edge does not exist" flag
//   image 1 / 2: the attr1 / attr2 block (columns 20..39 / 40..59); feature 20 = fc.0's bias (1.0 for the constant-one unit 60 that carries
//   fc.3's bias): only the SCATTER node's tile feeds a 1 there, so the bias enters once
template <int KIND> void pack_w1b(const float* fc0_w, const float* fc0_b, unsigned short* w1b) {
  using S = Spec<KIND>;
  static_assert(S::HID == HIDN && S::N_ATTR == 60, "node-side kernel: layer convs only");
  for (int img = 0; img < 3; ++img)
    for (int q = 0; q < 4; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int g = lane >> 4, k = 16 * q + (lane & 15), f = 8 * g + j;
          float v = 0.f;
          if (f < 20) v = k < HIDN ? fc0_w[k * 60 + 20 * img + f] : 0.f;
          else if (f == dpb::EMB_FLAG) v = img == 0 ? -1e30f : (k < HIDN ? fc0_b[k] : (k == HIDN ? 1.f : 0.f));
          w1b[(((size_t)img * 4 + q) * 64 + lane) * 8 + j] = bf16_rne(v);
        }
}
// second layer as GEMM-B A fragments [path][supertile][K chunk][M tile][wave q][lane][8]: lane (w = 16 mt + lane % 16, group g) holds
// W2[(u, w)][k = 16 q + 4 g + (j & 3)] for the chunk's two in-channels u = 10 st + 2 ch + (j >> 2); path coefficient, CG norm and the
// BatchNorm scale folded in as in pack_w2n, then rounded to bf16 once
template <int KIND> void pack_w2b(const float* fc3_w, const float* fc3_b, const float* bn_scale, unsigned short* w2b) {
  using S = Spec<KIND>;
  for (int p = 0; p < S::NPATH; ++p) {
    const PathC& pc = S::P[p];
    const int dout = d_out_of(pc.ztype);
    const double scale = __builtin_sqrt((double)(2 * (dout == 3 ? 1 : 0) + 1) / (double)pc.fan_in) * cg_const(pc.ztype);
    const int MT = path_mt<KIND>(p);
    for (int st = 0; st < dpb::st_count<KIND>(p); ++st)
      for (int ch = 0; ch < dpb::SUPER_CH; ++ch)
        for (int mt = 0; mt < MT; ++mt)
          for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane) {
              unsigned short* dst = w2b + ((size_t)dpb::w2b_path_off<KIND>(p) + (size_t)((st * dpb::SUPER_CH + ch) * MT + mt) * 256 + q * 64 + lane) * 8;
              for (int j = 0; j < 8; ++j) {
                const int g = lane >> 4, w = 16 * mt + (lane & 15);
                const int u = dpb::SUPER_NU * st + 2 * ch + (j >> 2), k = 16 * q + 4 * g + (j & 3);
                float v = 0.f;
                if (w < pc.mul_out && k <= HIDN) {
                  const int widx = pc.w_off + u * pc.mul_out + w;
                  const double bn = bn_scale ? (double)bn_scale[pc.out_off + w * dout] : 1.0;
                  v = (float)(scale * bn * (double)(k < HIDN ? fc3_w[widx * HIDN + k] : fc3_b[widx]));
                }
                dst[j] = bf16_rne(v);
              }
            }
  }
}

// emb [n, 20] fp32 -> [n + 1, 32] bf16 rows (RNE) + the flag row; one thread per 8 features
__global__ void edge_emb_bf16_kernel(const float* __restrict__ emb, int n_rows, uint16_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (n_rows + 1) * 4) return;
  const int row = i >> 2, g = i & 3;
  dpb::u32x4 v = {0u, 0u, 0u, 0u};
  if (row < n_rows) {
    if (g < 2) {
      const float* p = emb + (size_t)row * 20 + 8 * g;
      v = dpb::u32x4{dpb::pk2(p[0], p[1]), dpb::pk2(p[2], p[3]), dpb::pk2(p[4], p[5]), dpb::pk2(p[6], p[7])};
    } else if (g == 2) {
      const float* p = emb + (size_t)row * 20 + 16;
      v = dpb::u32x4{dpb::pk2(p[0], p[1]), dpb::pk2(p[2], p[3]), 0u, 0u};
    }
  } else if (g == 2) {
    v[2] = 0x3F80u;                                 // feature 20 = 1.0
  }
  *reinterpret_cast<dpb::u32x4*>(out + (size_t)row * dpb::EMB_ROW + 8 * g) = v;
}

// ---- DP_PREC_F16X3 (dp_conv_node_f16_impl.h): two-limb fp16 MFMA fragments, exact power-of-two scales ------------------------------------
// first layer: three B-operand images x two limbs per wave q, as pack_w1b, stored as 16 W1 (dpf::W1_SCALE; relu commutes with the scale, the kernel
// divides the hidden vectors by it); the flag row is dpf::FLAG_W: [img][limb][q][lane][8]
template <int KIND> void pack_w1h(const float* fc0_w, const float* fc0_b, unsigned short* w1h) {
  using S = Spec<KIND>;
  static_assert(S::HID == HIDN && S::N_ATTR == 60, "node-side kernel: layer convs only");
  for (int img = 0; img < 3; ++img)
    for (int q = 0; q < 4; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int g = lane >> 4, k = 16 * q + (lane & 15), f = 8 * g + j;
          float v = 0.f;
          if (f < 20) v = k < HIDN ? dpf::W1_SCALE * fc0_w[k * 60 + 20 * img + f] : 0.f;
          else if (f == dpb::EMB_FLAG) v = img == 0 ? dpf::FLAG_W : dpf::W1_SCALE * (k < HIDN ? fc0_b[k] : (k == HIDN ? 1.f : 0.f));
          unsigned short hi, lo;
          f16_split(v, hi, lo);
          w1h[((((size_t)img * 2 + 0) * 4 + q) * 64 + lane) * 8 + j] = hi;
          w1h[((((size_t)img * 2 + 1) * 4 + q) * 64 + lane) * 8 + j] = lo;
        }
}
// second layer as GEMM-B A fragments [header][path][supertile][limb][K chunk][M tile][wave q][lane][8] (fragment order of pack_w2b), stored as
// W2 * 2^w with w = the largest exponent that keeps max |W2 * 2^w| <= 2048; header float[0] = w
template <int KIND> void pack_w2h(const float* fc3_w, const float* fc3_b, const float* bn_scale, unsigned short* w2h) {
  using S = Spec<KIND>;
  auto value = [&](int p, int u, int w, int k) -> double {
    const PathC& pc = S::P[p];
    const int dout = d_out_of(pc.ztype);
    if (w >= pc.mul_out || k > HIDN) return 0.0;
    const double scale = __builtin_sqrt((double)(2 * (dout == 3 ? 1 : 0) + 1) / (double)pc.fan_in) * cg_const(pc.ztype);
    const int widx = pc.w_off + u * pc.mul_out + w;
    const double bn = bn_scale ? (double)bn_scale[pc.out_off + w * dout] : 1.0;
    return scale * bn * (double)(k < HIDN ? fc3_w[widx * HIDN + k] : fc3_b[widx]);
  };
  double mx = 0.0;
  for (int p = 0; p < S::NPATH; ++p)
    for (int u = 0; u < S::P[p].mul_in; ++u)
      for (int w = 0; w < S::P[p].mul_out; ++w)
        for (int k = 0; k <= HIDN; ++k) { const double v = __builtin_fabs(value(p, u, w, k)); if (v > mx) mx = v; }
  const int wexp = f16_image_exponent(mx);
  const double sc = __builtin_ldexp(1.0, wexp);
  float* hdr = reinterpret_cast<float*>(w2h);
  for (int i = 0; i < dpf::HDR_QUADS * 4; ++i) hdr[i] = 0.f;
  hdr[0] = (float)wexp;
  for (int p = 0; p < S::NPATH; ++p) {
    const int MT = path_mt<KIND>(p);
    for (int st = 0; st < dpb::st_count<KIND>(p); ++st)
      for (int ch = 0; ch < dpb::SUPER_CH; ++ch)
        for (int mt = 0; mt < MT; ++mt)
          for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j) {
                const int g = lane >> 4, w = 16 * mt + (lane & 15);
                const int u = dpb::SUPER_NU * st + 2 * ch + (j >> 2), k = 16 * q + 4 * g + (j & 3);
                unsigned short hi, lo;
                f16_split((float)(sc * value(p, u, w, k)), hi, lo);
                const size_t stq = (size_t)dpf::w2h_path_off<KIND>(p) + (size_t)(st * 2 * dpb::SUPER_CH * MT) * 256;
                w2h[(stq + (size_t)((0 * dpb::SUPER_CH + ch) * MT + mt) * 256 + q * 64 + lane) * 8 + j] = hi;
                w2h[(stq + (size_t)((1 * dpb::SUPER_CH + ch) * MT + mt) * 256 + q * 64 + lane) * 8 + j] = lo;
              }
  }
}
// emb [n, 20] fp32 -> [n + 1, 64] fp16 rows (hi limb: 20 features | zeros to 32; lo limb: the same) + the flag row (hi feature 20 = 1.0);
// one thread per 8 features of a row
__global__ void edge_emb_f16_kernel(const float* __restrict__ emb, int n_rows, uint16_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (n_rows + 1) * 4) return;
  const int row = i >> 2, g = i & 3;
  dpb::u32x4 vh = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
  if (row < n_rows) {
    if (g < 3) {
      const float* p = emb + (size_t)row * 20 + 8 * g;
      const int npair = g < 2 ? 4 : 2;
      for (int k = 0; k < npair; ++k) { const dpf::HL s = dpf::split2(p[2 * k], p[2 * k + 1]); vh[k] = s.h; vl[k] = s.l; }
    }
  } else if (g == 2) {
    vh[2] = 0x3C00u;                                // feature 20 = 1.0
  }
  *reinterpret_cast<dpb::u32x4*>(out + (size_t)row * dpf::EMBH_ROW + 8 * g) = vh;
  *reinterpret_cast<dpb::u32x4*>(out + (size_t)row * dpf::EMBH_ROW + 32 + 8 * g) = vl;
}

// conv split: out = (previous out | residual stream | 0) + ((part[0] + part[1]) + part[2]) + the convs' BatchNorm shifts, in this fixed order
struct CombineArgs {
  const float* part;
  int n_conv, part_stride, n_nodes, d_out;
  float* out;
  int out_stride, accumulate;
  const float* resid;
  int resid_stride, resid_width;
  const float* shift[3];
};
__global__ void conv_combine_kernel(CombineArgs c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c.n_nodes * c.d_out) return;
  const int node = i / c.d_out, ch = i - node * c.d_out;
  float p = c.part[(size_t)node * c.out_stride + ch], sh = c.shift[0][ch];
  for (int k = 1; k < c.n_conv; ++k) {
    p += c.part[(size_t)k * c.part_stride + (size_t)node * c.out_stride + ch];
    sh += c.shift[k][ch];
  }
  float* o = c.out + (size_t)node * c.out_stride + ch;
  const float base = c.accumulate ? *o : ((c.resid != nullptr && ch < c.resid_width) ? c.resid[(size_t)node * c.resid_stride + ch] : 0.f);
  *o = base + (p + sh);
}

static int launch_kind_bf16(int kind, const FusedArgs& fa, int xmode, hipStream_t st) {
  switch (kind) {
    case 0: return dpb_launch_k0(fa, xmode, st);
    case 1: return dpb_launch_k1(fa, xmode, st);
    case 2: return dpb_launch_k2(fa, xmode, st);
    default: return dpb_launch_k3(fa, xmode, st);
  }
}

static int launch_kind_f16(int kind, const FusedArgs& fa, int xmode, hipStream_t st) {
  switch (kind) {
    case 0: return dpf_launch_k0(fa, xmode, st);
    case 1: return dpf_launch_k1(fa, xmode, st);
    case 2: return dpf_launch_k2(fa, xmode, st);
    default: return dpf_launch_k3(fa, xmode, st);
  }
}

static int launch_kind(int kind, const FusedArgs& fa, int xmode, hipStream_t st) {
  if (fa.conv[0].precision == DP_PREC_BF16) return launch_kind_bf16(kind, fa, xmode, st);
  if (fa.conv[0].precision == DP_PREC_F16X3) return launch_kind_f16(kind, fa, xmode, st);
  switch (kind) {
    case 0: return dpn_launch_k0(fa, xmode, st);
    case 1: return dpn_launch_k1(fa, xmode, st);
    case 2: return dpn_launch_k2(fa, xmode, st);
    default: return dpn_launch_k3(fa, xmode, st);
  }
}

static int check_conv(const DpConvArgs* a) {
  if (a->B <= 0 || a->slices <= 0 || !a->scat_ptr || !a->lig_ptr || !a->gather_tab || !a->attr1_tab || !a->attr2_tab || !a->edge_emb ||
      !a->edge_sh || !a->w1n || !a->w2n || !a->bn_shift)
    return DP_E_BADARG;
  if (a->mode != DP_EDGES_P2L && a->mode != DP_EDGES_L2P && a->mode != DP_EDGES_LLPAIR) return DP_E_UNSUPPORTED;
  if (a->mode == DP_EDGES_LLPAIR ? (!a->pair_a || !a->pair_b || !a->pair_ptr || !a->rowptr || !a->rowend) : (!a->ph_ptr || !a->x_ptr)) return DP_E_BADARG;
  if (a->kind < 0 || a->kind > 3 || a->sh_stride < 9) return DP_E_UNSUPPORTED;
  if (a->precision != DP_PREC_F32 && a->precision != DP_PREC_BF16 && a->precision != DP_PREC_F16X3) return DP_E_UNSUPPORTED;
  if (a->precision != DP_PREC_F32 && (!a->edge_emb_bf || a->emb_flag_row < 0)) return DP_E_BADARG;
  if (a->max_scatter <= 0) return DP_E_BADARG;
  if ((a->max_scatter + a->slices - 1) / a->slices > NB) return DP_E_TOOLARGE;
  return DP_OK;
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

int dp_conv_w1b_floats(int32_t kind) { return kind >= 0 && kind <= 3 ? dpb::W1B_QUADS * 4 : DP_E_BADARG; }
int dp_conv_w2b_floats(int32_t kind) {
  switch (kind) {
    case 0: return dpb::w2b_quads<0>() * 4; case 1: return dpb::w2b_quads<1>() * 4;
    case 2: return dpb::w2b_quads<2>() * 4; case 3: return dpb::w2b_quads<3>() * 4;
    default: return DP_E_BADARG;
  }
}
int dp_pack_conv_weights_node_bf16(int32_t kind, const float* fc0_w, const float* fc0_b, const float* fc3_w, const float* fc3_b,
                                   const float* bn_scale, float* w1b, float* w2b) {
  if (!fc0_w || !fc0_b || !fc3_w || !fc3_b || !w1b || !w2b) return DP_E_BADARG;
  switch (kind) {
#define DP_CASE(K) case K: dpn::pack_w1b<K>(fc0_w, fc0_b, (unsigned short*)w1b); dpn::pack_w2b<K>(fc3_w, fc3_b, bn_scale, (unsigned short*)w2b); return DP_OK;
    DP_CASE(0) DP_CASE(1) DP_CASE(2) DP_CASE(3)
#undef DP_CASE
    default: return DP_E_UNSUPPORTED;
  }
}
int dp_conv_w1h_floats(int32_t kind) { return kind >= 0 && kind <= 3 ? dpf::W1H_QUADS * 4 : DP_E_BADARG; }
int dp_conv_w2h_floats(int32_t kind) {
  switch (kind) {
    case 0: return dpf::w2h_quads<0>() * 4; case 1: return dpf::w2h_quads<1>() * 4;
    case 2: return dpf::w2h_quads<2>() * 4; case 3: return dpf::w2h_quads<3>() * 4;
    default: return DP_E_BADARG;
  }
}
int dp_pack_conv_weights_node_f16(int32_t kind, const float* fc0_w, const float* fc0_b, const float* fc3_w, const float* fc3_b,
                                  const float* bn_scale, float* w1h, float* w2h) {
  if (!fc0_w || !fc0_b || !fc3_w || !fc3_b || !w1h || !w2h) return DP_E_BADARG;
  switch (kind) {
#define DP_CASE(K) case K: dpn::pack_w1h<K>(fc0_w, fc0_b, (unsigned short*)w1h); dpn::pack_w2h<K>(fc3_w, fc3_b, bn_scale, (unsigned short*)w2h); return DP_OK;
    DP_CASE(0) DP_CASE(1) DP_CASE(2) DP_CASE(3)
#undef DP_CASE
    default: return DP_E_UNSUPPORTED;
  }
}
int dp_edge_emb_f16(const float* emb, int32_t n_rows, uint16_t* out, void* stream) {
  if (!emb || !out || n_rows < 0) return DP_E_BADARG;
  const int total = (n_rows + 1) * 4;
  hipLaunchKernelGGL(dpn::edge_emb_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, emb, n_rows, out);
  return (int)hipGetLastError();
}
int dp_edge_emb_bf16(const float* emb, int32_t n_rows, uint16_t* out, void* stream) {
  if (!emb || !out || n_rows < 0) return DP_E_BADARG;
  const int total = (n_rows + 1) * 4;
  hipLaunchKernelGGL(dpn::edge_emb_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, emb, n_rows, out);
  return (int)hipGetLastError();
}

int dp_pack_conv_weights_node(int32_t kind, const float* fc0_w, const float* fc0_b, const float* fc3_w, const float* fc3_b,
                              const float* bn_scale, float* w1n, float* w2n) {
  if (!fc0_w || !fc0_b || !fc3_w || !fc3_b || !w1n || !w2n) return DP_E_BADARG;
  switch (kind) {
#define DP_CASE(K) case K: dpn::pack_w1n<K>(fc0_w, fc0_b, w1n); dpn::pack_w2n<K>(fc3_w, fc3_b, bn_scale, w2n); return DP_OK;
    DP_CASE(0) DP_CASE(1) DP_CASE(2) DP_CASE(3)
#undef DP_CASE
    default: return DP_E_UNSUPPORTED;
  }
}

// Fused node-side launch: convs[0..n) scatter onto the same node groups and their outputs add up (score_model_phore.py:703-710).
// Allowed sequences: [LLPAIR][P2L][P2L] (any prefix / suffix of it that is not empty) or [L2P][L2P].  convs[0] names out / resid /
// accumulate; kind, B, slices, scat_ptr, max_scatter and path_split must agree.
static int fused_launch(const DpConvArgs* convs, int32_t n, int32_t n_nodes, float* part, int64_t part_floats, void* stream, hipEvent_t before_combine = nullptr);

int dp_tp_conv_node_fused(const DpConvArgs* convs, int32_t n, void* stream) { return fused_launch(convs, n, 0, nullptr, 0, stream); }

// The same sum with every conv on its OWN workgroups (grid x n) - for launches that cannot fill the chip: a workgroup runs one conv's
// passes instead of all convs' back to back, the partial outputs meet in `part` and a second, tiny launch adds them up in a fixed order
// (deterministic; equal to the fused launch to fp32 rounding).  Needs path_split = 1 and scat_ptr[0] == 0; n_nodes = scat_ptr[B].
int dp_tp_conv_node_fused_split(const DpConvArgs* convs, int32_t n, int32_t n_nodes, float* part, int64_t part_floats, void* stream) {
  if (!convs || n < 2 || n_nodes <= 0 || !part || !convs[0].path_split) return DP_E_BADARG;
  return fused_launch(convs, n, n_nodes, part, part_floats, stream);
}

// (library-internal, dp_dev.h) the conv-split launch whose combine kernel waits for `before_combine` (an event of ANOTHER stream): dp_forward's third
// stream - the phore-phore conv writes the residual the combine reads while the split launch runs
int dpn_fused_split_after(const DpConvArgs* convs, int32_t n, int32_t n_nodes, float* part, int64_t part_floats, void* stream, void* before_combine) {
  if (!convs || n < 2 || n_nodes <= 0 || !part || !convs[0].path_split) return DP_E_BADARG;
  return fused_launch(convs, n, n_nodes, part, part_floats, stream, (hipEvent_t)before_combine);
}

static int fused_launch(const DpConvArgs* convs, int32_t n, int32_t n_nodes, float* part, int64_t part_floats, void* stream, hipEvent_t before_combine) {
  if (!convs || n < 1 || n > 3) return DP_E_BADARG;
  dpn::FusedArgs fa;
  memset(&fa, 0, sizeof(fa));
  int xmode = DP_EDGES_P2L;
  for (int i = 0; i < n; ++i) {
    const DpConvArgs* a = convs + i;
    const int rc = dpn::check_conv(a);
    if (rc) return rc;
    if (a->kind != convs[0].kind || a->B != convs[0].B || a->slices != convs[0].slices || a->scat_ptr != convs[0].scat_ptr ||
        a->max_scatter != convs[0].max_scatter || a->path_split != convs[0].path_split || a->precision != convs[0].precision)
      return DP_E_BADARG;
    if (a->mode == DP_EDGES_LLPAIR) {
      if (i != 0) return DP_E_UNSUPPORTED;
      fa.n_ll = 1;
    } else {
      if (fa.n_x == 0) xmode = a->mode;
      else if (a->mode != xmode) return DP_E_UNSUPPORTED;
      if (a->mode == DP_EDGES_L2P && fa.n_ll) return DP_E_UNSUPPORTED;
      if (fa.n_x == 2) return DP_E_UNSUPPORTED;
      ++fa.n_x;
    }
    fa.conv[i] = *a;
  }
  fa.lean = 1;
  for (int i = 0; i < n; ++i)
    if (!convs[i].no_tail) fa.lean = 0;
  if (!convs[0].out) return DP_E_BADARG;
  if (convs[0].precision == DP_PREC_BF16 || convs[0].precision == DP_PREC_F16X3) {        // the 16-bit kernels: no path split; conv split below
    for (int i = 0; i < n; ++i) fa.conv[i].path_split = 0;
  }
  if (part == nullptr) return dpn::launch_kind(convs[0].kind, fa, xmode, (hipStream_t)stream);
  fa.conv_split = 1;
  fa.part = part;
  fa.part_stride = n_nodes * convs[0].out_stride;
  fa.node0 = 0;
  if ((int64_t)fa.part_stride * n > part_floats) return DP_E_TOOLARGE;
  int rc = dpn::launch_kind(convs[0].kind, fa, xmode, (hipStream_t)stream);
  if (rc) return rc;
  static const int D_OUT[4] = {50, 80, 100, 100};
  dpn::CombineArgs c;
  c.part = part; c.n_conv = n; c.part_stride = fa.part_stride; c.n_nodes = n_nodes; c.d_out = D_OUT[convs[0].kind];
  c.out = convs[0].out; c.out_stride = convs[0].out_stride; c.accumulate = convs[0].accumulate;
  c.resid = convs[0].resid; c.resid_stride = convs[0].resid_stride; c.resid_width = convs[0].resid_width;
  for (int i = 0; i < 3; ++i) c.shift[i] = convs[i < n ? i : 0].bn_shift;
  const int total = n_nodes * c.d_out;
  if (before_combine && hipStreamWaitEvent((hipStream_t)stream, before_combine, 0) != hipSuccess) return (int)hipGetLastError();
  hipLaunchKernelGGL(dpn::conv_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c);
  return (int)hipGetLastError();
}

// node-side launch of one layer conv (kinds 0-3); called by dp_tp_conv when DpConvArgs.algo asks for it
int dp_tp_conv_node(const DpConvArgs* a, void* stream) {
  if (!a) return DP_E_BADARG;
  return dp_tp_conv_node_fused(a, 1, stream);
}

}  // extern "C"
