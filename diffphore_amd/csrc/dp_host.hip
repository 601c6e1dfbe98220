// dp_host.hip - the part of the C ABI a non-Python host needs besides the kernels: workspace sizing / carving, the parameter-blob
// layout, and the whole reverse-diffusion loop as one call (optionally captured into a hipGraph).
//
// Reference loop: sampling_phore, /root/reference/src/utils/sampling.py:204-255 (per step: set_time, model forward, SDE/ODE
// perturbation, modify_conformer per sample).  Here the S steps are enqueued back to back on one stream with no host
// synchronisation; diffusion times and pre-drawn noise are device-resident tables indexed by the step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/diffphore_amd.h"
#include "dp_dev.h"

namespace dph {

// ---------------------------------------------------------------------------------------------------------------------------
// workspace: one caller-owned allocation, carved here (single source of truth for the sizes dp_forward / dp_prepare assume)
// ---------------------------------------------------------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t off;
  bool dry;
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;                 // 256-byte aligned segments
    T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += (n ? n : 1) * sizeof(T);
    return p;
  }
};

static size_t carve(const DpBatch* b, void* base, DpWork* k, float** tr, float** rot, float** tor) {
  Carver c{reinterpret_cast<char*>(base), 0, base == nullptr};
  DpWork w;
  memset(&w, 0, sizeof(w));
  const size_t nl = b->n_lig, np = b->n_ph, npp = b->n_pp, nx = b->n_x, cap = b->ll_cap, B = b->B, nt = b->n_tor > 0 ? b->n_tor : 1;
  w.lig_x_emb = c.take<float>(nl * 20);
  w.ph_x_emb = c.take<float>(np * 20);
  w.pair_pre = c.take<float>(nl * 11 * 20);
  w.pair_fm = c.take<float>(nl * 11);
  w.pp_pre = c.take<float>(npp * 20);
  w.pp_sh = c.take<float>(npp * 9);
  w.tparams = c.take<float>(B * dp::TP_STRIDE);
  w.ll_rowptr = c.take<int32_t>(nl);
  w.ll_rowend = c.take<int32_t>(nl);
  w.ll_nbr = c.take<int32_t>(cap);
  w.ll_src = c.take<int32_t>(cap);
  w.ll_emb = c.take<float>(cap * 20);
  w.ll_sh = c.take<float>(cap * 9);
  w.ll_pair_rad = c.take<int32_t>((size_t)b->n_llp);
  w.ll_pair_bond = c.take<int32_t>((size_t)b->n_llp);
  w.pp_emb = c.take<float>(npp * 20);
  w.x_emb = c.take<float>(nx * 20);
  w.x_sh = c.take<float>(nx * 9);
  w.x_nsh = c.take<float>(nx * 9);
  w.x_ew = c.take<float>(nx);
  w.x_emb_bf = c.take<uint16_t>((nx + 1) * 64);
  w.ll_emb_bf = c.take<uint16_t>((cap + 1) * 64);
  w.h_l[0] = c.take<float>(nl * 100);
  w.h_l[1] = c.take<float>(nl * 100);
  w.h_p[0] = c.take<float>(np * 100);
  w.h_p[1] = c.take<float>(np * 100);
  w.c_emb = c.take<float>(nl * 20);
  w.c_sh = c.take<float>(nl * 9);
  w.gpred = c.take<float>(B * 12);
  w.t_rowptr = c.take<int32_t>(nt);
  w.t_rowend = c.take<int32_t>(nt);
  w.t_nbr = c.take<int32_t>(nt * dp::TOR_SLOTS);
  w.t_src = c.take<int32_t>(nt * dp::TOR_SLOTS);
  w.t_emb = c.take<float>(nt * dp::TOR_SLOTS * 20);
  w.t_sh = c.take<float>(nt * dp::TOR_SLOTS * 7);
  w.t_hb = c.take<float>(nt * 20);
  w.t_q = c.take<float>(nt * 40);
  w.status = c.take<int32_t>(B);
  w.z_noise = c.take<float>(B * 6 + nt);
  w.conv_part = c.take<float>(DP_CONV_PART_FLOATS);
  w.tp_table = c.take<float>((size_t)DP_TP_TABLE_STEPS * dp::TP_STRIDE);
  w.pp_tmp = c.take<float>(np * 100);
  // (a batch on one pharmacophore, DpBatch.ph_shared: room for one pose's edges is enough for the layer-0 conv that runs once)
  w.pp_msg = dp_pp_as_list(b) ? c.take<float>(npp * 100) : (b->ph_shared && B > 0 ? c.take<float>((npp / B + 1) * 100) : nullptr);
  w.xs_msg = c.take<float>((size_t)2 * (b->n_xs > 0 ? (size_t)b->n_xs : 0) * 100);
  float* tr_ = c.take<float>(B * 3);
  float* rot_ = c.take<float>(B * 3);
  float* tor_ = c.take<float>(nt);
  w.edge_bf16x6 = 1;                                  // default: the edge-side convs of an fp32 model on the bf16 pipe (fp32-grade split)
  w.node_f16 = 1;                                     // default: the node-side convs of an fp32 model as two fp16 limbs on the 16-bit pipe (fp32-grade)
  if (k) *k = w;
  if (tr) *tr = tr_;
  if (rot) *rot = rot_;
  if (tor) *tor = tor_;
  return (c.off + 255) & ~(size_t)255;
}

// ---------------------------------------------------------------------------------------------------------------------------
// parameter blob layout
// ---------------------------------------------------------------------------------------------------------------------------
struct SegList {
  DpSegment* out;
  int cap, n;
  int64_t off;
  void add(const char* name, int64_t numel) {
    if (out && n < cap) {
      snprintf(out[n].name, sizeof(out[n].name), "%s", name);
      out[n].offset = off;
      out[n].numel = numel;
    }
    ++n;
    off += (numel + 3) / 4 * 4;                       // 16-byte aligned segments
  }
};

static const char* const FAMS[6] = {"lig_conv", "ph_conv", "l2p_conv", "p2l_conv", "l2p_norm_conv", "p2l_norm_conv"};
static const int D_OUT_KIND[6] = {50, 80, 100, 100, 12, 40};

static int list_segments(int precision, DpSegment* out, int cap, int64_t* total) {
  if (dp_conv_w2p_floats(0, precision) < 0) return DP_E_BADARG;
  SegList s{out, cap, 0, 0};
  s.add("lig_emb", 199 * 20); s.add("lig_lin_w", 400); s.add("lig_lin_b", 20);
  s.add("boarder_emb", 200); s.add("boarder_lin_w", 20); s.add("boarder_lin_b", 20);
  s.add("ph_emb", 300); s.add("ph_lin_w", 440); s.add("ph_lin_b", 20);
  struct M { const char* n; int w0, b0, w3, b3; };
  static const M mlps[10] = {{"lig_edge", 880, 20, 400, 20}, {"ph_edge", 800, 20, 400, 20}, {"x_edge", 1460, 20, 400, 20},
                             {"cdt", 200, 10, 10, 1},        {"pdt", 11, 11, 11, 1},         {"pmt", 363, 11, 11, 1},
                             {"center_edge", 800, 20, 400, 20}, {"final_edge", 400, 20, 400, 20},
                             {"tr_final", 420, 20, 20, 1},   {"rot_final", 420, 20, 20, 1}};
  char nm[64];
  for (const M& m : mlps) {
    snprintf(nm, sizeof(nm), "%s.w0", m.n); s.add(nm, m.w0);
    snprintf(nm, sizeof(nm), "%s.b0", m.n); s.add(nm, m.b0);
    snprintf(nm, sizeof(nm), "%s.w3", m.n); s.add(nm, m.w3);
    snprintf(nm, sizeof(nm), "%s.b3", m.n); s.add(nm, m.b3);
  }
  s.add("tor_w0", 800); s.add("tor_w3", 20);
  for (int f = 0; f < 6; ++f)
    for (int l = 0; l < 4; ++l) {
      snprintf(nm, sizeof(nm), "%s.%d.w1p", FAMS[f], l); s.add(nm, dp_conv_w1p_floats(l));
      snprintf(nm, sizeof(nm), "%s.%d.w2p", FAMS[f], l); s.add(nm, dp_conv_w2p_floats(l, precision));
      snprintf(nm, sizeof(nm), "%s.%d.bn_scale", FAMS[f], l); s.add(nm, D_OUT_KIND[l]);
      snprintf(nm, sizeof(nm), "%s.%d.bn_shift", FAMS[f], l); s.add(nm, D_OUT_KIND[l]);
      if (f != 1 && precision == DP_PREC_F32) {       // every family but phore-phore (low degree: edge-side stays better): node-side layout as well
        snprintf(nm, sizeof(nm), "%s.%d.w1n", FAMS[f], l); s.add(nm, dp_conv_w1n_floats(l));
        snprintf(nm, sizeof(nm), "%s.%d.w2n", FAMS[f], l); s.add(nm, dp_conv_w2n_floats(l));
        // ... and once more as two-limb fp16 fragments (dp_pack_conv_weights_node_f16): DP_PREC_F16X3, what these convs run on by default
        snprintf(nm, sizeof(nm), "%s.%d.w1h", FAMS[f], l); s.add(nm, dp_conv_w1h_floats(l));
        snprintf(nm, sizeof(nm), "%s.%d.w2h", FAMS[f], l); s.add(nm, dp_conv_w2h_floats(l));
      }
      if (f != 1 && precision == DP_PREC_BF16) {      // the same families as bf16 MFMA fragments (dp_pack_conv_weights_node_bf16)
        snprintf(nm, sizeof(nm), "%s.%d.w1n", FAMS[f], l); s.add(nm, dp_conv_w1b_floats(l));
        snprintf(nm, sizeof(nm), "%s.%d.w2n", FAMS[f], l); s.add(nm, dp_conv_w2b_floats(l));
      }
      if ((((f == 1 && l < 3) || f == 0 || f >= 4) && precision == DP_PREC_F32) || (f >= 4 && precision == DP_PREC_BF16)) {     // phore-phore stays edge-side, ligand-ligand goes edge-side for
                                                                            // large ligands, the normals convs run their listed pairs edge-side
                                                                            // (DP_EDGES_LIST): the second layer for the 16-bit pipe as well (DpConvW.w2x)
        snprintf(nm, sizeof(nm), "%s.%d.w2x", FAMS[f], l); s.add(nm, dp_conv_w2p_floats(l, DP_PREC_F16X3));
        snprintf(nm, sizeof(nm), "%s.%d.bnx", FAMS[f], l); s.add(nm, D_OUT_KIND[l]);      // bn_scale * 2^-w of that image (DpConvW.bn_scale_x)
      }
    }
  static const char* const heads[2] = {"final_conv", "tor_conv"};
  for (int h = 0; h < 2; ++h) {
    const int kind = 4 + h;
    snprintf(nm, sizeof(nm), "%s.w1p", heads[h]); s.add(nm, dp_conv_w1p_floats(kind));
    snprintf(nm, sizeof(nm), "%s.w2p", heads[h]); s.add(nm, dp_conv_w2p_floats(kind, precision));
    snprintf(nm, sizeof(nm), "%s.bn_scale", heads[h]); s.add(nm, D_OUT_KIND[kind]);
    snprintf(nm, sizeof(nm), "%s.bn_shift", heads[h]); s.add(nm, D_OUT_KIND[kind]);
    if (h == 1 && precision == DP_PREC_F32) {
      snprintf(nm, sizeof(nm), "%s.w2x", heads[h]); s.add(nm, dp_conv_w2p_floats(kind, DP_PREC_F16X3));
      snprintf(nm, sizeof(nm), "%s.bnx", heads[h]); s.add(nm, D_OUT_KIND[kind]);
    }
  }
  s.add("so3_table", 1000); s.add("torus_table", 5001);
  s.add("lig_gauss_mu", 20); s.add("ph_gauss_mu", 20); s.add("x_gauss_mu", 20); s.add("center_gauss_mu", 20);
  s.add("emb_freq", 10);
  if (total) *total = s.off;
  return s.n;
}

}  // namespace dph

extern "C" {

size_t dp_work_bytes(const DpBatch* b) {
  if (!b || b->B <= 0) return 0;
  return dph::carve(b, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int dp_work_bind(const DpBatch* b, void* base, size_t bytes, DpWork* out, float** tr, float** rot, float** tor) {
  if (!b || b->B <= 0 || !base || !out) return DP_E_BADARG;
  if (((uintptr_t)base & 255) != 0) return DP_E_BADARG;
  if (bytes < dph::carve(b, nullptr, nullptr, nullptr, nullptr, nullptr)) return DP_E_TOOLARGE;
  dph::carve(b, base, out, tr, rot, tor);
  return DP_OK;
}

int dp_work_init(const DpBatch* b, DpWork* k, void* stream) {
  if (!b || !k || b->B <= 0) return DP_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  // node feature ping-pong buffers have row stride 100 for every layer: the unwritten tail of a row must read as zero
  for (int i = 0; i < 2; ++i) {
    if (hipMemsetAsync(k->h_l[i], 0, (size_t)b->n_lig * 100 * sizeof(float), st) != hipSuccess) return (int)hipGetLastError();
    if (hipMemsetAsync(k->h_p[i], 0, (size_t)b->n_ph * 100 * sizeof(float), st) != hipSuccess) return (int)hipGetLastError();
  }
  if (hipMemsetAsync(k->tparams, 0, (size_t)b->B * dp::TP_STRIDE * sizeof(float), st) != hipSuccess) return (int)hipGetLastError();
  if (hipMemsetAsync(k->status, 0, (size_t)b->B * sizeof(int32_t), st) != hipSuccess) return (int)hipGetLastError();
  return DP_OK;
}

#ifndef DP_SOURCE_HASH
#define DP_SOURCE_HASH "unknown"
#endif
#ifndef DP_NABL
#define DP_NABL 0
#endif
#ifndef DP_ABLATE
#define DP_ABLATE 0
#endif
#ifndef DP_BABL
#define DP_BABL 0
#endif
#ifndef DPF_ABL
#define DPF_ABL 0
#endif
#ifdef DP_SPLIT_LEAN
#define DP_SPLIT_LEAN_ON 1
#else
#define DP_SPLIT_LEAN_ON 0
#endif
// The switches are seen as THIS translation unit was compiled: tools/build_variant.sh therefore always recompiles dp_host.hip with the variant's
// flags (and a variant source hash), so that a timing / ablation build cannot pass for the product library.
int dp_build_info(DpBuildInfo* out) {
  if (!out) return DP_E_BADARG;
  memset(out, 0, sizeof(*out));
  out->abi_version = DP_ABI_VERSION;
  // (bf16 node-kernel ablations are reported as 100 + n, f16x3 node-kernel ablations as 200 + n, the lean path-split experiment as 300)
  out->nabl = DP_NABL ? DP_NABL : (DP_BABL ? 100 + DP_BABL : (DPF_ABL ? 200 + DPF_ABL : (DP_SPLIT_LEAN_ON ? 300 : 0)));
#ifdef DP_S0_TIMING       // dp_forward skips the sparse part of the normals convs (timing build, wrong results)
  if (!out->nabl) out->nabl = 400;
#endif
  out->ablate = DP_ABLATE;
#ifdef DP_PROF
  out->prof = 1;
#endif
  snprintf(out->source_hash, sizeof(out->source_hash), "%s", DP_SOURCE_HASH);
  return DP_OK;
}

int dp_weights_segments(int32_t precision, DpSegment* out, int32_t cap) { return dph::list_segments(precision, out, cap, nullptr); }

int64_t dp_weights_blob_floats(int32_t precision) {
  int64_t total = 0;
  return dph::list_segments(precision, nullptr, 0, &total) < 0 ? -1 : total;
}

int dp_weights_bind(const float* blob, int32_t precision, const DpModelScalars* sc, DpWeights* w) {
  if (!blob || !sc || !w) return DP_E_BADARG;
  DpSegment segs[384];
  const int n = dph::list_segments(precision, segs, 384, nullptr);
  if (n < 0 || n > 384) return DP_E_BADARG;
  memset(w, 0, sizeof(*w));
  auto find = [&](const char* name) -> const float* {
    for (int i = 0; i < n; ++i)
      if (strcmp(segs[i].name, name) == 0) return blob + segs[i].offset;
    return nullptr;
  };
  char nm[64];
  auto mlp = [&](const char* name, DpMlp& m) {
    snprintf(nm, sizeof(nm), "%s.w0", name); m.w0 = find(nm);
    snprintf(nm, sizeof(nm), "%s.b0", name); m.b0 = find(nm);
    snprintf(nm, sizeof(nm), "%s.w3", name); m.w3 = find(nm);
    snprintf(nm, sizeof(nm), "%s.b3", name); m.b3 = find(nm);
  };
  auto conv = [&](const char* name, DpConvW& c) {
    snprintf(nm, sizeof(nm), "%s.w1p", name); c.w1p = find(nm);
    snprintf(nm, sizeof(nm), "%s.w2p", name); c.w2p = find(nm);
    snprintf(nm, sizeof(nm), "%s.bn_scale", name); c.bn_scale = find(nm);
    snprintf(nm, sizeof(nm), "%s.bn_shift", name); c.bn_shift = find(nm);
    snprintf(nm, sizeof(nm), "%s.w1n", name); c.w1n = find(nm);
    snprintf(nm, sizeof(nm), "%s.w2n", name); c.w2n = find(nm);
    snprintf(nm, sizeof(nm), "%s.w2x", name); c.w2x = find(nm);
    snprintf(nm, sizeof(nm), "%s.bnx", name); c.bn_scale_x = find(nm);
    snprintf(nm, sizeof(nm), "%s.w1h", name); c.w1h = find(nm);
    snprintf(nm, sizeof(nm), "%s.w2h", name); c.w2h = find(nm);
  };
  w->lig_emb = find("lig_emb"); w->lig_lin_w = find("lig_lin_w"); w->lig_lin_b = find("lig_lin_b");
  w->boarder_emb = find("boarder_emb"); w->boarder_lin_w = find("boarder_lin_w"); w->boarder_lin_b = find("boarder_lin_b");
  w->ph_emb = find("ph_emb"); w->ph_lin_w = find("ph_lin_w"); w->ph_lin_b = find("ph_lin_b");
  mlp("lig_edge", w->lig_edge); mlp("ph_edge", w->ph_edge); mlp("x_edge", w->x_edge);
  mlp("cdt", w->cdt); mlp("pdt", w->pdt); mlp("pmt", w->pmt);
  mlp("center_edge", w->center_edge); mlp("final_edge", w->final_edge); mlp("tr_final", w->tr_final); mlp("rot_final", w->rot_final);
  w->tor_w0 = find("tor_w0"); w->tor_w3 = find("tor_w3");
  DpConvW* fams[6] = {w->lig_conv, w->ph_conv, w->l2p_conv, w->p2l_conv, w->l2p_norm_conv, w->p2l_norm_conv};
  for (int f = 0; f < 6; ++f)
    for (int l = 0; l < 4; ++l) {
      char base[48];
      snprintf(base, sizeof(base), "%s.%d", dph::FAMS[f], l);
      conv(base, fams[f][l]);
    }
  conv("final_conv", w->final_conv);
  conv("tor_conv", w->tor_conv);
  w->so3_table = find("so3_table"); w->torus_table = find("torus_table");
  w->lig_gauss_mu = find("lig_gauss_mu"); w->ph_gauss_mu = find("ph_gauss_mu"); w->x_gauss_mu = find("x_gauss_mu");
  w->center_gauss_mu = find("center_gauss_mu"); w->emb_freq = find("emb_freq");
  w->lig_gauss_coeff = sc->lig_gauss_coeff; w->ph_gauss_coeff = sc->ph_gauss_coeff; w->x_gauss_coeff = sc->x_gauss_coeff;
  w->center_gauss_coeff = sc->center_gauss_coeff; w->scaler = sc->scaler;
  w->tr_smin = sc->tr_smin; w->tr_smax = sc->tr_smax; w->rot_smin = sc->rot_smin; w->rot_smax = sc->rot_smax;
  w->tor_smin = sc->tor_smin; w->tor_smax = sc->tor_smax; w->emb_scale = sc->emb_scale;
  w->precision = precision;
  return DP_OK;
}

// e3nn BatchNorm(eval) of a conv's output irreps as a per-channel affine map: HOST pointers
int dp_pack_bn(int32_t kind, const float* running_mean, const float* running_var, const float* weight, const float* bias, float eps,
               float* scale, float* shift) {
  if (kind < 0 || kind > 5 || !running_var || !weight || !scale || !shift) return DP_E_BADARG;
  // out irreps per kind as (mul, dim, is 0e): seq1 / seq2 / seq3 / seq3 / 2x1o+2x1e / 20x0o+20x0e
  struct Ir { int mul, d, scalar_even; };
  static const Ir IRS[6][4] = {{{20, 1, 1}, {10, 3, 0}, {0, 0, 0}, {0, 0, 0}},   {{20, 1, 1}, {10, 3, 0}, {10, 3, 0}, {0, 0, 0}},
                               {{20, 1, 1}, {10, 3, 0}, {10, 3, 0}, {20, 1, 0}}, {{20, 1, 1}, {10, 3, 0}, {10, 3, 0}, {20, 1, 0}},
                               {{2, 3, 0}, {2, 3, 0}, {0, 0, 0}, {0, 0, 0}},     {{20, 1, 0}, {20, 1, 1}, {0, 0, 0}, {0, 0, 0}}};
  int iv = 0, im = 0, o = 0;
  for (int s = 0; s < 4; ++s) {
    const Ir& ir = IRS[kind][s];
    for (int u = 0; u < ir.mul; ++u) {
      const double sc = (double)weight[iv + u] / __builtin_sqrt((double)running_var[iv + u] + (double)eps);
      double sh = 0.0;
      if (ir.scalar_even) {
        if (!running_mean || !bias) return DP_E_BADARG;
        sh = (double)bias[im + u] - (double)running_mean[im + u] * sc;
      }
      for (int c = 0; c < ir.d; ++c) { scale[o] = (float)sc; shift[o] = (float)sh; ++o; }
    }
    iv += ir.mul;
    if (ir.scalar_even) im += ir.mul;
  }
  return DP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// whole reverse-diffusion loop
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int dpg_time_table(const float* t, int32_t S, const DpWeights* w, float* table, void* stream);      // dp_graph.hip
int dp_sample(const DpBatch* b, const DpWeights* w, DpWork* k, const DpSchedule* s, float* tr, float* rot, float* tor, void* stream) {
  if (!b || !w || !k || !s || s->steps <= 0 || !s->sp || !s->t || !tr || !rot) return DP_E_BADARG;
  if (s->noise_mode != DP_NOISE_TABLES && s->noise_mode != DP_NOISE_SEED) return DP_E_BADARG;
  DpWork ku = *k;
  ku.t_uniform = 1;                 // every pose gets the step's times (set_time_phore, sampling.py:218) ...
  // the schedule is known: the time parameters of all its steps in one launch, a step then only hands its row to the poses
  const bool tabled = k->tp_table != nullptr && s->steps <= DP_TP_TABLE_STEPS;
  if (tabled) {
    const int rt = dpg_time_table(s->t, s->steps, w, k->tp_table, stream);
    if (rt) return rt;
  }
  for (int i = 0; i < s->steps; ++i) {
    const DpStepParams& sp = s->sp[i];
    ku.t_row = s->t + 3 * i;        // ... by the first kernel of dp_forward (DpWork.t_row) instead of a launch of its own
    ku.tp_row = tabled ? k->tp_table + (size_t)i * dp::TP_STRIDE : nullptr;
    // a noise table is consumed only where its coefficient is non-zero (ODE, no_random, no_final_step_noise: sampling.py:226-244)
    const float *z_tr, *z_rot, *z_tor;
    if (s->noise_mode == DP_NOISE_SEED) {
      if (!k->z_noise) return DP_E_BADARG;
      float* zt = sp.tr_c2 != 0.f ? k->z_noise : nullptr;
      float* zr = sp.rot_c2 != 0.f ? k->z_noise + (size_t)b->B * 3 : nullptr;
      float* zo = (sp.tor_c2 != 0.f && !sp.no_torsion && b->n_tor > 0) ? k->z_noise + (size_t)b->B * 6 : nullptr;
      const int rn = dp_noise_normals(s->seed, i, b, s->pose_uid, s->uid_base, zt, zr, zo, stream);
      if (rn) return rn;
      z_tr = zt; z_rot = zr; z_tor = zo;
    } else {
      z_tr = (s->z_tr && sp.tr_c2 != 0.f) ? s->z_tr + (size_t)i * b->B * 3 : nullptr;
      z_rot = (s->z_rot && sp.rot_c2 != 0.f) ? s->z_rot + (size_t)i * b->B * 3 : nullptr;
      z_tor = (s->z_tor && sp.tor_c2 != 0.f && !sp.no_torsion) ? s->z_tor + (size_t)i * b->n_tor : nullptr;
    }
    const int rc = dp_sample_step(b, w, &ku, &sp, tr, rot, tor, z_tr, z_rot, z_tor, stream);
    if (rc) return rc;
  }
  return (int)hipGetLastError();
}

int dp_sample_graph_create(const DpBatch* b, const DpWeights* w, DpWork* k, const DpSchedule* s, float* tr, float* rot, float* tor,
                           void** graph_exec) {
  if (!graph_exec) return DP_E_BADARG;
  *graph_exec = nullptr;
  hipStream_t cs;
  if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return (int)hipGetLastError();
  int rc = (int)hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  hipGraph_t g = nullptr;
  if (rc == 0) {
    rc = dp_sample(b, w, k, s, tr, rot, tor, cs);
    const int rc2 = (int)hipStreamEndCapture(cs, &g);
    if (rc == 0) rc = rc2;
  }
  hipGraphExec_t ge = nullptr;
  if (rc == 0) rc = (int)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  if (g) (void)hipGraphDestroy(g);
  (void)hipStreamDestroy(cs);
  if (rc == 0) *graph_exec = ge;
  return rc;
}

// second stream + fork / join events for DpWork.aux_stream / ev_fork / ev_join, created by THIS library's HIP runtime (a host binding
// that dlopens its own libamdhip64 could hand over handles of a different runtime)
int dp_aux_create(void** aux_stream, void** ev_fork, void** ev_join) {
  if (!aux_stream || !ev_fork || !ev_join) return DP_E_BADARG;
  *aux_stream = *ev_fork = *ev_join = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  int rc = (int)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (rc == 0) rc = (int)hipEventCreateWithFlags(&a, hipEventDisableTiming);
  if (rc == 0) rc = (int)hipEventCreateWithFlags(&b, hipEventDisableTiming);
  if (rc != 0) {
    if (b) (void)hipEventDestroy(b);
    if (a) (void)hipEventDestroy(a);
    if (s) (void)hipStreamDestroy(s);
    return rc;
  }
  *aux_stream = s; *ev_fork = a; *ev_join = b;
  return DP_OK;
}

int dp_aux_destroy(void* aux_stream, void* ev_fork, void* ev_join) {
  int rc = 0;
  if (ev_fork) { const int r = (int)hipEventDestroy((hipEvent_t)ev_fork); if (r && !rc) rc = r; }
  if (ev_join) { const int r = (int)hipEventDestroy((hipEvent_t)ev_join); if (r && !rc) rc = r; }
  if (aux_stream) { const int r = (int)hipStreamDestroy((hipStream_t)aux_stream); if (r && !rc) rc = r; }
  return rc;
}

int dp_sample_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return DP_E_BADARG;
  return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
}

int dp_sample_graph_destroy(void* graph_exec) {
  if (!graph_exec) return DP_OK;
  return (int)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

}  // extern "C"
