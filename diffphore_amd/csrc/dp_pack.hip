// dp_pack.hip - HOST code: the collate step of the sampling loop in native code.
//
// The reference builds a PyG batch of the pose list once per chunk and step (`Batch.from_data_list` via DataLoader,
// /root/reference/src/utils/sampling.py:210-215).  Here the pose list is packed ONCE per device batch into the struct-of-arrays image
// dp_forward consumes (DpBatch: CSR offsets per pose + concatenated static arrays + the diffusion state).  This file does that packing
// from per-complex views of the caller's tensors: one C call concatenates, shifts indices, converts one-hots to indices, sorts the
// pharmacophore edges by source and validates - into ONE host block (the caller uploads it with one copy) - instead of ~200 interpreter
// operations per pose.  The samples of one complex share one DpGraphView (inference.py:183 deep-copies the graph N times; its static
// arrays are identical), only `pos` / `norm` are per pose.  Pure host arithmetic: callable without a GPU (tests/test_pack_cpu.py).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/diffphore_amd.h"

namespace dpk {

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Sizes { int64_t n_lig, n_ph, n_bond, n_tor, n_pp, n_x, ll_cap, n_llp, n_mask, n_xs; int32_t max_nl, max_np, max_t, ph_shared; };

// the pharmacophore of view b is the one of view a: same nodes (features, positions, normals, types) and the same edges in the same order
static bool same_phore(const DpGraphView& a, const DpGraphView& b) {
  if (&a == &b) return true;
  if (a.n_phore != b.n_phore || a.n_pp != b.n_pp) return false;
  if (!a.ph_x || !b.ph_x || !a.ph_pos || !b.ph_pos || !a.ph_norm || !b.ph_norm || !a.ph_type || !b.ph_type) return false;
  if (a.n_pp > 0 && (!a.pp_index || !b.pp_index)) return false;
  const size_t np = (size_t)a.n_phore;
  auto eq = [](const void* x, const void* y, size_t bytes) { return x == y || memcmp(x, y, bytes) == 0; };
  return eq(a.ph_x, b.ph_x, np * 5 * 4) && eq(a.ph_pos, b.ph_pos, np * 3 * 4) && eq(a.ph_norm, b.ph_norm, np * 3 * 4) && eq(a.ph_type, b.ph_type, np * 11 * 4) &&
         (a.n_pp == 0 || eq(a.pp_index, b.pp_index, (size_t)a.n_pp * 2 * 8));
}

constexpr int EX_TYPE = 10;      // the exclusion-sphere column of the one-hot `phoretype` (process_pharmacophore.py:28-38: the last of 11)
// type of pharmacophore node j: the first column that is exactly 1 (dp_pack_fill validates the one-hot), -1 if none
static inline int type_of(const float* t) {
  for (int k = 0; k < 11; ++k)
    if (t[k] == 1.f) return k;
  return -1;
}
// number of cross pairs of one complex with agreement != 0 (DpBatch.xs_*): node j is not an exclusion sphere and phorefp[i, type(j)] != 0
static int64_t count_xs(const DpGraphView& v) {
  int cnt[11] = {0};
  for (int i = 0; i < v.n_atoms; ++i)
    for (int t = 0; t < 11; ++t) cnt[t] += v.phorefp[(size_t)i * 11 + t] != 0.f ? 1 : 0;
  int64_t n = 0;
  for (int j = 0; j < v.n_phore; ++j) {
    const int ty = type_of(v.ph_type + (size_t)j * 11);
    if (ty >= 0 && ty != EX_TYPE) n += cnt[ty];
  }
  return n;
}

static int sizes_of(int32_t n, const DpPoseRef* poses, Sizes* s) {
  memset(s, 0, sizeof(*s));
  const DpGraphView* last = nullptr;               // the samples of one complex share a view and follow each other: counted once
  int64_t last_xs = 0;
  bool shared = n > 1;                             // DpBatch.ph_shared: every pose carries pose 0's pharmacophore (byte comparison; -0.0 vs 0.0 counts as different)
  for (int i = 0; i < n; ++i) {
    const DpGraphView* v = poses[i].view;
    if (!v || !poses[i].pos || !poses[i].norm) return DP_E_BADARG;
    const int64_t nl = v->n_atoms, np = v->n_phore, nb = v->n_bond_edges, nt = v->n_tor;
    if (nl < 1 || np < 1 || nb < 0 || nt < 0 || v->n_pp < 0) return DP_E_BADARG;
    if (nl > 256 || np > 256 || nt > 128) return DP_E_TOOLARGE;
    s->n_lig += nl; s->n_ph += np; s->n_bond += nb; s->n_tor += nt; s->n_pp += v->n_pp; s->n_x += nl * np;
    s->ll_cap += nb + nl * std::min<int64_t>(std::max<int64_t>(nl - 1, 0), 33);
    s->n_llp += nl * nl; s->n_mask += nt * nl;
    if (v != last) {
      if (!v->phorefp || !v->ph_type) return DP_E_BADARG;
      last = v; last_xs = count_xs(*v);
      if (shared && i > 0) shared = same_phore(*poses[0].view, *v);
    }
    s->n_xs += last_xs;
    s->max_nl = std::max<int32_t>(s->max_nl, (int32_t)nl); s->max_np = std::max<int32_t>(s->max_np, (int32_t)np);
    s->max_t = std::max<int32_t>(s->max_t, (int32_t)nt);
  }
  if (std::max(s->n_x, s->ll_cap) * 20 >= ((int64_t)1 << 31)) return DP_E_TOOLARGE;      // int32 indexing of the per-edge arrays
  s->ph_shared = shared ? 1 : 0;
  return DP_OK;
}

// byte offsets of every array inside the block, in DpPackLayout order
static size_t layout_of(int32_t n, const Sizes& s, DpPackLayout* L) {
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = al(off + std::max<size_t>(bytes, 4)); return (int64_t)o; };
  const size_t P = (size_t)(n + 1) * 4;
  L->lig_ptr = take(P); L->ph_ptr = take(P); L->bond_ptr = take(P); L->tor_ptr = take(P); L->pp_ptr = take(P); L->x_ptr = take(P);
  L->ll_ptr = take(P); L->mask_ptr = take(P); L->pose_ptr = take(P); L->llp_ptr = take(P);
  L->lig_x = take((size_t)s.n_lig * 16 * 4);
  L->bond_src = take((size_t)s.n_bond * 4); L->bond_dst = take((size_t)s.n_bond * 4); L->bond_type = take((size_t)s.n_bond * 4);
  L->tor_u = take((size_t)s.n_tor * 4); L->tor_v = take((size_t)s.n_tor * 4);
  L->mask_rotate = take((size_t)s.n_mask);
  L->phorefp = take((size_t)s.n_lig * 11 * 4); L->norm_angle1 = take((size_t)s.n_lig * 11 * 4); L->norm_angle2 = take((size_t)s.n_lig * 11 * 4);
  L->ph_pos = take((size_t)s.n_ph * 3 * 4); L->ph_norm = take((size_t)s.n_ph * 3 * 4); L->ph_x = take((size_t)s.n_ph * 3 * 4);
  L->ph_scalar = take((size_t)s.n_ph * 2 * 4); L->ph_type = take((size_t)s.n_ph * 4);
  L->pp_src = take((size_t)s.n_pp * 4); L->pp_dst = take((size_t)s.n_pp * 4); L->pp_rowptr = take((size_t)s.n_ph * 4); L->pp_rowend = take((size_t)s.n_ph * 4);
  L->pos = take((size_t)s.n_lig * 3 * 4); L->norm = take((size_t)s.n_lig * 33 * 4);
  L->xs_src = take((size_t)s.n_xs * 4); L->xs_nbr = take((size_t)s.n_xs * 4); L->xs_feat = take((size_t)s.n_xs * 4); L->xs_rowptr = take((size_t)(s.n_lig + 1) * 4);
  L->xt_src = take((size_t)s.n_xs * 4); L->xt_nbr = take((size_t)s.n_xs * 4); L->xt_feat = take((size_t)s.n_xs * 4); L->xt_rowptr = take((size_t)(s.n_ph + 1) * 4);
  L->bytes = (int64_t)off;
  return off;
}

}  // namespace dpk

extern "C" {

int dp_pack_query(int32_t n, const DpPoseRef* poses, DpBatch* totals, DpPackLayout* layout) {
  if (n <= 0 || !poses || !totals || !layout) return DP_E_BADARG;
  dpk::Sizes s;
  const int rc = dpk::sizes_of(n, poses, &s);
  if (rc) return rc;
  memset(totals, 0, sizeof(*totals));
  totals->B = n;
  totals->n_lig = (int32_t)s.n_lig; totals->n_ph = (int32_t)s.n_ph; totals->n_bond = (int32_t)s.n_bond; totals->n_tor = (int32_t)s.n_tor;
  totals->n_pp = (int32_t)s.n_pp; totals->n_x = (int32_t)s.n_x; totals->ll_cap = (int32_t)s.ll_cap; totals->n_llp = (int32_t)s.n_llp;
  totals->max_nl = s.max_nl; totals->max_np = s.max_np; totals->max_t = s.max_t;
  totals->n_xs = (int32_t)s.n_xs;
  totals->ph_shared = s.ph_shared;
  memset(layout, 0, sizeof(*layout));
  dpk::layout_of(n, s, layout);
  return DP_OK;
}

// Fill `block` (HOST, >= layout->bytes, 256-byte aligned is enough) with the packed batch.  *flags: bit 0 = some ligand lists the same ordered bond
// twice (no pair tables: DpBatch.llp_ptr = NULL, n_llp = 0), bit 1 = a pose ends on a short ligand pass (N_l mod 32 in 1..16), bit 2 = the same
// for N_p.  Returns DP_E_BADARG for malformed views (bond_attr / phoretype not one-hot, indices out of range, mask_rotate rows missing).
int dp_pack_fill(int32_t n, const DpPoseRef* poses, const DpPackLayout* L, void* block, int32_t* flags) {
  if (n <= 0 || !poses || !L || !block || !flags) return DP_E_BADARG;
  char* base = (char*)block;
  auto I = [&](int64_t off) { return (int32_t*)(base + off); };
  auto F = [&](int64_t off) { return (float*)(base + off); };
  int32_t *lig_ptr = I(L->lig_ptr), *ph_ptr = I(L->ph_ptr), *bond_ptr = I(L->bond_ptr), *tor_ptr = I(L->tor_ptr), *pp_ptr = I(L->pp_ptr),
          *x_ptr = I(L->x_ptr), *ll_ptr = I(L->ll_ptr), *mask_ptr = I(L->mask_ptr), *pose_ptr = I(L->pose_ptr), *llp_ptr = I(L->llp_ptr);
  lig_ptr[0] = ph_ptr[0] = bond_ptr[0] = tor_ptr[0] = pp_ptr[0] = x_ptr[0] = ll_ptr[0] = mask_ptr[0] = pose_ptr[0] = llp_ptr[0] = 0;
  int32_t fl = 0;
  int32_t *xs_src = I(L->xs_src), *xs_nbr = I(L->xs_nbr), *xs_feat = I(L->xs_feat), *xs_row = I(L->xs_rowptr);
  int32_t *xt_src = I(L->xt_src), *xt_nbr = I(L->xt_nbr), *xt_feat = I(L->xt_feat), *xt_row = I(L->xt_rowptr);
  int32_t n_xs = 0, n_xt = 0;
  const int64_t xs_cap = (L->xs_nbr - L->xs_src) / 4;      // room the layout reserved (dp_pack_query counted the pairs of the same views)
  std::vector<int32_t> deg, cur;
  std::vector<int64_t> keys;
  for (int i = 0; i < n; ++i) {
    const DpGraphView& v = *poses[i].view;
    const int nl = v.n_atoms, np = v.n_phore, nb = v.n_bond_edges, nt = v.n_tor, npp = v.n_pp;
    const int l0 = lig_ptr[i], p0 = ph_ptr[i], b0 = bond_ptr[i], t0 = tor_ptr[i], e0 = pp_ptr[i], m0 = mask_ptr[i];
    lig_ptr[i + 1] = l0 + nl; ph_ptr[i + 1] = p0 + np; bond_ptr[i + 1] = b0 + nb; tor_ptr[i + 1] = t0 + nt; pp_ptr[i + 1] = e0 + npp;
    x_ptr[i + 1] = x_ptr[i] + nl * np;
    ll_ptr[i + 1] = ll_ptr[i] + nb + nl * std::min(std::max(nl - 1, 0), 33);
    mask_ptr[i + 1] = m0 + nt * nl; pose_ptr[i + 1] = i + 1; llp_ptr[i + 1] = llp_ptr[i] + nl * nl;
    if (((nl - 1) % 32) < 16) fl |= 2;
    if (((np - 1) % 32) < 16) fl |= 4;
    // ligand, static
    int32_t* lx = I(L->lig_x) + (size_t)l0 * 16;
    for (int k = 0; k < nl * 16; ++k) lx[k] = (int32_t)v.lig_x[k];
    memcpy(F(L->phorefp) + (size_t)l0 * 11, v.phorefp, (size_t)nl * 11 * 4);
    memcpy(F(L->norm_angle1) + (size_t)l0 * 11, v.norm_angle1, (size_t)nl * 11 * 4);
    memcpy(F(L->norm_angle2) + (size_t)l0 * 11, v.norm_angle2, (size_t)nl * 11 * 4);
    int32_t *bs = I(L->bond_src) + b0, *bd = I(L->bond_dst) + b0, *bt = I(L->bond_type) + b0, *tu = I(L->tor_u) + t0, *tv = I(L->tor_v) + t0;
    keys.resize((size_t)nb);
    int tcount = 0;
    for (int e = 0; e < nb; ++e) {
      const int64_t s = v.bond_index[e], d = v.bond_index[nb + e];
      if (s < 0 || s >= nl || d < 0 || d >= nl) return DP_E_BADARG;
      bs[e] = l0 + (int32_t)s; bd[e] = l0 + (int32_t)d;
      const float* a = v.bond_attr + (size_t)e * 4;
      int arg = -1, ones = 0;
      float sum = 0.f;
      for (int k = 0; k < 4; ++k) { sum += a[k]; if (a[k] == 1.f) { ++ones; if (arg < 0) arg = k; } }
      if (ones != 1 || sum != 1.f) return DP_E_BADARG;                       // bond edge_attr must be one-hot over 4 bond types
      bt[e] = arg;
      keys[(size_t)e] = s * 1024 + d;
      if (v.edge_mask[e]) {
        if (tcount >= nt) return DP_E_BADARG;
        tu[tcount] = l0 + (int32_t)s; tv[tcount] = l0 + (int32_t)d; ++tcount;
      }
    }
    if (tcount != nt) return DP_E_BADARG;
    std::sort(keys.begin(), keys.end());
    for (size_t e = 1; e < keys.size(); ++e)
      if (keys[e] == keys[e - 1]) { fl |= 1; break; }
    if (nt > 0) {
      if (!v.mask_rotate) return DP_E_BADARG;
      memcpy(base + L->mask_rotate + m0, v.mask_rotate, (size_t)nt * nl);
    }
    // pharmacophore, static
    memcpy(F(L->ph_pos) + (size_t)p0 * 3, v.ph_pos, (size_t)np * 3 * 4);
    memcpy(F(L->ph_norm) + (size_t)p0 * 3, v.ph_norm, (size_t)np * 3 * 4);
    int32_t *phx = I(L->ph_x) + (size_t)p0 * 3, *pty = I(L->ph_type) + p0;
    float* psc = F(L->ph_scalar) + (size_t)p0 * 2;
    for (int j = 0; j < np; ++j) {
      const float* x = v.ph_x + (size_t)j * 5;
      phx[j * 3] = (int32_t)(int64_t)x[0]; phx[j * 3 + 1] = (int32_t)(int64_t)x[1]; phx[j * 3 + 2] = (int32_t)(int64_t)x[2];
      psc[j * 2] = x[3]; psc[j * 2 + 1] = x[4];
      const float* t = v.ph_type + (size_t)j * 11;
      int arg = -1, ones = 0;
      float sum = 0.f;
      for (int k = 0; k < 11; ++k) { sum += t[k]; if (t[k] == 1.f) { ++ones; if (arg < 0) arg = k; } }
      if (ones != 1 || sum != 1.f) return DP_E_BADARG;                       // data['phore'].phoretype must be one-hot (pdbbind_phore.py:1173)
      pty[j] = arg;
    }
    // pharmacophore edges: stable counting sort by source; CSR rows over the pose's nodes
    deg.assign((size_t)np, 0);
    for (int e = 0; e < npp; ++e) {
      const int64_t s = v.pp_index[e], d = v.pp_index[npp + e];
      if (s < 0 || s >= np || d < 0 || d >= np) return DP_E_BADARG;
      ++deg[(size_t)s];
    }
    int32_t *rp = I(L->pp_rowptr) + p0, *re = I(L->pp_rowend) + p0, *ps = I(L->pp_src) + e0, *pd = I(L->pp_dst) + e0;
    cur.resize((size_t)np);
    int run = 0;
    for (int j = 0; j < np; ++j) { rp[j] = e0 + run; cur[(size_t)j] = run; run += deg[(size_t)j]; re[j] = e0 + run; }
    for (int e = 0; e < npp; ++e) {
      const int s = (int)v.pp_index[e], d = (int)v.pp_index[npp + e];
      const int slot = cur[(size_t)s]++;
      ps[slot] = p0 + s; pd[slot] = p0 + d;
    }
    // cross pairs with agreement != 0, by atom and by node (DpBatch.xs_* / xt_*)
    for (int a = 0; a < nl; ++a) {
      xs_row[l0 + a] = n_xs;
      const float* fp = v.phorefp + (size_t)a * 11;
      for (int j = 0; j < np; ++j) {
        const int ty = pty[j];
        if (ty != dpk::EX_TYPE && fp[ty] != 0.f) {
          if (n_xs >= xs_cap) return DP_E_BADARG;
          xs_src[n_xs] = l0 + a; xs_nbr[n_xs] = p0 + j; xs_feat[n_xs] = x_ptr[i] + a * np + j; ++n_xs;
        }
      }
    }
    for (int j = 0; j < np; ++j) {
      xt_row[p0 + j] = n_xt;
      const int ty = pty[j];
      if (ty == dpk::EX_TYPE) continue;
      for (int a = 0; a < nl; ++a)
        if (v.phorefp[(size_t)a * 11 + ty] != 0.f) {
          if (n_xt >= xs_cap) return DP_E_BADARG;
          xt_src[n_xt] = p0 + j; xt_nbr[n_xt] = l0 + a; xt_feat[n_xt] = x_ptr[i] + a * np + j; ++n_xt;
        }
    }
    // diffusion state
    memcpy(F(L->pos) + (size_t)l0 * 3, poses[i].pos, (size_t)nl * 3 * 4);
    memcpy(F(L->norm) + (size_t)l0 * 33, poses[i].norm, (size_t)nl * 33 * 4);
  }
  xs_row[lig_ptr[n]] = n_xs;
  xt_row[ph_ptr[n]] = n_xt;
  if (n_xs != n_xt) return DP_E_BADARG;
  *flags = fl;
  return DP_OK;
}

// point the DpBatch fields at a block (device or host: `base` is just an address) holding the image dp_pack_fill wrote; `t` is left alone
int dp_pack_bind(const DpPackLayout* L, void* base, int32_t flags, DpBatch* b) {
  if (!L || !base || !b) return DP_E_BADARG;
  char* p = (char*)base;
#define DP_I(f) b->f = (const int32_t*)(p + L->f)
#define DP_F(f) b->f = (const float*)(p + L->f)
  DP_I(lig_ptr); DP_I(ph_ptr); DP_I(bond_ptr); DP_I(tor_ptr); DP_I(pp_ptr); DP_I(x_ptr); DP_I(ll_ptr); DP_I(mask_ptr); DP_I(pose_ptr); DP_I(llp_ptr);
  DP_I(lig_x); DP_I(bond_src); DP_I(bond_dst); DP_I(bond_type); DP_I(tor_u); DP_I(tor_v);
  b->mask_rotate = (const uint8_t*)(p + L->mask_rotate);
  DP_F(phorefp); DP_F(norm_angle1); DP_F(norm_angle2); DP_F(ph_pos); DP_F(ph_norm); DP_I(ph_x); DP_F(ph_scalar); DP_I(ph_type);
  DP_I(pp_src); DP_I(pp_dst); DP_I(pp_rowptr); DP_I(pp_rowend);
  DP_I(xs_src); DP_I(xs_nbr); DP_I(xs_feat); DP_I(xs_rowptr); DP_I(xt_src); DP_I(xt_nbr); DP_I(xt_feat); DP_I(xt_rowptr);
#undef DP_I
#undef DP_F
  b->pos = (float*)(p + L->pos);
  b->norm = (float*)(p + L->norm);
  if (flags & 1) { b->llp_ptr = nullptr; b->n_llp = 0; }
  b->no_tail_nl = (flags & 2) ? 0 : 1;
  b->no_tail_np = (flags & 4) ? 0 : 1;
  return DP_OK;
}

}  // extern "C"
