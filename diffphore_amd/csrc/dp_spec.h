// dp_spec.h - compile-time irreps / tensor-product path tables of the released DiffPhore configuration.
//
// Reference: irreps sequence /root/reference/src/models/score_model_phore.py:586-599; head convs :256-285.
// Path enumeration follows e3nn FullyConnectedTensorProduct (in-slot outer, sh-slot middle, out-slot inner),
// 'uvw' weights [mul_in, mul_out] row-major per path, coefficient sqrt((2 l_out+1) / sum_{paths->slot} mul_in).
// Feature layout: irreps in order, each [mul, 2l+1] mul-major.
#pragma once
#include <stdint.h>

namespace dp {

enum ZType : int {
  Z000 = 0,  // scalar x scalar -> scalar          z    = x * s0
  Z011 = 1,  // scalar x vector -> vector          z[k] = x * v[k]
  Z101 = 2,  // vector x scalar -> vector          z[k] = x[k] * s0
  Z110 = 3,  // vector . vector -> scalar          z    = x . v
  Z111 = 4,  // vector x vector -> vector          z    = x cross v
  Z121 = 5   // vector x l=2    -> vector          z[k] = sum_i x[i] M[i][k],  M from the five l=2 components
};

struct PathC {
  int ztype;
  int mul_in, mul_out;
  int in_off;    // offset of the in-slot in the D_in feature vector
  int out_off;   // offset of the out-slot in the D_out feature vector
  int sh_off;    // offset of the sh operand in the per-edge sh vector
  int w_off;     // offset of the [mul_in, mul_out] block in the reference weight vector
  int fan_in;    // sum of mul_in over all paths into the same out slot
};

constexpr int d_in_of(int z) { return (z == Z000 || z == Z011) ? 1 : 3; }
constexpr int d_out_of(int z) { return (z == Z000 || z == Z110) ? 1 : 3; }
// |C|-normalisation constant that is folded into the packed weights together with the path coefficient
constexpr double cg_const(int z) {
  return z == Z000 ? 1.0 : (z == Z011 || z == Z101 || z == Z110) ? 0.57735026918962576451  /* 1/sqrt3 */
         : z == Z111 ? 0.40824829046386301637 /* 1/sqrt6 */ : 0.31622776601683793320 /* 1/sqrt10 */;
}

// in-slot offsets: 0e@0 (20), 1o@20 (10x3), 1e@50 (10x3), 0o@80 (20).  sh: s0@0, l1@1, l2@4.
template <int KIND> struct Spec;

// L0: 20x0e -> 20x0e + 10x1o
template <> struct Spec<0> {
  static constexpr int D_IN = 20, D_OUT = 50, N_ATTR = 60, HID = 60, NPATH = 2, NUMEL = 600, SH_DIM = 9;
  static constexpr PathC P[NPATH] = {
      {Z000, 20, 20, 0, 0, 0, 0, 20},
      {Z011, 20, 10, 0, 20, 1, 400, 20},
  };
};
// L1: 20x0e+10x1o -> 20x0e + 10x1o + 10x1e
template <> struct Spec<1> {
  static constexpr int D_IN = 50, D_OUT = 80, N_ATTR = 60, HID = 60, NPATH = 6, NUMEL = 1100, SH_DIM = 9;
  static constexpr PathC P[NPATH] = {
      {Z000, 20, 20, 0, 0, 0, 0, 30},     {Z011, 20, 10, 0, 20, 1, 400, 40},
      {Z101, 10, 10, 20, 20, 0, 600, 40}, {Z110, 10, 20, 20, 0, 1, 700, 30},
      {Z111, 10, 10, 20, 50, 1, 900, 10}, {Z121, 10, 10, 20, 20, 4, 1000, 40},
  };
};
// L2: 20x0e+10x1o+10x1e -> 20x0e + 10x1o + 10x1e + 20x0o
template <> struct Spec<2> {
  static constexpr int D_IN = 80, D_OUT = 100, N_ATTR = 60, HID = 60, NPATH = 10, NUMEL = 1600, SH_DIM = 9;
  static constexpr PathC P[NPATH] = {
      {Z000, 20, 20, 0, 0, 0, 0, 30},      {Z011, 20, 10, 0, 20, 1, 400, 50},
      {Z101, 10, 10, 20, 20, 0, 600, 50},  {Z110, 10, 20, 20, 0, 1, 700, 30},
      {Z111, 10, 10, 20, 50, 1, 900, 30},  {Z121, 10, 10, 20, 20, 4, 1000, 50},
      {Z101, 10, 10, 50, 50, 0, 1100, 30}, {Z111, 10, 10, 50, 20, 1, 1200, 50},
      {Z110, 10, 20, 50, 80, 1, 1300, 10}, {Z121, 10, 10, 50, 50, 4, 1500, 30},
  };
};
// L3: 20x0e+10x1o+10x1e+20x0o -> same
template <> struct Spec<3> {
  static constexpr int D_IN = 100, D_OUT = 100, N_ATTR = 60, HID = 60, NPATH = 12, NUMEL = 2200, SH_DIM = 9;
  static constexpr PathC P[NPATH] = {
      {Z000, 20, 20, 0, 0, 0, 0, 30},      {Z011, 20, 10, 0, 20, 1, 400, 50},
      {Z101, 10, 10, 20, 20, 0, 600, 50},  {Z110, 10, 20, 20, 0, 1, 700, 30},
      {Z111, 10, 10, 20, 50, 1, 900, 50},  {Z121, 10, 10, 20, 20, 4, 1000, 50},
      {Z101, 10, 10, 50, 50, 0, 1100, 50}, {Z111, 10, 10, 50, 20, 1, 1200, 50},
      {Z110, 10, 20, 50, 80, 1, 1300, 30}, {Z121, 10, 10, 50, 50, 4, 1500, 50},
      {Z000, 20, 20, 80, 80, 0, 1600, 30}, {Z011, 20, 10, 80, 50, 1, 2000, 50},
  };
};
// final_conv: in = seq3, out = 2x1o + 2x1e ; fc 40 -> 40 -> 200
template <> struct Spec<4> {
  static constexpr int D_IN = 100, D_OUT = 12, N_ATTR = 40, HID = 40, NPATH = 8, NUMEL = 200, SH_DIM = 9;
  static constexpr PathC P[NPATH] = {
      {Z011, 20, 2, 0, 0, 1, 0, 50},   {Z101, 10, 2, 20, 0, 0, 40, 50},  {Z111, 10, 2, 20, 6, 1, 60, 50},
      {Z121, 10, 2, 20, 0, 4, 80, 50}, {Z101, 10, 2, 50, 6, 0, 100, 50}, {Z111, 10, 2, 50, 0, 1, 120, 50},
      {Z121, 10, 2, 50, 6, 4, 140, 50}, {Z011, 20, 2, 80, 6, 1, 160, 50},
  };
};
// tor_bond_conv: in = seq3, sh = consumed part of FullTP(sh, 2e): [s0e | v1o(3) | v1e(3)], out = 20x0o + 20x0e
template <> struct Spec<5> {
  static constexpr int D_IN = 100, D_OUT = 40, N_ATTR = 60, HID = 60, NPATH = 6, NUMEL = 1600, SH_DIM = 7;
  static constexpr PathC P[NPATH] = {
      {Z000, 20, 20, 0, 20, 0, 0, 40},     {Z110, 10, 20, 20, 20, 1, 400, 40}, {Z110, 10, 20, 20, 0, 4, 600, 40},
      {Z110, 10, 20, 50, 0, 1, 800, 40},   {Z110, 10, 20, 50, 20, 4, 1000, 40}, {Z000, 20, 20, 80, 0, 0, 1200, 40},
  };
};

// A "triple" = (path p, in-channel pair q, out channel w): the two lane halves of the 32x32 MFMA hold the weights of
// in-channel u = half*(mul_in/2) + q for the SAME (p, q, w), so accumulator registers are half-independent.
// Triples are processed in two phases - phase 0: paths with scalar outputs, phase 1: paths with vector outputs - so at
// most max(#scalar, #vector) output channels are live in registers at a time.  Each phase starts on a tile boundary.
struct Triple { int p, q, w; bool valid; };

template <int KIND> constexpr int n_triples_phase(int ph) {
  int n = 0;
  for (int i = 0; i < Spec<KIND>::NPATH; ++i)
    if ((d_out_of(Spec<KIND>::P[i].ztype) == 3) == (ph == 1)) n += (Spec<KIND>::P[i].mul_in / 2) * Spec<KIND>::P[i].mul_out;
  return n;
}
template <int KIND> constexpr int n_tiles_phase(int ph) { return (n_triples_phase<KIND>(ph) + 15) / 16; }
template <int KIND> constexpr int n_tiles() { return n_tiles_phase<KIND>(0) + n_tiles_phase<KIND>(1); }
template <int KIND> constexpr Triple triple_at(int n) {   // n = tile*16 + r over the concatenated (padded) phases
  const int nt0 = n_tiles_phase<KIND>(0);
  const int ph = (n / 16) < nt0 ? 0 : 1;
  if (ph == 1) n -= nt0 * 16;
  for (int i = 0; i < Spec<KIND>::NPATH; ++i) {
    if ((d_out_of(Spec<KIND>::P[i].ztype) == 3) != (ph == 1)) continue;
    int cnt = (Spec<KIND>::P[i].mul_in / 2) * Spec<KIND>::P[i].mul_out;
    if (n < cnt) return Triple{i, n / Spec<KIND>::P[i].mul_out, n % Spec<KIND>::P[i].mul_out, true};
    n -= cnt;
  }
  return Triple{0, 0, 0, false};
}
// compact accumulator index of output channel `out_off` within its phase: channels of the other class are skipped.
// Out slots are identified by the distinct out_off values of the paths; a slot's width is mul_out * d_out.
template <int KIND> constexpr int compact_off(int path) {
  using S = Spec<KIND>;
  const int d = d_out_of(S::P[path].ztype);
  int off = 0;
  // sum the widths of the distinct out slots of the same class that start before this path's slot
  for (int i = 0; i < S::NPATH; ++i) {
    if (d_out_of(S::P[i].ztype) != d || S::P[i].out_off >= S::P[path].out_off) continue;
    bool first = true;
    for (int j = 0; j < i; ++j)
      if (S::P[j].out_off == S::P[i].out_off) first = false;
    if (first) off += S::P[i].mul_out * d;
  }
  return off;
}
template <int KIND> constexpr int n_acc_phase(int ph) {
  using S = Spec<KIND>;
  int n = 0;
  for (int i = 0; i < S::NPATH; ++i) {
    if ((d_out_of(S::P[i].ztype) == 3) != (ph == 1)) continue;
    bool first = true;
    for (int j = 0; j < i; ++j)
      if (S::P[j].out_off == S::P[i].out_off) first = false;
    if (first) n += S::P[i].mul_out * d_out_of(S::P[i].ztype);
  }
  return n;
}
template <int KIND> constexpr int n_acc() {
  return n_acc_phase<KIND>(0) > n_acc_phase<KIND>(1) ? n_acc_phase<KIND>(0) : n_acc_phase<KIND>(1);
}
// inverse map for the phase epilogue: compact index -> output channel
template <int KIND> constexpr int channel_of(int ph, int ci) {
  using S = Spec<KIND>;
  for (int i = 0; i < S::NPATH; ++i) {
    if ((d_out_of(S::P[i].ztype) == 3) != (ph == 1)) continue;
    const int w = S::P[i].mul_out * d_out_of(S::P[i].ztype);
    const int co = compact_off<KIND>(i);
    if (ci >= co && ci < co + w) return S::P[i].out_off + (ci - co);
  }
  return -1;
}

// GEMM geometry
template <int KIND> constexpr int k1_steps() { return Spec<KIND>::N_ATTR / 2 + 1; }              // +1: bias step
template <int KIND> constexpr int k1_steps_pad() { return (k1_steps<KIND>() + 3) / 4 * 4; }
// hidden rows live in two 32-row MFMA D tiles; K-step sigma of GEMM2 pairs rows rho(sigma) (lane half 0) and rho+4 (half 1)
constexpr int hid_row(int sigma, int half) { return 32 * (sigma >> 4) + ((sigma & 15) & 3) + 8 * ((sigma & 15) >> 2) + 4 * half; }
// Which hidden unit sits in D-tile row `row`:  0..HID-1 = unit, HID = the constant-one unit that carries fc.3's bias, -1 = padding.
// For HID = 60 unit 59 is moved to row 61 so that rows 59 and 63 - the pair consumed by K-step 31 - are both padding and the
// step can be dropped (31 instead of 32 fp32 MFMAs per tile).
template <int KIND> constexpr int unit_of_row(int row) {
  constexpr int H = Spec<KIND>::HID;
  if (H == 60) {
    if (row < 59) return row;
    if (row == 60) return H;
    if (row == 61) return 59;
    return -1;
  }
  if (row < H) return row;
  if (row == H) return H;
  return -1;
}
template <int KIND> constexpr int k2_steps() {   // number of leading sigma steps that touch a non-padding row
  int last = 0;
  for (int s = 0; s < 32; ++s)
    if (unit_of_row<KIND>(hid_row(s, 0)) >= 0 || unit_of_row<KIND>(hid_row(s, 1)) >= 0) last = s + 1;
  return last;
}
template <int KIND> constexpr int k2_steps_pad() { return (k2_steps<KIND>() + 3) / 4 * 4; }

}  // namespace dp
