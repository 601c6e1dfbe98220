// dp_conv_node_f16_impl.h - the node-side formulation of the fused tensor-product convolution (dp_conv_node_impl.h) at fp32 GRADE on the
// 16-bit matrix pipe of gfx950: every operand of the three GEMMs is split into TWO fp16 limbs (hi = rne16(v), lo = rne16(v - hi): 22
// significand bits, the fp32 exponent range is mapped into fp16's by exact power-of-two scales), the cross products hi*hi + hi*lo + lo*hi
// (+ lo*lo in GEMM-B) run on v_mfma_f32_16x16x32_f16 and accumulate in fp32.  DP_PREC_F16X3.
//
// Replaces TensorProductConvLayer.forward (/root/reference/src/models/score_model_phore.py:134-149) for the ligand-side and cross convs of
// an fp32 model (21 per forward, :676-710) - the launches that were bound by the fp32-input matrix pipe (v_mfma_f32_16x16x4_f32 runs at
// 1/16 of the 16-bit rate; there is no TF32 on gfx950).  Same association as the fp32 and the bf16 node-side kernels:
//     T_p[n][k][u,m]  = sum_{e in n} hid[e][k] z_p[e][u,m]            GEMM-A: 3 MFMAs per (node, 15-column tile, 32-edge pass)
//     out[n][w,m]    += sum_{k,u} W2_p[(u,w)][k] T_p[n][k][u,m]        GEMM-B: K = 32 = two in-channels x the wave's 16 hidden rows
// and the same k-split: a workgroup = 4 waves, wave q owns hidden rows 16q..16q+15, GEMM1 -> GEMM-A -> LDS transpose -> GEMM-B is
// wave-private, the four k-partials meet in LDS once at the end.
//
// What is different from the bf16 kernel (dp_conv_node_bf16_impl.h), and why:
//   * a workgroup owns 8 scatter nodes, not 16: the hidden vectors stay in registers as GEMM-A's A operand, two limbs of 8 nodes are the 64
//     registers one limb of 16 nodes took.  GEMM-B's N = 16 is then (8 nodes) x (2 limbs of T): node slot n < 8 of the T image holds hi(T),
//     slot n + 8 holds lo(T) of the same node, so ONE MFMA per W2 limb covers W2_limb x (T_hi | T_lo) - two MFMAs give all four cross
//     products - and the two column halves of the accumulators are added once, in the epilogue.  The launch has twice the workgroups.
//   * limb extraction on the VALU: v_cvt_pk_f16_f32 (hi of a pair), then v_fma_mixlo_f16 + v_fma_mixhi_f16 (f16(v * 1.0 - hi) written into the
//     two halves of one register, the fp16 half of hi read in place; for the one-term tensor-product operands f16(x * s - hi) with the product
//     exact inside the fma): 2 extra instructions per pair of values against the bf16 kernel's single conversion.
//   * exact power-of-two scales keep the limbs inside fp16's range: W2 (path coefficient, CG norm, BatchNorm scale folded) is stored as
//     W2 * 2^w with max |W2 * 2^w| in (1024, 2048] (w in the image header), the hidden vectors carry 2^(S - w) / degree with S = min over the
//     launch's convs of w + 5, so that every conv of a fused launch accumulates out * 2^S; the epilogue multiplies by 2^-S.  W1 is stored
//     as 16 W1 (relu commutes with the scale; the "edge does not exist" flag row is -60000).  Features, spherical-harmonic factors and the
//     edge embedding are used as they are: |v| < 65504 is the precondition of this precision (a larger value becomes inf -> NaN, which the
//     conformer update reports in status bit 0; values below 2^-14 keep an ABSOLUTE precision of 2^-25).
//   * gathers go through a buffer resource limited to the pose's own rows: an edge slot past the pose's last node reads 0, never a
//     neighbouring pose's features (0 * NaN would leak a diverged neighbour into this pose).
// Accuracy (tests/test_conv_f16_gpu.py): held to the SAME tolerances as the fp32 kernel against the fp64 oracle.
#pragma once
#include "dp_conv_node_bf16_impl.h"

#ifndef DPF_ABL     // timing ablations of THIS kernel (tools/build_variant.sh; results are wrong by construction): 1 no gathered-operand loads, 2 no SH-factor
#define DPF_ABL 0   // reads, 3 no z arithmetic / limb extraction, 4 no GEMM-A MFMAs, 5 no T split / stores, 6 no T reads, 7 no GEMM-B MFMAs, 8 no W2 loads,
#endif              // 9 no edge-embedding loads (GEMM1), 10 z built for every fourth node only (bound of a z build shared by the four waves),
                    // 11 OPTIMISTIC cost model of "the SH factor on the hidden-vector side" (profiles/r6_experiments.md): no per-node
                    // operand build at all (the gathered input's limbs serve every node), the extra GEMM-A MFMA sets that form needs (one per
                    // (sh component, input component) pair with a non-zero CG coefficient), and ONE build of the nine (hid * s_j) limb pairs
                    // per node and pass - as if 576 registers were there to keep them

namespace dpf {
using namespace dpn;
using dpb::u32x4;
using dpb::u32x2;
using dpb::f32x2;
using dpb::lds_ptr;
using dpb::XSrc;
using dpb::uni;
using dpb::make_rsrc;
using dpb::t_col;
using dpb::term_sign;
using dpb::st_count;
using dpb::SUPER_NU;
using dpb::SUPER_CH;
using dpb::COLB;
using dpb::TIMG_B;
using dpb::TIMG_F;
using dpb::TDUM_B;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int NBH = 8;                // scatter nodes per workgroup; node slots NBH..15 of hid / the T image / the accumulator columns = lo limbs
constexpr int EMBH_ROW = 64;          // fp16 per row of the two-limb edge-embedding table: hi (20 features | flag | padding to 32) | lo (32) = 128 bytes
constexpr int HDR_QUADS = 4;          // header of the W2 image (16-byte quads): float[0] = w, the exponent of the image's scale
constexpr int A0 = 5;                 // T carries 2^(A0 + w_min - w) <= 2^5
constexpr float W1_SCALE = 16.f;      // first layer stored as 16 W1
constexpr float FLAG_W = -60000.f;    // W1 row of the "edge does not exist" flag (in units of 16 W1)

// The gathered features of the pass's 32 edge slots are staged in LDS once per pass and workgroup (they are the same rows for every scatter node
// of the group and for all four waves), in two halves of 50 input columns: every layer kind's paths read columns 0..49 first and 50..99 last
// (dp_spec.h), so the buffer is re-filled once per pass, between those two groups of paths.  [32 rows][XROW floats]; 52 = 50 + 2: a row is a
// whole number of 16-byte quads, and two rows read by one ds_read_b32 (lane groups g, g + 1: rows 4 apart = 208 dwords = 16 banks) never share
// a bank for the <= 15 consecutive columns a tile touches.
constexpr int XHALF = 50;             // input columns per staged half
constexpr int XROW = 52;              // floats per staged row
constexpr int XQUADS = XROW / 4;      // 16-byte quads per staged row
template <int KIND> constexpr int x_half_of(int p) { return Spec<KIND>::P[p].in_off >= XHALF ? 1 : 0; }
template <int KIND> constexpr bool x_halves_ok() {
  using S = Spec<KIND>;
  for (int i = 0; i < S::NPATH; ++i) {
    const int lo = S::P[i].in_off, hi = lo + S::P[i].mul_in * d_in_of(S::P[i].ztype);
    if (lo < XHALF && hi > XHALF) return false;                       // a path's columns lie inside one half
    if (hi - XHALF * x_half_of<KIND>(i) > XHALF) return false;
    if (i > 0 && x_half_of<KIND>(i) < x_half_of<KIND>(i - 1)) return false;     // ... and the halves are visited in order
  }
  return true;
}
static_assert(x_halves_ok<0>() && x_halves_ok<1>() && x_halves_ok<2>() && x_halves_ok<3>(), "paths read input columns 0..49, then 50..99");
// all 256 threads: rows 32 et .. 32 et + 31 of the pose's gather table (through the pose-limited buffer resource: a row past the pose's last node
// reads 0), columns 50 half .. 50 half + 51
__device__ __forceinline__ void stage_x(__amdgpu_buffer_rsrc_t xrs, unsigned gstride4, int et, int half, int tid, float* s_X) {
#pragma unroll
  for (int idx = tid; idx < 32 * XQUADS; idx += 256) {
    const int row = idx / XQUADS, qd = idx - row * XQUADS;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, (unsigned)(32 * et + row) * gstride4 + (unsigned)((XHALF * half + 4 * qd) * 4), 0, 0);
    *reinterpret_cast<u32x4*>(s_X + row * XROW + 4 * qd) = v;
  }
}

template <int KIND> constexpr int w2h_path_off(int p) {             // in 16-byte quads
  int o = HDR_QUADS;
  for (int i = 0; i < p; ++i) o += st_count<KIND>(i) * 2 * SUPER_CH * path_mt<KIND>(i) * 4 * 64;
  return o;
}
template <int KIND> constexpr int w2h_quads() { return w2h_path_off<KIND>(Spec<KIND>::NPATH); }
constexpr int W1H_QUADS = 3 * 2 * 4 * 64;                            // three images (edge embedding, attr1, attr2) x 2 limbs x 4 waves

// ReLU as ONE instruction: v_med3_f32 x, 0, FLT_MAX (with +inf as the upper bound this toolchain folds the median into a maximum and emits
// v_max x, x (canonicalise) + v_max x, 0 for a value that comes out of an MFMA)
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 3.40282347e38f); }
__device__ __forceinline__ unsigned pkh(float lo, float hi) {       // v_cvt_pk_f16_f32 (round to nearest even)
  const f16x2 v = __builtin_convertvector((f32x2){lo, hi}, f16x2);
  return __builtin_bit_cast(unsigned, v);
}
// Limb extraction.  hi limbs of a pair: v_cvt_pk_f16_f32.  lo limbs: v_fma_mixlo_f16 / v_fma_mixhi_f16 compute f16(a * b - (float)half) from
// fp32 sources with the fp16 half read IN PLACE, and write one half of the destination (the other half is preserved): two instructions give the
// packed lo limbs of a pair - no unpack, no second conversion.  The residual v - hi is exact (hi is within half an fp16 ulp of v).
// Written as plain C that the backend's mad-mix patterns turn into exactly those instructions - (_Float16)fma(v, one, -(float)h) with an OPAQUE
// one (a literal 1.0 would be folded into a subtraction and lose the pattern) - in translation units built with -fno-slp-vectorize (the SLP
// vectoriser would pack the two scalar fmas into v_pk_fma_f32 + unpack + convert).  NOT inline asm: an instruction that writes a 16-bit half
// needs one wait state before a dependent instruction (gfx940 dst-sel forwarding), and the compiler only knows that for instructions it
// emitted itself - behind an inline-asm v_fma_mixhi_f16 it guards VALU / MFMA consumers but not an LDS store of the result (measured: the
// ds_write stored the stale half).  The helpers work on TWO pairs at a time, in an order that leaves an independent instruction between
// dependent ones, so the wait states cost nothing.
__device__ __forceinline__ float opaque_one() {
  float o = 1.0f;
  asm("" : "+s"(o));
  return o;
}
__device__ __forceinline__ unsigned hbits(f16x2 v) { return __builtin_bit_cast(unsigned, v); }
struct HL { unsigned h, l; };
struct HL2 { unsigned h0, h1, l0, l1; };
// two pairs (v0, v1), (v2, v3) -> hi / lo limbs, 6 instructions
__device__ __forceinline__ HL2 split4(float v0, float v1, float v2, float v3) {
  const float one = opaque_one();
  const f16x2 h0 = __builtin_convertvector((f32x2){v0, v1}, f16x2), h1 = __builtin_convertvector((f32x2){v2, v3}, f16x2);
  f16x2 l0, l1;
  l0[0] = (_Float16)__builtin_fmaf(v0, one, -(float)h0[0]);
  l1[0] = (_Float16)__builtin_fmaf(v2, one, -(float)h1[0]);
  l0[1] = (_Float16)__builtin_fmaf(v1, one, -(float)h0[1]);
  l1[1] = (_Float16)__builtin_fmaf(v3, one, -(float)h1[1]);
  return HL2{hbits(h0), hbits(h1), hbits(l0), hbits(l1)};
}
// two pairs of PRODUCTS (x0 s0, x1 s1) = xa * sa, (x2 s2, x3 s3) = xb * sb: hi = rne16(fl32(x s)), lo = rne16(x s - hi) with the product EXACT inside
// the fma (so hi + lo also recovers what the fp32 rounding of the product dropped), 8 instructions.  hi is computed ONCE, from the packed fp32
// products: written as (_Float16)(x * s) per element the backend selects v_cvt_pk(v_mul, v_mul) for the limb that feeds the MFMA and
// v_fma_mixlo_f16(x, s, 0) for the copy the residual subtracts - two roundings of the same product that differ in rare cases by an fp16 ulp
// (measured: 1e-5 instead of 3e-7 of the output's magnitude).
__device__ __forceinline__ HL2 split4_mul(f32x2 xa, f32x2 sa, f32x2 xb, f32x2 sb) {
  const f32x2 pa = xa * sa, pb = xb * sb;
  const f16x2 h0 = __builtin_convertvector(pa, f16x2), h1 = __builtin_convertvector(pb, f16x2);
  f16x2 l0, l1;
  l0[0] = (_Float16)__builtin_fmaf(xa[0], sa[0], -(float)h0[0]);
  l1[0] = (_Float16)__builtin_fmaf(xb[0], sb[0], -(float)h1[0]);
  l0[1] = (_Float16)__builtin_fmaf(xa[1], sa[1], -(float)h0[1]);
  l1[1] = (_Float16)__builtin_fmaf(xb[1], sb[1], -(float)h1[1]);
  return HL2{hbits(h0), hbits(h1), hbits(l0), hbits(l1)};
}
__device__ __forceinline__ HL split2(float v0, float v1) {         // one pair (cold paths)
  const float one = opaque_one();
  const f16x2 h = __builtin_convertvector((f32x2){v0, v1}, f16x2);
  f16x2 l;
  l[0] = (_Float16)__builtin_fmaf(v0, one, -(float)h[0]);
  l[1] = (_Float16)__builtin_fmaf(v1, one, -(float)h[1]);
  return HL{hbits(h), hbits(l)};
}
__device__ __forceinline__ f16x8 as_h(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(a), as_h(b), c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_a(u32x4 a, u32x4 b, f32x4 c) {      // GEMM-A
#if DPF_ABL == 4
  asm volatile("" ::"v"(a), "v"(b));
  return c + f32x4{__builtin_bit_cast(float, b[0]), 1.f, 2.f, 3.f};
#else
  return mfma_h(a, b, c);
#endif
}
__device__ __forceinline__ f32x4 mfma_b(u32x4 a, u32x4 b, f32x4 c) {      // GEMM-B
#if DPF_ABL == 7
  asm volatile("" ::"v"(a), "v"(b));
  return c;
#else
  return mfma_h(a, b, c);
#endif
}
// hi*hi + hi*lo + lo*hi, smallest terms first
__device__ __forceinline__ f32x4 mfma3(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x4 c) {
  c = mfma_h(al, bh, c);
  c = mfma_h(ah, bl, c);
  return mfma_h(ah, bh, c);
}

__device__ __forceinline__ Grp grp_of8(const DpConvArgs& a, int bid) {     // as grp_of, with groups of <= 8 nodes on 2 * slices workgroup slots per pose
  Grp r;
  const int sl2 = 2 * a.slices;
  r.pose = bid / sl2;
  const int slice = bid % sl2;
  const int sc0 = a.scat_ptr[r.pose];
  const int nsc = a.scat_ptr[r.pose + 1] - sc0;
  const int ng = (nsc + NBH - 1) / NBH;
  if (slice >= ng) { r.n_lo = sc0; r.n_loc = 0; return r; }
  const int lo = (nsc * slice) / ng, hi = (nsc * (slice + 1)) / ng;
  r.n_lo = sc0 + lo;
  r.n_loc = hi - lo;
  return r;
}

// first path that reads the second half of the input columns (NPATH if none)
template <int KIND> constexpr int first_half1_path() {
  for (int i = 0; i < Spec<KIND>::NPATH; ++i)
    if (x_half_of<KIND>(i) == 1) return i;
  return Spec<KIND>::NPATH;
}
// S0: the conv's spherical harmonics are e0 = (1, 0, ..., 0) on every edge the launch covers (DpConvArgs.s0_only: the normals convs, whose
// few other edges the caller adds through the edge-side kernel): the paths that multiply by an l > 0 component have a zero operand and are
// skipped - GEMM-A and GEMM-B of the l = 0 paths only, whose operand is the gathered input itself (no per-node operand build at all)
template <int KIND, int MODE, bool HALF, bool S0, int P>
struct StageH {
  static __device__ __forceinline__ void run(__amdgpu_buffer_rsrc_t wrs, __amdgpu_buffer_rsrc_t xrs, unsigned gstride4, int et, int tid, float* sX, int q,
                                             int lane, const u32x4 (&hid)[2 * NBH], unsigned tl_base, unsigned dummy_base, const float* sS,
                                             f32x4 (&acc)[n_acc4<KIND>()]) {
    using S = Spec<KIND>;
    if constexpr (P < S::NPATH) {
      constexpr PathC pc = S::P[P];
      constexpr int ZT = pc.ztype;
      constexpr bool VEC = d_out_of(ZT) == 3;
      constexpr int NG = VEC ? 3 : 1, NTERM = n_terms<ZT>(), MT = path_mt<KIND>(P), AB = acc_base<KIND>(P), NTILE = VEC ? 2 : 1;
      constexpr int NJ = HALF ? 4 : 8, NT2 = HALF ? 1 : 2;
      static_assert(!S0 || first_half1_path<KIND>() == S::NPATH || !needs_s<S::P[first_half1_path<KIND>() < S::NPATH ? first_half1_path<KIND>() : 0].ztype>(),
                    "s0-only: the path that re-stages the input columns is an l = 0 path");
      if constexpr (!(S0 && needs_s<ZT>())) {
      if constexpr (P > 0 && x_half_of<KIND>(P) != x_half_of<KIND>(P > 0 ? P - 1 : 0)) {
        __syncthreads();                            // every wave is done with the first half of the input columns
        stage_x(xrs, gstride4, et, 1, tid, sX);
        __syncthreads();
      }
#pragma unroll 1
      for (int st = 0; st < st_count<KIND>(P); ++st) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));          // lane-derived constants are rebuilt per supertile (keeps LICM from hoisting and spilling them)
        const int g = lane_l >> 4, c = lane_l & 15;
        const int m = VEC ? (c >= 10 ? 2 : (c >= 5 ? 1 : 0)) : 0;
        const bool colv = VEC ? c < 15 : c < 10;
        // gathered operand of the lane's column in both tiles, for the lane's 8 (HALF: 4) edge slots, as pairs of consecutive slots (the operand
        // layout of v_pk_*_f32), from the staged rows: slot pair jp = rows 16 (jp >> 1) + 4 g + 2 (jp & 1), + 1 (the same rows for every node)
        f32x2 xa[NTILE][NTERM][NJ / 2];
        {
          constexpr int DIN = d_in_of(ZT);
          const int ul = VEC ? c - 5 * m : c;
#pragma unroll
          for (int tl = 0; tl < NTILE; ++tl) {
            const int u = colv ? SUPER_NU * st + 5 * tl + ul : 0;
            const float* xb = sX + (4 * g) * XROW + (pc.in_off - XHALF * x_half_of<KIND>(P)) + u * DIN;
#pragma unroll
            for (int t = 0; t < NTERM; ++t) {
              const float* xt = xb + term_of<ZT>(t, m).xcomp;
#pragma unroll
              for (int jp = 0; jp < NJ / 2; ++jp) {
                const int row = 16 * (jp >> 1) + 2 * (jp & 1);
#if DPF_ABL == 1
                xa[tl][t][jp] = f32x2{(float)(size_t)xt, (float)row};
#else
                xa[tl][t][jp] = f32x2{xt[row * XROW], xt[(row + 1) * XROW]};
#endif
              }
            }
          }
        }
        // T image: byte address of (column, k quad g, node slot) = column * 520 + g * 128 + slot * 8; slot nn = hi limb, slot nn + 8 = lo limb
        unsigned twr[NTILE];
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl)
          twr[tl] = (colv ? tl_base + (unsigned)((tl * TCOLS + c) * COLB) : dummy_base + (unsigned)(c * 8)) + (unsigned)(g * 128);
        // the supertile's W2 fragments (GEMM-B's A operands: 2 limbs x 5 K chunks x MT), requested before GEMM-A, which hides the L2 round trip
        // (through a buffer resource: one VGPR offset per lane, the fragment's offset is wave-uniform - no 64-bit address arithmetic per load)
        const unsigned wvoff = (unsigned)(q * 64 + lane_l) * 16u;
        const unsigned wsoff = uni((unsigned)(w2h_path_off<KIND>(P) + st * (2 * SUPER_CH * MT * 256)) * 16u);
        u32x4 wall[2][SUPER_CH][MT];
#pragma unroll
        for (int lb = 0; lb < 2; ++lb)
#pragma unroll
          for (int ch = 0; ch < SUPER_CH; ++ch)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#if DPF_ABL == 8
              wall[lb][ch][mt] = u32x4{wvoff, wsoff, (unsigned)ch, (unsigned)(lb + mt)};
#else
              wall[lb][ch][mt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff + (unsigned)(((lb * SUPER_CH + ch) * MT + mt) * 256 * 16), 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM-A: three MFMAs per (node, tile); T leaves as two fp16 limbs ---------------------------------------------------------
        // node nn's MFMAs go into a ring of two accumulator sets, their results are split / stored one node LATER (behind the next node's
        // z arithmetic); the SH factors of node nn + 1 are requested before node nn's arithmetic
        f32x4 Dq[2][NTILE];
        auto store_t = [&](int nn) {
#pragma unroll
          for (int tl = 0; tl < NTILE; ++tl) {
            const f32x4 D = Dq[nn & 1][tl];
#if DPF_ABL == 5
            asm volatile("" ::"v"(D), "v"(twr[tl]));
#else
            const HL2 t = split4(D[0], D[1], D[2], D[3]);
            *lds_ptr(twr[tl] + (unsigned)(nn << 3)) = u32x2{t.h0, t.h1};
            *lds_ptr(twr[tl] + (unsigned)((nn + NBH) << 3)) = u32x2{t.l0, t.l1};
#endif
          }
        };
        if constexpr (needs_s<ZT>() && DPF_ABL != 11) {
          const char* sbase = reinterpret_cast<const char*>(sS);
          int soff[NTERM];
#pragma unroll
          for (int t = 0; t < NTERM; ++t) soff[t] = (term_of<ZT>(t, m).scomp * 16 + 4 * g) * 4;
          auto loadS = [&](int nn, f32x4 (&sv)[NTERM][NT2]) {
#pragma unroll
            for (int t = 0; t < NTERM; ++t)
#pragma unroll
              for (int t2 = 0; t2 < NT2; ++t2)
#if DPF_ABL == 2
                sv[t][t2] = f32x4{(float)soff[t], (float)nn, 1.f, 2.f};
#else
                sv[t][t2] = *reinterpret_cast<const f32x4*>(sbase + (2 * nn + t2) * (SCOMP * 64) + soff[t]);
#endif
          };
          constexpr int SD = (VEC && NTERM == 3) ? 1 : 2;
          f32x4 sv[SD][NTERM][NT2];
          loadS(0, sv[0]);
#if DPF_ABL == 10
          u32x4 zh_keep[NTILE], zl_keep[NTILE];
#endif
#pragma unroll
          for (int nn = 0; nn < NBH; ++nn) {
            const int cur = SD == 2 ? (nn & 1) : 0;
            if constexpr (SD == 2) {
              if (nn + 1 < NBH) loadS(nn + 1, sv[(nn + 1) & 1]);
            }
#if DPF_ABL == 10        // upper bound of a z build shared by the four waves: only every fourth node builds its z (the others reuse it)
            static_assert(true, "");
#endif
            u32x4 zh[NTILE], zl[NTILE];
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) {
#if DPF_ABL == 10
              if ((nn & 3) != 0) { zh[tl] = zh_keep[tl]; zl[tl] = zl_keep[tl]; continue; }
#endif
              zh[tl] = u32x4{0u, 0u, 0u, 0u};
              zl[tl] = u32x4{0u, 0u, 0u, 0u};
#if DPF_ABL == 3
              zh[tl] = __builtin_bit_cast(u32x4, f32x4{xa[tl][0][0][0], xa[tl][NTERM - 1][NJ / 2 - 1][1], sv[cur][0][0][0], sv[cur][NTERM - 1][NT2 - 1][3]});
              zl[tl] = zh[tl];
              continue;
#endif
              // edge slots 2 jp, 2 jp + 1 of the lane = components (2 jp & 3, + 1) of the factor quad of 16-edge tile jp >> 1
              auto spair = [&](int t, int jp) { const f32x4 q4 = sv[cur][t][jp >> 1]; return (jp & 1) ? f32x2{q4[2], q4[3]} : f32x2{q4[0], q4[1]}; };
              if constexpr (NTERM == 1) {
#pragma unroll
                for (int j2 = 0; j2 < NJ / 4; ++j2) {           // two pairs at a time (see split4_mul)
                  const HL2 r = split4_mul(xa[tl][0][2 * j2], spair(0, 2 * j2), xa[tl][0][2 * j2 + 1], spair(0, 2 * j2 + 1));
                  zh[tl][2 * j2] = r.h0; zh[tl][2 * j2 + 1] = r.h1;
                  zl[tl][2 * j2] = r.l0; zl[tl][2 * j2 + 1] = r.l1;
                }
              } else {
                // term by term over all pairs (independent chains side by side), then the limbs of two pairs at a time
                f32x2 zz[NJ / 2];
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) zz[jp] = xa[tl][0][jp] * spair(0, jp);
#pragma unroll
                for (int t = 1; t < NTERM; ++t)
#pragma unroll
                  for (int jp = 0; jp < NJ / 2; ++jp) {
                    if (term_sign<ZT>(t) < 0) zz[jp] = __builtin_elementwise_fma(-xa[tl][t][jp], spair(t, jp), zz[jp]);
                    else zz[jp] = __builtin_elementwise_fma(xa[tl][t][jp], spair(t, jp), zz[jp]);
                  }
#pragma unroll
                for (int j2 = 0; j2 < NJ / 4; ++j2) {
                  const HL2 r = split4(zz[2 * j2][0], zz[2 * j2][1], zz[2 * j2 + 1][0], zz[2 * j2 + 1][1]);
                  zh[tl][2 * j2] = r.h0; zh[tl][2 * j2 + 1] = r.h1;
                  zl[tl][2 * j2] = r.l0; zl[tl][2 * j2 + 1] = r.l1;
                }
              }
            }
#if DPF_ABL == 10
            if ((nn & 3) == 0) {
#pragma unroll
              for (int tl = 0; tl < NTILE; ++tl) { zh_keep[tl] = zh[tl]; zl_keep[tl] = zl[tl]; }
            }
#endif
            if constexpr (SD == 1) {
              __builtin_amdgcn_sched_barrier(0);
              if (nn + 1 < NBH) loadS(nn + 1, sv[0]);
            }
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) Dq[nn & 1][tl] = mfma_a(hid[NBH + nn], zh[tl], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) Dq[nn & 1][tl] = mfma_a(hid[nn], zl[tl], Dq[nn & 1][tl]);
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) Dq[nn & 1][tl] = mfma_a(hid[nn], zh[tl], Dq[nn & 1][tl]);
            if (nn > 0) store_t(nn - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // z is the gathered input itself (sh_0 = 1): the B operands are the same for every node
          u32x4 zh[NTILE], zl[NTILE];
#pragma unroll
          for (int tl = 0; tl < NTILE; ++tl) {
            zh[tl] = u32x4{0u, 0u, 0u, 0u};
            zl[tl] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int j2 = 0; j2 < NJ / 4; ++j2) {
              const HL2 r = split4(xa[tl][0][2 * j2][0], xa[tl][0][2 * j2][1], xa[tl][0][2 * j2 + 1][0], xa[tl][0][2 * j2 + 1][1]);
              zh[tl][2 * j2] = r.h0; zh[tl][2 * j2 + 1] = r.h1;
              zl[tl][2 * j2] = r.l0; zl[tl][2 * j2 + 1] = r.l1;
            }
          }
          // nodes in PAIRS: the three dependent MFMAs of a (node, tile) chain are issued round-robin over the pair's 2 x NTILE chains, and the
          // previous pair's results are split / stored in their shadow (there is no z arithmetic here to hide the MFMA latency behind)
          f32x4 Dp[2][2][NTILE];
          auto store_p = [&](int np) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
              for (int tl = 0; tl < NTILE; ++tl) {
                const f32x4 D = Dp[np & 1][k][tl];
#if DPF_ABL == 5
                asm volatile("" ::"v"(D), "v"(twr[tl]));
#else
                const HL2 t = split4(D[0], D[1], D[2], D[3]);
                const int nn = 2 * np + k;
                *lds_ptr(twr[tl] + (unsigned)(nn << 3)) = u32x2{t.h0, t.h1};
                *lds_ptr(twr[tl] + (unsigned)((nn + NBH) << 3)) = u32x2{t.l0, t.l1};
#endif
              }
          };
#pragma unroll
          for (int np = 0; np < NBH / 2; ++np) {
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl)
#pragma unroll
              for (int k = 0; k < 2; ++k) Dp[np & 1][k][tl] = mfma_a(hid[NBH + 2 * np + k], zh[tl], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl)
#pragma unroll
              for (int k = 0; k < 2; ++k) Dp[np & 1][k][tl] = mfma_a(hid[2 * np + k], zl[tl], Dp[np & 1][k][tl]);
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl)
#pragma unroll
              for (int k = 0; k < 2; ++k) Dp[np & 1][k][tl] = mfma_a(hid[2 * np + k], zh[tl], Dp[np & 1][k][tl]);
#if DPF_ABL == 11
            {
              // MFMA sets of the re-associated form per supertile and node: Z011 3 (one per m), Z110 3 (one per i), Z111 6, Z121 9 - minus the NTILE above
              constexpr int XSETS = !needs_s<ZT>() ? 0 : (ZT == Z011 ? 3 : ZT == Z110 ? 3 : ZT == Z111 ? 6 : 9) - NTILE;
#pragma unroll
              for (int x = 0; x < XSETS; ++x)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                  f32x4& D = Dp[np & 1][k][x % NTILE];
                  D = mfma_a(hid[NBH + 2 * np + k], zh[x % NTILE], D);
                  D = mfma_a(hid[2 * np + k], zl[x % NTILE], D);
                  D = mfma_a(hid[2 * np + k], zh[x % NTILE], D);
                }
            }
#endif
            if (np > 0) store_p(np - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          store_p(NBH / 2 - 1);
        }
        if constexpr (needs_s<ZT>() && DPF_ABL != 11) store_t(NBH - 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM-B: out[w][slot] += W2_limb[w][(u,k)] (T_hi | T_lo)[(u,k)][slot], K = (two in-channels) x (the wave's 16 hidden rows) ----------
        {
          const unsigned rbase = tl_base + (unsigned)(g * 128 + c * 8);          // GEMM-B view: node slot c, k quad g; the column is an immediate offset
          auto readT = [&](int ch, u32x4 (&tb)[NG]) {
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
#if DPF_ABL == 6
              const u32x2 lo = {rbase + ch, (unsigned)grp}, hi = {rbase, 7u};
#else
              const u32x2 lo = *lds_ptr(rbase + (unsigned)(t_col<VEC>(2 * ch, grp) * COLB));
              const u32x2 hi = *lds_ptr(rbase + (unsigned)(t_col<VEC>(2 * ch + 1, grp) * COLB));
#endif
              tb[grp] = u32x4{lo.x, lo.y, hi.x, hi.y};
            }
          };
          u32x4 tb[2][NG];
          readT(0, tb[0]);
#pragma unroll
          for (int ch = 0; ch < SUPER_CH; ++ch) {
            if (ch + 1 < SUPER_CH) readT(ch + 1, tb[(ch + 1) & 1]);
#pragma unroll
            for (int lb = 1; lb >= 0; --lb)            // lo limb of W2 first (smaller terms first)
#pragma unroll
              for (int grp = 0; grp < NG; ++grp)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[AB + grp * MT + mt] = mfma_b(wall[lb][ch][mt], tb[ch & 1][grp], acc[AB + grp * MT + mt]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      }   // !(S0 && needs_s)
      StageH<KIND, MODE, HALF, S0, P + 1>::run(wrs, xrs, gstride4, et, tid, sX, q, lane, hid, tl_base, dummy_base, sS, acc);
    }
  }
};

// one pass (32 edge slots per node; HALF: 16) of one conv: stage the SH factors, GEMM1 -> hid (two fp16 limbs, A fragments), then the path stages
template <int KIND, int MODE, bool HALF, bool S0>
__device__ __forceinline__ void pass_f16(const DpConvArgs& a, __amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t ers, __amdgpu_buffer_rsrc_t wrs, const Grp& grp, const Wg& r, int et, int tid,
                                         float* s_T, float* s_S, float* s_X, const float* s_inv, f32x4 (&acc)[n_acc4<KIND>()]) {
  const int lane = tid & 63, q = tid >> 6, g = lane >> 4, c = lane & 15;
  constexpr int NT2 = HALF ? 1 : 2;
  constexpr int NSTEP = NBH * NT2;
  float* sT = s_T + q * TIMG_F;
  int* const s_pair = reinterpret_cast<int*>(s_T + TIMG_F - 2 * NBH * 32);     // LL mode: [2][NBH][32] edge slots of the pass's pairs
  (void)s_pair;
  __syncthreads();                               // readers of the previous pass's factors (and of s_pair / the T images) are done
  // ---- operands that do not depend on the staging: the wave's W1 images (2 limbs each) and the pass's node rows -----------------------
  const u32x4* w1 = reinterpret_cast<const u32x4*>(a.w1n);
  auto w1q = [&](int img, int lb) { return w1[((img * 2 + lb) * 4 + q) * 64 + lane]; };
  const u32x4 Wemb_h = w1q(0, 0), Wemb_l = w1q(0, 1);
  constexpr int IMG_S = MODE == DP_EDGES_L2P ? 2 : 1, IMG_G = MODE == DP_EDGES_L2P ? 1 : 2;      // attr block of the scatter / gather node
  // a node tile's A fragment (two limbs): lane (row c, group g) holds features 8g..8g+7 of the row's first 20, feature 20 = 1 carries the bias
  auto attr_frag = [&](const float* rowp, bool bias, u32x4& fh, u32x4& fl) {
    const int o = g < 2 ? 8 * g : 16;
    f32x4 lo = *reinterpret_cast<const f4u*>(rowp + o);
    f32x4 hi = *reinterpret_cast<const f4u*>(rowp + (g < 2 ? o + 4 : 16));
    if (g >= 2) hi = f32x4{(g == 2 && bias) ? 1.f : 0.f, 0.f, 0.f, 0.f};
    if (g == 3) lo = f32x4{0.f, 0.f, 0.f, 0.f};
    const HL2 s0 = split4(lo[0], lo[1], lo[2], lo[3]), s1 = split4(hi[0], hi[1], hi[2], hi[3]);
    fh = u32x4{s0.h0, s0.h1, s1.h0, s1.h1};
    fl = u32x4{s0.l0, s0.l1, s1.l0, s1.l1};
  };
  u32x4 ag_h[NT2], ag_l[NT2], as_h, as_l;
  {
    const float* s_tab = MODE == DP_EDGES_L2P ? a.attr2_tab : a.attr1_tab;
    const float* g_tab = MODE == DP_EDGES_L2P ? a.attr1_tab : a.attr2_tab;
    const int s_stride = MODE == DP_EDGES_L2P ? a.attr2_stride : a.attr1_stride, g_stride = MODE == DP_EDGES_L2P ? a.attr1_stride : a.attr2_stride;
#pragma unroll
    for (int t2 = 0; t2 < NT2; ++t2) {
      const int e0 = 32 * et + 16 * t2 + c;
      const int grow = (MODE == DP_EDGES_P2L ? r.p0 : r.l0) + (e0 < r.En ? e0 : 0);
      attr_frag(g_tab + (size_t)grow * g_stride, false, ag_h[t2], ag_l[t2]);
    }
    const int srow = grp.n_lo + (c < grp.n_loc ? c : 0);
    attr_frag(s_tab + (size_t)srow * s_stride, true, as_h, as_l);
  }
  // ---- stage the first half of the gathered rows, the SH-derived factors of the pass (fp32) and, in the pair mode, the pass's pair ->
  // edge-slot entries ------------------------------------------------------------------------------------------------------------------------
  const unsigned gstride4 = uni((unsigned)a.gather_stride * 4u);
  stage_x(xrs, gstride4, et, 0, tid, s_X);
  constexpr int SLOTS = HALF ? 16 : 32;
  static_assert(!(S0 && MODE == LL), "s0-only is a mode of the dense cross convs");
  if constexpr (!S0) {          // (s0-only: no path reads the factors)
#pragma unroll
  for (int idx = tid; idx < NBH * SLOTS; idx += 256) {
    const int nn = idx / SLOTS, el = idx % SLOTS, e = 32 * et + el;
    bool ok = nn < grp.n_loc && e < r.En;
    int frow = 0;
    if constexpr (MODE == LL) {
      int sa = -1, sb = -1;
      if (ok) { sa = pair_slot<0>(a, r, grp.n_lo + nn, e); sb = pair_slot<1>(a, r, grp.n_lo + nn, e); }
      s_pair[nn * 32 + el] = sa;
      s_pair[NBH * 32 + nn * 32 + el] = sb;
      frow = sa >= 0 ? sa : sb;
      ok = frow >= 0;
    } else if (ok) {
      frow = feat_row<MODE>(r, grp.n_lo + nn, e);
    }
    float v[SCOMP];
#pragma unroll
    for (int i = 0; i < SCOMP; ++i) v[i] = 0.f;
    if (ok) {
      const float* ps = a.edge_sh + (size_t)frow * a.sh_stride;
      const float cc = 0.57735026918962576451f * ps[6];
      v[0] = ps[1]; v[1] = ps[2]; v[2] = ps[3];
      v[3] = -cc - ps[8]; v[4] = ps[5]; v[5] = ps[4]; v[6] = 2.f * cc; v[7] = ps[7]; v[8] = -cc + ps[8];
    }
#pragma unroll
    for (int i = 0; i < SCOMP; ++i) s_S[((nn * 2 + (el >> 4)) * SCOMP + i) * 16 + (el & 15)] = v[i];
  }
  }
  __syncthreads();

  // ---- GEMM1: 16 pre[e][k] = 16 (W1_emb emb[e] + u_s[scatter node] + u_g[gather node]); hid = relu(.) * 2^(S-w) / (16 degree), two limbs ----
  f32x4 ug[NT2];
  {
    const u32x4 Wga_h = w1q(IMG_G, 0), Wga_l = w1q(IMG_G, 1);
#pragma unroll
    for (int t2 = 0; t2 < NT2; ++t2) ug[t2] = mfma3(ag_h[t2], ag_l[t2], Wga_h, Wga_l, f32x4{0.f, 0.f, 0.f, 0.f});     // rows = the pass's edge slots
    const u32x4 Wsc_h = w1q(IMG_S, 0), Wsc_l = w1q(IMG_S, 1);
    const f32x4 us = mfma3(as_h, as_l, Wsc_h, Wsc_l, f32x4{0.f, 0.f, 0.f, 0.f});                                       // rows = the group's nodes, + bias
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sT[(4 * g + rr) * 16 + c] = us[rr];             // [node][k]: read back as a broadcast over rows
  }
  // A fragment (two limbs) of (node nn, 16-edge tile t2): 2 x 16 bytes of the edge's embedding row; the flag row for an edge that does not exist
  auto load_a = [&](int step, int which, u32x4& eh, u32x4& el) {
    int nn_o = HALF ? step : step >> 1;
    asm volatile("" : "+s"(nn_o));                // opaque: keeps the nodes' row arithmetic from being hoisted out of the pass (spills)
    const int t2s = HALF ? 0 : step & 1;
    const int e0 = 32 * et + 16 * t2s + c;
    int frow;
    if constexpr (MODE == LL) {
      frow = s_pair[which * (NBH * 32) + nn_o * 32 + 16 * t2s + c];
    } else {
      frow = (nn_o < grp.n_loc && e0 < r.En) ? feat_row<MODE>(r, grp.n_lo + nn_o, e0) : -1;
    }
    if (frow < 0) frow = a.emb_flag_row;
    const unsigned off = (unsigned)frow * (unsigned)(EMBH_ROW * 2) + (unsigned)(16 * g);
#if DPF_ABL == 9
    eh = u32x4{off, 0u, 0u, 7u};
    el = u32x4{off, 1u, 0u, 7u};
#else
    eh = __builtin_amdgcn_raw_buffer_load_b128(ers, off, 0, 0);
    el = __builtin_amdgcn_raw_buffer_load_b128(ers, off, 64, 0);            // (soffset = 64: the lo limbs of the same features)
#endif
  };
  constexpr int PF = 3;
  u32x4 ea_h[PF + 1], ea_l[PF + 1], eb_h[PF + 1], eb_l[PF + 1];
  (void)eb_h; (void)eb_l;
#pragma unroll
  for (int s0 = 0; s0 < PF; ++s0) {
    load_a(s0, 0, ea_h[s0], ea_l[s0]);
    if constexpr (MODE == LL) load_a(s0, 1, eb_h[s0], eb_l[s0]);
  }
  u32x4 hid[2 * NBH];
  float hf[8];
  f32x4 Dg[2], Dg2[2];
  bool has2[2] = {false, false};
  (void)Dg2; (void)has2;
  auto finish = [&](int step) {
    const int nn = HALF ? step : step >> 1, t2 = HALF ? 0 : step & 1;
    const f32x4 D = Dg[step & 1];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) hf[4 * t2 + rr] = relu1(D[rr]);
    if constexpr (MODE == LL) {
      if (has2[step & 1]) {
        const f32x4 D2 = Dg2[step & 1];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hf[4 * t2 + rr] += relu1(D2[rr]);
      }
    }
    if (HALF || t2 == 1) {
      const float inv = s_inv[nn];               // 2^(S - w) / (16 degree): scatter MEAN and the conv's scale, applied to the hidden vectors
      const HL2 p0 = split4(hf[0] * inv, hf[1] * inv, hf[2] * inv, hf[3] * inv);
      hid[nn] = u32x4{p0.h0, p0.h1, 0u, 0u};
      hid[NBH + nn] = u32x4{p0.l0, p0.l1, 0u, 0u};
      if constexpr (!HALF) {
        const HL2 p1 = split4(hf[4] * inv, hf[5] * inv, hf[6] * inv, hf[7] * inv);
        hid[nn][2] = p1.h0; hid[nn][3] = p1.h1;
        hid[NBH + nn][2] = p1.l0; hid[NBH + nn][3] = p1.l1;
      }
    }
  };
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int nn = HALF ? step : step >> 1, t2 = HALF ? 0 : step & 1;
    if (step + PF < NSTEP) {
      load_a(step + PF, 0, ea_h[(step + PF) % (PF + 1)], ea_l[(step + PF) % (PF + 1)]);
      if constexpr (MODE == LL) load_a(step + PF, 1, eb_h[(step + PF) % (PF + 1)], eb_l[(step + PF) % (PF + 1)]);
    }
    const float us_n = sT[nn * 16 + c];
    const f32x4 init = {us_n + ug[t2][0], us_n + ug[t2][1], us_n + ug[t2][2], us_n + ug[t2][3]};
    Dg[step & 1] = mfma3(ea_h[step % (PF + 1)], ea_l[step % (PF + 1)], Wemb_h, Wemb_l, init);
    if constexpr (MODE == LL) {
      // the bond edges of the same pairs: second layer-1 evaluation over the same u_s + u_g, skipped per 16-pair tile when none of them is
      // bonded.  An absent bond edge reads the flag row: its slot 20 (lane group 2, third dword, low half of the hi limb) is 1.0
      const u32x4 b4 = eb_h[step % (PF + 1)];
      has2[step & 1] = __builtin_amdgcn_ballot_w64(g == 2 && (b4[2] & 0xFFFFu) == 0u) != 0;
      if (has2[step & 1]) Dg2[step & 1] = mfma3(b4, eb_l[step % (PF + 1)], Wemb_h, Wemb_l, init);
    }
    if (step > 0) finish(step - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  finish(NSTEP - 1);
#if DPF_ABL == 11
  {   // nine (hid * s_j) limb pairs per node, built once per pass (16 VALU instructions each), results discarded
    const f32x2 fa0 = {ug[0][0], ug[0][1]}, fa1 = {ug[0][2], ug[0][3]};
#pragma unroll
    for (int nn = 0; nn < NBH; ++nn)
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const float sj = s_inv[nn] + (float)j;
        const HL2 r0 = split4_mul(fa0, f32x2{sj, sj}, fa1, f32x2{sj, sj});
        const HL2 r1 = split4_mul(fa1, f32x2{sj, -sj}, fa0, f32x2{-sj, sj});
        asm volatile("" ::"v"(r0.h0), "v"(r0.h1), "v"(r0.l0), "v"(r0.l1), "v"(r1.h0), "v"(r1.h1), "v"(r1.l0), "v"(r1.l1));
      }
  }
#endif
  if constexpr (MODE == LL) __syncthreads();      // s_pair lives in wave 0's T image: every wave is done reading it before T tiles land there
  const unsigned tl_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)(s_T) + (unsigned)(q * TIMG_B);
  const unsigned dummy_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)(s_T) + (unsigned)(4 * TIMG_B);
  StageH<KIND, MODE, HALF, S0, 0>::run(wrs, xrs, gstride4, et, tid, s_X, q, lane, hid, tl_base, dummy_base, s_S, acc);
}

// every pass of one conv of the (possibly fused) launch; `hs` = 2^(S - w) of this conv
template <int KIND, int MODE, bool HALFOK, bool S0 = false>
__device__ __forceinline__ void conv_f16(const DpConvArgs& a, const Grp& grp, int tid, float hs, float* s_T, float* s_S, float* s_X, float* s_inv,
                                         f32x4 (&acc)[n_acc4<KIND>()]) {
  const Wg r = wg_of<MODE>(a, grp.pose);
  __syncthreads();                                 // the previous conv's GEMM1 is done with s_inv
  if (tid < NBH) {
    const int node = grp.n_lo + (tid < grp.n_loc ? tid : 0);
    const int deg = MODE == LL ? a.rowend[node] - a.rowptr[node] : r.En;
    s_inv[tid] = hs / (W1_SCALE * (float)max(deg, 1));      // scatter MEAN (torch_scatter: / max(count, 1)) x the conv's power-of-two scale
  }
  const int n_et = (r.En + 31) >> 5;
  // buffer resources: the POSE's rows of the gathered-feature table (pharmacophore nodes for phore->ligand, ligand atoms otherwise; an edge
  // slot past the pose's last node reads 0) and the two-limb edge-embedding image (+ its flag row)
  const int g_row0 = MODE == DP_EDGES_P2L ? r.p0 : r.l0;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.gather_tab + (size_t)g_row0 * a.gather_stride, (unsigned)r.En * (unsigned)a.gather_stride * 4u);
  const __amdgpu_buffer_rsrc_t ers = make_rsrc(a.edge_emb_bf, (unsigned)(a.emb_flag_row + 1) * (unsigned)(EMBH_ROW * 2));
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w2n, (unsigned)w2h_quads<KIND>() * 16u);          // the conv's W2 image (GEMM-B's A fragments)
#pragma unroll 1
  for (int et = 0; et < n_et; ++et) {
    // HALFOK = false: the lean instantiation for launches whose poses never end on a short pass (FusedArgs.lean, from DpConvArgs.no_tail): half the
    // code, fewer registers parked across it; on such a batch the two instantiations are bit-identical
    if constexpr (HALFOK) {
      if (r.En - 32 * et <= 16) pass_f16<KIND, MODE, true, S0>(a, xrs, ers, wrs, grp, r, et, tid, s_T, s_S, s_X, s_inv, acc);
      else pass_f16<KIND, MODE, false, S0>(a, xrs, ers, wrs, grp, r, et, tid, s_T, s_S, s_X, s_inv, acc);
    } else {
      pass_f16<KIND, MODE, false, S0>(a, xrs, ers, wrs, grp, r, et, tid, s_T, s_S, s_X, s_inv, acc);
    }
  }
}

__device__ __forceinline__ float w2_exponent(const DpConvArgs& a) { return *reinterpret_cast<const float*>(a.w2n); }

// S0K: every conv of the launch is s0-only (DpConvArgs.s0_only; cross convs only) - its own instantiation, launched on its own: inlined beside the
// full conv behind a run-time branch the s0 passes change that function's register allocation (78 -> 150 spilled registers in the full conv's
// passes at layer 3, step -3 %), behind a real call the accumulators go to scratch (step -25 %): profiles/r6_experiments.md section 2
template <int KIND, bool HAS_LL, int XMODE, bool HALFOK, bool S0K = false>
#ifndef DP_WG_PER_CU
#define DP_WG_PER_CU 2            // 1 = timing experiment: one wave per SIMD with the whole 512-register file (r5_experiments.md section 7)
#endif
__global__ __launch_bounds__(256, DP_WG_PER_CU) void tp_conv_node_f16_kernel(FusedArgs fa) {
  __shared__ __attribute__((aligned(128))) float s_T[4 * TIMG_F + TDUM_B / 4];   // wave-private T images (fp16, hi | lo node slots) + the dummy column; reused for the reduction
  __shared__ __attribute__((aligned(16))) float s_S[NBH * SCOMP * 32];         // per-edge sh factors of the current pass (fp32)
  __shared__ __attribute__((aligned(16))) float s_X[32 * XROW];               // the pass's gathered rows, one half of the input columns at a time
  __shared__ float s_inv[NBH];
  using S = Spec<KIND>;
  static_assert(4 * NB * S::D_OUT <= 4 * TIMG_F, "reduction scratch fits in the T images");
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  // conv split (small batches: FusedArgs.conv_split, as in the other node-side kernels): blockIdx = conv * groups + group, this workgroup runs ONE
  // conv of the launch and leaves its un-scaled sum in part[conv][node]; shifts, residual and the sum over convs are dp_conv_node_combine's
  const int groups = fa.conv[0].B * fa.conv[0].slices * 2;
  const int bid = fa.conv_split ? (int)blockIdx.x % groups : (int)blockIdx.x;
  const int cx_only = fa.conv_split ? (int)blockIdx.x / groups : -1;
  const Grp grp = grp_of8(fa.conv[0], bid);
  if (grp.n_loc <= 0) return;
  f32x4 acc[n_acc4<KIND>()];
#pragma unroll
  for (int i = 0; i < n_acc4<KIND>(); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the launch's common exponent S = min over its convs of w + A0: every conv accumulates out * 2^S
  const int n_conv = (HAS_LL ? 1 : 0) + fa.n_x;
  float w_min = w2_exponent(fa.conv[0]);
  for (int k = 1; k < n_conv; ++k) w_min = fminf(w_min, w2_exponent(fa.conv[k]));
  const float s_exp = w_min + (float)A0;
  // what the epilogue adds to the accumulators: BatchNorm shifts, the residual / previous output, the listed pairs of an s0-only conv.  Read BEHIND
  // the passes (in front of them, to hide the list term's dependent loads, the s0-only kernels ran 8 % longer: 163.6k vs 165.8k pose-steps/s)
  constexpr int EP_ROUNDS = (NBH * S::D_OUT + 255) / 256;
  float er[EP_ROUNDS], eb[EP_ROUNDS];
  auto load_ep = [&]() {
    const DpConvArgs& a = fa.conv[0];
#pragma unroll
    for (int it = 0; it < EP_ROUNDS; ++it) {
      const int i = tid + 256 * it;
      const bool live = i < grp.n_loc * S::D_OUT;
      const int n = i / S::D_OUT, ch = i - n * S::D_OUT;
      const int node = grp.n_lo + (live ? n : 0);
      float sh = 0.f;
      for (int k = 0; k < (cx_only < 0 ? n_conv : 0); ++k) sh += fa.conv[k].bn_shift[live ? ch : 0];
      if constexpr (S0K) {
        // the conv's listed pairs (DpConvArgs.list_*: messages of a DP_EDGES_LIST launch): scatter mean over the conv's full degree, their BatchNorm scale
        for (int k = 0; k < (cx_only < 0 ? n_conv : 0); ++k) {
          const DpConvArgs& ck = fa.conv[k];
          if (ck.list_msg == nullptr || !live) continue;
          const int k0 = ck.list_rowptr[node], k1 = ck.list_rowptr[node + 1];
          if (k0 == k1) continue;
          float sm = 0.f;
          for (int kk = k0; kk < k1; ++kk) sm += ck.list_msg[(size_t)kk * S::D_OUT + ch];
          const int deg = wg_of<XMODE>(ck, grp.pose).En;
          sh += (sm / (float)max(deg, 1)) * ck.list_scale[ch];
        }
      }
      eb[it] = sh;
      const float* src = a.accumulate ? a.out + (size_t)node * a.out_stride + (live ? ch : 0)
                                      : (a.resid != nullptr && live && ch < a.resid_width ? a.resid + (size_t)node * a.resid_stride + ch : nullptr);
      er[it] = (live && src != nullptr && cx_only < 0) ? *src : 0.f;
    }
  };
  if constexpr (HAS_LL) {
    if (cx_only <= 0) conv_f16<KIND, LL, HALFOK>(fa.conv[0], grp, tid, ldexpf(1.f, (int)(s_exp - w2_exponent(fa.conv[0]))), s_T, s_S, s_X, s_inv, acc);
  }
  {
    constexpr int X0 = HAS_LL ? 1 : 0;
    const int c_lo = cx_only < 0 ? X0 : (cx_only < X0 ? n_conv : cx_only);
    const int c_hi = cx_only < 0 ? n_conv : (c_lo < n_conv ? c_lo + 1 : n_conv);
#pragma unroll 1
    for (int cx = c_lo; cx < c_hi; ++cx) {
      const DpConvArgs& a = fa.conv[cx];
      conv_f16<KIND, XMODE, HALFOK, S0K>(a, grp, tid, ldexpf(1.f, (int)(s_exp - w2_exponent(a))), s_T, s_S, s_X, s_inv, acc);
    }
  }
  // ---- sum the four k-partials and the two limb halves of the node slots, un-scale, BatchNorm shift, residual stream ----------------------
  load_ep();
  {
    const DpConvArgs& a = fa.conv[0];
    const float oscale = ldexpf(1.f, -(int)s_exp);
    __syncthreads();
    AccStore<KIND, 0>::run(acc, s_T + q * (NB * S::D_OUT), lane);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EP_ROUNDS; ++it) {
      const int i = tid + 256 * it;
      if (i >= grp.n_loc * S::D_OUT) continue;
      const int n = i / S::D_OUT, ch = i - n * S::D_OUT;
      constexpr int WS = NB * S::D_OUT, LO = NBH * S::D_OUT;        // wave stride; offset of the lo-limb node slots
      const float lo = ((s_T[LO + i] + s_T[WS + LO + i]) + s_T[2 * WS + LO + i]) + s_T[3 * WS + LO + i];
      const float hi = ((s_T[i] + s_T[WS + i]) + s_T[2 * WS + i]) + s_T[3 * WS + i];
      if (cx_only < 0) a.out[(size_t)(grp.n_lo + n) * a.out_stride + ch] = er[it] + ((hi + lo) * oscale + eb[it]);
      else fa.part[(size_t)cx_only * fa.part_stride + (size_t)(grp.n_lo - fa.node0 + n) * a.out_stride + ch] = (hi + lo) * oscale;
    }
  }
}

template <int KIND, bool HALFOK> int launch_f16_h(const FusedArgs& fa, int xmode, hipStream_t st) {
  const DpConvArgs& a = fa.conv[0];
  const dim3 grid((unsigned)(a.B * a.slices * 2 * (fa.conv_split ? fa.n_ll + fa.n_x : 1))), block(256);
  int n_s0 = 0;
  for (int i = 0; i < fa.n_ll + fa.n_x; ++i) n_s0 += fa.conv[i].s0_only ? 1 : 0;
  if (n_s0) {                                     // s0-only convs: their own instantiation, never mixed with full convs in one launch
    if (fa.n_ll || n_s0 != fa.n_x) return DP_E_UNSUPPORTED;
    if (xmode == DP_EDGES_P2L) hipLaunchKernelGGL((tp_conv_node_f16_kernel<KIND, false, DP_EDGES_P2L, HALFOK, true>), grid, block, 0, st, fa);
    else hipLaunchKernelGGL((tp_conv_node_f16_kernel<KIND, false, DP_EDGES_L2P, HALFOK, true>), grid, block, 0, st, fa);
    return (int)hipGetLastError();
  }
  if (fa.n_ll) hipLaunchKernelGGL((tp_conv_node_f16_kernel<KIND, true, DP_EDGES_P2L, HALFOK>), grid, block, 0, st, fa);
  else if (xmode == DP_EDGES_P2L) hipLaunchKernelGGL((tp_conv_node_f16_kernel<KIND, false, DP_EDGES_P2L, HALFOK>), grid, block, 0, st, fa);
  else hipLaunchKernelGGL((tp_conv_node_f16_kernel<KIND, false, DP_EDGES_L2P, HALFOK>), grid, block, 0, st, fa);
  return (int)hipGetLastError();
}
// the lean instantiation when every conv of the launch carries DpConvArgs.no_tail (a hint: the general kernel is always correct)
template <int KIND> int launch_f16(const FusedArgs& fa, int xmode, hipStream_t st) {
  return fa.lean ? launch_f16_h<KIND, false>(fa, xmode, st) : launch_f16_h<KIND, true>(fa, xmode, st);
}

}  // namespace dpf

int dpf_launch_k0(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpf_launch_k1(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpf_launch_k2(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpf_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
