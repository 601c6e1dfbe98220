// dp_conv_node_bf16_impl.h - the node-side formulation of the fused tensor-product convolution (dp_conv_node_impl.h) with bf16
// OPERANDS on the bf16 matrix pipe of gfx950 (v_mfma_f32_16x16x32_bf16: 16x the fp32-input MFMA rate), fp32 accumulate.
//
// Replaces TensorProductConvLayer.forward (/root/reference/src/models/score_model_phore.py:134-149) for BASELINE.json configs[4]
// ("bf16 mixed precision": bf16 radial MLP / weights, fp32 accumulate, fp32 geometry - SURVEY.md 8d).  What is bf16 here: the radial
// MLP's two layers (W1, W2 and their activations), the edge embedding and node scalars fed to layer 1, the tensor-product operand
// z = (gathered features x spherical-harmonic factors) - computed in fp32, rounded once - and T = hid^T z between the two GEMMs.
// What stays fp32: positions, distances, spherical harmonics, edge embeddings' own MLPs, every accumulation, BatchNorm shift, residual.
//
// Same association as the fp32 kernel:
//     T_p[n][k][u,m]  = sum_{e in n} hid[e][k] z_p[e][u,m]            GEMM-A: ONE 16x16x32 MFMA per (node, 15-column tile, 32-edge pass)
//     out[n][w,m]    += sum_{k,u} W2_p[(u,w)][k] T_p[n][k][u,m]        GEMM-B: K = 32 = two in-channels x the wave's 16 hidden rows
// and the same mapping: a workgroup = 4 waves = one group of <= 16 scatter nodes of one pose, wave q owns hidden rows 16q..16q+15, so
// GEMM1 -> GEMM-A -> LDS transpose -> GEMM-B is wave-private; the four k-partials meet in LDS once at the end.
//
// What changes with the 16x pipe - the kernel is no longer bound by the matrix pipe but by the VALU that builds z and by the LDS:
//   * GEMM1 is one MFMA per (node, 16-edge tile): A = a 64-byte bf16 row of the edge embedding (written by the edge-embedding kernels
//     beside the fp32 one; feature 20 is the "edge does not exist" flag whose W1 row is -1e30), B = the wave's 16 columns of W1.
//     The node terms u_s / u_g are two more MFMAs per pass, as before.  hid leaves the accumulator as 8 bf16 per lane = GEMM-A's A
//     operand (64 registers for the 16 nodes instead of 128).
//   * column tiles are processed in PAIRS (a "supertile" = 10 in-channels: 2 x (5 in-channels x 3 m), or the 10 in-channels of a
//     scalar-output tile): both tiles of a pair take the same SH factors per lane, so those are read from LDS once per (node, pair),
//     and GEMM-B's K = 32 is exactly two in-channels (5 K-chunks per supertile).
//   * T crosses from "columns on lanes" to "nodes on lanes" as bf16 through a wave-private LDS image [column][k quad][node][4] with 520-byte
//     columns (512 + 8: the 16 lanes of a ds_write_b64 - one column each - fall into 16 different 8-byte bank units without any swizzle
//     arithmetic, the node is an immediate offset): one ds_write_b64 per (node, tile), two ds_read_b64 per MFMA.
//   * W2 is packed offline as bf16 A fragments (16 bytes per lane and MFMA), BatchNorm scale folded in.
#pragma once
#include "dp_conv_node_impl.h"

#ifndef DP_BABL     // timing ablations of THIS kernel (tools/build_variant.sh; results are wrong by construction): 1 no gathered-operand loads,
#define DP_BABL 0   // 2 no SH-factor reads, 3 no z arithmetic, 4 no GEMM-A MFMAs, 5 no T stores, 6 no T reads, 7 no GEMM-B MFMAs, 8 no W2 loads,
#endif              // 9 no edge-embedding loads (GEMM1), 10 one workgroup per CU
#ifdef DP_PROF     // cycle stamps (tools/build_variant.sh prof "-DDP_PROF" dp_conv_node_bf16_k3.hip): per translation unit, read through dp_debug_prof_node
static __device__ unsigned long long dpb_prof[16];
#endif

namespace dpb {
using namespace dpn;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(4)));
typedef u32x2 __attribute__((address_space(3))) lds_u32x2;
__device__ __forceinline__ lds_u32x2* lds_ptr(unsigned a) { return (lds_u32x2*)(__UINTPTR_TYPE__)a; }      // LDS byte address -> pointer

constexpr int SUPER_NU = 10;          // in-channels per supertile
constexpr int SUPER_CH = 5;           // GEMM-B K chunks (two in-channels x 16 hidden rows) per supertile
constexpr int EMB_ROW = 32;           // bf16 per row of the bf16 edge-embedding table (20 features | flag | padding) = 64 bytes
constexpr int EMB_FLAG = 20;          // feature slot of the "edge does not exist" flag
constexpr int COLB = 520;             // bytes per T column: 4 k quads x 16 nodes x 4 bf16 = 512, + 8: the 16 lanes of a ds_write_b64 (one column each,
                                      // same node, same k quad) then fall into 16 different 8-byte bank units - no swizzle arithmetic per store
constexpr int TIMG_B = 15616;         // bytes per wave-private T image: 30 columns x 520, rounded up to 64
constexpr int TIMG_F = TIMG_B / 4;
constexpr int TDUM_B = 640;           // where the lanes without a column store (4 x 128 + 15 x 8 bytes are touched)
static_assert(2 * TCOLS * COLB <= TIMG_B && TIMG_B % 64 == 0, "T image holds a supertile");

template <int KIND> constexpr int st_count(int p) { return Spec<KIND>::P[p].mul_in / SUPER_NU; }
template <int KIND> constexpr int w2b_path_off(int p) {             // in 16-byte quads
  int o = 0;
  for (int i = 0; i < p; ++i) o += st_count<KIND>(i) * SUPER_CH * path_mt<KIND>(i) * 4 * 64;
  return o;
}
template <int KIND> constexpr int w2b_quads() { return w2b_path_off<KIND>(Spec<KIND>::NPATH); }
constexpr int W1B_QUADS = 3 * 4 * 64;                                // three images (edge embedding, attr1 block, attr2 block) x 4 waves
template <int KIND> constexpr bool supertiles_ok() {
  for (int i = 0; i < Spec<KIND>::NPATH; ++i)
    if (Spec<KIND>::P[i].mul_in % SUPER_NU != 0) return false;
  return true;
}
static_assert(supertiles_ok<0>() && supertiles_ok<1>() && supertiles_ok<2>() && supertiles_ok<3>(), "every path has a multiple of 10 in-channels");

__device__ __forceinline__ unsigned pk2(float lo, float hi) {       // v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
  return __builtin_bit_cast(unsigned, v);
}
// ReLU as ONE instruction (v_med3_f32 x, 0, +inf): fmaxf(x, 0) compiles to two (a canonicalising v_max x, x first)
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// column of in-channel `ul` (0..9 inside the supertile), component m, in the wave's T image
template <bool VEC> constexpr int t_col(int ul, int m) { return VEC ? (ul / 5) * TCOLS + m * 5 + (ul % 5) : ul; }

template <int ZT> constexpr int term_sign(int t) { return (ZT == Z111 && t == 1) ? -1 : 1; }        // (term_of(t, m).sign, which does not depend on m)

// the gathered-feature table as a buffer resource: one VGPR offset per (lane, tile), the eight edges of a lane through SGPR offsets, rows past
// the table's end read as 0 (no clamping arithmetic, no 64-bit addresses)
struct XSrc {
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff_row;       // byte offset of the row gathered by this lane group's first edge slot of the pass
  unsigned soff[8];        // byte offsets of the lane's other edge slots relative to it (wave-uniform)
};
// (base and size are wave-uniform, but they come out of a dynamically indexed kernel-argument array, i.e. through VGPRs: readfirstlane makes the
// descriptor provably scalar - without it every buffer load is wrapped in a waterfall loop)
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)(__UINTPTR_TYPE__)base;
  const unsigned long long pu = ((unsigned long long)uni((unsigned)(p >> 32)) << 32) | uni((unsigned)p);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(__UINTPTR_TYPE__)pu, 0, (int)uni(bytes), 0x00020000);
}

// gathered operands of supertile (path P, st) for this lane's 8 (HALF: 4) edges, both tiles, as pairs of consecutive edge slots (the operand
// layout of v_pk_*_f32); requested during the previous supertile's GEMM-B.  No masking: an edge slot that does not exist lies past the end of
// the pose-limited buffer resource and reads 0 (and has a zero hidden vector), a lane without a column stores to the dummy column.
template <int KIND, int MODE, bool HALF, int P>
__device__ __forceinline__ void load_x2(const XSrc& xs, int lane_l, int st, f32x2 (&xr)[2][3][4]) {
  using S = Spec<KIND>;
  (void)MODE;
  if constexpr (P < S::NPATH) {
    constexpr PathC pc = S::P[P];
    constexpr int ZT = pc.ztype;
    constexpr bool VEC = d_out_of(ZT) == 3;
    constexpr int DIN = d_in_of(ZT), NTERM = n_terms<ZT>(), NTILE = VEC ? 2 : 1, NJ = HALF ? 4 : 8;
    // (scalar dword loads only: with this toolchain a raw_buffer_load_b64 / b96 / b128 whose result is read element by element is narrowed to ONE
    //  dword - the other components come back undefined; the three components of a 3-term path therefore travel as three loads with
    //  immediate offsets, which cost no address arithmetic either)
    const int c = lane_l & 15;
    const int m = VEC ? (c >= 10 ? 2 : (c >= 5 ? 1 : 0)) : 0;
    const int ul = VEC ? c - 5 * m : c;
    const bool colv = VEC ? c < 15 : c < 10;
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) {
      const int u = colv ? SUPER_NU * st + 5 * tl + ul : 0;
      const unsigned voff = xs.voff_row + (unsigned)((pc.in_off + u * DIN) * 4);
      float v[3][8];
      unsigned vt[NTERM];
#pragma unroll
      for (int t = 0; t < NTERM; ++t) vt[t] = voff + (unsigned)(term_of<ZT>(t, m).xcomp * 4);       // (xcomp = t for the dot / l=2 paths: an immediate)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int t = 0; t < NTERM; ++t) {
#if DP_BABL == 1 || (defined(DPF_ABL) && DPF_ABL == 1)
          v[t][j] = __builtin_bit_cast(float, vt[t] + xs.soff[j]);
#else
          v[t][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xs.rs, vt[t], xs.soff[j], 0));
#endif
        }
#pragma unroll
      for (int t = 0; t < NTERM; ++t)
#pragma unroll
        for (int jp = 0; jp < NJ / 2; ++jp) xr[tl][t][jp] = f32x2{v[t][2 * jp], v[t][2 * jp + 1]};
    }
  }
}

// S0 (DpConvArgs.s0_only, as in dp_conv_node_f16_impl.h): the conv's harmonics are e0 on every pair the launch covers - only the paths through the
// l = 0 component (Z000 / Z101: the operand is the gathered input itself) are evaluated; the path after P that such a launch runs next:
template <int KIND, bool S0> constexpr int next_path(int p) {
  for (int i = p + 1; i < Spec<KIND>::NPATH; ++i)
    if (!(S0 && needs_s_rt(Spec<KIND>::P[i].ztype))) return i;
  return Spec<KIND>::NPATH;
}
template <int KIND, int MODE, bool HALF, bool S0, int P>
struct StageB {
  static __device__ __forceinline__ void run(__amdgpu_buffer_rsrc_t wrs, const XSrc& xs, int q, int lane, const u32x4 (&hid)[NB],
                                             unsigned tl_base, unsigned dummy_base, const float* sS, f32x4 (&acc)[n_acc4<KIND>()],
                                             f32x2 (&xr)[2][3][4], NProfT& prof_prev) {
    using S = Spec<KIND>;
    if constexpr (P < S::NPATH && S0 && needs_s_rt(S::P[P < S::NPATH ? P : 0].ztype)) {
      StageB<KIND, MODE, HALF, S0, P + 1>::run(wrs, xs, q, lane, hid, tl_base, dummy_base, sS, acc, xr, prof_prev);
    } else if constexpr (P < S::NPATH) {
      constexpr PathC pc = S::P[P];
      constexpr int ZT = pc.ztype;
      constexpr bool VEC = d_out_of(ZT) == 3;
      constexpr int NG = VEC ? 3 : 1, NTERM = n_terms<ZT>(), MT = path_mt<KIND>(P), AB = acc_base<KIND>(P), NTILE = VEC ? 2 : 1;
      constexpr int NJ = HALF ? 4 : 8, NT2 = HALF ? 1 : 2;
#pragma unroll 1
      for (int st = 0; st < st_count<KIND>(P); ++st) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));          // lane-derived constants are rebuilt per supertile (keeps LICM from hoisting and spilling them)
        const int g = lane_l >> 4, c = lane_l & 15;
        const int m = VEC ? (c >= 10 ? 2 : (c >= 5 ? 1 : 0)) : 0;
        const bool colv = VEC ? c < 15 : c < 10;
        // ---- gathered operand of the lane's column in both tiles (the same rows for every node of the group), as edge-slot pairs ------
        f32x2 xa[NTILE][NTERM][NJ / 2];
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl)
#pragma unroll
          for (int t = 0; t < NTERM; ++t)
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) xa[tl][t][jp] = xr[tl][t][jp];
        // T image: byte address of (column, k quad g, node nn) = column * 520 + g * 128 + nn * 8: the node is an immediate offset
        unsigned twr[NTILE];
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl)
          twr[tl] = (colv ? tl_base + (unsigned)((tl * TCOLS + c) * COLB) : dummy_base + (unsigned)(c * 8)) + (unsigned)(g * 128);
        // the supertile's W2 fragments (GEMM-B's A operands, 5 K chunks x MT) are requested NOW: a chunk is only 3-6 MFMAs of 16 cycles long, an L2
        // round trip under load several hundred cycles - requested inside GEMM-B they would stall every chunk; GEMM-A hides them
        // (through a buffer resource: one VGPR offset per lane, the fragment's offset is wave-uniform - no 64-bit address arithmetic per load)
        const unsigned wvoff = (unsigned)(q * 64 + lane_l) * 16u;
        const unsigned wsoff = uni((unsigned)(w2b_path_off<KIND>(P) + st * (SUPER_CH * MT * 256)) * 16u);
        u32x4 wall[SUPER_CH][MT];
#pragma unroll
        for (int ch = 0; ch < SUPER_CH; ++ch)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#if DP_BABL == 8
            wall[ch][mt] = u32x4{wvoff, wsoff, (unsigned)ch, (unsigned)mt};
#else
            wall[ch][mt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, wsoff + (unsigned)((ch * MT + mt) * 256 * 16), 0);
#endif
          }
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM-A: one MFMA per (node, tile); T leaves as bf16 ----------------------------------------------------------------
        // Software pipeline over the nodes: node nn's MFMAs are issued into a ring of two accumulator sets and their results are
        // converted / stored one node LATER (behind the next node's z arithmetic), so no VALU instruction waits on the MFMA it follows;
        // the SH factors of node nn + 1 are requested before node nn's arithmetic.  sched_barrier keeps the nodes apart (hoisting every
        // LDS read to the top spills).
        f32x4 Dq[2][NTILE];
        auto store_t = [&](int nn) {
#pragma unroll
          for (int tl = 0; tl < NTILE; ++tl) {
            const f32x4 D = Dq[nn & 1][tl];
#if DP_BABL == 5
            asm volatile("" ::"v"(pk2(D[0], D[1])), "v"(pk2(D[2], D[3])), "v"(twr[tl] + (unsigned)(nn << 3)));
#else
            *lds_ptr(twr[tl] + (unsigned)(nn << 3)) = u32x2{pk2(D[0], D[1]), pk2(D[2], D[3])};
#endif
          }
        };
        if constexpr (needs_s<ZT>()) {
          const char* sbase = reinterpret_cast<const char*>(sS);
          int soff[NTERM];
#pragma unroll
          for (int t = 0; t < NTERM; ++t) soff[t] = (term_of<ZT>(t, m).scomp * 16 + 4 * g) * 4;
          auto loadS = [&](int nn, f32x4 (&sv)[NTERM][NT2]) {
#pragma unroll
            for (int t = 0; t < NTERM; ++t)
#pragma unroll
              for (int t2 = 0; t2 < NT2; ++t2)
#if DP_BABL == 2
                sv[t][t2] = f32x4{(float)soff[t], (float)nn, 1.f, 2.f};
#else
                sv[t][t2] = *reinterpret_cast<const f32x4*>(sbase + (2 * nn + t2) * (SCOMP * 64) + soff[t]);
#endif
          };
          // factor registers: double-buffered (node nn + 1 requested before node nn's arithmetic) except for the three-term vector paths,
          // where the second set would not fit beside the two tiles' gathered operands: there the next node's factors are requested as
          // soon as this node's z values exist
          constexpr int SD = (VEC && NTERM == 3) ? 1 : 2;
          f32x4 sv[SD][NTERM][NT2];
          loadS(0, sv[0]);
#pragma unroll
          for (int nn = 0; nn < NB; ++nn) {
            const int cur = SD == 2 ? (nn & 1) : 0;
            if constexpr (SD == 2) {
              if (nn + 1 < NB) loadS(nn + 1, sv[(nn + 1) & 1]);
            }
            u32x4 zb[NTILE];
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) {
              f32x2 z[NJ / 2];
#pragma unroll
              for (int jp = 0; jp < NJ / 2; ++jp) {
#if DP_BABL == 3
                z[jp] = xa[tl][0][jp];
                if (jp == 0) asm volatile("" ::"v"(sv[cur][NTERM - 1][0]), "v"(sv[cur][0][NT2 - 1]));
                continue;
#endif
                // edge slots 2 jp, 2 jp + 1 of the lane = components (2 jp & 3, + 1) of the factor quad of 16-edge tile jp >> 1
                auto spair = [&](int t) { const f32x4 q4 = sv[cur][t][jp >> 1]; return (jp & 1) ? f32x2{q4[2], q4[3]} : f32x2{q4[0], q4[1]}; };
                f32x2 zz = xa[tl][0][jp] * spair(0);
#pragma unroll
                for (int t = 1; t < NTERM; ++t) {
                  if (term_sign<ZT>(t) < 0) zz = __builtin_elementwise_fma(-xa[tl][t][jp], spair(t), zz);
                  else zz = __builtin_elementwise_fma(xa[tl][t][jp], spair(t), zz);
                }
                z[jp] = zz;
              }
              zb[tl] = u32x4{pk2(z[0][0], z[0][1]), pk2(z[1][0], z[1][1]), 0u, 0u};
              if constexpr (!HALF) { zb[tl][2] = pk2(z[NJ / 2 - 2][0], z[NJ / 2 - 2][1]); zb[tl][3] = pk2(z[NJ / 2 - 1][0], z[NJ / 2 - 1][1]); }
            }
            if constexpr (SD == 1) {
              __builtin_amdgcn_sched_barrier(0);
              if (nn + 1 < NB) loadS(nn + 1, sv[0]);
            }
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) {
#if DP_BABL == 4
              asm volatile("" ::"v"(hid[nn]), "v"(zb[tl]));
              Dq[nn & 1][tl] = f32x4{__builtin_bit_cast(float, zb[tl][0]), 1.f, 2.f, 3.f};
#else
              Dq[nn & 1][tl] = mfma_bf(as_bf(hid[nn]), as_bf(zb[tl]), f32x4{0.f, 0.f, 0.f, 0.f});
#endif
            }
            if (nn > 0) store_t(nn - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // z is the gathered input itself (sh_0 = 1): the B operand is the same for every node
          u32x4 zb[NTILE];
#pragma unroll
          for (int tl = 0; tl < NTILE; ++tl) {
            zb[tl] = u32x4{pk2(xa[tl][0][0][0], xa[tl][0][0][1]), pk2(xa[tl][0][1][0], xa[tl][0][1][1]), 0u, 0u};
            if constexpr (!HALF) { zb[tl][2] = pk2(xa[tl][0][NJ / 2 - 2][0], xa[tl][0][NJ / 2 - 2][1]); zb[tl][3] = pk2(xa[tl][0][NJ / 2 - 1][0], xa[tl][0][NJ / 2 - 1][1]); }
          }
#pragma unroll
          for (int nn = 0; nn < NB; ++nn) {
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) {
#if DP_BABL == 4
              asm volatile("" ::"v"(hid[nn]), "v"(zb[tl]));
              Dq[nn & 1][tl] = f32x4{__builtin_bit_cast(float, zb[tl][0]), 1.f, 2.f, 3.f};
#else
              Dq[nn & 1][tl] = mfma_bf(as_bf(hid[nn]), as_bf(zb[tl]), f32x4{0.f, 0.f, 0.f, 0.f});
#endif
            }
            if (nn > 0) store_t(nn - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        store_t(NB - 1);
        __builtin_amdgcn_sched_barrier(0);
        NPROF(2);
        // ---- GEMM-B: out[w][node] += W2[w][(u,k)] T[(u,k)][node], K = (two in-channels) x (the wave's 16 hidden rows) -----------------
        // chunk ch + 1's T fragments are read from LDS while chunk ch's MFMAs run; the next supertile's gathered operands are requested
        // at the start (they are consumed when that supertile begins: behind the W2 fragments already in flight, vector-memory returns are in order)
        {
          constexpr int PN = next_path<KIND, S0>(P);
          const unsigned rbase = tl_base + (unsigned)(g * 128 + c * 8);          // GEMM-B view: node c, k quad g; the column is an immediate offset
          auto readT = [&](int ch, u32x4 (&tb)[NG]) {
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
#if DP_BABL == 6
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
          if (st + 1 < st_count<KIND>(P)) load_x2<KIND, MODE, HALF, P>(xs, lane_l, st + 1, xr);
          else load_x2<KIND, MODE, HALF, PN>(xs, lane_l, 0, xr);
#pragma unroll
          for (int ch = 0; ch < SUPER_CH; ++ch) {
            if (ch + 1 < SUPER_CH) readT(ch + 1, tb[(ch + 1) & 1]);
#pragma unroll
            for (int grp = 0; grp < NG; ++grp)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
#if DP_BABL == 7
                asm volatile("" ::"v"(wall[ch][mt]), "v"(tb[ch & 1][grp]));
#else
                acc[AB + grp * MT + mt] = mfma_bf(as_bf(wall[ch][mt]), as_bf(tb[ch & 1][grp]), acc[AB + grp * MT + mt]);
#endif
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        NPROF(3);
      }
      StageB<KIND, MODE, HALF, S0, P + 1>::run(wrs, xs, q, lane, hid, tl_base, dummy_base, sS, acc, xr, prof_prev);
    }
  }
};

// one pass (32 edge slots per node; HALF: 16) of one conv: stage the SH factors, GEMM1 -> hid (bf16 A fragments), then the path stages
template <int KIND, int MODE, bool HALF, bool S0>
__device__ __forceinline__ void pass_bf16(const DpConvArgs& a, __amdgpu_buffer_rsrc_t xrs, __amdgpu_buffer_rsrc_t ers, __amdgpu_buffer_rsrc_t wrs, const Grp& grp, const Wg& r, int et, int tid,
                                          float* s_T, float* s_S, const float* s_inv, f32x4 (&acc)[n_acc4<KIND>()], NProfT& prof_prev) {
  const int lane = tid & 63, q = tid >> 6, g = lane >> 4, c = lane & 15;
  constexpr int NT2 = HALF ? 1 : 2;
  constexpr int NSTEP = NB * NT2;
  float* sT = s_T + q * TIMG_F;
  int* const s_pair = reinterpret_cast<int*>(s_T + TIMG_F - 2 * NB * 32);     // LL mode: [2][NB][32] edge slots of the pass's pairs
  (void)s_pair;
  __syncthreads();                               // readers of the previous pass's factors (and of s_pair / the T images) are done
  // ---- operands that do not depend on the staging: the wave's three W1 images and the pass's node rows ---------------------------
  const u32x4* w1 = reinterpret_cast<const u32x4*>(a.w1n);
  const u32x4 Wemb = w1[(0 * 4 + q) * 64 + lane], Wa1 = w1[(1 * 4 + q) * 64 + lane], Wa2 = w1[(2 * 4 + q) * 64 + lane];
  const u32x4 Wsc = MODE == DP_EDGES_L2P ? Wa2 : Wa1, Wga = MODE == DP_EDGES_L2P ? Wa1 : Wa2;      // attr block of the scatter / gather node
  // a node tile's A fragment: lane (row c, group g) holds features 8g..8g+7 of the row's first 20 (bf16), feature 20 = 1 carries the bias
  auto attr_frag = [&](const float* rowp, bool bias) {
    const int o = g < 2 ? 8 * g : 16;
    f32x4 lo = *reinterpret_cast<const f4u*>(rowp + o);
    f32x4 hi = *reinterpret_cast<const f4u*>(rowp + (g < 2 ? o + 4 : 16));
    if (g >= 2) hi = f32x4{(g == 2 && bias) ? 1.f : 0.f, 0.f, 0.f, 0.f};
    if (g == 3) lo = f32x4{0.f, 0.f, 0.f, 0.f};
    return u32x4{pk2(lo[0], lo[1]), pk2(lo[2], lo[3]), pk2(hi[0], hi[1]), pk2(hi[2], hi[3])};
  };
  u32x4 ag[NT2], as;
  {
    const float* s_tab = MODE == DP_EDGES_L2P ? a.attr2_tab : a.attr1_tab;
    const float* g_tab = MODE == DP_EDGES_L2P ? a.attr1_tab : a.attr2_tab;
    const int s_stride = MODE == DP_EDGES_L2P ? a.attr2_stride : a.attr1_stride, g_stride = MODE == DP_EDGES_L2P ? a.attr1_stride : a.attr2_stride;
#pragma unroll
    for (int t2 = 0; t2 < NT2; ++t2) {
      const int e0 = 32 * et + 16 * t2 + c;
      const int grow = (MODE == DP_EDGES_P2L ? r.p0 : r.l0) + (e0 < r.En ? e0 : 0);
      ag[t2] = attr_frag(g_tab + (size_t)grow * g_stride, false);
    }
    const int srow = grp.n_lo + (c < grp.n_loc ? c : 0);
    as = attr_frag(s_tab + (size_t)srow * s_stride, true);
  }
  // ---- stage the SH-derived factors of the pass (fp32) and, in the pair mode, the pass's pair -> edge-slot entries ----------------
  constexpr int SLOTS = HALF ? 16 : 32;
  static_assert(!(S0 && MODE == LL), "s0-only is a mode of the dense cross convs");
  if constexpr (!S0) {            // (s0-only: no path reads the factors)
#pragma unroll
  for (int idx = tid; idx < NB * SLOTS; idx += 256) {
    const int nn = idx / SLOTS, el = idx % SLOTS, e = 32 * et + el;
    bool ok = nn < grp.n_loc && e < r.En;
    int frow = 0;
    if constexpr (MODE == LL) {
      int sa = -1, sb = -1;
      if (ok) { sa = pair_slot<0>(a, r, grp.n_lo + nn, e); sb = pair_slot<1>(a, r, grp.n_lo + nn, e); }
      s_pair[nn * 32 + el] = sa;
      s_pair[NB * 32 + nn * 32 + el] = sb;
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
  NPROF(0);

  // ---- GEMM1: pre[e][k] = W1_emb emb[e] + u_s[scatter node] + u_g[gather node]; hid = relu(pre) / degree, as bf16 A fragments -----------
  f32x4 ug[NT2];
#pragma unroll
  for (int t2 = 0; t2 < NT2; ++t2) ug[t2] = mfma_bf(as_bf(ag[t2]), as_bf(Wga), f32x4{0.f, 0.f, 0.f, 0.f});     // rows = the pass's edge slots
  {
    const f32x4 us = mfma_bf(as_bf(as), as_bf(Wsc), f32x4{0.f, 0.f, 0.f, 0.f});                                   // rows = the group's nodes, + bias
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sT[(4 * g + rr) * 16 + c] = us[rr];             // [node][k]: read back as a broadcast over rows
  }
  // A fragment of (node nn, 16-edge tile t2): 16 bytes of the edge's bf16 embedding row; the flag row for an edge that does not exist
  auto load_a = [&](int step, int which) {
    int nn_o = HALF ? step : step >> 1;
    asm volatile("" : "+s"(nn_o));                // opaque: keeps the 16 nodes' row arithmetic from being hoisted out of the pass (spills)
    const int t2s = HALF ? 0 : step & 1;
    const int e0 = 32 * et + 16 * t2s + c;
    int frow;
    if constexpr (MODE == LL) {
      frow = s_pair[which * (NB * 32) + nn_o * 32 + 16 * t2s + c];
    } else {
      frow = (nn_o < grp.n_loc && e0 < r.En) ? feat_row<MODE>(r, grp.n_lo + nn_o, e0) : -1;
    }
    if (frow < 0) frow = a.emb_flag_row;
#if DP_BABL == 9
    return u32x4{(unsigned)frow, 0u, 0u, 7u};
#else
    return __builtin_amdgcn_raw_buffer_load_b128(ers, (unsigned)frow * (unsigned)(EMB_ROW * 2) + (unsigned)(16 * g), 0, 0);
#endif
  };
  constexpr int PF = 4;
  u32x4 ea[PF + 1], eb[PF + 1];
  (void)eb;
#pragma unroll
  for (int s0 = 0; s0 < PF; ++s0) {
    ea[s0] = load_a(s0, 0);
    if constexpr (MODE == LL) eb[s0] = load_a(s0, 1);
  }
  u32x4 hid[NB];
  f32x2 xr0[2][3][4];
  XSrc xs;
  xs.rs = xrs;
  const unsigned gstride4 = uni((unsigned)a.gather_stride * 4u);
  xs.voff_row = (unsigned)(32 * et + 4 * g) * gstride4;                         // relative to the pose's first gather row (the resource's base)
#pragma unroll
  for (int j = 0; j < 8; ++j) xs.soff[j] = (unsigned)(16 * (j >> 2) + (j & 3)) * gstride4;
  float hf[8];
  // step = (node, 16-edge tile).  The MFMA of step s is issued into a ring of two accumulators and its activation (ReLU, mean, bf16 pack)
  // happens one step later, so the VALU never waits on the MFMA it follows.
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
      const float inv = s_inv[nn];               // scatter MEAN (1 / degree) applied to the hidden vectors: T and the output are linear in them
      hid[nn] = u32x4{pk2(hf[0] * inv, hf[1] * inv), pk2(hf[2] * inv, hf[3] * inv), 0u, 0u};
      if constexpr (!HALF) { hid[nn][2] = pk2(hf[4] * inv, hf[5] * inv); hid[nn][3] = pk2(hf[6] * inv, hf[7] * inv); }
    }
  };
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int nn = HALF ? step : step >> 1, t2 = HALF ? 0 : step & 1;
    if (step + PF < NSTEP) {
      ea[(step + PF) % (PF + 1)] = load_a(step + PF, 0);
      if constexpr (MODE == LL) eb[(step + PF) % (PF + 1)] = load_a(step + PF, 1);
    }
    if (step == NSTEP - 3) load_x2<KIND, MODE, HALF, 0>(xs, lane, 0, xr0);          // first supertile's operands: behind the last emb request
    const float us_n = sT[nn * 16 + c];
    const f32x4 init = {us_n + ug[t2][0], us_n + ug[t2][1], us_n + ug[t2][2], us_n + ug[t2][3]};
    Dg[step & 1] = mfma_bf(as_bf(ea[step % (PF + 1)]), as_bf(Wemb), init);
    if constexpr (MODE == LL) {
      // the bond edges of the same pairs (1-4 per atom): second layer-1 evaluation over the same u_s + u_g, skipped per 16-pair tile when
      // none of them is bonded; the pair's two hidden vectors add (same geometry, same gathered row).  An absent bond edge reads the
      // flag row: its slot 20 (lane group 2, third dword, low half) is 1.0
      const u32x4 b4 = eb[step % (PF + 1)];
      has2[step & 1] = __builtin_amdgcn_ballot_w64(g == 2 && (b4[2] & 0xFFFFu) == 0u) != 0;
      if (has2[step & 1]) Dg2[step & 1] = mfma_bf(as_bf(b4), as_bf(Wemb), init);
    }
    if (step > 0) finish(step - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  finish(NSTEP - 1);
  if constexpr (MODE == LL) __syncthreads();      // s_pair lives in wave 0's T image: every wave is done reading it before T tiles land there
  const unsigned tl_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)(s_T) + (unsigned)(q * TIMG_B);
  const unsigned dummy_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)(s_T) + (unsigned)(4 * TIMG_B);
  NPROF(1);
  StageB<KIND, MODE, HALF, S0, 0>::run(wrs, xs, q, lane, hid, tl_base, dummy_base, s_S, acc, xr0, prof_prev);
}

// every pass of one conv of the (possibly fused) launch
template <int KIND, int MODE, bool HALFOK, bool S0 = false>
__device__ __forceinline__ void conv_bf16(const DpConvArgs& a, const Grp& grp, int tid, float* s_T, float* s_S, float* s_inv,
                                          f32x4 (&acc)[n_acc4<KIND>()], NProfT& prof_prev) {
  const Wg r = wg_of<MODE>(a, grp.pose);
  __syncthreads();                                 // the previous conv's GEMM1 is done with s_inv
  if (tid < NB) {
    const int node = grp.n_lo + (tid < grp.n_loc ? tid : 0);
    const int deg = MODE == LL ? a.rowend[node] - a.rowptr[node] : r.En;
    s_inv[tid] = 1.f / (float)max(deg, 1);         // scatter MEAN (torch_scatter: / max(count, 1)), applied to the hidden vectors
  }
  const int n_et = (r.En + 31) >> 5;
  // buffer resources: the POSE's rows of the gathered-feature table (pharmacophore nodes for phore->ligand, ligand atoms otherwise: an edge
  // slot past the pose's last node reads 0 - never the next pose's rows, whose NaN would survive the multiplication by a zero hidden vector
  // and leak a diverged neighbour into this pose) and the bf16 edge-embedding image (+ its flag row)
  const int g_row0 = MODE == DP_EDGES_P2L ? r.p0 : r.l0;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.gather_tab + (size_t)g_row0 * a.gather_stride, (unsigned)r.En * (unsigned)a.gather_stride * 4u);
  const __amdgpu_buffer_rsrc_t ers = make_rsrc(a.edge_emb_bf, (unsigned)(a.emb_flag_row + 1) * (unsigned)(EMB_ROW * 2));
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w2n, (unsigned)w2b_quads<KIND>() * 16u);
#pragma unroll 1
  for (int et = 0; et < n_et; ++et) {
    // HALFOK = false: the lean instantiation for launches in which no pose ends on a pass of <= 16 slots (DpConvArgs.no_tail on every conv, as
    // for the other node-side kernels): without the half-pass body the kernel spills less
    if constexpr (HALFOK) {
      if (r.En - 32 * et <= 16) { pass_bf16<KIND, MODE, true, S0>(a, xrs, ers, wrs, grp, r, et, tid, s_T, s_S, s_inv, acc, prof_prev); continue; }
    }
    pass_bf16<KIND, MODE, false, S0>(a, xrs, ers, wrs, grp, r, et, tid, s_T, s_S, s_inv, acc, prof_prev);
  }
}

// S0K: every conv of the launch is s0-only (its own instantiation and launch, as for the f16x3 kernel)
template <int KIND, bool HAS_LL, int XMODE, bool HALFOK, bool S0K = false>
#ifndef DP_WG_PER_CU
#define DP_WG_PER_CU 2            // 1 = timing experiment: one wave per SIMD with the whole 512-register file (r5_experiments.md section 7)
#endif
__global__ __launch_bounds__(256, DP_WG_PER_CU) void tp_conv_node_bf16_kernel(FusedArgs fa) {
  __shared__ __attribute__((aligned(128))) float s_T[4 * TIMG_F + TDUM_B / 4];   // wave-private T images (bf16) + the dummy column; reused for the reduction
  __shared__ __attribute__((aligned(16))) float s_S[NB * SCOMP * 32];         // per-edge sh factors of the current pass (fp32)
  __shared__ float s_inv[NB];
#if DP_BABL == 10
  __shared__ float s_pad[2048];
  if (threadIdx.x == 1023) s_pad[threadIdx.x] = 0.f;
#endif
  using S = Spec<KIND>;
  static_assert(4 * NB * S::D_OUT <= 4 * TIMG_F, "reduction scratch fits in the T images");
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  // conv split (small batches: FusedArgs.conv_split, as in the fp32 kernel): blockIdx = conv * groups + group, this workgroup runs ONE conv of the
  // launch and leaves the sum of its four k-partials in part[conv][node] - shifts, residual and the sum over convs are dp_conv_node_combine's
  const int n_conv_all = (HAS_LL ? 1 : 0) + fa.n_x;
  const int groups = fa.conv[0].B * fa.conv[0].slices;
  const int bid = fa.conv_split ? (int)blockIdx.x % groups : (int)blockIdx.x;
  const int cx_only = fa.conv_split ? (int)blockIdx.x / groups : -1;
  const Grp grp = grp_of(fa.conv[0], bid);
  if (grp.n_loc <= 0) return;
  f32x4 acc[n_acc4<KIND>()];
#pragma unroll
  for (int i = 0; i < n_acc4<KIND>(); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  NProfT prof_prev{};
  (void)prof_prev;
  NPROF(-1);
  if constexpr (HAS_LL) {
    if (cx_only <= 0) conv_bf16<KIND, LL, HALFOK>(fa.conv[0], grp, tid, s_T, s_S, s_inv, acc, prof_prev);
  }
  {
    const int c_lo = cx_only < 0 ? (HAS_LL ? 1 : 0) : (cx_only < (HAS_LL ? 1 : 0) ? n_conv_all : cx_only);
    const int c_hi = cx_only < 0 ? n_conv_all : (c_lo < n_conv_all ? c_lo + 1 : n_conv_all);
#pragma unroll 1
    for (int cx = c_lo; cx < c_hi; ++cx) conv_bf16<KIND, XMODE, HALFOK, S0K>(fa.conv[cx], grp, tid, s_T, s_S, s_inv, acc, prof_prev);
  }
  // ---- sum the four k-partials, BatchNorm shift, residual stream (mean and BatchNorm scale are inside the operands) -----------------
  {
    const DpConvArgs& a = fa.conv[0];
    const int n_conv = cx_only < 0 ? n_conv_all : 0;       // (conv split: no shifts, no residual here)
    constexpr int EP_ROUNDS = (NB * S::D_OUT + 255) / 256;
    float er[EP_ROUNDS], eb[EP_ROUNDS];
#pragma unroll
    for (int it = 0; it < EP_ROUNDS; ++it) {
      const int i = tid + 256 * it;
      const bool live = i < grp.n_loc * S::D_OUT;
      const int n = i / S::D_OUT, ch = i - n * S::D_OUT;
      const int node = grp.n_lo + (live ? n : 0);
      float sh = 0.f;
      for (int k = 0; k < n_conv; ++k) sh += fa.conv[k].bn_shift[live ? ch : 0];
      if constexpr (S0K) {
        // the conv's listed pairs (DpConvArgs.list_*: messages of a DP_EDGES_LIST launch): scatter mean over the conv's full degree, their BatchNorm scale
        for (int k = 0; k < n_conv; ++k) {
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
    __syncthreads();
    AccStore<KIND, 0>::run(acc, s_T + q * (NB * S::D_OUT), lane);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EP_ROUNDS; ++it) {
      const int i = tid + 256 * it;
      if (i >= grp.n_loc * S::D_OUT) continue;
      const int n = i / S::D_OUT, ch = i - n * S::D_OUT;
      const float part = ((s_T[i] + s_T[NB * S::D_OUT + i]) + s_T[2 * NB * S::D_OUT + i]) + s_T[3 * NB * S::D_OUT + i];
      if (cx_only < 0) a.out[(size_t)(grp.n_lo + n) * a.out_stride + ch] = er[it] + (part + eb[it]);
      else fa.part[(size_t)cx_only * fa.part_stride + (size_t)(grp.n_lo - fa.node0 + n) * a.out_stride + ch] = part;
    }
  }
  NPROF(4);
#ifdef DP_PROF
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&dpb_prof[i], prof_prev.acc[i]);
#endif
}

template <int KIND, bool HALFOK> int launch_bf16_h(const FusedArgs& fa, int xmode, hipStream_t st);
template <int KIND, bool HALFOK> int launch_bf16_s0(const FusedArgs& fa, int xmode, hipStream_t st, dim3 grid, dim3 block) {
  if (xmode == DP_EDGES_P2L) hipLaunchKernelGGL((tp_conv_node_bf16_kernel<KIND, false, DP_EDGES_P2L, HALFOK, true>), grid, block, 0, st, fa);
  else hipLaunchKernelGGL((tp_conv_node_bf16_kernel<KIND, false, DP_EDGES_L2P, HALFOK, true>), grid, block, 0, st, fa);
  return (int)hipGetLastError();
}
template <int KIND, bool HALFOK> int launch_bf16_h(const FusedArgs& fa, int xmode, hipStream_t st) {
  const DpConvArgs& a = fa.conv[0];
  const dim3 grid((unsigned)(a.B * a.slices * (fa.conv_split ? fa.n_ll + fa.n_x : 1))), block(256);
  int n_s0 = 0;
  for (int i = 0; i < fa.n_ll + fa.n_x; ++i) n_s0 += fa.conv[i].s0_only ? 1 : 0;
  if (n_s0) {                                     // s0-only convs: their own instantiation, never mixed with full convs in one launch
    if (fa.n_ll || n_s0 != fa.n_x) return DP_E_UNSUPPORTED;
    return launch_bf16_s0<KIND, HALFOK>(fa, xmode, st, grid, block);
  }
  if (fa.n_ll) hipLaunchKernelGGL((tp_conv_node_bf16_kernel<KIND, true, DP_EDGES_P2L, HALFOK>), grid, block, 0, st, fa);
  else if (xmode == DP_EDGES_P2L) hipLaunchKernelGGL((tp_conv_node_bf16_kernel<KIND, false, DP_EDGES_P2L, HALFOK>), grid, block, 0, st, fa);
  else hipLaunchKernelGGL((tp_conv_node_bf16_kernel<KIND, false, DP_EDGES_L2P, HALFOK>), grid, block, 0, st, fa);
  return (int)hipGetLastError();
}
template <int KIND> int launch_bf16(const FusedArgs& fa, int xmode, hipStream_t st) {
  return fa.lean ? launch_bf16_h<KIND, false>(fa, xmode, st) : launch_bf16_h<KIND, true>(fa, xmode, st);
}

}  // namespace dpb

int dpb_launch_k0(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpb_launch_k1(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpb_launch_k2(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
int dpb_launch_k3(const dpn::FusedArgs& fa, int xmode, hipStream_t st);
