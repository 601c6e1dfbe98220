/*
 * diffphore_amd.h - C ABI of the MI355X-native DiffPhore reverse-diffusion hot path (libdiffphore_hip.so).
 *
 * Drop-in boundary.  The reference is pure Python; the "FFI" a maintainer would add is a ctypes
 * binding from the two reference callables to these entry points:
 *
 *   TensorProductScoreModel.forward(data)      /root/reference/src/models/score_model_phore.py:294-310  -> dp_forward
 *   sampling_phore(...) per-step body          /root/reference/src/utils/sampling.py:204-255           -> dp_sample_step
 *   modify_conformer(...)                      /root/reference/src/utils/diffusion_utils.py:23-79      -> dp_conformer_update
 *   TensorProductConvLayer.forward(...)        /root/reference/src/models/score_model_phore.py:134-149 -> dp_tp_conv
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm tensors in the shipped host code); the
 *     library never allocates, frees or synchronises; all work is enqueued on the hipStream_t passed as `void*`.
 *   - return 0 on success, negative DP_E_* on bad arguments, positive hipError_t on a launch failure.  Never throws.
 *   - fp32 everywhere; int32 indices; "ptr" arrays are CSR offsets with B+1 entries (B = poses in the batch).
 *   - thread-compatible: no mutable globals.
 */
#ifndef DIFFPHORE_AMD_H
#define DIFFPHORE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DP_OK 0
#define DP_E_BADARG (-1)
#define DP_E_UNSUPPORTED (-2)
#define DP_E_TOOLARGE (-3)

#define DP_ABI_VERSION 12

/* tensor-product conv "kinds": which irreps / path table the kernel is specialised for */
enum DpConvKind {
  DP_CONV_L0 = 0,   /* 20x0e                    -> +10x1o           numel  600 */
  DP_CONV_L1 = 1,   /* 20x0e+10x1o              -> +10x1e           numel 1100 */
  DP_CONV_L2 = 2,   /* 20x0e+10x1o+10x1e        -> +20x0o           numel 1600 */
  DP_CONV_L3 = 3,   /* 20x0e+10x1o+10x1e+20x0o  -> same             numel 2200 */
  DP_CONV_FINAL = 4,/* ... -> 2x1o+2x1e  (fc 40->40->200)           score_model_phore.py:256-264 */
  DP_CONV_TOR = 5   /* ... x FullTP(sh,2e) -> 20x0o+20x0e (1600)    score_model_phore.py:276-285 */
};

/* edge-set "modes": how (scatter node, gather node, attr nodes, feature row) are derived from a sorted edge slot */
enum DpEdgeMode {
  DP_EDGES_CSR = 0,     /* explicit CSR: rowptr/rowend per scatter node, nbr[] = gather node (lig-lig, phore-phore, torsion) */
  DP_EDGES_P2L = 1,     /* dense cross, scatter = ligand atom i, gather = phore node j      (:682-689) */
  DP_EDGES_L2P = 2,     /* dense cross, scatter = phore node j, gather = ligand atom i      (:695-700) */
  DP_EDGES_CENTER = 3,  /* scatter = pose, one edge per ligand atom                         (:381-406) */
  DP_EDGES_LLPAIR = 4,  /* node-side kernel only: ligand-ligand edges as a dense atom x atom pair table of one pose; pair (i, j) carries
                           up to two edges (radius + bond, build_lig_conv_graph :715-739) found through pair_a / pair_b */
  DP_EDGES_LIST = 5     /* edge-side kernel, DP_PREC_F16X3 only: an explicit edge list whose per-edge messages are written out (DpConvArgs.msg_out) */
};

/* arithmetic of the radial-MLP second layer (the 60 x numel contraction that dominates the path) */
enum DpPrecision {
  DP_PREC_F32 = 0,      /* exact fp32 MFMA (edge-side kernel: v_mfma_f32_32x32x2_f32; node-side kernel: v_mfma_f32_16x16x4_f32) */
  DP_PREC_BF16X6 = 3,   /* 3-way bf16 operand split, 6 cross products, fp32 accumulate: ~2^-24 relative, fp32-grade */
  DP_PREC_BF16X3 = 2,   /* 2-way bf16 split, 3 cross products: ~2^-17 relative */
  DP_PREC_BF16X9 = 4,   /* 3-way split, all 9 cross products: every fp32 x fp32 product exact (3 x 8 significand bits), fp32 accumulate */
  DP_PREC_BF16 = 5,     /* plain bf16 OPERANDS, fp32 accumulate ("bf16 mixed precision", BASELINE.json configs[4]): radial MLP weights and
                           activations, the tensor-product operand and the node-side intermediate T are rounded to bf16 once (2^-9
                           relative each); geometry, spherical harmonics, edge-embedding MLPs, accumulation, BatchNorm shift and the
                           residual stream stay fp32.  Node-side kernel: dp_conv_node_bf16_impl.h (one v_mfma_f32_16x16x32_bf16 per node,
                           column tile and pass).  Edge-side kernel: the bf16x3 arithmetic (its weights are packed as for DP_PREC_BF16X3). */
  DP_PREC_F16X3 = 6     /* fp32-GRADE on the 16-bit pipe.  Node-side kernel (dp_conv_node_f16_impl.h): every operand of the three GEMMs as TWO fp16 limbs
                           (hi = rne16(v), lo = rne16(v - hi): 22 significand bits; exact power-of-two scales map the operands into fp16's
                           range), cross products hi*hi + hi*lo + lo*hi (+ lo*lo in the second contraction) on v_mfma_f32_16x16x32_f16, fp32
                           accumulate.  Held to the same tolerances as DP_PREC_F32.  PRECONDITIONS (what "inside fp16's range" means here):
                           the node-side intermediate T = sum over a node's edges of hidden x tensor-product operand, divided by the degree,
                           carries up to 2^5: |T| < ~2047; gathered features, SH factors, edge embeddings and hidden activations < 65504; the
                           packed second layer is scaled by ONE power of two per conv (max |W2 x bn_scale| -> (1024, 2048]), so a weight more
                           than 2^25 below the conv's largest keeps less than 11 + 11 bits (accuracy is relative to that maximum, as the
                           tests' tolerance is).  Beyond the range: non-finite scores -> DpWork.status bit 2 (the caller replays the batch with
                           node_f16 = 0, edge_bf16x6 = 0: tests/c_host/dp_host_forward.c, sampling._run_parts).
                           Edge-side kernel (dp_conv.hip, round 5): the radial contraction W2 hid with both operands as two fp16 limbs, 3 cross
                           products on v_mfma_f32_32x32x16_f16 (weights from dp_pack_conv_weights(kind, DP_PREC_F16X3): W2 * 2^w, w in a trailing
                           quad; the caller passes bn_scale * 2^-w); hidden activations are converted unscaled (< 65504).  Not a model precision:
                           dp_forward uses it for the convs of a DP_PREC_F32 model (DpWork.node_f16 for the node-side launches, DpWork.edge_bf16x6
                           for the convs that stay edge-side and for the DP_EDGES_LIST launches). */
};

/* formulation of the fused conv: per-edge radial GEMM, or node-side contraction (dense cross edge sets; DP_PREC_F32 or DP_PREC_BF16) */
enum DpConvAlgo {
  DP_ALGO_AUTO = 0,     /* library's choice by precision and batch size (dp_resolve_conv_algo) */
  DP_ALGO_EDGE = 1,     /* dp_conv.hip: w[e] = W2 hid[e] on the matrix pipe per edge, CG contraction on the VALU, segmented mean */
  DP_ALGO_NODE = 2      /* dp_conv_node*.{hip,h}: T[n] = hid^T z over the node's edges, out[n] = W2 T[n] batched over 16 nodes */
};

typedef struct DpConvArgs {
  int32_t kind;          /* DpConvKind */
  int32_t mode;          /* DpEdgeMode */
  int32_t B;             /* poses */
  int32_t slices;        /* workgroups per pose (>=1); scatter nodes of a pose are split evenly between them (node-side kernel: a pose
                            uses only the ceil(nodes / 16) groups it needs, the pose's other workgroups leave at once) */
  int32_t precision;     /* DpPrecision; must match the precision w2p was packed for */
  int32_t max_scatter;   /* largest number of scatter nodes of any pose; ceil(max_scatter / slices) must be <= dp_conv_max_nodes() */
  /* pose CSRs (B+1 entries each) */
  const int32_t* scat_ptr;   /* scatter-node offsets per pose (ligand atoms / phore nodes / rotatable bonds / poses) */
  const int32_t* lig_ptr;    /* ligand atom offsets per pose */
  const int32_t* ph_ptr;     /* phore node offsets per pose */
  const int32_t* x_ptr;      /* dense cross-edge offsets per pose (sum N_l*N_p) */
  /* DP_EDGES_CSR only */
  const int32_t* rowptr;     /* [n_scatter] first edge slot of each scatter node */
  const int32_t* rowend;     /* [n_scatter] one-past-last edge slot */
  const int32_t* nbr;        /* [n_edge_slots] gather node of each edge */
  const int32_t* edge_scat;  /* [n_edge_slots] scatter node of each edge (edges sorted by it) */
  /* node tables */
  const float* gather_tab;   /* features gathered through the TP: [n, D_in] */
  int32_t gather_stride;
  const float* attr1_tab;    /* h[:, :20] of edge_index[0]-side node (ligand atom for cross edges) */
  int32_t attr1_stride;
  const float* attr2_tab;    /* h[:, :20] of edge_index[1]-side node (phore node for cross edges; bond sums for torsion) */
  int32_t attr2_stride;
  int32_t attr_swap;         /* CSR mode: 0 -> attr1 = scatter node, attr2 = gather node (lig-lig, phore-phore);
                                          1 -> attr1 = gather node, attr2 = scatter node (torsion: atom, bond) */
  /* per-edge inputs */
  const float* edge_emb;     /* [n_edge_slots, 20] embedded edge attributes */
  const float* edge_sh;      /* [n_edge_slots, sh_stride] spherical harmonics (9) or contracted torsion sh (7) */
  int32_t sh_stride;
  /* packed weights (see diffphore_amd/packing.py) */
  const float* w1p;          /* radial MLP layer 1 (+bias) in MFMA A-operand order */
  const float* w2p;          /* radial MLP layer 2 (+bias), path coefficient and CG constants folded, MFMA A-operand order */
  const float* bn_scale;     /* [D_out]  weight / sqrt(running_var + eps) expanded per component */
  const float* bn_shift;     /* [D_out]  bias - running_mean*scale on 0e channels, 0 elsewhere */
  /* output */
  float* out;                /* [n_scatter, out_stride] */
  int32_t out_stride;
  int32_t accumulate;        /* 0: out = BN(mean) (+ resid zero-padded), 1: out += BN(mean) */
  const float* resid;        /* optional [n_scatter, resid_stride], added on the first resid_width channels */
  int32_t resid_stride;
  int32_t resid_width;
  int32_t phase_split;       /* 1: two workgroups per (pose, slice), one per accumulation phase (scalar / vector output channels) - halves a
                                conv's latency when B*slices cannot fill the chip; results are bit-identical.  The fp32-MFMA kernel, and DP_EDGES_LIST
                                launches that carry a whole conv (sh0_zero = 0: two workgroups per 128-edge chunk, each emits the output channels of its
                                phase - dp_forward: the phore-phore conv of a batch with at most 64 chunks); ignored otherwise and for kinds with a
                                single phase */
  /* node-side formulation (DP_ALGO_NODE): weights in the layout of dp_pack_conv_weights_node; `slices` = node groups per pose with
   * ceil(max_scatter / slices) <= dp_conv_node_max_nodes().  Kinds 0-3; DP_PREC_F32 (weights from dp_pack_conv_weights_node) or
   * DP_PREC_BF16 (weights from dp_pack_conv_weights_node_bf16, plus edge_emb_bf below; no path_split / conv split variants: such
   * launches run unsplit); modes P2L / L2P (dense cross edges) and LLPAIR (ligand-ligand edges through the per-pose pair tables below).
   * PRECONDITION on edge_sh: component 0 of every row is exactly 1.0 - true for e3nn's 'component'-normalised spherical harmonics
   * (Y_0 = 1, score_model_phore.py:737,754,891-894), which is all dp_forward ever passes.  The node-side kernel never reads sh[0]:
   * the l=0 paths use the gathered input itself as the GEMM-A operand.  A caller of dp_tp_conv with other edge_sh[:,0] values must
   * use DP_ALGO_EDGE (the edge-side kernel multiplies by sh[0]). */
  const float* w1n;
  const float* w2n;
  int32_t algo;              /* DpConvAlgo; AUTO / EDGE -> the edge-side kernel */
  int32_t path_split;        /* node-side only. 1: two workgroups per node group, each owning half of the output irrep slots - halves a conv's
                                latency when B*slices cannot fill the chip (both repeat the cheap radial layer 1); bit-identical results */
  /* DP_EDGES_LLPAIR: per pose a [N_l x N_l] table at pair_ptr[pose]; entry (i, j) = edge slot (row of edge_emb / edge_sh, CSR order)
   * of the radius edge scatter i <- gather j (pair_a) / of the bond edge (pair_b), -1 if absent.  rowptr / rowend give the degree. */
  const int32_t *pair_a, *pair_b, *pair_ptr;
  /* DP_PREC_BF16 node-side kernel only: the edge embedding once more as bf16 rows of 32 (64 bytes: features 0..19 | 0 ... ), plus ONE
   * flag row (all zero but feature 20 = 1.0) at row index emb_flag_row that stands for "this edge does not exist" (dp_forward: written by
   * the edge-embedding kernels beside edge_emb; standalone callers: dp_edge_emb_bf16).  DP_PREC_F16X3: rows of 64 fp16 (128 bytes: hi limbs |
   * lo limbs, dp_edge_emb_f16) */
  const uint16_t* edge_emb_bf;
  int32_t emb_flag_row;
  int32_t no_tail;           /* node-side only, a HINT: 1 = the caller guarantees that no pose of the batch ends on a pass of <= 16 edge slots
                                (edges per node mod 32 is 0 or > 16 for every pose), so the launch may use the instantiation without the
                                half-pass variant (half the code, ~40 % fewer spilled registers, ~2 % faster).  0 = unknown: the general
                                kernel.  Results are identical either way when the hint is true. */
  int32_t s0_only;           /* node-side DP_PREC_F16X3 kernel, modes P2L / L2P: 1 = the caller guarantees that edge_sh is e0 = (1, 0, ..., 0) on every
                                edge this conv is asked to cover, so only the tensor-product paths through the l = 0 component are evaluated (the
                                other paths' operands are identically zero).  dp_forward sets it on the two "normals" convs of a layer
                                (phore_to_lig_norm / lig_to_phore_norm, score_model_phore.py:686-689,698-700): their spherical harmonics are e0
                                wherever the pharmacophore type does not agree with the atom's fingerprint (:874-894 - r = 0 there), and the few
                                pairs that do agree (DpBatch.xs_*) are added by an edge-side launch over exactly those pairs (DP_EDGES_LIST) */
  /* DP_EDGES_LIST (edge-side kernel at DP_PREC_F16X3 only): an explicit list of n_list edges anywhere in the batch - edge k scatters onto node
   * edge_scat[k], gathers node nbr[k], reads row feat_idx[k] (NULL: row k) of edge_emb / edge_sh; attr_swap as for DP_EDGES_CSR.  The launch writes
   * the edges' MESSAGES msg_out[k][D_out] (no mean, no BatchNorm: dp_list_reduce / dp_list_finish / DpConvArgs.list_* finish the conv) - 128 edges
   * per workgroup whatever nodes (and poses) they belong to: no partly filled chunk per pose */
  const int32_t* feat_idx;
  float* msg_out;
  int32_t n_list;
  int32_t sh0_zero;          /* DP_EDGES_LIST: 1 = component 0 of every edge_sh row is read as 0: the tensor-product paths through the l = 0 harmonic are
                                not part of this launch (that part of the conv is an s0_only launch); 0 = the whole conv */
  /* s0_only convs: the messages of the conv's listed pairs (a DP_EDGES_LIST launch that ran before this one), folded in by this launch's epilogue
   * instead of a dp_list_reduce launch: out[n][c] also gets list_scale[c] * sum(list_msg[k][c], k = list_rowptr[n] .. list_rowptr[n + 1]) / degree.
   * All three or none. */
  const float* list_msg;
  const int32_t* list_rowptr;
  const float* list_scale;
} DpConvArgs;

/* Fused gather -> radial MLP (MFMA) -> Clebsch-Gordan tensor product -> per-node segmented mean -> e3nn BatchNorm(eval).
 * Replaces TensorProductConvLayer.forward, score_model_phore.py:134-149. */
int dp_tp_conv(const DpConvArgs* args, void* stream);
/* Finishes a DP_EDGES_LIST launch: out[n][c] += bn_scale[c] * (sum of msg[k][c] over the listed edges k of node n, rowptr[n] .. rowptr[n + 1], in list
 * order) / max(deg, 1), deg = deg_ptr[p + 1] - deg_ptr[p] of the node's pose p (nodes scat_ptr[p] .. scat_ptr[p + 1], at most max_scatter of them): the scatter MEAN of
 * TensorProductConvLayer.forward (score_model_phore.py:141) runs over ALL edges of the dense conv the list is a part of, then BatchNorm's scale.
 * bn_scale: what the messages' weights were packed with (DpConvW.bn_scale_x for DP_PREC_F16X3).  Deterministic (one thread per node and channel). */
int dp_list_reduce(const float* msg, const int32_t* rowptr, int32_t B, const int32_t* scat_ptr, int32_t max_scatter, const int32_t* deg_ptr, int32_t d_out,
                   const float* bn_scale, float* out, int32_t out_stride, void* stream);
/* Finishes a DP_EDGES_LIST launch that carried a WHOLE conv (sh0_zero = 0, every edge listed, grouped by scatter node: edges rowptr[n] .. rowend[n] of
 * node n): out[n][c] = base + bn_scale[c] * mean of msg[k][c] over the node's edges + bn_shift[c], base = out[n][c] (accumulate) or resid[n][c] for
 * c < resid_width or 0 - the epilogue of the CSR kernel, for convs whose poses have too few edges to fill 128-edge chunks of their own
 * (dp_forward: the phore-phore convs of small pharmacophores). */
int dp_list_finish(const float* msg, const int32_t* rowptr, const int32_t* rowend, int32_t n_nodes, int32_t d_out, const float* bn_scale, const float* bn_shift,
                   const float* resid, int32_t resid_stride, int32_t resid_width, float* out, int32_t out_stride, int32_t accumulate, int32_t bcast_nodes,
                   void* stream);
/* (bcast_nodes > 0: the list holds the edges of the first bcast_nodes nodes only and node n takes the messages of node n mod bcast_nodes - one
 *  pharmacophore's conv handed to every pose that carries it, DpBatch.ph_shared; 0 = every node its own rows) */
/* the node-side kernel directly (dp_tp_conv dispatches here when args->algo == DP_ALGO_NODE) */
int dp_tp_conv_node(const DpConvArgs* args, void* stream);
int dp_conv_node_max_nodes(void);
int dp_conv_w1n_floats(int32_t kind);
int dp_conv_w2n_floats(int32_t kind);
/* HOST pointers: the same reference tensors as dp_pack_conv_weights, re-ordered for the node-side kernel */
int dp_pack_conv_weights_node(int32_t kind, const float* fc0_w, const float* fc0_b, const float* fc3_w, const float* fc3_b,
                              const float* bn_scale, float* w1n, float* w2n);
/* the same for the DP_PREC_BF16 node-side kernel: bf16 MFMA fragments (sizes in FLOATS of the blob, i.e. 2 bf16 each); DpConvArgs.w1n / w2n
 * point at these images when DpConvArgs.precision == DP_PREC_BF16 */
int dp_conv_w1b_floats(int32_t kind);
int dp_conv_w2b_floats(int32_t kind);
int dp_pack_conv_weights_node_bf16(int32_t kind, const float* fc0_w, const float* fc0_b, const float* fc3_w, const float* fc3_b,
                                   const float* bn_scale, float* w1b, float* w2b);
/* DEVICE: emb [n_rows, 20] fp32 -> out [(n_rows + 1), 32] bf16 (round to nearest even), row n_rows = the flag row (DpConvArgs.edge_emb_bf) */
int dp_edge_emb_bf16(const float* emb, int32_t n_rows, uint16_t* out, void* stream);
/* the same for the DP_PREC_F16X3 node-side kernel: two-limb fp16 MFMA fragments (w2h: W2 * 2^w, the exponent w in the image's header; sizes in
 * FLOATS of the blob); DpConvArgs.w1n / w2n point at these images when DpConvArgs.precision == DP_PREC_F16X3 */
int dp_conv_w1h_floats(int32_t kind);
int dp_conv_w2h_floats(int32_t kind);
int dp_pack_conv_weights_node_f16(int32_t kind, const float* fc0_w, const float* fc0_b, const float* fc3_w, const float* fc3_b,
                                  const float* bn_scale, float* w1h, float* w2h);
/* DEVICE: emb [n_rows, 20] fp32 -> out [(n_rows + 1), 64] fp16: per row the hi limbs (20 features | zeros to 32) then the lo limbs (32), row
 * n_rows = the flag row (hi feature 20 = 1.0); DpConvArgs.edge_emb_bf of a DP_PREC_F16X3 launch */
int dp_edge_emb_f16(const float* emb, int32_t n_rows, uint16_t* out, void* stream);
/* `bn_scale` (HOST, [D_out], the `scale` output of dp_pack_bn; NULL = ones) is folded into the packed second layer: the node-side
 * kernel applies the scatter mean and the BatchNorm scale inside its operands and only ADDS bn_shift in its epilogue, which is what lets
 * several convs share one set of accumulators (below).  DpConvArgs.bn_scale is therefore ignored by the node-side kernel. */

/* Fused node-side launch: convs[0..n) (n <= 3) scatter onto the same node groups and their outputs are summed, as the reference sums
 * the outputs of lig_conv + phore_to_lig + phore_to_lig_norm (score_model_phore.py:703-706) and of lig_to_phore + lig_to_phore_norm
 * (:708-710).  Allowed sequences: [LLPAIR] [P2L] [P2L] (any non-empty sub-sequence in that order) or [L2P] [L2P].  convs[0] names
 * out / resid / accumulate; kind, B, slices, scat_ptr, max_scatter and path_split must agree.  One prologue, one reduction of the
 * k-partials, one epilogue and one write of `out` instead of one read-modify-write per conv. */
int dp_tp_conv_node_fused(const DpConvArgs* convs, int32_t n, void* stream);
/* The same sum for launches that cannot fill the chip (a single complex's 40 samples): every conv of the sequence runs on its OWN
 * workgroups (grid x n, with path_split = 1 that is 2 n workgroups per node group), leaves the sum of its k-partials in
 * part[conv][node][out_stride] (DEVICE scratch, part_floats >= n * n_nodes * out_stride) and a second, tiny launch adds the convs, their
 * BatchNorm shifts and the residual / previous output up in a fixed order: deterministic, equal to dp_tp_conv_node_fused to fp32
 * rounding, n x shorter when latency-bound.  Requires convs[0].path_split == 1, scat_ptr[0] == 0, n_nodes == scat_ptr[B], 2 <= n <= 3.
 * DP_PREC_BF16 / DP_PREC_F16X3 (round 5): the conv split without the path split (grid x n; the f16x3 kernel's 8-node groups already double
 * the workgroups); dp_forward asks for it while the fused launch would have at most 384 (f16x3) / 256 (bf16) workgroups. */
int dp_tp_conv_node_fused_split(const DpConvArgs* convs, int32_t n, int32_t n_nodes, float* part, int64_t part_floats, void* stream);
/* what DP_ALGO_AUTO resolves to for a batch of B poses at `precision` (returns DP_ALGO_EDGE or DP_ALGO_NODE) */
int dp_resolve_conv_algo(int32_t algo, int32_t precision, int32_t B);

/* scatter nodes one workgroup can own: callers of dp_tp_conv pick `slices` so that ceil(nodes per pose / slices) <= this */
int dp_conv_max_nodes(int32_t precision);
/* numbers the host packer needs */
int dp_conv_w1p_floats(int32_t kind);
int dp_conv_w2p_floats(int32_t kind, int32_t precision);
int dp_conv_dims(int32_t kind, int32_t* d_in, int32_t* d_out, int32_t* numel, int32_t* n_attr, int32_t* hidden);
/* HOST pointers: reorder one TensorProductConvLayer's fc parameters (reference state_dict layout) into MFMA operand order */
int dp_pack_conv_weights(int32_t kind, int32_t precision, const float* fc0_w, const float* fc0_b, const float* fc3_w,
                         const float* fc3_b, float* w1p, float* w2p);

/* ------------------------------------------------------------------------------------------------------------------
 * Packed batch of B poses (struct of arrays, CSR offsets per pose).  One "pose" = one (ligand conformer, pharmacophore)
 * sample of the reference's data_list (sampling.py:184,210).  Static fields come from the reference's graph builder
 * (pdbbind_phore.py:1143-1188); `pos` / `norm` are the diffusion state.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct DpBatch {
  int32_t B;
  int32_t n_lig, n_ph, n_bond, n_tor, n_pp, n_x, ll_cap, n_llp;  /* totals over the batch (n_llp = sum N_l^2) */
  int32_t max_nl, max_np, max_t;                           /* per-pose maxima (kernel limits: 256 / 256 / 128) */
  int32_t no_tail_nl, no_tail_np;   /* HINTS for the node-side convs (DpConvArgs.no_tail): 1 = no pose has N_l (N_p) mod 32 in 1..16; 0 = unknown */
  /* pose CSRs, B+1 entries */
  const int32_t *lig_ptr, *ph_ptr, *bond_ptr, *tor_ptr, *pp_ptr, *x_ptr, *ll_ptr, *mask_ptr;
  const int32_t* pose_ptr;    /* [B+1] = 0..B (scatter CSR of the centre conv) */
  const int32_t* llp_ptr;     /* [B+1] offsets of the per-pose N_l x N_l ligand pair tables (DpWork.ll_pair_*) */
  /* ligand, static */
  const int32_t* lig_x;       /* [n_lig,16] categorical atom features                     process_mols.py:162-179 */
  const int32_t *bond_src, *bond_dst, *bond_type; /* [n_bond] directed bonds, global atom ids, type 0..3 (one-hot index) */
  const int32_t *tor_u, *tor_v;                   /* [n_tor] rotatable bonds (edge_index[:, edge_mask])  torsion.py:13-61 */
  const uint8_t* mask_rotate; /* per pose T x N_l bytes at mask_ptr[p]                    torsion.py:54-61 */
  const float* phorefp;       /* [n_lig,11] */
  const float *norm_angle1, *norm_angle2;         /* [n_lig,11] */
  /* pharmacophore, static */
  const float* ph_pos;        /* [n_ph,3] */
  const float* ph_norm;       /* [n_ph,3] unit normal or 0 */
  const int32_t* ph_x;        /* [n_ph,3] categorical: type, is-EX idx, has-norm idx      process_pharmacophore.py:717-748 */
  const float* ph_scalar;     /* [n_ph,2] alpha, weight */
  const int32_t* ph_type;     /* [n_ph] argmax of the one-hot `phoretype`; 10 = exclusion sphere */
  const int32_t *pp_src, *pp_dst;                 /* [n_pp] phore-phore edges sorted by src, global node ids */
  const int32_t *pp_rowptr, *pp_rowend;           /* [n_ph] */
  /* diffusion state */
  float* pos;                 /* [n_lig,3] */
  float* norm;                /* [n_lig,33] ligand normal buffer (atom-major [N,11,3]) */
  const float* t;             /* [B,3] diffusion times (tr, rot, tor) per pose            diffusion_utils.py:181-207 */
  /* The cross pairs (atom i, pharmacophore node j) whose "normals" spherical harmonics can differ from e0 = (1, 0, ..., 0): the pairs with
   * agreement != 0, i.e. node j is not an exclusion sphere and phorefp[i, type(j)] != 0 (score_model_phore.py:790-793,874-894: the vector the
   * harmonics are taken of is cross(...) * sum(agreement), exactly 0 without agreement).  STATIC (pose-independent), built by dp_pack_fill;
   * listed twice, as CSR over the scatter nodes of each direction.  dp_forward evaluates the two normals convs of a layer as their l = 0 part on
   * every pair (DpConvArgs.s0_only) + the other paths on these pairs only (DP_EDGES_LIST).  NULL / n_xs < 0 = not provided: the dense convs. */
  int32_t n_xs;               /* number of such pairs in the batch */
  const int32_t *xs_src, *xs_nbr, *xs_feat;   /* [n_xs] sorted by (pose, atom, node): global atom id, global node id, dense cross row x_ptr[p] + i N_p + j */
  const int32_t* xs_rowptr;                   /* [n_lig + 1] the pairs of atom a are xs_rowptr[a] .. xs_rowptr[a + 1] */
  const int32_t *xt_src, *xt_nbr, *xt_feat;   /* [n_xs] sorted by (pose, node, atom): global node id, global atom id, the same dense cross row */
  const int32_t* xt_rowptr;                   /* [n_ph + 1] */
  /* 1 = every pose of the batch carries the SAME pharmacophore (same node features, positions, normals, the same edges in the same order): a virtual
   * screen against one pharmacophore, the N samples of one complex (inference.py:183-209).  Found by dp_pack_query (content comparison with pose 0).
   * What the model computes from the pharmacophore alone is then the same for every pose while the diffusion times are (DpWork.t_uniform): dp_forward
   * evaluates the layer-0 phore-phore conv - inputs h_p^0, e_pp, sh_pp: score_model_phore.py:691-693 at l = 0 - on pose 0's edges only and hands every
   * pose that output (dp_list_finish, bcast_nodes).  0 = not known / not so. */
  int32_t ph_shared;
} DpBatch;

/* ------------------------------------------------------------------------------------------------------------------
 * Native collate (csrc/dp_pack.hip, HOST code - callable without a GPU).  Stands where the reference builds a PyG batch of the pose list
 * (DataLoader / Batch.from_data_list, src/utils/sampling.py:210-215): one call turns N poses into the DpBatch image.  All pointers below
 * are HOST pointers into the caller's own tensors (reference field names in the comments); the samples of one complex share ONE view.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct DpGraphView {
  int32_t n_atoms, n_phore, n_bond_edges, n_tor, n_pp;
  const int64_t* lig_x;         /* data['ligand'].x            [n_atoms,16] int64 */
  const int64_t* bond_index;    /* data['ligand','ligand'].edge_index [2,n_bond_edges] int64, local atom ids */
  const float* bond_attr;       /* .edge_attr                  [n_bond_edges,4] one-hot */
  const uint8_t* edge_mask;     /* data['ligand'].edge_mask    [n_bond_edges] bool: rotatable bonds, n_tor of them set */
  const uint8_t* mask_rotate;   /* .mask_rotate                [n_tor,n_atoms] bool (NULL when n_tor == 0) */
  const float *phorefp, *norm_angle1, *norm_angle2;   /* [n_atoms,11] */
  const float* ph_x;            /* data['phore'].x             [n_phore,5] float: type, is-EX idx, has-norm idx, alpha, weight */
  const float *ph_pos, *ph_norm;                      /* [n_phore,3] */
  const float* ph_type;         /* .phoretype                  [n_phore,11] one-hot */
  const int64_t* pp_index;      /* data['phore','phore'].edge_index [2,n_pp] int64, local node ids */
} DpGraphView;
typedef struct DpPoseRef {
  const DpGraphView* view;      /* the pose's complex */
  const float* pos;             /* data['ligand'].pos  [n_atoms,3]  (this sample's diffusion state) */
  const float* norm;            /* data['ligand'].norm [n_atoms,33] */
} DpPoseRef;
/* byte offsets of every DpBatch array inside the ONE block dp_pack_fill writes (256-byte aligned segments); `bytes` = size of the block */
typedef struct DpPackLayout {
  int64_t lig_ptr, ph_ptr, bond_ptr, tor_ptr, pp_ptr, x_ptr, ll_ptr, mask_ptr, pose_ptr, llp_ptr;
  int64_t lig_x, bond_src, bond_dst, bond_type, tor_u, tor_v, mask_rotate, phorefp, norm_angle1, norm_angle2;
  int64_t ph_pos, ph_norm, ph_x, ph_scalar, ph_type, pp_src, pp_dst, pp_rowptr, pp_rowend, pos, norm;
  int64_t xs_src, xs_nbr, xs_feat, xs_rowptr, xt_src, xt_nbr, xt_feat, xt_rowptr;
  int64_t bytes;
} DpPackLayout;
/* totals (B, n_*, max_*, n_xs, ph_shared) of the batch into *totals and the block layout into *layout; DP_E_TOOLARGE for a pose over the kernel limits
 * (256 atoms / 256 pharmacophore nodes / 128 torsions) or a batch beyond int32 edge indexing */
int dp_pack_query(int32_t n, const DpPoseRef* poses, DpBatch* totals, DpPackLayout* layout);
/* fill a HOST block of layout->bytes: CSR offsets, concatenated static arrays with global ids, one-hots as indices, pharmacophore edges
 * sorted by source (stable), pos / norm.  *flags: bit 0 = a ligand lists the same ordered bond twice (no pair tables), bit 1 / 2 = some
 * pose ends on a short ligand / pharmacophore pass (the no_tail hints).  DP_E_BADARG for a malformed view (not one-hot, ids out of range). */
int dp_pack_fill(int32_t n, const DpPoseRef* poses, const DpPackLayout* layout, void* host_block, int32_t* flags);
/* set the pointer fields of *b (and llp / no_tail from `flags`) to a block at `base` - typically the DEVICE copy of the filled block.  *b = the totals
 * dp_pack_query returned (the scalar fields - B, n_*, max_*, n_xs, ph_shared - are not touched here) */
int dp_pack_bind(const DpPackLayout* layout, void* base, int32_t flags, DpBatch* b);

typedef struct DpMlp { const float *w0, *b0, *w3, *b3; } DpMlp;   /* nn.Sequential(Linear, act, Dropout, Linear) */
typedef struct DpConvW {
  const float *w1p, *w2p, *bn_scale, *bn_shift;
  const float *w1n, *w2n;     /* node-side layout or NULL */
  const float* w2x;           /* the second layer packed for the EDGE-side kernel at DP_PREC_F16X3 beside an fp32 `w2p`, or NULL: the convs that stay on
                                 the edge-side kernel in an fp32 model (phore-phore: low degree; torsion bonds; ligand-ligand of large ligands) then
                                 run their radial contraction on the 16-bit matrix pipe as a 2-way fp16 split with 3 cross products (22 significand
                                 bits, fp32 accumulate; rounds 3-4: a 3-way bf16 split with 6 products, DP_PREC_BF16X6, still selectable as a model
                                 precision) - held to the same tolerances as the fp32 kernel (DpWork.edge_bf16x6 = 0 turns it off) */
  const float* bn_scale_x;    /* [D_out] bn_scale * 2^-w for `w2x` (its image stores W2 * 2^w; w = the float in the image's last quad) */
  const float *w1h, *w2h;     /* the node-side convs of an fp32 model once more as two-limb fp16 fragments (dp_pack_conv_weights_node_f16), or NULL:
                                 DP_PREC_F16X3, fp32-grade on the 16-bit matrix pipe (DpWork.node_f16 = 0 turns it off) */
} DpConvW;

/* Device pointers into the uploaded parameter blob.  Reference names: score_model_phore.py:248-291, 524-641. */
typedef struct DpWeights {
  const float* lig_emb;        /* 16 embedding tables concatenated, [sum dims, 20] */
  const float *lig_lin_w, *lig_lin_b;            /* encoder.lig_node_embedding.linear   [20,20] */
  const float* boarder_emb;    /* [5,2,20] */
  const float *boarder_lin_w, *boarder_lin_b;    /* [20,1], [20] */
  const float* ph_emb;         /* 3 tables concatenated [15,20] */
  const float *ph_lin_w, *ph_lin_b;              /* [20,22] (alpha, weight, sigma emb) */
  DpMlp lig_edge, ph_edge, x_edge;               /* 44->20->20, 40->20->20, 73->20->20 */
  DpMlp cdt, pdt, pmt;                           /* cross_distance / phore_direction / phoretype_match transitions */
  DpMlp center_edge, final_edge, tr_final, rot_final;
  const float *tor_w0, *tor_w3;                  /* tor_final_layer: [20,40], [1,20], no bias */
  DpConvW lig_conv[4], ph_conv[4], l2p_conv[4], p2l_conv[4], l2p_norm_conv[4], p2l_norm_conv[4];
  DpConvW final_conv, tor_conv;
  const float* so3_table;      /* [1000] so3._exp_score_norms as float                 utils/so3.py:92-96 */
  const float* torus_table;    /* [5001] torus.score_norm_ as float                    utils/torus.py:82-86 */
  const float *lig_gauss_mu, *ph_gauss_mu, *x_gauss_mu, *center_gauss_mu;     /* GaussianSmearing.offset buffers [20] */
  const float* emb_freq;       /* [10] exp(-k ln(10000)/9) as the reference computes it (float32, torch CPU) */
  float lig_gauss_coeff, ph_gauss_coeff, x_gauss_coeff, center_gauss_coeff;  /* GaussianSmearing.coeff (python floats) */
  float scaler;                /* 100.0 */
  float tr_smin, tr_smax, rot_smin, rot_smax, tor_smin, tor_smax;
  float emb_scale;             /* 10000 */
  int32_t precision;           /* DpPrecision the conv weights were packed for */
} DpWeights;

/* Caller-owned scratch.  Size it with dp_work_bytes(batch), allocate ONE device block (256-byte aligned), carve it with
 * dp_work_bind() and zero the few arrays that need it with dp_work_init(); the array shapes below are what those functions assume. */
typedef struct DpWork {
  /* one-time static tables (dp_prepare) */
  float* lig_x_emb;    /* [n_lig,20] */
  float* ph_x_emb;     /* [n_ph,20]  */
  float* pair_pre;     /* [n_lig,11,20] static part of cross_edge_embedding layer 0 per (atom, phore type) */
  float* pair_fm;      /* [n_lig,11] softplus(phoretype_match_transition) per (atom, phore type) */
  float* pp_pre;       /* [n_pp,20] static part of phore_edge_embedding layer 0 */
  float* pp_sh;        /* [n_pp,9] */
  /* per-forward */
  float* tparams;      /* [B,32]: sigma emb (20) | tr_sigma, rot_sigma, tor_sigma, so3 norm, sqrt(torus norm) | pad */
  int32_t *ll_rowptr, *ll_rowend;   /* [n_lig] */
  int32_t *ll_nbr, *ll_src;         /* [ll_cap] */
  float *ll_emb, *ll_sh;            /* [ll_cap,20], [ll_cap,9] */
  int32_t *ll_pair_rad, *ll_pair_bond;  /* [n_llp] pair (scatter i, gather j) -> lig-lig edge slot of the radius / bond edge, -1 = none */
  float* pp_emb;                    /* [n_pp,20] */
  float *x_emb, *x_sh, *x_nsh;      /* [n_x,20], [n_x,9], [n_x,9] */
  float* x_ew;                      /* [n_x] exp(total_weight) */
  uint16_t *x_emb_bf, *ll_emb_bf;   /* [(n_x + 1), 64], [(ll_cap + 1), 64] 16-bit images of x_emb / ll_emb + flag row (DpConvArgs.edge_emb_bf), written by the
                                       edge-embedding kernels for the node-side convs that read them: bf16 rows of 32 when the model's precision is
                                       DP_PREC_BF16, two-limb fp16 rows of 64 for DP_PREC_F16X3 (fp32 model, node_f16), untouched otherwise */
  float *h_l[2], *h_p[2];           /* [n_lig,100], [n_ph,100] ping-pong */
  float *c_emb, *c_sh;              /* [n_lig,20], [n_lig,9] centre edges */
  float* gpred;                     /* [B,12] */
  int32_t *t_rowptr, *t_rowend, *t_nbr, *t_src;   /* [n_tor], [n_tor], [n_tor*32], [n_tor*32] */
  float *t_emb, *t_sh, *t_hb, *t_q;               /* [n_tor*32,20], [n_tor*32,7], [n_tor,20], [n_tor,40] */
  int32_t* status;                  /* [B] 0 ok; bit 0 (1) the conformer update produced a non-finite pose from FINITE scores (degenerate Kabsch alignment: what
                                       the reference reports as "linalg.svd ... failed to converge", inference.py:211); bit 2 (4) the SCORES the update was
                                       handed were non-finite while the pose was still fine (first cause wins, both are sticky): with the two-limb fp16
                                       kernels (node_f16 / edge_bf16x6, the defaults) this is how a feature, hidden activation or intermediate beyond
                                       fp16's range surfaces (precondition of DP_PREC_F16X3: |T| 2^5 < 65504, i.e. node-side intermediates below ~2047,
                                       hidden activations and features below 65504) - the caller re-runs the batch from its start state with node_f16 = 0,
                                       edge_bf16x6 = 0 (tests/c_host/dp_host_forward.c, sampling._run_parts); with the exact kernels the model itself
                                       diverged; bit1 ligand graph with a repeated (bond or radius) edge
                                       between the same ordered atom pair (not representable in the pair tables of the node-side lig-lig
                                       conv; only raised when that conv runs: pass DpBatch.llp_ptr = NULL for such ligands and the
                                       ligand-ligand convs use the edge-side CSR kernel, which handles multigraphs) */
  /* optional timing hook: if prof_start/prof_stop are hipEvent_t handles, dp_forward records them on the launch stream
   * immediately before / after the tensor-product conv (prof_layer, prof_conv); prof_conv: 0 lig-lig, 1 phore->lig,
   * 2 phore->lig (normals), 3 phore-phore, 4 lig->phore, 5 lig->phore (normals). */
  void *prof_start, *prof_stop;
  int32_t prof_layer, prof_conv;
  /* optional: HOST array of 4 x 6 x 2 hipEvent_t handles, [layer][conv][start, stop] in the numbering above; dp_forward brackets EVERY
   * tensor-product launch of the interaction layers with the pair of the first conv it carries (a fused launch: its first conv; NULL
   * handles are skipped).  For per-launch time tables (bench.py roofline.step_weighted); the events are recorded on the stream the launch
   * runs on. */
  void** prof_events;
  int32_t conv_algo;                /* DpConvAlgo for the dense cross convs of dp_forward */
  /* optional second stream + two hipEvent_t handles (all three or none): dp_forward then runs the pharmacophore-side convs of a
   * layer (and the torsion head) on `aux_stream`, forked from / joined back to the launch stream with the events.  Results are
   * bit-identical; it only helps batches too small to fill the chip with one conv at a time.  Capturable into a hipGraph. */
  void *aux_stream, *ev_fork, *ev_join;
  int32_t aux_mode;                 /* what runs on aux_stream: 0 = the phore-side convs of a layer, the torsion head and the ligand radius graph;
                                       1 = the ligand radius graph only (large homogeneous batches: two matrix-bound convs side by side lose) */
  int32_t ll_node_max_atoms;        /* the ligand-ligand conv runs node-side (pair-dense) while the batch's largest ligand has at most this many
                                       atoms, edge-side (CSR) beyond; 0 = library default */
  int32_t edge_bf16x6;              /* 1 = the edge-side convs of an fp32 model (phore-phore, torsion bonds, ligand-ligand above the pair cap) use
                                       DpConvW.w2x / bn_scale_x when present: fp32-grade operand split on the 16-bit pipe (the default dp_work_bind
                                       sets; the field keeps its round-3 name: until round 4 the split was bf16x6, since round 5 it is the two-limb
                                       fp16 form, DP_PREC_F16X3); 0 = exact fp32 MFMA */
  float* conv_part;                 /* [DP_CONV_PART_FLOATS] scratch of the conv-split launches of small batches (dp_tp_conv_node_fused_split) */
  float* z_noise;                   /* [6 B + n_tor] scratch of DP_NOISE_SEED: one step's z_tr | z_rot | z_tor */
  int32_t path_split;               /* node-side fp32 convs: 0 = library's choice (two workgroups per node group while B x groups <= 256: small batches), 1 = always,
                                       2 = never (A/B runs: profiles/r4_experiments.md) */
  int32_t conv_fuse;                /* node-side convs of a layer that scatter onto the same nodes: 0 = one fused launch per side (default),
                                       1 = one launch per conv, 2 = the two cross convs fused but the ligand-ligand conv on its own launch,
                                       3 = fused launches but never the conv split of small batches (1-3: A/B runs and tests) */
  int32_t node_f16;                 /* 1 = the node-side convs of an fp32 model use DpConvW.w1h / w2h when present and the launch is not path-split:
                                       DP_PREC_F16X3, fp32-grade two-limb fp16 operands on the 16-bit matrix pipe (the default dp_work_bind sets);
                                       0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32) */
  /* optional THIRD stream + one more event (both or none; created with dp_aux_create like the second): small batches whose pharmacophore-side launch takes
   * the conv split then run the layer's phore-phore conv on it, beside the ligand->phore convs instead of in front of them - it writes pad(h_p) + its
   * output to pp_tmp, which the combine launch of the conv split adds up with the other convs' parts.  A latency-bound step's longest chain per layer
   * (phore-phore conv -> ligand->phore convs -> combine) becomes max(...) -> combine (taken while the layer's launches together have at most 640
   * workgroups: 20 poses of 30 atoms / 32 nodes 0.510 -> 0.479 ms per step; from ~30 poses on one more concurrent launch only stretches the others).
   * Needs aux_stream. */
  void *aux2_stream, *ev_join2;
  void* ev_x2;                      /* one more event (optional; with ev_join2 it lets the list launches of a layer's normals convs run somewhere else than in front
                                       of their s0-only launch, see xs_place; eager enqueue with a third stream: the second stream's position at a layer
                                       boundary under a name of its own, so that the third stream waits for both sides side by side) */
  float* pp_tmp;                    /* [n_ph, 100] */
  float* pp_msg;                    /* [n_pp, 100] per-edge messages of a phore-phore conv that runs as one DP_EDGES_LIST launch over the whole batch's edges +
                                       dp_list_finish (dp_forward: when the poses' own edge lists would leave their 128-edge chunks less than 70 % full -
                                       small pharmacophores - and the two-limb fp16 edge-side weights are there; a batch on one pharmacophore, DpBatch.ph_shared: room
                                       for one pose's edges at least, for the layer-0 conv that runs once), or NULL (dp_work_bind: not needed) */
  float* xs_msg;                    /* [2, n_xs, 100] per-edge messages of the listed pairs of the two normals convs of a layer (DP_EDGES_LIST; the two
                                       sides of a layer may run on two streams: one half each) */
  int32_t xs_place;                 /* the DP_EDGES_LIST launches of a layer's two normals convs (small grids, one chunk's latency each) and how their messages
                                       reach the output: 0 = library's choice (4); 4 = list launch in front of its s0-only launch on that launch's stream,
                                       dp_list_reduce launch behind it; 3 = the same, messages folded into the s0-only launch's epilogue (DpConvArgs.list_*:
                                       one launch less, measured 0.7 % slower at 512 poses); 1 = CROSSED: each side's list launch opens the OTHER side's
                                       stream at the start of the layer (needs aux_stream, ev_join2, ev_x2; measured 2.6 % slower: a small launch beside
                                       the big ones stretches them); 2 = both on aux2_stream from the start of the layer (1.6 % slower).  1 / 2 fold. */
  int32_t xn_sparse;                /* the two normals convs of a layer (phore_to_lig_norm / lig_to_phore_norm): 0 = library's choice - when they run on the
                                       f16x3 node-side kernel, the batch carries the pair lists (DpBatch.xs_*) and at most a quarter of the cross pairs is
                                       listed, each is evaluated as its l = 0 part on every pair (DpConvArgs.s0_only, its own node-side launch) + the
                                       other paths on the listed pairs (DP_EDGES_LIST + dp_list_reduce): the same sum, term by term, without the terms
                                       that multiply by an exact zero; 1 = whenever possible, 2 = never (the dense conv; A/B runs and tests) */
  int32_t t_uniform;                /* the caller's statement about THIS forward: every pose carries the same (t_tr, t_rot, t_tor) - what set_time_phore does
                                       in the sampling loop (sampling.py:218; dp_sample states it itself: it writes the times).  0 = not known */
  int32_t ph_share;                 /* 0 = with DpBatch.ph_shared and t_uniform, the layer-0 phore-phore conv once for the batch; 2 = never (A/B runs, tests) */
  const float* t_row;               /* optional DEVICE [3]: THE (t_tr, t_rot, t_tor) of every pose for this forward - set_time_phore (sampling.py:218) folded into
                                       the first kernel of the step: it reads this row instead of DpBatch.t and writes it there for every pose.  Implies
                                       t_uniform.  dp_sample points it at row i of DpSchedule.t.  NULL = the per-pose times of DpBatch.t */
  float* tp_table;                  /* workspace [DP_TP_TABLE_STEPS, 32]: dp_sample's time parameters (sigma embedding, sigmas, score norms) of every step of
                                       the schedule, computed once per run by the same kernel that computes them per pose */
  const float* tp_row;              /* optional DEVICE [32], with t_row: that row's time parameters, computed before - dp_forward then only hands them to
                                       every pose (dp_sample: row i of tp_table) */
} DpWork;
#define DP_TP_TABLE_STEPS 1024

typedef struct DpStepParams {       /* per reverse-diffusion step scalars (sampling.py:205-244), computed by the host */
  float tr_c1, tr_c2;               /* tr_perturb  = tr_c1 * tr_score  + tr_c2 * z_tr   (c1 = g^2 dt or g^2 dt / 2 for ODE) */
  float rot_c1, rot_c2;
  float tor_c1, tor_c2;
  int32_t no_torsion;
} DpStepParams;

/* one-time per batch: embedding sums and other step-invariant tables */
int dp_prepare(const DpBatch* b, const DpWeights* w, DpWork* k, void* stream);
/* score model forward: writes tr [B,3], rot [B,3], tor [n_tor].   score_model_phore.py:294-310 */
int dp_forward(const DpBatch* b, const DpWeights* w, DpWork* k, float* tr, float* rot, float* tor, void* stream);

/* Which launch of dp_forward carries conv `conv` (0 lig_conv, 1 phore_to_lig, 2 phore_to_lig_norm, 3 phore_conv, 4 lig_to_phore,
 * 5 lig_to_phore_norm) of interaction layer `layer` for this batch / workspace: the launch runs convs [*first, *first + *count) (count > 1:
 * a fused node-side launch, dp_tp_conv_node_fused); *node_side (may be NULL) = 1 when that launch is a node-side kernel, 2 when it is the s0-only
 * node-side launch of a normals conv whose other paths run over the listed pairs (DpWork.xn_sparse; the bracketing events then span the DP_EDGES_LIST
 * launch and the s0-only launch).  The same conv
 * numbering as DpWork.prof_conv: the prof_start / prof_stop events bracket exactly this launch.  Pure host logic, no GPU work. */
int dp_forward_launch_of(const DpBatch* b, const DpWeights* w, const DpWork* k, int32_t layer, int32_t conv, int32_t* first, int32_t* count,
                         int32_t* node_side);

/* SDE/ODE perturbation + modify_conformer for every pose (updates b->pos / b->norm in place).
 * sampling.py:223-254, diffusion_utils.py:23-79.  z_* may be NULL (treated as zeros). */
int dp_conformer_update(const DpBatch* b, const DpStepParams* sp, const float* tr, const float* rot, const float* tor,
                        const float* z_tr, const float* z_rot, const float* z_tor, int32_t* status, void* stream);
/* randomize_position (sampling.py:16-63): theta [n_tor] random torsion angles, rot_mats [B,3,3] random rotations, tr [B,3]
 * translation (NULL = no_random).  The random numbers are drawn by the caller, exactly as the reference draws them. */
int dp_randomize_position(const DpBatch* b, const float* theta, const float* rot_mats, const float* tr, int32_t no_torsion, void* stream);

/* Pharmacophore-fit SURROGATE for ranking poses on the device.  Stands where the reference shells out to the
 * closed AncPhore binary (`calculate_fitscore`, src/utils/sampling.py:447-498) and is NOT that program: a Gaussian-volume overlap
 * with the constants of `get_perfect_similarity` (src/inference.py:273-295); formula in dp_update.hip.  `pos` = candidate ligand
 * coordinates [n_lig,3] or NULL for the batch's current state; `score` [B]. */
int dp_fit_overlap(const DpBatch* b, const float* pos, float ex_lambda, float* score, void* stream);
/* dp_forward + dp_conformer_update */
int dp_sample_step(const DpBatch* b, const DpWeights* w, DpWork* k, const DpStepParams* sp, float* tr, float* rot, float* tor,
                   const float* z_tr, const float* z_rot, const float* z_tor, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * What a host without diffphore_amd/engine.py needs (tests/c_host/dp_host_forward.c uses nothing else):
 * ------------------------------------------------------------------------------------------------------------------ */
/* bytes of the single device block that holds every DpWork array plus the tr / rot / tor outputs of `b` (0 on bad input) */
size_t dp_work_bytes(const DpBatch* b);
/* carve `base` (device pointer, 256-byte aligned, >= dp_work_bytes) into *out and the three output arrays; touches no memory */
int dp_work_bind(const DpBatch* b, void* base, size_t bytes, DpWork* out, float** tr, float** rot, float** tor);
/* stream-ordered zeroing of the arrays that must start at zero (feature ping-pong buffers, tparams, status) */
int dp_work_init(const DpBatch* b, DpWork* k, void* stream);

/* The parameter blob: every tensor DpWeights points to lives in ONE flat float array (one RCCL broadcast at start-up).  Its layout
 * is defined here: segment names ("lig_conv.2.w2p", "x_edge.w0", "so3_table", ...), offsets and sizes in floats.  Conv segments
 * hold the output of dp_pack_conv_weights / dp_pack_conv_weights_node / dp_pack_bn; the others are the reference tensors
 * flattened row-major (embedding tables concatenated in feature order; src/models/score_model_phore.py:248-291,524-641). */
typedef struct DpSegment { char name[48]; int64_t offset; int64_t numel; } DpSegment;
int dp_weights_segments(int32_t precision, DpSegment* out, int32_t cap);   /* returns the number of segments (also when out == NULL) */
int64_t dp_weights_blob_floats(int32_t precision);
typedef struct DpModelScalars {
  float lig_gauss_coeff, ph_gauss_coeff, x_gauss_coeff, center_gauss_coeff;  /* GaussianSmearing.coeff = -0.5 / (offset[1]-offset[0])^2 */
  float scaler;                                                                /* 100.0 */
  float tr_smin, tr_smax, rot_smin, rot_smax, tor_smin, tor_smax;              /* model_parameters.yml sigma ranges */
  float emb_scale;                                                             /* 10000 */
} DpModelScalars;
/* fill *w with pointers into the DEVICE blob (layout above) and the scalars */
int dp_weights_bind(const float* blob, int32_t precision, const DpModelScalars* sc, DpWeights* w);
/* HOST pointers: e3nn BatchNorm(eval) of conv kind `kind` as out = x*scale + shift per output channel (running_mean / bias cover
 * the 0e channels only; e3nn/nn/_batchnorm.py semantics) */
int dp_pack_bn(int32_t kind, const float* running_mean, const float* running_var, const float* weight, const float* bias, float eps,
               float* scale, float* shift);

/* The whole reverse-diffusion loop of sampling_phore (src/utils/sampling.py:204-255) as one call: for step i, set every pose's
 * diffusion times to t[i], run the score model, perturb and update the conformers.  Nothing synchronises with the host. */
typedef struct DpSchedule {
  int32_t steps;
  const DpStepParams* sp;        /* HOST array [steps]: the per-step scalars the reference computes on the host (:205-244) */
  const float* t;                /* DEVICE [steps,3]: t_tr, t_rot, t_tor of every step (set_time_phore broadcasts them to all poses) */
  const float *z_tr, *z_rot;     /* DEVICE pre-drawn normals [steps,B,3] or NULL */
  const float* z_tor;            /* DEVICE [steps,n_tor] or NULL */
  /* noise_mode DP_NOISE_TABLES (0): the z_* tables above (NULL = zeros) - the parity path, identical noise for a CPU run.
   * noise_mode DP_NOISE_SEED (1): z_* are ignored; every step's normals are drawn on the device into DpWork.z_noise by
   * dp_noise_normals(seed, step, ...) - a pure function of (seed, pose uid, step, component), so a pose's trajectory does not depend on
   * the rank / device batch / position it is processed at.  Stands where the reference draws inside its loop (sampling.py:230-243). */
  int32_t noise_mode;
  int32_t uid_base;              /* pose p's uid = pose_uid ? pose_uid[p] : uid_base + p */
  const int32_t* pose_uid;       /* DEVICE [B] or NULL */
  uint64_t seed;
} DpSchedule;
#define DP_CONV_PART_FLOATS (2 * 3 * 4096 * 100)   /* two sides (they may run on two streams) x 3 convs x 256 node groups x 16 nodes x row stride 100 */
#define DP_NOISE_TABLES 0
#define DP_NOISE_SEED 1
#define DP_RNG_STEP_START 0xFFFFFFFFu    /* the `step` word of the start-randomisation draws */

/* N(0, 1) draws of reverse step `step` for every pose of the batch: z_tr [B,3], z_rot [B,3], z_tor [n_tor] (DEVICE; any may be NULL).
 * Philox-4x32-10, key = seed, counter = (pose uid, step, block, stream) - layout in csrc/dp_rng.hip. */
int dp_noise_normals(uint64_t seed, int32_t step, const DpBatch* b, const int32_t* pose_uid, int32_t uid_base, float* z_tr, float* z_rot,
                     float* z_tor, void* stream);
/* the draws of randomize_position (sampling.py:16-63) on the device, same keying: theta [n_tor] ~ U(-pi, pi), rot_mats [B,3,3] uniform
 * on SO(3), tr [B,3] ~ N(0, tr_sigma_max) - the arguments dp_randomize_position takes (any may be NULL). */
int dp_randomize_draws(uint64_t seed, const DpBatch* b, const int32_t* pose_uid, int32_t uid_base, float tr_sigma_max, float* theta,
                       float* rot_mats, float* tr, void* stream);
/* the generator itself on the host (known-answer tests) */
int dp_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int dp_sample(const DpBatch* b, const DpWeights* w, DpWork* k, const DpSchedule* s, float* tr, float* rot, float* tor, void* stream);
/* the same loop captured into a hipGraph (all pointers are baked in): create once per (batch, workspace, schedule), replay with
 * dp_sample_graph_launch; bit-identical to dp_sample */
int dp_sample_graph_create(const DpBatch* b, const DpWeights* w, DpWork* k, const DpSchedule* s, float* tr, float* rot, float* tor,
                           void** graph_exec);
int dp_sample_graph_launch(void* graph_exec, void* stream);
int dp_sample_graph_destroy(void* graph_exec);

/* the optional second stream + fork / join events of DpWork (aux_stream, ev_fork, ev_join), created / destroyed by the library's own HIP
 * runtime; the caller keeps them alive while any DpWork that names them is in use */
int dp_aux_create(void** aux_stream, void** ev_fork, void** ev_join);
int dp_aux_destroy(void* aux_stream, void* ev_fork, void* ev_join);

/* sizeof() of DpConvArgs, DpBatch, DpWeights, DpWork, DpStepParams, DpMlp, DpSchedule (binding self-check); returns how many exist */
int dp_struct_sizes(int32_t* out, int32_t n);

int dp_abi_version(void);

/* What this build of the library is: compile-time switches that change what the kernels compute or time (all 0 in a product build) and a
 * hash of the kernel sources it was compiled from.  A host should refuse a library whose nabl / ablate are non-zero unless it asked for an
 * ablation build (those kernels compute wrong numbers by design). */
typedef struct DpBuildInfo {
  int32_t abi_version;
  int32_t nabl;        /* -DDP_NABL   (node-side kernel ablations) */
  int32_t ablate;      /* -DDP_ABLATE (edge-side kernel ablations) */
  int32_t prof;        /* -DDP_PROF   (cycle stamps)               */
  char source_hash[20];   /* first 16 hex digits of sha256 over the .hip / .h files of csrc/ and include/ (bench.kernel_source_hash), NUL-terminated;
                             "unknown" when built without the build script */
} DpBuildInfo;
int dp_build_info(DpBuildInfo* out);

#ifdef __cplusplus
}
#endif
#endif /* DIFFPHORE_AMD_H */
