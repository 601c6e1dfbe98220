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

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DP_OK 0
#define DP_E_BADARG (-1)
#define DP_E_UNSUPPORTED (-2)
#define DP_E_TOOLARGE (-3)

#define DP_ABI_VERSION 1

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
  DP_EDGES_CENTER = 3   /* scatter = pose, one edge per ligand atom                         (:381-406) */
};

typedef struct DpConvArgs {
  int32_t kind;          /* DpConvKind */
  int32_t mode;          /* DpEdgeMode */
  int32_t B;             /* poses */
  int32_t slices;        /* workgroups per pose (>=1); scatter nodes of a pose are split evenly between them */
  /* pose CSRs (B+1 entries each) */
  const int32_t* scat_ptr;   /* scatter-node offsets per pose (ligand atoms / phore nodes / rotatable bonds / poses) */
  const int32_t* lig_ptr;    /* ligand atom offsets per pose */
  const int32_t* ph_ptr;     /* phore node offsets per pose */
  const int32_t* x_ptr;      /* dense cross-edge offsets per pose (sum N_l*N_p) */
  /* DP_EDGES_CSR only */
  const int32_t* rowptr;     /* [n_scatter] first edge slot of each scatter node */
  const int32_t* rowend;     /* [n_scatter] one-past-last edge slot */
  const int32_t* nbr;        /* [n_edge_slots] gather node of each edge */
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
} DpConvArgs;

/* Fused gather -> radial MLP (MFMA) -> Clebsch-Gordan tensor product -> per-node segmented mean -> e3nn BatchNorm(eval).
 * Replaces TensorProductConvLayer.forward, score_model_phore.py:134-149. */
int dp_tp_conv(const DpConvArgs* args, void* stream);

/* numbers the host packer needs */
int dp_conv_w1p_floats(int32_t kind);
int dp_conv_w2p_floats(int32_t kind);
int dp_conv_dims(int32_t kind, int32_t* d_in, int32_t* d_out, int32_t* numel, int32_t* n_attr, int32_t* hidden);
/* HOST pointers: reorder one TensorProductConvLayer's fc parameters (reference state_dict layout) into MFMA operand order */
int dp_pack_conv_weights(int32_t kind, const float* fc0_w, const float* fc0_b, const float* fc3_w, const float* fc3_b,
                         float* w1p, float* w2p);

int dp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DIFFPHORE_AMD_H */
