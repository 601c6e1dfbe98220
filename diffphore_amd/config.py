"""Effective hyper-parameters of the released DiffPhore score model and the irreps / tensor-product path tables.

Values: /root/reference/weights/diffphore_calibrated_warmuped_ft/model_parameters.yml mapped through
`get_model` (src/utils/utils.py:113-176) onto the constructor defaults (src/models/score_model_phore.py:166-180).
Keys the reference's `get_model` never forwards (`multiple`, `phoretype_match`, `sh_lmax`, `phore_max_radius`,
`center_max_distance`) keep their constructor defaults.

The HIP kernels are specialised for exactly this configuration (ns=20, nv=10, 4 layers, lmax=2, first-order irreps);
`ModelConfig.check_supported()` rejects anything else loudly.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

NS, NV = 20, 10
SIGMA_EMBED_DIM = 20
DIST_EMBED_DIM = 20
NUM_PHORETYPE = 11


@dataclass
class ModelConfig:
    ns: int = 20
    nv: int = 10
    num_conv_layers: int = 4
    sigma_embed_dim: int = 20
    distance_embed_dim: int = 20
    cross_distance_embed_dim: int = 20
    in_lig_edge_features: int = 4
    sh_lmax: int = 2
    lig_max_radius: float = 5.0
    phore_max_radius: float = 5.0
    cross_max_distance: float = 25.0
    center_max_distance: float = 30.0
    dropout: float = 0.1                     # eval => identity
    embedding_type: str = "sinusoidal"
    embedding_scale: float = 10000.0
    scale_by_sigma: bool = True
    batch_norm: bool = True
    use_second_order_repr: bool = False
    consider_norm: bool = True
    auto_phorefp: bool = False
    angle_match: bool = True
    new: bool = True
    phoretype_match: bool = True
    boarder: bool = True
    by_radius: bool = False
    clash_cutoff: Tuple[float, ...] = (1.0, 2.0, 3.0, 4.0, 5.0)
    use_att: bool = False
    use_phore_match_feat: bool = True
    atom_weight: str = "phore"
    cross_distance_transition: bool = True
    phore_direction_transition: bool = True
    phoretype_match_transition: bool = True
    scaler: float = 100.0
    multiple: bool = False
    no_torsion: bool = False
    # diffusion
    tr_sigma_min: float = 0.1
    tr_sigma_max: float = 5.0
    rot_sigma_min: float = 0.1
    rot_sigma_max: float = 1.5
    tor_sigma_min: float = 0.0314
    tor_sigma_max: float = 3.14
    inference_steps: int = 20

    def check_supported(self):
        want = ModelConfig()
        bad = [k for k in ("ns", "nv", "num_conv_layers", "sigma_embed_dim", "distance_embed_dim",
                           "cross_distance_embed_dim", "in_lig_edge_features", "sh_lmax", "embedding_type",
                           "batch_norm", "use_second_order_repr", "consider_norm", "auto_phorefp", "angle_match",
                           "new", "phoretype_match", "boarder", "by_radius", "use_att", "use_phore_match_feat",
                           "atom_weight", "cross_distance_transition", "phore_direction_transition",
                           "phoretype_match_transition", "multiple", "scale_by_sigma",
                           # cut-offs: the ligand radius graph and the torsion-bond graph are built with 5 A inside the kernels (dp_graph.hip:
                           # lig_graph_kernel, tor_graph_kernel); the other three only set the range of a Gaussian basis, which the kernels read
                           # from the state_dict's offsets - held to the released values all the same: nothing else has been checked
                           "lig_max_radius", "phore_max_radius", "cross_max_distance", "center_max_distance")
               if getattr(self, k) != getattr(want, k)]
        if tuple(self.clash_cutoff) != want.clash_cutoff:
            bad.append("clash_cutoff")
        if bad:
            raise NotImplementedError(
                f"diffphore_amd HIP kernels are specialised for the released DiffPhore configuration; unsupported: {bad}")


# ------------------------------------------------------------------------------------------------------------------
# irreps and FullyConnectedTensorProduct path tables (e3nn rules; src/models/score_model_phore.py:586-599,256-285)
# ------------------------------------------------------------------------------------------------------------------
# an irrep is (mul, l, parity); parity +1 = 'e', -1 = 'o'
SEQ = [
    [(20, 0, 1)],
    [(20, 0, 1), (10, 1, -1)],
    [(20, 0, 1), (10, 1, -1), (10, 1, 1)],
    [(20, 0, 1), (10, 1, -1), (10, 1, 1), (20, 0, -1)],
]
SH = [(1, 0, 1), (1, 1, -1), (1, 2, 1)]
FINAL_OUT = [(2, 1, -1), (2, 1, 1)]
TOR_OUT = [(20, 0, -1), (20, 0, 1)]
# FullTensorProduct(1x0e+1x1o+1x2e, 2e) output, sorted by (l, parity) with odd first, un-simplified (45 floats)
TOR_SH = [(1, 0, 1), (1, 1, -1), (1, 1, 1), (1, 2, -1), (1, 2, 1), (1, 2, 1), (1, 3, -1), (1, 3, 1), (1, 4, 1)]


def dim(irreps) -> int:
    return sum(m * (2 * l + 1) for m, l, _ in irreps)


def offsets(irreps) -> List[int]:
    o, out = 0, []
    for m, l, _ in irreps:
        out.append(o)
        o += m * (2 * l + 1)
    return out


@dataclass
class TPPath:
    i_in: int
    i_sh: int
    i_out: int
    l_in: int
    l_sh: int
    l_out: int
    mul_in: int
    mul_out: int
    w_off: int
    coef: float = 0.0
    in_off: int = 0
    sh_off: int = 0
    out_off: int = 0


def tp_paths(irreps_in, irreps_sh, irreps_out) -> Tuple[List[TPPath], int]:
    """Instruction order: in-slot outer, sh-slot middle, out-slot inner; weight blocks [mul_in, mul_out] row-major."""
    io, so, oo = offsets(irreps_in), offsets(irreps_sh), offsets(irreps_out)
    paths, off = [], 0
    for i1, (m1, l1, p1) in enumerate(irreps_in):
        for i2, (m2, l2, p2) in enumerate(irreps_sh):
            assert m2 == 1
            for i3, (m3, l3, p3) in enumerate(irreps_out):
                if abs(l1 - l2) <= l3 <= l1 + l2 and p3 == p1 * p2:
                    paths.append(TPPath(i1, i2, i3, l1, l2, l3, m1, m3, off, 0.0, io[i1], so[i2], oo[i3]))
                    off += m1 * m3
    for p in paths:
        fan = sum(q.mul_in for q in paths if q.i_out == p.i_out)
        p.coef = math.sqrt((2 * p.l_out + 1) / fan)
    return paths, off


def layer_irreps(l: int):
    return SEQ[min(l, 3)], SEQ[min(l + 1, 3)]


def layer_numel(l: int) -> int:
    return tp_paths(layer_irreps(l)[0], SH, layer_irreps(l)[1])[1]


def bn_sizes(irreps) -> Tuple[int, int]:
    """(# scalar (0e) channels -> running_mean/bias, # multiplicity channels -> running_var/weight)."""
    return sum(m for m, l, p in irreps if l == 0 and p == 1), sum(m for m, _, _ in irreps)


CONV_FAMILIES = ("lig_conv_layers", "phore_conv_layers", "lig_to_phore_conv_layers", "phore_to_lig_conv_layers",
                 "lig_to_phore_norm_conv_layers", "phore_to_lig_norm_conv_layers")

LIG_FEATURE_DIMS = [119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2]
PHORE_FEATURE_DIMS = [11, 2, 2]
