#!/usr/bin/env python3
"""Golden vectors for the conformer update at BASELINE configs[2]'s shape (80 atoms x 25 torsions) and at the kernel limit shape
(128 atoms x 40 torsions), produced by IMPORTING the reference's own `utils/diffusion_utils.py::modify_conformer`
(+ `utils/torsion.py`, `utils/geometry.py`) in the build container.  Data only (inputs + expected outputs).

    python tests/golden/make_torsion25_golden.py      ->  tests/golden/ref_torsion25.npz

The sequential bond loop of `modify_conformer_torsion_angles` (/root/reference/src/utils/torsion.py:64-109) is where a long chain
of rotations would expose an ordering / accumulation difference; ref_utils.npz stops at 7 torsions.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _Graph, _stub_modules, make_tree  # noqa: E402


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    from utils import diffusion_utils  # noqa: E402  (reference import)
    out = {}
    g = torch.Generator().manual_seed(2525)
    rng = np.random.default_rng(2525)
    for case, (n, T) in enumerate([(80, 25), (80, 25), (128, 40), (45, 12)]):
        pos, edge_index, edge_mask, mask_rotate = make_tree(rng, n, T)
        Treal = int(edge_mask.sum())
        assert Treal == T, (Treal, T)
        norm = rng.normal(size=(n, 33)).astype(np.float32)
        norm[rng.random(size=(n, 11)).repeat(3, axis=1) < 0.6] = 0.0
        theta = rng.uniform(-np.pi, np.pi, size=Treal).astype(np.float32)
        if case == 1:
            theta *= 0.05                    # late-schedule sized updates
            theta[3] = 0.0                   # the `== 0: continue` branch (torsion.py:88-89)
        data = _Graph()
        data["ligand"].pos = torch.from_numpy(pos).clone()
        data["ligand"].x = torch.zeros(n, 16)
        data["ligand"].norm = torch.from_numpy(norm).clone()
        data["ligand"].edge_mask = torch.from_numpy(edge_mask)
        data["ligand"].mask_rotate = mask_rotate
        data["ligand", "ligand"].edge_index = torch.from_numpy(edge_index)
        tr = torch.randn(1, 3, generator=g)
        rot = torch.randn(3, generator=g) * 0.4
        diffusion_utils.modify_conformer(data, tr, rot, theta)
        pre = f"c{case}_"
        out[pre + "pos"], out[pre + "edge_index"], out[pre + "edge_mask"] = pos, edge_index, edge_mask
        out[pre + "mask_rotate"], out[pre + "norm"], out[pre + "theta"] = mask_rotate, norm, theta
        out[pre + "mc_tr"], out[pre + "mc_rot"] = tr.numpy(), rot.numpy()
        out[pre + "mc_pos_out"] = data["ligand"].pos.numpy()
        out[pre + "mc_norm_out"] = data["ligand"].norm.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_torsion25.npz"), **out)
    print("wrote ref_torsion25.npz:", {k: v.shape for k, v in out.items() if k.endswith("pos")})


if __name__ == "__main__":
    main()
