#!/usr/bin/env python3
"""End-to-end use of the drop-in on synthetic complexes (no dataset, no checkpoint, no RDKit, no AncPhore needed):

    python examples/fit_synthetic.py --complexes 32 --samples 40 --out /tmp/diffphore_demo

mirrors `python src/inference.py ...` of the reference from the point where the complex graphs exist: model construction through
`get_model`, `fit` (randomise -> 20-step reverse diffusion -> score -> ranked SDF + json log per complex), `analyze_results`
(`ranked_results.csv`).  With a real checkpoint replace `init_random()` by `load_state_dict(torch.load(path))`; with real inputs
replace the synthetic graphs by the reference's featurised `HeteroData` objects (same fields); with AncPhore pass your wrapper as
`scorer=` - here the on-device Gaussian-overlap surrogate ranks the poses."""
import argparse
import os
import time
import types

import numpy as np
import torch

from diffphore_amd import config, sdf_io, synthetic
from diffphore_amd.diffusion import t_to_sigma_fn
from diffphore_amd.fit import analyze_results, fit
from diffphore_amd.fitness import overlap_scorer
from diffphore_amd.score_model import get_model


def carbon_skeleton(g, name):
    """A V2000 template for the synthetic ligand (one carbon per atom) so that ranked poses can be written as SDF."""
    pos = g["ligand"].pos.numpy().astype(np.float64) + g.original_center.numpy().reshape(1, 3)
    m = sdf_io.Mol(title=name, program="  synthetic", comment="", coords=pos, symbols=["C"] * len(pos),
                   atom_tail=[" C   0  0  0  0  0  0  0  0  0  0  0  0"] * len(pos))
    seen = set()
    for a, b in g["ligand", "ligand"].edge_index.numpy().T:
        if (min(a, b), max(a, b)) not in seen:
            seen.add((min(a, b), max(a, b)))
            m.bonds.append([int(a) + 1, int(b) + 1, 1])
            m.bond_tail.append("  0")
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--complexes", type=int, default=16)
    ap.add_argument("--samples", type=int, default=40)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default="/tmp/diffphore_demo")
    ap.add_argument("--precision", default=None, help="f32 (default) | bf16x6 | bf16x3")
    a = ap.parse_args()
    cfg = config.ModelConfig()
    args = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")},
                                 no_torsion=False, run_dir=a.out, out_dir=a.out, sample_per_complex=a.samples, inference_steps=a.steps,
                                 overwrite=True, batch_size=20, no_random=False, ode=False, no_final_step_noise=True, cutoff=None,
                                 max_radius=5.0, scale_by_sigma=True, boarder=True, use_phore_match_feat=True, atom_weight="phore", scaler=100.0)
    if a.precision:
        os.environ["DIFFPHORE_PRECISION"] = a.precision
    t_to_sigma = t_to_sigma_fn(args)
    model = get_model(args, torch.device("cuda"), t_to_sigma, no_parallel=True).init_random()
    graphs = synthetic.synthetic_pairs(2025, a.complexes, (18, 40), (2, 8), (20, 79))
    for i, g in enumerate(graphs):
        g.name = f"target{i // 4}__ligand{i}"
        g.mol = carbon_skeleton(g, g.name)
    t0 = time.time()
    results = fit(args, model, graphs, "cuda", t_to_sigma, scorer=overlap_scorer)
    dt = time.time() - t0
    df = analyze_results(args, results)
    n = a.complexes * a.samples
    print(df[["target", "ligand", "max_fitscore", "top5_mean_fitscore"]].head(8).to_string(index=False))
    print(f"{a.complexes} complexes x {a.samples} samples x {a.steps} steps in {dt:.2f} s ({n / dt:.0f} poses/s incl. scoring and SDF writing); results in {a.out}")


if __name__ == "__main__":
    main()
