#!/usr/bin/env python3
"""Golden vectors for the `.phore` parser + pharmacophore graph builder, from the reference's own code.

    python tests/golden/make_phore_golden.py      ->  tests/golden/ref_phore_graph.npz  (+ the example data file it parses)

`src/datasets/process_pharmacophore.py` is imported for real (`parse_phore`, `get_phore_graph`, `phore_featurizer`); its
module-level imports of rdkit and of two `datasets.process_mols` helpers are satisfied by inert placeholders that this path never
calls.  Inputs: the example pharmacophore shipped with the reference (`examples/phore/sQC_QFA_complex.phore`, a data file, copied
next to the vectors so the test runs where /root/reference does not exist) and two small generated records (written below,
also stored in the npz as text).
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _placeholder(*a, **k):
    raise RuntimeError("placeholder: never called on the parser / graph-builder path")


def main():
    chem = types.ModuleType("rdkit.Chem")
    rd = types.ModuleType("rdkit")
    rd.Chem = chem
    pm = types.ModuleType("datasets.process_mols")
    pm.generate_ligand_phore_feat = pm.analyze_phorefp = _placeholder
    pkg = types.ModuleType("datasets")          # the reference's `datasets` package (an installed HF package has the same name)
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules.update({"rdkit": rd, "rdkit.Chem": chem, "datasets": pkg, "datasets.process_mols": pm})
    sys.path.insert(0, REF)
    import importlib
    ref = importlib.import_module("datasets.process_pharmacophore")
    from diffphore_amd.hetero import HeteroData

    rng = np.random.default_rng(20250321)
    types_ = ["MB", "HD", "AR", "PO", "HA", "HY", "NE", "CV", "CR", "XB"]
    lines = ["generated_A"]
    for i in range(7):
        t = types_[int(rng.integers(len(types_)))]
        xyz = rng.normal(scale=3.0, size=3)
        has_norm = int(rng.random() < 0.6)
        nrm = xyz + rng.normal(size=3) if has_norm else np.zeros(3)
        lines.append("\t".join([t, f"{rng.choice([1.0, 0.7]):.3f}", f"{rng.uniform(0.5, 1.5):.3f}", "1"] + [f"{v:.3f}" for v in xyz]
                               + [str(has_norm)] + [f"{v:.3f}" for v in nrm] + [str(i), "1"]))
    for i in range(9):
        xyz = rng.normal(scale=6.0, size=3)
        lines.append("\t".join(["EX", "0.837", "0.5", "1"] + [f"{v:.3f}" for v in xyz] + ["0", "0.000", "0.000", "0.000", "0", "1"]))
    lines += ["$$$$", "generated_B", "HA\t1.000\t1.000\t1\t0.000\t0.000\t0.000\t1\t0.000\t1.000\t0.000\t1\t1",
              "EX\t0.837\t0.5\t1\t40.000\t0.000\t0.000\t0\t0.000\t0.000\t0.000\t0\t1", "$$$$", ""]
    gen_text = "\n".join(lines)
    out = {"generated_text": np.asarray(gen_text)}
    with tempfile.TemporaryDirectory() as td:
        gen = os.path.join(td, "gen.phore")
        open(gen, "w").write(gen_text)
        k = 0
        for src in ("/root/reference/examples/phore/sQC_QFA_complex.phore", gen):
            for ph in ref.parse_phore(phore_file=src):
                g = HeteroData()
                ref.get_phore_graph(ph, g)
                out[f"g{k}_id"] = np.asarray(ph.id)
                out[f"g{k}_x"] = g["phore"].x.numpy()
                out[f"g{k}_pos"] = g["phore"].pos.numpy()
                out[f"g{k}_norm"] = g["phore"].norm.numpy()
                out[f"g{k}_edge_index"] = g["phore", "phore"].edge_index.numpy()
                print(k, ph.id, out[f"g{k}_x"].shape, out[f"g{k}_edge_index"].shape)
                k += 1
    out["n_graphs"] = np.int64(k)
    np.savez_compressed(os.path.join(HERE, "ref_phore_graph.npz"), **out)
    shutil.copyfile("/root/reference/examples/phore/sQC_QFA_complex.phore", os.path.join(HERE, "sQC_QFA_complex.phore"))
    os.chmod(os.path.join(HERE, "sQC_QFA_complex.phore"), 0o644)


if __name__ == "__main__":
    main()
