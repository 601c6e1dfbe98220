#!/usr/bin/env python3
"""Golden vectors for two self-contained functions of the fitting driver, from the reference's own `src/inference.py`:
`get_perfect_similarity` (:273-311) and `analyze_results` (:321-350).

    python tests/golden/make_fit_golden.py      ->  tests/golden/ref_fit_helpers.npz

`inference.py` imports most of the code base at module level (rdkit, torch_geometric, the dataset classes, ...).  None of that is
executed by the two functions, so every absent package is satisfied by a permissive placeholder module (attributes resolve to inert
objects); `utils.so3` / `utils.torus` (which build tables for minutes and write a cache at import) are replaced by the committed
table lookups.  The functions themselves run unmodified."""
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
ABSENT = ("rdkit", "torch_geometric", "spyrmsd", "Bio", "prody", "e3nn", "torch_scatter", "torch_cluster", "torch_sparse", "openbabel", "wandb", "esm")


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, k):
        return _Inert()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _Module(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Inert()


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        return importlib.machinery.ModuleSpec(name, self, is_package=True) if name.split(".")[0] in ABSENT else None

    def create_module(self, spec):
        m = _Module(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


def main():
    from oracle import tables
    sys.meta_path.insert(0, _Finder())
    pkg = types.ModuleType("datasets")                       # the reference's package, not the installed HF one
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = pkg
    sys.path.insert(0, REF)
    import utils as up
    for nm, fn in (("so3", lambda eps: torch.from_numpy(tables.so3_score_norm(eps.numpy()))), ("torus", lambda s: tables.torus_score_norm(s))):
        m = types.ModuleType("utils." + nm)
        m.score_norm = fn
        sys.modules["utils." + nm] = m
        setattr(up, nm, m)
    import inference as ref

    from diffphore_amd import synthetic
    out = {}
    rng = np.random.default_rng(3)
    sims = []
    for k in range(6):
        (g,) = synthetic.synthetic_pairs(40 + k, 1, 14, 2, 10 + k)
        vol = g["phore"].phoretype.sum(0)
        ph = torch.as_tensor(rng.integers(0, 3, size=11)).float() if k else vol.clone()
        g["ligand"].ph = ph
        g.name = f"g{k}"
        out[f"sim{k}_ph"] = ph.numpy()
        sims.append(ref.get_perfect_similarity(g))
    out["sims"] = np.asarray(sims, dtype=np.float64)
    print("similarities:", sims)

    res = {"name": ["t1__a", "t1__b", "t2__c", "t3__d"], "run_time": [1.0, 2.5, 3.0, 0.5],
           "fitscore": [[0.1, 0.7, 0.3], [0.9, 0.2, 0.8, 0.1, 0.5, 0.6], [-2.0, -2.0], [0.7, 0.7, 0.1, 0.0, 0.2, 0.3]]}
    with tempfile.TemporaryDirectory() as td:
        args = types.SimpleNamespace(out_dir=td, cutoff=0.6, report_results=False)
        ref.analyze_results(args, {k: list(v) for k, v in res.items()})
        out["csv"] = np.asarray(open(os.path.join(td, "ranked_results.csv")).read())
        out["csv_cut"] = np.asarray(open(os.path.join(td, "ranked_results_gt0.6.csv")).read())
    np.savez_compressed(os.path.join(HERE, "ref_fit_helpers.npz"), **out)
    print(str(out["csv"]))


if __name__ == "__main__":
    main()
