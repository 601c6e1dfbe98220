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


def _import_reference_inference():
    """`import inference` with (a) the FUNCTIONAL adapters of make_model_golden.py for what the model / sampler path executes and
    (b) permissive placeholders for every other absent package."""
    sys.path.insert(0, HERE)
    import make_model_golden as mm
    so3, torus = mm.install_adapters()
    dims = {k: getattr(sys.modules[k], a) for k, a in (("datasets.process_mols", "lig_feature_dims"), ("datasets.process_pharmacophore", "phore_feature_dims"))}
    for k in [k for k in list(sys.modules) if k.split(".")[0] in ("rdkit", "spyrmsd", "datasets") or k in ("torch_geometric.nn", "torch_geometric.nn.data_parallel", "torch_geometric.data")]:
        del sys.modules[k]
    tg = sys.modules["torch_geometric"]
    tg.__path__ = []
    for a in ("nn", "data"):
        if hasattr(tg, a):
            delattr(tg, a)
    sys.meta_path.insert(0, _Finder())
    pkg = types.ModuleType("datasets")                       # the reference's package, not the installed HF one
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = pkg
    for k, a in (("datasets.process_mols", "lig_feature_dims"), ("datasets.process_pharmacophore", "phore_feature_dims")):
        m = _Module(k)          # everything else these two modules export is chemistry the driver path never executes
        setattr(m, a, dims[k])
        sys.modules[k] = m

    from diffphore_amd.hetero import Batch

    class DataLoader:            # torch_geometric.loader.DataLoader on the duck-typed graphs: in-order chunks, PyG-style collation
        def __init__(self, dataset=None, batch_size=1, shuffle=False, **kw):
            self.d, self.bs = dataset, batch_size

        def __len__(self):
            n = self.d.len() if hasattr(self.d, "len") else len(self.d)
            return (n + self.bs - 1) // self.bs

        def __iter__(self):
            n = self.d.len() if hasattr(self.d, "len") else len(self.d)
            get = self.d.get if hasattr(self.d, "get") else self.d.__getitem__
            for lo in range(0, n, self.bs):
                items = [get(i) for i in range(lo, min(lo + self.bs, n))]
                for g in items:   # a batch-of-1 carries `mask_rotate` as [array]; modify_conformer un-nests exactly one level
                    mr = g["ligand"].mask_rotate
                    if isinstance(mr, list):
                        g["ligand"].mask_rotate = mr[0]
                b = Batch.from_data_list(items)
                for a in ("name", "mol", "phore_file"):      # PyG collates python attributes into lists
                    if all(a in g for g in items):
                        setattr(b, a, [getattr(g, a) for g in items])
                if all("original_center" in g for g in items):
                    b.original_center = torch.cat([torch.as_tensor(g.original_center).reshape(-1, 3) for g in items])
                yield b

    mm._module("torch_geometric.loader", DataLoader=DataLoader, DataListLoader=_Inert)
    sys.path.insert(0, REF)
    import utils as up
    up.so3, up.torus = so3, torus
    import inference as ref
    return ref, mm


def main():
    ref, mm = _import_reference_inference()

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
    print(str(out["csv"]))

    # ---- the driver itself: inference.py::fit around the reference model and sampler, AncPhore replaced by a geometric score ----
    import json
    import yaml
    from argparse import Namespace
    from functools import partial
    margs = Namespace(**yaml.full_load(open("/root/reference/weights/diffphore_calibrated_warmuped_ft/model_parameters.yml")))
    t_to_sigma = partial(ref.t_to_sigma_compl, args=margs)
    model = ref.get_model(margs, torch.device("cpu"), t_to_sigma=t_to_sigma, no_parallel=True)
    model.eval()
    from diffphore_amd import weights
    model.load_state_dict(weights.init_state_dict(), strict=True)

    def fake_fitscore(args, ligand_pos, name, mol, phore_file=None, store_ranked_pose=True):
        os.makedirs(os.path.join(args.run_dir, f"mapping_process/{name}"), exist_ok=True)     # calculate_fitscore does (sampling.py:462-463)
        return [float(-np.linalg.norm(p.mean(0) - np.array([10.0, -3.0, 2.0]))) for p in ligand_pos]
    ref.calculate_fitscore = fake_fitscore
    gs = synthetic.synthetic_pairs(313, 3, 14, 3, 12)
    for i, g in enumerate(gs):
        g.name = f"tgt__lig{i}"
        g.original_center = torch.tensor([[10.0 + i, -3.0, 2.0]])
        g.mol = "placeholder-mol"
        g.phore_file = "none.phore"
    with tempfile.TemporaryDirectory() as td:
        args = Namespace(**{**vars(margs), "run_dir": td, "sample_per_complex": 3, "inference_steps": 3, "overwrite": True, "save_single": False,
                            "min_similarity": 0.0, "no_random": False, "ode": False, "batch_size": 20, "no_final_step_noise": True})
        np.random.seed(13)
        torch.manual_seed(13)
        with torch.no_grad():
            metrics = ref.fit(args, model, gs, torch.device("cpu"), t_to_sigma)
        logs = [json.load(open(os.path.join(td, f"mapping_process/{n}/{n}_dock.log"))) for n in metrics["name"]]
    out["fit_names"] = np.asarray(metrics["name"])
    out["fit_scores"] = np.asarray(metrics["fitscore"], dtype=np.float64)
    out["fit_log_scores"] = np.asarray([l["fitscore"] for l in logs], dtype=np.float64)
    print("fit:", metrics["name"], out["fit_scores"])
    # ---- the validation-time caller: utils/training.py::inference_epoch (RMSD metrics), same adapters -----------------------
    gs = synthetic.synthetic_pairs(314, 3, 14, 3, 12)
    for i, g in enumerate(gs):
        g.name = f"val__lig{i}"
        g.original_center = torch.tensor([[1.0 + i, 2.0, -3.0]])
        g["ligand"].orig_pos = g["ligand"].pos.numpy() + g.original_center.numpy()
    eargs = Namespace(**{**vars(margs), "dataset": "pdbbind", "sample_per_complex": 3, "inference_steps": 3, "keep_update": False,
                         "fitscore": False})
    np.random.seed(17)
    torch.manual_seed(17)
    with torch.no_grad():
        em = ref.inference_epoch(model, gs, torch.device("cpu"), t_to_sigma, eargs)
    out["epoch_rmsd"] = np.asarray(em["rmsd"], dtype=np.float64)
    out["epoch_lt2"], out["epoch_lt5"] = np.float64(em["rmsds_lt2"]), np.float64(em["rmsds_lt5"])
    print("inference_epoch rmsd:", out["epoch_rmsd"], em["rmsds_lt2"], em["rmsds_lt5"], sorted(em))
    np.savez_compressed(os.path.join(HERE, "ref_fit_helpers.npz"), **out)


if __name__ == "__main__":
    main()
