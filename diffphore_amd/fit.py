"""Ligand-pharmacophore fitting driver: `src/inference.py::fit` (:139-270) and `analyze_results` (:321-350) on the HIP hot path
(SURVEY.md 8f-1).

What is kept from the reference, per complex: resume from `ranked_poses/<name>_ranked.sdf` + `mapping_process/<name>/<name>_dock.log`
unless `args.overwrite`; N = `args.sample_per_complex` copies, `randomize_position`, `sampling_phore`; the SVD retry convention
(an exception whose text contains 'failed to converge' -> up to 5 re-randomised retries, anything else -> skip the complex);
heavy-atom filter `x[:,0] != 0`; docked pose = pos + original_center; fit scores; the ranked SDF (best score first, `fitscore`
data item, titles `<name>_rank_<i>`); the per-complex json log; the `tmp_log` progress dump; the returned metrics dict.

What differs:
  * Fit scores.  The reference shells out to the closed AncPhore binary (`calculate_fitscore`, `src/utils/sampling.py:447-498`),
    which is not available.  `scorer(args, dock_pose [N,n,3], name, graph) -> list[float] | None` is a hook; without one the
    scores are the reference's own failure value (-2.0 per pose, `inference.py:233-235`) and poses keep sampling order.
  * RDKit.  `graph.mol` is a `diffphore_amd.sdf_io.Mol` (or absent: then no SDF is written).
  * Throughput.  Complexes whose results are not on disk are sampled `args.complexes_per_launch` (default 64) at a time in one
    device-resident batch instead of one 40-pose call each; a batch that trips the SVD flag is redone complex by complex so the
    retry bookkeeping stays per complex.
"""
from __future__ import annotations

import copy
import json
import os
import time
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from . import sdf_io
from .diffusion import get_t_schedule
from .sampling import randomize_position, sampling_phore

Scorer = Callable[[object, np.ndarray, str, object], Optional[Sequence[float]]]


def _name(g) -> str:
    n = g.name
    return n[0] if isinstance(n, (list, tuple)) else n


def _first(v):
    return v[0] if isinstance(v, (list, tuple)) else v


def _paths(args, name):
    return (os.path.join(args.run_dir, f"ranked_poses/{name}_ranked.sdf"),
            os.path.join(args.run_dir, f"mapping_process/{name}/{name}_dock.log"))


def store_ranked_poses(args, dock_pose: np.ndarray, name: str, mol: Optional[sdf_io.Mol], scores) -> None:
    """The `store_ranked_pose=True` tail of `calculate_fitscore` (`src/utils/sampling.py:480-496`)."""
    if mol is None or scores is None:
        return
    d = os.path.join(args.run_dir, "ranked_poses/")
    os.makedirs(d, exist_ok=True)
    perm = np.argsort(np.array(scores))[::-1]
    sdf_io.write_mol_with_multi_coords(mol, dock_pose[perm], os.path.join(d, f"{name}_ranked.sdf"), name, marker="rank",
                                       properties={"fitscore": np.array(scores)[perm]})


def _pose_copy(g):
    """One more sample of a complex (inference.py:183 deep-copies the graph; only the ligand state is ever written)."""
    return g.sample_copy() if hasattr(g, "sample_copy") else copy.deepcopy(g)


def _sample(args, model, graphs, N, device, t_to_sigma, sched):
    data_list = [_pose_copy(g) for g in graphs for _ in range(N)]
    rz = dict(no_torsion=args.no_torsion, no_random=False, tr_sigma_max=args.tr_sigma_max, keep_update=getattr(args, "keep_traj", False))
    initial = None
    t0 = time.time()
    if getattr(args, "keep_update", False):        # the caller wants the randomised start poses back: two calls, like the reference
        randomize_position(data_list, rz["no_torsion"], False, rz["tr_sigma_max"], keep_update=rz["keep_update"], device=device)
        initial = [g["ligand"].pos.cpu().numpy() for g in data_list]
        rz = None
        t0 = time.time()
    # otherwise randomisation is fused into the sampler call: packed once, randomised and sampled on the device (same RNG order)
    sampling_phore(data_list=data_list, model=model, inference_steps=args.inference_steps, tr_schedule=sched, rot_schedule=sched,
                   tor_schedule=sched, device=device, t_to_sigma=t_to_sigma, model_args=args, no_random=getattr(args, "no_random", False),
                   ode=getattr(args, "ode", False), batch_size=getattr(args, "batch_size", 20),
                   no_final_step_noise=getattr(args, "no_final_step_noise", False), max_poses=max(8192, len(data_list)), randomize=rz)
    return data_list, initial, time.time() - t0


def fit(args, model, complex_graphs, device, t_to_sigma, tmp_log: str = "", n_report: int = 1000, scorer: Optional[Scorer] = None):
    sched = get_t_schedule(inference_steps=args.inference_steps)
    N = getattr(args, "sample_per_complex", 1)
    per_launch = max(1, int(getattr(args, "complexes_per_launch", 64)))
    keep = getattr(args, "keep_update", False)
    fitscore, run_times, names, initial_poses, dock_poses = [], [], [], [], []
    results = {}                       # index -> (name, scores, run_time, initial, docked) in input order
    std_time = time.time()

    todo = []
    for i, g in enumerate(complex_graphs):
        name = _name(g)
        if getattr(args, "min_similarity", 0) > 0:          # inference.py:170-178
            try:
                from .fitness import get_perfect_similarity
                max_sim = get_perfect_similarity(g)
                if max_sim < args.min_similarity:
                    print(f"[I] `{name}` is excluded due to pharmacophore fingerprint similarity constraints ({max_sim:.2f} < {args.min_similarity:.2f}).")
                    continue
            except Exception as e:
                print(f"[W] Failed to calculate the pharmacophore fingerprint similarity of `{name}`, skipped. {e}")
        docked_file, log_file = _paths(args, name)
        if os.path.exists(docked_file) and os.path.exists(log_file) and not getattr(args, "overwrite", False):
            log = json.load(open(log_file))
            results[i] = (log["name"], log["fitscore"] if log["fitscore"] is not None else [-2.0] * N, log["run_time"], None, None)
        elif g["ligand"].pos.shape[0] == 0:
            print(f"[W] Graph {[name]} with 0 atoms, skipped")
        else:
            todo.append(i)

    def finish(i, poses, initial, run_time):
        g = complex_graphs[i]
        name = _name(g)
        filter_hs = torch.not_equal(g["ligand"].x[:, 0], 0).cpu().numpy()
        ligand_pos = np.asarray([p["ligand"].pos.cpu().numpy()[filter_hs] for p in poses])
        center = torch.as_tensor(g.original_center).reshape(1, 3).cpu().numpy() if "original_center" in g else np.zeros((1, 3))
        dock_pose = ligand_pos + center
        mol = _first(g.mol) if "mol" in g else None
        if mol is not None:
            mol = sdf_io.remove_hs(mol)
        scores = None
        if scorer is not None:
            scores = scorer(args, dock_pose, name, g)
        scored = scores is not None and len(scores) > 0
        if scored:
            scores = [float(s) for s in scores]
            # the reference writes the ranked SDF inside calculate_fitscore, i.e. only when scoring succeeded (sampling.py:447-498):
            # without it a later run finds no `*_ranked.sdf`, does not resume from the log and scores the complex again
            store_ranked_poses(args, dock_pose, name, mol, scores)
        else:
            print(f"[W] fitscore calculated with error and set as -2.0 for `{name}`")
        _, log_file = _paths(args, name)
        os.makedirs(os.path.dirname(log_file), exist_ok=True)
        json.dump({"name": name, "fitscore": scores if scored else None, "run_time": run_time}, open(log_file, "w"), indent=4)   # inference.py:241-245
        results[i] = (name, scores if scored else [-2.0] * N, run_time, initial, [p.docked_poses for p in poses] if keep else None)

    def finish_safe(i, poses, initial, run_time):
        """Scoring / writing problems concern one complex only: never a reason to sample anything again."""
        try:
            finish(i, poses, initial, run_time)
        except Exception as e:      # noqa: BLE001
            print(f"[W] Error occured when fitting {_name(complex_graphs[i])} to the reference pharamcophore, skipped. {e}")

    done = 0
    for lo in range(0, len(todo), per_launch):
        group = todo[lo:lo + per_launch]
        sampled = None
        try:
            sampled = _sample(args, model, [complex_graphs[i] for i in group], N, device, t_to_sigma, sched)
        except Exception as e:                                   # only SAMPLER errors lead to re-sampling (inference.py:196-218)
            if "failed to converge" not in str(e):
                print(f"[W] Error occured when fitting a batch of {len(group)} complexes, retrying one by one. {e}")
        if sampled is not None:
            poses, initial, dt = sampled
            for c, i in enumerate(group):
                finish_safe(i, poses[c * N:(c + 1) * N], initial[c * N:(c + 1) * N] if initial else None, dt / len(group))
        else:
            for i in group:                                      # per-complex retry bookkeeping of the reference
                if i in results:
                    continue
                name = _name(complex_graphs[i])
                failed = 0
                while True:
                    try:
                        poses, initial, dt = _sample(args, model, [complex_graphs[i]], N, device, t_to_sigma, sched)
                    except Exception as e2:
                        if "failed to converge" in str(e2):
                            failed += 1
                            if failed > 5:
                                print("[W] SVD failed to converge 5 times - skipping the complex")
                                break
                            print("[W] SVD failed to converge - trying again with a new sample")
                            continue
                        print(f"[W] Error occured when fitting {name} to the reference pharamcophore, skipped. {e2}")
                        break
                    finish_safe(i, poses, initial, dt)
                    break
        done += len(group)
        if tmp_log != "" and (done // n_report) > ((done - len(group)) // n_report):
            print(f"[I] {done}/{len(todo)} processed...")
            ordered = [results[k] for k in sorted(results)]
            json.dump({"name": [r[0] for r in ordered], "fitscore": [r[1] for r in ordered], "run_time": [r[2] for r in ordered],
                       "batch": done - 1, "total_time": time.time() - std_time}, open(tmp_log, "w"), indent=4)

    for k in sorted(results):
        n, s, rt, ini, dk = results[k]
        names.append(n)
        fitscore.append(s)
        run_times.append(rt)
        if keep and ini is not None:
            initial_poses.append(ini)
            dock_poses.append(dk)
    metrics = {"name": names, "fitscore": fitscore, "run_time": run_times}
    if keep:
        metrics.update(initial_poses=initial_poses, dock_poses=dock_poses)
    return metrics


def my_max(x):
    return max(x) if len(x) else -2.0


def analyze_results(args, results):
    """`ranked_results.csv` exactly as `src/inference.py:321-350` lays it out (tab-separated, sorted by max then top-5 mean)."""
    import pandas as pd
    df = pd.DataFrame({k: results[k] for k in ("name", "fitscore", "run_time")})
    df["max_fitscore"] = df["fitscore"].map(my_max)
    df["top5_mean_fitscore"] = df["fitscore"].map(lambda x: np.sort(x)[-5:].mean().item())
    df["target"] = df["name"].map(lambda x: x.split("__")[0])
    df["ligand"] = df["name"].map(lambda x: x.split("__")[1] if "__" in x else x)
    df = df.sort_values(by=["max_fitscore", "top5_mean_fitscore"], ascending=False)
    os.makedirs(args.out_dir, exist_ok=True)
    dump_file = os.path.join(args.out_dir, "ranked_results.csv")
    df = df[["target", "ligand", "name", "run_time", "max_fitscore", "top5_mean_fitscore", "fitscore"]]
    df.to_csv(dump_file, sep="\t", index=False)
    cutoff = getattr(args, "cutoff", None)
    if cutoff is not None:
        df[df["max_fitscore"] >= cutoff].to_csv(os.path.join(args.out_dir, f"ranked_results_gt{cutoff}.csv"), sep="\t", index=False)
    return df


def inference_epoch(model, complex_graphs, device, t_to_sigma, args, scorer: Optional[Scorer] = None):
    """`src/utils/training.py::inference_epoch` (:370-552) - the validation-time caller of the hot path: N samples per complex,
    randomised start, reverse diffusion, heavy-atom RMSD of every sample to the reference pose (`orig_pos - original_center`,
    no symmetry correction), success rates.  Same metrics dict (`rmsds_lt2`, `rmsds_lt5`, `rmsd`, `fitscore`, `run_time`,
    `failed_indices`, `fitscore_gt0.7/0.4`, and `dock_process` when `args.keep_update`).  Like the reference it calls the sampler
    with its DEFAULT keywords (batch_size 20, noise on every step), and switches to `sampling_phore_with_fitscore` when
    `args.random_samples > 1` (then `scorer`-independent `args.fitscore_fn` must be set: AncPhore is not available).
    Complexes are sampled `args.complexes_per_launch` (default 64) at a time; 1 reproduces the reference's RNG order."""
    from .sampling import sampling_phore_with_fitscore
    sched = get_t_schedule(inference_steps=args.inference_steps)
    N = getattr(args, "sample_per_complex", 1)
    per_launch = max(1, int(getattr(args, "complexes_per_launch", 64)))
    keep = bool(getattr(args, "keep_update", False))
    branching = getattr(args, "random_samples", 0) and args.random_samples > 1
    rmsds, fitscore, run_times, names, failed_indices = [], [], [], [], []
    rec = {k: [] for k in ("fw_tr_update", "fw_tor_update", "fw_rot_update", "rvs_tr_update", "rvs_tor_update", "rvs_rot_update",
                           "origin_pos", "initial_pos", "perturb_pos", "docked_poses", "original_centers")}

    def sample(graphs):
        data_list = [_pose_copy(g) for g in graphs for _ in range(N)]
        if keep or branching:
            randomize_position(data_list, args.no_torsion, False, args.tr_sigma_max, keep_update=keep, device=device)
            pert = [g["ligand"].pos.cpu().numpy() for g in data_list] if keep else None
            if branching:
                sampling_phore_with_fitscore(data_list, model, args.inference_steps, sched, sched, sched, device, t_to_sigma, args,
                                             fitscore_fn=getattr(args, "fitscore_fn", None))
            else:
                sampling_phore(data_list, model, args.inference_steps, sched, sched, sched, device, t_to_sigma, args,
                               max_poses=max(8192, len(data_list)))
            return data_list, pert
        sampling_phore(data_list, model, args.inference_steps, sched, sched, sched, device, t_to_sigma, args, max_poses=max(8192, len(data_list)),
                       randomize=dict(no_torsion=args.no_torsion, no_random=False, tr_sigma_max=args.tr_sigma_max))
        return data_list, None

    def finish(i, g, poses, pert, dt):
        name = _name(g)
        run_times.append(dt)
        center = torch.as_tensor(g.original_center).reshape(1, 3).cpu().numpy()
        if args.no_torsion and "orig_pos" not in g["ligand"]:
            g["ligand"].orig_pos = g["ligand"].pos.cpu().numpy() + center
        orig = g["ligand"].orig_pos
        orig = np.asarray(orig[0] if isinstance(orig, list) else orig)
        filter_hs = torch.not_equal(g["ligand"].x[:, 0], 0).cpu().numpy()
        ligand_pos = np.asarray([p["ligand"].pos.cpu().numpy()[filter_hs] for p in poses])
        rmsds.append(np.sqrt(((ligand_pos - np.expand_dims(orig[filter_hs] - center, 0)) ** 2).sum(axis=2).mean(axis=1)))
        if keep:
            for k in ("fw_tr_update", "fw_tor_update", "fw_rot_update", "rvs_tr_update", "rvs_tor_update", "rvs_rot_update", "docked_poses"):
                rec[k].append([getattr(p, k) if k in p else None for p in poses])
            rec["initial_pos"].append(g["ligand"].pos.cpu().numpy())
            rec["origin_pos"].append(orig)
            rec["perturb_pos"].append(pert)
            rec["original_centers"].append([center for _ in poses])
            names.append(name)
        if getattr(args, "fitscore", False):
            scores = scorer(args, ligand_pos + center, name, g) if scorer is not None else None
            if scores is None or len(scores) == 0:
                fitscore.append([-2.0] * N)
                print(f"[W] Warning fitscore calculated with error and set as -2.0 for `{name}`")
            else:
                fitscore.append([float(s) for s in scores])

    todo = [i for i, g in enumerate(complex_graphs) if g["ligand"].pos.shape[0] > 0]
    for lo in range(0, len(todo), per_launch):
        group = todo[lo:lo + per_launch]
        t0 = time.time()
        try:
            poses, pert = sample([complex_graphs[i] for i in group])
            dt = (time.time() - t0) / len(group)
            for c, i in enumerate(group):
                finish(i, complex_graphs[i], poses[c * N:(c + 1) * N], pert[c * N:(c + 1) * N] if pert else None, dt)
        except Exception as e:
            if "failed to converge" not in str(e):
                raise
            for i in group:                                 # per-complex retry bookkeeping (:445-467)
                failed = 0
                while True:
                    t0 = time.time()
                    try:
                        poses, pert = sample([complex_graphs[i]])
                        finish(i, complex_graphs[i], poses, pert, time.time() - t0)
                        break
                    except Exception as e2:
                        if "failed to converge" not in str(e2):
                            raise
                        failed += 1
                        if failed > 5:
                            print("| WARNING: SVD failed to converge 5 times - skipping the complex")
                            failed_indices.append(i)
                            break
                        print("| WARNING: SVD failed to converge - trying again with a new sample")
    rm = np.array(rmsds).reshape(-1, N) if N != 1 else np.array(rmsds).reshape(-1)
    fs = np.array(fitscore).reshape(-1, N) if N != 1 else np.array(fitscore).reshape(-1)
    metrics = {"rmsds_lt2": (100 * (rm < 2).sum() / max(len(rm), 1) / N), "rmsds_lt5": (100 * (rm < 5).sum() / max(len(rm), 1) / N),
               "rmsd": rm.tolist(), "fitscore": fs.tolist(), "run_time": run_times, "failed_indices": failed_indices}
    if len(fs) > 0:
        metrics["fitscore_gt0.7"] = 100 * (fs > 0.7).sum() / len(fs)
        metrics["fitscore_gt0.4"] = 100 * (fs > 0.4).sum() / len(fs)
    if keep:
        metrics["dock_process"] = dict(name=names, rmsd=rm, **rec)
    return metrics
