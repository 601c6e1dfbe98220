#!/usr/bin/env python3
"""Virtual screening with the streaming driver on a lazily generated library (no dataset, no checkpoint, no RDKit needed):

    python examples/screen_synthetic.py --ligands 200 --samples 40 --seed 7 --out /tmp/diffphore_screen
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/screen_synthetic.py --ligands 20000 ...

One pharmacophore x many ligands x N samples each (BASELINE configs[3]; the reference's `for graph in loader: [deepcopy(graph) x N] ->
randomize_position -> sampling_phore`, src/inference.py:139-228).  `distributed.screen` shards the ligand list over the ranks by a cost
estimate, builds each rank's ligands on demand while the GPU samples the previous device batch, draws start randomisation and noise on
the device from (`seed`, pose uid) - 1, 2 or 8 ranks produce the same poses - and hands every finished ligand to a sink.  Here the sink
ranks the samples by the on-device pharmacophore-overlap surrogate (not AncPhore) and appends `ligand, best score, centroid` to a csv.
With real inputs `load_unit(i)` is the featuriser (SMILES -> conformer -> HeteroData) and `costs[i]` its heavy-atom count times the
number of pharmacophore nodes."""
import argparse
import os
import time
import types

import numpy as np
import torch

from diffphore_amd import config, synthetic
from diffphore_amd import distributed as D
from diffphore_amd.diffusion import t_to_sigma_fn
from diffphore_amd.engine import PackedBatch
from diffphore_amd.fitness import overlap_scores
from diffphore_amd.score_model import get_model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ligands", type=int, default=64)
    ap.add_argument("--samples", type=int, default=40)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--seed", type=int, default=20250321)
    ap.add_argument("--max-poses", type=int, default=2560, help="poses per device batch")
    ap.add_argument("--out", default="/tmp/diffphore_screen")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if torch.cuda.device_count() >= world else "gloo", device_id=dev if torch.cuda.device_count() >= world else None)
    cfg = config.ModelConfig()
    args = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False, inference_steps=a.steps,
                                 # the released model_parameters.yml (the configuration the kernels are specialised for)
                                 max_radius=5.0, scale_by_sigma=True, boarder=True, use_phore_match_feat=True, atom_weight="phore", scaler=100.0)
    model = get_model(args, dev, t_to_sigma_fn(args), no_parallel=True).init_random()     # with a checkpoint: .load_state_dict(torch.load(path))
    if world > 1:
        D.broadcast_weights(model.device_model.blob, src=0)  # the job's only collective
    lib = synthetic.ScreenLibrary(4, a.ligands)              # unit i = a pure function of (seed, i): nothing is built until asked for
    costs = [lib.cost(i) for i in range(len(lib))]           # sizes only
    os.makedirs(a.out, exist_ok=True)
    kept = {}                                                # the graphs of the units in flight (the scorer needs the pharmacophore)

    def load(i):
        kept[i] = lib.unit(i)
        return kept[i]

    rows = []

    def sink(i, poses):                                      # poses: float32 [samples, n_atoms, 3], pharmacophore-centred frame
        g = kept.pop(i)
        cand = []
        for p in poses:                                      # one packed batch of the ligand's samples, scored in one launch
            c = g.sample_copy()
            c["ligand"].pos = torch.from_numpy(p)
            cand.append(c)
        s = overlap_scores(PackedBatch(cand, dev)).cpu().numpy().astype(np.float64)
        best = int(np.argmax(s))
        rows.append((i, poses.shape[1], float(s[best]), *(poses[best].mean(0) + g.original_center.numpy().reshape(3)).tolist()))

    t0 = time.time()
    mine, _ = D.screen(len(lib), load, model, args, costs=costs, samples_per_complex=a.samples, max_poses=a.max_poses, seed=a.seed, on_result=sink)
    torch.cuda.synchronize()
    dt = time.time() - t0
    path = os.path.join(a.out, f"screen_rank{rank}.csv")
    with open(path, "w") as f:
        f.write("ligand,atoms,best_overlap,cx,cy,cz\n")
        for r in sorted(rows):
            f.write("%d,%d,%.5f,%.4f,%.4f,%.4f\n" % r)
    n_pose = len(mine) * a.samples
    print(f"rank {rank}/{world}: {len(mine)} ligands x {a.samples} samples x {a.steps} steps in {dt:.2f} s = {n_pose * a.steps / dt:,.0f} pose-steps/s "
          f"({len(mine) / dt:.1f} ligands/s) -> {path}")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
