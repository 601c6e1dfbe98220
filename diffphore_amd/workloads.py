"""Per-rank synthetic workloads for the five BASELINE.json configs (SURVEY.md 8d "Configs -> shapes").

`config` here is BASELINE.json's configs index + 1 (the numbering of `synthetic.CONFIGS`):

    1  configs[0]  one complex of the shipped example's shape (20 atoms, 5 torsions, 79 phore nodes) x 40 samples
    2  configs[1]  512 distinct ligand-pharmacophore pairs (30 atoms, 8 torsions, 32 phore nodes)          [bench default]
    3  configs[2]  256 pairs of large ligands (80 atoms, 25 torsions, 32 phore nodes)
    4  configs[3]  virtual screen: ONE 79-node pharmacophore x many ligands (N_l ~ U{15..45}, T ~ U{2..12}) x 40 samples;
                   the ligand list is cost-sharded over the ranks (LPT, `distributed.shard_indices`), the phore replicated.
                   This is the reference's `[deepcopy(graph) x N]` (src/inference.py:183-185) over a ligand library.
    5  configs[4]  target fishing: many pharmacophores (N_p ~ U{12..80}) x ONE ligand (30 atoms, 8 torsions) x 40 samples,
                   100-step schedule; the phore list is sharded, the ligand replicated.

Every rank builds the SAME global unit list from the same seed and keeps its own shard, so no data crosses ranks.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import synthetic
from .distributed import pose_cost, shard_indices

DEFAULT_UNITS = {1: 512, 2: 512, 3: 256, 4: 16, 5: 16}        # per GPU: poses (1-3) or complexes (4, 5; x 40 samples each)
SAMPLES = 40                                                   # inference.py --sample_per_complex default of the examples


def describe(config: int, units: int) -> str:
    s = synthetic.CONFIGS[config]
    if config in (1, 2, 3):
        return (f"BASELINE.json configs[{config - 1}]: {units} synthetic lig-phore pairs per GPU ({s['n_atoms']} atoms, "
                f"{s['n_tor']} torsions, {s['n_phore']} phore nodes)")
    if config == 4:
        return (f"BASELINE.json configs[3] (virtual screen, subsample): 1 pharmacophore (79 nodes) x {units} ligands per GPU "
                f"(N_l~U{{15..45}}, T~U{{2..12}}) x {SAMPLES} samples = {units * SAMPLES} poses per GPU, ligands LPT-sharded over ranks")
    return (f"BASELINE.json configs[4] (target fishing, subsample): {units} pharmacophores per GPU (N_p~U{{12..80}}) x 1 ligand "
            f"(30 atoms, 8 torsions) x {SAMPLES} samples = {units * SAMPLES} poses per GPU, 100-step schedule, phores LPT-sharded over ranks")


def steps_of(config: int) -> int:
    return int(synthetic.CONFIGS[config]["steps"])


def build(config: int, units_per_gpu: int, rank: int = 0, world: int = 1) -> Tuple[List, List[int]]:
    """-> (this rank's pose list, global unit indices it owns).  A pose list entry is one HeteroData sample; samples of one
    complex share every static tensor (`HeteroData.sample_copy`)."""
    s = synthetic.CONFIGS[config]
    if config in (1, 2, 3):
        # independent pairs: every rank draws its own (seed offset by rank), weak scaling by construction
        graphs = synthetic.synthetic_pairs(config, units_per_gpu, s["n_atoms"], s["n_tor"], s["n_phore"],
                                           seed=synthetic.BASE_SEED + config + 1000 * rank)
        return graphs, list(range(rank * units_per_gpu, (rank + 1) * units_per_gpu))
    n_units = units_per_gpu * world
    cplx = synthetic.synthetic_pairs(config, n_units, s["n_atoms"], s["n_tor"], s["n_phore"], seed=synthetic.BASE_SEED + config,
                                     shared_phore=(config == 4), shared_ligand=(config == 5))
    costs = [pose_cost(g) for g in cplx]
    mine = shard_indices(costs, world)[rank]
    # heaviest complexes first: workgroups are dispatched in pose order, so a launch then ends on its cheapest poses instead of on a straggler
    mine = sorted(mine, key=lambda i: (-costs[i], i))
    poses = [cplx[i].sample_copy() for i in mine for _ in range(SAMPLES)]
    return poses, mine


def shard_balance(config: int, units_per_gpu: int, world: int) -> List[int]:
    """Cost per rank of the LPT partition (for reporting)."""
    s = synthetic.CONFIGS[config]
    rng_cost = []
    cplx = synthetic.synthetic_pairs(config, units_per_gpu * world, s["n_atoms"], s["n_tor"], s["n_phore"],
                                     seed=synthetic.BASE_SEED + config, shared_phore=(config == 4), shared_ligand=(config == 5))
    costs = [pose_cost(g) for g in cplx]
    for sh in shard_indices(costs, world):
        rng_cost.append(int(np.sum([costs[i] for i in sh])))
    return rng_cost
