"""Multi-GPU plumbing: independent poses sharded over ranks, one weight broadcast, optional result gather.

The reference has no distributed code; poses are independent through all reverse-diffusion steps
(sampling.py:213-255, BatchNorm in eval mode), so there is no data-path collective.  One process per GPU,
`torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm); "gloo" works for the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def pose_cost(g) -> int:
    """Work estimate of one pose per step: dense cross edges dominate (4 of 6 conv families)."""
    nl, npn = int(g["ligand"].pos.shape[0]), int(g["phore"].pos.shape[0])
    return nl * npn + 25 * nl + 8 * npn


def shard_indices(costs: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first partition of pose indices into `world` shards (deterministic)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


def shard_poses(graphs: Sequence, rank: int, world: int) -> Tuple[List[int], List]:
    idx = shard_indices([pose_cost(g) for g in graphs], world)[rank]
    return idx, [graphs[i] for i in idx]


def broadcast_weights(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """The job's only collective: rank `src`'s packed parameter blob to every rank (in place)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def gather_positions(local_idx: Sequence[int], local_pos: Sequence[torch.Tensor], n_total: int, dst: int = 0):
    """Collect final ligand coordinates on rank `dst` in the original pose order (CPU tensors; tiny: 12 B per atom)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = [None] * n_total
        for i, p in zip(local_idx, local_pos):
            out[i] = p
        return out
    payload = (list(local_idx), [p.cpu() for p in local_pos])
    gathered = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    out = [None] * n_total
    for idx, pos in gathered:
        for i, p in zip(idx, pos):
            out[i] = p
    return out


def fit_complexes(graphs: Sequence, model, args, samples_per_complex: int = 40, device="cuda", batch_size: int = 20,
                  max_poses: int = 8192):
    """The sampler-facing part of `inference.py::fit` (:139-228) on this rank's share of `graphs`:

        data_list = [deepcopy(graph) x N] ; randomize_position(...) ; sampling_phore(...) ; collect ligand_pos

    for every complex, but with ALL complexes x samples of the rank batched into device-resident launches instead of one
    40-pose call per complex.  Returns (indices of the complexes this rank owns, list of [N, n_atoms, 3] arrays in that order).
    Scoring / SDF writing (AncPhore, RDKit) stay with the caller.  `args` needs the reference's fields: inference_steps,
    tr/rot/tor_sigma_min/max, no_torsion, no_random, ode, no_final_step_noise (missing ones default like inference.py:54-96).
    """
    import copy

    import torch.distributed as dist

    from .diffusion import get_t_schedule, t_to_sigma_fn
    from .sampling import sampling_phore
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    idx, mine = shard_poses(graphs, rank, world)
    steps = int(getattr(args, "inference_steps", 20))
    sched = get_t_schedule(steps)
    out = []
    # One fused call: every device batch (whole complexes, <= max_poses poses) is packed once, randomised and sampled on the device,
    # and the host stage of the next batch overlaps the device loop of the current one (sampling_phore's pipeline).
    per = max(1, max_poses // samples_per_complex)
    data_list = [g.sample_copy() if hasattr(g, "sample_copy") else copy.deepcopy(g) for g in mine for _ in range(samples_per_complex)]
    sampling_phore(data_list, model, steps, sched, sched, sched, device, t_to_sigma_fn(args), args,
                   no_random=getattr(args, "no_random", False), ode=getattr(args, "ode", False), batch_size=batch_size,
                   no_final_step_noise=getattr(args, "no_final_step_noise", False), max_poses=per * samples_per_complex,
                   randomize=dict(no_torsion=getattr(args, "no_torsion", False), no_random=False, tr_sigma_max=args.tr_sigma_max))
    for c in range(len(mine)):
        poses = data_list[c * samples_per_complex:(c + 1) * samples_per_complex]
        out.append(torch.stack([p["ligand"].pos for p in poses]).numpy())
    return idx, out


def screen_plan(n_units: int, costs, rank: int, world: int, samples_per_complex: int = 40, max_poses: int = 8192) -> List[List[int]]:
    """Which units rank `rank` of `world` processes, cut into device batches (lists of unit indices, in processing order).  Pure host
    arithmetic on the cost list: every rank computes its own plan without building or exchanging anything.  With `costs` the shard is
    the LPT partition (`shard_indices`) walked heaviest first - similar sizes share a device batch (its launch geometry follows the
    largest pose), and inside a launch the workgroups, dispatched in pose order, end on the cheapest poses instead of on a straggler;
    without, units are dealt round-robin.  A batch holds whole complexes, at most `max_poses` poses (at least one complex)."""
    if costs is not None:
        if len(costs) != n_units:
            raise ValueError("costs needs one entry per unit")
        mine = sorted(shard_indices(costs, world)[rank], key=lambda i: (-costs[i], i))
    else:
        mine = list(range(rank, n_units, world))
    if int(n_units) * samples_per_complex >= 2 ** 31:
        raise ValueError("pose uids are 32-bit: n_units * samples_per_complex must stay below 2^31")
    per = max(1, max_poses // samples_per_complex)
    return [mine[lo:lo + per] for lo in range(0, len(mine), per)]


def screen(n_units: int, load_unit, model, args, *, costs: Sequence[int] = None, samples_per_complex: int = 40, device="cuda",
           max_poses: int = 8192, seed: int = 0, on_result=None):
    """Streaming form of `fit_complexes` for library-scale screening (BASELINE configs[3] / configs[4]: 10^4-10^5 complexes x 40 samples
    per GPU).  `load_unit(i)` builds complex i on demand (featurisation, parsing, ... stay with the caller); `costs[i]` (optional, cheap:
    e.g. atoms x pharmacophore nodes) drives the LPT shard, otherwise units are dealt round-robin.  This rank

        * takes ITS indices only (no rank ever builds or holds another rank's units),
        * walks them in device batches of <= max_poses poses (whole complexes), building batch k+1 - `load_unit` calls included - on the
          host while the GPU samples batch k (`sampling_phore_stream`),
        * draws start randomisation and step noise on the device from (`seed`, pose uid = unit * samples + sample): results do not
          depend on the world size, the batch size or the order of the stream,
        * hands each finished complex to `on_result(i, poses [samples, n_atoms, 3] float32)` and drops it; with on_result=None the arrays
          are collected and returned (only sensible for small runs).

    Returns (indices owned by this rank in processing order - heaviest first when `costs` is given -, list of pose arrays or None).  Peak host memory: two device batches of
    pose objects, whatever the library size."""
    import torch.distributed as dist

    from .diffusion import get_t_schedule, t_to_sigma_fn
    from .sampling import sampling_phore_stream
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    batches = screen_plan(n_units, costs, rank, world, samples_per_complex, max_poses)
    mine = [i for bt in batches for i in bt]
    steps = int(getattr(args, "inference_steps", 20))
    sched = get_t_schedule(steps)

    def parts():
        for bt in batches:
            poses, uids = [], []
            for i in bt:
                g = load_unit(i)
                for s_ in range(samples_per_complex):
                    poses.append(g.sample_copy() if hasattr(g, "sample_copy") else __import__("copy").deepcopy(g))
                    uids.append(i * samples_per_complex + s_)
            yield poses, uids

    collected = [] if on_result is None else None
    done = 0
    stream = sampling_phore_stream(parts(), model, steps, sched, sched, sched, device, t_to_sigma_fn(args), args, seed=seed,
                                   no_random=getattr(args, "no_random", False), ode=getattr(args, "ode", False),
                                   no_final_step_noise=getattr(args, "no_final_step_noise", False),
                                   randomize=dict(no_torsion=getattr(args, "no_torsion", False), no_random=False, tr_sigma_max=args.tr_sigma_max))
    for part in stream:
        for c in range(len(part) // samples_per_complex):
            arr = torch.stack([p["ligand"].pos for p in part[c * samples_per_complex:(c + 1) * samples_per_complex]]).numpy()
            i = mine[done]
            done += 1
            if on_result is not None:
                on_result(i, arr)
            else:
                collected.append(arr)
    return mine, collected
