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
