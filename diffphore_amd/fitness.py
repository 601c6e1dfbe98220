"""On-device pharmacophore-fit surrogate for ranking poses - `dp_fit_overlap` behind the reference's scoring hooks.

This is NOT AncPhore (`calculate_fitscore`, src/utils/sampling.py:447-498 shells out to a closed binary): it is a Gaussian-volume
overlap with the constants the reference itself quotes in `get_perfect_similarity` (src/inference.py:273-295).  Use it where a
ranking signal is needed and AncPhore is unavailable; scores are not comparable with published DiffPhore fit scores.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .engine import PackedBatch, _declare, _stream


def overlap_scores(pb: PackedBatch, pos: Optional[torch.Tensor] = None, ex_lambda: float = 1.0) -> torch.Tensor:
    """Scores [B] of the packed batch's current state, or of candidate coordinates `pos` [sum N_l, 3] on the batch's device."""
    L = _declare()
    L.dp_fit_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    L.dp_fit_overlap.restype = C.c_int
    out = torch.empty(pb.B, dtype=torch.float32, device=pb.device)
    if pos is not None:
        pos = pos.to(pb.device, torch.float32).contiguous()
        if pos.shape != pb.pos.shape:
            raise ValueError(f"candidate coordinates {tuple(pos.shape)} do not match the batch {tuple(pb.pos.shape)}")
    _lib.check(L.dp_fit_overlap(C.byref(pb.struct), _lib.ptr(pos), C.c_float(ex_lambda), out.data_ptr(), _stream()), "dp_fit_overlap")
    return out


def overlap_scorer(args, dock_pose: np.ndarray, name: str, graph, device="cuda"):
    """`scorer` for `diffphore_amd.fit.fit`: scores the N docked poses of one complex (original frame, heavy atoms)."""
    import copy
    center = torch.as_tensor(graph.original_center).reshape(1, 3).numpy() if "original_center" in graph else np.zeros((1, 3))
    heavy = torch.not_equal(torch.as_tensor(graph["ligand"].x)[:, 0], 0).numpy()
    if not heavy.all():
        raise NotImplementedError("graphs with explicit hydrogens: score the heavy-atom graph")
    gs = []
    for p in dock_pose:
        g = copy.deepcopy(graph)
        g["ligand"].pos = torch.as_tensor(p - center, dtype=torch.float32)
        gs.append(g)
    return overlap_scores(PackedBatch(gs, device), ex_lambda=float(getattr(args, "ex_lambda", 1.0))).cpu().tolist()


def get_perfect_similarity(g, weights=(1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0), alpha=(1.0, 1.0, 0.7, 1.0, 1.0, 0.7, 1.0, 1.0, 0.7, 1.0, 0.837)):
    """`src/inference.py:273-311`: type/count-only fingerprint similarity behind the `--min_similarity` filter of `fit`.
    `g['ligand'].ph` = per-type feature counts of the ligand (written by the reference's featuriser)."""
    phore_volume = torch.as_tensor(g["phore"].phoretype).float().sum(dim=0)
    phore_overlap = torch.min(torch.as_tensor(g["ligand"].ph).float().reshape(-1), phore_volume)
    coeff = torch.tensor(weights).float()
    if alpha is not None:
        coeff = coeff * (7.999999999 * (torch.tensor(alpha) * torch.pi / 2) ** 1.5)
    weighted_volume = (phore_volume * coeff).sum()
    if weighted_volume == 0:
        return -1.0
    return ((phore_overlap * coeff).sum() / weighted_volume).item()
