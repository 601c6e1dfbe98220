"""Time / sigma plumbing of the sampler - same pure functions as the reference (src/utils/diffusion_utils.py)."""
from __future__ import annotations

from functools import partial

import numpy as np
import torch


def t_to_sigma(t_tr, t_rot, t_tor, args):
    """diffusion_utils.py:16-20"""
    tr_sigma = args.tr_sigma_min ** (1 - t_tr) * args.tr_sigma_max ** t_tr
    rot_sigma = args.rot_sigma_min ** (1 - t_rot) * args.rot_sigma_max ** t_rot
    tor_sigma = args.tor_sigma_min ** (1 - t_tor) * args.tor_sigma_max ** t_tor
    return tr_sigma, rot_sigma, tor_sigma


def t_to_sigma_fn(args):
    """`partial(t_to_sigma_compl, args=args)` as built at inference.py:408."""
    return partial(t_to_sigma, args=args)


def get_t_schedule(inference_steps):
    """diffusion_utils.py:135-145"""
    return np.linspace(1, 0, inference_steps + 1)[:-1]


def set_time_phore(graphs, t_tr, t_rot, t_tor, batchsize, device=None):
    """diffusion_utils.py:181-207 (device kept for signature compatibility; the HIP path uploads times itself)."""
    for nt in ("ligand", "phore"):
        n = graphs[nt].num_nodes
        graphs[nt].node_t = {"tr": t_tr * torch.ones(n), "rot": t_rot * torch.ones(n), "tor": t_tor * torch.ones(n)}
    graphs.complex_t = {"tr": t_tr * torch.ones(batchsize), "rot": t_rot * torch.ones(batchsize),
                        "tor": t_tor * torch.ones(batchsize)}
