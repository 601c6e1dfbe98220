"""Oracle restatement of the reverse-diffusion sampler (TEST INFRASTRUCTURE ONLY).

  sampling_phore     /root/reference/src/utils/sampling.py:174-280
  set_time_phore     /root/reference/src/utils/diffusion_utils.py:181-207
  t_to_sigma         /root/reference/src/utils/diffusion_utils.py:16-20
  get_t_schedule     /root/reference/src/utils/diffusion_utils.py:135-145

The reference collates with PyG's DataLoader; here `diffphore_amd.hetero.Batch` (a plain container, no compute) plays
that role.  `noise` lets the CPU oracle and the HIP path consume identical pre-drawn normals; with noise=None the
draw order is the reference's (per batch: z_tr, z_rot, z_tor from the global torch CPU generator).
Generalisation over the reference: per-graph torsion counts may differ inside a batch (reference assumes
`len(tor)//b`, sampling.py:246); identical when T is uniform.
"""
from __future__ import annotations

import numpy as np
import torch

from diffphore_amd.hetero import Batch

from .conformer import modify_conformer


def t_to_sigma(t_tr, t_rot, t_tor, args):
    tr_sigma = args.tr_sigma_min ** (1 - t_tr) * args.tr_sigma_max ** t_tr
    rot_sigma = args.rot_sigma_min ** (1 - t_rot) * args.rot_sigma_max ** t_rot
    tor_sigma = args.tor_sigma_min ** (1 - t_tor) * args.tor_sigma_max ** t_tor
    return tr_sigma, rot_sigma, tor_sigma


def get_t_schedule(inference_steps):
    return np.linspace(1, 0, inference_steps + 1)[:-1]


def set_time_phore(graphs, t_tr, t_rot, t_tor, batchsize):
    for nt in ("ligand", "phore"):
        n = graphs[nt].num_nodes
        graphs[nt].node_t = {"tr": t_tr * torch.ones(n), "rot": t_rot * torch.ones(n), "tor": t_tor * torch.ones(n)}
    graphs.complex_t = {"tr": t_tr * torch.ones(batchsize), "rot": t_rot * torch.ones(batchsize),
                        "tor": t_tor * torch.ones(batchsize)}


def sampling_phore(data_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, t_to_sigma_fn, model_args,
                   no_random=False, ode=False, batch_size=20, no_final_step_noise=False, noise=None, trace=None, pos_trace=None, margin_trace=None):
    N = len(data_list)
    for t_idx in range(inference_steps):
        t_tr, t_rot, t_tor = tr_schedule[t_idx], rot_schedule[t_idx], tor_schedule[t_idx]
        last = t_idx == inference_steps - 1
        dt_tr = tr_schedule[t_idx] - tr_schedule[t_idx + 1] if not last else tr_schedule[t_idx]
        dt_rot = rot_schedule[t_idx] - rot_schedule[t_idx + 1] if not last else rot_schedule[t_idx]
        dt_tor = tor_schedule[t_idx] - tor_schedule[t_idx + 1] if not last else tor_schedule[t_idx]
        new_data_list = []
        g0 = 0
        tor0 = 0
        for lo in range(0, N, batch_size):
            graphs = data_list[lo:lo + batch_size]
            batch = Batch.from_data_list(graphs)
            b = batch.num_graphs
            tr_sigma, rot_sigma, tor_sigma = t_to_sigma_fn(t_tr, t_rot, t_tor)
            set_time_phore(batch, t_tr, t_rot, t_tor, b)
            with torch.no_grad():
                tr_score, rot_score, tor_score = model(batch)
            tr_score, rot_score, tor_score = tr_score.float(), rot_score.float(), tor_score.float()
            if trace is not None:
                trace.append((tr_score.clone(), rot_score.clone(), tor_score.clone()))
            if getattr(model, "track_margins", False) and margin_trace is not None:
                margin_trace.append((t_idx, lo, {k: v.clone() for k, v in model.margins.items()}))
            tr_g = tr_sigma * torch.sqrt(torch.tensor(2 * np.log(model_args.tr_sigma_max / model_args.tr_sigma_min)))
            rot_g = 2 * rot_sigma * torch.sqrt(torch.tensor(np.log(model_args.rot_sigma_max / model_args.rot_sigma_min)))
            zero = no_random or (no_final_step_noise and last)
            nT = tor_score.shape[0]
            if ode:
                tr_perturb = 0.5 * tr_g ** 2 * dt_tr * tr_score
                rot_perturb = 0.5 * rot_score * dt_rot * rot_g ** 2
            else:
                if zero:
                    tr_z, rot_z = torch.zeros((b, 3)), torch.zeros((b, 3))
                elif noise is not None:
                    tr_z, rot_z = noise["tr"][t_idx, g0:g0 + b], noise["rot"][t_idx, g0:g0 + b]
                else:
                    tr_z = torch.normal(mean=0, std=1, size=(b, 3))
                    rot_z = torch.normal(mean=0, std=1, size=(b, 3))
                tr_perturb = tr_g ** 2 * dt_tr * tr_score + tr_g * np.sqrt(dt_tr) * tr_z
                rot_perturb = rot_score * dt_rot * rot_g ** 2 + rot_g * np.sqrt(dt_rot) * rot_z
            tor_perturb = None
            if not model_args.no_torsion:
                tor_g = tor_sigma * torch.sqrt(torch.tensor(2 * np.log(model_args.tor_sigma_max / model_args.tor_sigma_min)))
                if ode:
                    tor_perturb = (0.5 * tor_g ** 2 * dt_tor * tor_score).numpy()
                else:
                    if zero:
                        tor_z = torch.zeros(tor_score.shape)
                    elif noise is not None:
                        tor_z = noise["tor"][t_idx, tor0:tor0 + nT]
                    else:
                        tor_z = torch.normal(mean=0, std=1, size=tor_score.shape)
                    tor_perturb = (tor_g ** 2 * dt_tor * tor_score + tor_g * np.sqrt(dt_tor) * tor_z).numpy()
            o = 0
            for i, g in enumerate(batch.to_data_list()):
                T = int(g["ligand"].edge_mask.sum())
                tp = tor_perturb[o:o + T] if tor_perturb is not None else None
                o += T
                new_data_list.append(modify_conformer(g, tr_perturb[i:i + 1], rot_perturb[i:i + 1].squeeze(0),
                                                      tp if (tp is not None and T > 0) else None))
            g0 += b
            tor0 += nT
        data_list = new_data_list
        if pos_trace is not None:          # per-step coordinates (diagnostics: tools/diag_traj2.py)
            pos_trace.append([g["ligand"].pos.clone() for g in data_list])
    return data_list, None
