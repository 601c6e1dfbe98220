"""Oracle restatement of the conformer update (TEST INFRASTRUCTURE ONLY).  PINNED against tests/golden/ref_utils.npz.

  axis_angle_to_matrix / quaternion route      /root/reference/src/utils/geometry.py:6-85
  rigid_transform_Kabsch_3D_torch              /root/reference/src/utils/geometry.py:88-136
  modify_conformer_torsion_angles              /root/reference/src/utils/torsion.py:64-109
  modify_conformer                             /root/reference/src/utils/diffusion_utils.py:23-79
  randomize_position                           /root/reference/src/utils/sampling.py:16-63

scipy's `Rotation` is the same third-party call the reference makes (torsion.py:100, sampling.py:49).
"""
from __future__ import annotations

import copy
import math

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = 0.5 * angles
    small = angles.abs() < 1e-6
    s = torch.empty_like(angles)
    s[~small] = torch.sin(half[~small]) / angles[~small]
    s[small] = 0.5 - (angles[small] * angles[small]) / 48
    q = torch.cat([torch.cos(half), axis_angle * s], dim=-1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def kabsch(A: torch.Tensor, B: torch.Tensor):
    """A, B: [3, N].  Returns R [3,3], t [3,1] with R A + t ~ B."""
    cA = A.mean(dim=1, keepdim=True)
    cB = B.mean(dim=1, keepdim=True)
    H = (A - cA) @ (B - cB).T
    U, S, Vt = torch.linalg.svd(H)
    Rm = Vt.T @ U.T
    if torch.linalg.det(Rm) < 0:
        SS = torch.diag(torch.tensor([1.0, 1.0, -1.0], dtype=A.dtype))
        Rm = (Vt.T @ SS) @ U.T
    assert math.fabs(torch.linalg.det(Rm) - 1) < 3e-3
    t = -Rm @ cA + cB
    return Rm, t


def modify_conformer_torsion_angles(pos, edge_index, mask_rotate, torsion_updates, norm=None):
    """pos [N,3] (torch/np f32), edge_index [T,2] rotatable bonds, mask_rotate [T,N] bool, norm [K,N,3] points or None."""
    pos = copy.deepcopy(pos)
    if not isinstance(pos, np.ndarray):
        pos = pos.cpu().numpy()
    if norm is not None and not isinstance(norm, np.ndarray):
        norm = norm.cpu().numpy().copy()
    edge_index = edge_index.cpu().numpy() if torch.is_tensor(edge_index) else np.asarray(edge_index)
    for idx_edge, e in enumerate(edge_index):
        if torsion_updates[idx_edge] == 0:
            continue
        u, v = e[0], e[1]
        assert not mask_rotate[idx_edge, u]
        assert mask_rotate[idx_edge, v]
        rot_vec = pos[u] - pos[v]
        rot_vec = rot_vec * torsion_updates[idx_edge] / np.linalg.norm(rot_vec)
        rot_mat = R.from_rotvec(rot_vec).as_matrix()
        pos[mask_rotate[idx_edge]] = (pos[mask_rotate[idx_edge]] - pos[v]) @ rot_mat.T + pos[v]
        if norm is not None:
            norm[:, mask_rotate[idx_edge]] = (norm[:, mask_rotate[idx_edge]] - pos[v]) @ rot_mat.T + pos[v]
    pos = torch.from_numpy(pos.astype(np.float32))
    norm = torch.from_numpy(norm.astype(np.float32)) if norm is not None else None
    return pos, norm


def _mask_rotate(L):
    mr = L.mask_rotate
    return mr if isinstance(mr, np.ndarray) else mr[0]


def modify_conformer(data, tr_update, rot_update, torsion_updates):
    """In-place on data['ligand'].pos / .norm; the [N,33] normal buffer is reinterpreted as [11,N,3] (quirk N6)."""
    L = data["ligand"]
    n = L.x.shape[0]
    lig_center = torch.mean(L.pos, dim=0, keepdim=True)
    lig_norm = L.norm.reshape(-1, n, 3) + L.pos.unsqueeze(0) if "norm" in L and L.norm is not None else None
    rot_mat = axis_angle_to_matrix(rot_update.squeeze())
    rigid_new_pos = (L.pos - lig_center) @ rot_mat.T + tr_update + lig_center
    rigid_new_norm = (lig_norm - lig_center) @ rot_mat.T + tr_update + lig_center if lig_norm is not None else None
    if torsion_updates is not None:
        rot_bonds = data["ligand", "ligand"].edge_index.T[L.edge_mask.bool()]
        flex_pos, flex_norm = modify_conformer_torsion_angles(rigid_new_pos, rot_bonds, _mask_rotate(L), torsion_updates,
                                                              norm=rigid_new_norm)
        Rk, tk = kabsch(flex_pos.T, rigid_new_pos.T)
        aligned_pos = flex_pos @ Rk.T + tk.T
        L.pos = aligned_pos
        if flex_norm is not None:
            L.norm = (flex_norm @ Rk.T + tk.T - aligned_pos).reshape(n, -1)
    else:
        L.pos = rigid_new_pos
        if rigid_new_norm is not None:
            L.norm = (rigid_new_norm - rigid_new_pos).reshape(n, -1)
    return data


def randomize_position(data_list, no_torsion, no_random, tr_sigma_max, draws=None):
    """sampling.py:16-63.  `draws` (optional, for reproducible tests): list of dicts with pre-drawn
    'tor' [T] float64, 'rot' [3,3] float64 rotation matrix, 'tr' [1,3] float32; otherwise the same global RNG calls
    as the reference (np.random.uniform, scipy R.random, torch.normal)."""
    for i, g in enumerate(data_list):
        L = g["ligand"]
        n = L.x.shape[0]
        if not no_torsion:
            T = int(L.edge_mask.sum())
            tor = draws[i]["tor"] if draws is not None else np.random.uniform(low=-np.pi, high=np.pi, size=T)
            norm = L.norm.reshape(-1, n, 3) + L.pos.unsqueeze(0)
            L.pos, L.norm = modify_conformer_torsion_angles(
                L.pos, g["ligand", "ligand"].edge_index.T[L.edge_mask.bool()], _mask_rotate(L), tor, norm=norm)
    for i, g in enumerate(data_list):
        L = g["ligand"]
        n = L.pos.shape[0]
        center = torch.mean(L.pos, dim=0, keepdim=True)
        rot = draws[i]["rot"] if draws is not None else R.random().as_matrix()
        rot = torch.from_numpy(np.asarray(rot)).float()
        L.pos = (L.pos - center) @ rot.T
        # NB (reference as written): when no_torsion the [N,33] vector buffer is used here un-reshaped and would not
        # broadcast; the reference only runs this with torsions enabled, where L.norm holds [11,N,3] absolute points.
        L.norm = ((L.norm - center) @ rot.T - L.pos).reshape(n, -1)
        if not no_random:
            tr = draws[i]["tr"] if draws is not None else torch.normal(mean=0, std=tr_sigma_max, size=(1, 3))
            L.pos = L.pos + torch.as_tensor(tr)
