"""CPU restatement of the pharmacophore-fit SURROGATE (`dp_fit_overlap`).  TEST INFRASTRUCTURE ONLY.

Parity unpinned: the reference ranks poses with the closed AncPhore binary (`src/utils/sampling.py:447-498`), which is not in
/root/reference; the only part of its scoring the reference itself restates is the Gaussian-volume kernel and constants in
`get_perfect_similarity` (`src/inference.py:273-295`), which this follows.  The checker below pins the HIP kernel to this formula,
not to AncPhore.
"""
import numpy as np


def gaussian_volume(a1, a2, r2):
    """inference.py:287-289 (comment quoting AncPhore): GCI2 * (pi/(a1+a2))^1.5 * exp(-(a1 a2) r2 / (a1+a2)), GCI2 = 7.999999999."""
    return 7.999999999 * (np.pi / (a1 + a2)) ** 1.5 * np.exp(-(a1 * a2) * r2 / (a1 + a2))


def overlap_score(lig_pos, phorefp, ph_pos, ph_type, ph_alpha, ph_weight, ex_lambda=1.0):
    lig_pos, ph_pos = np.asarray(lig_pos, np.float64), np.asarray(ph_pos, np.float64)
    r2 = ((lig_pos[:, None, :] - ph_pos[None, :, :]) ** 2).sum(-1)            # [N_l, N_p]
    V = R = X = 0.0
    for m in range(ph_pos.shape[0]):
        a, w, t = float(ph_alpha[m]), float(ph_weight[m]), int(ph_type[m])
        if t != 10:
            match = np.asarray(phorefp)[:, t] != 0
            best = gaussian_volume(a, a, r2[match, m]).max() if match.any() else 0.0
            V += w * best
            R += w * gaussian_volume(a, a, 0.0)
        else:
            X += np.exp(-0.5 * a * r2[:, m]).sum()
    return (V / R if R > 0 else -1.0) - ex_lambda * X / max(lig_pos.shape[0], 1)


def overlap_score_graph(g, pos=None, ex_lambda=1.0):
    x = np.asarray(g["phore"].x)
    return overlap_score(np.asarray(g["ligand"].pos if pos is None else pos), np.asarray(g["ligand"].phorefp), np.asarray(g["phore"].pos),
                         np.asarray(g["phore"].phoretype).argmax(1), x[:, 3], x[:, 4], ex_lambda)
