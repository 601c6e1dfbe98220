"""Property tests of the oracle score model itself (CPU): SE(3) equivariance and batch independence.  These are the
locks on the restated score-model half, whose parity with e3nn-based DiffPhore cannot be pinned here."""
import copy

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from diffphore_amd import config, synthetic, weights
from diffphore_amd.hetero import Batch
from oracle import sampler
from oracle.model import OracleScoreModel
from oracle.tables import load_tables


def _model():
    return OracleScoreModel(weights.init_state_dict(), config.ModelConfig(), dtype=torch.float64, tables=load_tables())


def _run(m, gs, t=0.4):
    b = Batch.from_data_list(gs)
    sampler.set_time_phore(b, t, t, t, b.num_graphs)
    return m(b)


def _transform(g, R, shift):
    g = copy.deepcopy(g)
    R = torch.tensor(R, dtype=torch.float32)
    g["ligand"].pos = g["ligand"].pos @ R.T + shift
    g["phore"].pos = g["phore"].pos @ R.T + shift
    g["ligand"].norm = (g["ligand"].norm.reshape(-1, 11, 3) @ R.T).reshape(-1, 33)
    g["phore"].norm = g["phore"].norm @ R.T
    return g


def test_score_model_equivariant_without_normal_branch():
    """With all ligand pharmacophore fingerprints zero the (non-covariant) normal-vector clip quirk is inactive and the model
    must be exactly SE(3)-equivariant: tr, rot rotate as vectors, torsion scores are invariant."""
    gs = synthetic.synthetic_pairs(42, 2, 14, 3, 12)
    for g in gs:
        g["ligand"].phorefp = torch.zeros_like(g["ligand"].phorefp)
        g["ligand"].norm = torch.zeros_like(g["ligand"].norm)
    m = _model()
    tr, rot, tor = _run(m, gs)
    R = Rotation.random(random_state=6).as_matrix()
    gs2 = [_transform(g, R, torch.tensor([0.5, 1.0, -0.25])) for g in gs]
    tr2, rot2, tor2 = _run(m, gs2)
    Rt = torch.tensor(R, dtype=torch.float64)
    assert torch.allclose(tr2, tr @ Rt.T, atol=5e-5 * float(tr.abs().max()) + 1e-9)
    assert torch.allclose(rot2, rot @ Rt.T, atol=5e-5 * float(rot.abs().max()) + 1e-9)
    assert torch.allclose(tor2, tor, atol=5e-5 * float(tor.abs().max()) + 1e-9)


def test_poses_in_a_batch_are_independent():
    gs = synthetic.synthetic_pairs(43, 3, (10, 16), (0, 4), (8, 14))
    m = _model()
    tr, rot, tor = _run(m, gs)
    tr1, rot1, tor1 = _run(m, gs[1:2])
    assert torch.allclose(tr[1:2], tr1, atol=1e-10) and torch.allclose(rot[1:2], rot1, atol=1e-10)
    t0 = int(gs[0]["ligand"].edge_mask.sum())
    t1 = int(gs[1]["ligand"].edge_mask.sum())
    assert torch.allclose(tor[t0:t0 + t1], tor1, atol=1e-10)
