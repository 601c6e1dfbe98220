"""Pin the oracle's conformer-update half against outputs of the reference's own functions.

tests/golden/ref_utils.npz was produced by tests/golden/make_golden.py importing /root/reference/src/utils/{geometry,
torsion,diffusion_utils}.py in the build container.  CPU only.
"""
import os
import types

import numpy as np
import pytest
import torch

from diffphore_amd.hetero import HeteroData
from oracle import conformer, sampler, tables
from oracle.model import sinusoidal_embedding


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_utils.npz"))


def test_axis_angle_to_matrix(G):
    out = conformer.axis_angle_to_matrix(torch.from_numpy(G["aa_in"]))
    np.testing.assert_array_equal(out.numpy(), G["aa_out"])


def test_kabsch(G):
    for i in range(G["kabsch_A"].shape[0]):
        R, t = conformer.kabsch(torch.from_numpy(G["kabsch_A"][i]), torch.from_numpy(G["kabsch_B"][i]))
        np.testing.assert_allclose(R.numpy(), G["kabsch_R"][i], atol=1e-6)
        np.testing.assert_allclose(t.numpy(), G["kabsch_t"][i], atol=1e-6)
    # last case is the reflection branch
    assert np.linalg.det(G["kabsch_R"][-1]) > 0.99


def _graph(G, pre):
    g = HeteroData()
    n = G[pre + "pos"].shape[0]
    g["ligand"].pos = torch.from_numpy(G[pre + "pos"]).clone()
    g["ligand"].x = torch.zeros(n, 16)
    g["ligand"].norm = torch.from_numpy(G[pre + "norm"]).clone()
    g["ligand"].edge_mask = torch.from_numpy(G[pre + "edge_mask"])
    g["ligand"].mask_rotate = G[pre + "mask_rotate"]
    g["ligand", "ligand"].edge_index = torch.from_numpy(G[pre + "edge_index"])
    return g


@pytest.mark.parametrize("case", [0, 1, 2])
def test_torsion_update(G, case):
    pre = f"tor{case}_"
    pos = torch.from_numpy(G[pre + "pos"])
    n = pos.shape[0]
    pts = torch.from_numpy(G[pre + "norm"]).reshape(-1, n, 3) + pos.unsqueeze(0)
    rot_edges = torch.from_numpy(G[pre + "edge_index"]).T[torch.from_numpy(G[pre + "edge_mask"])]
    p2, n2 = conformer.modify_conformer_torsion_angles(pos, rot_edges, G[pre + "mask_rotate"], G[pre + "theta"], norm=pts)
    np.testing.assert_array_equal(p2.numpy(), G[pre + "pos_out"])
    np.testing.assert_array_equal(n2.numpy(), G[pre + "normpts_out"])


@pytest.mark.parametrize("case", [0, 1, 2])
def test_modify_conformer(G, case):
    pre = f"tor{case}_"
    g = _graph(G, pre)
    conformer.modify_conformer(g, torch.from_numpy(G[pre + "mc_tr"]), torch.from_numpy(G[pre + "mc_rot"]), G[pre + "theta"])
    np.testing.assert_allclose(g["ligand"].pos.numpy(), G[pre + "mc_pos_out"], atol=2e-6)
    np.testing.assert_allclose(g["ligand"].norm.numpy(), G[pre + "mc_norm_out"], atol=4e-6)
    g = _graph(G, pre)
    conformer.modify_conformer(g, torch.from_numpy(G[pre + "mc_tr"]), torch.from_numpy(G[pre + "mc_rot"]), None)
    np.testing.assert_array_equal(g["ligand"].pos.numpy(), G[pre + "rigid_pos_out"])
    np.testing.assert_array_equal(g["ligand"].norm.numpy(), G[pre + "rigid_norm_out"])


def test_norm_slot_quirk(G):
    """N6: slot s of the [N,33] buffer moves with atom (s mod N), not atom s // 11."""
    pre = "tor1_"
    g = _graph(G, pre)
    n = g["ligand"].pos.shape[0]
    theta = np.zeros_like(G[pre + "theta"])
    theta[0] = 1.0
    before = g["ligand"].norm.clone().reshape(-1, 3)
    conformer.modify_conformer(g, torch.zeros(1, 3), torch.zeros(3), theta)
    after = g["ligand"].norm.reshape(-1, 3)
    moved = ((after - before).abs().max(dim=1).values > 1e-4).numpy()
    nonzero = (before.abs().max(dim=1).values > 0).numpy()
    # after the Kabsch re-alignment every non-zero slot changes a little, but slots attached to rotated atoms change most;
    # check the slot->atom rule on the un-aligned torsion update instead
    pos = torch.from_numpy(G[pre + "pos"])
    pts = torch.from_numpy(G[pre + "norm"]).reshape(-1, n, 3) + pos.unsqueeze(0)
    rot_edges = torch.from_numpy(G[pre + "edge_index"]).T[torch.from_numpy(G[pre + "edge_mask"])]
    p2, n2 = conformer.modify_conformer_torsion_angles(pos, rot_edges, G[pre + "mask_rotate"], theta, norm=pts)
    vec_before = (pts - pos.unsqueeze(0)).reshape(-1, 3)
    vec_after = (n2 - p2.unsqueeze(0)).reshape(-1, 3)
    changed = ((vec_after - vec_before).abs().max(dim=1).values > 1e-5).numpy()
    slot_atom = np.arange(11 * n) % n
    expect = G[pre + "mask_rotate"][0][slot_atom] & (vec_before.abs().max(dim=1).values > 0).numpy()
    np.testing.assert_array_equal(changed, expect)
    assert moved.sum() >= expect.sum() and nonzero.sum() > 0


def test_schedule_sigma_embedding(G):
    np.testing.assert_array_equal(sampler.get_t_schedule(20), G["t_schedule20"])
    np.testing.assert_array_equal(sampler.get_t_schedule(7), G["t_schedule7"])
    args = types.SimpleNamespace(tr_sigma_min=0.1, tr_sigma_max=5.0, rot_sigma_min=0.1, rot_sigma_max=1.5,
                                 tor_sigma_min=0.0314, tor_sigma_max=3.14)
    sig = np.asarray([sampler.t_to_sigma(t, t, t, args) for t in G["t_schedule20"]])
    np.testing.assert_array_equal(sig, G["sigma20"])
    emb = sinusoidal_embedding(10000 * torch.from_numpy(G["sin_emb_t"]), 20)
    np.testing.assert_array_equal(emb.numpy(), G["sin_emb20"])


def test_so3_table_rows_match_reference():
    so3, _ = tables.load_tables()
    assert so3.shape == (1000,)
    idx = [0, 1, 250, 500, 777, 999]
    rows = tables.so3_exp_score_norm_rows(idx)
    np.testing.assert_allclose(rows, so3[idx], rtol=1e-9)


def test_torus_table_rows_within_mc_noise():
    _, tor = tables.load_tables()
    assert tor.shape == (5001,)
    idx = [0, 1000, 2500, 3640, 4000, 5000]
    rows = tables.torus_score_norm_rows(idx, seed=7)
    np.testing.assert_allclose(rows, tor[idx], rtol=0.08)     # 10k-sample Monte-Carlo estimate on both sides


def test_table_lookups_match_reference(golden_dir):
    P = np.load(os.path.join(golden_dir, "ref_tables_probe.npz"))
    np.testing.assert_array_equal(tables.so3_score_norm(P["so3_eps"]), P["so3_score_norm"])
    np.testing.assert_array_equal(tables.torus_score_norm(P["torus_sigma"]), P["torus_score_norm"])


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_modify_conformer_many_torsions(golden_dir, case):
    """BASELINE configs[2]'s shape (80 atoms x 25 torsions), the kernel-limit shape (128 x 40) and a 45 x 12 tree: the oracle's
    modify_conformer against the reference's own outputs (tests/golden/make_torsion25_golden.py)."""
    G25 = np.load(os.path.join(golden_dir, "ref_torsion25.npz"))
    pre = f"c{case}_"
    g = _graph(G25, pre)
    conformer.modify_conformer(g, torch.from_numpy(G25[pre + "mc_tr"]), torch.from_numpy(G25[pre + "mc_rot"]), G25[pre + "theta"])
    np.testing.assert_allclose(g["ligand"].pos.numpy(), G25[pre + "mc_pos_out"], atol=4e-6)
    np.testing.assert_allclose(g["ligand"].norm.numpy(), G25[pre + "mc_norm_out"], atol=8e-6)
