"""BASELINE configs[2] (large ligands: ~80 atoms / 25 torsions) - the sampler half and the node-side kernel above its ligand-ligand cap.

  * conformer update at 25 / 40 torsions held DIRECTLY to the reference's own `modify_conformer` outputs
    (tests/golden/ref_torsion25.npz; /root/reference/src/utils/torsion.py:64-109 is sequential over bonds) and to the oracle;
  * 6-step SDE / ODE trajectories of 80-atom x 25-torsion poses, HIP {edge, node} vs the fp64 oracle, every pose <= 1e-4 A or an
    oracle-recorded near-tie decision;
  * a batch of >= 20 poses with >= 100 atoms each forced onto the node-side formulation: cross convs node-side, ligand-ligand convs on
    the edge-side CSR kernel (DP_NODE_LL_MAX_ATOMS = 96), against the fp64 oracle.
"""
import copy
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffphore_amd import config, synthetic, weights  # noqa: E402
from diffphore_amd.hetero import Batch  # noqa: E402
from test_forward_gpu import (_assert_poses_or_decision_flip, _golden_graph, _model, _oracle_run_with_margins, _rel, _rmsd)  # noqa: E402


def _margs(**kw):
    cfg = config.ModelConfig()
    return types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False, **kw)


def test_conformer_update_many_torsions_matches_reference_golden(golden_dir):
    """dp_conformer_update at 25 and 40 torsions vs outputs of the REFERENCE's own modify_conformer, one batch with all four cases."""
    from diffphore_amd.engine import DpStepParams, PackedBatch, _declare, _stream
    G = np.load(os.path.join(golden_dir, "ref_torsion25.npz"))
    n_case = 4
    graphs = [_golden_graph(G, f"c{c}_") for c in range(n_case)]
    assert [int(g["ligand"].edge_mask.sum()) for g in graphs] == [25, 25, 40, 12]
    L = _declare()
    pb = PackedBatch(graphs, "cuda")
    tr = torch.cat([torch.from_numpy(G[f"c{c}_mc_tr"]) for c in range(n_case)]).cuda().contiguous()
    rot = torch.stack([torch.from_numpy(G[f"c{c}_mc_rot"]) for c in range(n_case)]).cuda().contiguous()
    tor = torch.cat([torch.from_numpy(G[f"c{c}_theta"]) for c in range(n_case)]).cuda().contiguous()
    sp = DpStepParams()
    sp.tr_c1 = sp.rot_c1 = sp.tor_c1 = 1.0
    status = torch.zeros(n_case, dtype=torch.int32, device="cuda")
    rc = L.dp_conformer_update(C.byref(pb.struct), C.byref(sp), tr.data_ptr(), rot.data_ptr(), tor.data_ptr(), None, None, None,
                               status.data_ptr(), _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0
    pos, norm = pb.pos.cpu().numpy(), pb.norm.cpu().numpy()
    for c in range(n_case):
        lo, hi = int(pb.lig_ptr[c]), int(pb.lig_ptr[c + 1])
        err = np.abs(pos[lo:hi] - G[f"c{c}_mc_pos_out"]).max()
        print(f"case {c}: {hi - lo} atoms, max |dpos| vs reference = {err:.2e} A")
        np.testing.assert_allclose(pos[lo:hi], G[f"c{c}_mc_pos_out"], atol=5e-6)
        np.testing.assert_allclose(norm[lo:hi], G[f"c{c}_mc_norm_out"], atol=1e-5)


def test_conformer_update_configs2_shape_matches_oracle():
    """One dp_conformer_update call on synthetic configs[2] poses (80 atoms, 25 torsions) vs oracle.conformer.modify_conformer
    (itself pinned to the reference at this torsion count by tests/test_oracle_golden.py) <= 5e-6 A."""
    from diffphore_amd.engine import DpStepParams, PackedBatch, _declare, _stream
    from oracle import conformer
    gs = synthetic.synthetic_pairs(3, 4, 80, 25, 32)
    assert all(int(g["ligand"].edge_mask.sum()) == 25 for g in gs)
    rng = np.random.default_rng(31)
    tr = rng.normal(size=(4, 3)).astype(np.float32)
    rot = (0.4 * rng.normal(size=(4, 3))).astype(np.float32)
    tor = rng.uniform(-np.pi, np.pi, size=(4, 25)).astype(np.float32)
    ref = copy.deepcopy(gs)
    for i, g in enumerate(ref):
        conformer.modify_conformer(g, torch.from_numpy(tr[i:i + 1]), torch.from_numpy(rot[i]), tor[i])
    L = _declare()
    pb = PackedBatch(gs, "cuda")
    sp = DpStepParams()
    sp.tr_c1 = sp.rot_c1 = sp.tor_c1 = 1.0
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    dtr, drot, dtor = (torch.from_numpy(x).cuda().contiguous() for x in (tr, rot, tor.reshape(-1)))
    assert L.dp_conformer_update(C.byref(pb.struct), C.byref(sp), dtr.data_ptr(), drot.data_ptr(), dtor.data_ptr(), None, None, None,
                                 status.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0
    pos, norm = pb.pos.cpu().numpy(), pb.norm.cpu().numpy()
    for i, g in enumerate(ref):
        lo, hi = int(pb.lig_ptr[i]), int(pb.lig_ptr[i + 1])
        np.testing.assert_allclose(pos[lo:hi], g["ligand"].pos.numpy(), atol=5e-6)
        np.testing.assert_allclose(norm[lo:hi], g["ligand"].norm.numpy().reshape(-1, 33), atol=1e-5)


@pytest.mark.parametrize("ode,algo", [(False, "edge"), (False, "node"), (True, "edge"), (True, "node")])
def test_configs2_sampler_trajectory_matches_oracle(ode, algo, monkeypatch):
    """configs[2]'s sampler half: 4 poses of 80 atoms x 25 torsions x 32 phore nodes, randomised start, 6 steps."""
    monkeypatch.setenv("DIFFPHORE_CONV_ALGO", algo)
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import sampling_phore
    from diffphore_amd.score_model import TensorProductScoreModel
    from oracle import conformer
    from scipy.spatial.transform import Rotation
    gs = synthetic.synthetic_pairs(3, 4, 80, 25, 32)
    rng = np.random.default_rng(23)
    draws = [dict(tor=rng.uniform(-np.pi, np.pi, size=int(g["ligand"].edge_mask.sum())),
                  rot=Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix(),
                  tr=torch.from_numpy(rng.normal(size=(1, 3)).astype(np.float32))) for g in gs]
    conformer.randomize_position(gs, False, False, 5.0, draws=draws)
    for g in gs:
        g["ligand"].norm = g["ligand"].norm.reshape(-1, 33)
    S = 6
    sched = get_t_schedule(S)
    args = _margs()
    gen = torch.Generator().manual_seed(25)
    nT = sum(int(g["ligand"].edge_mask.sum()) for g in gs)
    assert nT == 100
    noise = dict(tr=torch.randn(S, len(gs), 3, generator=gen), rot=torch.randn(S, len(gs), 3, generator=gen),
                 tor=torch.randn(S, nT, generator=gen))
    sd, om = _model()
    a, margin, which = _oracle_run_with_margins(om, gs, S, sched, args, ode=ode, batch_size=4, noise=noise)
    m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
    b, _ = sampling_phore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, ode=ode, batch_size=4, noise=noise)
    r = np.asarray([_rmsd(x["ligand"].pos.numpy(), y["ligand"].pos.numpy()) for x, y in zip(a, b)])
    print(f"configs[2] 6-step {'ode' if ode else 'sde'} {algo}: per-pose RMSD (A):", r)
    _assert_poses_or_decision_flip(r, margin, which, f"configs[2] 6-step {'ode' if ode else 'sde'} {algo}")
    # the torsion updates really moved the conformers (not a rigid run)
    d0 = torch.cdist(gs[0]["ligand"].pos, gs[0]["ligand"].pos)
    d1 = torch.cdist(b[0]["ligand"].pos, b[0]["ligand"].pos)
    assert (d0 - d1).abs().max() > 1e-2


def test_node_side_above_the_ligand_ligand_cap():
    """20 poses of 100-110 atoms forced onto DIFFPHORE_CONV_ALGO=node: the cross convs run node-side (7 node groups per pose), the
    ligand-ligand convs fall back to the edge-side CSR kernel (more than DP_NODE_LL_MAX_ATOMS = 96 atoms) - vs the fp64 oracle."""
    from diffphore_amd.engine import Engine, PackedBatch
    from diffphore_amd.score_model import TensorProductScoreModel
    from oracle import sampler as osampler
    gs = synthetic.synthetic_pairs(811, 10, 100, 14, 24) + synthetic.synthetic_pairs(812, 10, 110, 10, 40)
    sd, om = _model()
    batch = Batch.from_data_list(gs)
    osampler.set_time_phore(batch, 0.45, 0.45, 0.45, batch.num_graphs)
    tr_o, rot_o, tor_o = om(batch)
    os.environ["DIFFPHORE_CONV_ALGO"] = "node"
    try:
        m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
        tr, rot, tor = m(batch)
        torch.cuda.synchronize()
        assert m._engine.conv_algo == "node"
        errs = dict(h_l4=_rel(m._engine.ws.t["h_l0"][:, :100], om.debug["h_l4"]), tr=_rel(tr, tr_o), rot=_rel(rot, rot_o), tor=_rel(tor, tor_o))
        print("node-side above the lig-lig cap, relative errors vs fp64 oracle:", {k: f"{v:.1e}" for k, v in errs.items()})
        assert all(v < 1e-5 for v in errs.values()), errs
        # and the edge-side formulation agrees with it on the same batch
        os.environ["DIFFPHORE_CONV_ALGO"] = "edge"
        m2 = TensorProductScoreModel(device="cuda").load_state_dict(sd)
        tr2, rot2, tor2 = m2(batch)
        assert _rel(tr2, tr.cpu()) < 1e-5 and _rel(rot2, rot.cpu()) < 1e-5 and _rel(tor2, tor.cpu()) < 1e-5
    finally:
        os.environ.pop("DIFFPHORE_CONV_ALGO", None)
    del Engine, PackedBatch


def test_repeated_ordered_bond_falls_back_to_the_edge_side_lig_conv():
    """a ligand multigraph (same ordered bond listed twice) used to raise ValueError on the default path.  Now the batch is
    packed without pair tables, the ligand-ligand convs run on the edge-side CSR kernel (cross convs stay node-side), no status bit is
    raised and the scores match the fp64 oracle, which processes the duplicated edge like the reference would."""
    from diffphore_amd.score_model import TensorProductScoreModel
    from oracle import sampler as osampler
    gs = synthetic.synthetic_pairs(640, 24, 14, 3, 12)
    g = gs[5]
    ei, ea = g["ligand", "ligand"].edge_index, g["ligand", "ligand"].edge_attr
    g["ligand", "ligand"].edge_index = torch.cat([ei, ei[:, :1]], 1)
    g["ligand", "ligand"].edge_attr = torch.cat([ea, ea[:1]], 0)
    g["ligand"].edge_mask = torch.cat([torch.as_tensor(g["ligand"].edge_mask), torch.zeros(1, dtype=torch.bool)])
    sd, om = _model()
    batch = Batch.from_data_list(gs)
    osampler.set_time_phore(batch, 0.6, 0.6, 0.6, batch.num_graphs)
    tr_o, rot_o, tor_o = om(batch)
    m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
    tr, rot, tor = m(batch)
    torch.cuda.synchronize()
    eng = m._engine
    assert eng.conv_algo == "node" and not eng.pb.ll_pairs_ok
    assert int(eng.ws.t["status"].abs().sum()) == 0
    assert _rel(tr, tr_o) < 1e-5 and _rel(rot, rot_o) < 1e-5 and _rel(tor, tor_o) < 1e-5
