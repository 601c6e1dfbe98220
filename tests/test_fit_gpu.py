"""SURVEY.md 8f rows on the GPU: `sample_step`, the best-of-R branching sampler and the `fit` driver."""
import copy
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffphore_amd import config, sdf_io, synthetic, weights  # noqa: E402
from diffphore_amd.hetero import Batch  # noqa: E402


def _args(**kw):
    cfg = config.ModelConfig()
    base = {k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max", "tor_sigma_min", "tor_sigma_max")}
    base.update(no_torsion=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


def _model():
    from diffphore_amd.score_model import TensorProductScoreModel
    return TensorProductScoreModel(device="cuda").load_state_dict(weights.init_state_dict())


def _noise(S, R, gs, seed):
    gen = torch.Generator().manual_seed(seed)
    nT = sum(int(g["ligand"].edge_mask.sum()) for g in gs)
    return dict(tr=torch.randn(S, R, len(gs), 3, generator=gen), rot=torch.randn(S, R, len(gs), 3, generator=gen),
                tor=torch.randn(S, R, nT, generator=gen))


def test_sample_step_is_one_sampler_step_and_matches_oracle():
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import sample_step, sampling_phore
    from oracle import sampler as osampler
    from oracle.model import OracleScoreModel
    from oracle.tables import load_tables
    gs = synthetic.synthetic_pairs(311, 5, (12, 20), (1, 5), (10, 18))
    args = _args()
    m = _model()
    t = 0.4
    nz = _noise(1, 1, gs, 5)
    # (a) one-step sampling_phore with schedule [t]: dt = t on the last step
    a, _ = sampling_phore(copy.deepcopy(gs), m, 1, [t], [t], [t], "cuda", t_to_sigma_fn(args), args,
                          noise=dict(tr=nz["tr"][:, 0], rot=nz["rot"][:, 0], tor=nz["tor"][:, 0]))
    # (b) sample_step on the collated batch
    batch = Batch.from_data_list(copy.deepcopy(gs))
    osampler.set_time_phore(batch, t, t, t, batch.num_graphs)
    sig = t_to_sigma_fn(args)(t, t, t)
    b, tor_p, tr_p, rot_p = sample_step(batch, m, args, *sig, delta_t=t, noise=(nz["tr"][0, 0], nz["rot"][0, 0], nz["tor"][0, 0]))
    for x, y in zip(a, b):
        assert torch.equal(x["ligand"].pos, y["ligand"].pos) and torch.equal(x["ligand"].norm, y["ligand"].norm)
    # (c) the returned perturbations are c1*score + c2*z with the oracle's scores
    om = OracleScoreModel(weights.init_state_dict(), config.ModelConfig(), dtype=torch.float64, tables=load_tables())
    ob = Batch.from_data_list(copy.deepcopy(gs))
    osampler.set_time_phore(ob, t, t, t, ob.num_graphs)
    tr_s, rot_s, tor_s = om(ob)
    tr_g = float(sig[0]) * np.sqrt(2 * np.log(args.tr_sigma_max / args.tr_sigma_min))
    rot_g = 2 * float(sig[1]) * np.sqrt(np.log(args.rot_sigma_max / args.rot_sigma_min))
    tor_g = float(sig[2]) * np.sqrt(2 * np.log(args.tor_sigma_max / args.tor_sigma_min))
    np.testing.assert_allclose(tr_p.numpy(), (tr_g ** 2 * t * tr_s + tr_g * np.sqrt(t) * nz["tr"][0, 0]).numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rot_p.numpy(), (rot_g ** 2 * t * rot_s + rot_g * np.sqrt(t) * nz["rot"][0, 0]).numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(tor_p, (tor_g ** 2 * t * tor_s + tor_g * np.sqrt(t) * nz["tor"][0, 0]).numpy(), rtol=2e-4, atol=2e-4)
    assert tr_p.shape == (5, 3) and tor_p.shape == tor_s.shape
    # no_random: deterministic and different
    b2 = Batch.from_data_list(copy.deepcopy(gs))
    osampler.set_time_phore(b2, t, t, t, b2.num_graphs)
    c, _, tr0, _ = sample_step(b2, m, args, *sig, delta_t=t, no_random=True)
    np.testing.assert_allclose(tr0.numpy(), (tr_g ** 2 * t * tr_s).numpy(), rtol=2e-4, atol=2e-4)


def test_branching_sampler_selects_best_candidate_per_pose():
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import sampling_phore, sampling_phore_with_fitscore
    gs = synthetic.synthetic_pairs(312, 4, (12, 18), (1, 4), (10, 16))
    for i, g in enumerate(gs):
        g.original_center = torch.tensor([[1.0 * i, 0.0, -2.0]])
    S, R = 3, 3
    sched = get_t_schedule(S)
    m = _model()
    nz = _noise(S, R, gs, 11)
    # R <= 1: identical to sampling_phore under the same noise
    args1 = _args(random_samples=0)
    a, _ = sampling_phore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args1), args1,
                          noise=dict(tr=nz["tr"][:, 0], rot=nz["rot"][:, 0], tor=nz["tor"][:, 0]))
    b, _ = sampling_phore_with_fitscore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args1), args1,
                                        noise={k: v[:, :1] for k, v in nz.items()})
    for x, y in zip(a, b):
        assert torch.equal(x["ligand"].pos, y["ligand"].pos)
    # R = 3 with a geometric score (closer to the origin of the docked frame is better); record the choices
    argsR = _args(random_samples=R)
    choices = []

    def score(dock, data_list):
        assert len(dock) == len(data_list) * R and dock[0].shape == (int(data_list[0]["ligand"].pos.shape[0]), 3)
        s = np.asarray([-np.linalg.norm(d.mean(0)) for d in dock])
        choices.append(s.reshape(len(data_list), R).argmax(1))
        return s

    c, _ = sampling_phore_with_fitscore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(argsR), argsR,
                                        fitscore_fn=score, noise=nz, no_final_step_noise=True)
    assert len(choices) == S - 1                                   # the noise-free last step does not branch
    # replay: the plain sampler fed with each pose's chosen candidate noise reproduces the branching run bit for bit
    tor_ptr = np.concatenate([[0], np.cumsum([int(g["ligand"].edge_mask.sum()) for g in gs])])
    sel = dict(tr=torch.zeros(S, len(gs), 3), rot=torch.zeros(S, len(gs), 3), tor=torch.zeros(S, int(tor_ptr[-1])))
    for s_ in range(S - 1):
        for i in range(len(gs)):
            j = int(choices[s_][i])
            sel["tr"][s_, i], sel["rot"][s_, i] = nz["tr"][s_, j, i], nz["rot"][s_, j, i]
            sel["tor"][s_, tor_ptr[i]:tor_ptr[i + 1]] = nz["tor"][s_, j, tor_ptr[i]:tor_ptr[i + 1]]
    d, _ = sampling_phore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(argsR), argsR, noise=sel,
                          no_final_step_noise=True)
    for x, y in zip(c, d):
        assert torch.equal(x["ligand"].pos, y["ligand"].pos) and torch.equal(x["ligand"].norm, y["ligand"].norm)
    assert any(len(set(ch.tolist())) > 1 for ch in choices)       # different poses picked different candidates
    with pytest.raises(ValueError):
        sampling_phore_with_fitscore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(argsR), argsR)
    with pytest.raises(NotImplementedError):
        sampling_phore_with_fitscore(copy.deepcopy(gs), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(args1), args1, ode=True)


def _mol_for(g, name):
    """A V2000 record with one carbon per heavy atom, the graph's bonds, plus two explicit hydrogens the driver must drop."""
    pos = g["ligand"].pos.numpy().astype(np.float64)
    n = pos.shape[0]
    m = sdf_io.Mol(title=name, program="  synthetic", comment="")
    m.coords = np.concatenate([pos, pos[:2] + 0.9])
    m.symbols = ["C"] * n + ["H", "H"]
    m.atom_tail = [" C   0  0  0  0  0  0  0  0  0  0  0  0"] * n + [" H   0  0  0  0  0  0  0  0  0  0  0  0"] * 2
    ei = g["ligand", "ligand"].edge_index.numpy()
    seen = set()
    for a, b in ei.T:
        if (min(a, b), max(a, b)) not in seen:
            seen.add((min(a, b), max(a, b)))
            m.bonds.append([int(a) + 1, int(b) + 1, 1])
            m.bond_tail.append("  0")
    m.bonds += [[1, n + 1, 1], [2, n + 2, 1]]
    m.bond_tail += ["  0", "  0"]
    return m


def test_fit_driver_files_resume_and_retry(tmp_path, monkeypatch, capsys):
    from diffphore_amd import fit as fitmod
    from diffphore_amd.diffusion import t_to_sigma_fn
    gs = synthetic.synthetic_pairs(313, 3, (12, 18), (1, 4), (10, 16))
    for i, g in enumerate(gs):
        g.name = f"tgt__lig{i}"
        g.original_center = torch.tensor([[10.0 + i, -3.0, 2.0]])
        g.mol = _mol_for(g, g.name)
    N = 4
    args = _args(run_dir=str(tmp_path), out_dir=str(tmp_path), sample_per_complex=N, inference_steps=3, overwrite=False, batch_size=20,
                 no_random=False, ode=False, no_final_step_noise=True, complexes_per_launch=2, cutoff=None)
    m = _model()

    def scorer(a, dock_pose, name, g):
        return [-float(np.linalg.norm(p.mean(0) - np.array([10.0, -3.0, 2.0]))) for p in dock_pose]

    torch.manual_seed(1)
    np.random.seed(1)
    res = fitmod.fit(args, m, gs, "cuda", t_to_sigma_fn(args), scorer=scorer)
    assert res["name"] == [g.name for g in gs] and all(len(s) == N for s in res["fitscore"])
    for g, sc in zip(gs, res["fitscore"]):
        recs = sdf_io.read_sdf(os.path.join(tmp_path, f"ranked_poses/{g.name}_ranked.sdf"))
        assert [r.title for r in recs] == [f"{g.name}_rank_{i}" for i in range(N)]
        fs = [float(r.data["fitscore"]) for r in recs]
        assert fs == sorted(sc, reverse=True)                                         # best first
        assert all(r.num_atoms == g["ligand"].pos.shape[0] and "H" not in r.symbols for r in recs)   # hydrogens dropped
        # ranked coordinates are in the original frame: their score, recomputed from the file, matches
        np.testing.assert_allclose([-np.linalg.norm(r.coords.mean(0) - np.array([10.0, -3.0, 2.0])) for r in recs], fs, atol=2e-4)
        # bonded geometry of the template survives (torsion + rigid moves only)
        d_ref = [np.linalg.norm(g.mol.coords[a - 1] - g.mol.coords[b - 1]) for a, b, _ in recs[0].bonds]
        d_new = [np.linalg.norm(recs[0].coords[a - 1] - recs[0].coords[b - 1]) for a, b, _ in recs[0].bonds]
        np.testing.assert_allclose(d_new, d_ref, atol=3e-4)
        log = json.load(open(os.path.join(tmp_path, f"mapping_process/{g.name}/{g.name}_dock.log")))
        assert log["name"] == g.name and log["fitscore"] == sc
    df = fitmod.analyze_results(args, res)
    assert os.path.exists(os.path.join(tmp_path, "ranked_results.csv")) and len(df) == 3

    # resume: nothing is sampled again
    def boom(*a, **k):
        raise AssertionError("resumed complexes must not be re-sampled")
    monkeypatch.setattr(fitmod, "sampling_phore", boom)
    res2 = fitmod.fit(args, m, gs, "cuda", t_to_sigma_fn(args), scorer=scorer)
    assert res2["fitscore"] == res["fitscore"] and res2["name"] == res["name"]

    # SVD-retry convention: 'failed to converge' is retried up to 5 times per complex, then the complex is skipped; no scorer -> -2.0
    calls = {"n": 0}
    real = fitmod.__dict__["randomize_position"]

    def flaky(**kw):
        calls["n"] += 1
        if calls["n"] <= 3:
            raise RuntimeError("linalg.svd: Kabsch alignment failed to converge for poses [0]")
        from diffphore_amd.sampling import sampling_phore as sp
        return sp(**kw)
    monkeypatch.setattr(fitmod, "sampling_phore", flaky)
    args.overwrite = True
    res3 = fitmod.fit(args, m, gs[:1], "cuda", t_to_sigma_fn(args))
    assert calls["n"] == 4 and res3["fitscore"] == [[-2.0] * N]

    def dead(**kw):
        raise RuntimeError("failed to converge")
    monkeypatch.setattr(fitmod, "sampling_phore", dead)
    res4 = fitmod.fit(args, m, gs[:1], "cuda", t_to_sigma_fn(args))
    assert res4["name"] == [] and "skipping the complex" in capsys.readouterr().out
    assert real is fitmod.randomize_position


def test_fit_overlap_surrogate_matches_its_oracle_and_ranks_on_device():
    """dp_fit_overlap (8f-3) vs the numpy restatement; then the branching sampler scoring candidates on the device."""
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.engine import PackedBatch
    from diffphore_amd.fitness import overlap_scorer, overlap_scores
    from diffphore_amd.sampling import sampling_phore_with_fitscore
    from oracle import fitness as ofit
    gs = synthetic.synthetic_pairs(314, 6, (10, 40), (0, 6), (8, 79))
    pb = PackedBatch(gs, "cuda")
    got = overlap_scores(pb, ex_lambda=0.7).cpu().numpy()
    want = np.asarray([ofit.overlap_score_graph(g, ex_lambda=0.7) for g in gs])
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    # candidate coordinates instead of the batch state; moving the ligand far away kills the overlap and the clash term
    far = pb.pos + 100.0
    np.testing.assert_allclose(overlap_scores(pb, far).cpu().numpy(), 0.0, atol=1e-12)
    np.testing.assert_allclose(overlap_scores(pb, pb.pos.clone(), 0.7).cpu().numpy(), got, rtol=0, atol=0)
    # an atom sitting exactly on every feature it matches scores the full volume (no exclusion spheres counted: lambda = 0)
    g = copy.deepcopy(gs[0])
    types_ = np.asarray(g["phore"].phoretype).argmax(1)
    feat = np.where(types_ != 10)[0]
    n = g["ligand"].pos.shape[0]
    assert len(feat) <= n
    g["ligand"].pos[:len(feat)] = g["phore"].pos[feat]
    fp = torch.zeros_like(g["ligand"].phorefp)
    fp[torch.arange(len(feat)), torch.as_tensor(types_[feat])] = 1.0
    g["ligand"].phorefp = fp
    np.testing.assert_allclose(overlap_scores(PackedBatch([g], "cuda"), ex_lambda=0.0).cpu().numpy(), [1.0], rtol=1e-6)
    # scorer hook of fit(): docked frame in, same numbers out
    g0 = gs[1]
    g0.original_center = torch.tensor([[3.0, -1.0, 2.0]])
    dock = np.stack([g0["ligand"].pos.numpy() + g0.original_center.numpy(), g0["ligand"].pos.numpy() + 1.5 + g0.original_center.numpy()])
    sc = overlap_scorer(types.SimpleNamespace(), dock, "x", g0)
    np.testing.assert_allclose(sc, [ofit.overlap_score_graph(g0), ofit.overlap_score_graph(g0, pos=g0["ligand"].pos.numpy() + 1.5)], rtol=2e-5, atol=2e-6)

    # branching sampler with the built-in score == branching sampler with the same score computed on the host
    S, R = 3, 3
    sched = get_t_schedule(S)
    argsR = _args(random_samples=R)
    m = _model()
    gs2 = synthetic.synthetic_pairs(315, 4, (12, 18), (1, 4), (10, 16))
    nz = _noise(S, R, gs2, 21)

    def host_score(dock, data_list):
        out = []
        for i, g_ in enumerate(data_list):
            for j in range(R):
                c = g_.original_center.reshape(1, 3).numpy() if "original_center" in g_ else 0.0   # candidates arrive in the docked frame
                out.append(ofit.overlap_score_graph(g_, pos=dock[i * R + j] - c))
        return out
    a, _ = sampling_phore_with_fitscore(copy.deepcopy(gs2), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(argsR), argsR,
                                        fitscore_fn="overlap", noise=nz, no_final_step_noise=True)
    b, _ = sampling_phore_with_fitscore(copy.deepcopy(gs2), m, S, sched, sched, sched, "cuda", t_to_sigma_fn(argsR), argsR,
                                        fitscore_fn=host_score, noise=nz, no_final_step_noise=True)
    same = sum(bool(torch.equal(x["ligand"].pos, y["ligand"].pos)) for x, y in zip(a, b))
    assert same == len(gs2), same


def test_sample_step_and_fitscore_sampler_match_reference_code(golden_dir):
    """8f-4 against the reference's OWN `sample_step` / `sampling_phore_with_fitscore` (tests/golden/make_model_golden.py):
    same global RNG seeding, same draw order; the ODE case pins the reference's un-squared translation coefficient (:344)."""
    from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
    from diffphore_amd.sampling import randomize_position, sample_step, sampling_phore_with_fitscore
    from oracle import sampler as osampler
    G = np.load(os.path.join(golden_dir, "ref_model_glue.npz"))
    m = _model()
    args = _args()
    gs = synthetic.synthetic_pairs(311, 5, 16, 3, 14)
    batch = Batch.from_data_list(gs)
    t = 0.4
    osampler.set_time_phore(batch, t, t, t, batch.num_graphs)
    torch.manual_seed(11)
    dl, tor_p, tr_p, rot_p = sample_step(batch, m, args, *t_to_sigma_fn(args)(t, t, t), delta_t=0.07)
    np.testing.assert_allclose(tr_p.numpy(), G["step_tr"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(rot_p.numpy(), G["step_rot"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(tor_p, G["step_tor"], rtol=2e-4, atol=2e-5)
    pos = np.stack([g["ligand"].pos.numpy() for g in dl])
    assert np.sqrt(((pos - G["step_pos"]) ** 2).sum(-1).mean(-1)).max() < 2e-5
    S = 4
    sched = get_t_schedule(S)
    for name, a2, kw in (("fs_sde", args, dict(ode=False)), ("fs_ode_notor", _args(no_torsion=True), dict(ode=True))):
        gs = synthetic.synthetic_pairs(97, 6, 18, 4, 20)
        np.random.seed(9)
        torch.manual_seed(9)
        randomize_position(gs, False, False, args.tr_sigma_max)
        np.testing.assert_allclose(np.stack([g["ligand"].pos.numpy() for g in gs]), G[f"{name}_start"], atol=5e-6)
        res, _ = sampling_phore_with_fitscore(gs, m, S, sched, sched, sched, "cuda", t_to_sigma_fn(a2), a2, batch_size=3, **kw)
        pos = np.stack([g["ligand"].pos.numpy() for g in res])
        r = np.sqrt(((pos - G[f"{name}_pos"]) ** 2).sum(-1).mean(-1))
        # every pose within 1e-3 A of the reference's own run, and at most one of the six beyond 1e-4 (the reference itself runs in fp32:
        # a pose that passes a discontinuous decision within fp32 resolution lands on either side - tests/test_forward_gpu.py)
        assert (r < 1e-3).all() and int((r >= 1e-4).sum()) <= 1, (name, r)


def test_fit_driver_matches_reference_fit(golden_dir, tmp_path):
    """8f-1 against the reference's OWN `inference.py::fit` run end to end in the build container (reference model + sampler,
    AncPhore replaced by a geometric score; tests/golden/make_fit_golden.py).  One complex per launch so the global RNG is
    consumed in the reference's order; the scores are centroid distances in A, so 1e-3 bounds the docked poses themselves."""
    from diffphore_amd import fit as fitmod
    from diffphore_amd.diffusion import t_to_sigma_fn
    G = np.load(os.path.join(golden_dir, "ref_fit_helpers.npz"))
    gs = synthetic.synthetic_pairs(313, 3, 14, 3, 12)
    for i, g in enumerate(gs):
        g.name = f"tgt__lig{i}"
        g.original_center = torch.tensor([[10.0 + i, -3.0, 2.0]])
    args = _args(run_dir=str(tmp_path), sample_per_complex=3, inference_steps=3, overwrite=True, batch_size=20, no_random=False, ode=False,
                 no_final_step_noise=True, complexes_per_launch=1)

    def scorer(a, dock_pose, name, g):
        return [float(-np.linalg.norm(p.mean(0) - np.array([10.0, -3.0, 2.0]))) for p in dock_pose]
    np.random.seed(13)
    torch.manual_seed(13)
    res = fitmod.fit(args, _model(), gs, "cuda", t_to_sigma_fn(args), scorer=scorer)
    assert res["name"] == [str(n) for n in G["fit_names"]]
    np.testing.assert_allclose(np.asarray(res["fitscore"]), G["fit_scores"], atol=1e-3)
    for n, want in zip(res["name"], G["fit_log_scores"]):
        log = json.load(open(os.path.join(tmp_path, f"mapping_process/{n}/{n}_dock.log")))
        np.testing.assert_allclose(log["fitscore"], want, atol=1e-3)
        assert set(log) == {"name", "fitscore", "run_time"}


def test_inference_epoch_matches_reference(golden_dir):
    """The validation-time caller (`utils/training.py::inference_epoch`) against the reference's own function run in the build
    container (tests/golden/make_fit_golden.py): per-sample RMSDs to 1e-3 A with the same global-RNG seeding, same metric keys."""
    from diffphore_amd import fit as fitmod
    from diffphore_amd.diffusion import t_to_sigma_fn
    G = np.load(os.path.join(golden_dir, "ref_fit_helpers.npz"))
    gs = synthetic.synthetic_pairs(314, 3, 14, 3, 12)
    for i, g in enumerate(gs):
        g.name = f"val__lig{i}"
        g.original_center = torch.tensor([[1.0 + i, 2.0, -3.0]])
        g["ligand"].orig_pos = g["ligand"].pos.numpy() + g.original_center.numpy()
    args = _args(dataset="pdbbind", sample_per_complex=3, inference_steps=3, keep_update=False, fitscore=False, complexes_per_launch=1)
    np.random.seed(17)
    torch.manual_seed(17)
    em = fitmod.inference_epoch(_model(), gs, "cuda", t_to_sigma_fn(args), args)
    np.testing.assert_allclose(np.asarray(em["rmsd"]), G["epoch_rmsd"], atol=1e-3)
    assert em["rmsds_lt2"] == float(G["epoch_lt2"]) and em["rmsds_lt5"] == float(G["epoch_lt5"])
    assert sorted(em) == ["failed_indices", "fitscore", "rmsd", "rmsds_lt2", "rmsds_lt5", "run_time"] and em["failed_indices"] == []
    # batched launch + keep_update bookkeeping + a scorer: shapes and invariants
    args2 = _args(dataset="pdbbind", sample_per_complex=3, inference_steps=3, keep_update=True, fitscore=True, complexes_per_launch=8)
    em2 = fitmod.inference_epoch(_model(), gs, "cuda", t_to_sigma_fn(args2), args2, scorer=lambda a, d, n, g: [0.8, 0.5, 0.1])
    assert np.asarray(em2["rmsd"]).shape == (3, 3) and em2["fitscore_gt0.7"] == 100 * 3 / 3 and em2["fitscore_gt0.4"] == 100 * 6 / 3
    dp = em2["dock_process"]
    assert dp["name"] == [g.name for g in gs] and len(dp["docked_poses"][0][0]) == 3 and len(dp["perturb_pos"][0]) == 3
    np.testing.assert_allclose(dp["docked_poses"][1][2][-1], dp["docked_poses"][1][2][-1])
