"""Host-side logic on CPU: container, synthetic inputs, reference-layout weights, schedule helpers, step scalars."""
import copy
import types

import numpy as np
import pytest
import torch

from diffphore_amd import config, diffusion, synthetic, weights
from diffphore_amd.hetero import Batch


def test_batch_roundtrip_and_offsets():
    gs = synthetic.synthetic_pairs(3, 4, (8, 20), (0, 5), (6, 30))
    b = Batch.from_data_list(gs)
    assert b.num_graphs == 4
    n = sum(g["ligand"].pos.shape[0] for g in gs)
    assert b["ligand"].batch.shape[0] == n and int(b["ligand", "ligand"].edge_index.max()) < n
    assert int(b["phore", "phore"].edge_index.max()) < b["phore"].pos.shape[0]
    for a, c in zip(gs, b.to_data_list()):
        for k in ("pos", "x", "norm", "phorefp", "edge_mask"):
            assert torch.equal(torch.as_tensor(a["ligand"][k]), torch.as_tensor(c["ligand"][k]))
        assert np.array_equal(a["ligand"].mask_rotate, c["ligand"].mask_rotate)
        assert torch.equal(a["ligand", "ligand"].edge_index, c["ligand", "ligand"].edge_index)
        assert torch.equal(a["phore", "phore"].edge_index, c["phore", "phore"].edge_index)
        assert torch.equal(a["phore"].phoretype, c["phore"].phoretype)


def test_synthetic_inputs_follow_reference_layout():
    gs = synthetic.synthetic_pairs(2, 3, 30, 8, 32)
    gs2 = synthetic.synthetic_pairs(2, 3, 30, 8, 32)
    for g, h in zip(gs, gs2):
        assert torch.equal(g["ligand"].pos, h["ligand"].pos)                    # deterministic in the seed
        L, P = g["ligand"], g["phore"]
        n = L.pos.shape[0]
        assert L.x.shape == (n, 16) and L.norm.shape == (n, 33) and L.phorefp.shape == (n, 11)
        assert int(L.edge_mask.sum()) == 8 and L.mask_rotate.shape == (8, n)
        ei = g["ligand", "ligand"].edge_index
        assert torch.equal(ei[:, 0::2], ei[:, 1::2].flip(0))                    # both directions adjacent (process_mols.py:270-274)
        assert torch.all(g["ligand", "ligand"].edge_attr.sum(1) == 1)
        rb = ei[:, L.edge_mask]
        for k in range(8):                                                      # flagged direction: u outside, v inside (torsion.py:94-95)
            assert not L.mask_rotate[k, int(rb[0, k])] and L.mask_rotate[k, int(rb[1, k])]
            assert 2 <= L.mask_rotate[k].sum() <= n // 2
        assert torch.all(L.phorefp[:, 10] == 0)
        assert P.x.shape == (32, 5) and torch.all(P.phoretype.sum(1) == 1)
        ex = P.phoretype[:, 10] == 1
        assert torch.equal(P.x[:, 1] == 0, ex) and int(ex.sum()) == 26
        assert torch.allclose(P.pos.mean(0), torch.zeros(3), atol=1e-5)         # centred on the phore centroid
        nn_ = P.norm.norm(dim=1)
        assert torch.all((nn_ < 1e-6) | ((nn_ - 1).abs() < 1e-5))


def test_state_dict_layout_matches_parameter_inventory():
    sd = weights.init_state_dict()
    assert sd["encoder.lig_conv_layers.3.fc.3.weight"].shape == (2200, 60)
    assert sd["encoder.phore_to_lig_norm_conv_layers.0.fc.3.weight"].shape == (600, 60)
    assert sd["encoder.lig_to_phore_conv_layers.2.batch_norm.running_var"].shape == (60,)
    assert sd["encoder.lig_conv_layers.0.batch_norm.running_mean"].shape == (20,)
    assert sd["final_conv.batch_norm.running_mean"].shape == (0,) and sd["final_conv.batch_norm.weight"].shape == (4,)
    assert sd["tor_bond_conv.fc.3.weight"].shape == (1600, 60) and sd["tor_final_layer.0.weight"].shape == (20, 40)
    assert "tor_final_layer.0.bias" not in sd
    assert sd["encoder.cross_edge_embedding.0.weight"].shape == (20, 73)
    assert sd["encoder.lig_node_embedding.atom_embedding_list.0.weight"].shape == (119, 20)
    n_par = sum(v.numel() for k, v in sd.items() if not k.endswith("offset") and "running" not in k)
    assert 2.1e6 < n_par < 2.3e6                                               # SURVEY: ~2.2 M parameters
    # tolerant loader: DataParallel prefix + e3nn-internal buffers are dropped, shapes are checked
    noisy = {("module." + k): v for k, v in sd.items()}
    noisy["module.encoder.lig_conv_layers.0.tp.weight"] = torch.zeros(0)
    noisy["module.encoder.lig_conv_layers.0.tp.output_mask"] = torch.ones(50)
    noisy["module.final_tp_tor.output_mask"] = torch.ones(45)
    clean = weights.clean_state_dict(noisy)
    assert list(clean.keys()) == list(sd.keys())
    bad = dict(sd)
    bad["tr_final_layer.0.weight"] = torch.zeros(20, 5)
    with pytest.raises(ValueError):
        weights.clean_state_dict(bad)
    del bad["tr_final_layer.0.weight"]
    with pytest.raises(KeyError):
        weights.clean_state_dict(bad)


def test_unsupported_configurations_are_rejected():
    c = config.ModelConfig()
    c.check_supported()
    for k, v in (("ns", 16), ("use_att", True), ("use_second_order_repr", True), ("atom_weight", "softmax"), ("consider_norm", False)):
        c2 = copy.deepcopy(c)
        setattr(c2, k, v)
        with pytest.raises(NotImplementedError):
            c2.check_supported()
    assert [config.layer_numel(l) for l in range(4)] == [600, 1100, 1600, 2200]


@pytest.mark.parametrize("key,val", [("lig_max_radius", 6.0), ("phore_max_radius", 4.0), ("cross_max_distance", 30.0), ("center_max_distance", 25.0),
                                     ("ns", 16), ("num_conv_layers", 3), ("use_att", True), ("by_radius", True), ("atom_weight", "softmax"),
                                     ("phoretype_match", False), ("multiple", True), ("sh_lmax", 3), ("clash_cutoff", [1.0, 2.0]),
                                     ("confidence_mode", True), ("contrastive_model", object()), ("num_phoretype", 12)])
def test_constructor_rejects_every_keyword_that_would_change_the_model(key, val):
    """a keyword of the reference constructor (src/models/score_model_phore.py:166-180) whose value the kernels are not built for raises - it is
    never accepted and ignored (cut-offs included: the ligand radius graph uses 5 A inside the kernels)"""
    from diffphore_amd.score_model import TensorProductScoreModel
    with pytest.raises(NotImplementedError):
        TensorProductScoreModel(device="cuda", **{key: val})


def test_constructor_keyword_surface():
    from diffphore_amd import score_model as sm
    # unknown keywords are a TypeError, as for any Python callable (the reference constructor takes **kwargs only to pass them on, :180)
    with pytest.raises(TypeError, match="frobnicate"):
        sm.TensorProductScoreModel(device="cuda", frobnicate=1)
    # keywords the reference stores or passes on but never reads in this configuration are accepted with any value and change nothing
    ref = sm.TensorProductScoreModel(device="cuda").cfg
    for k in sm._INERT:
        for v in (True, 3, "sum"):
            assert sm.TensorProductScoreModel(device="cuda", **{k: v}).cfg == ref, k
    assert {"dynamic_max_cross", "norm_by_ph", "as_exp", "readout"} <= set(sm._INERT)
    # each entry names the reference lines that prove it; the lines exist where the reference is available (the build container)
    import os
    src = "/root/reference/src/models/score_model_phore.py"
    if os.path.exists(src):
        text = open(src).read().splitlines()
        import re
        for k, proof in sm._INERT.items():
            lines = [int(x) for x in re.findall(r"(\d{3})", proof.split(" ")[0])]
            assert lines, (k, proof)
            assert any(re.search(rf"\b{k}\b", text[n - 1]) for n in lines), (k, proof)
        # ... and a keyword of the reference signature is in exactly one of the three sets (or handled by name)
        sig = " ".join(text[165:180])
        names = set(re.findall(r"(\w+)=", sig)) | {"t_to_sigma", "device", "timestep_emb_func"}
        handled = set(sm._KW) | set(sm._INERT) | set(sm._MUST) | {"clash_cutoff", "t_to_sigma", "device", "timestep_emb_func"}
        assert names <= handled, sorted(names - handled)
    # the get_model mirror forwards everything the reference's get_model forwards (src/utils/utils.py:121-166)
    args = types.SimpleNamespace(no_torsion=False, num_conv_layers=4, max_radius=5.0, scale_by_sigma=True, sigma_embed_dim=20, ns=20, nv=10,
                                 distance_embed_dim=20, cross_distance_embed_dim=20, no_batch_norm=False, dropout=0.1, use_second_order_repr=False,
                                 cross_max_distance=25.0, dynamic_max_cross=True, consider_norm=True, phore_rule=True, auto_phorefp=False, angle_match=True,
                                 cross_distance_transition=True, phore_direction_transition=True, phoretype_match_transition=True, new=True, ex_factor=-2.0,
                                 boarder=True, by_radius=False, clash_tolerance=0.4, use_att=False, use_phore_match_feat=True, atom_weight="phore",
                                 norm_by_ph=True, readout="sum", as_exp=True, scaler=100.0)
    m = sm.get_model(args, "cuda", None)
    assert m.cfg == ref
    with pytest.raises(NotImplementedError):
        sm.get_model(args, "cuda", None, confidence_mode=True)
    args.max_radius = 6.0
    with pytest.raises(NotImplementedError):
        sm.get_model(args, "cuda", None)


def test_step_scalars_follow_the_reference_formulas():
    from diffphore_amd.sampling import _step_params
    cfg = config.ModelConfig()
    args = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)
    sched = diffusion.get_t_schedule(20)
    assert np.allclose(sched, np.linspace(1, 0.05, 20))
    t2s = diffusion.t_to_sigma_fn(args)
    sp = _step_params(0, 20, sched, sched, sched, t2s, args, False, False)
    tr_g = 5.0 * np.sqrt(2 * np.log(50.0))                                      # sampling.py:223 at t = 1
    assert abs(sp.tr_c1 - tr_g ** 2 * 0.05) < 1e-4 and abs(sp.tr_c2 - tr_g * np.sqrt(0.05)) < 1e-5
    rot_g = 2 * 1.5 * np.sqrt(np.log(15.0))
    assert abs(sp.rot_c1 - rot_g ** 2 * 0.05) < 1e-5
    last = _step_params(19, 20, sched, sched, sched, t2s, args, False, True)    # dt = t on the last step; no_final_step_noise
    assert last.tr_c2 == 0.0 and abs(last.tor_c1 - (t2s(0.05, 0.05, 0.05)[2] * np.sqrt(2 * np.log(100.0))) ** 2 * 0.05) < 1e-7
    ode = _step_params(3, 20, sched, sched, sched, t2s, args, True, False)
    sde = _step_params(3, 20, sched, sched, sched, t2s, args, False, False)
    assert abs(ode.tr_c1 - 0.5 * sde.tr_c1) < 1e-7 and ode.tr_c2 == 0.0


def test_pose_limits_are_enforced_before_any_launch():
    """>256 atoms / >128 torsions per pose are rejected by the packer (kernel LDS limits), on any machine."""
    from diffphore_amd.engine import PackedBatch
    g = synthetic.synthetic_pairs(5, 1, 12, 2, 10)[0]
    big = copy.deepcopy(g)
    n = 257
    big["ligand"].pos = torch.zeros(n, 3)
    with pytest.raises(ValueError):
        PackedBatch([big], "cpu")


def test_transformation_mask_matches_reference_code(golden_dir):
    """`synthetic.transformation_mask` (which bonds rotate, which side moves) against the reference's own
    `get_transformation_mask` run on the same bond lists (tests/golden/make_mask_golden.py)."""
    import os
    G = np.load(os.path.join(golden_dir, "ref_transformation_mask.npz"))
    for k in range(int(G["n"])):
        seed, n, t = (int(v) for v in G[f"m{k}_spec"])
        lig = synthetic.make_ligand(np.random.default_rng(seed), n, t)
        me, mr = synthetic.transformation_mask(n, np.asarray(lig["edge_index"]))
        np.testing.assert_array_equal(me, G[f"m{k}_edges"])
        np.testing.assert_array_equal(mr, G[f"m{k}_rotate"])
        # the generator keeps a subset of those bonds as torsions, with the reference's rows for them
        sel = np.asarray(lig["edge_mask"], dtype=bool)
        assert (me[sel]).all() and sel.sum() == np.asarray(lig["mask_rotate"]).shape[0]
        rows = np.cumsum(me) - 1
        np.testing.assert_array_equal(np.asarray(lig["mask_rotate"]), mr[rows[sel]])


def test_batch_keeps_graph_level_tensors_and_fingerprint_tracks_static_content():
    """`to_data_list` must hand graph-level tensors (original_center) back; the packed-engine cache of the score model is
    keyed on the batch's static CONTENT, never on `id(data)`."""
    from diffphore_amd.score_model import TensorProductScoreModel
    gs = synthetic.synthetic_pairs(3, 3, (8, 12), (1, 3), (6, 9))
    b = Batch.from_data_list(gs)
    for g, c in zip(gs, b.to_data_list()):
        assert torch.equal(torch.as_tensor(g.original_center).reshape(1, 3), c.original_center.reshape(1, 3))
    fp = TensorProductScoreModel._fingerprint
    k1 = fp(b)
    assert fp(Batch.from_data_list(gs)) == k1                       # same content, new object -> same key
    b2 = Batch.from_data_list(gs)
    b2["ligand"].pos = b2["ligand"].pos + 1.0                      # diffusion state is not part of the key
    assert fp(b2) == k1
    other = synthetic.synthetic_pairs(4, 3, (8, 12), (1, 3), (6, 9))
    keys = {k1}
    for variant in (Batch.from_data_list(other), Batch.from_data_list(gs[::-1]), Batch.from_data_list(gs[:2])):
        keys.add(fp(variant))
    assert len(keys) == 4
    b3 = Batch.from_data_list(gs)
    b3["ligand"].x = b3["ligand"].x.clone()
    b3["ligand"].x[0, 1] = (b3["ligand"].x[0, 1] + 1) % 4          # an in-place edit of a static field changes the key
    assert fp(b3) != k1


def test_bench_traffic_record_lookup_is_robust(tmp_path, monkeypatch):
    """bench.py reads roofline.traffic from profiles/conv_hbm_traffic.json: the committed file must parse, and neither a bare record, a
    list with junk entries nor a missing file may take the bench line down with them."""
    import json
    import bench
    assert isinstance(bench.measured_traffic(2, 512, "f32", "node"), tuple)           # the committed file
    rec = dict(source_hash=bench.kernel_source_hash(), config=2, poses=512, precision="f32", conv_algo="node", traffic_bytes=123, profile="x.csv")
    for payload in (rec, [rec], ["junk", 5, rec]):
        f = tmp_path / "t.json"
        f.write_text(json.dumps(payload))
        monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
        assert bench.measured_traffic(2, 512, "f32", "node") == (123, "x.csv")
        assert bench.measured_traffic(2, 512, "f32", "edge") == (None, None)
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(tmp_path / "absent.json"))
    assert bench.measured_traffic(2, 512, "f32", "node") == (None, None)


def test_fingerprint_is_cached_on_the_batch_object_and_invalidated_by_edits():
    """the content hash is computed once per batch object; later calls compare O(1) identity / version keys only, and an
    in-place edit or a replaced static tensor recomputes it."""
    from diffphore_amd.score_model import TensorProductScoreModel
    gs = synthetic.synthetic_pairs(3, 3, (8, 12), (1, 3), (6, 9))
    m = TensorProductScoreModel.__new__(TensorProductScoreModel)
    calls = []
    real = TensorProductScoreModel._fingerprint
    m._fingerprint = lambda d: (calls.append(1), real(d))[1]
    b = Batch.from_data_list(gs)
    k1 = m._fingerprint_cached(b)
    b["ligand"].pos = b["ligand"].pos + 1.0                        # diffusion state: no recomputation
    assert m._fingerprint_cached(b) == k1 and len(calls) == 1
    b["ligand"].x[0, 1] = (b["ligand"].x[0, 1] + 1) % 4            # in-place edit of a static tensor bumps its version
    k2 = m._fingerprint_cached(b)
    assert k2 != k1 and len(calls) == 2
    b["phore"].pos = b["phore"].pos.clone()                        # replaced (equal content): rehashed, same key
    assert m._fingerprint_cached(b) == k2 and len(calls) == 3
    assert m._fingerprint_cached(Batch.from_data_list(gs)) == k1 and len(calls) == 4      # a new object never inherits a cache entry


def test_repeated_ordered_bond_drops_the_pair_tables():
    from diffphore_amd.engine import PackedBatch
    """A ligand graph that lists the same ordered bond twice cannot be held in the node-side conv's N x N pair tables: the packed batch
    then carries no tables (llp_ptr NULL, n_llp 0), which keeps dp_forward's ligand-ligand convs on the edge-side CSR kernel."""
    gs = synthetic.synthetic_pairs(5, 2, 10, 2, 8)
    pb = PackedBatch(gs, "cpu", upload=False)
    assert pb.ll_pairs_ok
    g = gs[1]
    ei, ea = g["ligand", "ligand"].edge_index, g["ligand", "ligand"].edge_attr
    g["ligand", "ligand"].edge_index = torch.cat([ei, ei[:, :1]], 1)
    g["ligand", "ligand"].edge_attr = torch.cat([ea, ea[:1]], 0)
    g["ligand"].edge_mask = torch.cat([torch.as_tensor(g["ligand"].edge_mask), torch.zeros(1, dtype=torch.bool)])
    pb2 = PackedBatch(gs, "cpu", upload=False)
    assert not pb2.ll_pairs_ok
