"""The native collate (csrc/dp_pack.hip: dp_pack_query / dp_pack_fill / dp_pack_bind, host code) against the numpy packer it replaces
(engine.PackedBatch._pack_numpy, the readable statement of the layout): every array of the DpBatch image byte for byte, the totals, the
hints and the error behaviour.  The reference's counterpart is PyG's Batch.from_data_list over the pose list (src/utils/sampling.py:210-215).
No GPU: both packers are host code."""
import ctypes as C

import numpy as np
import pytest
import torch

from diffphore_amd import engine, synthetic
from diffphore_amd.engine import PackedBatch


def _native_arrays(pb):
    lay, tot, bn = pb._layout, pb._totals, pb._block.numpy()
    B, n_lig, n_ph = pb.B, int(tot.n_lig), int(tot.n_ph)
    cnt = dict(lig_ptr=B + 1, ph_ptr=B + 1, bond_ptr=B + 1, tor_ptr=B + 1, pp_ptr=B + 1, x_ptr=B + 1, ll_ptr=B + 1, mask_ptr=B + 1, pose_ptr=B + 1,
               llp_ptr=B + 1, lig_x=n_lig * 16, bond_src=tot.n_bond, bond_dst=tot.n_bond, bond_type=tot.n_bond, tor_u=tot.n_tor, tor_v=tot.n_tor,
               mask_rotate=int(pb.mask_ptr[-1]), phorefp=n_lig * 11, norm_angle1=n_lig * 11, norm_angle2=n_lig * 11, ph_pos=n_ph * 3, ph_norm=n_ph * 3,
               ph_x=n_ph * 3, ph_scalar=n_ph * 2, ph_type=n_ph, pp_src=tot.n_pp, pp_dst=tot.n_pp, pp_rowptr=n_ph, pp_rowend=n_ph, pos=n_lig * 3,
               norm=n_lig * 33, xs_src=tot.n_xs, xs_nbr=tot.n_xs, xs_feat=tot.n_xs, xs_rowptr=n_lig + 1, xt_src=tot.n_xs, xt_nbr=tot.n_xs, xt_feat=tot.n_xs,
               xt_rowptr=n_ph + 1)
    fl = {"phorefp", "norm_angle1", "norm_angle2", "ph_pos", "ph_norm", "ph_scalar", "pos", "norm"}
    out = {}
    for k, n in cnt.items():
        dt = np.uint8 if k == "mask_rotate" else (np.float32 if k in fl else np.int32)
        off = int(getattr(lay, k))
        assert off % 256 == 0
        out[k] = bn[off:off + int(n) * np.dtype(dt).itemsize].view(dt).copy()
    return out


def _batch():
    gs = synthetic.synthetic_pairs(31, 5, (9, 40), (0, 9), (6, 70))                 # ragged ligands / pharmacophores, some without torsions
    gs += synthetic.synthetic_pairs(32, 1, 16, 7, 33)                               # N_l = 16, N_p = 33: both "short last pass" hints
    poses = [g.sample_copy() for g in gs for _ in range(3)]                        # samples share their complex's static tensors
    for k, p in enumerate(poses):
        p["ligand"].pos = p["ligand"].pos + 0.01 * k                              # ... and own their diffusion state
    return poses


def test_native_pack_equals_numpy_pack():
    poses = _batch()
    a = PackedBatch(poses, "cpu", upload=False, host_pack="native")
    b = PackedBatch(poses, "cpu", upload=False, host_pack="numpy")
    na = _native_arrays(a)
    for k, v in b._host.items():
        w = v.numpy().reshape(-1)
        if k == "mask_rotate" and int(b.mask_ptr[-1]) == 0:
            continue
        if k[:3] in ("xs_", "xt_") and not k.endswith("rowptr") and b._n_xs == 0:
            continue
        assert np.array_equal(na[k], w.view(na[k].dtype) if w.dtype != na[k].dtype and w.dtype.itemsize == na[k].dtype.itemsize else w.astype(na[k].dtype)), k
    assert np.array_equal(na["pos"], b._host_pos.numpy().reshape(-1)) and np.array_equal(na["norm"], b._host_norm.numpy().reshape(-1))
    for k in ("lig_ptr", "ph_ptr", "bond_ptr", "tor_ptr", "pp_ptr", "x_ptr", "ll_ptr", "mask_ptr", "llp_ptr"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert np.array_equal(a.nl, b.nl) and np.array_equal(a.np_, b.np_) and np.array_equal(a.nt, b.nt) and a.ll_pairs_ok == b.ll_pairs_ok
    # totals and hints of the DpBatch the numpy path would fill
    t = a._totals
    assert (t.B, t.n_lig, t.n_ph, t.n_bond, t.n_tor, t.n_pp, t.n_x, t.ll_cap, t.n_llp) == (
        len(poses), int(b.lig_ptr[-1]), int(b.ph_ptr[-1]), int(b.bond_ptr[-1]), int(b.tor_ptr[-1]), int(b.pp_ptr[-1]), int(b.x_ptr[-1]), int(b.ll_ptr[-1]),
        int(b.llp_ptr[-1]))
    assert (t.max_nl, t.max_np, t.max_t) == (int(b.nl.max()), int(b.np_.max()), int(b.nt.max()))
    # the cross pairs with agreement != 0 (DpBatch.xs_* / xt_*): the same set in both orders, exactly the pairs the definition names
    assert t.n_xs == b._n_xs > 0
    fp, ty = b._host["phorefp"].numpy().reshape(-1, 11), b._host["ph_type"].numpy()
    want = set()
    for p in range(len(poses)):
        for i in range(int(b.lig_ptr[p]), int(b.lig_ptr[p + 1])):
            for j in range(int(b.ph_ptr[p]), int(b.ph_ptr[p + 1])):
                if ty[j] != 10 and fp[i, ty[j]] != 0:
                    want.add((i, j, int(b.x_ptr[p]) + (i - int(b.lig_ptr[p])) * int(b.np_[p]) + (j - int(b.ph_ptr[p]))))
    assert set(zip(na["xs_src"].tolist(), na["xs_nbr"].tolist(), na["xs_feat"].tolist())) == want and len(want) == t.n_xs
    assert set(zip(na["xt_nbr"].tolist(), na["xt_src"].tolist(), na["xt_feat"].tolist())) == want
    assert np.all(np.diff(na["xs_src"]) >= 0) and np.all(np.diff(na["xt_src"]) >= 0)
    assert na["xs_rowptr"][-1] == t.n_xs == na["xt_rowptr"][-1]
    for a_ in range(int(t.n_lig)):
        assert np.all(na["xs_src"][na["xs_rowptr"][a_]:na["xs_rowptr"][a_ + 1]] == a_)
    assert a._flags & 2 and a._flags & 4                                            # the 16-atom / 33-node complex ends on short passes
    # dp_pack_bind: pointers = base + offsets, hints from the flags
    s = engine.DpBatch()
    L = engine._declare()
    assert L.dp_pack_bind(C.byref(a._layout), 1 << 30, a._flags, C.byref(s)) == 0
    assert s.lig_x == (1 << 30) + a._layout.lig_x and s.norm == (1 << 30) + a._layout.norm and s.no_tail_nl == 0 and s.no_tail_np == 0


def test_native_pack_of_a_large_batch_and_a_replaced_static_on_a_later_sample():
    """210 poses (70 complexes x 3 samples): the same image and flags as the numpy packer; then ONE sample - not the first of its complex - gets a
    replaced (malformed) static tensor: it must not ride on the view its siblings validated in the same batch"""
    gs = synthetic.synthetic_pairs(36, 70, (12, 36), (0, 3), (5, 40))
    poses = [g.sample_copy() for g in gs for _ in range(3)]                        # 210 poses
    for k, p in enumerate(poses):
        p["ligand"].pos = p["ligand"].pos + 0.001 * k
    a = PackedBatch(poses, "cpu", upload=False, host_pack="native")
    b = PackedBatch(poses, "cpu", upload=False, host_pack="numpy")
    na = _native_arrays(a)
    for k, v in b._host.items():
        w = v.numpy().reshape(-1)
        if (k == "mask_rotate" and int(b.mask_ptr[-1]) == 0) or (k[:3] in ("xs_", "xt_") and not k.endswith("rowptr") and b._n_xs == 0):
            continue
        assert np.array_equal(na[k], w.view(na[k].dtype) if w.dtype != na[k].dtype and w.dtype.itemsize == na[k].dtype.itemsize else w.astype(na[k].dtype)), k
    assert np.array_equal(na["pos"], b._host_pos.numpy().reshape(-1)) and np.array_equal(na["norm"], b._host_norm.numpy().reshape(-1))
    assert a.ll_pairs_ok == b.ll_pairs_ok and int(a._totals.n_xs) == b._n_xs
    tail = lambda v: bool((((v - 1) % 32) < 16).any())                              # noqa: E731
    assert bool(a._flags & 2) == tail(np.asarray(b.nl)) and bool(a._flags & 4) == tail(np.asarray(b.np_))
    bad = poses[-2]
    pt = bad["phore"].phoretype.clone()
    pt[0] = 0.0
    bad["phore"].phoretype = pt
    with pytest.raises(ValueError, match="one-hot"):
        PackedBatch(poses, "cpu", upload=False, host_pack="native")


def test_ph_shared_is_found_by_both_packers():
    """DpBatch.ph_shared: every pose on pose 0's pharmacophore (content, not tensor identity) - the samples of one complex, a screen against one
    pharmacophore; not a mixed batch, not a single pose, not a pharmacophore that differs in one coordinate"""
    def both(poses):
        a = PackedBatch(poses, "cpu", upload=False, host_pack="native")
        b = PackedBatch(poses, "cpu", upload=False, host_pack="numpy")
        assert int(a._totals.ph_shared) == b._ph_shared
        return b._ph_shared
    gs = synthetic.synthetic_pairs(35, 3, (10, 14), 3, 12)
    assert both(gs) == 0                                                            # three different pharmacophores
    assert both([gs[0].sample_copy() for _ in range(4)]) == 1                       # samples of one complex
    assert both([gs[0]]) == 0                                                       # a single pose: nothing to share
    lib = synthetic.ScreenLibrary(4, 3)
    units = [lib.unit(i) for i in range(3)]
    assert units[0]["phore"].pos is not units[1]["phore"].pos
    screen = [u.sample_copy() for u in units for _ in range(2)]
    assert both(screen) == 1                                                        # different ligands, the same pharmacophore (equal content)
    units[2]["phore"].pos = units[2]["phore"].pos.clone()
    units[2]["phore"].pos[3, 1] += 1e-6
    assert both([u.sample_copy() for u in units for _ in range(2)]) == 0
    units = [lib.unit(i) for i in range(3)]
    ei = units[1]["phore", "phore"].edge_index.clone()
    ei[:, [0, 1]] = ei[:, [1, 0]]                                                   # the same edges in another order
    units[1]["phore", "phore"].edge_index = ei
    assert both([u.sample_copy() for u in units]) == 0


def test_views_are_cached_per_complex_and_rebuilt_when_a_static_changes():
    gs = synthetic.synthetic_pairs(33, 2, 12, 3, 10)
    poses = [g.sample_copy() for g in gs for _ in range(4)]
    PackedBatch(poses, "cpu", upload=False, host_pack="native")
    views = {id(p._dp_view) for p in poses}
    assert len(views) == 2                                                          # one view per complex, shared by its samples
    v0 = poses[0]._dp_view
    PackedBatch(poses, "cpu", upload=False, host_pack="native")
    assert poses[0]._dp_view is v0                                                  # reused
    poses[0]["ligand"].phorefp = poses[0]["ligand"].phorefp.clone()                 # a replaced static tensor on ONE sample
    a = PackedBatch(poses, "cpu", upload=False, host_pack="native")
    assert poses[0]._dp_view is not v0 and poses[1]._dp_view is v0
    poses[1]["ligand"].x[0, 1] = (poses[1]["ligand"].x[0, 1] + 1) % 4               # an in-place edit (shared by samples 1-3 of complex 0)
    a = PackedBatch(poses, "cpu", upload=False, host_pack="native")
    b = PackedBatch(poses, "cpu", upload=False, host_pack="numpy")
    assert np.array_equal(_native_arrays(a)["lig_x"], b._host["lig_x"].numpy().reshape(-1))


def test_native_pack_rejects_what_the_numpy_packer_rejects():
    gs = synthetic.synthetic_pairs(34, 2, 12, 3, 10)
    bad = gs[1]
    pt = bad["phore"].phoretype.clone()
    pt[0] = 0.0
    bad["phore"].phoretype = pt
    for hp in ("native", "numpy"):
        with pytest.raises(ValueError, match="one-hot"):
            PackedBatch(gs, "cpu", upload=False, host_pack=hp)
    gs = synthetic.synthetic_pairs(34, 2, 12, 3, 10)
    ea = gs[0]["ligand", "ligand"].edge_attr.clone()
    ea[0] = torch.tensor([0.5, 0.5, 0.0, 0.0])
    gs[0]["ligand", "ligand"].edge_attr = ea
    for hp in ("native", "numpy"):
        with pytest.raises(ValueError, match="one-hot"):
            PackedBatch(gs, "cpu", upload=False, host_pack=hp)
    L = engine._declare()
    assert L.dp_pack_query(0, None, None, None) == -1


def test_repeated_ordered_bond_is_flagged_by_both_packers():
    gs = synthetic.synthetic_pairs(5, 2, 10, 2, 8)
    g = gs[1]
    ei, ea = g["ligand", "ligand"].edge_index, g["ligand", "ligand"].edge_attr
    g["ligand", "ligand"].edge_index = torch.cat([ei, ei[:, :1]], 1)
    g["ligand", "ligand"].edge_attr = torch.cat([ea, ea[:1]], 0)
    g["ligand"].edge_mask = torch.cat([torch.as_tensor(g["ligand"].edge_mask), torch.zeros(1, dtype=torch.bool)])
    for hp in ("native", "numpy"):
        assert not PackedBatch(gs, "cpu", upload=False, host_pack=hp).ll_pairs_ok


class _PlainGraph:
    """a graph type WITHOUT a custom __deepcopy__ (stands for PyG's HeteroData, which the reference uses): default copy / pickle walk __dict__"""

    def __init__(self, g):
        self.g = g

    def __getitem__(self, k):
        return self.g[k]


def test_a_packed_graph_still_pickles_and_deepcopies():
    """the cached host view (_dp_view: a ctypes struct of pointers) is a cache, not state: pickle / deepcopy of a graph that went through
    PackedBatch drop it (it is rebuilt at the next packing) instead of failing with 'ctypes objects containing pointers cannot be pickled'"""
    import copy
    import pickle
    gs = synthetic.synthetic_pairs(35, 2, 12, 3, 10)
    a = PackedBatch(gs, "cpu", upload=False, host_pack="native")
    assert gs[0]._dp_view is not None
    g2 = pickle.loads(pickle.dumps(gs[0]))
    assert getattr(g2, "_dp_view", None) is None
    g3 = copy.deepcopy(gs[1])
    assert getattr(g3, "_dp_view", None) is None
    b = PackedBatch([g2, g3], "cpu", upload=False, host_pack="native")
    assert all(np.array_equal(x, y) for x, y in zip(_native_arrays(a).values(), _native_arrays(b).values()))
    # a wrapper type with default copy semantics around a packed graph
    w = _PlainGraph(gs[0])
    object.__setattr__(w, "_dp_view", gs[0]._dp_view)
    assert copy.deepcopy(w)._dp_view is None and pickle.loads(pickle.dumps(w))._dp_view is None


def test_numpy_statics_are_tracked_by_content():
    """mask_rotate kept as an ndarray (the reference's layout) has no version counter: an in-place edit must still rebuild the view"""
    gs = synthetic.synthetic_pairs(36, 1, 12, 3, 10)
    g = gs[0]
    mr = g["ligand"].mask_rotate
    g["ligand"].mask_rotate = np.ascontiguousarray((mr[0] if isinstance(mr, (list, tuple)) else mr).numpy() if torch.is_tensor(mr[0] if isinstance(mr, (list, tuple)) else mr) else np.asarray(mr))
    a = PackedBatch([g], "cpu", upload=False, host_pack="native")
    v0 = g._dp_view
    m = g["ligand"].mask_rotate
    m[0, 0] = not m[0, 0]
    b = PackedBatch([g], "cpu", upload=False, host_pack="native")
    assert g._dp_view is not v0
    c = PackedBatch([g], "cpu", upload=False, host_pack="numpy")
    assert np.array_equal(_native_arrays(b)["mask_rotate"], c._host["mask_rotate"].numpy().reshape(-1))
    assert not np.array_equal(_native_arrays(a)["mask_rotate"], _native_arrays(b)["mask_rotate"])
