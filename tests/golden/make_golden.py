#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's own utilities (build container only).

Run from the repo root in the build container (where /root/reference exists):

    python tests/golden/make_golden.py            # fast fixtures (seconds)
    python tests/golden/make_golden.py --tables   # + SO(3)/torus score-norm tables (~10 min)

Outputs (data only: inputs and expected outputs; never reference source text):
    tests/golden/ref_utils.npz                 geometry / torsion / diffusion_utils known answers
    diffphore_amd/data/so3_exp_score_norms.npy reference `utils/so3.py:_exp_score_norms`  (1000 f64, deterministic)
    diffphore_amd/data/torus_score_norm.npy    reference `utils/torus.py:score_norm_`     (5001 f64, ONE Monte-Carlo draw,
                                               np.random.seed(20250321) set before the import; canonical for oracle+GPU)

The score model (`models/score_model_phore.py`) and `utils/sampling.py` need e3nn, torch_scatter, torch_cluster, torch_geometric
and rdkit, which are absent: their golden vectors come from `make_model_golden.py`, which runs the reference's
model / sampler code with those packages adapted onto the oracle's restatements (and says what that does and does not pin).

`torch_geometric` is stubbed in sys.modules purely so `utils/torsion.py` and `utils/diffusion_utils.py` import; none
of the stubbed symbols is called by the functions exercised here.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _stub_modules():
    tg = types.ModuleType("torch_geometric")
    tgu = types.ModuleType("torch_geometric.utils")
    tgd = types.ModuleType("torch_geometric.data")
    tgu.to_networkx = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))

    class Data:  # noqa: D401 - stub
        pass

    tgd.Data = Data
    tg.utils, tg.data = tgu, tgd
    sys.modules.update({"torch_geometric": tg, "torch_geometric.utils": tgu, "torch_geometric.data": tgd})


class _Store(dict):
    """Attribute+item store standing in for a PyG node/edge store."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Graph:
    """Duck-typed HeteroData: g['ligand'], g['ligand','ligand']."""

    def __init__(self):
        self._s = {}

    def __getitem__(self, k):
        return self._s.setdefault(k, _Store())


def make_tree(rng, n, n_tor):
    """Random tree ligand (reference-independent toy): returns pos, directed bond list, rotatable bonds, mask_rotate."""
    import networkx as nx

    parent = [-1]
    pos = [np.zeros(3)]
    deg = [0]
    for k in range(1, n):
        while True:
            p = int(rng.integers(0, k))
            if deg[p] >= 4:
                continue
            v = rng.normal(size=3)
            v /= np.linalg.norm(v)
            cand = pos[p] + 1.5 * v
            if all(np.linalg.norm(cand - q) > 1.2 for q in pos):
                break
        parent.append(p)
        deg[p] += 1
        deg.append(1)
        pos.append(cand)
    pos = np.asarray(pos, dtype=np.float32)
    und = [(parent[k], k) for k in range(1, n)]
    G = nx.Graph(und)
    cands = []
    for (a, b) in und:
        G2 = G.copy()
        G2.remove_edge(a, b)
        comps = sorted(nx.connected_components(G2), key=len)
        small = comps[0]
        if len(small) > 1:
            cands.append((a, b, small))
    rng.shuffle(cands)
    cands = cands[:n_tor]
    rot_keys = {}
    for (a, b, small) in cands:
        # flagged direction (u, v): u NOT in the rotating set, v in it   (reference utils/torsion.py:94-95)
        u, v = (a, b) if b in small else (b, a)
        rot_keys[(u, v)] = small
    edges = []
    for (a, b) in und:
        edges += [(a, b), (b, a)]
    edge_index = np.asarray(edges, dtype=np.int64).T
    edge_mask = np.asarray([(int(a), int(b)) in rot_keys for a, b in edges], dtype=bool)
    mask_rotate = np.zeros((int(edge_mask.sum()), n), dtype=bool)
    r = 0
    for (a, b) in edges:
        if (a, b) in rot_keys:
            mask_rotate[r, list(rot_keys[(a, b)])] = True
            r += 1
    return pos, edge_index, edge_mask, mask_rotate


def fast_fixtures():
    _stub_modules()
    sys.path.insert(0, REF)
    from utils import geometry, torsion, diffusion_utils  # noqa: E402  (reference imports)

    out = {}
    g = torch.Generator().manual_seed(1234)
    rng = np.random.default_rng(4321)

    # (1) axis_angle_to_matrix incl. the |v|<1e-6 branch                     geometry.py:54-85
    rv = torch.randn(16, 3, generator=g)
    rv[0] = torch.tensor([1e-8, -2e-8, 3e-9])
    rv[1] = torch.zeros(3)
    rv[2] = torch.tensor([3.0, 0.1, -0.2])
    out["aa_in"] = rv.numpy()
    out["aa_out"] = geometry.axis_angle_to_matrix(rv).numpy()

    # (2) Kabsch incl. a reflection case                                       geometry.py:88-136
    A = torch.randn(4, 3, 12, generator=g)
    Rs = geometry.axis_angle_to_matrix(torch.randn(4, 3, generator=g))
    B = torch.einsum("bij,bjn->bin", Rs, A) + torch.randn(4, 3, 1, generator=g) + 0.01 * torch.randn(4, 3, 12, generator=g)
    # reflection case: planar-ish point cloud mirrored
    A[3, 2] *= 1e-3
    B[3] = A[3] * torch.tensor([1.0, 1.0, -1.0]).view(3, 1)
    kr, kt = [], []
    for i in range(4):
        R, t = geometry.rigid_transform_Kabsch_3D_torch(A[i], B[i])
        kr.append(R.numpy())
        kt.append(t.numpy())
    out["kabsch_A"], out["kabsch_B"] = A.numpy(), B.numpy()
    out["kabsch_R"], out["kabsch_t"] = np.stack(kr), np.stack(kt)

    # (3,4) torsion update and modify_conformer on toy trees incl. the norm-slot quirk
    for case, (n, T) in enumerate([(9, 3), (14, 5), (23, 7)]):
        pos, edge_index, edge_mask, mask_rotate = make_tree(rng, n, T)
        Treal = int(edge_mask.sum())
        norm = rng.normal(size=(n, 33)).astype(np.float32)
        norm[rng.random(size=(n, 11)).repeat(3, axis=1) < 0.6] = 0.0
        theta = rng.uniform(-np.pi, np.pi, size=Treal).astype(np.float32)
        if Treal > 1:
            theta[1] = 0.0  # the `== 0: continue` branch  (torsion.py:88-89)
        tpos = torch.from_numpy(pos)
        tnorm_pts = torch.from_numpy(norm).reshape(-1, n, 3) + tpos.unsqueeze(0)
        rot_edges = torch.from_numpy(edge_index).T[torch.from_numpy(edge_mask)]
        p2, n2 = torsion.modify_conformer_torsion_angles(tpos, rot_edges, mask_rotate, theta, norm=tnorm_pts)
        pre = f"tor{case}_"
        out[pre + "pos"], out[pre + "edge_index"], out[pre + "edge_mask"] = pos, edge_index, edge_mask
        out[pre + "mask_rotate"], out[pre + "norm"], out[pre + "theta"] = mask_rotate, norm, theta
        out[pre + "pos_out"], out[pre + "normpts_out"] = p2.numpy(), n2.numpy()

        data = _Graph()
        data["ligand"].pos = tpos.clone()
        data["ligand"].x = torch.zeros(n, 16)
        data["ligand"].norm = torch.from_numpy(norm).clone()
        data["ligand"].edge_mask = torch.from_numpy(edge_mask)
        data["ligand"].mask_rotate = mask_rotate
        data["ligand", "ligand"].edge_index = torch.from_numpy(edge_index)
        tr = torch.randn(1, 3, generator=g)
        rot = torch.randn(3, generator=g) * 0.4
        diffusion_utils.modify_conformer(data, tr, rot, theta)
        out[pre + "mc_tr"], out[pre + "mc_rot"] = tr.numpy(), rot.numpy()
        out[pre + "mc_pos_out"] = data["ligand"].pos.numpy()
        out[pre + "mc_norm_out"] = data["ligand"].norm.numpy()
        # rigid-only branch (torsion_updates=None)
        data2 = _Graph()
        data2["ligand"].pos = tpos.clone()
        data2["ligand"].x = torch.zeros(n, 16)
        data2["ligand"].norm = torch.from_numpy(norm).clone()
        diffusion_utils.modify_conformer(data2, tr, rot, None)
        out[pre + "rigid_pos_out"] = data2["ligand"].pos.numpy()
        out[pre + "rigid_norm_out"] = data2["ligand"].norm.numpy()

    # (5) schedule / sigma / time embedding                               diffusion_utils.py:16-20,82-93,135-145
    args = types.SimpleNamespace(tr_sigma_min=0.1, tr_sigma_max=5.0, rot_sigma_min=0.1, rot_sigma_max=1.5,
                                 tor_sigma_min=0.0314, tor_sigma_max=3.14)
    sched = diffusion_utils.get_t_schedule(20)
    out["t_schedule20"] = np.asarray(sched)
    out["t_schedule7"] = np.asarray(diffusion_utils.get_t_schedule(7))
    sig = np.asarray([diffusion_utils.t_to_sigma(t, t, t, args) for t in sched], dtype=np.float64)
    out["sigma20"] = sig
    emb = diffusion_utils.get_timestep_embedding("sinusoidal", 20, 10000)
    tt = torch.tensor(np.asarray(sched), dtype=torch.float32)
    out["sin_emb20"] = emb(tt).numpy()
    out["sin_emb_t"] = tt.numpy()

    np.savez_compressed(os.path.join(HERE, "ref_utils.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_utils.npz"), len(out), "arrays")


def tables():
    _stub_modules()
    sys.path.insert(0, REF)
    real_save = np.save
    np.save = lambda *a, **k: None  # reference tries to cache into a read-only, non-existent dir
    try:
        np.random.seed(20250321)
        from utils import so3  # noqa: E402   ~5 min
        real_save(os.path.join(ROOT, "diffphore_amd", "data", "so3_exp_score_norms.npy"),
                  np.asarray(so3._exp_score_norms, dtype=np.float64))
        print("so3 table:", so3._exp_score_norms[[0, 500, 999]])
        np.random.seed(20250321)
        from utils import torus  # noqa: E402  ~4 min
        real_save(os.path.join(ROOT, "diffphore_amd", "data", "torus_score_norm.npy"),
                  np.asarray(torus.score_norm_, dtype=np.float64))
        print("torus table:", torus.score_norm(np.asarray([0.0314, 0.5, 3.14])))
        # known-answer probes of the lookup functions themselves
        probes = {
            "so3_eps": np.asarray([0.1, 0.3, 0.5, 1.0, 1.5, 0.005, 3.0], dtype=np.float32),
            "torus_sigma": np.asarray([0.0314, 0.1, 0.5, 1.0, 3.14, 1e-3, 10.0], dtype=np.float32),
        }
        probes["so3_score_norm"] = so3.score_norm(torch.from_numpy(probes["so3_eps"])).numpy()
        probes["torus_score_norm"] = torus.score_norm(probes["torus_sigma"])
        np.savez_compressed(os.path.join(HERE, "ref_tables_probe.npz"), **probes)
    finally:
        np.save = real_save


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", action="store_true")
    ap.add_argument("--only-tables", action="store_true")
    a = ap.parse_args()
    if not a.only_tables:
        fast_fixtures()
    if a.tables or a.only_tables:
        tables()
