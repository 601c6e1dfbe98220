"""Deterministic synthetic ligand / pharmacophore complexes with the exact input layout the hot path consumes.

There is no RDKit here and no dataset, so BASELINE configs 2-5 (and config 1's shape) run on synthetic graphs that
reproduce what the reference's out-of-scope preprocessing emits (SURVEY.md §8a "Inputs", §8d):

  ligand  : src/datasets/process_mols.py:162-179 (16 categorical features), :255-284 (bond graph, both directions
            adjacent, one-hot bond type), :376-417 (phorefp [N,11], norm [N,33], norm_angle1/2 [N,11]),
            src/utils/torsion.py:13-61 (edge_mask / mask_rotate: the smaller side of a bridge rotates, it must have >1 atom,
            flagged direction (u,v) has u outside and v inside the rotating set)
  phore   : src/datasets/process_pharmacophore.py:634-748 (x = [type, is-EX idx, has-norm idx, alpha, weight], unit normals or 0,
            feature-feature fully connected, EX -> everything within 5 A with nearest-neighbour fallback)
  complex : src/datasets/pdbbind_phore.py:1143-1188 (phoretype one-hot, everything centred on the phore centroid)
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import networkx as nx
import numpy as np
import torch

from .hetero import HeteroData

LIG_FEATURE_DIMS = [119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2]   # process_mols.py:162-179
PHORE_FEATURE_DIMS = [11, 2, 2]                                             # process_pharmacophore.py:34-38
NUM_PHORETYPE = 11
EX_TYPE = 10

BASE_SEED = 20250321


def transformation_mask(n_atoms: int, edge_index: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Restatement of `get_transformation_mask` (src/utils/torsion.py:13-61) on a plain bond list."""
    edges = edge_index.T
    G = nx.Graph()
    G.add_nodes_from(range(n_atoms))
    G.add_edges_from([tuple(map(int, e)) for e in edges])
    to_rotate: List[List[int]] = []
    for i in range(0, edges.shape[0], 2):
        assert edges[i, 0] == edges[i + 1, 1]
        G2 = G.copy()
        G2.remove_edge(int(edges[i, 0]), int(edges[i, 1]))
        if not nx.is_connected(G2):
            comp = list(sorted(nx.connected_components(G2), key=len)[0])
            if len(comp) > 1:
                if int(edges[i, 0]) in comp:
                    to_rotate += [[], comp]
                else:
                    to_rotate += [comp, []]
                continue
        to_rotate += [[], []]
    mask_edges = np.asarray([len(c) > 0 for c in to_rotate], dtype=bool)
    mask_rotate = np.zeros((int(mask_edges.sum()), n_atoms), dtype=bool)
    r = 0
    for i, c in enumerate(to_rotate):
        if mask_edges[i]:
            mask_rotate[r, np.asarray(c, dtype=int)] = True
            r += 1
    return mask_edges, mask_rotate


def make_ligand(rng: np.random.Generator, n_atoms: int, n_tor: int) -> dict:
    """Random tree + a few ring closures; exactly `n_tor` rotatable bonds when the tree allows it.  A torsion-dense request (more than one
    rotatable bond per three atoms: the 256-atom / 128-torsion kernel-limit pose of the tests) grows long chains and closes no rings -
    every other (n_atoms, n_tor) keeps the generator it always had."""
    dense = 3 * n_tor > n_atoms
    for _attempt in range(200):
        parent, pos, deg = [-1], [np.zeros(3)], [0]
        ok = True
        for k in range(1, n_atoms):
            for _try in range(2000):
                # bias toward recent atoms so chains (and therefore rotatable bridges) are common
                p = int(rng.integers(max(0, k - (2 if dense else 6)), k)) if rng.random() < (0.9 if dense else 0.7) else int(rng.integers(0, k))
                if deg[p] >= 4:
                    continue
                v = rng.normal(size=3)
                v /= np.linalg.norm(v)
                cand = pos[p] + 1.5 * v
                if all(np.linalg.norm(cand - q) > 1.2 for q in pos):
                    break
            else:
                ok = False
                break
            parent.append(p)
            deg[p] += 1
            deg.append(1)
            pos.append(cand)
        if not ok:
            continue
        pos_a = np.asarray(pos)
        und = [(parent[k], k) for k in range(1, n_atoms)]
        have = set(und)
        n_ring = 0 if dense else n_atoms // 10
        D = np.linalg.norm(pos_a[:, None] - pos_a[None], axis=-1)
        cand = [(a, b) for a in range(n_atoms) for b in range(a + 1, n_atoms)
                if 2.0 <= D[a, b] <= 3.0 and (a, b) not in have and (b, a) not in have and deg[a] < 4 and deg[b] < 4]
        rng.shuffle(cand)
        rings = []
        for (a, b) in cand:
            if len(rings) >= n_ring:
                break
            if deg[a] < 4 and deg[b] < 4:
                rings.append((a, b))
                deg[a] += 1
                deg[b] += 1
        bonds = und + rings
        edges = []
        for (a, b) in bonds:
            edges += [(a, b), (b, a)]
        edge_index = np.asarray(edges, dtype=np.int64).T
        mask_edges, mask_rotate = transformation_mask(n_atoms, edge_index)
        T_all = int(mask_edges.sum())
        if T_all < n_tor:
            continue
        if T_all > n_tor:
            keep = np.sort(rng.choice(T_all, size=n_tor, replace=False))
            idx = np.nonzero(mask_edges)[0]
            mask_edges = np.zeros_like(mask_edges)
            mask_edges[idx[keep]] = True
            mask_rotate = mask_rotate[keep]
        break
    else:
        raise RuntimeError(f"could not build a ligand with {n_atoms} atoms / {n_tor} torsions")

    x = np.stack([rng.integers(0, d, size=n_atoms) for d in LIG_FEATURE_DIMS], axis=1).astype(np.int64)
    x[:, 0] = rng.choice([5, 6, 7, 15], size=n_atoms)
    btype = rng.choice(4, size=len(bonds), p=[0.7, 0.15, 0.05, 0.10])
    edge_attr = np.zeros((2 * len(bonds), 4), dtype=np.float32)
    edge_attr[np.arange(2 * len(bonds)), np.repeat(btype, 2)] = 1.0
    phorefp = (rng.random((n_atoms, NUM_PHORETYPE)) < 0.15).astype(np.float32)
    phorefp[:, EX_TYPE] = 0.0
    nv = rng.normal(size=(n_atoms, NUM_PHORETYPE, 3))
    nv /= np.linalg.norm(nv, axis=-1, keepdims=True)
    norm = (nv * phorefp[..., None]).astype(np.float32).reshape(n_atoms, 33)
    ang_choices = np.asarray([0.0, np.pi / 3, np.pi], dtype=np.float32)
    a1 = ang_choices[rng.integers(0, 3, size=(n_atoms, NUM_PHORETYPE))] * phorefp
    a2 = ang_choices[rng.integers(0, 3, size=(n_atoms, NUM_PHORETYPE))] * phorefp
    return dict(x=x, pos=pos_a.astype(np.float32), edge_index=edge_index, edge_attr=edge_attr,
                edge_mask=mask_edges, mask_rotate=mask_rotate, phorefp=phorefp, norm=norm,
                norm_angle1=a1.astype(np.float32), norm_angle2=a2.astype(np.float32))


def make_phore(rng: np.random.Generator, n_nodes: int, lig_pos: np.ndarray, n_feat: int = 6) -> dict:
    n_feat = min(n_feat, n_nodes)
    r = 6.0 * rng.random(n_feat) ** (1 / 3)
    d = rng.normal(size=(n_feat, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    fpos = lig_pos.mean(0) + d * r[:, None]
    ftype = rng.integers(0, 10, size=n_feat)
    fhas = rng.random(n_feat) < 0.7
    fn = rng.normal(size=(n_feat, 3))
    fn /= np.linalg.norm(fn, axis=-1, keepdims=True)
    fn *= fhas[:, None]
    ex = []
    n_ex = n_nodes - n_feat
    tries = 0
    lo_lig = 3.0
    while len(ex) < n_ex:
        tries += 1
        if tries > 20000:   # dense small ligands: relax the clearance rather than fail
            lo_lig *= 0.9
            tries = 0
        a = lig_pos[int(rng.integers(0, lig_pos.shape[0]))]
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        c = a + v * rng.uniform(3.0, 5.0)
        if np.min(np.linalg.norm(lig_pos - c, axis=-1)) < lo_lig:
            continue
        if ex and np.min(np.linalg.norm(np.asarray(ex) - c, axis=-1)) < 0.8:
            continue
        ex.append(c)
    pos = np.concatenate([fpos, np.asarray(ex).reshape(-1, 3)], axis=0)
    types = np.concatenate([ftype, np.full(n_ex, EX_TYPE)])
    has_norm = np.concatenate([fhas, np.zeros(n_ex, dtype=bool)])
    norm = np.concatenate([fn, np.zeros((n_ex, 3))], axis=0)
    alpha = np.concatenate([rng.choice([1.0, 0.7], size=n_feat), np.full(n_ex, 0.837)])
    weight = np.concatenate([rng.uniform(0.5, 1.5, size=n_feat), np.full(n_ex, 0.5)])
    x = np.stack([types, np.where(types == EX_TYPE, 0, 1), np.where(has_norm, 0, 1), alpha, weight], axis=1)
    D = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    src, dst = [], []
    for i in range(n_nodes):
        if i < n_feat:
            dd = [j for j in range(n_feat) if j != i]
        else:
            dd = [int(j) for j in np.where(D[i] < 5.0)[0] if j != i]
        if len(dd) == 0:
            dd = [int(np.argsort(D[i])[1])]
        src += [i] * len(dd)
        dst += dd
    phoretype = np.zeros((n_nodes, NUM_PHORETYPE), dtype=np.float32)
    phoretype[np.arange(n_nodes), types] = 1.0
    return dict(x=x.astype(np.float32), pos=pos.astype(np.float32), norm=norm.astype(np.float32),
                phoretype=phoretype, edge_index=np.asarray([src, dst], dtype=np.int64))


def make_complex(lig: dict, phore: dict, name: str = "syn") -> HeteroData:
    g = HeteroData()
    g["name"] = name
    center = phore["pos"].mean(0, keepdims=True)
    L, P = g["ligand"], g["phore"]
    L.x = torch.from_numpy(lig["x"])
    L.pos = torch.from_numpy(lig["pos"] - center).float()
    L.edge_mask = torch.from_numpy(lig["edge_mask"])
    L.mask_rotate = lig["mask_rotate"]
    L.phorefp = torch.from_numpy(lig["phorefp"])
    L.norm = torch.from_numpy(lig["norm"])
    L.norm_angle1 = torch.from_numpy(lig["norm_angle1"])
    L.norm_angle2 = torch.from_numpy(lig["norm_angle2"])
    g["ligand", "ligand"].edge_index = torch.from_numpy(lig["edge_index"])
    g["ligand", "ligand"].edge_attr = torch.from_numpy(lig["edge_attr"])
    P.x = torch.from_numpy(phore["x"])
    P.pos = torch.from_numpy(phore["pos"] - center).float()
    P.norm = torch.from_numpy(phore["norm"])
    P.phoretype = torch.from_numpy(phore["phoretype"])
    g["phore", "phore"].edge_index = torch.from_numpy(phore["edge_index"])
    g["original_center"] = torch.from_numpy(center).float()
    return g


def synthetic_pairs(cfg_id: int, n_pairs: int, n_atoms, n_tor, n_phore, seed: Optional[int] = None,
                    shared_phore: bool = False, shared_ligand: bool = False) -> List[HeteroData]:
    """`n_pairs` distinct ligand-pharmacophore complexes.  n_atoms/n_tor/n_phore: int or (lo, hi) inclusive ranges."""
    rng = np.random.Generator(np.random.PCG64(BASE_SEED + cfg_id if seed is None else seed))

    def draw(v):
        return int(v) if np.isscalar(v) else int(rng.integers(v[0], v[1] + 1))

    out = []
    lig0 = ph0 = None
    for i in range(n_pairs):
        if shared_ligand and lig0 is not None:
            lig = lig0
        else:
            lig = make_ligand(rng, draw(n_atoms), draw(n_tor))
            lig0 = lig0 or lig
        if shared_phore and ph0 is not None:
            ph = ph0
        else:
            ph = make_phore(rng, draw(n_phore), lig["pos"])
            ph0 = ph0 or ph
        out.append(make_complex(lig, ph, name=f"cfg{cfg_id}_{i}"))
    return out


def example_complex(sdf_path: str, phore_path: str, seed: int = BASE_SEED + 1, name: str = "sQC_Substrate__STK936575") -> HeteroData:
    """BASELINE configs[0] on the reference's REAL example inputs: geometry, bond list and bond orders of the ligand from its SDF
    (`diffphore_amd.sdf_io`, explicit hydrogens removed as `read_mol(..., remove_hs=True)` does, process_mols.py:823), the pharmacophore
    graph from its `.phore` file (`diffphore_amd.phore_io`, pinned against the reference's own parser), rotatable-bond masks from the
    real bond graph (`transformation_mask`, torsion.py:13-61).  What RDKit would compute - the 16 categorical atom features beyond the
    element, pharmacophore fingerprints, normals - is out of scope and drawn like the synthetic ligands (seeded)."""
    from . import phore_io, sdf_io
    rng = np.random.Generator(np.random.PCG64(seed))
    mol = sdf_io.remove_hs(sdf_io.read_mol(sdf_path))
    n = mol.num_atoms
    pt = {"H": 1, "B": 5, "C": 6, "N": 7, "O": 8, "F": 9, "Si": 14, "P": 15, "S": 16, "Cl": 17, "Br": 35, "I": 53}
    x = np.stack([rng.integers(0, d, size=n) for d in LIG_FEATURE_DIMS], axis=1).astype(np.int64)
    x[:, 0] = [pt.get(sy, 6) - 1 for sy in mol.symbols]          # allowable atomic_num list is 1..118 (process_mols.py:33): index = Z - 1
    edges, btype = [], []
    for b in mol.bonds:
        a1, a2, order = int(b[0]) - 1, int(b[1]) - 1, int(b[2])
        edges += [(a1, a2), (a2, a1)]
        btype += [{1: 0, 2: 1, 3: 2, 4: 3}.get(order, 0)] * 2   # single / double / triple / aromatic (process_mols.py:255-284)
    edge_index = np.asarray(edges, dtype=np.int64).T
    edge_attr = np.zeros((len(edges), 4), dtype=np.float32)
    edge_attr[np.arange(len(edges)), btype] = 1.0
    mask_edges, mask_rotate = transformation_mask(n, edge_index)
    phorefp = (rng.random((n, NUM_PHORETYPE)) < 0.15).astype(np.float32)
    phorefp[:, EX_TYPE] = 0.0
    nv = rng.normal(size=(n, NUM_PHORETYPE, 3))
    nv /= np.linalg.norm(nv, axis=-1, keepdims=True)
    ang = np.asarray([0.0, np.pi / 3, np.pi], dtype=np.float32)
    lig = dict(x=x, pos=np.asarray(mol.coords, dtype=np.float32), edge_index=edge_index, edge_attr=edge_attr, edge_mask=mask_edges,
               mask_rotate=mask_rotate, phorefp=phorefp, norm=(nv * phorefp[..., None]).astype(np.float32).reshape(n, 33),
               norm_angle1=(ang[rng.integers(0, 3, size=(n, NUM_PHORETYPE))] * phorefp).astype(np.float32),
               norm_angle2=(ang[rng.integers(0, 3, size=(n, NUM_PHORETYPE))] * phorefp).astype(np.float32))
    (ph,) = phore_io.parse_phore(phore_path)
    return make_complex(lig, phore_io.phore_graph(ph), name=name)


class ScreenLibrary:
    """A lazily generated synthetic screening library with the shapes of BASELINE configs[3] (cfg_id 4: ONE 79-node pharmacophore x
    ligands of 15-45 atoms) or configs[4] (cfg_id 5: pharmacophores of 12-80 nodes x ONE ligand).  Unit i is a pure function of
    (seed, i): a rank can size the whole library (`sizes`, `cost`: a few integers per unit, no graph is built), shard it, and then build
    only the units it owns, one at a time - the reference walks its dataset the same way (`for idx, graph in enumerate(loader)`,
    src/inference.py:139), here without ever holding more than the units of two device batches."""

    def __init__(self, cfg_id: int, n_units: int, seed: Optional[int] = None):
        if cfg_id not in (4, 5):
            raise ValueError("screening libraries exist for configs 4 (virtual screen) and 5 (target fishing)")
        self.cfg_id, self.n_units = cfg_id, int(n_units)
        self.seed = BASE_SEED + cfg_id if seed is None else int(seed)
        self.shape = CONFIGS[cfg_id]
        rng = np.random.Generator(np.random.PCG64([self.seed, 0x5EED]))
        na, nt, npn = self.shape["n_atoms"], self.shape["n_tor"], self.shape["n_phore"]
        # the replicated side: the pharmacophore (built around a template ligand) or the ligand
        tmpl = make_ligand(rng, 30 if not np.isscalar(na) else int(na), 8 if not np.isscalar(nt) else int(nt))
        self._shared_lig = tmpl if cfg_id == 5 else None
        self._shared_ph = make_phore(rng, int(npn), tmpl["pos"]) if cfg_id == 4 else None
        self._tmpl_pos = tmpl["pos"]

    def __len__(self):
        return self.n_units

    def _rng(self, i: int) -> np.random.Generator:
        return np.random.Generator(np.random.PCG64([self.seed, int(i)]))

    def _draw_sizes(self, rng):
        def draw(v):
            return int(v) if np.isscalar(v) else int(rng.integers(v[0], v[1] + 1))
        nl, nt, npn = draw(self.shape["n_atoms"]), draw(self.shape["n_tor"]), draw(self.shape["n_phore"])
        return nl, min(nt, nl // 3), npn            # a small random tree cannot carry many rotatable bonds

    def sizes(self, i: int) -> Tuple[int, int, int]:
        """(atoms, requested torsions, pharmacophore nodes) of unit i"""
        return self._draw_sizes(self._rng(i))

    def cost(self, i: int) -> int:
        """the same estimate as `distributed.pose_cost`, from the sizes alone"""
        nl, _, npn = self.sizes(i)
        return nl * npn + 25 * nl + 8 * npn

    def unit(self, i: int) -> HeteroData:
        rng = self._rng(i)
        nl, nt, npn = self._draw_sizes(rng)
        lig = self._shared_lig if self._shared_lig is not None else make_ligand(rng, nl, nt)
        ph = self._shared_ph if self._shared_ph is not None else make_phore(rng, npn, self._tmpl_pos)
        return make_complex(lig, ph, name=f"lib{self.cfg_id}_{i}")


# BASELINE.json configs -> shapes
CONFIGS = {
    1: dict(n_atoms=20, n_tor=5, n_phore=79, samples=40, steps=20),     # shape of STK936575 x sQC phore example
    2: dict(n_atoms=30, n_tor=8, n_phore=32, pairs=512, steps=20),
    3: dict(n_atoms=80, n_tor=25, n_phore=32, pairs=256, steps=20),
    4: dict(n_atoms=(15, 45), n_tor=(2, 12), n_phore=79, samples=40, steps=20),
    5: dict(n_atoms=30, n_tor=8, n_phore=(12, 80), samples=40, steps=100),
}
