#!/usr/bin/env python3
"""Golden vectors for the rotatable-bond masks (`get_transformation_mask`, src/utils/torsion.py:13-61) from the reference's own code.

    python tests/golden/make_mask_golden.py      ->  tests/golden/ref_transformation_mask.npz

`utils/torsion.py` is imported for real.  Its one torch_geometric dependency, `to_networkx`, is adapted here with networkx itself
(directed graph over `range(num_nodes)` with one edge per edge_index column - what PyG's helper builds for a homogeneous graph
without attributes).  Inputs: the bond lists of synthetic ligands (regenerated from seeds by the test)."""
import os
import sys
import types

import networkx as nx
import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def to_networkx(data, to_undirected=False, **kw):
    G = nx.Graph() if to_undirected else nx.DiGraph()
    G.add_nodes_from(range(int(data.num_nodes)))
    G.add_edges_from((int(a), int(b)) for a, b in data.edge_index.T.tolist())
    return G


class _Homo:
    def __init__(self, n, ei):
        self.num_nodes, self.edge_index = n, ei


class _Pyg:
    """The two accesses get_transformation_mask makes on its argument."""

    def __init__(self, n, ei):
        self.n, self.ei = n, ei

    def to_homogeneous(self):
        return _Homo(self.n, self.ei)

    def __getitem__(self, k):
        return types.SimpleNamespace(edge_index=self.ei)


def main():
    tg, tgu, tgd = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.utils"), types.ModuleType("torch_geometric.data")
    tgu.to_networkx = to_networkx
    tgd.Data = type("Data", (), {})
    tg.utils, tg.data = tgu, tgd
    sys.modules.update({"torch_geometric": tg, "torch_geometric.utils": tgu, "torch_geometric.data": tgd})
    sys.path.insert(0, REF)
    from utils.torsion import get_transformation_mask
    from diffphore_amd import synthetic
    out = {}
    specs = [(1, 12, 2), (2, 20, 5), (3, 30, 8), (4, 45, 12), (5, 80, 25), (6, 13, 0)]
    for k, (seed, n, t) in enumerate(specs):
        lig = synthetic.make_ligand(np.random.default_rng(seed), n, t)
        ei = torch.as_tensor(np.asarray(lig["edge_index"])).long()
        me, mr = get_transformation_mask(_Pyg(n, ei))
        out[f"m{k}_spec"] = np.asarray([seed, n, t])
        out[f"m{k}_edges"], out[f"m{k}_rotate"] = me, mr
        print(k, (seed, n, t), "rotatable:", int(me.sum()), mr.shape)
    out["n"] = np.int64(len(specs))
    np.savez_compressed(os.path.join(HERE, "ref_transformation_mask.npz"), **out)


if __name__ == "__main__":
    main()
