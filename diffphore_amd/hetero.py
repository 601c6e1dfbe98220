"""Duck-typed stand-in for the torch_geometric HeteroData / Batch objects the reference hot path consumes.

torch_geometric is not available on the GPU box, so the drop-in accepts any object with this access pattern
(a real PyG HeteroDataBatch satisfies it too):

    data['ligand'].pos / .x / .batch / .edge_mask / .mask_rotate / .phorefp / .norm / .norm_angle1 / .norm_angle2
    data['ligand', 'ligand'].edge_index / .edge_attr          (PyG resolves the partial key ('ligand','lig_bond','ligand'))
    data['phore'].pos / .x / .batch / .norm / .phoretype
    data['phore', 'phore'].edge_index
    data.complex_t = {'tr','rot','tor'} ; data.num_graphs ; data['ligand'].node_t ; data['phore'].node_t

Field meanings follow the reference graph builder: src/datasets/pdbbind_phore.py:1143-1188,
src/datasets/process_mols.py:255-284,376-417, src/datasets/process_pharmacophore.py:634-748.
"""
from __future__ import annotations

import copy
from typing import Dict, List

import numpy as np
import torch


class Store(dict):
    """Attribute-style node / edge store."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    @property
    def num_nodes(self):
        for k in ("pos", "x"):
            if k in self:
                return int(self[k].shape[0])
        raise AttributeError("num_nodes")


def _norm_key(k):
    if isinstance(k, tuple) and len(k) == 3:
        return (k[0], k[2])
    return k


class HeteroData:
    """Minimal hetero-graph container (one complex)."""

    def __init__(self):
        object.__setattr__(self, "_stores", {})
        object.__setattr__(self, "_attrs", {})

    def __getitem__(self, k):
        k = _norm_key(k)
        if isinstance(k, str) and k in self._attrs:
            return self._attrs[k]
        return self._stores.setdefault(k, Store())

    def __setitem__(self, k, v):
        self._attrs[k] = v

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self._attrs[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self._attrs[k] = v

    def __contains__(self, k):
        return k in self._attrs or _norm_key(k) in self._stores

    def stores(self):
        return self._stores

    def attrs(self):
        return self._attrs

    def __deepcopy__(self, memo):
        out = type(self)()
        for k, s in self._stores.items():
            out._stores[k] = Store({a: copy.deepcopy(v, memo) for a, v in s.items()})
        for k, v in self._attrs.items():
            out._attrs[k] = copy.deepcopy(v, memo)
        return out

    def sample_copy(self):
        """A new pose of the same complex: shares every static tensor with `self`, owns only the ligand's diffusion state
        (`pos`, `norm`).  What `copy.deepcopy(graph)` is used for in the reference's sampling drivers (inference.py:183), ~25x cheaper."""
        out = type(self)()
        for k, st in self._stores.items():
            out._stores[k] = Store(st)
        lig = out._stores.get("ligand")
        if lig is not None:
            for a in ("pos", "norm"):
                if a in lig and torch.is_tensor(lig[a]):
                    lig[a] = lig[a].clone()
        out._attrs.update(self._attrs)
        for a, v in vars(self).items():
            if a not in ("_stores", "_attrs"):
                object.__setattr__(out, a, v)
        return out

    def to(self, device):
        for s in self._stores.values():
            for a, v in list(s.items()):
                if torch.is_tensor(v):
                    s[a] = v.to(device)
                elif isinstance(v, dict):
                    s[a] = {kk: (vv.to(device) if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
        for a, v in list(self._attrs.items()):
            if torch.is_tensor(v):
                self._attrs[a] = v.to(device)
            elif isinstance(v, dict):
                self._attrs[a] = {kk: (vv.to(device) if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
        return self


_NODE_TYPES = ("ligand", "phore")


class Batch(HeteroData):
    """Collated batch of HeteroData (what PyG's DataLoader yields at src/utils/sampling.py:210-213)."""

    @classmethod
    def from_data_list(cls, data_list: List[HeteroData]) -> "Batch":
        b = cls()
        B = len(data_list)
        object.__setattr__(b, "_num_graphs", B)
        ptr: Dict[str, List[int]] = {}
        for nt in _NODE_TYPES:
            ptr[nt] = [0]
            for g in data_list:
                ptr[nt].append(ptr[nt][-1] + g[nt].num_nodes)
        keys = set()
        for g in data_list:
            keys.update(g.stores().keys())
        for k in keys:
            st = b[k]
            names = set()
            for g in data_list:
                names.update(g[k].keys())
            for a in names:
                vals = [g[k][a] for g in data_list]
                if a == "edge_index":
                    src_t, dst_t = (k if isinstance(k, tuple) else (k, k))
                    offs = [torch.tensor([[ptr[src_t][i]], [ptr[dst_t][i]]], dtype=torch.long) for i in range(B)]
                    st[a] = torch.cat([torch.as_tensor(v).long() + o for v, o in zip(vals, offs)], dim=1)
                elif a == "mask_rotate":
                    st[a] = [np.asarray(v) for v in vals]   # PyG keeps non-tensor attrs as a python list
                elif torch.is_tensor(vals[0]):
                    st[a] = torch.cat(vals, dim=0)
                else:
                    st[a] = vals
        for nt in _NODE_TYPES:
            b[nt].batch = torch.repeat_interleave(torch.arange(B), torch.tensor(
                [ptr[nt][i + 1] - ptr[nt][i] for i in range(B)], dtype=torch.long))
            b[nt].ptr = torch.tensor(ptr[nt], dtype=torch.long)
        names = set()
        for g in data_list:
            names.update(g.attrs().keys())
        for a in names:
            vals = [g.attrs().get(a) for g in data_list]
            b._attrs[a] = torch.cat(vals, dim=0) if torch.is_tensor(vals[0]) and vals[0].dim() > 0 else vals
        b._attrs["_slices"] = ptr
        return b

    @property
    def num_graphs(self):
        return self._num_graphs

    def to_data_list(self) -> List[HeteroData]:
        ptr = self._attrs["_slices"]
        B = self._num_graphs
        out = []
        for i in range(B):
            g = HeteroData()
            for k, st in self._stores.items():
                for a, v in st.items():
                    if a in ("batch", "ptr", "node_t", "node_sigma_emb"):
                        continue
                    if isinstance(k, tuple):
                        src_t, dst_t = k
                        ei = st["edge_index"]
                        sel = (ei[0] >= ptr[src_t][i]) & (ei[0] < ptr[src_t][i + 1])
                        if a == "edge_index":
                            off = torch.tensor([[ptr[src_t][i]], [ptr[dst_t][i]]], dtype=torch.long, device=ei.device)
                            g[k][a] = ei[:, sel] - off
                        elif torch.is_tensor(v):
                            g[k][a] = v[sel]
                    else:
                        lo, hi = ptr[k][i], ptr[k][i + 1]
                        if a == "mask_rotate":
                            g[k][a] = v[i]
                        elif a == "edge_mask":
                            # one flag per ligand bond edge: slice by this graph's bond-edge range
                            ei = self._stores[("ligand", "ligand")]["edge_index"]
                            sel = (ei[0] >= ptr["ligand"][i]) & (ei[0] < ptr["ligand"][i + 1])
                            g[k][a] = v[sel]
                        elif torch.is_tensor(v) and v.shape[0] == ptr[k][-1]:
                            g[k][a] = v[lo:hi]
                        elif isinstance(v, list) and len(v) == B:
                            g[k][a] = v[i]
            for a, v in self._attrs.items():
                if a in ("_slices", "complex_t", "graph_sigma_emb"):
                    continue
                if isinstance(v, list) and len(v) == B:
                    g._attrs[a] = v[i]
                elif torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B:
                    g._attrs[a] = v[i:i + 1]          # graph-level tensors (e.g. original_center [B,3]) come back as [1,...], like PyG
            out.append(g)
        return out
