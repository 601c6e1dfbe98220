"""`.phore` pharmacophore files -> the 'phore' half of the hot path's input graph  (first "next" row).

Restates /root/reference/src/datasets/process_pharmacophore.py:
    parse_phore / parse_phore_line   :78-152, :751-789    tab-separated records, "$$$$" terminates a pharmacophore
    get_phore_graph                  :634-714             nodes = features + exclusion spheres; feature<->feature fully connected,
                                                          exclusion sphere -> every node within `neighbor_cutoff`, nearest-neighbour fallback
    phore_featurizer                 :717-748             x = [type idx, is-EX idx, has-norm idx, alpha, weight]
and the one-hot `phoretype` / centring of pdbbind_phore.py:1172-1184.  Pure text + numpy (no RDKit); the ligand half of
the graph still needs the reference's RDKit featuriser.  Parity: pinned - `tests/golden/make_phore_golden.py` runs the reference's
own three functions on the shipped example and two generated records; `tests/test_phore_io_cpu.py` compares features, coordinates,
normals and the ordered edge list.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

PHORE_TYPES = ["MB", "HD", "AR", "PO", "HA", "HY", "NE", "CV", "CR", "XB", "EX"]     # process_pharmacophore.py:34


@dataclass
class PhoreFeature:
    type: str
    alpha: float
    weight: float
    factor: float
    coordinate: Tuple[float, float, float]
    has_norm: bool
    norm: Tuple[float, float, float]
    label: str
    anchor_weight: float


@dataclass
class Phore:
    id: str
    features: List[PhoreFeature] = field(default_factory=list)
    exclusion_volumes: List[PhoreFeature] = field(default_factory=list)


def parse_phore_line(record: str, cvs: bool = False):
    """None for the "$$$$" terminator; raises SyntaxError on a malformed line (reference: skip_wrong_lines=False)."""
    if record == "$$$$":
        return None
    try:
        ptype, alpha, weight, factor, x, y, z, has_norm, nx, ny, nz, label, anchor = record.split("\t")
        return PhoreFeature(ptype if cvs else ptype[:2], float(alpha), float(weight), float(factor),
                            (float(x), float(y), float(z)), bool(int(has_norm)), (float(nx), float(ny), float(nz)), label,
                            float(anchor))
    except Exception as e:  # noqa: BLE001
        raise SyntaxError(f"Invalid phore feature syntax: {record!r}") from e


def parse_phore_text(text: str, skip_ex: bool = False, skip_wrong_lines: bool = True) -> List[Phore]:
    phores: List[Phore] = []
    cur = None
    correct = True
    for raw in text.splitlines():
        record = raw.strip()
        if not record:
            break                                   # the reference stops at the first empty line (:113-140)
        if cur is None:
            cur = Phore(record)
            correct = True
            continue
        try:
            feat = parse_phore_line(record) if correct else False
        except SyntaxError:
            if not skip_wrong_lines:
                raise
            feat = False
        if feat is None:
            if cur.features and correct:
                phores.append(cur)
            cur = None
        elif feat is False:
            correct = False
        elif feat.type != "EX":
            cur.features.append(feat)
        elif not skip_ex:
            cur.exclusion_volumes.append(feat)
    return phores


def parse_phore(phore_file: str, **kw) -> List[Phore]:
    with open(phore_file, "r") as f:
        return parse_phore_text(f.read(), **kw)


def _safe_index(lst, e):
    return lst.index(e) if e in lst else len(lst) - 1


def phore_graph(phore: Phore, consider_ex: bool = True, neighbor_cutoff: float = 5.0, ex_connected: bool = True) -> dict:
    """Arrays in the layout `diffphore_amd.synthetic.make_complex` consumes (x, pos, norm, phoretype, edge_index), un-centred."""
    ex = phore.exclusion_volumes if consider_ex else []
    feats = list(phore.features) + list(ex)
    n, n_feat = len(feats), len(phore.features)
    pos = np.asarray([f.coordinate for f in feats], dtype=np.float64).reshape(n, 3)
    norm = np.asarray([[f.norm[k] - f.coordinate[k] for k in range(3)] if f.has_norm else [0.0, 0.0, 0.0] for f in feats])
    ln = np.linalg.norm(norm, axis=1)
    ln[ln == 0] = 1
    norm = norm / ln.reshape(-1, 1)
    D = np.linalg.norm(pos[:, None, :] - pos[None, :, :], axis=-1)
    src, dst = [], []
    for i in range(n):
        if i < n_feat:
            dd = [j for j in range(n_feat) if j != i]
        else:
            dd = [int(j) for j in np.where(D[i] < neighbor_cutoff)[0] if j != i]
            if not ex_connected:
                dd = [j for j in dd if j >= n_feat]
        if not dd:
            dd = [int(np.argsort(D[i])[1])]
        src += [i] * len(dd)
        dst += dd
    types = np.asarray([_safe_index(PHORE_TYPES, f.type) for f in feats])
    x = np.asarray([[t, _safe_index([True, False], f.type == "EX"), _safe_index([True, False], f.has_norm), f.alpha, f.weight]
                    for t, f in zip(types, feats)], dtype=np.float32)
    phoretype = np.zeros((n, len(PHORE_TYPES)), dtype=np.float32)
    phoretype[np.arange(n), types] = 1.0
    return dict(x=x, pos=pos.astype(np.float32), norm=norm.astype(np.float32), phoretype=phoretype,
                edge_index=np.asarray([src, dst], dtype=np.int64))
