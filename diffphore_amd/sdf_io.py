"""Minimal MOL/SDF (CTfile V2000) reader-writer without RDKit.

Only what the fitting driver needs from RDKit in the reference: read a ligand record, drop hydrogens the way
`Chem.RemoveAllHs` does for explicit H atoms (`src/inference.py:229`), and write one record per pose with replaced
coordinates, a `<name>_<marker>_<idx>` title and per-pose data items (`write_mol_with_multi_coords`,
`src/datasets/process_mols.py:888-921`).  Chemistry perception (aromaticity, valences, stereo) is out of scope: atom and bond
lines are carried through verbatim apart from coordinates and renumbering.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np


@dataclass
class Mol:
    """One V2000 record.  `atom_tail[i]` is the text of atom line i after the three coordinate fields (symbol, mass diff, ...)."""
    title: str = ""
    program: str = ""
    comment: str = ""
    coords: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    symbols: List[str] = field(default_factory=list)
    atom_tail: List[str] = field(default_factory=list)
    bonds: List[List[int]] = field(default_factory=list)        # [a1, a2, order, rest...] 1-based atom indices
    bond_tail: List[str] = field(default_factory=list)
    properties: List[str] = field(default_factory=list)          # "M  ..." lines except M  END
    data: Dict[str, str] = field(default_factory=dict)           # SDF data items
    counts_tail: str = "  0  0  0  0  0  0  0  0999 V2000"

    @property
    def num_atoms(self) -> int:
        return len(self.symbols)

    def heavy_atom_mask(self) -> np.ndarray:
        return np.asarray([s not in ("H", "D", "T") for s in self.symbols], dtype=bool)


def _int(s: str, default: int = 0) -> int:
    s = s.strip()
    return int(s) if s else default


def parse_mol_block(text: str) -> Mol:
    lines = text.replace("\r\n", "\n").split("\n")
    if len(lines) < 4:
        raise ValueError("not a MOL block: fewer than 4 lines")
    m = Mol(title=lines[0].rstrip(), program=lines[1].rstrip(), comment=lines[2].rstrip())
    counts = lines[3]
    if "V3000" in counts:
        raise NotImplementedError("V3000 CTfiles are not supported")
    na, nb = _int(counts[0:3]), _int(counts[3:6])
    m.counts_tail = counts[6:] if len(counts) > 6 else m.counts_tail
    xyz = np.zeros((na, 3), dtype=np.float64)
    for i in range(na):
        ln = lines[4 + i]
        try:
            xyz[i] = float(ln[0:10]), float(ln[10:20]), float(ln[20:30])
            sym, tail = ln[31:34].strip(), ln[30:]
        except ValueError:                     # free-format fallback
            f = ln.split()
            xyz[i] = float(f[0]), float(f[1]), float(f[2])
            sym, tail = f[3], " " + f"{f[3]:<3s}" + " 0  0  0  0  0  0  0  0  0  0  0  0"
        m.symbols.append(sym)
        m.atom_tail.append(tail)
    m.coords = xyz
    for j in range(nb):
        ln = lines[4 + na + j]
        m.bonds.append([_int(ln[0:3]), _int(ln[3:6]), _int(ln[6:9])])
        m.bond_tail.append(ln[9:])
    k = 4 + na + nb
    while k < len(lines):
        ln = lines[k]
        k += 1
        if ln.startswith("M  END"):
            break
        if ln.strip():
            m.properties.append(ln.rstrip())
    # SDF data items
    while k < len(lines):
        ln = lines[k]
        k += 1
        if ln.startswith(">"):
            lo, hi = ln.find("<"), ln.rfind(">")
            key = ln[lo + 1:hi] if 0 <= lo < hi else ln[1:].strip()
            val = []
            while k < len(lines) and lines[k].strip() != "" and not lines[k].startswith("$$$$"):
                val.append(lines[k])
                k += 1
            m.data[key] = "\n".join(val)
    return m


def read_sdf(path: str) -> List[Mol]:
    text = open(path).read().replace("\r\n", "\n")
    out = []
    for rec in text.split("$$$$\n"):
        if rec.strip():
            out.append(parse_mol_block(rec))
    return out


def read_mol(path: str) -> Mol:
    return parse_mol_block(open(path).read())


def remove_hs(mol: Mol) -> Mol:
    """Explicit hydrogens and their bonds removed, remaining atoms renumbered (cf. Chem.RemoveAllHs).  Atom-list `M` property
    lines (charges, isotopes, radicals) are renumbered; entries on removed atoms are dropped."""
    keep = mol.heavy_atom_mask()
    new_idx = np.cumsum(keep) * keep            # 1-based new index, 0 for removed
    out = copy.deepcopy(mol)
    out.coords = mol.coords[keep].copy()
    out.symbols = [s for s, k in zip(mol.symbols, keep) if k]
    out.atom_tail = [t for t, k in zip(mol.atom_tail, keep) if k]
    out.bonds, out.bond_tail = [], []
    for b, t in zip(mol.bonds, mol.bond_tail):
        if keep[b[0] - 1] and keep[b[1] - 1]:
            out.bonds.append([int(new_idx[b[0] - 1]), int(new_idx[b[1] - 1]), b[2]])
            out.bond_tail.append(t)
    props = []
    for ln in mol.properties:
        tag = ln[3:6]
        if tag in ("CHG", "ISO", "RAD"):
            f = ln[6:].split()
            pairs = [(int(f[1 + 2 * i]), int(f[2 + 2 * i])) for i in range(int(f[0]))]
            pairs = [(int(new_idx[a - 1]), v) for a, v in pairs if keep[a - 1]]
            if pairs:
                props.append(f"M  {tag}{len(pairs):3d}" + "".join(f"{a:4d}{v:4d}" for a, v in pairs))
        else:
            props.append(ln)
    out.properties = props
    return out


def mol_block(mol: Mol, coords: Optional[np.ndarray] = None, title: Optional[str] = None) -> str:
    xyz = mol.coords if coords is None else np.asarray(coords, dtype=np.float64)
    if xyz.shape != (mol.num_atoms, 3):
        raise ValueError(f"coordinates {xyz.shape} do not match {mol.num_atoms} atoms")
    lines = [mol.title if title is None else title, mol.program, mol.comment,
             f"{mol.num_atoms:3d}{len(mol.bonds):3d}{mol.counts_tail}"]
    for (x, y, z), tail in zip(xyz, mol.atom_tail):
        lines.append(f"{x:10.4f}{y:10.4f}{z:10.4f}{tail}")
    for b, tail in zip(mol.bonds, mol.bond_tail):
        lines.append(f"{b[0]:3d}{b[1]:3d}{b[2]:3d}{tail}")
    lines.extend(mol.properties)
    lines.append("M  END")
    return "\n".join(lines) + "\n"


def write_mol_with_multi_coords(mol: Mol, multi_new_coords: Sequence[np.ndarray], path: str, name: str, marker: str = "",
                                properties: Optional[Dict[str, Sequence]] = None) -> None:
    """Same signature and record layout as the reference's writer (`process_mols.py:888-921`): one record per coordinate set,
    title `<name>_<marker>_<idx>`, and for every key of `properties` the idx-th value as an SDF data item."""
    with open(path, "w") as f:
        for idx, xyz in enumerate(multi_new_coords):
            f.write(mol_block(mol, np.asarray(xyz, dtype=np.float64), title=f"{name}_{marker}_{idx}"))
            if properties is not None:
                for k, v in properties.items():
                    f.write(f">  <{k}>  ({idx + 1}) \n{v[idx]}\n\n")
            f.write("$$$$\n")
