"""SURVEY.md 8f-1 host pieces that need no GPU: the RDKit-free MOL/SDF reader-writer and `analyze_results`."""
import os
import types

import numpy as np
import pytest

from diffphore_amd import fit as fitmod
from diffphore_amd import sdf_io

ETHANOL = """ethanol
  handmade

  9  8  0  0  0  0  0  0  0  0999 V2000
    0.0000    0.0000    0.0000 C   0  0  0  0  0  0  0  0  0  0  0  0
    1.5200    0.0000    0.0000 C   0  0  0  0  0  0  0  0  0  0  0  0
    2.0500    1.3000    0.0000 O   0  0  0  0  0  0  0  0  0  0  0  0
   -0.3600    1.0300    0.0000 H   0  0  0  0  0  0  0  0  0  0  0  0
   -0.3600   -0.5100    0.8900 H   0  0  0  0  0  0  0  0  0  0  0  0
   -0.3600   -0.5100   -0.8900 H   0  0  0  0  0  0  0  0  0  0  0  0
    1.8800   -0.5100    0.8900 H   0  0  0  0  0  0  0  0  0  0  0  0
    1.8800   -0.5100   -0.8900 H   0  0  0  0  0  0  0  0  0  0  0  0
    3.0100    1.2500    0.0000 H   0  0  0  0  0  0  0  0  0  0  0  0
  1  2  1  0
  2  3  1  0
  1  4  1  0
  1  5  1  0
  1  6  1  0
  2  7  1  0
  2  8  1  0
  3  9  1  0
M  CHG  2   3  -1   9   1
M  END
>  <source>  (1)
unit test

$$$$
"""


def test_parse_remove_hs_and_write(tmp_path):
    m = sdf_io.parse_mol_block(ETHANOL)
    assert m.title == "ethanol" and m.num_atoms == 9 and len(m.bonds) == 8 and m.symbols[:3] == ["C", "C", "O"]
    assert m.data == {"source": "unit test"}
    np.testing.assert_allclose(m.coords[2], [2.05, 1.3, 0.0])
    h = sdf_io.remove_hs(m)
    assert h.num_atoms == 3 and [b[:2] for b in h.bonds] == [[1, 2], [2, 3]]
    assert h.properties == ["M  CHG  1   3  -1"]                         # the charge on the removed H is dropped, O keeps its index
    poses = np.stack([h.coords + k for k in range(3)])
    out = tmp_path / "poses.sdf"
    sdf_io.write_mol_with_multi_coords(h, poses, str(out), "lig", marker="rank", properties={"fitscore": [0.9, 0.5, 0.1]})
    recs = sdf_io.read_sdf(str(out))
    assert [r.title for r in recs] == ["lig_rank_0", "lig_rank_1", "lig_rank_2"]
    assert [r.data["fitscore"] for r in recs] == ["0.9", "0.5", "0.1"]
    for k, r in enumerate(recs):
        np.testing.assert_allclose(r.coords, poses[k], atol=1e-4)
        assert r.symbols == ["C", "C", "O"] and [b[:3] for b in r.bonds] == [[1, 2, 1], [2, 3, 1]]
    # the mol block survives a parse -> write -> parse round trip verbatim
    assert sdf_io.mol_block(sdf_io.parse_mol_block(sdf_io.mol_block(m))) == sdf_io.mol_block(m)
    with pytest.raises(ValueError):
        sdf_io.mol_block(h, np.zeros((4, 3)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples/ligands"), reason="reference examples only exist in the build container")
def test_shipped_example_ligands_parse():
    d = "/root/reference/examples/ligands"
    n = 0
    for f in sorted(os.listdir(d)):
        (m,) = sdf_io.read_sdf(os.path.join(d, f))
        heavy = sdf_io.remove_hs(m)
        assert 5 < heavy.num_atoms <= m.num_atoms and all(1 <= a <= heavy.num_atoms and 1 <= b <= heavy.num_atoms for a, b, _ in heavy.bonds)
        again = sdf_io.parse_mol_block(sdf_io.mol_block(m))
        np.testing.assert_allclose(again.coords, m.coords, atol=1e-4)
        assert again.symbols == m.symbols and again.bonds == m.bonds
        n += 1
    assert n == 18
    (m,) = sdf_io.read_sdf(os.path.join(d, "STK936575.sdf"))
    assert int(sdf_io.remove_hs(m).num_atoms) == 20                    # cfg1 ligand has 20 heavy atoms, 22 heavy bonds
    assert len(sdf_io.remove_hs(m).bonds) == 22


def test_analyze_results_layout(tmp_path):
    args = types.SimpleNamespace(out_dir=str(tmp_path), cutoff=0.6)
    res = {"name": ["t1__a", "t1__b", "t2__c"], "fitscore": [[0.1, 0.7, 0.3], [0.9, 0.2, 0.8, 0.1, 0.5, 0.6], []], "run_time": [1.0, 2.0, 3.0]}
    res["fitscore"][2] = [-2.0, -2.0]
    df = fitmod.analyze_results(args, res)
    assert df["name"].tolist() == ["t1__b", "t1__a", "t2__c"]
    assert df.columns.tolist() == ["target", "ligand", "name", "run_time", "max_fitscore", "top5_mean_fitscore", "fitscore"]
    np.testing.assert_allclose(df["top5_mean_fitscore"].tolist(), [np.mean([0.9, 0.8, 0.6, 0.5, 0.2]), np.mean([0.1, 0.7, 0.3]), -2.0])
    lines = open(tmp_path / "ranked_results.csv").read().splitlines()
    assert lines[0].split("\t")[:3] == ["target", "ligand", "name"] and lines[1].split("\t")[:3] == ["t1", "b", "t1__b"]
    assert len(open(tmp_path / "ranked_results_gt0.6.csv").read().splitlines()) == 3
    assert fitmod.my_max([]) == -2.0


def test_perfect_similarity_and_surrogate_formula():
    import torch
    from diffphore_amd import synthetic
    from diffphore_amd.fitness import get_perfect_similarity
    from oracle import fitness as ofit
    (g,) = synthetic.synthetic_pairs(5, 1, 14, 2, 12)
    vol = g["phore"].phoretype.sum(0)
    g["ligand"].ph = vol.clone()                       # the ligand offers every feature the model asks for
    assert abs(get_perfect_similarity(g) - 1.0) < 1e-6
    g["ligand"].ph = torch.zeros(11)
    assert get_perfect_similarity(g) == 0.0
    half = vol.clone()
    k = int(torch.nonzero(vol[:10])[0])
    half[k] = 0
    g["ligand"].ph = half
    coeff = torch.tensor([1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0.]) * 7.999999999 * (torch.tensor([1.0, 1.0, 0.7, 1.0, 1.0, 0.7, 1.0, 1.0, 0.7, 1.0, 0.837]) * torch.pi / 2) ** 1.5
    want = 1.0 - float(vol[k] * coeff[k] / (vol * coeff).sum())
    assert abs(get_perfect_similarity(g) - want) < 1e-6
    # Gaussian volume: the self-overlap constant and the 1/e distance
    assert abs(ofit.gaussian_volume(1.0, 1.0, 0.0) - 7.999999999 * (np.pi / 2) ** 1.5) < 1e-12
    assert abs(ofit.gaussian_volume(0.7, 0.7, 2 / 0.7) / ofit.gaussian_volume(0.7, 0.7, 0.0) - np.exp(-1)) < 1e-12


def test_fit_helpers_match_reference_code(golden_dir, tmp_path):
    """`get_perfect_similarity` and `analyze_results` against the reference's own functions (tests/golden/make_fit_golden.py):
    similarities to 1e-6, the csv files byte for byte."""
    import torch
    from diffphore_amd import synthetic
    from diffphore_amd.fitness import get_perfect_similarity
    G = np.load(os.path.join(golden_dir, "ref_fit_helpers.npz"))
    for k in range(6):
        (g,) = synthetic.synthetic_pairs(40 + k, 1, 14, 2, 10 + k)
        g["ligand"].ph = torch.as_tensor(G[f"sim{k}_ph"])
        assert abs(get_perfect_similarity(g) - float(G["sims"][k])) < 1e-6
    res = {"name": ["t1__a", "t1__b", "t2__c", "t3__d"], "run_time": [1.0, 2.5, 3.0, 0.5],
           "fitscore": [[0.1, 0.7, 0.3], [0.9, 0.2, 0.8, 0.1, 0.5, 0.6], [-2.0, -2.0], [0.7, 0.7, 0.1, 0.0, 0.2, 0.3]]}
    args = types.SimpleNamespace(out_dir=str(tmp_path), cutoff=0.6)
    fitmod.analyze_results(args, res)
    assert open(tmp_path / "ranked_results.csv").read() == str(G["csv"])
    assert open(tmp_path / "ranked_results_gt0.6.csv").read() == str(G["csv_cut"])
