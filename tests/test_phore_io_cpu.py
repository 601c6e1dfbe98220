"""`.phore` parser + pharmacophore graph builder.  The shipped example is read from /root/reference when
present (build container only); the hand-made record below always runs."""
import os

import numpy as np
import pytest

from diffphore_amd import phore_io, synthetic

TOY = "\n".join([
    "toy_phore",
    "HD\t1\t1.2\t1\t0.0\t0.0\t0.0\t1\t1.0\t0.0\t0.0\t55\t1",
    "AR\t0.7\t1.0\t1\t3.0\t0.0\t0.0\t0\t0\t0\t0\t7\t1",
    "HA\t1\t1.2\t1\t0.0\t4.0\t0.0\t1\t0.0\t6.0\t0.0\t56\t1",
    "EX\t0.837\t0.5\t1\t1.0\t1.0\t1.0\t0\t0\t0\t0\t0\t1",
    "EX\t0.837\t0.5\t1\t30.0\t30.0\t30.0\t0\t0\t0\t0\t0\t1",
    "$$$$", ""])


def test_toy_phore_graph():
    (ph,) = phore_io.parse_phore_text(TOY)
    assert ph.id == "toy_phore" and len(ph.features) == 3 and len(ph.exclusion_volumes) == 2
    g = phore_io.phore_graph(ph)
    assert g["x"].shape == (5, 5) and g["phoretype"].shape == (5, 11)
    assert g["x"][:, 0].tolist() == [1, 2, 4, 10, 10]
    assert g["x"][:, 1].tolist() == [1, 1, 1, 0, 0]          # is-EX index: True -> 0
    assert g["x"][:, 2].tolist() == [0, 1, 0, 1, 1]          # has-norm index: True -> 0
    np.testing.assert_allclose(g["norm"][0], [1, 0, 0])
    np.testing.assert_allclose(g["norm"][2], [0, 1, 0])
    np.testing.assert_allclose(g["norm"][1], [0, 0, 0])
    ei = set(map(tuple, g["edge_index"].T.tolist()))
    assert {(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)} <= ei      # features fully connected
    assert (3, 0) in ei and (3, 1) in ei and (3, 2) in ei and (3, 4) not in ei   # EX -> everything within 5 A
    assert [e for e in ei if e[0] == 4] == [(4, 2)]                      # isolated EX: nearest-neighbour fallback (node 2 is closest)
    # plugs straight into the complex builder used for the synthetic workloads
    lig = synthetic.make_ligand(np.random.default_rng(0), 12, 2)
    c = synthetic.make_complex(lig, g)
    assert c["phore"].pos.shape == (5, 3) and abs(float(c["phore"].pos.mean())) < 1e-5


def test_malformed_record_is_rejected():
    bad = TOY.replace("0.7\t1.0", "zero\t1.0")
    assert phore_io.parse_phore_text(bad) == []               # reference default: skip the whole pharmacophore
    with pytest.raises(SyntaxError):
        phore_io.parse_phore_text(bad, skip_wrong_lines=False)


@pytest.mark.skipif(not os.path.exists("/root/reference/examples/phore/sQC_QFA_complex.phore"), reason="reference examples not mounted")
def test_shipped_example_counts():
    (ph,) = phore_io.parse_phore("/root/reference/examples/phore/sQC_QFA_complex.phore")
    g = phore_io.phore_graph(ph, consider_ex=True, neighbor_cutoff=5.0, ex_connected=True)
    # 5 features + 74 exclusion spheres = 79 nodes, 811 phore-phore edges
    assert len(ph.features) == 5 and len(ph.exclusion_volumes) == 74
    assert g["pos"].shape == (79, 3) and g["edge_index"].shape == (2, 811)


def test_parser_and_graph_match_reference_code(golden_dir):
    """`phore_io` against the reference's own `parse_phore` + `get_phore_graph` + `phore_featurizer`
    (tests/golden/make_phore_golden.py): node features, coordinates, unit normals and the directed edge list, in order."""
    G = np.load(os.path.join(golden_dir, "ref_phore_graph.npz"))
    phs = phore_io.parse_phore(os.path.join(golden_dir, "sQC_QFA_complex.phore")) + phore_io.parse_phore_text(str(G["generated_text"]))
    assert len(phs) == int(G["n_graphs"]) == 3
    for k, ph in enumerate(phs):
        assert ph.id == str(G[f"g{k}_id"])
        g = phore_io.phore_graph(ph)
        np.testing.assert_allclose(np.asarray(g["x"], dtype=np.float64), G[f"g{k}_x"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(g["pos"], G[f"g{k}_pos"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(g["norm"], G[f"g{k}_norm"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(np.asarray(g["edge_index"]), G[f"g{k}_edge_index"])
        # phoretype (set by the dataset class in the reference, pdbbind_phore.py) is the one-hot of x[:, 0]
        assert (np.asarray(g["phoretype"]).argmax(1) == G[f"g{k}_x"][:, 0].astype(int)).all()
