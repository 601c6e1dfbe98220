"""Parity of the fused HIP tensor-product conv (dp_tp_conv, through the C ABI) against the oracle's
TensorProductConvLayer restatement on the same seeded inputs.  fp32 kernel vs fp64 oracle; tolerance 2e-5 of the
per-tensor max magnitude (fp32 accumulation over <=2200 weights x <=80 edges)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffphore_amd import _lib, config, packing, weights  # noqa: E402

TOL = {"f32": 2e-5, "bf16x6": 2e-5, "bf16x9": 2e-5, "bf16x3": 3e-4}
PRECS = ["f32", "bf16x6", "bf16x9", "bf16x3"]


def _oracle_conv(sd, prefix, kind, x_gather, attr, sh, scatter, n_out):
    from oracle import e3
    from oracle.model import OracleScoreModel
    m = OracleScoreModel(sd, config.ModelConfig(), dtype=torch.float64, tables=(np.ones(1000), np.ones(5001)))
    ir_in = m.seq[min(kind, 3)] if kind < 4 else m.seq[3]
    if kind < 4:
        ir_out, ir_sh = m.seq[min(kind + 1, 3)], m.sh_irreps
    elif kind == 4:
        ir_out, ir_sh = m.final_out, m.sh_irreps
    else:
        ir_out, ir_sh = m.tor_out, m.tor_sh_irreps
    w = m.mlp(prefix + ".fc", attr.double())
    tp = e3.fctp_apply(x_gather.double(), sh.double(), w, ir_in, ir_sh, ir_out)
    from oracle.model import scatter_mean
    out = scatter_mean(tp, scatter, n_out)
    bn = prefix + ".batch_norm"
    return e3.batchnorm_eval(out, ir_out, m.sd[bn + ".running_mean"], m.sd[bn + ".running_var"], m.sd[bn + ".weight"],
                             m.sd[bn + ".bias"])


def _dev(t):
    return t.contiguous().cuda()


def _run(args_dict, keep):
    a = _lib.DpConvArgs()
    for k, v in args_dict.items():
        if torch.is_tensor(v):
            keep.append(v)
            v = v.data_ptr()
        setattr(a, k, v)
    _lib.check(_lib.lib().dp_tp_conv(C.byref(a), torch.cuda.current_stream().cuda_stream), "dp_tp_conv")
    torch.cuda.synchronize()


def _i32(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.int32)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("slices", [1, 3])
def test_conv_csr_ragged(kind, slices, prec):
    """lig-lig / phore-phore style: explicit CSR with ragged degrees incl. empty rows, several poses, residual + accumulate."""
    g = torch.Generator().manual_seed(100 + kind)
    sd = weights.init_state_dict(7)
    prefix = f"encoder.lig_conv_layers.{kind}"
    dims = _lib.conv_dims(kind)
    B = 5
    n_nodes = [9, 1, 17, 30, 12]
    scat_ptr = np.concatenate([[0], np.cumsum(n_nodes)])
    N = int(scat_ptr[-1])
    h = torch.randn(N, dims["d_in"], generator=g)
    rowptr, rowend, nbr, scatter = [], [], [], []
    e0 = 0
    for b in range(B):
        e0 += int(torch.randint(0, 5, (1,), generator=g))          # gap between poses (capacity padding)
        for n in range(scat_ptr[b], scat_ptr[b + 1]):
            deg = int(torch.randint(0, 40, (1,), generator=g)) if n_nodes[b] > 1 else 3
            if n % 7 == 3:
                deg = 0
            rowptr.append(e0)
            rowend.append(e0 + deg)
            for _ in range(deg):
                nbr.append(int(torch.randint(int(scat_ptr[b]), int(scat_ptr[b + 1]), (1,), generator=g)))
                scatter.append(n)
            e0 += deg
    E_slots = e0 + 3
    slot_of = []
    for n in range(N):
        slot_of += list(range(rowptr[n], rowend[n]))
    slot_of = torch.tensor(slot_of)
    nbr_slots = torch.zeros(E_slots, dtype=torch.int32)
    nbr_slots[slot_of] = torch.tensor(nbr, dtype=torch.int32)
    scat_slots = torch.full((E_slots,), -1, dtype=torch.int32)
    scat_slots[slot_of] = torch.tensor(scatter, dtype=torch.int32)
    emb = torch.randn(E_slots, 20, generator=g)
    sh = torch.randn(E_slots, 9, generator=g)
    sh[:, 0] = 1.0
    scatter_t = torch.tensor(scatter)
    nbr_t = torch.tensor(nbr)
    attr = torch.cat([emb[slot_of], h[scatter_t, :20], h[nbr_t, :20]], 1)
    ref = _oracle_conv(sd, prefix, kind, h[nbr_t], attr, sh[slot_of], scatter_t, N)
    pk = packing.pack_conv(sd, prefix, kind, _lib.PRECISIONS[prec])
    resid = torch.randn(N, dims["d_in"], generator=g)
    out = torch.full((N, dims["d_out"]), float("nan")).cuda()
    keep = []
    base = dict(kind=kind, mode=_lib.EDGES_CSR, B=B, slices=slices, precision=_lib.PRECISIONS[prec], max_scatter=max(n_nodes), scat_ptr=_dev(_i32(scat_ptr)),
                rowptr=_dev(_i32(rowptr)), rowend=_dev(_i32(rowend)), nbr=_dev(nbr_slots), edge_scat=_dev(scat_slots),
                gather_tab=_dev(h), gather_stride=dims["d_in"], attr1_tab=_dev(h), attr1_stride=dims["d_in"],
                attr2_tab=_dev(h), attr2_stride=dims["d_in"], attr_swap=0, edge_emb=_dev(emb), edge_sh=_dev(sh), sh_stride=9,
                w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]), bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]),
                out=out, out_stride=dims["d_out"], accumulate=0, resid=_dev(resid), resid_stride=dims["d_in"],
                resid_width=dims["d_in"])
    _run(base, keep)
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err < TOL[prec], (prec, err)
    # accumulate mode: out += conv
    base["accumulate"] = 1
    _run(base, keep)
    got2 = out.cpu().double()
    err = (got2 - (want + ref)).abs().max().item() / want.abs().max().item()
    assert err < 2 * TOL[prec], (prec, err)
    # phase-split launch (one workgroup per accumulation phase): per-channel sums are untouched -> bit-identical output
    base["accumulate"] = 0
    out.fill_(float("nan"))
    base["phase_split"] = 1
    _run(base, keep)
    assert torch.equal(out.cpu().double(), got), "phase-split launch must reproduce the unsplit result bit for bit"


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("kind", [0, 3])
@pytest.mark.parametrize("mode", ["p2l", "l2p"])
def test_conv_dense_cross(kind, mode, prec):
    g = torch.Generator().manual_seed(300 + kind)
    sd = weights.init_state_dict(11)
    fam = "phore_to_lig_conv_layers" if mode == "p2l" else "lig_to_phore_norm_conv_layers"
    if kind == 3 and mode == "l2p":
        fam = "lig_to_phore_conv_layers"
    prefix = f"encoder.{fam}.{kind}"
    dims = _lib.conv_dims(kind)
    NL, NP = [7, 30, 13], [32, 5, 79]
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    ph_ptr = np.concatenate([[0], np.cumsum(NP)])
    x_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) * np.asarray(NP))])
    hl = torch.randn(int(lig_ptr[-1]), dims["d_in"], generator=g)
    hp = torch.randn(int(ph_ptr[-1]), dims["d_in"], generator=g)
    E = int(x_ptr[-1])
    emb = torch.randn(E, 20, generator=g)
    sh = torch.randn(E, 9, generator=g)
    src = torch.cat([torch.arange(lig_ptr[b], lig_ptr[b + 1]).repeat_interleave(NP[b]) for b in range(B)])
    dst = torch.cat([torch.arange(ph_ptr[b], ph_ptr[b + 1]).repeat(NL[b]) for b in range(B)])
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    if mode == "p2l":
        ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, hl.shape[0])
        n_out, gt, scat_ptr, m = hl.shape[0], hp, lig_ptr, _lib.EDGES_P2L
    else:
        ref = _oracle_conv(sd, prefix, kind, hl[src], attr, sh, dst, hp.shape[0])
        n_out, gt, scat_ptr, m = hp.shape[0], hl, ph_ptr, _lib.EDGES_L2P
    pk = packing.pack_conv(sd, prefix, kind, _lib.PRECISIONS[prec])
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    keep = []
    _run(dict(kind=kind, mode=m, B=B, slices=3, precision=_lib.PRECISIONS[prec], max_scatter=max(NL) if mode == "p2l" else max(NP), scat_ptr=_dev(_i32(scat_ptr)), lig_ptr=_dev(_i32(lig_ptr)),
              ph_ptr=_dev(_i32(ph_ptr)), x_ptr=_dev(_i32(x_ptr)), gather_tab=_dev(gt), gather_stride=dims["d_in"],
              attr1_tab=_dev(hl), attr1_stride=dims["d_in"], attr2_tab=_dev(hp), attr2_stride=dims["d_in"], attr_swap=0,
              edge_emb=_dev(emb), edge_sh=_dev(sh), sh_stride=9, w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]),
              bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]), out=out, out_stride=dims["d_out"],
              accumulate=0, resid=None, resid_stride=0, resid_width=0), keep)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < TOL[prec], (prec, err)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["p2l", "l2p"])
@pytest.mark.parametrize("big", [False, True])
def test_conv_node_side_dense_cross(kind, mode, big):
    """Node-side formulation (dp_conv_node.hip, DP_ALGO_NODE) of the dense cross convs vs the fp64 oracle AND vs the edge-side
    kernel: ragged pose sizes (1 node, < 16, > 16 nodes; 5 / 32 / 79 edges per node = 1-3 edge passes), residual + accumulate.
    `big`: ligands above the ligand-ligand cap of dp_forward (100 / 110 atoms: 7 node groups per pose, four edge passes)."""
    g = torch.Generator().manual_seed(700 + kind)
    sd = weights.init_state_dict(11)
    fam = "phore_to_lig_conv_layers" if mode == "p2l" else "lig_to_phore_norm_conv_layers"
    if kind == 3 and mode == "l2p":
        fam = "lig_to_phore_conv_layers"
    prefix = f"encoder.{fam}.{kind}"
    dims = _lib.conv_dims(kind)
    NL, NP = ([7, 30, 13, 1, 40], [32, 5, 79, 33, 17]) if not big else ([100, 110, 97], [24, 40, 16])
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    ph_ptr = np.concatenate([[0], np.cumsum(NP)])
    x_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) * np.asarray(NP))])
    hl = torch.randn(int(lig_ptr[-1]), dims["d_in"], generator=g)
    hp = torch.randn(int(ph_ptr[-1]), dims["d_in"], generator=g)
    E = int(x_ptr[-1])
    emb = torch.randn(E, 20, generator=g)
    sh = torch.randn(E, 9, generator=g)
    sh[:, 0] = 1.0                                  # Y_0 = 1 for every edge ('component' normalisation): the node-side kernel relies on it
    src = torch.cat([torch.arange(lig_ptr[b], lig_ptr[b + 1]).repeat_interleave(NP[b]) for b in range(B)])
    dst = torch.cat([torch.arange(ph_ptr[b], ph_ptr[b + 1]).repeat(NL[b]) for b in range(B)])
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    if mode == "p2l":
        ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, hl.shape[0])
        n_out, gt, scat_ptr, m, mx = hl.shape[0], hp, lig_ptr, _lib.EDGES_P2L, max(NL)
    else:
        ref = _oracle_conv(sd, prefix, kind, hl[src], attr, sh, dst, hp.shape[0])
        n_out, gt, scat_ptr, m, mx = hp.shape[0], hl, ph_ptr, _lib.EDGES_L2P, max(NP)
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_F32, node_side=True)
    nb = _lib.lib().dp_conv_node_max_nodes()
    resid = torch.randn(n_out, dims["d_in"], generator=g)
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    keep = []
    base = dict(kind=kind, mode=m, B=B, slices=(mx + nb - 1) // nb, precision=_lib.PREC_F32, max_scatter=mx, scat_ptr=_dev(_i32(scat_ptr)),
                lig_ptr=_dev(_i32(lig_ptr)), ph_ptr=_dev(_i32(ph_ptr)), x_ptr=_dev(_i32(x_ptr)), gather_tab=_dev(gt), gather_stride=dims["d_in"],
                attr1_tab=_dev(hl), attr1_stride=dims["d_in"], attr2_tab=_dev(hp), attr2_stride=dims["d_in"], attr_swap=0,
                edge_emb=_dev(emb), edge_sh=_dev(sh), sh_stride=9, w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]), w1n=_dev(pk["w1n"]), w2n=_dev(pk["w2n"]),
                bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]), out=out, out_stride=dims["d_out"],
                accumulate=0, resid=_dev(resid), resid_stride=dims["d_in"], resid_width=dims["d_in"], algo=_lib.ALGO_NODE)
    _run(base, keep)
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    scale = want.abs().max().item()
    err = (got - want).abs().max().item() / scale
    assert err < TOL["f32"], err
    base["accumulate"] = 1
    _run(base, keep)
    err = (out.cpu().double() - (want + ref)).abs().max().item() / scale
    assert err < 2 * TOL["f32"], err
    # the edge-side kernel on the same inputs: same operator, different association -> equal to fp32 rounding
    out2 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    base.update(algo=_lib.ALGO_EDGE, accumulate=0, out=out2, slices=3)
    _run(base, keep)
    err = (out2.cpu().double() - got).abs().max().item() / scale
    assert err < TOL["f32"], err
    # reruns are bit-identical (fixed summation order, no atomics)
    out3 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    base.update(algo=_lib.ALGO_NODE, out=out3, slices=(mx + nb - 1) // nb)
    _run(base, keep)
    assert torch.equal(out3.cpu().double(), got)
    # path split (two workgroups per node group, each owning half of the out slots): every channel is computed as before
    out4 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    base.update(out=out4, path_split=1)
    _run(base, keep)
    assert torch.equal(out4.cpu().double(), got)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_conv_node_side_ligand_pairs(kind):
    """Node-side lig-lig conv (DP_EDGES_LLPAIR): the ligand multigraph (radius edges + bond edges, bonded pairs carry BOTH) as a dense
    atom x atom pair table.  Held to the fp64 oracle run on the explicit edge list and to the edge-side CSR kernel on the same edges."""
    g = torch.Generator().manual_seed(900 + kind)
    sd = weights.init_state_dict(11)
    prefix = f"encoder.lig_conv_layers.{kind}"
    dims = _lib.conv_dims(kind)
    NL = [9, 30, 1, 45, 17]
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    N = int(lig_ptr[-1])
    h = torch.randn(N, dims["d_in"], generator=g)
    rowptr, rowend, nbr, scat, pair_rad, pair_bond, is_bond = [], [], [], [], [], [], []
    pair_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) ** 2)])
    e0 = 0
    for b in range(B):
        n = NL[b]
        adj = (torch.rand(n, n, generator=g) < 0.6).numpy()
        np.fill_diagonal(adj, False)
        bond = (torch.rand(n, n, generator=g) < 0.08).numpy() & ~np.eye(n, dtype=bool)
        adj[3 % n] = False                                        # an atom without radius edges
        if n > 5:
            adj[5], bond[5] = False, False                        # ... and one without any edge
        pr, pb_ = -np.ones((n, n), dtype=np.int32), -np.ones((n, n), dtype=np.int32)
        for a_ in range(n):
            rowptr.append(e0)
            for y in np.nonzero(bond[a_])[0]:                      # bond edges first, as lig_graph_kernel lays a row out
                pb_[a_, y] = e0
                nbr.append(int(lig_ptr[b]) + int(y)); scat.append(int(lig_ptr[b]) + a_); is_bond.append(True)
                e0 += 1
            for y in np.nonzero(adj[a_])[0]:
                pr[a_, y] = e0
                nbr.append(int(lig_ptr[b]) + int(y)); scat.append(int(lig_ptr[b]) + a_); is_bond.append(False)
                e0 += 1
            rowend.append(e0)
        pair_rad.append(pr.reshape(-1)); pair_bond.append(pb_.reshape(-1))
    E = e0
    nbr_t, scat_t = torch.tensor(nbr), torch.tensor(scat)
    # geometry is a property of the PAIR: the radius and the bond edge of a pair share sh (and the gathered row), only emb differs
    pos = torch.randn(N, 3, generator=g)
    from oracle import e3
    sh = e3.spherical_harmonics_lmax2((pos[nbr_t] - pos[scat_t]).double()).float()
    emb = torch.randn(E, 20, generator=g)
    attr = torch.cat([emb, h[scat_t, :20], h[nbr_t, :20]], 1)
    ref = _oracle_conv(sd, prefix, kind, h[nbr_t], attr, sh, scat_t, N)
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_F32, node_side=True)
    nb = _lib.lib().dp_conv_node_max_nodes()
    resid = torch.randn(N, dims["d_in"], generator=g)
    out = torch.full((N, dims["d_out"]), float("nan")).cuda()
    keep = []
    common = dict(kind=kind, B=B, precision=_lib.PREC_F32, max_scatter=max(NL), scat_ptr=_dev(_i32(lig_ptr)), lig_ptr=_dev(_i32(lig_ptr)),
                  rowptr=_dev(_i32(rowptr)), rowend=_dev(_i32(rowend)), nbr=_dev(_i32(nbr)), edge_scat=_dev(_i32(scat)),
                  gather_tab=_dev(h), gather_stride=dims["d_in"], attr1_tab=_dev(h), attr1_stride=dims["d_in"], attr2_tab=_dev(h),
                  attr2_stride=dims["d_in"], attr_swap=0, edge_emb=_dev(emb), edge_sh=_dev(sh), sh_stride=9, w1p=_dev(pk["w1p"]),
                  w2p=_dev(pk["w2p"]), w1n=_dev(pk["w1n"]), w2n=_dev(pk["w2n"]), bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]),
                  out_stride=dims["d_out"], accumulate=0, resid=_dev(resid), resid_stride=dims["d_in"], resid_width=dims["d_in"],
                  pair_a=_dev(_i32(np.concatenate(pair_rad))), pair_b=_dev(_i32(np.concatenate(pair_bond))), pair_ptr=_dev(_i32(pair_ptr)))
    _run(dict(common, mode=_lib.EDGES_LLPAIR, algo=_lib.ALGO_NODE, slices=(max(NL) + nb - 1) // nb, out=out), keep)
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    scale = want.abs().max().item()
    err = (got - want).abs().max().item() / scale
    assert err < TOL["f32"], err
    out2 = torch.full((N, dims["d_out"]), float("nan")).cuda()
    _run(dict(common, mode=_lib.EDGES_CSR, algo=_lib.ALGO_EDGE, slices=2, out=out2), keep)
    err = (out2.cpu().double() - got).abs().max().item() / scale
    assert err < TOL["f32"], err
    out3 = torch.full((N, dims["d_out"]), float("nan")).cuda()
    _run(dict(common, mode=_lib.EDGES_LLPAIR, algo=_lib.ALGO_NODE, slices=(max(NL) + nb - 1) // nb, out=out3, path_split=1), keep)
    assert torch.equal(out3.cpu().double(), got)               # path split: bit-identical


@pytest.mark.parametrize("prec", PRECS)
def test_conv_center_and_torsion_kinds(prec):
    g = torch.Generator().manual_seed(500)
    sd = weights.init_state_dict(13)
    # final_conv: one scatter node per pose, one edge per atom, attr = [emb | h_l[atom,:20]]
    dims = _lib.conv_dims(4)
    NL = [20, 3, 44]
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    n = int(lig_ptr[-1])
    hl = torch.randn(n, 100, generator=g)
    emb = torch.randn(n, 20, generator=g)
    sh = torch.randn(n, 9, generator=g)
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(NL))
    attr = torch.cat([emb, hl[:, :20]], 1)
    ref = _oracle_conv(sd, "final_conv", 4, hl, attr, sh, batch, B)
    pk = packing.pack_conv(sd, "final_conv", 4, _lib.PRECISIONS[prec])
    out = torch.full((B, 12), float("nan")).cuda()
    keep = []
    _run(dict(kind=4, mode=_lib.EDGES_CENTER, B=B, slices=1, precision=_lib.PRECISIONS[prec], max_scatter=1, scat_ptr=_dev(_i32(np.arange(B + 1))), lig_ptr=_dev(_i32(lig_ptr)),
              gather_tab=_dev(hl), gather_stride=100, attr1_tab=_dev(hl), attr1_stride=100, attr2_tab=None, attr2_stride=0,
              attr_swap=0, edge_emb=_dev(emb), edge_sh=_dev(sh), sh_stride=9, w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]),
              bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]), out=out, out_stride=12, accumulate=0,
              resid=None, resid_stride=0, resid_width=0), keep)
    got = out.cpu().double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < TOL[prec], (prec, err)

    # tor_bond_conv: scatter = bonds (CSR), gather = atoms, attr = [emb | h_l[atom,:20] | hb[bond,:20]], sh = 45-dim FullTP
    from oracle import e3
    T = [4, 0, 9]
    tor_ptr = np.concatenate([[0], np.cumsum(T)])
    nb = int(tor_ptr[-1])
    hb = torch.randn(nb, 20, generator=g)
    rowptr, rowend, nbr, scat = [], [], [], []
    e0 = 0
    for b in range(B):
        for k in range(tor_ptr[b], tor_ptr[b + 1]):
            deg = int(torch.randint(1, 33, (1,), generator=g))
            rowptr.append(e0)
            rowend.append(e0 + deg)
            nbr += [int(torch.randint(int(lig_ptr[b]), int(lig_ptr[b + 1]), (1,), generator=g)) for _ in range(deg)]
            scat += [k] * deg
            e0 += deg
    E = e0
    nbr_t, scat_t = torch.tensor(nbr), torch.tensor(scat)
    temb = torch.randn(E, 20, generator=g)
    sh9 = e3.spherical_harmonics_lmax2(torch.randn(E, 3, generator=g).double())
    b5 = e3.spherical_harmonics_2e(torch.randn(nb, 3, generator=g).double())[scat_t]
    sh45 = e3.full_tp_apply(sh9, b5, e3.SH_IRREPS, e3.parse_irreps("1x2e"))
    sh7 = sh45[:, :7].float()                      # [0e | 1o | 1e] is all tor_bond_conv can consume
    attr = torch.cat([temb, hl[nbr_t, :20], hb[scat_t]], 1)
    ref = _oracle_conv(sd, "tor_bond_conv", 5, hl[nbr_t], attr, sh45, scat_t, nb)
    pk = packing.pack_conv(sd, "tor_bond_conv", 5, _lib.PRECISIONS[prec])
    out = torch.full((nb, 40), float("nan")).cuda()
    _run(dict(kind=5, mode=_lib.EDGES_CSR, B=B, slices=1, precision=_lib.PRECISIONS[prec], max_scatter=max(T), scat_ptr=_dev(_i32(tor_ptr)), rowptr=_dev(_i32(rowptr)),
              rowend=_dev(_i32(rowend)), nbr=_dev(_i32(nbr)), edge_scat=_dev(_i32(scat)), gather_tab=_dev(hl), gather_stride=100,
              attr1_tab=_dev(hl), attr1_stride=100, attr2_tab=_dev(hb), attr2_stride=20, attr_swap=1,
              edge_emb=_dev(temb), edge_sh=_dev(sh7), sh_stride=7, w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]),
              bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]), out=out, out_stride=40, accumulate=0,
              resid=None, resid_stride=0, resid_width=0), keep)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < TOL[prec], (prec, err)


def test_conv_rejects_too_many_nodes_per_workgroup():
    a = _lib.DpConvArgs()
    dummy = torch.zeros(16, device="cuda")
    for f in ("scat_ptr", "gather_tab", "attr1_tab", "attr2_tab", "edge_emb", "edge_sh", "w1p", "w2p", "bn_scale", "bn_shift", "out",
              "lig_ptr", "ph_ptr", "x_ptr"):
        setattr(a, f, dummy.data_ptr())
    a.kind, a.mode, a.B, a.slices, a.precision, a.max_scatter = 0, _lib.EDGES_P2L, 1, 1, _lib.PREC_BF16X6, 79
    assert _lib.lib().dp_tp_conv(C.byref(a), None) == -3          # DP_E_TOOLARGE: 79 nodes > 32 per workgroup
    a.max_scatter = 0
    assert _lib.lib().dp_tp_conv(C.byref(a), None) == -1


@pytest.mark.parametrize("kind", [1, 3])
def test_conv_node_lean_instantiation_is_bit_identical(kind):
    """DpConvArgs.no_tail = 1 selects the node-side instantiation without the half-pass variant.  On a batch where the hint is true (no
    pose ends on a pass of <= 16 edge slots) it must reproduce the general kernel bit for bit; where the hint is FALSE the lean kernel
    still computes the right answer (a short pass run as a full, zero-padded one) - only the guarantee of identical bits is gone."""
    g = torch.Generator().manual_seed(2700 + kind)
    sd = weights.init_state_dict(11)
    prefix = f"encoder.phore_to_lig_conv_layers.{kind}"
    dims = _lib.conv_dims(kind)
    for NL, NP, hint_true in (([30, 7, 20], [32, 64, 17], True), ([30, 7, 20], [40, 5, 33], False)):
        B = len(NL)
        lig_ptr = np.concatenate([[0], np.cumsum(NL)])
        ph_ptr = np.concatenate([[0], np.cumsum(NP)])
        x_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) * np.asarray(NP))])
        hl = torch.randn(int(lig_ptr[-1]), dims["d_in"], generator=g)
        hp = torch.randn(int(ph_ptr[-1]), dims["d_in"], generator=g)
        E = int(x_ptr[-1])
        emb = torch.randn(E, 20, generator=g)
        sh = torch.randn(E, 9, generator=g)
        sh[:, 0] = 1.0
        src = torch.cat([torch.arange(lig_ptr[b], lig_ptr[b + 1]).repeat_interleave(NP[b]) for b in range(B)])
        dst = torch.cat([torch.arange(ph_ptr[b], ph_ptr[b + 1]).repeat(NL[b]) for b in range(B)])
        attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
        ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, hl.shape[0])
        pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_F32, node_side=True)
        nb = _lib.lib().dp_conv_node_max_nodes()
        outs, keep = [], []
        base = dict(kind=kind, mode=_lib.EDGES_P2L, B=B, slices=(max(NL) + nb - 1) // nb, precision=_lib.PREC_F32, max_scatter=max(NL),
                    scat_ptr=_dev(_i32(lig_ptr)), lig_ptr=_dev(_i32(lig_ptr)), ph_ptr=_dev(_i32(ph_ptr)), x_ptr=_dev(_i32(x_ptr)), gather_tab=_dev(hp),
                    gather_stride=dims["d_in"], attr1_tab=_dev(hl), attr1_stride=dims["d_in"], attr2_tab=_dev(hp), attr2_stride=dims["d_in"],
                    attr_swap=0, edge_emb=_dev(emb), edge_sh=_dev(sh), sh_stride=9, w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]), w1n=_dev(pk["w1n"]),
                    w2n=_dev(pk["w2n"]), bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]), out_stride=dims["d_out"], accumulate=0,
                    algo=_lib.ALGO_NODE)
        for hint in (0, 1):
            out = torch.full((hl.shape[0], dims["d_out"]), float("nan")).cuda()
            base.update(out=out, no_tail=hint)
            _run(base, keep)
            outs.append(out.cpu())
            assert torch.isfinite(outs[-1]).all()
            assert (outs[-1].double() - ref).abs().max().item() / ref.abs().max().item() < TOL["f32"]
        if hint_true:
            assert torch.equal(outs[0], outs[1])
        else:
            assert (outs[0] - outs[1]).abs().max().item() / ref.abs().max().item() < 2e-6


@pytest.mark.parametrize("kind,mode", [(0, "p2l"), (2, "p2l"), (3, "p2l"), (1, "l2p"), (2, "l2p")])
def test_conv_node_fused_pair_equals_the_two_launches(kind, mode):
    """dp_tp_conv_node_fused([conv, norm-conv]) - one launch, shared accumulators - against the two convs launched one after the other
    (accumulate = 1 for the second): same sums in a different order -> equal to fp32 rounding; and against the fp64 oracle's sum.  Ragged
    poses as above (1-3 edge passes incl. half passes)."""
    g = torch.Generator().manual_seed(1700 + kind)
    sd = weights.init_state_dict(11)
    fams = ("phore_to_lig_conv_layers", "phore_to_lig_norm_conv_layers") if mode == "p2l" else ("lig_to_phore_conv_layers", "lig_to_phore_norm_conv_layers")
    dims = _lib.conv_dims(kind)
    NL, NP = [7, 30, 13, 1, 40, 16], [32, 5, 79, 33, 17, 48]
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    ph_ptr = np.concatenate([[0], np.cumsum(NP)])
    x_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) * np.asarray(NP))])
    hl = torch.randn(int(lig_ptr[-1]), dims["d_in"], generator=g)
    hp = torch.randn(int(ph_ptr[-1]), dims["d_in"], generator=g)
    E = int(x_ptr[-1])
    emb = torch.randn(E, 20, generator=g)
    shs = []
    for _ in range(2):
        sh = torch.randn(E, 9, generator=g)
        sh[:, 0] = 1.0
        shs.append(sh)
    src = torch.cat([torch.arange(lig_ptr[b], lig_ptr[b + 1]).repeat_interleave(NP[b]) for b in range(B)])
    dst = torch.cat([torch.arange(ph_ptr[b], ph_ptr[b + 1]).repeat(NL[b]) for b in range(B)])
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    if mode == "p2l":
        n_out, gt, scat_ptr, m, mx = hl.shape[0], hp, lig_ptr, _lib.EDGES_P2L, max(NL)
        refs = [_oracle_conv(sd, f"encoder.{fams[i]}.{kind}", kind, hp[dst], attr, shs[i], src, n_out) for i in range(2)]
    else:
        n_out, gt, scat_ptr, m, mx = hp.shape[0], hl, ph_ptr, _lib.EDGES_L2P, max(NP)
        refs = [_oracle_conv(sd, f"encoder.{fams[i]}.{kind}", kind, hl[src], attr, shs[i], dst, n_out) for i in range(2)]
    pks = [packing.pack_conv(sd, f"encoder.{fams[i]}.{kind}", kind, _lib.PREC_F32, node_side=True) for i in range(2)]
    nb = _lib.lib().dp_conv_node_max_nodes()
    resid = torch.randn(n_out, dims["d_in"], generator=g)
    keep = []
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream

    D = dict(scat_ptr=_dev(_i32(scat_ptr)), lig_ptr=_dev(_i32(lig_ptr)), ph_ptr=_dev(_i32(ph_ptr)), x_ptr=_dev(_i32(x_ptr)), gt=_dev(gt),
             hl=_dev(hl), hp=_dev(hp), emb=_dev(emb), resid=_dev(resid), sh=[_dev(x) for x in shs],
             pk=[{k_: _dev(v) for k_, v in pk.items()} for pk in pks])          # one device copy each: the convs of a fused launch share them

    def args(i, out, accumulate, path_split=0):
        pk = D["pk"][i]
        d = dict(kind=kind, mode=m, B=B, slices=(mx + nb - 1) // nb, precision=_lib.PREC_F32, max_scatter=mx, scat_ptr=D["scat_ptr"],
                 lig_ptr=D["lig_ptr"], ph_ptr=D["ph_ptr"], x_ptr=D["x_ptr"], gather_tab=D["gt"], gather_stride=dims["d_in"],
                 attr1_tab=D["hl"], attr1_stride=dims["d_in"], attr2_tab=D["hp"], attr2_stride=dims["d_in"], attr_swap=0,
                 edge_emb=D["emb"], edge_sh=D["sh"][i], sh_stride=9, w1p=pk["w1p"], w2p=pk["w2p"], w1n=pk["w1n"],
                 w2n=pk["w2n"], bn_scale=pk["bn_scale"], bn_shift=pk["bn_shift"], out=out, out_stride=dims["d_out"],
                 accumulate=accumulate, resid=D["resid"], resid_stride=dims["d_in"], resid_width=dims["d_in"], algo=_lib.ALGO_NODE,
                 path_split=path_split)
        a = _lib.DpConvArgs()
        for k_, v in d.items():
            if torch.is_tensor(v):
                keep.append(v)
                v = v.data_ptr()
            setattr(a, k_, v)
        return a

    want = refs[0] + refs[1]
    want[:, :dims["d_in"]] += resid.double()
    scale = want.abs().max().item()
    out_seq = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    _lib.check(L.dp_tp_conv_node(C.byref(args(0, out_seq, 0)), st), "conv 0")
    _lib.check(L.dp_tp_conv_node(C.byref(args(1, out_seq, 1)), st), "conv 1")
    for ps in (0, 1):
        out_f = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
        arr = (_lib.DpConvArgs * 2)(args(0, out_f, 0, ps), args(1, out_f, 0, ps))
        _lib.check(L.dp_tp_conv_node_fused(arr, 2, st), "fused")
        torch.cuda.synchronize()
        got = out_f.cpu().double()
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() / scale < TOL["f32"]
        assert (got - out_seq.cpu().double()).abs().max().item() / scale < 2e-6
    # conv split (small batches): every conv on its own workgroups + a combine launch - the same sum to fp32 rounding, bit-identical reruns,
    # residual stream and a previous output (accumulate) both honoured
    part = torch.full((2 * n_out * dims["d_out"] + 64,), float("nan")).cuda()
    runs = []
    for acc_mode in (0, 1):
        for _rep in range(2):
            out_s = torch.full((n_out, dims["d_out"]), 0.25).cuda() if acc_mode else torch.full((n_out, dims["d_out"]), float("nan")).cuda()
            arr = (_lib.DpConvArgs * 2)(args(0, out_s, acc_mode, 1), args(1, out_s, 0, 1))
            _lib.check(L.dp_tp_conv_node_fused_split(arr, 2, n_out, part.data_ptr(), part.numel(), st), "fused split")
            torch.cuda.synchronize()
            runs.append(out_s.cpu())
        assert torch.equal(runs[-1], runs[-2])
        want_s = want if not acc_mode else (refs[0] + refs[1] + 0.25)
        assert (runs[-1].double() - want_s).abs().max().item() / scale < TOL["f32"]
    arr = (_lib.DpConvArgs * 2)(args(0, out_seq, 0, 0), args(1, out_seq, 0, 0))
    assert L.dp_tp_conv_node_fused_split(arr, 2, n_out, part.data_ptr(), part.numel(), st) == -1            # needs path_split
    arr = (_lib.DpConvArgs * 2)(args(0, out_seq, 0, 1), args(1, out_seq, 0, 1))
    assert L.dp_tp_conv_node_fused_split(arr, 2, n_out, part.data_ptr(), 10, st) == -3                        # scratch too small
    # sequences the kernel has no instantiation for are refused, not mis-run
    bad = (_lib.DpConvArgs * 2)(args(0, out_seq, 0), args(1, out_seq, 0))
    bad[1].mode = _lib.EDGES_L2P if mode == "p2l" else _lib.EDGES_P2L
    assert L.dp_tp_conv_node_fused(bad, 2, st) == -2
    assert L.dp_tp_conv_node_fused(bad, 4, st) == -1
