"""DP_PREC_BF16 ("bf16 mixed precision", BASELINE.json configs[4]): the bf16-operand node-side conv (dp_conv_node_bf16_impl.h) and the
one-product edge-side kernel, through the C ABI.

Two references per case:
  * tests/bf16_model.py - the kernel's SPECIFICATION (numpy float64 with bf16 roundings exactly where the kernel rounds).  The kernel may
    differ from it by fp32 summation order and by a handful of values that sit on a bf16 rounding boundary: <= 1e-3 of the output's largest
    magnitude (measured ~1e-4).  This is the tight check of the implementation: layouts, fragments, masks, passes, fusion.
  * the fp64 oracle (oracle/e3.py: the reference operator, /root/reference/src/models/score_model_phore.py:134-149).  bf16 operands carry
    8 significand bits (2^-9 relative rounding each): <= 2e-2 of the output's largest magnitude (measured 2-6e-3).  This is the stated
    precision of the mode, not a bug bound."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bf16_model import node_conv_bf16  # noqa: E402
from diffphore_amd import _lib, packing, weights  # noqa: E402
from test_conv_gpu import _dev, _i32, _oracle_conv, _run  # noqa: E402

TOL_MODEL, TOL_ORACLE = 1e-3, 2e-2


def _emb_bf(emb_dev):
    """[n + 1, 32] bf16 image (+ flag row) of an edge-embedding table, made by the library's own conversion kernel"""
    n = emb_dev.shape[0]
    out = torch.zeros((n + 1) * 32, dtype=torch.int16, device="cuda")
    _lib.check(_lib.lib().dp_edge_emb_bf16(emb_dev.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "dp_edge_emb_bf16")
    return out, n


def test_edge_emb_bf16_rows():
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(37, 20, generator=g)
    out, n = _emb_bf(_dev(emb))
    got = out.cpu().view(n + 1, 32)
    want = emb.to(torch.bfloat16).view(torch.int16)
    assert torch.equal(got[:n, :20], want) and int(got[:n, 20:].abs().sum()) == 0
    flag = torch.zeros(32, dtype=torch.int16)
    flag[20] = 0x3F80
    assert torch.equal(got[n], flag)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["p2l", "l2p"])
def test_conv_node_bf16_dense_cross(kind, mode):
    """ragged poses: 1 node, < 16, > 16 nodes; 5 / 17 / 32 / 33 / 79 edges per node = full, half and several passes; residual + accumulate"""
    g = torch.Generator().manual_seed(1700 + kind)
    sd = weights.init_state_dict(11)
    fam = "phore_to_lig_conv_layers" if mode == "p2l" else "lig_to_phore_norm_conv_layers"
    if kind == 3 and mode == "l2p":
        fam = "lig_to_phore_conv_layers"
    prefix = f"encoder.{fam}.{kind}"
    dims = _lib.conv_dims(kind)
    NL, NP = [7, 30, 13, 1, 40, 17], [32, 5, 79, 33, 17, 48]
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
    slot_p2l = torch.cat([torch.arange(NP[b]).repeat(NL[b]) for b in range(B)])
    slot_l2p = torch.cat([torch.arange(NL[b]).repeat_interleave(NP[b]) for b in range(B)])
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_BF16, node_bf16=True)
    if mode == "p2l":
        ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, hl.shape[0])
        model = node_conv_bf16(kind, pk["w1n"].numpy(), pk["w2n"].numpy(), pk["bn_shift"].numpy(), emb.numpy(), hl[src, :20].numpy(), hp[dst, :20].numpy(),
                               hp[dst].numpy(), sh.numpy(), src.numpy(), hl.shape[0], s_img=1, slot=slot_p2l.numpy())
        n_out, gt, scat_ptr, m, mx = hl.shape[0], hp, lig_ptr, _lib.EDGES_P2L, max(NL)
    else:
        ref = _oracle_conv(sd, prefix, kind, hl[src], attr, sh, dst, hp.shape[0])
        model = node_conv_bf16(kind, pk["w1n"].numpy(), pk["w2n"].numpy(), pk["bn_shift"].numpy(), emb.numpy(), hp[dst, :20].numpy(), hl[src, :20].numpy(),
                               hl[src].numpy(), sh.numpy(), dst.numpy(), hp.shape[0], s_img=2, slot=slot_l2p.numpy())
        n_out, gt, scat_ptr, m, mx = hp.shape[0], hl, ph_ptr, _lib.EDGES_L2P, max(NP)
    nb = _lib.lib().dp_conv_node_max_nodes()
    resid = torch.randn(n_out, dims["d_in"], generator=g)
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    keep = []
    emb_d = _dev(emb)
    embbf, flag_row = _emb_bf(emb_d)
    base = dict(kind=kind, mode=m, B=B, slices=(mx + nb - 1) // nb, precision=_lib.PREC_BF16, max_scatter=mx, scat_ptr=_dev(_i32(scat_ptr)),
                lig_ptr=_dev(_i32(lig_ptr)), ph_ptr=_dev(_i32(ph_ptr)), x_ptr=_dev(_i32(x_ptr)), gather_tab=_dev(gt), gather_stride=dims["d_in"],
                attr1_tab=_dev(hl), attr1_stride=dims["d_in"], attr2_tab=_dev(hp), attr2_stride=dims["d_in"], attr_swap=0,
                edge_emb=emb_d, edge_emb_bf=embbf, emb_flag_row=flag_row, edge_sh=_dev(sh), sh_stride=9, w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]),
                w1n=_dev(pk["w1n"]), w2n=_dev(pk["w2n"]), bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]), out=out, out_stride=dims["d_out"],
                accumulate=0, resid=_dev(resid), resid_stride=dims["d_in"], resid_width=dims["d_in"], algo=_lib.ALGO_NODE)
    _run(base, keep)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    want_m = torch.from_numpy(model).clone()
    want_m[:, :dims["d_in"]] += resid.double()
    want_o = ref.clone()
    want_o[:, :dims["d_in"]] += resid.double()
    scale = ref.abs().max().item()
    err_m = (got - want_m).abs().max().item() / scale
    err_o = (got - want_o).abs().max().item() / scale
    print(f"kind {kind} {mode}: kernel vs bf16 model {err_m:.2e}, vs fp64 oracle {err_o:.2e}")
    assert err_m < TOL_MODEL, err_m
    assert err_o < TOL_ORACLE, err_o
    base["accumulate"] = 1
    _run(base, keep)
    err = (out.cpu().double() - (want_m + torch.from_numpy(model))).abs().max().item() / scale
    assert err < 2 * TOL_MODEL, err
    # reruns are bit-identical (fixed summation order, no atomics)
    out3 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    base.update(accumulate=0, out=out3)
    _run(base, keep)
    assert torch.equal(out3.cpu().double(), got)
    # the one-product edge-side kernel (DP_PREC_BF16 on DP_ALGO_EDGE) on the same inputs: bf16-grade as well
    pe = packing.pack_conv(sd, prefix, kind, _lib.PREC_BF16)
    out2 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    base.update(algo=_lib.ALGO_EDGE, out=out2, slices=3, w1p=_dev(pe["w1p"]), w2p=_dev(pe["w2p"]))
    _run(base, keep)
    err = (out2.cpu().double() - want_o).abs().max().item() / scale
    print(f"    edge-side one-product bf16 vs fp64 oracle {err:.2e}")
    assert err < TOL_ORACLE, err


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_conv_node_bf16_ligand_pairs(kind):
    """pair-dense ligand-ligand mode: radius + bond edges on one pair add their hidden vectors; atoms without edges; <= 32 and > 32 atoms"""
    g = torch.Generator().manual_seed(1900 + kind)
    sd = weights.init_state_dict(11)
    prefix = f"encoder.lig_conv_layers.{kind}"
    dims = _lib.conv_dims(kind)
    NL = [9, 30, 1, 45, 17]
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    N = int(lig_ptr[-1])
    h = torch.randn(N, dims["d_in"], generator=g)
    rowptr, rowend, nbr, scat, pair_rad, pair_bond = [], [], [], [], [], []
    pair_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) ** 2)])
    # per ordered pair (for the model): radius edge slot / bond edge slot
    P_scat, P_gath, P_rad, P_bond, P_slot = [], [], [], [], []
    e0 = 0
    for b in range(B):
        n = NL[b]
        adj = (torch.rand(n, n, generator=g) < 0.6).numpy()
        np.fill_diagonal(adj, False)
        bond = (torch.rand(n, n, generator=g) < 0.08).numpy() & ~np.eye(n, dtype=bool)
        adj[3 % n] = False
        if n > 5:
            adj[5], bond[5] = False, False
        pr, pb_ = -np.ones((n, n), dtype=np.int32), -np.ones((n, n), dtype=np.int32)
        for a_ in range(n):
            rowptr.append(e0)
            for y in np.nonzero(bond[a_])[0]:
                pb_[a_, y] = e0
                nbr.append(int(lig_ptr[b]) + int(y)); scat.append(int(lig_ptr[b]) + a_)
                e0 += 1
            for y in np.nonzero(adj[a_])[0]:
                pr[a_, y] = e0
                nbr.append(int(lig_ptr[b]) + int(y)); scat.append(int(lig_ptr[b]) + a_)
                e0 += 1
            rowend.append(e0)
            for y in range(n):
                if pr[a_, y] >= 0 or pb_[a_, y] >= 0:
                    P_scat.append(int(lig_ptr[b]) + a_); P_gath.append(int(lig_ptr[b]) + y); P_rad.append(pr[a_, y]); P_bond.append(pb_[a_, y]); P_slot.append(y)
        pair_rad.append(pr.reshape(-1)); pair_bond.append(pb_.reshape(-1))
    E = e0
    nbr_t, scat_t = torch.tensor(nbr), torch.tensor(scat)
    pos = torch.randn(N, 3, generator=g)
    from oracle import e3
    sh = e3.spherical_harmonics_lmax2((pos[nbr_t] - pos[scat_t]).double()).float()
    emb = torch.randn(E, 20, generator=g)
    attr = torch.cat([emb, h[scat_t, :20], h[nbr_t, :20]], 1)
    ref = _oracle_conv(sd, prefix, kind, h[nbr_t], attr, sh, scat_t, N)
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_BF16, node_bf16=True)
    # the model works per PAIR: first edge = the radius edge if present, else the bond edge; second = the bond edge when both exist
    P_scat, P_gath, P_rad, P_bond, P_slot = map(np.asarray, (P_scat, P_gath, P_rad, P_bond, P_slot))
    first = np.where(P_rad >= 0, P_rad, P_bond)
    both = (P_rad >= 0) & (P_bond >= 0)
    second = np.where(both, P_bond, 0)
    model = node_conv_bf16(kind, pk["w1n"].numpy(), pk["w2n"].numpy(), pk["bn_shift"].numpy(), emb.numpy()[first], h.numpy()[P_scat, :20], h.numpy()[P_gath, :20],
                           h.numpy()[P_gath], sh.numpy()[first], P_scat, N, s_img=1, extra_hid=(emb.numpy()[second], both), slot=P_slot)
    nb = _lib.lib().dp_conv_node_max_nodes()
    resid = torch.randn(N, dims["d_in"], generator=g)
    out = torch.full((N, dims["d_out"]), float("nan")).cuda()
    keep = []
    emb_d = _dev(emb)
    embbf, flag_row = _emb_bf(emb_d)
    common = dict(kind=kind, B=B, precision=_lib.PREC_BF16, max_scatter=max(NL), scat_ptr=_dev(_i32(lig_ptr)), lig_ptr=_dev(_i32(lig_ptr)),
                  rowptr=_dev(_i32(rowptr)), rowend=_dev(_i32(rowend)), nbr=_dev(_i32(nbr)), edge_scat=_dev(_i32(scat)),
                  gather_tab=_dev(h), gather_stride=dims["d_in"], attr1_tab=_dev(h), attr1_stride=dims["d_in"], attr2_tab=_dev(h),
                  attr2_stride=dims["d_in"], attr_swap=0, edge_emb=emb_d, edge_emb_bf=embbf, emb_flag_row=flag_row, edge_sh=_dev(sh), sh_stride=9,
                  w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]), w1n=_dev(pk["w1n"]), w2n=_dev(pk["w2n"]), bn_scale=_dev(pk["bn_scale"]),
                  bn_shift=_dev(pk["bn_shift"]), out_stride=dims["d_out"], accumulate=0, resid=_dev(resid), resid_stride=dims["d_in"],
                  resid_width=dims["d_in"], pair_a=_dev(_i32(np.concatenate(pair_rad))), pair_b=_dev(_i32(np.concatenate(pair_bond))),
                  pair_ptr=_dev(_i32(pair_ptr)))
    _run(dict(common, mode=_lib.EDGES_LLPAIR, algo=_lib.ALGO_NODE, slices=(max(NL) + nb - 1) // nb, out=out), keep)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    want_m = torch.from_numpy(model).clone()
    want_m[:, :dims["d_in"]] += resid.double()
    want_o = ref.clone()
    want_o[:, :dims["d_in"]] += resid.double()
    err_m = (got - want_m).abs().max().item() / scale
    err_o = (got - want_o).abs().max().item() / scale
    print(f"kind {kind} ligand pairs: kernel vs bf16 model {err_m:.2e}, vs fp64 oracle {err_o:.2e}")
    assert err_m < TOL_MODEL, err_m
    assert err_o < TOL_ORACLE, err_o


@pytest.mark.parametrize("kind,mode", [(0, "p2l"), (2, "p2l"), (3, "p2l"), (1, "l2p"), (2, "l2p")])
def test_conv_node_bf16_fused_equals_separate_launches(kind, mode):
    """dp_tp_conv_node_fused with two bf16 cross convs == the same two convs launched one after the other (accumulate): the operands of
    every MFMA are the same, only the fp32 summation order of the final adds differs"""
    g = torch.Generator().manual_seed(2100 + kind)
    sd = weights.init_state_dict(11)
    fams = ("phore_to_lig_conv_layers", "phore_to_lig_norm_conv_layers") if mode == "p2l" else ("lig_to_phore_conv_layers", "lig_to_phore_norm_conv_layers")
    dims = _lib.conv_dims(kind)
    NL, NP = [20, 33, 5], [40, 16, 79]
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    ph_ptr = np.concatenate([[0], np.cumsum(NP)])
    x_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) * np.asarray(NP))])
    hl = torch.randn(int(lig_ptr[-1]), dims["d_in"], generator=g)
    hp = torch.randn(int(ph_ptr[-1]), dims["d_in"], generator=g)
    E = int(x_ptr[-1])
    emb = torch.randn(E, 20, generator=g)
    shs = [torch.randn(E, 9, generator=g) for _ in range(2)]
    for s in shs:
        s[:, 0] = 1.0
    n_out, gt, scat_ptr, m, mx = ((hl.shape[0], hp, lig_ptr, _lib.EDGES_P2L, max(NL)) if mode == "p2l" else (hp.shape[0], hl, ph_ptr, _lib.EDGES_L2P, max(NP)))
    nb = _lib.lib().dp_conv_node_max_nodes()
    keep = []
    emb_d = _dev(emb)
    embbf, flag_row = _emb_bf(emb_d)
    resid = torch.randn(n_out, dims["d_in"], generator=g)
    shared = dict(kind=kind, mode=m, B=B, slices=(mx + nb - 1) // nb, precision=_lib.PREC_BF16, max_scatter=mx, scat_ptr=_dev(_i32(scat_ptr)),
                  lig_ptr=_dev(_i32(lig_ptr)), ph_ptr=_dev(_i32(ph_ptr)), x_ptr=_dev(_i32(x_ptr)), gather_tab=_dev(gt), gather_stride=dims["d_in"],
                  attr1_tab=_dev(hl), attr1_stride=dims["d_in"], attr2_tab=_dev(hp), attr2_stride=dims["d_in"], attr_swap=0, edge_emb=emb_d,
                  edge_emb_bf=embbf, emb_flag_row=flag_row, sh_stride=9, out_stride=dims["d_out"], resid=_dev(resid), resid_stride=dims["d_in"],
                  resid_width=dims["d_in"], algo=_lib.ALGO_NODE)
    convs = []
    for i, fam in enumerate(fams):
        pk = packing.pack_conv(sd, f"encoder.{fam}.{kind}", kind, _lib.PREC_BF16, node_bf16=True)
        convs.append(dict(shared, edge_sh=_dev(shs[i]), w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]), w1n=_dev(pk["w1n"]), w2n=_dev(pk["w2n"]),
                          bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"])))
    out_a = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    _run(dict(convs[0], out=out_a, accumulate=0), keep)
    _run(dict(convs[1], out=out_a, accumulate=1), keep)
    out_b = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    arr = (_lib.DpConvArgs * 2)()
    for i in range(2):
        for k_, v in dict(convs[i], out=out_b, accumulate=0).items():
            if torch.is_tensor(v):
                keep.append(v)
                v = v.data_ptr()
            setattr(arr[i], k_, v)
    _lib.check(_lib.lib().dp_tp_conv_node_fused(arr, 2, torch.cuda.current_stream().cuda_stream), "dp_tp_conv_node_fused")
    torch.cuda.synchronize()
    a, b_ = out_a.cpu().double(), out_b.cpu().double()
    assert torch.isfinite(b_).all()
    err = (a - b_).abs().max().item() / a.abs().max().item()
    assert err < 1e-5, err
    # conv split (small batches): every conv on its own workgroups + the combine launch - the same MFMA operands per conv, the convs' outputs
    # added in fp32 by dp_conv_node_combine instead of inside the accumulators; deterministic (bit-identical reruns)
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    part = torch.zeros(2 * n_out * dims["d_out"], device="cuda")
    outs = []
    for rep in range(2):
        out_c = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
        for i in range(2):
            arr[i].out, arr[i].path_split = out_c.data_ptr(), 1
        _lib.check(L.dp_tp_conv_node_fused_split(arr, 2, n_out, part.data_ptr(), part.numel(), st), "dp_tp_conv_node_fused_split (bf16)")
        torch.cuda.synchronize()
        outs.append(out_c.cpu())
    assert torch.equal(outs[0], outs[1])
    c_ = outs[0].double()
    assert torch.isfinite(c_).all()
    err_s = (c_ - b_).abs().max().item() / b_.abs().max().item()
    print(f"kind {kind} {mode}: bf16 conv split vs fused launch {err_s:.1e}")
    assert err_s < 1e-5, err_s
