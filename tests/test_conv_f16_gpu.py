"""DP_PREC_F16X3: the fp32-GRADE node-side conv on the 16-bit matrix pipe (dp_conv_node_f16_impl.h: every GEMM operand as two fp16 limbs,
cross products on v_mfma_f32_16x16x32_f16, fp32 accumulate), through the C ABI, against the fp64 oracle (oracle/e3.py: the reference
operator, /root/reference/src/models/score_model_phore.py:134-149) at the SAME tolerance as the exact-fp32 kernels of tests/test_conv_gpu.py:
2e-5 of the output's largest magnitude."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffphore_amd import _lib, packing, weights  # noqa: E402
from test_conv_gpu import _dev, _i32, _oracle_conv, _run  # noqa: E402

TOL = 2e-5


def _emb_h(emb_dev):
    """[n + 1, 64] two-limb fp16 image (+ flag row) of an edge-embedding table, made by the library's own conversion kernel"""
    n = emb_dev.shape[0]
    out = torch.zeros((n + 1) * 64, dtype=torch.int16, device="cuda")
    _lib.check(_lib.lib().dp_edge_emb_f16(emb_dev.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "dp_edge_emb_f16")
    return out, n


def test_edge_emb_f16_rows():
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(37, 20, generator=g) * torch.logspace(-6, 3, 37).unsqueeze(1)       # 1e-6 .. 1e3: normal, subnormal and large limbs
    out, n = _emb_h(_dev(emb))
    got = out.cpu().view(n + 1, 64)
    hi = emb.to(torch.float16)
    lo = (emb - hi.float()).to(torch.float16)
    assert torch.equal(got[:n, :20], hi.view(torch.int16)) and torch.equal(got[:n, 32:52], lo.view(torch.int16))
    assert int(got[:n, 20:32].abs().sum()) == 0 and int(got[:n, 52:].abs().sum()) == 0
    flag = torch.zeros(64, dtype=torch.int16)
    flag[20] = 0x3C00
    assert torch.equal(got[n], flag)
    # two limbs carry >= 21 significant bits; where the lo limb is a subnormal fp16 the error is absolute: half of 2^-24
    rec = hi.double() + lo.double()
    assert ((rec - emb.double()).abs() <= torch.clamp(emb.double().abs() * 2.0 ** -21, min=2.0 ** -25)).all()


def _cross_case(kind, mode, seed, NL, NP, feat_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    sd = weights.init_state_dict(11)
    fam = "phore_to_lig_conv_layers" if mode == "p2l" else "lig_to_phore_norm_conv_layers"
    if kind == 3 and mode == "l2p":
        fam = "lig_to_phore_conv_layers"
    prefix = f"encoder.{fam}.{kind}"
    dims = _lib.conv_dims(kind)
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    ph_ptr = np.concatenate([[0], np.cumsum(NP)])
    x_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) * np.asarray(NP))])
    hl = torch.randn(int(lig_ptr[-1]), dims["d_in"], generator=g) * feat_scale
    hp = torch.randn(int(ph_ptr[-1]), dims["d_in"], generator=g) * feat_scale
    E = int(x_ptr[-1])
    emb = torch.randn(E, 20, generator=g)
    sh = torch.randn(E, 9, generator=g)
    sh[:, 0] = 1.0
    src = torch.cat([torch.arange(lig_ptr[b], lig_ptr[b + 1]).repeat_interleave(NP[b]) for b in range(B)])
    dst = torch.cat([torch.arange(ph_ptr[b], ph_ptr[b + 1]).repeat(NL[b]) for b in range(B)])
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_F32, node_f16=True)
    if mode == "p2l":
        ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, hl.shape[0])
        n_out, gt, scat_ptr, m, mx = hl.shape[0], hp, lig_ptr, _lib.EDGES_P2L, max(NL)
    else:
        ref = _oracle_conv(sd, prefix, kind, hl[src], attr, sh, dst, hp.shape[0])
        n_out, gt, scat_ptr, m, mx = hp.shape[0], hl, ph_ptr, _lib.EDGES_L2P, max(NP)
    nb = _lib.lib().dp_conv_node_max_nodes()
    resid = torch.randn(n_out, dims["d_in"], generator=g)
    emb_d = _dev(emb)
    embh, flag_row = _emb_h(emb_d)
    base = dict(kind=kind, mode=m, B=B, slices=(mx + nb - 1) // nb, precision=_lib.PREC_F16X3, max_scatter=mx, scat_ptr=_dev(_i32(scat_ptr)),
                lig_ptr=_dev(_i32(lig_ptr)), ph_ptr=_dev(_i32(ph_ptr)), x_ptr=_dev(_i32(x_ptr)), gather_tab=_dev(gt), gather_stride=dims["d_in"],
                attr1_tab=_dev(hl), attr1_stride=dims["d_in"], attr2_tab=_dev(hp), attr2_stride=dims["d_in"], attr_swap=0,
                edge_emb=emb_d, edge_emb_bf=embh, emb_flag_row=flag_row, edge_sh=_dev(sh), sh_stride=9, w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]),
                w1n=_dev(pk["w1h"]), w2n=_dev(pk["w2h"]), bn_scale=_dev(pk["bn_scale"]), bn_shift=_dev(pk["bn_shift"]), out_stride=dims["d_out"],
                accumulate=0, resid=_dev(resid), resid_stride=dims["d_in"], resid_width=dims["d_in"], algo=_lib.ALGO_NODE)
    return base, ref, resid, dims, n_out, pk


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["p2l", "l2p"])
def test_conv_node_f16_dense_cross(kind, mode):
    """ragged poses: 1 node, < 8, 8..16, > 16 nodes; 5 / 17 / 32 / 33 / 79 edges per node = full, half and several passes; residual + accumulate"""
    base, ref, resid, dims, n_out, _ = _cross_case(kind, mode, 1700 + kind, [7, 30, 13, 1, 40, 17], [32, 5, 79, 33, 17, 48])
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    keep = []
    _run(dict(base, out=out), keep)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    scale = ref.abs().max().item()
    err = (got - want).abs().max().item() / scale
    print(f"kind {kind} {mode}: f16x3 node-side kernel vs fp64 oracle {err:.2e}")
    assert err < TOL, err
    _run(dict(base, out=out, accumulate=1), keep)
    err = (out.cpu().double() - (want + ref)).abs().max().item() / scale
    assert err < 2 * TOL, err
    # reruns are bit-identical (fixed summation order, no atomics)
    out3 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    _run(dict(base, out=out3), keep)
    assert torch.equal(out3.cpu().double(), got)


@pytest.mark.parametrize("scale", [1e-3, 30.0])
def test_conv_node_f16_feature_magnitudes(scale):
    """features far from O(1): the limbs of small values keep an absolute precision of 2^-25, large ones stay below fp16's 65504"""
    base, ref, resid, dims, n_out, _ = _cross_case(3, "p2l", 77, [30, 9], [32, 20], feat_scale=scale)
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    keep = []
    _run(dict(base, out=out, resid=None, resid_width=0), keep)
    got = out.cpu().double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"feature scale {scale}: {err:.2e}")
    assert err < TOL, err


def test_conv_node_f16_neighbour_nan_does_not_leak():
    """a pose whose features are NaN (a diverged ligand in a screening batch) leaves the poses packed around it untouched: gathers past a pose's
    last node read 0 through the pose-limited buffer resource, never the next pose's rows"""
    base, ref, resid, dims, n_out, _ = _cross_case(2, "p2l", 78, [12, 20, 9], [20, 17, 40])
    gt = base["gather_tab"].clone()
    ph_ptr = base["ph_ptr"].cpu().numpy()
    gt[ph_ptr[1]:ph_ptr[2]] = float("nan")                       # pose 1's pharmacophore features
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    keep = []
    _run(dict(base, out=out, gather_tab=gt, attr2_tab=gt), keep)
    got = out.cpu().double()
    lig_ptr = base["lig_ptr"].cpu().numpy()
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    good = np.r_[0:lig_ptr[1], lig_ptr[2]:lig_ptr[3]]
    assert torch.isfinite(got[good]).all()
    assert (got[good] - want[good]).abs().max().item() / ref.abs().max().item() < TOL
    assert not torch.isfinite(got[lig_ptr[1]:lig_ptr[2]]).all()


def test_conv_node_f16_out_of_range_features_fail_loudly_and_locally():
    """the kernel's precondition: |gathered features| < 65504 (fp16's range; BatchNorm keeps real features at O(1-10)).  A value beyond it becomes
    inf in the hi limb and NaN in the products: the pose that holds it comes out non-finite - never a silently wrong finite number - and the
    poses packed around it are untouched (the status / screen() failure path then reports that pose)"""
    base, ref, resid, dims, n_out, _ = _cross_case(1, "p2l", 79, [12, 20, 9], [20, 17, 40])
    gt = base["gather_tab"].clone()
    ph_ptr = base["ph_ptr"].cpu().numpy()
    gt[ph_ptr[1] + 3, 25] = 1.0e5                                 # one vector component of one pharmacophore node of pose 1
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    keep = []
    _run(dict(base, out=out, gather_tab=gt), keep)
    got = out.cpu().double()
    lig_ptr = base["lig_ptr"].cpu().numpy()
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    good = np.r_[0:lig_ptr[1], lig_ptr[2]:lig_ptr[3]]
    assert torch.isfinite(got[good]).all()
    assert (got[good] - want[good]).abs().max().item() / ref.abs().max().item() < TOL
    bad = got[lig_ptr[1]:lig_ptr[2]]
    assert not torch.isfinite(bad).all(dim=1).any(), "every atom of the pose gathers the out-of-range node (dense cross edges): all of its rows are flagged"


def _ll_case(kind, seed, NL):
    g = torch.Generator().manual_seed(seed)
    sd = weights.init_state_dict(11)
    prefix = f"encoder.lig_conv_layers.{kind}"
    dims = _lib.conv_dims(kind)
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    N = int(lig_ptr[-1])
    h = torch.randn(N, dims["d_in"], generator=g)
    rowptr, rowend, nbr, scat, pair_rad, pair_bond = [], [], [], [], [], []
    pair_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) ** 2)])
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
        pair_rad.append(pr.reshape(-1)); pair_bond.append(pb_.reshape(-1))
    E = e0
    nbr_t, scat_t = torch.tensor(nbr), torch.tensor(scat)
    pos = torch.randn(N, 3, generator=g)
    from oracle import e3
    sh = e3.spherical_harmonics_lmax2((pos[nbr_t] - pos[scat_t]).double()).float()
    emb = torch.randn(E, 20, generator=g)
    attr = torch.cat([emb, h[scat_t, :20], h[nbr_t, :20]], 1)
    ref = _oracle_conv(sd, prefix, kind, h[nbr_t], attr, sh, scat_t, N)
    pk = packing.pack_conv(sd, prefix, kind, _lib.PREC_F32, node_f16=True)
    nb = _lib.lib().dp_conv_node_max_nodes()
    resid = torch.randn(N, dims["d_in"], generator=g)
    emb_d = _dev(emb)
    embh, flag_row = _emb_h(emb_d)
    args = dict(kind=kind, B=B, precision=_lib.PREC_F16X3, max_scatter=max(NL), scat_ptr=_dev(_i32(lig_ptr)), lig_ptr=_dev(_i32(lig_ptr)),
                rowptr=_dev(_i32(rowptr)), rowend=_dev(_i32(rowend)), nbr=_dev(_i32(nbr)), edge_scat=_dev(_i32(scat)),
                gather_tab=_dev(h), gather_stride=dims["d_in"], attr1_tab=_dev(h), attr1_stride=dims["d_in"], attr2_tab=_dev(h),
                attr2_stride=dims["d_in"], attr_swap=0, edge_emb=emb_d, edge_emb_bf=embh, emb_flag_row=flag_row, edge_sh=_dev(sh), sh_stride=9,
                w1p=_dev(pk["w1p"]), w2p=_dev(pk["w2p"]), w1n=_dev(pk["w1h"]), w2n=_dev(pk["w2h"]), bn_scale=_dev(pk["bn_scale"]),
                bn_shift=_dev(pk["bn_shift"]), out_stride=dims["d_out"], accumulate=0, resid=_dev(resid), resid_stride=dims["d_in"],
                resid_width=dims["d_in"], pair_a=_dev(_i32(np.concatenate(pair_rad))), pair_b=_dev(_i32(np.concatenate(pair_bond))),
                pair_ptr=_dev(_i32(pair_ptr)), mode=_lib.EDGES_LLPAIR, algo=_lib.ALGO_NODE, slices=(max(NL) + nb - 1) // nb)
    return args, ref, resid, dims, N, h


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_conv_node_f16_ligand_pairs(kind):
    """pair-dense ligand-ligand mode: radius + bond edges on one pair add their hidden vectors; atoms without edges; <= 32 and > 32 atoms"""
    args, ref, resid, dims, N, _ = _ll_case(kind, 1900 + kind, [9, 30, 1, 45, 17])
    out = torch.full((N, dims["d_out"]), float("nan")).cuda()
    keep = []
    _run(dict(args, out=out), keep)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    err = (got - want).abs().max().item() / ref.abs().max().item()
    print(f"kind {kind} ligand pairs: f16x3 vs fp64 oracle {err:.2e}")
    assert err < TOL, err


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_conv_node_f16_fused_ligand_side(kind):
    """dp_tp_conv_node_fused: ligand-ligand + phore->ligand + phore->ligand(normals) in one launch (three W2 images with their own exponents, one
    common accumulator scale) == the sum of the three oracle convs (score_model_phore.py:703-706)"""
    NL, NP = [9, 30, 1, 24, 17], [32, 5, 40, 33, 16]
    ll, ref_ll, resid, dims, N, h = _ll_case(kind, 2300 + kind, NL)
    g = torch.Generator().manual_seed(2400 + kind)
    sd = weights.init_state_dict(11)
    B = len(NL)
    lig_ptr = np.concatenate([[0], np.cumsum(NL)])
    ph_ptr = np.concatenate([[0], np.cumsum(NP)])
    x_ptr = np.concatenate([[0], np.cumsum(np.asarray(NL) * np.asarray(NP))])
    hp = torch.randn(int(ph_ptr[-1]), dims["d_in"], generator=g)
    E = int(x_ptr[-1])
    emb = torch.randn(E, 20, generator=g)
    src = torch.cat([torch.arange(lig_ptr[b], lig_ptr[b + 1]).repeat_interleave(NP[b]) for b in range(B)])
    dst = torch.cat([torch.arange(ph_ptr[b], ph_ptr[b + 1]).repeat(NL[b]) for b in range(B)])
    attr = torch.cat([emb, h[src, :20], hp[dst, :20]], 1)
    emb_d = _dev(emb)
    embh, flag_row = _emb_h(emb_d)
    keep = []
    convs, want = [ll], ref_ll.clone()
    for i, fam in enumerate(("phore_to_lig_conv_layers", "phore_to_lig_norm_conv_layers")):
        sh = torch.randn(E, 9, generator=g)
        sh[:, 0] = 1.0
        prefix = f"encoder.{fam}.{kind}"
        # different magnitudes of the second layer per conv: the three images get different exponents
        sd2 = dict(sd)
        sd2[prefix + ".fc.3.weight"] = sd[prefix + ".fc.3.weight"] * (0.1 if i == 0 else 7.0)
        pk = packing.pack_conv(sd2, prefix, kind, _lib.PREC_F32, node_f16=True)
        want += _oracle_conv(sd2, prefix, kind, hp[dst], attr, sh, src, N)
        convs.append(dict(ll, mode=_lib.EDGES_P2L, ph_ptr=_dev(_i32(ph_ptr)), x_ptr=_dev(_i32(x_ptr)), gather_tab=_dev(hp), attr2_tab=_dev(hp),
                          edge_emb=emb_d, edge_emb_bf=embh, emb_flag_row=flag_row, edge_sh=_dev(sh), w1n=_dev(pk["w1h"]), w2n=_dev(pk["w2h"]),
                          bn_shift=_dev(pk["bn_shift"]), pair_a=None, pair_b=None, pair_ptr=None))
    out = torch.full((N, dims["d_out"]), float("nan")).cuda()
    arr = (_lib.DpConvArgs * 3)()
    for i in range(3):
        for k_, v in dict(convs[i], out=out, accumulate=0).items():
            if torch.is_tensor(v):
                keep.append(v)
                v = v.data_ptr()
            setattr(arr[i], k_, v)
    _lib.check(_lib.lib().dp_tp_conv_node_fused(arr, 3, torch.cuda.current_stream().cuda_stream), "dp_tp_conv_node_fused")
    torch.cuda.synchronize()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    scale = want.abs().max().item()
    want[:, :dims["d_in"]] += resid.double()
    err = (got - want).abs().max().item() / scale
    print(f"kind {kind} fused ligand side: f16x3 vs fp64 oracle {err:.2e}")
    assert err < TOL, err
    # conv split (the small-batch form: one conv per workgroup, dp_conv_node_combine adds them up): the same limb products per conv, the three
    # un-scaled sums added in fp32 outside the accumulators; bit-identical reruns; the convs keep their different exponents
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    part = torch.zeros(3 * N * dims["d_out"], device="cuda")
    outs = []
    for rep in range(2):
        out_s = torch.full((N, dims["d_out"]), float("nan")).cuda()
        for i in range(3):
            arr[i].out, arr[i].path_split = out_s.data_ptr(), 1
        _lib.check(L.dp_tp_conv_node_fused_split(arr, 3, N, part.data_ptr(), part.numel(), st), "dp_tp_conv_node_fused_split (f16x3)")
        torch.cuda.synchronize()
        outs.append(out_s.cpu())
    assert torch.equal(outs[0], outs[1])
    err_s = (outs[0].double() - want).abs().max().item() / scale
    err_f = (outs[0].double() - got).abs().max().item() / scale
    print(f"kind {kind} ligand side, conv split: vs fp64 oracle {err_s:.2e}, vs the fused launch {err_f:.2e}")
    assert err_s < TOL and err_f < 1e-6, (err_s, err_f)


@pytest.mark.parametrize("kind,mode", [(1, "p2l"), (3, "p2l"), (2, "l2p")])
def test_conv_node_f16_lean_instantiation_is_bit_identical(kind, mode):
    """DpConvArgs.no_tail = 1 (no pose ends on a pass of <= 16 edge slots) selects the instantiation without the half-pass variant (half the
    code, ~40 % fewer spilled registers, +2.8 % on the configs[1] step); on such a batch it must equal the general kernel bit for bit"""
    base, ref, resid, dims, n_out, _ = _cross_case(kind, mode, 2600 + kind, [30, 17, 32, 49], [32, 64, 30, 51])      # atoms and phore nodes per pose: both mod 32 in {0, 17..31}
    keep = []
    out_g = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    _run(dict(base, out=out_g, no_tail=0), keep)
    out_l = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    _run(dict(base, out=out_l, no_tail=1), keep)
    assert torch.isfinite(out_g).all() and torch.equal(out_g, out_l)
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    assert (out_l.cpu().double() - want).abs().max().item() / ref.abs().max().item() < TOL


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["p2l", "l2p"])
def test_edge_side_f16x3_vs_oracle(kind, mode):
    """DP_PREC_F16X3 on the EDGE-side kernel (dp_conv.hip, round 5: what the phore-phore / torsion / large-ligand convs of an fp32 model run
    instead of the 3-way bf16 split): both operands of the radial contraction as two fp16 limbs, 3 cross products; weights `w2x` = W2 * 2^w and
    the BatchNorm scale `bnx` = bn_scale * 2^-w from packing.pack_conv(x6=True).  Same tolerance as the exact-fp32 kernel."""
    base, ref, resid, dims, n_out, _ = _cross_case(kind, mode, 2800 + kind, [7, 30, 13, 1, 40, 17], [32, 5, 79, 33, 17, 48])
    sd = weights.init_state_dict(11)
    fam = "phore_to_lig_conv_layers" if mode == "p2l" else "lig_to_phore_norm_conv_layers"
    if kind == 3 and mode == "l2p":
        fam = "lig_to_phore_conv_layers"
    pk = packing.pack_conv(sd, f"encoder.{fam}.{kind}", kind, _lib.PREC_F32, x6=True)
    wexp = float(pk["w2x"][-4])
    assert torch.allclose(pk["bnx"].double() * 2.0 ** wexp, pk["bn_scale"].double(), rtol=1e-7)
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    keep = []
    args = dict(base, algo=_lib.ALGO_EDGE, precision=_lib.PREC_F16X3, slices=3, w2p=_dev(pk["w2x"]), bn_scale=_dev(pk["bnx"]), out=out)
    _run(args, keep)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    err = (got - want).abs().max().item() / ref.abs().max().item()
    print(f"kind {kind} {mode}: edge-side f16x3 vs fp64 oracle {err:.2e}")
    assert err < TOL, err
    # the 3-way bf16 split it replaces, on the same inputs
    pk6 = packing.pack_conv(sd, f"encoder.{fam}.{kind}", kind, _lib.PREC_BF16X6)
    out6 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    _run(dict(base, algo=_lib.ALGO_EDGE, precision=_lib.PREC_BF16X6, slices=3, w2p=_dev(pk6["w2p"]), bn_scale=_dev(pk6["bn_scale"]), out=out6), keep)
    err6 = (out6.cpu().double() - want).abs().max().item() / ref.abs().max().item()
    print(f"    bf16x6 on the same inputs {err6:.2e}")
    assert err6 < TOL


def _sparse_sh_case(kind, mode, seed, NL, NP, density):
    """a dense cross conv whose spherical harmonics are e0 = (1, 0, ..., 0) on all but a random `density` of the pairs (the normals convs of the
    model, score_model_phore.py:874-894), + the list of the other pairs by scatter node (what dp_pack_fill builds from the type agreement)"""
    base, _, resid, dims, n_out, pk = _cross_case(kind, mode, seed, NL, NP)
    g = torch.Generator().manual_seed(seed + 1)
    sh = base["edge_sh"].cpu()
    E = sh.shape[0]
    keep = torch.rand(E, generator=g) < density
    sh[~keep, 1:] = 0.0
    sd = weights.init_state_dict(11)
    fam = "phore_to_lig_conv_layers" if mode == "p2l" else "lig_to_phore_norm_conv_layers"
    if kind == 3 and mode == "l2p":
        fam = "lig_to_phore_conv_layers"
    prefix = f"encoder.{fam}.{kind}"
    hl, hp, emb = base["attr1_tab"].cpu(), base["attr2_tab"].cpu(), base["edge_emb"].cpu()
    B = len(NL)
    lig_ptr, ph_ptr, x_ptr = (base[k].cpu().numpy() for k in ("lig_ptr", "ph_ptr", "x_ptr"))
    src = torch.cat([torch.arange(lig_ptr[b], lig_ptr[b + 1]).repeat_interleave(NP[b]) for b in range(B)])
    dst = torch.cat([torch.arange(ph_ptr[b], ph_ptr[b + 1]).repeat(NL[b]) for b in range(B)])
    attr = torch.cat([emb, hl[src, :20], hp[dst, :20]], 1)
    if mode == "p2l":
        ref = _oracle_conv(sd, prefix, kind, hp[dst], attr, sh, src, hl.shape[0])
        scat, gath = src, dst
    else:
        ref = _oracle_conv(sd, prefix, kind, hl[src], attr, sh, dst, hp.shape[0])
        scat, gath = dst, src
    # the listed pairs, sorted by scatter node (stable: dense order inside a node), CSR over the scatter nodes
    idx = torch.nonzero(keep).flatten()
    order = torch.argsort(scat[idx], stable=True)
    idx = idx[order]
    rowptr = torch.zeros(n_out + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(scat[idx], minlength=n_out), 0)
    x6 = packing.pack_conv(sd, prefix, kind, _lib.PREC_F32, x6=True)
    lst = dict(edge_scat=_dev(_i32(scat[idx].numpy())), nbr=_dev(_i32(gath[idx].numpy())), feat_idx=_dev(_i32(idx.numpy())), n_list=int(idx.numel()),
               rowptr=_dev(_i32(rowptr.numpy())), w2x=_dev(x6["w2x"]), bnx=_dev(x6["bnx"]))
    return dict(base, edge_sh=_dev(sh)), ref, resid, dims, n_out, lst


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["p2l", "l2p"])
@pytest.mark.parametrize("fold", [True, False])
def test_s0_only_launch_plus_listed_pairs_equals_the_dense_conv(kind, mode, fold):
    """What dp_forward does with the two normals convs of a layer (DpWork.xn_sparse): the conv's l = 0 paths on every pair (node-side kernel,
    DpConvArgs.s0_only) + the other paths on the pairs whose harmonics are not e0 (DP_EDGES_LIST messages, folded into the s0-only launch's
    epilogue or finished by dp_list_reduce) - against the fp64 oracle's DENSE conv, at the dense kernels' tolerance."""
    NL, NP = [7, 30, 13, 1, 40, 17], [32, 5, 79, 33, 17, 48]
    base, ref, resid, dims, n_out, lst = _sparse_sh_case(kind, mode, 2300 + kind, NL, NP, 0.06)
    assert 0 < lst["n_list"] < 0.1 * base["edge_sh"].shape[0]
    keep = []
    st = torch.cuda.current_stream().cuda_stream
    msg = torch.full((lst["n_list"], dims["d_out"]), float("nan")).cuda()
    larg = dict(kind=kind, mode=_lib.EDGES_LIST, B=len(NL), slices=1, precision=_lib.PREC_F16X3, max_scatter=1, n_list=lst["n_list"], sh0_zero=1, msg_out=msg,
                edge_scat=lst["edge_scat"], nbr=lst["nbr"], feat_idx=lst["feat_idx"], attr_swap=0 if mode == "p2l" else 1,
                gather_tab=base["gather_tab"], gather_stride=dims["d_in"], attr1_tab=base["attr1_tab"], attr1_stride=dims["d_in"],
                attr2_tab=base["attr2_tab"], attr2_stride=dims["d_in"], edge_emb=base["edge_emb"], edge_sh=base["edge_sh"], sh_stride=9,
                w1p=base["w1p"], w2p=lst["w2x"])
    _run(larg, keep)
    assert torch.isfinite(msg).all()
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    s0 = dict(base, out=out, s0_only=1)
    if fold:
        s0.update(list_msg=msg, list_rowptr=lst["rowptr"], list_scale=lst["bnx"])
    _run(s0, keep)
    if not fold:
        deg_ptr = base["ph_ptr"] if mode == "p2l" else base["lig_ptr"]
        _lib.check(_lib.lib().dp_list_reduce(msg.data_ptr(), lst["rowptr"].data_ptr(), len(NL), base["scat_ptr"].data_ptr(), max(NL if mode == "p2l" else NP), deg_ptr.data_ptr(), dims["d_out"],
                                             lst["bnx"].data_ptr(), out.data_ptr(), dims["d_out"], st), "dp_list_reduce")
        torch.cuda.synchronize()
    got = out.cpu().double()
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    scale = ref.abs().max().item()
    err = (got - want).abs().max().item() / scale
    # the same inputs through the dense kernel
    dense = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    _run(dict(base, out=dense), keep)
    err_dense = (dense.cpu().double() - want).abs().max().item() / scale
    print(f"kind {kind} {mode} fold={fold}: s0-only + {lst['n_list']} listed pairs vs fp64 oracle {err:.2e} (dense kernel {err_dense:.2e})")
    assert err < TOL and err_dense < TOL, (err, err_dense)
    # without the listed pairs the result is NOT the conv (the list carries real terms)
    out0 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    _run(dict(base, out=out0, s0_only=1), keep)
    assert (out0.cpu().double() - want).abs().max().item() / scale > 100 * TOL
    # bit-identical reruns
    out2 = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    if fold:
        _run(dict(s0, out=out2), keep)
        assert torch.equal(out2.cpu().double(), got)


def test_list_launch_rejects_what_it_does_not_cover():
    base, ref, resid, dims, n_out, lst = _sparse_sh_case(1, "p2l", 5, [9], [12], 0.2)
    a = _lib.DpConvArgs()
    for k_, v in dict(kind=1, mode=_lib.EDGES_LIST, B=1, slices=1, precision=_lib.PREC_F16X3, n_list=lst["n_list"], sh0_zero=0).items():
        setattr(a, k_, v)
    import ctypes as C
    assert _lib.lib().dp_tp_conv(C.byref(a), None) == -1          # missing pointers
    msg = torch.zeros(lst["n_list"], dims["d_out"]).cuda()
    for k_, v in dict(msg_out=msg, edge_scat=lst["edge_scat"], nbr=lst["nbr"], feat_idx=lst["feat_idx"], gather_tab=base["gather_tab"], attr1_tab=base["attr1_tab"],
                      attr2_tab=base["attr2_tab"], edge_emb=base["edge_emb"], edge_sh=base["edge_sh"], w1p=base["w1p"], w2p=lst["w2x"]).items():
        setattr(a, k_, v.data_ptr())
    a.kind = 5
    assert _lib.lib().dp_tp_conv(C.byref(a), None) == -2          # layer convs only
    a.kind, a.precision = 1, _lib.PREC_F32
    assert _lib.lib().dp_tp_conv(C.byref(a), None) == -2          # the two-limb fp16 form only


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_whole_conv_as_one_list_launch(kind):
    """DP_EDGES_LIST with sh0_zero = 0 + dp_list_finish: a WHOLE conv (every edge listed, grouped by scatter node, rows of the edge tables = list
    positions) - what dp_forward does with the phore-phore convs of small pharmacophores, whose poses cannot fill 128-edge chunks of their own.
    Against the fp64 oracle at the edge-side kernels' tolerance."""
    NL, NP = [7, 30, 13, 1, 40, 17], [32, 5, 79, 33, 17, 48]
    base, ref, resid, dims, n_out, lst = _sparse_sh_case(kind, "p2l", 3100 + kind, NL, NP, 1.1)       # density > 1: every pair is listed
    E = base["edge_sh"].shape[0]
    assert lst["n_list"] == E
    keep = []
    msg = torch.full((E, dims["d_out"]), float("nan")).cuda()
    larg = dict(kind=kind, mode=_lib.EDGES_LIST, B=len(NL), slices=1, precision=_lib.PREC_F16X3, max_scatter=1, n_list=E, sh0_zero=0, msg_out=msg,
                edge_scat=lst["edge_scat"], nbr=lst["nbr"], feat_idx=None, attr_swap=0, gather_tab=base["gather_tab"], gather_stride=dims["d_in"],
                attr1_tab=base["attr1_tab"], attr1_stride=dims["d_in"], attr2_tab=base["attr2_tab"], attr2_stride=dims["d_in"], edge_emb=base["edge_emb"],
                edge_sh=base["edge_sh"], sh_stride=9, w1p=base["w1p"], w2p=lst["w2x"])
    _run(larg, keep)
    # DpConvArgs.phase_split on a list launch: two workgroups per chunk, one accumulation phase each - the same messages, bit for bit
    msg2 = torch.full((E, dims["d_out"]), float("nan")).cuda()
    _run(dict(larg, msg_out=msg2, phase_split=1), keep)
    torch.cuda.synchronize()
    assert torch.equal(msg, msg2)
    out = torch.full((n_out, dims["d_out"]), float("nan")).cuda()
    rp = lst["rowptr"]
    _lib.check(_lib.lib().dp_list_finish(msg.data_ptr(), rp.data_ptr(), rp[1:].contiguous().data_ptr(), n_out, dims["d_out"], lst["bnx"].data_ptr(),
                                         base["bn_shift"].data_ptr(), base["resid"].data_ptr(), dims["d_in"], dims["d_in"], out.data_ptr(), dims["d_out"], 0, 0,
                                         torch.cuda.current_stream().cuda_stream), "dp_list_finish")
    torch.cuda.synchronize()
    want = ref.clone()
    want[:, :dims["d_in"]] += resid.double()
    err = (out.cpu().double() - want).abs().max().item() / ref.abs().max().item()
    print(f"kind {kind}: whole conv as a list launch vs fp64 oracle {err:.2e}")
    assert err < TOL, err
