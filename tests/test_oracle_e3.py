"""Property tests that lock the oracle's restatement of the e3nn operators (parity with e3nn itself is unpinned: e3nn is
absent; see oracle/__init__.py).  CPU only."""
import math

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from diffphore_amd import config as C
from oracle import e3

torch.manual_seed(0)
SL = {0: slice(0, 1), 1: slice(1, 4), 2: slice(4, 9)}


def wigner_D(R: torch.Tensor, l: int) -> torch.Tensor:
    """D_l(R) defined by Y_l(R v) = D_l(R) Y_l(v) (least squares over random v; exact because Y_l spans the irrep)."""
    v = torch.randn(64, 3, dtype=torch.float64)
    Y = e3.spherical_harmonics_lmax2(v)[:, SL[l]]
    YR = e3.spherical_harmonics_lmax2(v @ R.T)[:, SL[l]]
    D = torch.linalg.lstsq(Y, YR).solution.T
    assert (Y @ D.T - YR).abs().max() < 1e-10
    return D


def rot(seed, improper=False):
    R = torch.tensor(Rotation.random(random_state=seed).as_matrix())
    return -R if improper else R


def test_sh_normalisation_and_values():
    v = torch.randn(100, 3, dtype=torch.float64)
    Y = e3.spherical_harmonics_lmax2(v)
    for l in (0, 1, 2):                                        # 'component': |Y_l|^2 = 2l+1
        assert torch.allclose((Y[:, SL[l]] ** 2).sum(-1), torch.full((100,), 2.0 * l + 1, dtype=torch.float64))
    n = v / v.norm(dim=-1, keepdim=True)
    assert torch.allclose(Y[:, 1:4], math.sqrt(3) * n)         # l=1 is the (scaled) vector itself
    z = e3.spherical_harmonics_lmax2(torch.zeros(1, 3))
    assert torch.equal(z, torch.tensor([[1.0, 0, 0, 0, 0, 0, 0, 0, 0]]))      # zero vector


@pytest.mark.parametrize("ls", [(0, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1), (1, 2, 1), (2, 2, 0), (2, 2, 1), (1, 1, 2), (2, 2, 2)])
def test_wigner3j_invariant_normalised(ls):
    l1, l2, l3 = ls
    Cg = e3.wigner_3j(l1, l2, l3)
    assert abs(np.linalg.norm(Cg) - 1) < 1e-12
    R = rot(3)
    D1, D2, D3 = (wigner_D(R, l).numpy() for l in ls)
    assert np.abs(np.einsum("ia,jb,kc,abc->ijk", D1, D2, D3, Cg) - Cg).max() < 1e-10


def test_wigner3j_sign_conventions():
    assert abs(e3.wigner_3j(1, 1, 1)[0, 1, 2] - 1 / math.sqrt(6)) < 1e-12
    assert np.allclose(e3.wigner_3j(1, 1, 0)[:, :, 0], np.eye(3) / math.sqrt(3))
    assert np.allclose(e3.wigner_3j(2, 2, 0)[:, :, 0], np.eye(5) / math.sqrt(5))
    v = torch.randn(7, 3, dtype=torch.float64)
    Y = e3.spherical_harmonics_lmax2(v)
    for (l1, l2, l3), const in {(1, 1, 0): math.sqrt(3), (1, 1, 2): 0.4898979485566356, (1, 2, 1): 0.816496580927726,
                                (2, 2, 0): math.sqrt(5), (2, 2, 2): 0.5345224838248488}.items():
        r = torch.einsum("ijk,zi,zj->zk", torch.tensor(e3.wigner_3j(l1, l2, l3)), Y[:, SL[l1]], Y[:, SL[l2]])
        assert torch.allclose(r, const * Y[:, SL[l3]], atol=1e-10)          # positive constants


def test_path_tables_match_product_config():
    """The product's compile-time tables (diffphore_amd/config.py, dp_spec.h) and the oracle's e3nn-rule enumeration agree."""
    seqs = ["20x0e", "20x0e + 10x1o", "20x0e + 10x1o + 10x1e", "20x0e + 10x1o + 10x1e + 20x0o"]
    for l in range(4):
        po, numel = e3.fctp_paths(e3.parse_irreps(seqs[min(l, 3)]), e3.SH_IRREPS, e3.parse_irreps(seqs[min(l + 1, 3)]))
        pc, numel_c = C.tp_paths(*[C.layer_irreps(l)[0], C.SH, C.layer_irreps(l)[1]])
        assert numel == numel_c == [600, 1100, 1600, 2200][l]
        assert [(p.l_in, p.l_sh, p.l_out, p.w_off) for p in po] == [(p.l_in, p.l_sh, p.l_out, p.w_off) for p in pc]
        assert np.allclose([p.coef for p in po], [p.coef for p in pc])
    po, n = e3.fctp_paths(e3.parse_irreps(seqs[3]), e3.SH_IRREPS, e3.parse_irreps("2x1o + 2x1e"))
    assert n == 200 and len(po) == 8
    tor_sh, _ = e3.full_tp_irreps(e3.SH_IRREPS, e3.parse_irreps("1x2e"))
    assert e3.irreps_dim(tor_sh) == 45
    assert [(l, p) for _, (l, p) in tor_sh] == [(0, 1), (1, -1), (1, 1), (2, -1), (2, 1), (2, 1), (3, -1), (3, 1), (4, 1)]
    po, n = e3.fctp_paths(e3.parse_irreps(seqs[3]), tor_sh, e3.parse_irreps("20x0o + 20x0e"))
    assert n == 1600 and [p.w_off for p in po] == [0, 400, 600, 800, 1000, 1200]


def _rotate_features(x, irreps, R, improper):
    out, off = [], 0
    for mul, (l, p) in irreps:
        d = 2 * l + 1
        blk = x[:, off:off + mul * d].reshape(-1, mul, d)
        D = wigner_D(R if not improper else -R, l) if l > 0 else torch.eye(1, dtype=torch.float64)
        # parity: an improper rotation multiplies an irrep of parity p by p (on top of the l-dependent (-1)^l inside Y_l(-v))
        sign = 1.0
        if improper:
            sign = p * ((-1) ** l)
        out.append((sign * blk @ D.T).reshape(-1, mul * d))
        off += mul * d
    return torch.cat(out, -1)


@pytest.mark.parametrize("improper", [False, True])
def test_fctp_equivariance(improper):
    ir_in = e3.parse_irreps("20x0e + 10x1o + 10x1e + 20x0o")
    ir_out = ir_in
    E = 5
    x = torch.randn(E, e3.irreps_dim(ir_in), dtype=torch.float64)
    v = torch.randn(E, 3, dtype=torch.float64)
    w = torch.randn(E, 2200, dtype=torch.float64)
    R = rot(11)
    Rv = (-R if improper else R)
    y = e3.fctp_apply(x, e3.spherical_harmonics_lmax2(v), w, ir_in, e3.SH_IRREPS, ir_out)
    y2 = e3.fctp_apply(_rotate_features(x, ir_in, R, improper), e3.spherical_harmonics_lmax2(v @ Rv.T), w, ir_in, e3.SH_IRREPS, ir_out)
    assert torch.allclose(y2, _rotate_features(y, ir_out, R, improper), atol=1e-9)


def test_tp_output_variance_is_normalised():
    """component/element normalisation: unit-variance inputs and weights give ~unit-variance outputs."""
    ir = e3.parse_irreps("20x0e + 10x1o + 10x1e + 20x0o")
    E = 4000
    g = torch.Generator().manual_seed(1)
    x = torch.randn(E, 100, generator=g, dtype=torch.float64)
    sh = e3.spherical_harmonics_lmax2(torch.randn(E, 3, generator=g, dtype=torch.float64))
    w = torch.randn(E, 2200, generator=g, dtype=torch.float64)
    y = e3.fctp_apply(x, sh, w, ir, e3.SH_IRREPS, ir)
    assert 0.8 < float(y.var()) < 1.25


def test_batchnorm_eval_algebra():
    ir = e3.parse_irreps("3x0e + 2x1o + 2x0o")
    x = torch.randn(6, 3 + 6 + 2, dtype=torch.float64)
    rm, rv = torch.randn(3, dtype=torch.float64), torch.rand(7, dtype=torch.float64) + 0.5
    w, b = torch.rand(7, dtype=torch.float64) + 0.5, torch.randn(3, dtype=torch.float64)
    y = e3.batchnorm_eval(x, ir, rm, rv, w, b)
    s = w / torch.sqrt(rv + 1e-5)
    assert torch.allclose(y[:, :3], (x[:, :3] - rm) * s[:3] + b)               # 0e: mean, scale, bias
    assert torch.allclose(y[:, 3:9], x[:, 3:9] * s[3:5].repeat_interleave(3))   # 1o: scale only, shared over components
    assert torch.allclose(y[:, 9:], x[:, 9:] * s[5:])                           # 0o is NOT a scalar: no mean / bias


# ---------------------------------------------------------------------------------------------------------------------------
# Independent cross-checks of the e3nn conventions: e3nn itself cannot be imported anywhere in this project, so the
# restatement in oracle/e3.py is held to two libraries that know nothing about it - sympy's Clebsch-Gordan coefficients and scipy's
# complex spherical harmonics - through e3nn's published real<->complex change of basis.  This is the ceiling for row (c): what stays
# unpinned is only that e3nn 0.5.1 really uses the construction its source documents (o3/_wigner.py, o3/_spherical_harmonics.py).
# ---------------------------------------------------------------------------------------------------------------------------
TRIPLES = [(0, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1), (1, 2, 1), (2, 2, 0), (2, 2, 1), (1, 1, 2), (2, 2, 2), (2, 1, 1), (2, 2, 3), (2, 2, 4)]


@pytest.mark.parametrize("ls", TRIPLES)
def test_su2_clebsch_gordan_matches_sympy(ls):
    from sympy import S
    from sympy.physics.wigner import clebsch_gordan
    j1, j2, j3 = ls
    for m1 in range(-j1, j1 + 1):
        for m2 in range(-j2, j2 + 1):
            for m3 in range(-j3, j3 + 1):
                want = float(clebsch_gordan(S(j1), S(j2), S(j3), S(m1), S(m2), S(m3)))
                assert abs(e3._su2_cg_coeff(j1, m1, j2, m2, j3, m3) - want) < 1e-13, (ls, m1, m2, m3)


@pytest.mark.parametrize("ls", TRIPLES)
def test_real_wigner3j_rebuilt_from_sympy(ls):
    """e3nn's recipe (o3/_wigner.py `_so3_clebsch_gordan`): C_real = Re einsum('ij,kl,mn,ikn->jlm', Q1, Q2, conj(Q3^T), CG_su2) / |.|
    with Q(l) = (-i)^l q(l) - every ingredient here comes from sympy or is written out again, independently of oracle/e3.py."""
    from sympy import S
    from sympy.physics.wigner import clebsch_gordan
    l1, l2, l3 = ls

    def Q(l):
        q = np.zeros((2 * l + 1, 2 * l + 1), dtype=complex)
        for m in range(-l, l + 1):
            if m < 0:
                q[l + m, l - m], q[l + m, l + m] = 1 / math.sqrt(2), -1j / math.sqrt(2)
            elif m == 0:
                q[l, l] = 1
            else:
                q[l + m, l + m], q[l + m, l - m] = (-1) ** m / math.sqrt(2), 1j * (-1) ** m / math.sqrt(2)
        return (-1j) ** l * q
    cg = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1), dtype=complex)
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                cg[l1 + m1, l2 + m2, l3 + m1 + m2] = float(clebsch_gordan(S(l1), S(l2), S(l3), S(m1), S(m2), S(m1 + m2)))
    cr = np.einsum("ij,kl,mn,ikn->jlm", Q(l1), Q(l2), np.conj(Q(l3).T), cg)
    assert np.abs(cr.imag).max() < 1e-12
    cr = cr.real / np.linalg.norm(cr.real)
    assert np.abs(cr - e3.wigner_3j(l1, l2, l3)).max() < 1e-12


@pytest.mark.parametrize("l", [1, 2])
def test_real_spherical_harmonics_vs_scipy_complex_Ylm(l):
    """The oracle's real harmonics ARE scipy's complex Y_l^m (Condon-Shortley phase) seen through e3nn's basis change: with the polar
    axis along y - e3nn's convention: (x, y, z) = (sin(theta) sin(phi), cos(theta), sin(theta) cos(phi)) - and 'component'
    normalisation (sqrt(4 pi) Y), the complex vector with components m = -l..l equals Q(l) Y_real up to ONE constant phase per l
    (e3nn's (-i)^l), which the explicit l = 1 values pin.  This fixes the component order and every relative sign of the real basis
    independently of oracle/e3.py's formulas."""
    import scipy.special as sp
    rng = np.random.default_rng(0)
    v = rng.normal(size=(50, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    theta = np.arccos(np.clip(v[:, 1], -1, 1))                # polar angle from the y axis
    phi = np.arctan2(v[:, 0], v[:, 2])                        # azimuth in the (z, x) plane
    if hasattr(sp, "sph_harm_y"):
        Yc = np.stack([sp.sph_harm_y(l, m, theta, phi) for m in range(-l, l + 1)], 1)
    else:
        Yc = np.stack([sp.sph_harm(m, l, phi, theta) for m in range(-l, l + 1)], 1)
    Yc = Yc * math.sqrt(4 * math.pi)
    Yr = e3.spherical_harmonics_lmax2(torch.tensor(v))[:, SL[l]].numpy()
    Ycr = Yr @ e3._real_to_complex(l).T                        # Q Y_real
    ratio = (Ycr * np.conj(Yc)).sum() / (np.abs(Yc) ** 2).sum()
    assert abs(abs(ratio) - 1) < 1e-12                         # same functions up to a phase ...
    assert np.abs(Ycr - ratio * Yc).max() < 1e-12              # ... that is ONE constant for all m and all directions
    assert abs(ratio - (-1j) ** l) < 1e-12 or abs(ratio - (1j) ** l) < 1e-12    # and it is e3nn's (-i)^l (or its conjugate: the
    #                                                            real harmonics come out identical either way)
