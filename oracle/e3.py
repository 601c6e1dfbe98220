"""Oracle restatement of the e3nn==0.5.1 operators the score model calls (TEST INFRASTRUCTURE ONLY).

e3nn is NOT under /root/reference (pinned in `src/environment_diffphore.yml:117`); its algorithm is restated from the
published definitions.  Call sites in the reference that these functions stand in for:

  o3.spherical_harmonics(irreps, v, normalize=True, normalization='component')
        src/models/score_model_phore.py:365,404,434,737,754,891,893
  o3.FullyConnectedTensorProduct(in, sh, out, shared_weights=False)       :123,137
  o3.FullTensorProduct(sh_irreps, "2e")                                     :276,366
  e3nn.nn.BatchNorm(out_irreps)  (eval mode)                                :132,148

Status: parity unpinned (no e3nn here, no reference tests/vectors).  Locked by property tests instead.
"""
from __future__ import annotations

import math
from fractions import Fraction
from functools import lru_cache
from typing import List, Tuple

import numpy as np
import torch

Irrep = Tuple[int, int]            # (l, parity) parity in {+1 (e), -1 (o)}
Irreps = List[Tuple[int, Irrep]]   # [(mul, (l, p)), ...]


def parse_irreps(s: str) -> Irreps:
    """'20x0e + 10x1o' -> [(20,(0,1)), (10,(1,-1))]"""
    out = []
    for tok in s.replace(" ", "").split("+"):
        if "x" in tok:
            mul, ir = tok.split("x")
        else:
            mul, ir = "1", tok
        l, p = int(ir[:-1]), {"e": 1, "o": -1}[ir[-1]]
        out.append((int(mul), (l, p)))
    return out


def irreps_dim(irreps: Irreps) -> int:
    return sum(mul * (2 * l + 1) for mul, (l, _) in irreps)


def irreps_slices(irreps: Irreps):
    off, out = 0, []
    for mul, (l, _) in irreps:
        d = mul * (2 * l + 1)
        out.append(slice(off, off + d))
        off += d
    return out


SH_IRREPS = parse_irreps("1x0e + 1x1o + 1x2e")


# --------------------------------------------------------------------------------------------------------------
# spherical harmonics, l <= 2, e3nn conventions (polar axis y; l=1 is the vector itself; 'component' normalisation
# => |Y_l|^2 = 2l+1 on the unit sphere)
# --------------------------------------------------------------------------------------------------------------
def spherical_harmonics_lmax2(vec: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """[...,3] -> [...,9] = [Y0 | Y1(3) | Y2(5)], 'component' normalisation.

    normalize=True follows torch.nn.functional.normalize: v / max(|v|, 1e-12)  (zero vector -> [1,0,...,0]).
    """
    if normalize:
        vec = vec / vec.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    x, y, z = vec[..., 0], vec[..., 1], vec[..., 2]
    s3, s5, s15 = math.sqrt(3.0), math.sqrt(5.0), math.sqrt(15.0)
    return torch.stack([
        torch.ones_like(x),
        s3 * x, s3 * y, s3 * z,
        s15 * x * z,
        s15 * x * y,
        s5 * (y * y - 0.5 * (x * x + z * z)),
        s15 * y * z,
        0.5 * s15 * (z * z - x * x),
    ], dim=-1)


def spherical_harmonics_2e(vec: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    return spherical_harmonics_lmax2(vec, normalize)[..., 4:9]


# --------------------------------------------------------------------------------------------------------------
# Wigner 3j in e3nn's real basis
# --------------------------------------------------------------------------------------------------------------
def _f(n: int) -> int:
    return math.factorial(n)


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3) -> float:
    """<j1 m1 j2 m2 | j3 m3>, integer spins, Racah's formula (exact rational under the root)."""
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))
    C = Fraction((2 * j3 + 1) * _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3) * _f(j3 + m3) * _f(j3 - m3),
                 _f(j1 + j2 + j3 + 1) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2))
    S = Fraction(0)
    for v in range(vmin, vmax + 1):
        S += Fraction((-1) ** (v + j2 + m2) * _f(j2 + j3 + m1 - v) * _f(j1 - m1 + v),
                      _f(v) * _f(j3 - j1 + j2 - v) * _f(j3 + m3 - v) * _f(v + j1 - j2 - m3))
    return math.sqrt(float(C)) * float(S)


def _real_to_complex(l: int) -> np.ndarray:
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / math.sqrt(2)
        q[l + m, l - abs(m)] = -1j / math.sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / math.sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / math.sqrt(2)
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1: int, l2: int, l3: int) -> np.ndarray:
    """Real-basis Wigner 3j tensor [2l1+1, 2l2+1, 2l3+1], Frobenius norm 1 (e3nn `o3.wigner_3j`)."""
    assert abs(l1 - l2) <= l3 <= l1 + l2
    C = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                C[l1 + m1, l2 + m2, l3 + m1 + m2] = _su2_cg_coeff(l1, m1, l2, m2, l3, m1 + m2)
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    Cr = np.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, np.conj(Q3.T), C.astype(np.complex128))
    assert np.abs(Cr.imag).max() < 1e-9
    Cr = Cr.real
    return Cr / np.linalg.norm(Cr)


# --------------------------------------------------------------------------------------------------------------
# FullyConnectedTensorProduct ('uvw', per-edge weights) and FullTensorProduct ('uvuv')
# --------------------------------------------------------------------------------------------------------------
class Path:
    __slots__ = ("i_in", "i_sh", "i_out", "l_in", "l_sh", "l_out", "mul_in", "mul_sh", "mul_out",
                 "w_off", "w_numel", "coef")

    def __repr__(self):
        return (f"Path(in{self.i_in}(l{self.l_in}x{self.mul_in}) * sh{self.i_sh}(l{self.l_sh}) -> out{self.i_out}"
                f"(l{self.l_out}x{self.mul_out}) w[{self.w_off}:{self.w_off + self.w_numel}] c={self.coef:.5f})")


def fctp_paths(irreps_in: Irreps, irreps_sh: Irreps, irreps_out: Irreps) -> Tuple[List[Path], int]:
    """Instruction list of o3.FullyConnectedTensorProduct: loop in1, then in2, then out; keep ir_out in ir1 x ir2.

    irrep_normalization='component', path_normalization='element':
        coef = sqrt( (2 l_out + 1) / sum_{paths into the same out slot} mul_in * mul_sh ).
    Weight vector = concatenation of [mul_in, mul_sh, mul_out] blocks in instruction order.
    """
    paths: List[Path] = []
    off = 0
    for i1, (m1, (l1, p1)) in enumerate(irreps_in):
        for i2, (m2, (l2, p2)) in enumerate(irreps_sh):
            for io, (mo, (lo, po)) in enumerate(irreps_out):
                if abs(l1 - l2) <= lo <= l1 + l2 and po == p1 * p2:
                    p = Path()
                    p.i_in, p.i_sh, p.i_out = i1, i2, io
                    p.l_in, p.l_sh, p.l_out = l1, l2, lo
                    p.mul_in, p.mul_sh, p.mul_out = m1, m2, mo
                    p.w_off, p.w_numel = off, m1 * m2 * mo
                    off += p.w_numel
                    paths.append(p)
    for p in paths:
        fan = sum(q.mul_in * q.mul_sh for q in paths if q.i_out == p.i_out)
        p.coef = math.sqrt((2 * p.l_out + 1) / fan)
    return paths, off


def fctp_apply(x: torch.Tensor, sh: torch.Tensor, w: torch.Tensor, irreps_in: Irreps, irreps_sh: Irreps,
               irreps_out: Irreps) -> torch.Tensor:
    """x [E,Din], sh [E,Dsh], w [E,numel] -> [E,Dout]   (e3nn 'uvw' einsum zuvw,ijk,zuvij->zwk times path weight)."""
    paths, numel = fctp_paths(irreps_in, irreps_sh, irreps_out)
    assert w.shape[-1] == numel, (w.shape, numel)
    s_in, s_sh, s_out = irreps_slices(irreps_in), irreps_slices(irreps_sh), irreps_slices(irreps_out)
    E = x.shape[0]
    out = x.new_zeros(E, irreps_dim(irreps_out))
    for p in paths:
        x1 = x[:, s_in[p.i_in]].reshape(E, p.mul_in, 2 * p.l_in + 1)
        x2 = sh[:, s_sh[p.i_sh]].reshape(E, p.mul_sh, 2 * p.l_sh + 1)
        ww = w[:, p.w_off:p.w_off + p.w_numel].reshape(E, p.mul_in, p.mul_sh, p.mul_out)
        C = torch.as_tensor(wigner_3j(p.l_in, p.l_sh, p.l_out), dtype=x.dtype)
        # == einsum("zuvw,ijk,zui,zvj->zwk", ww, C, x1, x2), contracted in two cheap stages
        zz = torch.einsum("ijk,zui,zvj->zuvk", C, x1, x2).reshape(E, p.mul_in * p.mul_sh, 2 * p.l_out + 1)
        r = torch.bmm(ww.reshape(E, p.mul_in * p.mul_sh, p.mul_out).transpose(1, 2), zz)
        out[:, s_out[p.i_out]] += p.coef * r.reshape(E, -1)
    return out


def full_tp_irreps(irreps_1: Irreps, irreps_2: Irreps):
    """Output irreps + instructions of o3.FullTensorProduct (mode 'uvuv', outputs sorted by (l, p) with odd first)."""
    outs = []
    for i1, (m1, (l1, p1)) in enumerate(irreps_1):
        for i2, (m2, (l2, p2)) in enumerate(irreps_2):
            for lo in range(abs(l1 - l2), l1 + l2 + 1):
                outs.append(((lo, p1 * p2), len(outs), m1 * m2, i1, i2))
    outs_sorted = sorted(outs, key=lambda t: (t[0][0], t[0][1], t[1]))  # Irrep tuple order: (l, p) with p=-1 < p=+1
    irreps_out = [(t[2], t[0]) for t in outs_sorted]
    instr = [(t[3], t[4], k) for k, t in enumerate(outs_sorted)]
    return irreps_out, instr


def full_tp_apply(x1: torch.Tensor, x2: torch.Tensor, irreps_1: Irreps, irreps_2: Irreps) -> torch.Tensor:
    """o3.FullTensorProduct forward: out[(u,v),k] = sqrt(2 l_out + 1) * sum_ij C[i,j,k] x1[u,i] x2[v,j]."""
    irreps_out, instr = full_tp_irreps(irreps_1, irreps_2)
    s1, s2, so = irreps_slices(irreps_1), irreps_slices(irreps_2), irreps_slices(irreps_out)
    E = x1.shape[0]
    out = x1.new_zeros(E, irreps_dim(irreps_out))
    for (i1, i2, io) in instr:
        m1, (l1, _) = irreps_1[i1]
        m2, (l2, _) = irreps_2[i2]
        _, (lo, _) = irreps_out[io]
        a = x1[:, s1[i1]].reshape(E, m1, 2 * l1 + 1)
        b = x2[:, s2[i2]].reshape(E, m2, 2 * l2 + 1)
        C = torch.as_tensor(wigner_3j(l1, l2, lo), dtype=x1.dtype)
        r = torch.einsum("ijk,zui,zvj->zuvk", C, a, b)
        out[:, so[io]] = math.sqrt(2 * lo + 1) * r.reshape(E, -1)
    return out


# --------------------------------------------------------------------------------------------------------------
# e3nn.nn.BatchNorm, eval mode, affine, eps=1e-5, normalization='component'
# --------------------------------------------------------------------------------------------------------------
def batchnorm_eval(x: torch.Tensor, irreps: Irreps, running_mean: torch.Tensor, running_var: torch.Tensor,
                   weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    out = []
    ix = irm = irv = 0
    ib = 0
    N = x.shape[0]
    for mul, (l, p) in irreps:
        d = 2 * l + 1
        f = x[:, ix:ix + mul * d].reshape(N, mul, d)
        ix += mul * d
        scalar = (l == 0 and p == 1)
        if scalar:
            f = f - running_mean[irm:irm + mul].reshape(1, mul, 1)
            irm += mul
        fn = (running_var[irv:irv + mul] + eps).pow(-0.5) * weight[irv:irv + mul]
        irv += mul
        f = f * fn.reshape(1, mul, 1)
        if scalar:
            f = f + bias[ib:ib + mul].reshape(1, mul, 1)
            ib += mul
        out.append(f.reshape(N, mul * d))
    return torch.cat(out, dim=-1)


def bn_num_features(irreps: Irreps):
    """(num scalar channels, num multiplicity channels) = sizes of (running_mean/bias, running_var/weight)."""
    ns = sum(mul for mul, (l, p) in irreps if l == 0 and p == 1)
    nf = sum(mul for mul, _ in irreps)
    return ns, nf
