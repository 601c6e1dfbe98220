"""Oracle restatement of the SO(3) / torus score-norm tables and their lookups (TEST INFRASTRUCTURE ONLY).

  so3:   /root/reference/src/utils/so3.py:6-67 (table build), :92-96 (`score_norm` lookup)
  torus: /root/reference/src/utils/torus.py:11-43,52-86 (table build incl. the Monte-Carlo `score_norm_`), :82-86 (lookup)

PINNED: `tests/test_oracle_golden.py` checks the rebuilt SO(3) rows bit-close against the table produced by importing
the reference itself (`diffphore_amd/data/so3_exp_score_norms.npy`, made by tests/golden/make_golden.py), the torus
rows within Monte-Carlo tolerance, and both lookups against `tests/golden/ref_tables_probe.npz`.
"""
from __future__ import annotations

import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffphore_amd", "data")

SO3_MIN_EPS, SO3_MAX_EPS, SO3_N_EPS, SO3_X_N = 0.01, 2, 1000, 2000
TORUS_X_MIN, TORUS_X_N = 1e-5, 5000
TORUS_SIGMA_MIN, TORUS_SIGMA_MAX, TORUS_SIGMA_N = 3e-3, 2, 5000


def load_tables():
    return (np.load(os.path.join(_DATA, "so3_exp_score_norms.npy")),
            np.load(os.path.join(_DATA, "torus_score_norm.npy")))


_cache = None


def _tables():
    global _cache
    if _cache is None:
        _cache = load_tables()
    return _cache


def so3_index(eps) -> np.ndarray:
    eps = np.asarray(eps)
    idx = (np.log10(eps) - np.log10(SO3_MIN_EPS)) / (np.log10(SO3_MAX_EPS) - np.log10(SO3_MIN_EPS)) * SO3_N_EPS
    return np.clip(np.around(idx).astype(int), a_min=0, a_max=SO3_N_EPS - 1)


def so3_score_norm(eps, table=None) -> np.ndarray:
    """so3.score_norm (so3.py:92-96): nearest-index lookup, returned as float32 like `.float()` there."""
    table = _tables()[0] if table is None else table
    return table[so3_index(eps)].astype(np.float32)


def torus_index(sigma) -> np.ndarray:
    s = np.log(np.asarray(sigma) / np.pi)
    s = (s - np.log(TORUS_SIGMA_MIN)) / (np.log(TORUS_SIGMA_MAX) - np.log(TORUS_SIGMA_MIN)) * TORUS_SIGMA_N
    return np.round(np.clip(s, 0, TORUS_SIGMA_N)).astype(int)


def torus_score_norm(sigma, table=None) -> np.ndarray:
    """torus.score_norm (torus.py:82-86)."""
    table = _tables()[1] if table is None else table
    return table[torus_index(sigma)]


# --------------------------------------------------------------------------------------------------------------
# table builders (restated; row subsets so tests stay fast)
# --------------------------------------------------------------------------------------------------------------
def so3_exp_score_norm_rows(eps_idx, L: int = 2000) -> np.ndarray:
    """Rows of `_exp_score_norms` (so3.py:49-60) for the given eps indices.

    Same summation order over l as the reference's python loops (accumulate l = 0..L-1), vectorised over omega.
    """
    eps_array = 10 ** np.linspace(np.log10(SO3_MIN_EPS), np.log10(SO3_MAX_EPS), SO3_N_EPS)
    omega = np.linspace(0, np.pi, SO3_X_N + 1)[1:]
    out = []
    for i in np.atleast_1d(eps_idx):
        eps = eps_array[i]
        p = 0
        for l in range(L):                                              # _expansion (so3.py:21-25)
            p = p + (2 * l + 1) * np.exp(-l * (l + 1) * eps ** 2) * np.sin(omega * (l + 1 / 2)) / np.sin(omega / 2)
        pdf = p * (1 - np.cos(omega)) / np.pi                           # _density marginal (so3.py:28-32)
        dS = 0
        for l in range(L):                                              # _score (so3.py:35-43)
            hi = np.sin(omega * (l + 1 / 2))
            dhi = (l + 1 / 2) * np.cos(omega * (l + 1 / 2))
            lo = np.sin(omega / 2)
            dlo = 1 / 2 * np.cos(omega / 2)
            dS = dS + (2 * l + 1) * np.exp(-l * (l + 1) * eps ** 2) * (lo * dhi - hi * dlo) / lo ** 2
        score = dS / p
        out.append(np.sqrt(np.sum(score ** 2 * pdf) / np.sum(pdf) / np.pi))
    return np.asarray(out)


def torus_score_norm_rows(sigma_idx, n_samples: int = 10000, seed: int = 0, N: int = 100) -> np.ndarray:
    """Rows of the Monte-Carlo `score_norm_` (torus.py:75-79) for the given sigma indices (independent draw)."""
    rng = np.random.default_rng(seed)
    x = 10 ** np.linspace(np.log10(TORUS_X_MIN), 0, TORUS_X_N + 1) * np.pi
    sig_all = 10 ** np.linspace(np.log10(TORUS_SIGMA_MIN), np.log10(TORUS_SIGMA_MAX), TORUS_SIGMA_N + 1) * np.pi
    out = []
    for si in np.atleast_1d(sigma_idx):
        s = sig_all[si]
        p_ = np.zeros_like(x)
        g_ = np.zeros_like(x)
        for i in range(-N, N + 1):                                       # p / grad (torus.py:11-22)
            e = np.exp(-(x + 2 * np.pi * i) ** 2 / 2 / s ** 2)
            p_ += e
            g_ += (x + 2 * np.pi * i) / s ** 2 * e
        score_row = g_ / p_
        smp = s * rng.standard_normal(n_samples)                          # sample (torus.py:69-72)
        smp = (smp + np.pi) % (2 * np.pi) - np.pi
        sign = np.sign(smp)                                               # score (torus.py:46-55)
        xi = np.log(np.abs(smp) / np.pi)
        xi = (xi - np.log(TORUS_X_MIN)) / (0 - np.log(TORUS_X_MIN)) * TORUS_X_N
        xi = np.round(np.clip(xi, 0, TORUS_X_N)).astype(int)
        sc = -sign * score_row[xi]
        out.append((sc ** 2).mean())
    return np.asarray(out)
