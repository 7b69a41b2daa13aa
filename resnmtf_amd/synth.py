"""Synthetic multi-view inputs for tests and benchmarks (host side, NumPy only).

Recipe = the reference's planted-bicluster test data
(``tests/testthat/test-resnmtf.R:38-52``: k disjoint blocks of height 10 plus
0.1*|N(0,1)| noise) scaled to arbitrary shapes, followed by the reference's
pre-processing (non-negativity shift is a no-op for X >= 0, ``R/utils.r:20-27``; column
L1 normalisation, ``R/utils.r:86-88``).  Seeds and shapes are those of SURVEY.md
section 8(d2) / BASELINE.md: NumPy PCG64, seed 1000+v for data, 2000+v for the initial
factors.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np


def planted_blocks(n: int, k: int) -> np.ndarray:
    """0/1 membership matrix (n, k): block p covers rows floor(p*n/k) .. floor((p+1)*n/k)-1."""
    r = np.zeros((n, k))
    for p in range(k):
        r[(p * n) // k:((p + 1) * n) // k, p] = 1.0
    return r


def planted_view(n: int, m: int, k: int, seed: int, height: float = 10.0,
                 noise: float = 0.1, normalise: bool = True) -> np.ndarray:
    """One view: height * R C^T + noise * |N(0,1)|, then X / colSums(X)."""
    rng = np.random.default_rng(seed)
    x = height * (planted_blocks(n, k) @ planted_blocks(m, k).T)
    x += noise * np.abs(rng.standard_normal((n, m)))
    if normalise:
        x /= x.sum(axis=0)[None, :]
    return x


def random_init(n: int, m: int, k: int, seed: int, sigma: float = 0.05):
    """Strictly positive, column-L1-normalised F0 (n,k), G0 (m,k) from U(0.1, 1) and a
    diagonally dominant S0 = I + |N(0, sigma)| (shape of ``R/update_steps.r:95-105`` with
    unit singular values; the column rescale by colSums(F0)*colSums(G0) is the identity
    because both are normalised)."""
    rng = np.random.default_rng(seed)
    f = rng.uniform(0.1, 1.0, size=(n, k))
    g = rng.uniform(0.1, 1.0, size=(m, k))
    f /= f.sum(axis=0)[None, :]
    g /= g.sum(axis=0)[None, :]
    s = np.eye(k) + np.abs(rng.normal(0.0, np.sqrt(sigma), size=(k, k)))
    return f, s, g


@dataclass
class Problem:
    """A complete input set for ``res_nmtf_inner`` (explicit-init entry, R/main.r:32-37)."""
    data: List[np.ndarray]
    init_f: List[np.ndarray]
    init_s: List[np.ndarray]
    init_g: List[np.ndarray]
    phi: np.ndarray
    xi: np.ndarray
    psi: np.ndarray
    k: int
    name: str = ""
    row_names: Optional[List[List[str]]] = None
    col_names: Optional[List[List[str]]] = None
    extras: dict = field(default_factory=dict)


def make_problem(shapes: Sequence[Sequence[int]], k: int, phi: float = 0.0, xi: float = 0.0,
                 psi: float = 0.0, name: str = "", seed_base: int = 0) -> Problem:
    """V views with the given (n, m) shapes.  A non-zero ``phi``/``psi``/``xi`` couples every
    pair of views with that weight (value of the symmetrised matrix, i.e. what
    ``init_rest_mats`` returns); views coupled through phi (psi) must have equal n (m) and
    share all rows (columns) in the same order -- the reference's auto-naming semantics
    (``R/utils.r:482-491``)."""
    n_v = len(shapes)
    data, f0, s0, g0 = [], [], [], []
    for v, (n, m) in enumerate(shapes):
        data.append(planted_view(n, m, k, 1000 + v + seed_base))
        f, s, g = random_init(n, m, k, 2000 + v + seed_base)
        f0.append(f); s0.append(s); g0.append(g)
    off = 1.0 - np.eye(n_v)
    prob = Problem(data, f0, s0, g0, phi * off, xi * off, psi * off, k, name)
    rn, cn = [], []
    rbase = cbase = 1
    for v, (n, m) in enumerate(shapes):
        if phi != 0.0:
            if n != shapes[0][0]:
                raise ValueError("phi-coupled synthetic views must share n")
            rn.append([f"row_{t}" for t in range(1, n + 1)])
        else:
            rn.append([f"row_{t}" for t in range(rbase, rbase + n)]); rbase += n
        if psi != 0.0:
            if m != shapes[0][1]:
                raise ValueError("psi-coupled synthetic views must share m")
            cn.append([f"col_{t}" for t in range(1, m + 1)])
        else:
            cn.append([f"col_{t}" for t in range(cbase, cbase + m)]); cbase += m
    prob.row_names, prob.col_names = rn, cn
    return prob


# BASELINE.json configs
def config(name: str) -> Problem:
    if name == "c1":
        return make_problem([(100, 50)], 3, name="c1: 1 view 100x50 k=3")
    if name == "c2":
        return make_problem([(10000, 2000)], 16, name="c2: 1 view 10000x2000 k=16")
    if name == "c3":
        return make_problem([(10000, 2000), (10000, 1500)], 16, phi=200.0,
                            name="c3: 2 views 10000x{2000,1500} k=16 phi=200")
    if name == "c4":
        return make_problem([(20000, 4000)] * 4, 32, phi=200.0, psi=200.0,
                            name="c4: 4 views 20000x4000 k=32 phi=psi=200")
    if name == "c5":
        return make_problem([(50000, 8000)] * 8, 64, phi=200.0, xi=200.0, psi=200.0,
                            name="c5: 8 views 50000x8000 k=64 phi=xi=psi=200")
    raise KeyError(name)
