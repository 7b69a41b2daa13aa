"""Shared helpers for the test-suite (test infrastructure; may import the oracle)."""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_NAMES = [
    "g1_single_60x40_k3",
    "g2_two_views_phi_partial",
    "g3_three_views_phi_psi_xi",
    "g4_dead_component_nan",
    "g5_psi_one_pair_k20",
]


def load_golden(name: str) -> dict:
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_v = int(z["n_views"])
    g = {"n_views": n_v, "n_iters": int(z["n_iters"]), "phi": z["phi"], "xi": z["xi"], "psi": z["psi"],
         "all_error": z["all_error"], "error": float(z["error"])}
    for key in ("x", "f0_", "s0_", "g0_", "out_f", "out_s", "out_g", "raw_f", "raw_s", "raw_g", "rc", "cc", "lam", "mu"):
        g[key.rstrip("_")] = [z[f"{key}{v}"] for v in range(n_v)]
    g["row_names"] = [[str(s) for s in z[f"rn{v}"]] for v in range(n_v)]
    g["col_names"] = [[str(s) for s in z[f"cn{v}"]] for v in range(n_v)]
    return g


def rel_fro(a: np.ndarray, b: np.ndarray) -> float:
    """relative Frobenius distance ||a - b|| / ||b||, NaN positions must coincide and are skipped."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), "NaN patterns differ"
    a = np.where(na, 0.0, a); b = np.where(nb, 0.0, b)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))


def run_oracle(prob, n_iters=None, max_iters=None):
    from oracle import resnmtf_oracle as O
    return O.res_nmtf_inner(prob.data, prob.init_f, prob.init_s, prob.init_g, prob.phi, prob.xi, prob.psi,
                            row_names=prob.row_names, col_names=prob.col_names, n_iters=n_iters,
                            max_iters=max_iters)


def run_hip(prob, n_iters=None, max_iters=100000, **engine_opts):
    """Through the package's public mirror of res_nmtf_inner -> ctypes -> C-ABI -> HIP."""
    from resnmtf_amd import naming, res_nmtf_inner
    k_vec = [f.shape[1] for f in prob.init_f]
    return res_nmtf_inner(prob.data, naming.shared_names(prob.row_names), naming.shared_names(prob.col_names),
                          prob.init_f, prob.init_s, prob.init_g, k_vec, prob.phi, prob.xi, prob.psi,
                          n_iters=n_iters, spurious=False, row_names=prob.row_names, col_names=prob.col_names,
                          max_iters=max_iters, engine_opts=engine_opts or None)


def golden_problem(g: dict):
    from resnmtf_amd.synth import Problem
    return Problem(g["x"], g["f0"], g["s0"], g["g0"], g["phi"], g["xi"], g["psi"], g["f0"][0].shape[1],
                   row_names=g["row_names"], col_names=g["col_names"])


def coupled_problem(shapes, k, seed, phi_w=0.0, psi_w=0.0, xi_w=0.0, overlap=0.75, na_pairs=(), same_order_views=()):
    """Several views whose row / column names are drawn from common pools so that every pair of views shares a PART of
    its names at DIFFERENT positions (the situation of test-resnmtf.R:140-184, generalised to many views): the integer
    maps of star_prod_relevant (R/utils.r:63-78) are then neither identities nor complete, and the coupling-count
    buckets of the update kernels are selected by the number of views.  Restriction weights vary per pair (symmetric,
    zero diagonal -- what init_rest_mats returns, R/update_steps.r:12-24), some pairs have weight zero.
    ``na_pairs``: pairs (v, w) given disjoint ROW names (the reference's NA, R/utils.r:70), weight kept.
    ``same_order_views``: views that take the pool's first names in pool order (identity maps among them)."""
    from resnmtf_amd import synth
    from resnmtf_amd.synth import Problem
    rng = np.random.default_rng(seed)
    n_v = len(shapes)

    def names_for(sizes, prefix):
        pool = int(max(sizes) / overlap) + 1
        out, idx = [], []
        for v, sz in enumerate(sizes):
            pick = np.arange(sz) if v in same_order_views else rng.permutation(pool)[:sz]
            out.append([f"{prefix}{t}" for t in pick])
            idx.append((np.asarray(pick), pool))
        return out, idx
    rn, ridx = names_for([s[0] for s in shapes], "r")
    cn, cidx = names_for([s[1] for s in shapes], "c")
    for (v, w) in na_pairs:                     # view w gets row names nobody else has
        rn[w] = [f"only{w}_{t}" for t in range(shapes[w][0])]
    # planted blocks BY NAME (name t of a pool of P belongs to block floor(t k / P)): the same name sits in the same
    # bicluster in every view, so the restrictions pull consistent rows together (test-resnmtf.R:38-52, :140-160)
    data, f0, s0, g0 = [], [], [], []
    for v, (n, m) in enumerate(shapes):
        rb = (ridx[v][0] * k) // ridx[v][1]; cb = (cidx[v][0] * k) // cidx[v][1]
        x = 10.0 * (rb[:, None] == cb[None, :]) + 0.1 * np.abs(rng.standard_normal((n, m)))
        data.append(x / x.sum(axis=0)[None, :])
        f, s, g = synth.random_init(n, m, k, seed * 100 + 50 + v)
        f0.append(f); s0.append(s); g0.append(g)

    def weights(scale):
        if scale == 0.0:
            return np.zeros((n_v, n_v))
        a = rng.uniform(0.3, 1.0, size=(n_v, n_v)) * scale
        a = np.triu(a, 1)
        if n_v > 3:
            a[0, n_v - 1] = 0.0                 # one uncoupled pair
        return a + a.T
    return Problem(data, f0, s0, g0, weights(phi_w), weights(xi_w), weights(psi_w), k, f"{n_v} views coupled",
                   row_names=rn, col_names=cn)
