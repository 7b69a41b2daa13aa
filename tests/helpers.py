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
