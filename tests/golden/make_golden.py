"""Generates the golden fixtures in this directory.

RESTATEMENT-DERIVED: the reference (eso28599/resnmtf) is an R package and cannot be run in
the build container or on the GPU box (no R interpreter), and its own tests hold no numeric
vectors.  These files are therefore made by ``oracle/resnmtf_oracle.py`` -- the literal fp64
restatement of R/update_steps.r:141-319, R/utils.r:39-78,157-195, R/main.r:48-130 and
R/obtain_bicl.r:162-180 -- on small seeded inputs.  They pin the oracle against silent edits
and give the HIP path committed input/output pairs that need no generator at test time.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Each .npz holds the inputs (data, initial factors, restriction matrices, names) and the
expected outputs after ``n_iters`` fixed sweeps.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import resnmtf_oracle as O  # noqa: E402


def planted(n, m, k, rng, height=10.0, noise=0.1, perm_rows=None, perm_cols=None):
    r = np.zeros((n, k)); c = np.zeros((m, k))
    for p in range(k):
        r[(p * n) // k:((p + 1) * n) // k, p] = 1
        c[(p * m) // k:((p + 1) * m) // k, p] = 1
    x = height * r @ c.T + noise * np.abs(rng.standard_normal((n, m)))
    if perm_rows is not None:
        x = x[perm_rows]
    if perm_cols is not None:
        x = x[:, perm_cols]
    return O.matrix_normalisation(O.make_non_neg(x))


def rand_init(n, m, k, rng):
    f = rng.uniform(0.1, 1.0, (n, k)); g = rng.uniform(0.1, 1.0, (m, k))
    f /= f.sum(0); g /= g.sum(0)
    s = np.eye(k) + np.abs(rng.normal(0, np.sqrt(0.05), (k, k)))
    return f, s, g


def pack(name, data, f0, s0, g0, phi, xi, psi, row_names, col_names, n_iters):
    res = O.res_nmtf_inner(data, f0, s0, g0, phi, xi, psi, row_names=row_names, col_names=col_names,
                           n_iters=n_iters)
    out = {"n_views": len(data), "n_iters": n_iters, "phi": phi, "xi": xi, "psi": psi,
           "all_error": res["All_Error"], "error": res["Error"]}
    for v in range(len(data)):
        out[f"x{v}"] = data[v]; out[f"f0_{v}"] = f0[v]; out[f"s0_{v}"] = s0[v]; out[f"g0_{v}"] = g0[v]
        out[f"rn{v}"] = np.array(row_names[v]); out[f"cn{v}"] = np.array(col_names[v])
        out[f"out_f{v}"] = res["output_f"][v]; out[f"out_s{v}"] = res["output_s"][v]
        out[f"out_g{v}"] = res["output_g"][v]
        out[f"raw_f{v}"] = res["raw"]["f"][v]; out[f"raw_s{v}"] = res["raw"]["s"][v]
        out[f"raw_g{v}"] = res["raw"]["g"][v]
        out[f"rc{v}"] = res["row_clusters"][v]; out[f"cc{v}"] = res["col_clusters"][v]
        out[f"lam{v}"] = res["lambda"][v]; out[f"mu{v}"] = res["mu"][v]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, final err {res['Error']:.6g}")


def main():
    z1 = np.zeros((1, 1))
    # g1: single view 60x40, k=3, uncoupled (NaN->1 branches, R/update_steps.r:152-155,190-193,226-229)
    rng = np.random.default_rng(101)
    x = planted(60, 40, 3, rng)
    f, s, g = rand_init(60, 40, 3, rng)
    pack("g1_single_60x40_k3", [x], [f], [s], [g], z1, z1, z1,
         [[f"row_{i}" for i in range(1, 61)]], [[f"col_{i}" for i in range(1, 41)]], 30)

    # g2: two views, rows partially shared and at DIFFERENT positions, phi only
    # (the situation of tests/testthat/test-resnmtf.R:140-184)
    rng = np.random.default_rng(102)
    n0, n1 = 90, 75
    names0 = [f"r{i}" for i in range(n0)]
    names1_sorted = [f"r{i}" for i in range(30, 30 + n1)]          # r30..r104 : 60 shared with view 0
    perm1 = rng.permutation(n1)
    names1 = [names1_sorted[p] for p in perm1]
    x0 = planted(n0, 60, 3, rng)
    x1 = planted(n1, 50, 3, rng, perm_rows=perm1)
    data = [x0, x1]
    inits = [rand_init(n0, 60, 3, rng), rand_init(n1, 50, 3, rng)]
    phi = O.init_rest_mats(np.array([[0.0, 200.0], [0.0, 0.0]]), 2)
    z2 = np.zeros((2, 2))
    pack("g2_two_views_phi_partial", data, [i[0] for i in inits], [i[1] for i in inits], [i[2] for i in inits],
         phi, z2, z2, [names0, names1],
         [[f"c0_{i}" for i in range(60)], [f"c1_{i}" for i in range(50)]], 40)

    # g3: three views, phi + psi + xi all non-zero, unequal n and m, one pair with NO shared columns
    rng = np.random.default_rng(103)
    shapes = [(70, 50), (64, 44), (56, 50)]
    k = 4
    rown = [[f"r{i}" for i in range(70)],
            [f"r{i}" for i in rng.permutation(np.arange(10, 74))],          # 60 shared with view 0
            [f"r{i}" for i in rng.permutation(np.arange(0, 56))]]           # subset of view 0
    coln = [[f"c{i}" for i in range(50)],
            [f"d{i}" for i in range(44)],                                   # shares no column with anyone (NA)
            [f"c{i}" for i in rng.permutation(np.arange(0, 50))]]           # same columns as view 0, shuffled
    data = [planted(n, m, k, rng) for n, m in shapes]
    inits = [rand_init(n, m, k, rng) for n, m in shapes]
    up = np.triu(np.ones((3, 3)), 1)
    phi = O.init_rest_mats(2.0 * up, 3)          # moderate weights: data and coupling terms both matter
    psi = O.init_rest_mats(1.5 * up, 3)
    xi = O.init_rest_mats(0.5 * up, 3)
    pack("g3_three_views_phi_psi_xi", data, [i[0] for i in inits], [i[1] for i in inits], [i[2] for i in inits],
         phi, xi, psi, rown, coln, 40)

    # g4: dead component -> 0/0 in every update rule, exercising NaN -> 1 (single view, k = 5)
    rng = np.random.default_rng(104)
    x = planted(48, 36, 4, rng)
    f, s, g = rand_init(48, 36, 5, rng)
    f[:, 2] = 0.0; g[:, 2] = 0.0; s[2, :] = 0.0; s[:, 2] = 0.0
    pack("g4_dead_component_nan", [x], [f], [s], [g], z1, z1, z1,
         [[f"row_{i}" for i in range(1, 49)]], [[f"col_{i}" for i in range(1, 37)]], 25)

    # g5: psi non-zero for ONE pair only, so the third view takes the restricted G formula with
    # zero coupling (whole-matrix branch, R/update_steps.r:190) ; k = 20 (two MFMA N-tiles)
    rng = np.random.default_rng(105)
    shapes = [(48, 40), (44, 40), (40, 24)]
    k = 20
    data = [planted(n, m, 4, rng) for n, m in shapes]
    inits = [rand_init(n, m, k, rng) for n, m in shapes]
    psi = O.init_rest_mats(np.array([[0.0, 120.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]), 3)
    z3 = np.zeros((3, 3))
    rown = [[f"a{i}" for i in range(48)], [f"b{i}" for i in range(44)], [f"c{i}" for i in range(40)]]
    coln = [[f"c{i}" for i in range(40)], [f"c{i}" for i in range(40)], [f"e{i}" for i in range(24)]]
    pack("g5_psi_one_pair_k20", data, [i[0] for i in inits], [i[1] for i in inits], [i[2] for i in inits],
         z3, z3, psi, rown, coln, 30)


if __name__ == "__main__":
    main()
