"""CPU ORACLE for the ResNMTF multiplicative-update inner loop  --  TEST INFRASTRUCTURE ONLY.

This file is a *literal* fp64 NumPy restatement of the reference's arithmetic for the one
path this repository accelerates.  It is the checker the HIP path is
compared against.  It is NOT part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing
under ``resnmtf_amd/`` does, and the product path has no CPU fallback.

PARITY UNPINNED (by the reference's own vectors): the reference (eso28599/resnmtf) is a
pure-R package; there is no R interpreter in the build container or on the GPU box, and the
reference's tests hold no numeric golden vectors (only shape / column-sum / planted-cluster
recovery properties on unseeded data, tests/testthat/test-resnmtf.R:38-184).  The restatement
is therefore pinned by (i) those property tests re-created with seeds in
``tests/test_oracle_properties.py`` and (ii) line-by-line citation below.  The golden files in
``tests/golden/`` are *restatement-derived* (made by ``tests/golden/make_golden.py`` from
this file).

Every function cites the reference file:line it follows (paths relative to the reference
repository root).  Operation order (association of matrix products, order of additions) is
the reference's; R's ``%*%`` is left-associative.

Conventions
-----------
* Views are 0-based here (the reference is 1-based).
* ``names[v]`` is the list of row (or column) names of view v.  ``shared[v][w]`` is the
  vector of names shared between views v and w, or ``None`` for the reference's ``NA``
  ("no shared names", R/utils.r:587-596).
* Restriction matrices ``phi/xi/psi`` are (V, V) float arrays *already symmetrised with a
  zero diagonal* as produced by ``init_rest_mats`` (R/update_steps.r:12-24).
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Optional, Sequence

import numpy as np

__all__ = [
    "init_rest_mats", "star_prod", "star_prod_relevant", "update_f", "update_g", "update_s",
    "update_lm", "update_matrices", "calculate_error", "normalisation_check",
    "binary_clusters", "res_nmtf_inner", "init_mats_inner", "matrix_normalisation",
    "make_non_neg", "give_names", "reorder_data", "explicit_init_lm",
]


# --------------------------------------------------------------------------------------
# host-side hygiene that defines the inputs of the path
# --------------------------------------------------------------------------------------
def init_rest_mats(mat: Optional[np.ndarray], n_v: int) -> np.ndarray:
    """R/update_steps.r:12-24 -- NULL -> zeros; else diag<-0 and M + t(M)."""
    if mat is None:
        return np.zeros((n_v, n_v))
    m = np.array(mat, dtype=np.float64, copy=True)
    np.fill_diagonal(m, 0.0)                       # update_steps.r:21
    return m + m.T                                 # update_steps.r:22


def make_non_neg(x: np.ndarray) -> np.ndarray:
    """R/utils.r:20-27 -- per-COLUMN shift by |min(0, min(col))|."""
    x = np.asarray(x, dtype=np.float64)
    shift = np.abs(np.minimum(0.0, x.min(axis=0)))
    return x + shift[None, :]


def matrix_normalisation(x: np.ndarray) -> np.ndarray:
    """R/utils.r:86-88 -- sweep(matrix, 2, colSums(matrix), '/')."""
    x = np.asarray(x, dtype=np.float64)
    return x / x.sum(axis=0)[None, :]


def give_names(data: Sequence[np.ndarray], phi=None, psi=None,
               row_names=None, col_names=None):
    """R/utils.r:469-542 for the two supported situations: every view unnamed (auto names
    ``row_<n>``/``col_<n>`` with a running counter, names copied to view j>i when
    phi[i,j]>0 / psi[i,j]>0, R/utils.r:474-491,503-521) or every view named by the caller."""
    n_views = len(data)

    def _auto(prefix, sizes, rest):
        names: List[Optional[List[str]]] = [None] * n_views
        n = 1
        for i in range(n_views):
            if names[i] is None:                                   # utils.r:477 / 506
                names[i] = [f"{prefix}_{t}" for t in range(n, n + sizes[i])]
                n += sizes[i]
            if rest is not None:                                   # utils.r:483-491
                for j in range(min(i + 1, n_views - 1), n_views):
                    if rest[i, j] > 0 and sizes[i] != sizes[j]:
                        raise ValueError("restriction implies shared unnamed rows/cols of differing number")
                    elif rest[i, j] > 0:
                        names[j] = list(names[i])
        return names

    if row_names is None:
        row_names = _auto("row", [d.shape[0] for d in data], None if phi is None else np.asarray(phi))
    if col_names is None:
        col_names = _auto("col", [d.shape[1] for d in data], None if psi is None else np.asarray(psi))
    return [list(r) for r in row_names], [list(c) for c in col_names]


def reorder_data(names: Sequence[Sequence[str]]):
    """R/utils.r:619-662 + 560-601 for one axis: power-set partition of the names into
    'existing view subsets' (names present in exactly the views of subset A), then for every
    ordered pair (v, w) the concatenation of the name lists of all subsets containing both;
    empty -> None (the reference's NA)."""
    n_views = len(names)
    sets = [set(nm) for nm in names]
    subsets, lists = [], []
    views_all = list(range(n_views))
    # rje::powerSetCond enumerates the non-empty subsets; order is irrelevant to the result
    for size in range(1, n_views + 1):
        for views in itertools.combinations(views_all, size):
            neg = [u for u in views_all if u not in views]
            rows = [nm for nm in names[views[0]] if all(nm in sets[u] for u in views[1:])]   # Reduce(intersect)
            neg_union = set().union(*[sets[u] for u in neg]) if neg else set()
            rows_in_a = [nm for nm in rows if nm not in neg_union]                          # setdiff
            if len(rows_in_a) != 0:
                subsets.append(views)
                lists.append(rows_in_a)
    shared: List[Dict[int, Optional[List[str]]]] = []
    for v1 in range(n_views):
        d: Dict[int, Optional[List[str]]] = {}
        for v2 in range(n_views):
            if v2 == v1:
                continue
            common = [nm for sub, lst in zip(subsets, lists) if (v1 in sub and v2 in sub) for nm in lst]
            d[v2] = common if len(common) != 0 else None            # utils.r:587-591
        shared.append(d)
    return shared


# --------------------------------------------------------------------------------------
# the hot path
# --------------------------------------------------------------------------------------
def star_prod(vec: np.ndarray, mat_list: Sequence[np.ndarray]):
    """R/utils.r:39-47."""
    vec_mat = 0.0
    for i in range(len(vec)):
        if vec[i] != 0:
            vec_mat = vec_mat + vec[i] * mat_list[i]
    return vec_mat


def star_prod_relevant(vec, mat_list, current_mat, indices, names_v, names_all):
    """R/utils.r:63-78.  ``indices[w]`` = shared NAMES with view w or None (NA).  Rows are
    matched BY NAME on both sides (utils.r:71)."""
    vec_mat = 0.0
    for i in range(len(vec)):
        if vec[i] != 0:
            masked = np.array(current_mat, copy=True)               # utils.r:67
            rows = indices[i]                                       # utils.r:69
            if rows is not None:                                    # utils.r:70  !any(is.na(rows))
                pos_v = _positions(names_v, rows)
                pos_i = _positions(names_all[i], rows)
                masked[pos_v, :] = mat_list[i][pos_i, :]            # utils.r:71
                vec_mat = vec_mat + vec[i] * masked * mat_list[i].shape[0]   # utils.r:73
    return vec_mat / current_mat.shape[0]                           # utils.r:77


_pos_cache: Dict[int, tuple] = {}


def _positions(names: Sequence[str], wanted: Sequence[str]) -> np.ndarray:
    """name -> position lookup (R character indexing; first occurrence wins).  The lookup
    table is cached per names-object (the object is kept alive by the cache entry)."""
    entry = _pos_cache.get(id(names))
    if entry is None or entry[0] is not names:
        lut: Dict[str, int] = {}
        for p, nm in enumerate(names):
            lut.setdefault(nm, p)
        if len(_pos_cache) > 64:
            _pos_cache.clear()
        entry = (names, lut)
        _pos_cache[id(names)] = entry
    lut = entry[1]
    return np.fromiter((lut[w] for w in wanted), dtype=np.int64, count=len(wanted))


def update_f(x, input_f, input_s, input_g, lambda_in, phi, v, row_indices, names_v, names_all):
    """R/update_steps.r:141-165."""
    current_f = input_f[v]
    numerator = (x @ input_g) @ input_s.T                                          # :146
    denominator = (current_f @ input_s) @ ((input_g.T @ input_g) @ input_s.T)      # :147-148
    phi_vec = phi[:, v]                                                            # :150
    lambda_mat = 0.5 * np.broadcast_to(lambda_in[None, :], current_f.shape)        # :151
    if np.sum(phi_vec) == 0:                                                       # :152
        with np.errstate(divide="ignore", invalid="ignore"):
            mat = numerator / (denominator + lambda_mat)                           # :153
        mat[np.isnan(mat)] = 1.0                                                   # :154
        output_f = current_f * mat                                                 # :155
    else:
        num_prod = star_prod_relevant(phi_vec, input_f, current_f, row_indices, names_v, names_all)  # :157
        denom_prod = np.sum(phi_vec) * current_f                                   # :158
        with np.errstate(divide="ignore", invalid="ignore"):
            output_f = current_f * ((numerator + num_prod) /
                                    (denominator + denom_prod + lambda_mat))       # :159-162
    return np.abs(output_f)                                                        # :164


def update_g(x, input_f, input_s, input_g, mu_in, psi, v, col_indices, names_v, names_all):
    """R/update_steps.r:180-207.  NOTE the branch is on sum(psi) of the WHOLE matrix (:190)."""
    current_g = input_g[v]
    numerator = (x.T @ input_f) @ input_s                                          # :185
    denominator = (current_g @ input_s.T) @ ((input_f.T @ input_f) @ input_s)      # :186-187
    mu_mat = 0.5 * np.broadcast_to(mu_in[None, :], current_g.shape)                # :188
    if np.sum(psi) == 0:                                                           # :190
        with np.errstate(divide="ignore", invalid="ignore"):
            mat = numerator / (denominator + mu_mat)                               # :191
        mat[np.isnan(mat)] = 1.0                                                   # :192
        output_g = current_g * mat                                                 # :193
    else:
        psi_vec = psi[:, v]                                                        # :195
        num_prod = star_prod_relevant(psi_vec, input_g, current_g, col_indices, names_v, names_all)  # :196-199
        denom_prod = np.sum(psi_vec) * current_g                                   # :200
        with np.errstate(divide="ignore", invalid="ignore"):
            output_g = current_g * ((numerator + num_prod) /
                                    (denominator + denom_prod + mu_mat))           # :201-204
    return np.abs(output_g)                                                        # :206


def update_s(x, input_f, input_s, input_g, xi, v):
    """R/update_steps.r:220-240.  Branch on sum(xi) of the WHOLE matrix (:226)."""
    current_s = input_s[v]
    numerator = (input_f.T @ x) @ input_g                                          # :223
    denominator = ((input_f.T @ input_f) @ current_s) @ (input_g.T @ input_g)      # :224
    if np.sum(xi) == 0:                                                            # :226
        with np.errstate(divide="ignore", invalid="ignore"):
            mat = numerator / denominator                                          # :227
        mat[np.isnan(mat)] = 1.0                                                   # :228
        output_s = current_s * mat                                                 # :229
    else:
        xi_vec = xi[:, v]                                                          # :231
        num_prod = star_prod(xi_vec, input_s)                                      # :232
        denom_prod = np.sum(xi_vec) * current_s                                    # :233
        with np.errstate(divide="ignore", invalid="ignore"):
            output_s = current_s * ((numerator + num_prod) / (denominator + denom_prod))   # :234-237
    return np.abs(output_s)                                                        # :239


def update_lm(vec, matrix):
    """R/update_steps.r:249-251."""
    return matrix.sum(axis=0) * vec


def update_matrices(x, input_f, input_s, input_g, lam, mu, phi, xi, psi,
                    row_indices, col_indices, row_names, col_names):
    """R/update_steps.r:272-319 -- one Gauss-Seidel sweep, views in index order, in place in
    the running lists."""
    n_v = len(x)
    cur_f, cur_s, cur_g = list(input_f), list(input_s), list(input_g)              # :276-278
    cur_lam, cur_mu = list(lam), list(mu)                                          # :279-280
    for v in range(n_v):                                                           # :282
        cur_f[v] = update_f(x[v], cur_f, cur_s[v], cur_g[v], cur_lam[v], phi, v,
                            row_indices[v], row_names[v], row_names)               # :284-293
        cur_g[v] = update_g(x[v], cur_f[v], cur_s[v], cur_g, cur_mu[v], psi, v,
                            col_indices[v], col_names[v], col_names)               # :295-303
        cur_s[v] = update_s(x[v], cur_f[v], cur_s, cur_g[v], xi, v)                # :305-311
        cur_lam[v] = update_lm(cur_lam[v], cur_f[v])                               # :312
        cur_mu[v] = update_lm(cur_mu[v], cur_g[v])                                 # :313
    return cur_f, cur_s, cur_g, cur_lam, cur_mu


def calculate_error(data, cur_f, cur_s, cur_g, data_norms):
    """R/utils.r:157-166 -- explicit residual, x_hat materialised."""
    err = np.zeros(len(data))
    for v in range(len(data)):
        x_hat = (cur_f[v] @ cur_s[v]) @ cur_g[v].T                                 # :161
        err[v] = np.linalg.norm(data[v] - x_hat, "fro") ** 2                       # :162
    return err / data_norms                                                        # :164


def normalisation_check(cur_f, cur_g, cur_s):
    """R/utils.r:176-195 -- S column sweep by cF*cG (pre-normalisation sums), then F, G."""
    out_f, out_g, out_s = [], [], []
    for f, g, s in zip(cur_f, cur_g, cur_s):
        cf, cg = f.sum(axis=0), g.sum(axis=0)
        out_s.append(s * (cf * cg)[None, :])                                       # :178-181
        with np.errstate(divide="ignore", invalid="ignore"):                       # 0/0 -> NaN as in R
            out_f.append(f / cf[None, :])                                          # :182-185
            out_g.append(g / cg[None, :])                                          # :186-189
    return out_f, out_g, out_s


def binary_clusters(out_f, out_g, out_s):
    """R/obtain_bicl.r:162-180 with remove_spurious = FALSE: thresholds 1/n, 1/m, pairing by
    which.max of each S column (first maximum), row clusters re-ordered by it."""
    row_cl, col_cl = [], []
    for f, g, s in zip(out_f, out_g, out_s):
        rc = (f > (1.0 / f.shape[0])).astype(np.float64)                           # :163-166
        cc = (g > (1.0 / g.shape[0])).astype(np.float64)                           # :168-171
        relations = np.argmax(s, axis=0)                                           # :179 (first max)
        row_cl.append(rc[:, relations])                                            # :180
        col_cl.append(cc)
    return row_cl, col_cl


def explicit_init_lm(init_f, init_g):
    """R/update_steps.r:49-56 -- explicit-init branch: lambda = colSums(F), mu = colSums(G)."""
    return [f.sum(axis=0) for f in init_f], [g.sum(axis=0) for g in init_g]


def init_mats_inner(x, k_vec, rng: np.random.Generator, sigma: float = 0.05):
    """R/update_steps.r:78-125.  The reference draws the noise with MASS::mvrnorm from R's
    RNG (not reproducible here); ``rng.normal`` with the same covariance (sigma * I) stands in
    -- statistically, not bitwise, equivalent."""
    init_f, init_s, init_g, init_lam, init_mu = [], [], [], [], []
    for xi_, k in zip(x, k_vec):
        u, d, vt = np.linalg.svd(xi_, full_matrices=False)                         # :92
        f = np.abs(u[:, :k])                                                       # :93
        g = np.abs(vt.T[:, :k])                                                    # :94
        s = np.abs(np.diag(d)[:k, :k])                                             # :95
        s = s + np.abs(rng.normal(0.0, np.sqrt(sigma), size=(k, k)))               # :96-99
        cf, cg = f.sum(axis=0), g.sum(axis=0)                                      # :100-101
        s = s * (cf * cg)[None, :]                                                 # :102-105
        f = f / cf[None, :]                                                        # :106-109
        g = g / cg[None, :]                                                        # :110-113
        init_f.append(f); init_s.append(s); init_g.append(g)
        init_lam.append(f.sum(axis=0)); init_mu.append(g.sum(axis=0))              # :114-115
    return init_f, init_s, init_g, init_lam, init_mu


def res_nmtf_inner(data, init_f, init_s, init_g, phi, xi, psi,
                   row_names=None, col_names=None, row_indices=None, col_indices=None,
                   n_iters: Optional[int] = None, max_iters: Optional[int] = None,
                   init_lam=None, init_mu=None, tol: float = 1.0e-6):
    """R/main.r:32-140 with explicit inits and no_clusts-style outputs plus the binary
    matrices of R/obtain_bicl.r:162-180 (spurious = FALSE).

    ``n_iters=None`` -> convergence loop (main.r:50-81), else fixed count (main.r:83-108).
    ``max_iters`` is a guard the reference lacks (its while-loop has no cap, main.r:55).
    """
    n_v = len(data)
    data = [np.asarray(d, dtype=np.float64) for d in data]
    if row_names is None or col_names is None:
        rn, cn = give_names(data, phi, psi)
        row_names = row_names or rn
        col_names = col_names or cn
    if row_indices is None:
        row_indices = reorder_data(row_names)
    if col_indices is None:
        col_indices = reorder_data(col_names)
    cur_f = [np.array(f, dtype=np.float64) for f in init_f]
    cur_s = [np.array(s, dtype=np.float64) for s in init_s]
    cur_g = [np.array(g, dtype=np.float64) for g in init_g]
    if init_lam is None or init_mu is None:
        cur_lam, cur_mu = explicit_init_lm(cur_f, cur_g)                           # update_steps.r:55-56
    else:
        cur_lam = [np.array(t, dtype=np.float64) for t in init_lam]
        cur_mu = [np.array(t, dtype=np.float64) for t in init_mu]
    data_norms = np.array([np.linalg.norm(d, "fro") ** 2 for d in data])           # main.r:48
    total_err: List[float] = []

    def sweep():
        nonlocal cur_f, cur_s, cur_g, cur_lam, cur_mu
        cur_f, cur_s, cur_g, cur_lam, cur_mu = update_matrices(
            data, cur_f, cur_s, cur_g, cur_lam, cur_mu, phi, xi, psi,
            row_indices, col_indices, row_names, col_names)
        return float(np.mean(calculate_error(data, cur_f, cur_s, cur_g, data_norms)))

    if n_iters is None:                                                            # main.r:50
        err_diff, err_temp = 1.0, 0.0                                              # :53-54
        while err_diff > tol:                                                      # :55
            mean_err = sweep()
            total_err.append(mean_err)                                             # :78
            err_diff = abs(mean_err - err_temp)                                    # :79
            err_temp = total_err[-1]                                               # :80
            if max_iters is not None and len(total_err) >= max_iters:
                break
        error = float(np.mean(total_err[-10:]))                                    # :127
    else:
        for _ in range(n_iters):                                                   # :84
            total_err.append(sweep())                                              # :104-107
        error = total_err[-1]                                                      # :129
    raw = dict(f=[a.copy() for a in cur_f], s=[a.copy() for a in cur_s], g=[a.copy() for a in cur_g])
    out_f, out_g, out_s = normalisation_check(cur_f, cur_g, cur_s)                 # main.r:110
    row_cl, col_cl = binary_clusters(out_f, out_g, out_s)
    return {
        "output_f": out_f, "output_s": out_s, "output_g": out_g,
        "Error": error, "All_Error": np.array(total_err),
        "row_clusters": row_cl, "col_clusters": col_cl,
        "lambda": cur_lam, "mu": cur_mu, "raw": raw,
    }
