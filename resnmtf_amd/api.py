"""Host-side mirror of the reference's two exported functions for the path this repository
accelerates: ``res_nmtf_inner`` (``R/main.r:32-140``) and ``apply_resnmtf``
(``R/main.r:214-335``).  Same argument names and meaning, same keys in the result; the
loop itself (update_matrices x T, calculate_error x T, normalisation_check, binary cluster
matrices) runs in the HIP library through the C-ABI.  R is not available in the build or
run environment, so this Python module is the tested stand-in for the thin R wrapper shown
in INTEGRATION.md.

Deliberately NOT implemented here (out of scope): spurious-bicluster
removal, the bisilhouette score, the k sweep and stability selection -- they are statistics
on top of finished factorisations and stay on the R side.  Requests for them raise
``NotImplementedError`` instead of silently doing something else.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import naming
from .engine import Engine

_DISTANCES = ("euclidean", "manhattan", "cosine")


def _as_list(x):
    if isinstance(x, np.ndarray) and x.ndim == 2:
        return [x]                      # check_lists, R/utils.r:313-316
    return list(x)


def svd_init(data: Sequence[np.ndarray], k_vec: Sequence[int], seed: Optional[int] = None, sigma: float = 0.05):
    """``init_mats_inner`` (``R/update_steps.r:78-125``) on the host, as in the reference
    (the initialisation is outside the accelerated loop).  The noise on S comes from NumPy's
    generator instead of ``MASS::mvrnorm`` + R's RNG: statistically, not bitwise, equivalent."""
    rng = np.random.default_rng(seed)
    init_f, init_s, init_g, init_lam, init_mu = [], [], [], [], []
    for x, k in zip(data, k_vec):
        u, d, vt = np.linalg.svd(x, full_matrices=False)
        f = np.abs(u[:, :k]); g = np.abs(vt.T[:, :k])
        s = np.abs(np.diag(d)[:k, :k]) + np.abs(rng.normal(0.0, np.sqrt(sigma), size=(k, k)))
        cf, cg = f.sum(axis=0), g.sum(axis=0)
        s = s * (cf * cg)[None, :]
        f = f / cf[None, :]; g = g / cg[None, :]
        init_f.append(f); init_s.append(s); init_g.append(g)
        init_lam.append(f.sum(axis=0)); init_mu.append(g.sum(axis=0))
    return init_f, init_s, init_g, init_lam, init_mu


def _load_engine(eng: Engine, data, init_f, init_s, init_g, lam, mu, phi, xi, psi,
                 row_names, col_names, row_indices, column_indices, seed=None):
    """``init_f is None``: the initial factors come from the device (``resnmtf_init_svd``)."""
    n_v = eng.n_views
    for v in range(n_v):
        if eng.owned[v]:
            eng.set_view(v, data[v])
        if init_f is None:
            eng.init_svd(v, seed=(0 if seed is None else int(seed)) + v)                          # update_steps.r:78-125
        else:
            eng.set_factors(v, init_f[v], init_s[v], init_g[v],
                            None if lam is None else lam[v], None if mu is None else mu[v])
    eng.set_restrictions(phi, xi, psi)
    for v in range(n_v):
        for w in range(n_v):
            if w == v:
                continue
            if n_v > 1:
                iv, iw = naming.index_pairs(row_names[v], row_names[w], row_indices[v].get(w))
                eng.set_shared_rows(v, w, iv, iw)
                iv, iw = naming.index_pairs(col_names[v], col_names[w], column_indices[v].get(w))
                eng.set_shared_cols(v, w, iv, iw)


def res_nmtf_inner(data, row_indices, column_indices,
                   init_f=None, init_s=None, init_g=None,
                   k_vec=None, phi=None, xi=None, psi=None,
                   n_iters=None, num_repeats=5, spurious=True, distance="euclidean",
                   no_clusts=False, *, row_names=None, col_names=None, device_id: int = 0,
                   max_iters: int = 100000, seed: Optional[int] = None, engine_opts: Optional[dict] = None,
                   host_init: bool = False, return_init: bool = False):
    """``res_nmtf_inner`` (``R/main.r:32-140``).

    ``data``: list of pre-processed (non-negative, column-normalised) matrices; ``row_indices[v][w]``
    / ``column_indices[v][w]``: shared names between views v and w or None (NA), as produced by
    ``naming.shared_names``; ``phi/xi/psi``: symmetrised restriction matrices (the reference's
    ``res_nmtf_inner`` needs them non-NULL, ``R/update_steps.r:150``); ``n_iters=None`` runs to
    convergence.  Keyword-only extras: ``row_names``/``col_names`` (the reference reads them off
    the matrices' dimnames), ``max_iters`` (a guard the reference lacks), ``seed`` for the SVD
    init noise, ``host_init`` (without explicit initial factors: ``False`` = ``init_mats_inner`` on
    the device, randomized top-k SVD on the pass kernels, milliseconds; ``True`` = NumPy's full SVD
    on the host as the reference's ``svd()``, seconds to minutes -- statistically equivalent),
    ``return_init`` (adds ``"init"``: the (F, S, G, lambda, mu) per view the loop started from).
    """
    data = [np.asarray(d, dtype=np.float64) for d in _as_list(data)]
    n_v = len(data)
    if k_vec is None:
        raise ValueError("k_vec is required")
    k_vec = [int(k) for k in np.atleast_1d(k_vec)]
    if len(k_vec) != n_v:
        raise ValueError("k_vec must be a vector of the same length as the number of views.")   # utils.r:440
    if not no_clusts and spurious:
        raise NotImplementedError(
            "spurious-bicluster removal (R/obtain_bicl.r:31-133) is outside the accelerated path; "
            "pass spurious=False or do it on the R side (INTEGRATION.md).")
    if distance not in _DISTANCES:
        raise ValueError("distance must be one of 'euclidean', 'manhattan' or 'cosine'.")         # utils.r:425
    phi = np.zeros((n_v, n_v)) if phi is None else np.asarray(phi, dtype=np.float64)
    xi = np.zeros((n_v, n_v)) if xi is None else np.asarray(xi, dtype=np.float64)
    psi = np.zeros((n_v, n_v)) if psi is None else np.asarray(psi, dtype=np.float64)
    if row_names is None or col_names is None:
        rn, cn = naming.give_names(data, None, None)
        row_names = row_names or rn
        col_names = col_names or cn
    if row_indices is None:
        row_indices = naming.shared_names(row_names)
    if column_indices is None:
        column_indices = naming.shared_names(col_names)

    lam = mu = None
    if init_f is None or init_g is None or init_s is None:                                        # update_steps.r:41
        init_f = init_s = init_g = None
        if host_init:
            init_f, init_s, init_g, lam, mu = svd_init(data, k_vec, seed)
    if init_f is not None:
        init_f, init_s, init_g = _as_list(init_f), _as_list(init_s), _as_list(init_g)

    eng = Engine([d.shape[0] for d in data], [d.shape[1] for d in data], k_vec, device_id=device_id,
                 **(engine_opts or {}))
    try:
        _load_engine(eng, data, init_f, init_s, init_g, lam, mu, phi, xi, psi,
                     row_names, col_names, row_indices, column_indices, seed=seed)
        init_state = [eng.get_factors(v) for v in range(n_v)] if return_init else None      # (F, S, G, lambda, mu) the loop starts from
        total_err = eng.run(n_iters=n_iters, tol=1.0e-6, max_iters=max_iters)
        out_f, out_s, out_g, row_cl, col_cl, lams, mus = [], [], [], [], [], [], []
        for v in range(n_v):
            f, s, g, rc, cc = eng.finalise(v)                                                     # main.r:110 + obtain_bicl.r:162-180
            out_f.append(f); out_s.append(s); out_g.append(g); row_cl.append(rc); col_cl.append(cc)
            _, _, _, lv, mv = eng.get_factors(v)
            lams.append(lv); mus.append(mv)
    finally:
        eng.close()
    if no_clusts:                                                                                 # main.r:115-120
        res = {"output_f": out_f, "output_s": out_s, "output_g": out_g}
        if return_init:
            res["init"] = init_state
        return res
    if n_iters is None:
        error = float(np.mean(total_err[-10:]))                                                   # main.r:127
    else:
        error = float(total_err[-1])                                                              # main.r:129
    res = {
        "output_f": out_f, "output_s": out_s, "output_g": out_g,
        "Error": error, "All_Error": total_err,
        "bisil": None,            # bisilhouette::bisilhouette is not available offline
        "row_clusters": row_cl, "col_clusters": col_cl,
        "lambda": lams, "mu": mus,
    }
    if return_init:               # (test hook: the initial state the device built, for a reference run from the same start)
        res["init"] = init_state
    return res


def apply_resnmtf(data, init_f=None, init_s=None, init_g=None, k_val=None,
                  phi=None, xi=None, psi=None, n_iters=None, k_min=3, k_max=8,
                  distance="euclidean", spurious=True, num_repeats=5, no_clusts=False,
                  sample_rate=0.9, n_stability=5, stability=True, stab_thres=0.4,
                  remove_unstable=True, use_parallel=True, *, row_names=None, col_names=None,
                  device_id: int = 0, max_iters: int = 100000, seed: Optional[int] = None):
    """``apply_resnmtf`` (``R/main.r:214-335``) for a known ``k_val`` without stability
    selection: naming, shared-name maps, restriction symmetrisation, non-negativity shift and
    column normalisation on the host, then the device loop."""
    data = [np.asarray(d, dtype=np.float64) for d in _as_list(data)]
    n_v = len(data)
    if k_val is None:
        raise NotImplementedError("the k sweep (R/main.r:279-321) needs the bisilhouette score, which is "
                                  "outside the accelerated path; pass k_val")
    if stability:
        raise NotImplementedError("stability selection (R/stability_analysis.r) is outside the accelerated "
                                  "path; pass stability=False or do it on the R side (INTEGRATION.md)")
    for name, val in (("n_iters", n_iters), ("num_repeats", num_repeats), ("n_stability", n_stability)):
        if val is not None and (int(val) != val or val < 1):
            raise ValueError(f"{name} must be a positive integer.")                               # utils.r:220-253
    k_vec = [int(np.atleast_1d(k_val)[0])] * n_v                                                  # main.r:226
    ranks = [d.shape[1] for d in data]
    if any(k < 1 for k in k_vec):
        raise ValueError("k_vec must be a vector of integers greater than 1.")                    # utils.r:437
    if any(k > r for k, r in zip(k_vec, ranks)):
        raise ValueError("k_vec must be a vector of integers less than or equal to the ranks of the views.")
    rn, cn = naming.give_names(data, phi, psi, row_names, col_names)                              # main.r:228
    row_idx, col_idx = naming.shared_names(rn), naming.shared_names(cn)                           # main.r:230
    phi_m = naming.init_rest_mats(phi, n_v)                                                       # main.r:233-235
    psi_m = naming.init_rest_mats(psi, n_v)
    xi_m = naming.init_rest_mats(xi, n_v)
    data = naming.check_data(data)                                                                # main.r:237
    return res_nmtf_inner(data, row_idx, col_idx, init_f, init_s, init_g, k_vec, phi_m, xi_m, psi_m,
                          n_iters, num_repeats, spurious, distance, no_clusts,
                          row_names=rn, col_names=cn, device_id=device_id, max_iters=max_iters, seed=seed)
