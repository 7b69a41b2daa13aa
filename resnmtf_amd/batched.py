"""Batched driver for the repeated factorisations around the hot path.

The reference calls ``res_nmtf_inner`` many times per ``apply_resnmtf``: once per candidate k
(``R/main.r:279-321``), ``num_repeats`` times on shuffled data for spurious-bicluster removal
(``R/obtain_bicl.r:31-42``) and ``n_stability`` times on sub-samples (``R/stability_analysis.r:215-278``)
-- 36 to 66 independent factorisations.  They share nothing, so they are replicas: this module
builds the job lists with the reference's own sampling rules and runs them, sharded round-robin
over the ranks of a ``torch.distributed`` process group when there is one (one process per GPU, no
collective in the data path; the results are gathered as Python objects at the end).

What is NOT here, on purpose: the scores computed from the factorisations (bisilhouette, JSD,
relevance) -- statistics on finished results that stay on the R side (``bisilhouette`` is not even
available offline).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


@dataclass
class Job:
    """One independent factorisation: the arguments of ``apply_resnmtf`` with a known k."""
    data: List[np.ndarray]
    k_val: int
    phi: Optional[np.ndarray] = None
    xi: Optional[np.ndarray] = None
    psi: Optional[np.ndarray] = None
    n_iters: Optional[int] = None
    seed: int = 0
    row_names: Optional[List[List[str]]] = None
    col_names: Optional[List[List[str]]] = None
    tag: str = ""
    extras: Dict = field(default_factory=dict)


def shuffle_view(x: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """``shuffle_view`` (``R/obtain_bicl.r:11-22``): all entries permuted uniformly, redrawn while a
    row or a column sums to zero."""
    x = np.asarray(x, dtype=np.float64)
    while True:
        messed = rng.permutation(x.ravel()).reshape(x.shape)
        if not ((messed.sum(axis=0) == 0).any() or (messed.sum(axis=1) == 0).any()):
            return messed


def subsample_views(data: Sequence[np.ndarray], sample_rate: float, rng: np.random.Generator,
                    row_names: Optional[Sequence[Sequence[str]]] = None,
                    col_names: Optional[Sequence[Sequence[str]]] = None, max_attempts: int = 20):
    """The sampling of ``stability_repeat`` (``R/stability_analysis.r:215-249`` with ``initial_shuffle``
    ``:111-132`` and ``sample_view`` ``:156-193``): view 1 draws ``floor(dim * sample_rate)`` rows and
    columns without replacement; a later view re-uses view 1's draw along every axis on which it
    has view 1's extent, else draws its own; all-zero rows / columns of a sub-sample are dropped
    (from every earlier view sharing that draw).  The sub-samples are NOT re-normalised (Appendix
    B11).  Returns ``(new_data, row_samples, col_samples, new_row_names, new_col_names)`` or ``None``
    after ``max_attempts`` failures (the reference prints and gives up at 20, ``:223-226``)."""
    data = [np.asarray(d, dtype=np.float64) for d in data]
    n_v = len(data)
    dim_1 = data[0].shape
    for _ in range(max_attempts - 1):
        rows: List[np.ndarray] = [None] * n_v
        cols: List[np.ndarray] = [None] * n_v
        new: List[np.ndarray] = [None] * n_v
        rows[0] = rng.choice(dim_1[0], int(dim_1[0] * sample_rate), replace=False)           # :230
        cols[0] = rng.choice(dim_1[1], int(dim_1[1] * sample_rate), replace=False)           # :231
        new[0] = data[0][np.ix_(rows[0], cols[0])]
        if (new[0].sum(axis=0) == 0).any() or (new[0].sum(axis=1) == 0).any():               # :233-240
            keep_c = new[0].sum(axis=0) != 0; keep_r = new[0].sum(axis=1) != 0
            rows[0] = rows[0][keep_r]; cols[0] = cols[0][keep_c]
            new[0] = data[0][np.ix_(rows[0], cols[0])]
        for i in range(1, n_v):
            dims = data[i].shape
            same_r, same_c = dims[0] == dim_1[0], dims[1] == dim_1[1]
            rows[i] = rows[0] if same_r else rng.choice(dims[0], int(dims[0] * sample_rate), replace=False)   # :114-118
            cols[i] = cols[0] if same_c else rng.choice(dims[1], int(dims[1] * sample_rate), replace=False)   # :119-123
            new[i] = data[i][np.ix_(rows[i], cols[i])]
            if (new[i].sum(axis=0) == 0).any() or (new[i].sum(axis=1) == 0).any():           # :165-190
                keep_c = new[i].sum(axis=0) != 0; keep_r = new[i].sum(axis=1) != 0
                for p in (range(i + 1) if same_r else [i]):
                    rows[p] = rows[p][keep_r]
                for p in (range(i + 1) if same_c else [i]):
                    cols[p] = cols[p][keep_c]
                for p in range(i + 1):
                    new[p] = data[p][np.ix_(rows[p], cols[p])]
        if all((d.sum(axis=0) != 0).all() and (d.sum(axis=1) != 0).all() for d in new):      # test_cond
            rn = None if row_names is None else [[row_names[v][t] for t in rows[v]] for v in range(n_v)]
            cn = None if col_names is None else [[col_names[v][t] for t in cols[v]] for v in range(n_v)]
            return new, rows, cols, rn, cn
    return None


# ---------------------------------------------------------------------------------------------
# job lists
# ---------------------------------------------------------------------------------------------
def k_sweep_jobs(data, k_min: int = 3, k_max: int = 8, phi=None, xi=None, psi=None, n_iters=None, seed: int = 0,
                 row_names=None, col_names=None) -> List[Job]:
    """The factorisations of the k sweep, ``R/main.r:270-290`` (one per k in ``k_min:k_max``; the
    reference then scores them with the bisilhouette and may extend the range, ``:291-312``)."""
    return [Job(list(data), k, phi, xi, psi, n_iters, seed + k, row_names, col_names, tag=f"k={k}")
            for k in range(k_min, k_max + 1)]


def shuffled_jobs(data, n_clusts: int, num_repeats: int = 5, seed: int = 0, n_iters=None) -> List[Job]:
    """``obtain_shuffled_f`` (``R/obtain_bicl.r:31-42``): ``num_repeats`` factorisations of
    independently shuffled copies of every view -- no restrictions, fresh names
    (``temp_row_*``, ``:19-20``), so the views are uncoupled."""
    rng = np.random.default_rng(seed)
    return [Job([shuffle_view(x, rng) for x in data], n_clusts, None, None, None, n_iters, seed + 1000 + r,
                tag=f"shuffle={r}") for r in range(num_repeats)]


def stability_jobs(data, k: int, n_stability: int = 5, sample_rate: float = 0.9, phi=None, xi=None, psi=None,
                   n_iters=None, seed: int = 0, row_names=None, col_names=None) -> List[Job]:
    """The factorisations of ``stability_check`` (``R/stability_analysis.r:305-323``): one per
    repeat on a sub-sample drawn by ``subsample_views``; the draws are kept in ``extras`` for the
    relevance computation on the R side.  Repeats whose sampling fails are skipped, as the reference
    does (``stability_performed = FALSE``)."""
    from . import naming
    rng = np.random.default_rng(seed)
    data = [np.asarray(d, dtype=np.float64) for d in data]
    if row_names is None or col_names is None:
        rn, cn = naming.give_names(data, phi, psi, row_names, col_names)
        row_names = row_names or rn
        col_names = col_names or cn
    jobs = []
    for r in range(n_stability):
        s = subsample_views(data, sample_rate, rng, row_names, col_names)
        if s is None:
            continue
        new, rows, cols, rn, cn = s
        jobs.append(Job(new, k, phi, xi, psi, n_iters, seed + 2000 + r, rn, cn, tag=f"stability={r}",
                        extras={"row_samples": rows, "col_samples": cols}))
    return jobs


# ---------------------------------------------------------------------------------------------
# execution
# ---------------------------------------------------------------------------------------------
def run_job(job: Job, device_id: int = 0, pre_processed: bool = False, return_init: bool = False) -> dict:
    """One factorisation through the accelerated path: naming, symmetrisation and (unless
    ``pre_processed``) non-negativity shift + normalisation as ``apply_resnmtf`` does, device-side SVD
    initialisation, the loop, finalise."""
    from . import api, naming
    data = [np.asarray(d, dtype=np.float64) for d in job.data]
    n_v = len(data)
    rn, cn = naming.give_names(data, job.phi, job.psi, job.row_names, job.col_names)
    row_idx, col_idx = naming.shared_names(rn), naming.shared_names(cn)
    phi = naming.init_rest_mats(job.phi, n_v); psi = naming.init_rest_mats(job.psi, n_v); xi = naming.init_rest_mats(job.xi, n_v)
    if not pre_processed:
        data = naming.check_data(data)
    res = api.res_nmtf_inner(data, row_idx, col_idx, None, None, None, [job.k_val] * n_v, phi, xi, psi,
                             job.n_iters, spurious=False, row_names=rn, col_names=cn, device_id=device_id,
                             seed=job.seed, return_init=return_init)
    res["tag"] = job.tag
    res["extras"] = job.extras
    return res


def run_jobs(jobs: Sequence[Job], device_id: int = 0, group=None, runner: Optional[Callable] = None,
             pre_processed: bool = False) -> List[dict]:
    """Run independent jobs; with an initialised ``torch.distributed`` group of W ranks, rank r runs
    jobs r, r + W, ... on its GPU and every rank returns the complete list in job order.  ``runner``
    replaces ``run_job`` (the CPU tests inject a stand-in)."""
    runner = runner or (lambda job: run_job(job, device_id=device_id, pre_processed=pre_processed))
    rank, world, dist = 0, 1, None
    try:
        import torch.distributed as dist_mod
        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    except ImportError:
        pass
    mine = {i: runner(jobs[i]) for i in range(rank, len(jobs), world)}
    if world == 1:
        return [mine[i] for i in range(len(jobs))]
    gathered: List[Optional[dict]] = [None] * world
    dist.all_gather_object(gathered, mine, group=group)
    merged: Dict[int, dict] = {}
    for part in gathered:
        merged.update(part)
    return [merged[i] for i in range(len(jobs))]


# ---------------------------------------------------------------------------------------------
# the same job kinds with the data resident on the device: one upload, copies / shuffles drawn there
# ---------------------------------------------------------------------------------------------
class DeviceData:
    """The pre-processed views of one data set, uploaded once (``resnmtf_set_view_raw``) and kept for
    any number of factorisations (``resnmtf_copy_view`` / ``resnmtf_shuffle_view``).  At c2 size an
    upload costs ~45 ms of PCIe + conversion, 500 sweeps ~23 ms: re-uploading per job would dominate."""

    def __init__(self, data, phi=None, xi=None, psi=None, row_names=None, col_names=None, device_id: int = 0):
        from . import naming
        from .engine import Engine
        self.data_shapes = [np.asarray(d).shape for d in data]
        n_v = len(data)
        self.rn, self.cn = naming.give_names([np.asarray(d) for d in data], phi, psi, row_names, col_names)
        self.phi = naming.init_rest_mats(phi, n_v); self.xi = naming.init_rest_mats(xi, n_v); self.psi = naming.init_rest_mats(psi, n_v)
        self.device_id = device_id
        self.base = Engine([s[0] for s in self.data_shapes], [s[1] for s in self.data_shapes], [2] * n_v, device_id=device_id)
        self.was_negative = [self.base.set_view_raw(v, np.asarray(data[v], dtype=np.float64)) for v in range(n_v)]

    def close(self):
        self.base.close()

    def _trim_samples(self, samples, max_rounds: int = 20):
        """``sample_view`` / ``stability_repeat`` (``R/stability_analysis.r:165-190``, ``:233-240``): all-zero rows and
        columns of a sub-sample are dropped -- from every earlier view that shares the draw (equal extent along that
        axis) -- and the sub-samples gathered again, until none is left.  The emptiness test runs on the device
        (``resnmtf_view_empty_lines``), on probes that hold only the data.  Returns the trimmed draws or ``None``."""
        from .engine import Engine
        n_v = len(self.data_shapes)
        rows = [np.asarray(r).copy() for r in samples[0]]; cols = [np.asarray(c).copy() for c in samples[1]]
        for _ in range(max_rounds):
            changed = False
            for i in range(n_v):
                if len(rows[i]) < 2 or len(cols[i]) < 2:
                    return None
                probe = Engine([len(rows[i])], [len(cols[i])], [2], device_id=self.device_id)
                try:
                    probe.subsample_view_from(0, self.base, i, rows[i], cols[i])
                    er, ec = probe.empty_lines(0)
                finally:
                    probe.close()
                if er.any() or ec.any():
                    changed = True
                    same_r = self.data_shapes[i][0] == self.data_shapes[0][0]
                    same_c = self.data_shapes[i][1] == self.data_shapes[0][1]
                    for p in (range(i + 1) if same_r else [i]):            # :168-174
                        if len(rows[p]) == len(er):
                            rows[p] = rows[p][~er]
                    for p in (range(i + 1) if same_c else [i]):            # :175-181
                        if len(cols[p]) == len(ec):
                            cols[p] = cols[p][~ec]
            if not changed:
                return rows, cols
        return None

    def factorise(self, k: int, n_iters: Optional[int] = None, seed: int = 0, shuffle_seed: Optional[int] = None,
                  max_iters: int = 100000, tag: str = "", samples=None, return_init: bool = False,
                  return_data: bool = False) -> dict:
        """One factorisation with k biclusters per view: views copied -- or, with ``shuffle_seed``, shuffled as
        ``obtain_shuffled_f`` does (no restrictions, fresh names; redrawn while a row or a column of the shuffled
        matrix sums to zero, ``R/obtain_bicl.r:14-18``), or, with ``samples = (row_samples, col_samples)``,
        sub-sampled as ``stability_repeat`` does (not re-normalised, names carried over; all-zero rows / columns
        dropped first, ``_trim_samples``) -- on the device, device SVD init, loop, finalise.
        ``return_init`` / ``return_data`` add the initial (F, S, G, lambda, mu) per view and the device's copy of the
        data actually factorised (fp32 precision) to the result: what a reference run needs to start from the same place."""
        from . import naming
        from .engine import Engine
        n_v = len(self.data_shapes)
        if samples is not None:
            samples = self._trim_samples(samples)
            if samples is None:
                return {"stability_performed": False, "tag": tag}          # R/stability_analysis.r:223-226
        shapes = self.data_shapes if samples is None else [(len(samples[0][v]), len(samples[1][v])) for v in range(n_v)]
        rn, cn = self.rn, self.cn
        if samples is not None:
            rn = [[self.rn[v][t] for t in samples[0][v]] for v in range(n_v)]
            cn = [[self.cn[v][t] for t in samples[1][v]] for v in range(n_v)]
        eng = Engine([s[0] for s in shapes], [s[1] for s in shapes], [k] * n_v, device_id=self.device_id)
        try:
            shuffled = shuffle_seed is not None
            for v in range(n_v):
                if shuffled:
                    for attempt in range(64):                              # R/obtain_bicl.r:14-18: redraw on an empty row / column
                        eng.shuffle_view_from(v, self.base, v, seed=(shuffle_seed + 7919 * attempt) * 1000003 + v)
                        er, ec = eng.empty_lines(v)
                        if not (er.any() or ec.any()):
                            break
                    else:
                        raise RuntimeError("shuffle_view: every draw left an all-zero row or column")
                elif samples is not None:
                    eng.subsample_view_from(v, self.base, v, samples[0][v], samples[1][v])
                else:
                    eng.copy_view_from(v, self.base, v)
                eng.init_svd(v, seed=seed + v)
            if shuffled:
                eng.set_restrictions(None, None, None)          # R/obtain_bicl.r:35-39: apply_resnmtf without phi/xi/psi
            else:
                eng.set_restrictions(self.phi, self.xi, self.psi)
                rs, cs = naming.shared_names(rn), naming.shared_names(cn)
                for v in range(n_v):
                    for w in range(n_v):
                        if v != w:
                            eng.set_shared_rows(v, w, *naming.index_pairs(rn[v], rn[w], rs[v].get(w)))
                            eng.set_shared_cols(v, w, *naming.index_pairs(cn[v], cn[w], cs[v].get(w)))
            init_state = [eng.get_factors(v) for v in range(n_v)] if return_init else None
            data_used = [eng.get_view(v) for v in range(n_v)] if return_data else None
            errs = eng.run(n_iters=n_iters, tol=1.0e-6, max_iters=max_iters)
            fin = [eng.finalise(v) for v in range(n_v)]
        finally:
            eng.close()
        error = float(np.mean(errs[-10:])) if n_iters is None else float(errs[-1])           # R/main.r:126-130
        res = {"output_f": [f[0] for f in fin], "output_s": [f[1] for f in fin], "output_g": [f[2] for f in fin],
               "row_clusters": [f[3] for f in fin], "col_clusters": [f[4] for f in fin],
               "Error": error, "All_Error": errs, "tag": tag,
               "extras": {} if samples is None else {"row_samples": samples[0], "col_samples": samples[1]},
               "row_names": rn, "col_names": cn}
        if return_init:
            res["init"] = init_state
        if return_data:
            res["data"] = data_used
        return res


def k_sweep_on_device(dev: DeviceData, k_min: int = 3, k_max: int = 8, n_iters=None, seed: int = 0, group=None) -> List[dict]:
    """The factorisations of the k sweep (``R/main.r:279-290``) from one upload; sharded round-robin over
    the ranks of an initialised process group (every rank holds its own ``DeviceData``)."""
    ks = list(range(k_min, k_max + 1))
    return run_jobs(ks, group=group, runner=lambda k: dev.factorise(k, n_iters, seed + k, tag=f"k={k}"))


def shuffles_on_device(dev: DeviceData, n_clusts: int, num_repeats: int = 5, n_iters=None, seed: int = 0, group=None) -> List[dict]:
    """``obtain_shuffled_f`` (``R/obtain_bicl.r:31-42``) with the shuffles drawn on the device."""
    reps = list(range(num_repeats))
    return run_jobs(reps, group=group,
                    runner=lambda r: dev.factorise(n_clusts, n_iters, seed + 1000 + r, shuffle_seed=seed * 7919 + r + 1, tag=f"shuffle={r}"))


def stability_on_device(dev: DeviceData, k: int, n_stability: int = 5, sample_rate: float = 0.9, n_iters=None, seed: int = 0,
                        group=None) -> List[dict]:
    """The factorisations of ``stability_check`` (``R/stability_analysis.r:305-323``): the draws follow
    ``subsample_views`` (shared draws for equal extents), the sub-samples are gathered on the device; all-zero rows /
    columns of a sub-sample -- the pre-processed data are non-negative, not positive: ``make_non_neg`` leaves a zero
    at every shifted column's minimum and sparse inputs stay sparse -- are dropped as the reference does
    (``DeviceData._trim_samples``); a repeat whose sampling fails returns ``stability_performed = False``."""
    rng = np.random.default_rng(seed)
    draws = []
    for _ in range(n_stability):
        rows, cols = [], []
        for v, (n, m) in enumerate(dev.data_shapes):
            same_r = v > 0 and n == dev.data_shapes[0][0]
            same_c = v > 0 and m == dev.data_shapes[0][1]
            rows.append(rows[0] if same_r else rng.choice(n, int(n * sample_rate), replace=False))      # :114-118, :230
            cols.append(cols[0] if same_c else rng.choice(m, int(m * sample_rate), replace=False))      # :119-123, :231
        draws.append((rows, cols))
    return run_jobs(list(range(n_stability)), group=group,
                    runner=lambda r: dev.factorise(k, n_iters, seed + 2000 + r, samples=draws[r], tag=f"stability={r}"))
