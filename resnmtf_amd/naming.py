"""Host-side input hygiene that feeds the C-ABI boundary (NumPy only, no device work).

Mirrors, for the fixed-k path, what ``apply_resnmtf`` does before it reaches the loop
(``R/main.r:225-249``): auto-naming of rows/columns (``give_names``, ``R/utils.r:469-542``),
the shared-name maps (``reorder_data`` / ``produce_indices``, ``R/utils.r:560-662``),
symmetrisation of the restriction matrices (``init_rest_mats``, ``R/update_steps.r:12-24``)
and non-negativity + column normalisation (``check_inputs``, ``R/utils.r:416-422``).

The reference matches shared rows BY NAME inside the loop (``R/utils.r:67-71``); the device
path wants integer index pairs, built here once (``index_pairs``).
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

SharedNames = List[Dict[int, Optional[List[str]]]]


def init_rest_mats(mat, n_v: int) -> np.ndarray:
    """``R/update_steps.r:12-24``: NULL -> zeros, else zero the diagonal and return M + t(M)."""
    if mat is None:
        return np.zeros((n_v, n_v))
    m = np.array(mat, dtype=np.float64, copy=True)
    if m.shape != (n_v, n_v):
        raise ValueError("restriction matrix must be of the same dimensions as data.")   # utils.r:351-353
    if (m < 0).any():
        raise ValueError("restriction matrix must be a non-negative matrix.")             # utils.r:348-350
    np.fill_diagonal(m, 0.0)
    return m + m.T


def check_data(data: Sequence[np.ndarray]) -> List[np.ndarray]:
    """``make_non_neg`` (per-column shift, ``R/utils.r:20-27``, with the reference's warning)
    followed by ``matrix_normalisation`` (``R/utils.r:86-88``)."""
    out = []
    for x in data:
        x = np.asarray(x, dtype=np.float64)
        if x.ndim != 2:
            raise ValueError("Data must be a list of matrices or a matrix.")               # utils.r:317
        colmin = x.min(axis=0)
        if (colmin < 0).any():
            warnings.warn("Matrix is not non-negative. Has been made non-negative.")       # utils.r:24
        x = x + np.abs(np.minimum(0.0, colmin))[None, :]
        out.append(x / x.sum(axis=0)[None, :])
    return out


def give_names(data: Sequence[np.ndarray], phi=None, psi=None,
               row_names: Optional[Sequence[Optional[Sequence[str]]]] = None,
               col_names: Optional[Sequence[Optional[Sequence[str]]]] = None
               ) -> Tuple[List[List[str]], List[List[str]]]:
    """``R/utils.r:469-542``.  ``row_names[v]`` / ``col_names[v]`` may be None for an unnamed
    view.  All views unnamed -> ``row_<n>`` / ``col_<n>`` with one running counter over the
    views; a positive ``phi[i, j]`` (``psi[i, j]``), j > i, copies view i's names onto view j
    and requires equal sizes.  Some-but-not-all views named -> error, as in the reference."""
    n_views = len(data)

    def one_axis(names, sizes, rest, what):
        names = [None] * n_views if names is None else [None if nm is None else list(nm) for nm in names]
        missing = [nm is None for nm in names]
        if all(missing):
            n = 1
            for i in range(n_views):
                if names[i] is None:
                    names[i] = [f"{what}_{t}" for t in range(n, n + sizes[i])]
                    n += sizes[i]
                if rest is not None:
                    for j in range(min(i + 1, n_views - 1), n_views):
                        if rest[i][j] > 0 and sizes[i] != sizes[j]:
                            raise ValueError(
                                f"{what.capitalize()} restriction matrices implies shared {what}s between views "
                                f"with differing number of unnamed {what}s. Please name {what}s.")
                        elif rest[i][j] > 0:
                            names[j] = list(names[i])
        elif any(missing):
            raise ValueError(f"At least one view is missing {what} names. Please name missing {what}s.")
        else:
            for nm, sz in zip(names, sizes):
                if len(nm) != sz or any(x is None for x in nm):
                    raise ValueError(f"Some {what}s missing names. Check {what} names.")
        return names

    rn = one_axis(row_names, [np.shape(d)[0] for d in data], None if phi is None else np.asarray(phi), "row")
    cn = one_axis(col_names, [np.shape(d)[1] for d in data], None if psi is None else np.asarray(psi), "col")
    return rn, cn


def shared_names(names: Sequence[Sequence[str]]) -> SharedNames:
    """``reorder_data`` + ``produce_indices`` for one axis (``R/utils.r:560-662``).

    The reference partitions the names over the power set of views and, for a pair (v, w),
    concatenates the parts whose view subset contains both: exactly the names present in
    both views.  Empty -> None (the reference's NA)."""
    n_views = len(names)
    sets = [set(nm) for nm in names]
    out: SharedNames = []
    for v in range(n_views):
        d: Dict[int, Optional[List[str]]] = {}
        for w in range(n_views):
            if w == v:
                continue
            common = [nm for nm in dict.fromkeys(names[v]) if nm in sets[w]]
            d[w] = common if common else None
        out.append(d)
    return out


def index_pairs(names_v: Sequence[str], names_w: Sequence[str], shared: Optional[Sequence[str]]):
    """Integer form of ``masked[rows, ] <- mat_list[[i]][rows, ]`` (``R/utils.r:71``): for each
    shared name its first position in view v and in view w (R's character indexing takes the
    first match).  Returns (None, None) for NA."""
    if shared is None:
        return None, None
    pos_v: Dict[str, int] = {}
    for p, nm in enumerate(names_v):
        pos_v.setdefault(nm, p)
    pos_w: Dict[str, int] = {}
    for p, nm in enumerate(names_w):
        pos_w.setdefault(nm, p)
    try:
        iv = np.fromiter((pos_v[s] for s in shared), dtype=np.int32, count=len(shared))
        iw = np.fromiter((pos_w[s] for s in shared), dtype=np.int32, count=len(shared))
    except KeyError as exc:      # R: "subscript out of bounds"
        raise KeyError(f"shared name {exc.args[0]!r} is not a name of both views") from None
    return iv, iw
