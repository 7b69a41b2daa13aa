"""Writes the golden fixtures of this directory as plain CSV (full fp64 precision, `repr`) under
tests/golden/r_export/<fixture>/ so that R can read them without any package:

    python tests/golden/export_for_r.py

    meta.csv                 n_views, n_iters
    x<v>.csv                 data matrix of view v WITH row / column names (first column = row names, header = column names)
    f0_<v>.csv s0_<v>.csv g0_<v>.csv      initial factors
    phi.csv xi.csv psi.csv                restriction matrices as init_rest_mats returns them (already symmetrised)
    out_f<v>.csv out_s<v>.csv out_g<v>.csv rc<v>.csv cc<v>.csv all_error.csv     expected outputs (the restatement's)

`check_with_r.R` next to this file feeds them to the reference's own exported res_nmtf_inner() -- for whoever has R
(SURVEY.md section 8 c5); nobody could run it where this repository is built."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import GOLDEN_NAMES, load_golden  # noqa: E402


def write_matrix(path, a, row_names=None, col_names=None):
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    with open(path, "w") as fh:
        if col_names is not None:
            fh.write(",".join(([""] if row_names is not None else []) + [f'"{c}"' for c in col_names]) + "\n")
        for i, row in enumerate(a):
            cells = [repr(float(v)) if np.isfinite(v) else ("NaN" if np.isnan(v) else ("Inf" if v > 0 else "-Inf")) for v in row]
            fh.write(",".join(([f'"{row_names[i]}"'] if row_names is not None else []) + cells) + "\n")


def main():
    out_root = os.path.join(HERE, "r_export")
    for name in GOLDEN_NAMES:
        g = load_golden(name)
        d = os.path.join(out_root, name)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "meta.csv"), "w") as fh:
            fh.write("n_views,n_iters\n%d,%d\n" % (g["n_views"], g["n_iters"]))
        for v in range(g["n_views"]):
            write_matrix(os.path.join(d, f"x{v + 1}.csv"), g["x"][v], g["row_names"][v], g["col_names"][v])
            for key, fn in (("f0", "f0_"), ("s0", "s0_"), ("g0", "g0_"), ("out_f", "out_f"), ("out_s", "out_s"), ("out_g", "out_g"),
                            ("rc", "rc"), ("cc", "cc")):
                write_matrix(os.path.join(d, f"{fn}{v + 1}.csv"), g[key][v])
        for key in ("phi", "xi", "psi"):
            write_matrix(os.path.join(d, key + ".csv"), g[key])
        write_matrix(os.path.join(d, "all_error.csv"), np.asarray(g["all_error"])[:, None])
    print("wrote", out_root)


if __name__ == "__main__":
    main()
