#!/usr/bin/env python3
"""CPU study (oracle only, no GPU): how far do F / G move when X is stored in a 2-byte format?
fp16 (11-bit mantissa per entry) against uniform 16-bit quantisation (one power-of-two step per view: absolute error
2^-16 of the largest entry everywhere).  Same eight seeded problems as tools/half_parity.py.
    python tools/quant_study.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import rel_fro, run_oracle
from resnmtf_amd import synth
from resnmtf_amd.synth import Problem


def q_fp16(x):
    s = 2.0 ** (14 - np.frexp(x.max())[1])
    return (x * s).astype(np.float16).astype(np.float64) / s


def q_u16(x):
    step = 2.0 ** (np.frexp(x.max())[1] - 16)          # max / step < 2^16
    return np.round(x / step) * step


cases = [([(100, 50)], 3, {}, 60), ([(300, 200)], 5, {}, 200), ([(1000, 333)], 16, {}, 100),
         ([(600, 200), (600, 150)], 16, {"phi": 200.0}, 100), ([(400, 300)] * 4, 8, {"phi": 2.0, "psi": 1.0}, 60),
         ([(320, 256)] * 3, 6, {"phi": 1.0, "psi": 1.0, "xi": 0.3}, 60), ([(2000, 700)], 16, {}, 300), ([(3000, 1000)], 12, {}, 500)]
for shapes, k, kw, iters in cases:
    prob = synth.make_problem(shapes, k, **kw)
    ref = run_oracle(prob, n_iters=iters)
    line = f"{str(shapes):38s} k={k:2d} it={iters:3d}"
    for label, q in (("fp16", q_fp16), ("u16", q_u16)):
        p2 = Problem([q(x) for x in prob.data], prob.init_f, prob.init_s, prob.init_g, prob.phi, prob.xi, prob.psi, prob.k,
                     row_names=prob.row_names, col_names=prob.col_names)
        res = run_oracle(p2, n_iters=iters)
        ef = max(rel_fro(res["output_f"][v], ref["output_f"][v]) for v in range(len(shapes)))
        eg = max(rel_fro(res["output_g"][v], ref["output_g"][v]) for v in range(len(shapes)))
        line += f" | {label}: F {ef:.1e} G {eg:.1e}"
    print(line, flush=True)

# ---- wider value distributions (single view 600 x 300, k = 8, 150 sweeps): what an upload-time guard has to catch
print("\nvalue distributions (u16 image): rel. quantisation error of X, F / G distance")
rng = np.random.default_rng(5)
n, m, k = 600, 300, 8
base = synth.planted_view(n, m, k, 77, normalise=False)
variants = {
    "planted blocks (the bench data)": base,
    "blocks + 20 outliers x100": base + 100.0 * base.max() * (rng.random((n, m)) < 20.0 / (n * m)),
    "blocks + 1 outlier x10000": base + 1e4 * base.max() * (np.arange(n * m).reshape(n, m) == 12345),
    "lognormal(0, 2), no structure": rng.lognormal(0.0, 2.0, (n, m)),
    "uniform(0, 1), no structure": rng.random((n, m)),
    "90 % zeros, blocks": base * (rng.random((n, m)) < 0.1),
    "columns scaled 1 .. 1e4 (undone by normalisation)": base * np.logspace(0, 4, m)[None, :],
    "rows scaled 1 .. 1e4": base * np.logspace(0, 4, n)[:, None],
}
f0, s0, g0 = synth.random_init(n, m, k, 99)
z = np.zeros((1, 1))
for name, raw in variants.items():
    x = raw / raw.sum(axis=0)[None, :]
    xq = q_u16(x)
    rel = np.linalg.norm(xq - x) / np.linalg.norm(x)
    pa = Problem([x], [f0], [s0], [g0], z, z, z, k)
    pb = Problem([xq], [f0], [s0], [g0], z, z, z, k)
    ra, rb = run_oracle(pa, n_iters=150), run_oracle(pb, n_iters=150)
    print(f"{name:52s} |dX|/|X| {rel:.1e}   F {rel_fro(rb['output_f'][0], ra['output_f'][0]):.1e}  G {rel_fro(rb['output_g'][0], ra['output_g'][0]):.1e}",
          flush=True)
