#!/usr/bin/env python3
"""Parity of the 2-byte-image passes (resnmtf_options.x_half: 1 = fp16, 2 = uniform 16-bit) against the fp64 oracle, next to the f32 passes.
    python tools/half_parity.py          (diagnostic; uses the oracle, so it lives outside the product)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import rel_fro, run_hip, run_oracle
from resnmtf_amd import synth

cases = [([(100, 50)], 3, {}, 60), ([(300, 200)], 5, {}, 200), ([(1000, 333)], 16, {}, 100),
         ([(600, 200), (600, 150)], 16, {"phi": 200.0}, 100), ([(400, 300)] * 4, 8, {"phi": 2.0, "psi": 1.0}, 60),
         ([(320, 256)] * 3, 6, {"phi": 1.0, "psi": 1.0, "xi": 0.3}, 60), ([(2000, 700)], 16, {}, 300), ([(3000, 1000)], 12, {}, 500)]
for shapes, k, kw, iters in cases:
    prob = synth.make_problem(shapes, k, **kw)
    ref = run_oracle(prob, n_iters=iters)
    line = f"{str(shapes):38s} k={k:2d} it={iters:3d}"
    for label, opts in (("f32", {}), ("f16", {"x_half": 1}), ("u16", {"x_half": 2})):
        res = run_hip(prob, n_iters=iters, **opts)
        ef = max(rel_fro(res["output_f"][v], ref["output_f"][v]) for v in range(len(shapes)))
        eg = max(rel_fro(res["output_g"][v], ref["output_g"][v]) for v in range(len(shapes)))
        es = max(rel_fro(res["output_s"][v], ref["output_s"][v]) for v in range(len(shapes)))
        ee = float(np.abs(res["All_Error"] - ref["All_Error"]).max())
        mism = sum(int((res["row_clusters"][v] != ref["row_clusters"][v]).sum() + (res["col_clusters"][v] != ref["col_clusters"][v]).sum())
                   for v in range(len(shapes)))
        line += f" | {label}: F {ef:.1e} G {eg:.1e} S {es:.1e} err {ee:.1e} cl {mism}"
    print(line, flush=True)

# ---- guarded mode over more seeds / shapes (support for making it the default): worst distances
print("\nx_half = 3 (guarded 16-bit image) over seeds:")
from resnmtf_amd.engine import Engine
worst = 0.0
for si, (shapes, k, kw, iters) in enumerate([([(500, 260)], 7, {}, 120), ([(1500, 400)], 16, {}, 150), ([(800, 800)], 10, {}, 100),
                                              ([(700, 180), (700, 220)], 9, {"phi": 5.0}, 80), ([(256, 1024)], 4, {}, 150),
                                              ([(640, 320)] * 3, 12, {"phi": 1.0, "xi": 0.5}, 80)]):
    for seed in (11, 23, 37):
        prob = synth.make_problem(shapes, k, seed_base=seed, **kw)
        ref = run_oracle(prob, n_iters=iters)
        res = run_hip(prob, n_iters=iters, x_half=3)
        ef = max(rel_fro(res["output_f"][v], ref["output_f"][v]) for v in range(len(shapes)))
        eg = max(rel_fro(res["output_g"][v], ref["output_g"][v]) for v in range(len(shapes)))
        worst = max(worst, ef, eg)
        print(f"  {str(shapes):34s} k={k:2d} seed {seed}: F {ef:.1e} G {eg:.1e}", flush=True)
print(f"worst F / G distance: {worst:.2e}")
