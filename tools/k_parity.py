#!/usr/bin/env python3
"""F / G / S distance from the fp64 oracle at k > 16 for the two arithmetic forms of the streaming passes
(bf16_split 0 = three bf16 pieces on the K = 32 MFMA, wide workgroups; 2 = plain f32 MFMA) on a few seeded problems."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import rel_fro, run_hip, run_oracle
from resnmtf_amd import synth

cases = [([(500, 384)], 64, {}, 30), ([(500, 384)], 64, {}, 100), ([(1000, 700)], 64, {}, 30), ([(400, 320)], 48, {}, 30),
         ([(700, 500)], 24, {}, 30), ([(3000, 1200)], 64, {}, 60), ([(2000, 900)], 32, {}, 60),
         ([(300, 200), (300, 150)], 40, dict(phi=50.0, xi=20.0), 30)]
for seed_base in (0, 7):
    for shapes, k, kw, iters in cases:
        prob = synth.make_problem(shapes, k, seed_base=seed_base, **kw)
        ref = run_oracle(prob, n_iters=iters)
        row = []
        for mode in (0, 2):
            res = run_hip(prob, n_iters=iters, bf16_split=mode)
            f = max(rel_fro(res["output_f"][v], ref["output_f"][v]) for v in range(len(shapes)))
            g = max(rel_fro(res["output_g"][v], ref["output_g"][v]) for v in range(len(shapes)))
            e = float(np.max(np.abs(res["All_Error"] - ref["All_Error"])))
            row.append(f"mode {mode}: F {f:.2e} G {g:.2e} err {e:.1e}")
        print(f"seed+{seed_base} {shapes} k={k} {iters} sweeps | " + " | ".join(row), flush=True)
