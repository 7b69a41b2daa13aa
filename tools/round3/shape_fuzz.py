#!/usr/bin/env python3
"""Random single-view shapes and k against the oracle (default options): planner / padding edge cases."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from resnmtf_amd import synth
from resnmtf_amd.engine import Engine
from helpers import rel_fro, run_oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
bad = 0
cases = [(k, k, k) for k in (1, 2, 16, 17, 32, 33, 48, 49, 64)] + [(64, 65, 64), (65, 64, 64), (1000, 64, 64), (64, 1000, 64), (33, 1500, 17), (1500, 33, 17)]
for _ in range(40):
    k = int(rng.integers(1, 65)); n = int(rng.integers(k, 900)); m = int(rng.integers(k, 900))
    cases.append((n, m, k))
for n, m, k in cases:
    prob = synth.make_problem([(n, m)], k, seed_base=int(rng.integers(0, 1000)))
    ref = run_oracle(prob, n_iters=4)
    e = Engine([n], [m], [k])
    try:
        e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
        errs = e.run(4); f, s, g, rc, cc = e.finalise(0)
    finally:
        e.close()
    df, dg, de = rel_fro(f, ref["output_f"][0]), rel_fro(g, ref["output_g"][0]), float(np.max(np.abs(errs - ref["All_Error"])))
    flag = "" if (df < 1e-4 and dg < 1e-4 and de < 5e-5) else "   <-- WRONG"
    bad += bool(flag)
    print(f"n={n:4d} m={m:4d} k={k:2d}: dF {df:.1e} dG {dg:.1e} dErr {de:.1e}{flag}", flush=True)
print("bad:", bad)
