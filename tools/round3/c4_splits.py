#!/usr/bin/env python3
"""c4-sized view: the launch model's row splits against forced combinations (pass_splits_xg, pass_splits_xtf), best of 3 x 100 sweeps, twice."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from resnmtf_amd import synth  # noqa: E402
from resnmtf_amd.engine import Engine  # noqa: E402
n, m, k = 20000, 4000, 32
prob = synth.make_problem([(n, m)], k)
for rep in range(2):
    for sx, st in ((0, 0), (3, 0), (0, 14), (3, 14), (3, 13), (3, 15), (6, 14)):
        e = Engine([n], [m], [k], pass_splits_xg=sx, pass_splits_xtf=st)
        e.set_view(0, prob.data[0]); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0]); e.set_restrictions()
        e.run(5)
        dt = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); e.run(100); dt = min(dt, time.perf_counter() - t0)
        e.close()
        print(f"xg {sx} xtf {st:2d}: {dt / 100 * 1e6:7.1f} us/sweep", flush=True)
