#!/usr/bin/env python3
"""Sweep rate in convergence mode (the reference's default: while |d err| > 1e-6, R/main.r:50-81) against the
fixed-iteration mode, c2, 500 sweeps each (tolerance 0 so that the test never fires)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resnmtf_amd import synth
from resnmtf_amd.engine import Engine
prob = synth.config("c2")
n, m = prob.data[0].shape
for ce in (8, 32):
    e = Engine([n], [m], [prob.k], check_every=ce)
    e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
    e.run(50)
    t0 = time.perf_counter(); e.run(500); t_fixed = time.perf_counter() - t0
    e.run(n_iters=None, tol=0.0, max_iters=50)
    t0 = time.perf_counter(); errs = e.run(n_iters=None, tol=0.0, max_iters=500); t_conv = time.perf_counter() - t0
    print(f"check_every={ce}: fixed {t_fixed/500*1e6:.2f} us/sweep, convergence mode {t_conv/len(errs)*1e6:.2f} us/sweep ({len(errs)} sweeps)", flush=True)
    e.close()
