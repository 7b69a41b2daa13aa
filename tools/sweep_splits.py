#!/usr/bin/env python3
"""Diagnostic: sweep rate and event-timed passes of one view against forced row-split counts of the two passes
(resnmtf_options.pass_splits_xg / pass_splits_xtf; 0 = the launch model's own choice):
    python tools/sweep_splits.py 20000 4000 32 xtf 0 10 12 14 15 16"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resnmtf_amd import synth  # noqa: E402
from resnmtf_amd.engine import Engine  # noqa: E402

n, m, k = (int(x) for x in sys.argv[1:4])
which = sys.argv[4]
prob = synth.make_problem([(n, m)], k)
for ns in (int(x) for x in sys.argv[5:]):
    opts = {"pass_splits_xg": ns} if which == "xg" else {"pass_splits_xtf": ns}
    def mk(**extra):
        e = Engine([n], [m], [k], **opts, **extra)
        e.set_view(0, prob.data[0]); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0]); e.set_restrictions()
        return e
    e = mk(); e.run(5)
    dt = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); e.run(100); dt = min(dt, time.perf_counter() - t0)
    e.close()
    e = mk(time_kernels=True); e.run(3); e.pass_timings(reset=True); e.run(20); t = e.pass_timings(); e.close()
    xg = t["xg_ms_total"] / t["xg_launches"] * 1e3; xtf = t["xtf_ms_total"] / t["xtf_launches"] * 1e3
    print(f"{which} splits {ns:2d}: {dt / 100 * 1e6:8.1f} us/sweep, X.G {xg:7.1f} us, Xt.F {xtf:7.1f} us", flush=True)
