#!/usr/bin/env python3
"""On-GPU tuning sweep of the streaming-pass geometry (waves per workgroup x row splits x LDS pad)
for one BASELINE config.  Prints per-config pass durations (HIP events) and whole-sweep rate.

    python tools/sweep_pass.py [c2] [--quick]
"""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from resnmtf_amd import synth  # noqa: E402
from resnmtf_amd.engine import Engine  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c2"
prob = synth.config(cfg)
n, m = prob.data[0].shape
k = prob.k


def measure(**kw):
    def mk(**extra):
        e = Engine([n], [m], [k], **kw, **extra)
        e.set_view(0, prob.data[0]); e.set_restrictions()
        e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
        return e
    e = mk()
    e.run(30)
    t0 = time.perf_counter(); e.run(300); dt = time.perf_counter() - t0
    e.close()
    e = mk(time_kernels=True)
    e.run(10); e.pass_timings(reset=True); e.run(100); t = e.pass_timings(); e.close()
    return 300 / dt, t["xg_ms_total"] / t["xg_launches"] * 1e3, t["xtf_ms_total"] / t["xtf_launches"] * 1e3


configs = []
for npp in (True, False):
    for nw, sx, st in ((8, 4, 8), (8, 4, 16), (4, 4, 16), (4, 8, 16), (16, 3, 8)):
        configs.append(dict(pass_waves=nw, pass_splits_xg=sx, pass_splits_xtf=st, no_pitch_pad=npp))
print(f"{'config':70s} {'sweeps/s':>10s} {'xg_us':>8s} {'xtf_us':>8s}", flush=True)
for c in configs:
    try:
        r, xg, xtf = measure(**c)
        print(f"{str(c):70s} {r:10.0f} {xg:8.2f} {xtf:8.2f}", flush=True)
    except Exception as exc:  # keep sweeping
        print(f"{str(c):70s} FAILED {exc}", flush=True)
