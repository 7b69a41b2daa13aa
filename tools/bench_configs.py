#!/usr/bin/env python3
"""Sweep rate and streaming-pass bandwidth of the other BASELINE shapes on ONE GPU (all views
local).  Shapes follow BASELINE.json configs; c4/c5 can be scaled down with --scale."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from resnmtf_amd import naming, synth, _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.environ["LIB"]
from resnmtf_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=["c2", "c3", "c4v1", "c5v1"])
ap.add_argument("--sweeps", type=int, default=100)
ap.add_argument("--bf16-split", type=int, default=0)
ap.add_argument("--update-blocks", type=int, default=0)
a = ap.parse_args()
SHAPES = {
    "c2": ([(10000, 2000)], 16, {}),
    "c3": ([(10000, 2000), (10000, 1500)], 16, dict(phi=200.0)),
    "c4v1": ([(20000, 4000)], 32, {}),                                   # one view of c4
    "c4": ([(20000, 4000)] * 4, 32, dict(phi=200.0, psi=200.0)),
    "c5v1": ([(50000, 8000)], 64, {}),                                   # one view of c5
    "c5v2": ([(50000, 8000)] * 2, 64, dict(phi=200.0, psi=200.0, xi=200.0)),
}
for name in a.configs:
    shapes, k, kw = SHAPES[name]
    t0 = time.perf_counter()
    prob = synth.make_problem(shapes, k, **kw)
    gen = time.perf_counter() - t0
    V = len(shapes)
    def mk(**extra):
        if os.environ.get("RESNMTF_XCD_ORDER") == "1":         # A/B: XCD-aware order of the wide pass's main workgroups (opt-in)
            extra.setdefault("xcd_order", True)
        e = Engine([s[0] for s in shapes], [s[1] for s in shapes], [k] * V, bf16_split=a.bf16_split, update_blocks=a.update_blocks, **extra)
        for v in range(V):
            e.set_view(v, prob.data[v]); e.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
        e.set_restrictions(prob.phi, prob.xi, prob.psi)
        rs, cs = naming.shared_names(prob.row_names), naming.shared_names(prob.col_names)
        for v in range(V):
            for w in range(V):
                if v != w:
                    e.set_shared_rows(v, w, *naming.index_pairs(prob.row_names[v], prob.row_names[w], rs[v].get(w)))
                    e.set_shared_cols(v, w, *naming.index_pairs(prob.col_names[v], prob.col_names[w], cs[v].get(w)))
        return e
    e = mk(); e.run(5)
    dt = 1e9
    for _ in range(3):                 # best of three (box-to-box and run-to-run noise is a few %)
        t0 = time.perf_counter(); errs = e.run(a.sweeps); dt = min(dt, time.perf_counter() - t0)
    e.close()
    e = mk(time_kernels=True); e.run(3); e.pass_timings(reset=True); e.run(20); t = e.pass_timings(); e.close()
    xg = t["xg_ms_total"] / t["xg_launches"] * 1e3; xtf = t["xtf_ms_total"] / t["xtf_launches"] * 1e3
    n, m = shapes[0]
    print(f"{name:5s} V={V} {n}x{m} k={k}: {a.sweeps*V/dt:9.1f} view-updates/s ({dt/a.sweeps*1e6:9.1f} us/sweep), "
          f"pass X.G {xg:8.1f} us = {t['xg_bytes']/xg/1e3:6.0f} GB/s, Xt.F {xtf:8.1f} us = {t['xtf_bytes']/xtf/1e3:6.0f} GB/s, "
          f"err {errs[-1]:.4g} (gen {gen:.0f}s)", flush=True)
