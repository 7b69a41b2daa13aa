#!/usr/bin/env python3
"""Sweep rate and pass-launch times of a config under different engine options (diagnostic).
    python tools/tune_c2.py [config] "opt=val,opt=val" "opt=val" ...      (one engine per argument)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resnmtf_amd import synth
from resnmtf_amd.engine import Engine

args = sys.argv[1:]
cfg = args.pop(0) if args and "=" not in args[0] and args[0] != "-" else "c2"
if "x" in cfg:                      # custom single view "NxMxK"
    n_, m_, k_ = (int(t) for t in cfg.split("x"))
    prob = synth.make_problem([(n_, m_)], k_)
else:
    prob = synth.config(cfg)
n, m = prob.data[0].shape
for spec in (args or ["-"]):
    kw = {} if spec == "-" else {k: int(v) for k, v in (kv.split("=") for kv in spec.split(","))}
    def mk(**extra):
        e = Engine([n], [m], [prob.k], **kw, **extra)
        e.set_view(0, prob.data[0]); e.set_restrictions(None, None, None)
        e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
        return e
    e = mk(); e.run(50)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); errs = e.run(500); best = min(best, time.perf_counter() - t0)
    e.close()
    e = mk(time_kernels=True); e.run(20); e.pass_timings(reset=True); e.run(100); t = e.pass_timings(); e.close()
    xg = t["xg_ms_total"] / t["xg_launches"] * 1e3; xtf = t["xtf_ms_total"] / t["xtf_launches"] * 1e3
    gb = t["xg_bytes"] / 1e3
    print(f"{cfg} {spec:45s}: {best/500*1e6:7.2f} us/sweep ({500/best:7.0f}/s)  X.G {xg:6.2f} us ({gb/xg:5.0f} GB/s)  Xt.F {xtf:6.2f} us ({gb/xtf:5.0f} GB/s)  err {errs[-1]:.6g}", flush=True)
