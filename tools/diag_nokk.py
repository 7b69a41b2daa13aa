#!/usr/bin/env python3
"""Diagnostic: time the sweep with the k x k jobs compiled out of the pass launches
(-DRESNMTF_DIAG_NO_KK; results are meaningless, only kernel durations matter)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "resnmtf_amd", "libresnmtf_hip_nokk.so")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRESNMTF_DIAG_NO_KK",
                "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "resnmtf_amd", "csrc"), "-o", so,
                os.path.join(ROOT, "resnmtf_amd", "csrc", "resnmtf_hip.hip")], check=True)
from resnmtf_amd import _lib, synth
_lib.LIB_PATH = so
from resnmtf_amd.engine import Engine
prob = synth.config("c2")
n, m = prob.data[0].shape
for kw in (dict(pass_waves=8, pass_splits_xg=4, pass_splits_xtf=16), dict(pass_waves=4, pass_splits_xg=8, pass_splits_xtf=32)):
    e = Engine([n], [m], [prob.k], **kw)
    e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
    e.run(30)
    t0 = time.perf_counter(); e.run(300); dt = time.perf_counter() - t0
    print(kw, f"{300/dt:.0f} sweeps/s, {dt/300*1e6:.1f} us/sweep")
    e.close()
