#!/usr/bin/env python3
"""Wall time of the repeated factorisations of one apply_resnmtf on a c2-sized view, data resident on the
device (batched.DeviceData): k sweep 3..8, 5 shuffles, 5 sub-samples, 500 sweeps each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from resnmtf_amd import batched, synth

x = [synth.planted_view(10000, 2000, 16, 1000)]
t0 = time.perf_counter(); dev = batched.DeviceData(x); t_up = time.perf_counter() - t0
for name, fn in (("k sweep 3..8", lambda: batched.k_sweep_on_device(dev, 3, 8, n_iters=500)),
                 ("5 shuffles, k = 8", lambda: batched.shuffles_on_device(dev, 8, 5, n_iters=500)),
                 ("5 sub-samples, k = 8", lambda: batched.stability_on_device(dev, 8, 5, n_iters=500))):
    fn()                                            # warm (module load, first graph captures)
    t0 = time.perf_counter(); res = fn(); dt = time.perf_counter() - t0
    errs = ", ".join("%.3g" % r["Error"] for r in res)
    print(f"{name:22s}: {len(res)} factorisations x 500 sweeps in {dt*1e3:7.1f} ms = {dt/len(res)*1e3:6.1f} ms each (errors {errs})", flush=True)
print(f"one upload + pre-processing: {t_up*1e3:.1f} ms")
dev.close()
