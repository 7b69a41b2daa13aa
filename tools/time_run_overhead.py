"""Host-side cost of one resnmtf_run call at c2: wall time of run(n) against n, and the per-stage
trace of the library (RESNMTF_TRACE_RUN=1).  The slope is the sweep time, the intercept the per-call overhead."""
import os
import sys
import time

import numpy as np
import torch  # (before the library initialises HIP)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RESNMTF_TRACE_RUN", "1")
from resnmtf_amd import synth  # noqa: E402
from resnmtf_amd.engine import Engine  # noqa: E402

prob = synth.config("c2")
n, m = prob.data[0].shape
for ce in (32, 20):
    e = Engine([n], [m], [prob.k], check_every=ce)
    e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
    e.run(50)
    print(f"check_every = {ce}", flush=True)
    rows = []
    for cnt in (1, 2, 4, 8, 16, 20, 32, 64, 128, 500):
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            e.run(cnt)
            best = min(best, time.perf_counter() - t0)
        rows.append((cnt, best * 1e6))
        print(f"  run({cnt}): {best * 1e6:9.1f} us  = {best * 1e6 / cnt:7.2f} us per sweep", flush=True)
    cnts = np.array([r[0] for r in rows], float); ts = np.array([r[1] for r in rows])
    slope, icpt = np.polyfit(cnts, ts, 1)
    print(f"  fit: {slope:.2f} us per sweep + {icpt:.1f} us per call")
    e.close()

# the driver's sequence (bench.py --steps 20 --warmup 5): first use of the 16-sweep rung inside the timed call?
for trial in range(3):
    e = Engine([n], [m], [prob.k])
    e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
    e.run(5)
    torch.cuda.synchronize()
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); e.run(20); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    print("fresh engine, run(5) then run(20) x4 [us]:", " ".join(f"{t:.1f}" for t in ts), flush=True)
    e.close()
