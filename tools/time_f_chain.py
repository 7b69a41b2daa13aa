#!/usr/bin/env python3
"""Time of RESNMTF_PHASE_F_ALL on one GPU: the fused F chain (f_chain_kernel) against one launch per view.

    python tools/time_f_chain.py [--views 8] [--rows 10000] [--cols 2000] [--k 16] [--reps 400]

One rank's share of a view-sharded run: view 0 is owned, the other views are F replicas whose exchange
blocks are copies of view 0's (what an all-gather would have delivered), all rows shared in the same order.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from resnmtf_amd import _lib, sharded  # noqa: E402
from resnmtf_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--rows", type=int, default=10000)
ap.add_argument("--cols", type=int, default=2000)
ap.add_argument("--k", type=int, default=16)
ap.add_argument("--reps", type=int, default=400)
a = ap.parse_args()

V = a.views
prob = sharded.local_problem(V, (a.rows, a.cols), a.k, phi=200.0, owned=[0])
out = {}
for label, off in (("fused", False), ("per_view", True)):
    st = torch.cuda.Stream()
    eng = Engine([a.rows] * V, [a.cols] * V, [a.k] * V, owned=[v == 0 for v in range(V)], stream=st.cuda_stream,
                 replicate_f=True, no_f_chain=off)
    eng.set_view(0, prob.data[0])
    for v in range(V):
        eng.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
    eng.set_restrictions(prob.phi, prob.xi, prob.psi)
    idx = np.arange(a.rows, dtype=np.int32)
    for v in range(V):
        for w in range(V):
            if v != w:
                eng.set_shared_rows(v, w, idx, idx)
                eng.set_shared_cols(v, w, None, None)
    eng.reserve_sweeps(1024)
    eng.prepare()
    eng.synchronize()
    ad = sharded.HipEngineAdapter(eng)
    blk0 = ad.factor_tensor(0, "FBLOCK")
    for v in range(1, V):
        ad.factor_tensor(v, "FBLOCK").copy_(blk0)
    torch.cuda.synchronize()
    for _ in range(20):
        eng.phase(0, _lib.PHASE_F_ALL, 0)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        eng.phase(0, _lib.PHASE_F_ALL, 0)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    out[label] = [ad.factor_tensor(v, "F").cpu().numpy().copy() for v in range(V)]
    print(f"{label:9s}: {dt * 1e6:8.2f} us per PHASE_F_ALL ({V} views {a.rows} x {a.k})", flush=True)
    ad._views.clear()
    eng.close()
worst = max(float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-300)) for x, y in zip(out["fused"], out["per_view"]))
print("max F per view (fused):", [float(x.max()) for x in out["fused"]][:3])
print(f"fused vs per-view after {a.reps + 20} chained updates: worst max-abs difference / max = {worst:.3e}")
