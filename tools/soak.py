#!/usr/bin/env python3
"""Soak check: handles created, run and destroyed in a loop (fixed and convergence mode, one and three views, k = 16 and
k = 40); free device memory before and after must agree, every run must return finite errors."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import psutil  # noqa: E402
import torch  # noqa: E402
from resnmtf_amd import naming, synth  # noqa: E402
from resnmtf_amd.engine import Engine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
probs = [synth.make_problem([(3000, 700)], 16), synth.make_problem([(1200, 500)] * 3, 40, phi=2.0, psi=1.0, xi=0.5)]
free0 = None
t0 = time.perf_counter()
for it in range(rounds + 2):
    if it == 2:                     # (the first handles load the code objects and the runtime's own pools: one-off)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        rss0 = psutil.Process().memory_info().rss
    prob = probs[it % 2]
    V = len(prob.data)
    e = Engine([d.shape[0] for d in prob.data], [d.shape[1] for d in prob.data], [prob.k] * V)
    for v in range(V):
        e.set_view(v, prob.data[v]); e.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
    e.set_restrictions(prob.phi, prob.xi, prob.psi)
    if V > 1:
        rs, cs = naming.shared_names(prob.row_names), naming.shared_names(prob.col_names)
        for v in range(V):
            for w in range(V):
                if v != w:
                    e.set_shared_rows(v, w, *naming.index_pairs(prob.row_names[v], prob.row_names[w], rs[v].get(w)))
                    e.set_shared_cols(v, w, *naming.index_pairs(prob.col_names[v], prob.col_names[w], cs[v].get(w)))
    errs = e.run(37 + it)                                   # a different run length every time (graph cache)
    assert np.isfinite(errs).all()
    errs = e.run(n_iters=None, tol=1e-7, max_iters=300)     # convergence mode
    assert np.isfinite(errs).all() and len(errs) >= 1
    e.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
rss1 = psutil.Process().memory_info().rss
print(f"{rounds} handles in {time.perf_counter() - t0:.1f} s; free device memory {free0 >> 20} MiB -> {free1 >> 20} MiB; "
      f"host RSS {rss0 >> 20} MiB -> {rss1 >> 20} MiB")
assert abs(free0 - free1) < (8 << 20), "device memory not returned"
assert rss1 - rss0 < (rounds * (256 << 10)) + (64 << 20), "host memory grows with the number of handles"
print("soak ok")
