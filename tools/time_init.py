#!/usr/bin/env python3
"""Time resnmtf_init_svd (device, randomized top-k SVD on the pass kernels) against numpy's full
SVD (what the reference's svd() costs on the host) for a config."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from resnmtf_amd import synth
from resnmtf_amd.engine import Engine

for cfg in sys.argv[1:] or ["c2"]:
    if "x" in cfg:
        n_, m_, k_ = (int(t) for t in cfg.split("x")); prob = synth.make_problem([(n_, m_)], k_)
    else:
        prob = synth.config(cfg)
    x = prob.data[0]; n, m = x.shape; k = prob.k
    e = Engine([n], [m], [k]); e.set_view(0, x); e.set_restrictions()
    e.init_svd(0, seed=1)
    t0 = time.perf_counter(); d = e.init_svd(0, seed=1); t_dev = time.perf_counter() - t0
    errs = e.run(50); e.close()
    t0 = time.perf_counter(); dref = np.linalg.svd(x, compute_uv=False)[:k]; t_host_vals = time.perf_counter() - t0
    t0 = time.perf_counter(); np.linalg.svd(x, full_matrices=False); t_host = time.perf_counter() - t0
    print(f"{cfg}: device init {t_dev*1e3:8.1f} ms | host full SVD {t_host*1e3:9.1f} ms (values only {t_host_vals*1e3:.1f} ms) | "
          f"max rel err of d[:k] {np.max(np.abs(d - dref) / dref):.2e} | error after 50 sweeps {errs[-1]:.4g}", flush=True)
