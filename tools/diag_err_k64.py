#!/usr/bin/env python3
"""Parity of the k > 32 path on small seeded problems (where rounding averages least): worst
rel-Frobenius distance of F / G / S to the oracle after 30 sweeps.  LIB=path selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # lives in tests/: it calls the oracle
import numpy as np
from resnmtf_amd import _lib, synth
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.environ["LIB"]
from helpers import rel_fro, run_hip, run_oracle
for shapes, k, kw in [([(400, 300)], 48, {}), ([(400, 300)], 64, {}), ([(1000, 700)], 64, {}), ([(300, 200), (300, 150)], 40, dict(phi=50.0)),
                      ([(3000, 1100)], 64, {})]:
    prob = synth.make_problem(shapes, k, **kw)
    ref = run_oracle(prob, n_iters=30); res = run_hip(prob, n_iters=30, bf16_split=int(os.environ.get('SPLIT', '0')))
    worst = max(max(rel_fro(res[key][v], ref[key][v]) for v in range(len(shapes))) for key in ("output_f", "output_g"))
    ws = max(rel_fro(res["output_s"][v], ref["output_s"][v]) for v in range(len(shapes)))
    print(f"{shapes} k={k}: worst F/G {worst:.3e}  S {ws:.3e}  err diff {np.max(np.abs(res['All_Error'] - ref['All_Error'])):.2e}", flush=True)
