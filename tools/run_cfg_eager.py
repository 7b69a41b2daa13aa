#!/usr/bin/env python3
"""A few EAGER sweeps (one dispatch per kernel, no graph) of one bench_configs.py shape: the target of the PMC passes
(rocprofv3 --pmc attributes counters per dispatch).   python tools/run_cfg_eager.py c5v1 --sweeps 6"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resnmtf_amd import synth
from resnmtf_amd.engine import Engine

SHAPES = {"c2": ((10000, 2000), 16), "c4v1": ((20000, 4000), 32), "c5v1": ((50000, 8000), 64)}
ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--sweeps", type=int, default=6)
ap.add_argument("--bf16-split", type=int, default=0)
a = ap.parse_args()
(n, m), k = SHAPES[a.config]
prob = synth.make_problem([(n, m)], k)
e = Engine([n], [m], [k], use_graph=False, bf16_split=a.bf16_split)
e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
errs = e.run(a.sweeps)
e.close()
print(a.config, "eager sweeps", a.sweeps, "final error", errs[-1])
