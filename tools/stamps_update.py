#!/usr/bin/env python3
"""Diagnostic: timeline of the workgroups of the factor-update kernels (stamps build, see stamps.py).
Runs a few sweeps of a config eagerly, then ONE PHASE_F and ONE PHASE_G with the stamp buffer
attached (the pass launches stamp other columns of other rows; only update-kernel rows are read:
the update kernels run first in each phase and their block ids are the lowest)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
so = os.path.join(ROOT, "resnmtf_amd", "libresnmtf_hip_stamps.so")
if not os.path.exists(so) or os.environ.get("STAMPS_REBUILD") == "1":
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRESNMTF_STAMPS",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "resnmtf_amd", "csrc"), "-o", so,
                    os.path.join(ROOT, "resnmtf_amd", "csrc", "resnmtf_hip.hip")], check=True)
from resnmtf_amd import _lib, synth
_lib.LIB_PATH = so
from resnmtf_amd.engine import Engine
import torch
lib = _lib.load()
lib.resnmtf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
if "x" in cfg:
    n_, m_, k_ = (int(t) for t in cfg.split("x")); prob = synth.make_problem([(n_, m_)], k_)
else:
    prob = synth.config(cfg)
n, m = prob.data[0].shape
e = Engine([n], [m], [prob.k], use_graph=False)
e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
e.run(5); e.reserve_sweeps(64); e.prepare()
for sw in range(3):
    e.phase(0, _lib.PHASE_F, sw); e.phase(0, _lib.PHASE_G, sw)
e.synchronize()
st = lambda x: f"min {x.min():6.2f}  med {np.median(x):6.2f}  p90 {np.percentile(x,90):6.2f}  max {x.max():6.2f}"
for name, phase, base in (("F update", _lib.PHASE_F, 0), ("G update", _lib.PHASE_G, 8)):
    buf = torch.zeros((16384, 16), dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
    if phase == _lib.PHASE_G:
        e.phase(0, _lib.PHASE_F, 3); e.synchronize()
    assert lib.resnmtf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
    assert lib.resnmtf_debug_set_stamp_select(2) == 0      # only the update kernels stamp
    e.phase(0, phase, 3); e.synchronize()
    lib.resnmtf_debug_set_stamp_buffer(None)
    t = buf.cpu().numpy().astype(np.int64)[:, base:base + 8]
    if phase == _lib.PHASE_G:       # the G form stamps columns 8..12; pass kernels of the same phase use 0..7 / 8..15 of
        pass                        # THEIR block rows -- rows with column 4 set and column 5.. unset belong to the update
    rows = t[(t[:, 0] > 0) & (t[:, 3] > 0) & (t[:, 4] > 0)]
    rows = rows[rows[:, 1] >= rows[:, 0]]
    t0 = rows[:, 0].min()
    us = lambda c: (rows[:, c] - t0) / 100.0
    print(f"== {name} ({cfg}): {len(rows)} workgroups")
    for c, label in ((0, "entry"), (1, "operands arrived"), (2, "first group updated"), (3, "all groups done"), (4, "partials written")):
        print(f"  {label:22s}", st(us(c)))
    if phase == _lib.PHASE_F:
        e.phase(0, _lib.PHASE_G, 3); e.synchronize()
e.close()
