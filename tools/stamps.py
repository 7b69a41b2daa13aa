#!/usr/bin/env python3
"""Diagnostic: where does a factor-update workgroup spend its time?  Builds
libresnmtf_hip_stamps.so (-DRESNMTF_STAMPS), runs a few sweeps of c2 eagerly, then ONE F update
(or G update) with the stamp buffer attached and prints per-stage statistics over the blocks.
Stamps are 100 MHz wall-clock ticks (10 ns)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "F"
so = os.path.join(ROOT, "resnmtf_amd", "libresnmtf_hip_stamps.so")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRESNMTF_STAMPS",
                "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "resnmtf_amd", "csrc"), "-o", so,
                os.path.join(ROOT, "resnmtf_amd", "csrc", "resnmtf_hip.hip")], check=True)
from resnmtf_amd import _lib, synth  # noqa: E402
_lib.LIB_PATH = so
from resnmtf_amd.engine import Engine  # noqa: E402
import torch  # noqa: E402  (device memory for the stamp buffer)

lib = _lib.load()
lib.resnmtf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
prob = synth.config("c2")
n, m = prob.data[0].shape
e = Engine([n], [m], [prob.k], use_graph=False)
e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
e.run(5)
e.reserve_sweeps(64); e.prepare()
for sw in range(3):
    e.phase(0, _lib.PHASE_F, sw); e.phase(0, _lib.PHASE_G, sw)
e.synchronize()
nblk = 4096
buf = torch.zeros((nblk, 16), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
which = sys.argv[1] if len(sys.argv) > 1 else "F"
if which == "F":
    assert lib.resnmtf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
    e.phase(0, _lib.PHASE_F, 3); e.synchronize()
else:
    e.phase(0, _lib.PHASE_F, 3); e.synchronize()
    assert lib.resnmtf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
    e.phase(0, _lib.PHASE_G, 3); e.synchronize()
lib.resnmtf_debug_set_stamp_buffer(None)
t = buf.cpu().numpy().astype(np.int64)
t = t[t[:, 0] > 0]
print(f"{which} update: {len(t)} blocks stamped")
t0 = t[:, 0].min()
names = {0: "entry", 1: "operands landed (1st barrier)", 2: "rows computed + stored", 3: "2nd barrier", 4: "partials stored"}
print(f"{'stage':34s} {'min':>8s} {'median':>8s} {'max':>8s}   (us since first block entry)")
for idx in range(5):
    col = t[:, idx]; col = col[col > 0]
    rel = (col - t0) / 100.0
    print(f"{names[idx]:34s} {rel.min():8.2f} {np.median(rel):8.2f} {rel.max():8.2f}")
e.close()
