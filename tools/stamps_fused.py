#!/usr/bin/env python3
"""Diagnostic: timeline of a FUSED pass launch (pass_fused_kernel: update blocks in the first workgroups, the rest wait on
the arrival flag).  -DRESNMTF_STAMPS build; stamps are 100 MHz ticks.  Columns (X.G launch 0.., Xt.F launch 8..):
+0 entry, +5 update block done, +6 wait over, +1 pass body done, +4 k x k job done.
    python tools/stamps_fused.py [fuse_updates: 1 | 2]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
so = os.path.join(ROOT, "resnmtf_amd", "libresnmtf_hip_stamps.so")
if not os.path.exists(so) or os.environ.get("STAMPS_REBUILD") == "1":
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRESNMTF_STAMPS",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "resnmtf_amd", "csrc"), "-o", so,
                    os.path.join(ROOT, "resnmtf_amd", "csrc", "resnmtf_hip.hip")], check=True)
from resnmtf_amd import _lib, synth  # noqa: E402
_lib.LIB_PATH = so
from resnmtf_amd.engine import Engine  # noqa: E402
import torch  # noqa: E402

lib = _lib.load()
lib.resnmtf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
prob = synth.config("c2")
n, m = prob.data[0].shape
e = Engine([n], [m], [prob.k], use_graph=False, fuse_updates=mode)
e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
e.run(6)
buf = torch.zeros((16384, 16), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
assert lib.resnmtf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
assert lib.resnmtf_debug_set_stamp_select(1) == 0
e.run(1)                                   # one eager sweep: fused Xt.F launch, fused X.G launch
lib.resnmtf_debug_set_stamp_buffer(None)
tall = buf.cpu().numpy().astype(np.int64)
st = lambda x: f"min {x.min():6.2f}  p10 {np.percentile(x,10):6.2f}  med {np.median(x):6.2f}  p90 {np.percentile(x,90):6.2f}  max {x.max():6.2f}" if len(x) else "-"
for name, base in (("Xt.F", 8), ("X.G", 0)):
    t = tall[:, base:base + 8]
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = lambda col: (col[col > 0] - t0) / 100.0
    print(f"== fused {name} launch (c2, mode {mode}): {len(t)} workgroups stamped")
    print("  entry        ", st(us(t[:, 0])))
    print("  update done  ", st(us(t[:, 5])), f"({int((t[:, 5] > 0).sum())} blocks)")
    print("  wait over    ", st(us(t[:, 6])))
    print("  pass done    ", st(us(t[:, 1])))
    main = t[t[:, 1] > 0]
    print("  pass duration", st((main[:, 1] - main[:, 6]) / 100.0))
    kk = t[t[:, 4] > 0]
    for row in kk:
        print(f"  k x k job: entry {(row[0]-t0)/100:.2f}  wait over {(row[6]-t0)/100:.2f}  done {(row[4]-t0)/100:.2f}")
e.close()
