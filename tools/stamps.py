#!/usr/bin/env python3
"""Diagnostic: timeline of the workgroups of the two streaming-pass launches.  Builds
libresnmtf_hip_stamps.so (-DRESNMTF_STAMPS), runs a few sweeps of a config eagerly, then ONE
PHASE_G (Xt.F launch, G update, X.G launch) with the stamp buffer attached.  Stamps are 100 MHz
wall-clock ticks (10 ns); columns 0-4 belong to the X.G launch, 8-12 to the Xt.F launch.
    python tools/stamps.py [config] [kk_mode]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
kk_mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
so = os.path.join(ROOT, "resnmtf_amd", "libresnmtf_hip_stamps.so")
if not os.path.exists(so) or os.environ.get("STAMPS_REBUILD") == "1":
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRESNMTF_STAMPS",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "resnmtf_amd", "csrc"), "-o", so,
                    os.path.join(ROOT, "resnmtf_amd", "csrc", "resnmtf_hip.hip")], check=True)
from resnmtf_amd import _lib, synth  # noqa: E402
_lib.LIB_PATH = so
from resnmtf_amd.engine import Engine  # noqa: E402
import torch  # noqa: E402  (device memory for the stamp buffer)

lib = _lib.load()
lib.resnmtf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
if "x" in cfg:
    n_, m_, k_ = (int(t) for t in cfg.split("x")); prob = synth.make_problem([(n_, m_)], k_)
else:
    prob = synth.config(cfg)
n, m = prob.data[0].shape
e = Engine([n], [m], [prob.k], use_graph=False, kk_mode=kk_mode)
e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
e.run(5)
e.reserve_sweeps(64); e.prepare()
for sw in range(3):
    e.phase(0, _lib.PHASE_F, sw); e.phase(0, _lib.PHASE_G, sw)
e.synchronize()
nblk = 16384
buf = torch.zeros((nblk, 16), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
e.phase(0, _lib.PHASE_F, 3); e.synchronize()
assert lib.resnmtf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
assert lib.resnmtf_debug_set_stamp_select(1) == 0      # the G update between the two passes shares their columns
e.phase(0, _lib.PHASE_G, 3); e.synchronize()
lib.resnmtf_debug_set_stamp_buffer(None)
tall = buf.cpu().numpy().astype(np.int64)
for name, base in (("Xt.F", 8), ("X.G", 0)):
    t = tall[:, base:base + 8]
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = lambda col: (col[col > 0] - t0) / 100.0
    st = lambda x: f"min {x.min():6.2f}  p10 {np.percentile(x,10):6.2f}  med {np.median(x):6.2f}  p90 {np.percentile(x,90):6.2f}  max {x.max():6.2f}"
    main = t[(t[:, 1] > 0) & (t[:, 2] == 0)]
    print(f"== {name} launch ({cfg}, kk_mode {kk_mode}): {len(t)} blocks stamped, {len(main)} main")
    print("  main entry    ", st(us(main[:, 0])))
    print("  main end      ", st(us(main[:, 1])))
    print("  main duration ", st((main[:, 1] - main[:, 0]) / 100.0))
    aux = t[t[:, 2] > 0]
    if len(aux):
        print("  aux entry     ", st(us(aux[:, 0])))
        print("  aux body end  ", st(us(aux[:, 1])))
        print("  aux ticket    ", st(us(aux[:, 2])))
    kk = t[t[:, 4] > 0]
    for row in kk:
        extra = "  ".join(f"s{c}={(row[c]-t0)/100:.2f}" for c in (5, 6, 7) if row[c] > 0)
        print(f"  k x k job: entry {(row[0]-t0)/100:.2f}  acquire done {((row[3]-t0)/100 if row[3] else float('nan')):.2f}  done {(row[4]-t0)/100:.2f}  {extra}")
    # concurrency profile: how many main workgroups are alive at each microsecond
    ent, end = us(main[:, 0]), us(main[:, 1])
    prof = [int(((ent <= x) & (end > x)).sum()) for x in np.arange(0, end.max() + 1, 1.0)]
    step = max(1, len(prof) // 60)
    print(f"  alive main WGs every {step} us:", prof[::step])
    # structure of the spread: mean end time by XCD (dispatch order round-robins workgroups over the
    # 8 XCDs), by tile and by split
    ids = np.nonzero((tall[:, base] > 0) & (tall[:, base + 1] > 0) & (tall[:, base + 2] == 0))[0]
    endt = (tall[ids, base + 1] - t0) / 100.0
    nt = (m + 63) // 64 if name == "Xt.F" else (n + 63) // 64
    if prob.k > 16:                      # wide form: tile GROUPS of 8 tiles, or of 4 when that does not divide the grid
        g8 = (nt + 7) // 8
        nt = g8 if (len(ids) % g8 == 0 and len(ids) // g8 <= 16) else (nt + 3) // 4
    off = ids.min()                      # first main block
    tile, split = (ids - off) % nt, (ids - off) // nt
    print("  mean end by XCD (block % 8):", np.round([endt[ids % 8 == x].mean() for x in range(8)], 1))
    print("  mean end by split:", np.round([endt[split == x].mean() for x in range(split.max() + 1)], 1))
    tl = [endt[tile == x].mean() for x in range(nt)]
    print("  mean end by tile (first 32):", np.round(tl[:32], 1))
e.close()
