#!/usr/bin/env python3
"""Diagnostic: what ONE rank of a view-sharded run with replicated F / G / S chains executes per sweep, without the
exchange (one process, one GPU: the rank owns view 0 of V views; the blocks of the other views keep their initial
contents, so the numbers mean nothing -- only the kernel times do).  Run under the kernel trace:
    rocprofv3 --kernel-trace --stats ... -- python3 tools/time_replica_updates.py 8 50000 8000 64 [sweeps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from resnmtf_amd import _lib, sharded  # noqa: E402

V, n, m, k = (int(x) for x in sys.argv[1:5])
sweeps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
prob = sharded.local_problem(V, (n, m), k, phi=200.0, xi=200.0, psi=200.0, owned=[0])
stream = torch.cuda.Stream()
eng = sharded.make_hip_engine(prob, [v == 0 for v in range(V)], 0, stream.cuda_stream, replicate_f=True, replicate_gs=True)
eng.reserve_sweeps(sweeps + 8)
eng.prepare()
ph = (_lib.PHASE_F_ALL, _lib.PHASE_XTF, _lib.PHASE_G_ALL, _lib.PHASE_XG, _lib.PHASE_S_ALL)
for t in range(3):
    for p in ph:
        eng.phase(0, p, t)
eng.synchronize()
t0 = time.perf_counter()
for t in range(3, 3 + sweeps):
    for p in ph:
        eng.phase(0, p, t)
eng.synchronize()
dt = (time.perf_counter() - t0) / sweeps
print(f"V={V} {n}x{m} k={k}: {dt*1e6:.1f} us per sweep of one rank's kernels (no exchange)")
eng.close()
