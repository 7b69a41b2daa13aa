#!/usr/bin/env python3
"""Diagnostic: what ONE rank of a view-sharded run executes per sweep, without the exchange (one process, one GPU: the
rank owns view 0 of V views).  Two layouts:
    replicated   F / G / S chains of every view on every rank (blocks of the other views keep their initial contents)
    --sliced     row-sliced F / G chains (slice 0 of V), replicated S chain; the exchange is stood in for by device
                 copies of the own chunks into every peer's slot, outside the timed kernels
The numbers mean nothing -- only the kernel times do.  Prints the wall time per sweep of the library's launches and the
per-kind kernel times (HIP events attached to the dispatches, resnmtf_kernel_timings).  Under the kernel trace:
    rocprofv3 --kernel-trace --stats ... -- python3 tools/time_replica_updates.py 8 50000 8000 64 [sweeps] [--sliced]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from resnmtf_amd import _lib, sharded  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
sliced = "--sliced" in sys.argv
V, n, m, k = (int(x) for x in args[:4])
sweeps = int(args[4]) if len(args) > 4 else 30
prob = sharded.local_problem(V, (n, m), k, phi=200.0, xi=200.0, psi=200.0, owned=[0])
stream = torch.cuda.Stream()


def make(timed):
    opts = dict(slice_chains=True, slice_index=0, slice_count=V) if sliced else {}
    if os.environ.get("RESNMTF_NO_F_CHAIN") == "1":      # A/B: one launch per view instead of the fused chain launches
        opts["no_f_chain"] = True
    e = sharded.make_hip_engine(prob, [v == 0 for v in range(V)], 0, stream.cuda_stream, replicate_f=True, replicate_gs=True,
                                time_kernels=timed, **opts)
    e.reserve_sweeps(sweeps + 8)
    e.prepare()
    return e


def fan_out(eng, kind):
    """stand-in for an all-to-all: every peer's chunk = a copy of the chunk this rank keeps for itself"""
    send, recv = eng.factor_tensor(0, kind + "_SEND"), eng.factor_tensor(0, kind + "_RECV")
    chunk = send.numel() // V
    recv.view(V, chunk).copy_(send[:chunk].expand(V, chunk))


if sliced:
    steps = ((_lib.PHASE_SLICE_F, "FNEW"), (_lib.PHASE_SLICE_XTF, "T"), (_lib.PHASE_SLICE_G, "GNEW"), (_lib.PHASE_SLICE_XG, "U"),
             (_lib.PHASE_S_ALL, None))
else:
    steps = tuple((p, None) for p in (_lib.PHASE_F_ALL, _lib.PHASE_XTF, _lib.PHASE_G_ALL, _lib.PHASE_XG, _lib.PHASE_S_ALL))

for timed in (False, True):
    eng = make(timed)
    with torch.cuda.stream(stream):
        if sliced:
            fan_out(eng, "U")
            s0 = eng.factor_tensor(0, "SBLOCK")
        for t in range(sweeps + 3):
            if t == 3:
                eng.synchronize()
                if timed:
                    eng.kernel_timings(reset=True)
                t0 = time.perf_counter()
            for p, kind in steps:
                if sliced and p == _lib.PHASE_S_ALL and timed:
                    for v in range(1, V):
                        eng.factor_tensor(v, "SBLOCK").copy_(s0)
                eng.phase(0, p, t)
                if kind and timed:                     # (the untimed pass measures the launches alone: stale peers' chunks)
                    fan_out(eng, kind)
        eng.synchronize()
    dt = (time.perf_counter() - t0) / sweeps
    if not timed:
        print(f"V={V} {n}x{m} k={k} {'sliced' if sliced else 'replicated'}: {dt*1e6:.1f} us per sweep of one rank's launches (no exchange)")
    else:
        kt = eng.kernel_timings()
        total = sum(ms for ms, _ in kt.values()) / sweeps * 1e3
        print("  per sweep by kind (events): " + ", ".join(f"{name} {ms / sweeps * 1e3:.1f} us" for name, (ms, cnt) in kt.items() if cnt) +
              f"  | sum {total:.1f} us")
    eng._views.clear() if hasattr(eng, "_views") else None
    eng.close()
