#!/usr/bin/env python3
"""Diagnostic: in-kernel timeline of slice_chain_kernel (-DRESNMTF_STAMPS build; 100 MHz ticks): one rank's sliced F and G
chain launches of a V-view problem (the rank owns view 0, exchange stood in for by copies).
Columns (F chain 0.., G chain 8..): +0 entry, +1 rows in LDS, +2 first pair's products stored, +3 first pair walked, +4 end.
    python tools/stamps_slice_chain.py 8 50000 8000 64"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

V, n, m, k = (int(x) for x in sys.argv[1:5])
so = os.path.join(ROOT, "resnmtf_amd", "libresnmtf_hip_stamps.so")
if not os.path.exists(so) or os.environ.get("STAMPS_REBUILD") == "1":
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRESNMTF_STAMPS",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "resnmtf_amd", "csrc"), "-o", so,
                    os.path.join(ROOT, "resnmtf_amd", "csrc", "resnmtf_hip.hip")], check=True)
from resnmtf_amd import _lib  # noqa: E402
_lib.LIB_PATH = so
import torch  # noqa: E402
from resnmtf_amd import sharded  # noqa: E402

lib = _lib.load()
lib.resnmtf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
prob = sharded.local_problem(V, (n, m), k, phi=200.0, xi=200.0, psi=200.0, owned=[0])
stream = torch.cuda.Stream()
eng = sharded.make_hip_engine(prob, [v == 0 for v in range(V)], 0, stream.cuda_stream, replicate_f=True, replicate_gs=True,
                              slice_chains=True, slice_index=0, slice_count=V)
eng.reserve_sweeps(16); eng.prepare()


def fan_out(kind):
    send, recv = eng.factor_tensor(0, kind + "_SEND"), eng.factor_tensor(0, kind + "_RECV")
    chunk = send.numel() // V
    recv.view(V, chunk).copy_(send[:chunk].expand(V, chunk))


steps = ((_lib.PHASE_SLICE_F, "FNEW"), (_lib.PHASE_SLICE_XTF, "T"), (_lib.PHASE_SLICE_G, "GNEW"), (_lib.PHASE_SLICE_XG, "U"), (_lib.PHASE_S_ALL, None))
buf = torch.zeros((16384, 16), dtype=torch.int64, device="cuda")
with torch.cuda.stream(stream):
    fan_out("U")
    for t in range(3):
        if t == 2:
            eng.synchronize(); torch.cuda.synchronize()
            assert lib.resnmtf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
            assert lib.resnmtf_debug_set_stamp_select(2) == 0
        for p, kind in steps:
            eng.phase(0, p, t)
            if kind:
                fan_out(kind)
    eng.synchronize()
lib.resnmtf_debug_set_stamp_buffer(None)
tall = buf.cpu().numpy().astype(np.int64)
st = lambda x: f"min {x.min():6.2f}  p10 {np.percentile(x,10):6.2f}  med {np.median(x):6.2f}  p90 {np.percentile(x,90):6.2f}  max {x.max():6.2f}" if len(x) else "-"
for name, base in (("F chain", 0), ("G chain", 8)):
    t = tall[:, base:base + 8]
    t = t[t[:, 0] > 0]
    if not len(t):
        continue
    t0 = t[:, 0].min()
    print(f"== sliced {name} ({V} views {n}x{m} k={k}): {len(t)} workgroups")
    print("  entry              ", st((t[:, 0] - t0) / 100.0))
    for col, label in ((1, "rows in LDS        "), (2, "pair 0 products    "), (3, "pair 0 walked      "), (4, "end                ")):
        print(f"  {label}", st((t[:, col] - t[:, 0]) / 100.0), " (since entry)")
    print("  end since launch   ", st((t[:, 4] - t0) / 100.0))
# s_chain_kernel (workgroup w = view w): columns 5 entry, 6 walk done, 7 two k x k products done, 13 error / lambda / mu done, 14 end
t = tall[:V]
if (t[:, 5] > 0).any():
    t0 = t[t[:, 5] > 0][:, 5].min()
    print(f"== s_chain_kernel ({V} views, k={k}): per workgroup (view): walk done | products done | error done | end   [us since the first entry]")
    for w in range(V):
        print(f"  view {w}: entry {(t[w,5]-t0)/100:6.2f}  walk {(t[w,6]-t0)/100:6.2f}  products {(t[w,7]-t0)/100:6.2f}  error {(t[w,13]-t0)/100:6.2f}  end {(t[w,14]-t0)/100:6.2f}")
eng.close()
