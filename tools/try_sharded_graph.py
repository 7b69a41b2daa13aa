#!/usr/bin/env python3
"""Experiment (one rank on RCCL): K sweeps of the view-sharded loop -- library kernels + the in-place all-gather --
captured in ONE torch.cuda.CUDAGraph and replayed, against the eager loop.
    RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python tools/try_sharded_graph.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from resnmtf_amd import sharded

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, m, k = 10000, 2000, 16
prob = sharded.local_problem(1, (n, m), k, phi=200.0, owned=[0])
drv = sharded.ShardedSweep(prob, [0], 0, 1, device_index=0, replicate_f="force")
drv.reserve(5000)
drv.run(50)
torch.cuda.synchronize()
t0 = time.perf_counter(); drv.run(500); torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 500
print(f"eager      : {te * 1e6:7.2f} us per sweep", flush=True)
K = 25
g = torch.cuda.CUDAGraph()
st = drv._tstream
try:
    with torch.cuda.graph(g, stream=st):
        drv._run(K)
    torch.cuda.synchronize()
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / (20 * K)
    print(f"graph of {K:2d}: {tg * 1e6:7.2f} us per sweep", flush=True)
except Exception as exc:
    print("capture failed:", repr(exc)[:500], flush=True)
dist.destroy_process_group()
