"""One rank of the world-size-2 test of resnmtf_amd.batched.run_jobs (gloo, CPU, stand-in runner)."""
import argparse
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

from resnmtf_amd import batched  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rank", type=int); ap.add_argument("--world", type=int); ap.add_argument("--port", type=int)
ap.add_argument("--out")
a = ap.parse_args()
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=a.rank, world_size=a.world)
x = np.arange(12.0).reshape(3, 4) + 1.0
jobs = batched.k_sweep_jobs([x], 2, 6)


def runner(job):       # stand-in for the GPU factorisation: records who ran what
    return {"tag": job.tag, "k": job.k_val, "rank": dist.get_rank(), "sum": float(sum(d.sum() for d in job.data))}


res = batched.run_jobs(jobs, runner=runner)
if a.rank == 0:
    pickle.dump(res, open(a.out, "wb"))
dist.barrier()
dist.destroy_process_group()
