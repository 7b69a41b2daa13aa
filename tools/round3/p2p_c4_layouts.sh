#!/bin/bash
# c4-shaped views, 3 ranks sharing the one GPU, peer-store exchange: row-sliced chains against replicated chains with stored blocks
set -u
out=gpurun_out/r3c; mkdir -p $out
export RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo RESNMTF_BENCH_SHAPE=20000,4000,32 RESNMTF_P2P=1
for rep in 1 2; do
  for sl in 1 0; do
    RESNMTF_SLICE_CHAINS=$sl timeout -k 10 300 python bench.py --gpus 3 --steps 200 --warmup 20 --no-cpu-baseline > $out/c4lay_$sl$rep.json 2> $out/c4lay_$sl$rep.err || echo "failed"
    python - <<PY
import json
try:
    d = json.loads(open("$out/c4lay_$sl$rep.json").read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("sliced=$sl rep=$rep ms_per_step", d["ms_per_step"], "value", d["value"], "| rank0 kernels/sweep", r.get("rank0_kernel_us_per_sweep_total"), "single", r.get("single_view_updates_per_s"))
except Exception as e:
    print("sliced=$sl: no line", e)
PY
  done
done
