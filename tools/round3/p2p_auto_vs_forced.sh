#!/bin/bash
# is the timed driver slower when the probe / cross-check drivers ran before it in the same processes?
set -u
out=gpurun_out/r3c; mkdir -p $out
export RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo RESNMTF_BENCH_SHAPE=20000,4000,32
for rep in 1 2; do
  for mode in auto 1; do
    RESNMTF_P2P=$mode timeout -k 10 400 python bench.py --gpus 3 --steps 100 --warmup 10 --no-cpu-baseline > $out/avf_$mode$rep.json 2> $out/avf_$mode$rep.err || echo "$mode failed"
    python - <<PY
import json
try:
    d = json.loads(open("$out/avf_$mode$rep.json").read().strip().splitlines()[-1])
    print("mode=$mode rep=$rep ms_per_step", d["ms_per_step"], "value", d["value"], "|", d["config"]["exchange"][:40])
except Exception as e:
    print("mode=$mode: no line", e)
PY
  done
done
