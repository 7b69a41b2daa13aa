#!/bin/bash
# bench.py's default exchange decision (self-test + bitwise cross-check), ranks sharing the one GPU over gloo
set -u
out=gpurun_out/r3c; mkdir -p $out
export RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo
for n in 2 3; do
  extra=""
  [ $n = 3 ] && export RESNMTF_BENCH_SHAPE=20000,4000,32
  timeout -k 10 400 python bench.py --gpus $n --steps 100 --warmup 10 > $out/auto_n$n.json 2> $out/auto_n$n.err || echo "n=$n failed"
  python - <<PY
import json
try:
    d = json.loads(open("$out/auto_n$n.json").read().strip().splitlines()[-1])
    print("n=$n ms_per_step", d["ms_per_step"], "value", d["value"], "| exchange:", d["config"]["exchange"], "| cpu:", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("n=$n: no line", e)
PY
done
