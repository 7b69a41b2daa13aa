#!/bin/bash
# more peer-store rehearsals on the one GPU: c4 with the replicated chains (2 exchanges per sweep) and N x c2 (1 exchange)
set -u
out=gpurun_out/r3c; mkdir -p $out
export RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo RESNMTF_P2P=1
run() {  # name, gpus, extra env...
  local name=$1 n=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --gpus $n --steps 200 --warmup 20 --no-cpu-baseline > $out/rehearse_$name.json 2> $out/rehearse_$name.err || echo "$name failed"
  python - <<PY
import json
try:
    d = json.loads(open("$out/rehearse_$name.json").read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("$name ms_per_step", d["ms_per_step"], "value", d["value"], "| rank0 kernels/sweep", r.get("rank0_kernel_us_per_sweep_total"), r.get("rank0_kernel_us_per_sweep"), "single", r.get("single_view_updates_per_s"))
except Exception as e:
    print("$name: no line", e)
PY
}
run c4_block 4 RESNMTF_SLICE_CHAINS=0
run c2x4_f 4 RESNMTF_BENCH_SAME_SHAPE=1
run c2x3_f 3 RESNMTF_BENCH_SAME_SHAPE=1
run c4_sliced3 3 RESNMTF_BENCH_SHAPE=20000,4000,32
run c4_sliced2 2 RESNMTF_BENCH_SHAPE=20000,4000,32
