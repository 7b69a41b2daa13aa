#!/bin/bash
# Ranks sharing the one GPU of the box (gloo for the set-up collectives): the sweep with the peer-store exchange against the
# same layout with its collectives staged through gloo -- what the exchange machinery itself costs when no wire is involved.
# The ranks' kernels share the GPU, so ms_per_step is about N x one rank's kernels + the hand-off latencies.
set -u
out=gpurun_out/r3c; mkdir -p $out
export RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo
for n in 2 4; do
  for p2p in 1 0; do
    RESNMTF_P2P=$p2p timeout -k 10 300 python bench.py --gpus $n --steps 200 --warmup 20 --no-cpu-baseline > $out/rehearse_n${n}_p2p${p2p}.json 2> $out/rehearse_n${n}_p2p${p2p}.err || echo "n=$n p2p=$p2p failed"
    python - <<PY
import json
try:
    d = json.loads(open("$out/rehearse_n${n}_p2p${p2p}.json").read().strip().splitlines()[-1])
    print("n=$n p2p=$p2p ms_per_step", d["ms_per_step"], "value", d["value"], "|", d["config"]["workload"][-130:], "|", (d.get("roofline") or {}).get("rank0_kernel_us_per_sweep_total"))
except Exception as e:
    print("n=$n p2p=$p2p: no line", e)
PY
  done
done
