#!/bin/bash
# one rank, the sharded driver with the peer-store exchange (degenerate: own arrivals only): sweeps replayed from a graph (wait
# kernels) / plain launches with stream waits / the collective form -- what the host-side launch sequence of a sweep costs
set -u
out=gpurun_out/r3c; mkdir -p $out
run() {
  local name=$1; shift
  env RESNMTF_FORCE_SHARDED=1 RESNMTF_FORCE_REPLICATE=1 "$@" timeout -k 10 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline > $out/one_rank_$name.json 2> $out/one_rank_$name.err || echo "$name failed"
  python - <<PY
import json
try:
    d = json.loads(open("$out/one_rank_$name.json").read().strip().splitlines()[-1])
    print("$name ms_per_step", d["ms_per_step"], "value", d["value"], "|", d["config"]["exchange"], "| replay", d["config"]["sweeps_per_graph_replay"])
except Exception as e:
    print("$name: no line", e)
PY
}
run p2p_graph RESNMTF_P2P=1 RESNMTF_P2P_GRAPH=1
run p2p_eager RESNMTF_P2P=1
run collective RESNMTF_P2P=0
