#!/bin/bash
# compile-time A/B: LLVM's AMDGPU scheduling strategies for the whole library (tools/micro/libresnmtf_<tag>.so built with
# -mllvm -amdgpu-sched-strategy=max-ilp | max-memory-clause, -amdgpu-schedule-metric-bias=0), rates of the one-GPU configs
set -u
out=gpurun_out/r3c; mkdir -p $out
for cfg in c2 c4v1 c5v1; do
  echo "== $cfg base"; timeout -k 10 200 python3 tools/bench_configs.py $cfg 2>&1 | tail -1
  for L in tools/micro/libresnmtf_*.so; do
    echo "== $cfg $(basename $L)"; timeout -k 10 200 python3 tools/run_with_lib.py $L tools/bench_configs.py $cfg 2>&1 | tail -1
  done
done
