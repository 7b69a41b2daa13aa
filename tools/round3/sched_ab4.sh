#!/bin/bash
# main-unit scheduler options (trackers, relaxed occupancy, rescheduling stages off) on the k > 16 configurations, two passes
for i in 1 2; do
  for cfg in c4v1 c5v1; do
    echo "base     $(timeout -k 10 100 python3 tools/bench_configs.py $cfg 2>&1 | tail -1 | cut -c1-140)"
    for L in tools/micro/libresnmtf_*.so; do
      echo "$(basename $L .so | sed s/libresnmtf_//) $(timeout -k 10 100 python3 tools/run_with_lib.py $L tools/bench_configs.py $cfg 2>&1 | tail -1 | cut -c1-140)"
    done
  done
done
