#!/bin/bash
# loads in flight per trip of the k <= 16 pass (UNROLL) under the max-ILP scheduler: variant libraries (tools/build_variant.py)
for i in 1 2; do
  echo "base     $(timeout -k 10 100 python3 tools/bench_configs.py c2 2>&1 | tail -1 | cut -c1-150)"
  for L in tools/micro/libresnmtf_unroll*.so; do
    echo "$(basename $L .so | sed s/libresnmtf_//) $(timeout -k 10 100 python3 tools/run_with_lib.py $L tools/bench_configs.py c2 2>&1 | tail -1 | cut -c1-150)"
  done
done
