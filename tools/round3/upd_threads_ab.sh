#!/bin/bash
# threads per workgroup of the k <= 16 factor update (RESNMTF_UPD16_THREADS): variant libraries against the product (512)
for i in 1 2; do
  echo "base    $(timeout -k 10 100 python3 tools/bench_configs.py c2 2>&1 | tail -1 | cut -c1-150)"
  for L in tools/micro/libresnmtf_upd*.so; do
    echo "$(basename $L .so | sed s/libresnmtf_//) $(timeout -k 10 100 python3 tools/run_with_lib.py $L tools/bench_configs.py c2 2>&1 | tail -1 | cut -c1-150)"
  done
done
