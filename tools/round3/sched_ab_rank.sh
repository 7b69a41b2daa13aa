#!/bin/bash
# the chain / S-chain / pack kernels of one c5 x 8 rank under other scheduling strategies of the main unit
for L in "" tools/micro/libresnmtf_*.so; do
  if [ -z "$L" ]; then echo "== base"; timeout -k 10 200 python3 tools/time_replica_updates.py 8 50000 8000 64 --sliced 2>&1 | tail -2
  else echo "== $(basename $L)"; timeout -k 10 200 python3 tools/run_with_lib.py $L tools/time_replica_updates.py 8 50000 8000 64 --sliced 2>&1 | tail -2; fi
done
