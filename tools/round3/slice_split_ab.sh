#!/bin/bash
# sliced chains of one rank of c5 x 8 / c4 x 4: the 16-row form in two launches (products | walk) against the fused kernel,
# and for the F slice against the 32-row wide chain
set -u
out=gpurun_out/r3c; mkdir -p $out
for cfg in "8 50000 8000 64" "4 20000 4000 32"; do
  echo "== $cfg: default (G: split; F: by size)"; timeout -k 10 200 python tools/time_replica_updates.py $cfg --sliced 2>&1 | tail -2
  echo "== $cfg: RESNMTF_SLICE_FUSED=1"; RESNMTF_SLICE_FUSED=1 timeout -k 10 200 python tools/time_replica_updates.py $cfg --sliced 2>&1 | tail -2
  echo "== $cfg: RESNMTF_SLICE_WIDE=0 (split form for both slices)"; RESNMTF_SLICE_WIDE=0 timeout -k 10 200 python tools/time_replica_updates.py $cfg --sliced 2>&1 | tail -2
done
