#!/bin/bash
out=gpurun_out/r3e; mkdir -p $out
summ='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], r["xg_avg_us"], r["xtf_avg_us"])'
for nap in 0 1 2 4 8; do
for nf in 0 2; do
echo "nap=$nap no_fuse=$nf"
RESNMTF_FUSE_NAP=$nap RESNMTF_FUSE_UPDATES=$nf timeout -k 10 120 python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>>$out/bench.err | python3 -c "$summ"
done; done
echo "unfused"; RESNMTF_FUSE_UPDATES=1 timeout -k 10 120 python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>>$out/bench.err | python3 -c "$summ"
