#!/bin/bash
# round 3: fused update launches -- parity (bitwise vs no_fuse) and the c2 rates with and without the fusion
set -o pipefail
out=gpurun_out/r3d; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or graph_and_eager or consecutive or resume or convergence" > $out/t_fused.log 2>&1
tail -4 $out/t_fused.log
summ='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], r["xg_avg_us"], r["xtf_avg_us"])'
for i in 1 2; do timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$out/bench.err | python3 -c "$summ"; done
timeout -k 10 120 python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>>$out/bench.err | python3 -c "$summ"
echo "--- no_fuse"
RESNMTF_FUSE_UPDATES=1 timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$out/bench.err | python3 -c "$summ"
RESNMTF_FUSE_UPDATES=1 timeout -k 10 120 python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>>$out/bench.err | python3 -c "$summ"
echo "--- fused without prefetch"
RESNMTF_FUSE_UPDATES=2 timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$out/bench.err | python3 -c "$summ"
RESNMTF_FUSE_UPDATES=2 timeout -k 10 120 python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>>$out/bench.err | python3 -c "$summ"
