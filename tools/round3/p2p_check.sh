#!/bin/bash
out=gpurun_out/r3j; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_sharded.py -x -q -m gpu -k "peer_stores" > $out/tests_p2p.log 2>&1; tail -3 $out/tests_p2p.log
(RESNMTF_P2P=1 RESNMTF_BENCH_DEVICE=0 RESNMTF_BENCH_BACKEND=gloo RESNMTF_BENCH_SHAPE=6000,2000,64 RESNMTF_SLICE_CHAINS=1 timeout -k 10 300 python3 bench.py --gpus 4 --steps 5 --warmup 2 > $out/bench_gpus4_p2p.json 2> $out/bench_gpus4_p2p.err; echo "rc=$?" >> $out/bench_gpus4_p2p.err); tail -2 $out/bench_gpus4_p2p.err; python3 -c "
import json; d=json.loads(open('$out/bench_gpus4_p2p.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['workload'][-220:]); print(d['roofline']['rank0_kernel_us_per_sweep'], d['cpu_baseline'] is not None)"
timeout -k 10 200 python3 tools/soak.py 40 > $out/soak40.log 2>&1; tail -3 $out/soak40.log
