#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/tests_all.log 2>&1; tail -4 $out/tests_all.log
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > $out/bench_c2_20.json 2> $out/bench_c2_20.err; python3 -c "import json; d=json.loads(open('$out/bench_c2_20.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
